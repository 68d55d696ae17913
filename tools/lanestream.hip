// lanestream.hip — can the vector memory path feed "one lane = one read" directly?  Every lane streams its OWN contiguous
// read with 16-byte loads (64 distinct lines per wave-instruction), CH bytes per round, no LDS transposition.
// usage: lanestream <n_streams> <len> ; prints GB/s for CH = 64/128/256 at 8/12/16 waves per CU (persistent blocks).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int CH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_lane(const uint8_t* plane, uint64_t n_streams, uint64_t len, unsigned* ticket, uint32_t* out) {
    constexpr int N = CH / 16;
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (;;) {
        unsigned g = 0;
        if (lane == 0) g = atomicAdd(ticket, 1u);
        g = __builtin_amdgcn_readfirstlane(g);
        if ((uint64_t)g * 64 >= n_streams) break;
        const uint4* p = reinterpret_cast<const uint4*>(plane + ((uint64_t)g * 64 + lane) * len);
        const uint64_t rounds = len / CH;
        uint4 cur[N], nxt[N];
#pragma unroll
        for (int i = 0; i < N; ++i) cur[i] = p[i];
        for (uint64_t t = 1; t <= rounds; ++t) {
            if (t < rounds) {
#pragma unroll
                for (int i = 0; i < N; ++i) nxt[i] = p[t * N + i];
            }
#pragma unroll
            for (int i = 0; i < N; ++i) acc ^= cur[i].x ^ cur[i].y ^ cur[i].z ^ cur[i].w;
#pragma unroll
            for (int i = 0; i < N; ++i) cur[i] = nxt[i];
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}

template <int CH, int WAVES>
void run(const uint8_t* d, uint64_t n, uint64_t len, unsigned* ticket, uint32_t* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(ticket, 0, 4);
    hipLaunchKernelGGL((k_lane<CH, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, n, len, ticket, out);
    hipDeviceSynchronize();
    hipMemset(ticket, 0, 4);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_lane<CH, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, n, len, ticket, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("lane-private streams  chunk %3d B  waves/CU %2d: %8.3f ms  %8.1f GB/s\n", CH, WAVES, ms, (double)n * len / ms / 1e6);
}

int main(int argc, char** argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000;
    uint64_t len = argc > 2 ? strtoull(argv[2], 0, 10) : 10240;
    uint8_t* d; uint32_t* out; unsigned* ticket;
    hipMalloc(&d, n * len + 4096); hipMalloc(&out, 64); hipMalloc(&ticket, 64);
    hipMemset(d, 1, n * len);
    run<64, 8>(d, n, len, ticket, out);
    run<128, 8>(d, n, len, ticket, out);
    run<256, 8>(d, n, len, ticket, out);
    run<64, 12>(d, n, len, ticket, out);
    run<128, 12>(d, n, len, ticket, out);
    run<256, 12>(d, n, len, ticket, out);
    run<64, 16>(d, n, len, ticket, out);
    run<128, 16>(d, n, len, ticket, out);
    return 0;
}
