// membench.hip — how fast can HBM feed the "64 reads per wavefront" access pattern?
// Each wave owns 64 streams (reads) of LEN bytes; per round it fetches CHUNK contiguous bytes of every
// stream with 16-byte loads (lane -> (stream, piece) map, like the scoring kernel) and XORs them.
// usage: membench <n_streams> <len> <waves_per_block> ; prints GB/s for CHUNK = 16..512
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int CHUNK>
__global__ void k_stream(const uint8_t* plane, uint64_t n_streams, uint64_t len, uint64_t stride, uint32_t* out) {
    constexpr int PPR = CHUNK / 16;                 // pieces per stream per round
    constexpr int LOADS = PPR;                      // loads per lane per round (64 streams * PPR / 64 lanes)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (w * 64 >= n_streams) return;
    uint32_t acc = 0;
    const uint64_t rounds = len / CHUNK;
    for (uint64_t t = 0; t < rounds; ++t) {
        uint4 v[LOADS];
#pragma unroll
        for (int m = 0; m < LOADS; ++m) {
            const int i = m * 64 + lane;
            const int r = i / PPR, k = i % PPR;
            const uint64_t s = w * 64 + r;
            v[m] = *reinterpret_cast<const uint4*>(plane + s * stride + t * CHUNK + k * 16);
        }
#pragma unroll
        for (int m = 0; m < LOADS; ++m) acc ^= v[m].x ^ v[m].y ^ v[m].z ^ v[m].w;
    }
    if (acc == 0x12345678) out[0] = acc;
}

template <int CHUNK>
void run(const uint8_t* d, uint64_t n, uint64_t len, uint64_t stride, int wpb, uint32_t* out) {
    const uint64_t waves = n / 64;
    dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(wpb * 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_stream<CHUNK>, grid, block, 0, 0, d, n, len, stride, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_stream<CHUNK>, grid, block, 0, 0, d, n, len, stride, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("chunk %4d B  wpb %d: %8.3f ms  %8.1f GB/s\n", CHUNK, wpb, ms, (double)n * (len / CHUNK * CHUNK) / ms / 1e6);
}

int main(int argc, char** argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000;
    uint64_t len = argc > 2 ? strtoull(argv[2], 0, 10) : 10240;
    int wpb = argc > 3 ? atoi(argv[3]) : 7;
    uint64_t stride = len;
    uint8_t* d; uint32_t* out;
    hipMalloc(&d, n * stride + 4096); hipMalloc(&out, 64);
    hipMemset(d, 1, n * stride);
    run<16>(d, n, len, stride, wpb, out);
    run<32>(d, n, len, stride, wpb, out);
    run<64>(d, n, len, stride, wpb, out);
    run<128>(d, n, len, stride, wpb, out);
    run<256>(d, n, len, stride, wpb, out);
    run<512>(d, n, len, stride, wpb, out);
    return 0;
}
