// randbench.hip — random 4-byte read rate vs working-set size on MI355X (L2 4 MiB/XCD, Infinity Cache 256 MiB, HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k(const uint32_t* buf, uint64_t mask_words, int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = buf[(x >> 20) & mask_words];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

int main() {
    const uint64_t max_bytes = 4ull << 30;
    uint32_t *buf, *out;
    hipMalloc(&buf, max_bytes); hipMalloc(&out, 64);
    hipMemset(buf, 1, max_bytes);
    for (uint64_t mb : {1ull, 4ull, 16ull, 32ull, 64ull, 128ull, 256ull, 512ull, 1024ull, 4096ull}) {
        const uint64_t words = mb * (1 << 20) / 4;
        const int iters = 200;
        dim3 grid(256 * 8), block(256);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, grid, block, 0, 0, buf, words - 1, 20, out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k, grid, block, 0, 0, buf, words - 1, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double n = (double)grid.x * 256 * iters * 16;
        printf("working set %5llu MiB: %7.3f ms  %7.1f G random reads/s\n", (unsigned long long)mb, ms, n / ms / 1e6);
    }
    return 0;
}
