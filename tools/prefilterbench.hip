// prefilterbench.hip — would a small L2-resident prefilter in front of the 512 MiB exact 16-mer bitmap pay?
// Every thread does N "queries": a random 4-byte read of a SMALL table (the prefilter, 1/2/4 MiB) and, for a fraction
// FAR/16 of the queries, a random 4-byte read of the 512 MiB table (the queries the prefilter cannot answer).
// Reported: queries/s.  The far reads are issued plain or non-temporal (do they evict the prefilter from L2?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int FAR, bool NT>
__global__ void __launch_bounds__(256) k(const uint32_t* small, uint64_t small_mask, const uint32_t* big, uint64_t big_mask,
                                         int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16], w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = small_mask ? small[(uint32_t)((((x >> 20) & 0xffffffffull) * small_mask) >> 32)] : 0u;  // small_mask = number of words here
            w[j] = 0;
            if (j < FAR) {
                const uint32_t* p = big + ((x >> 13) & big_mask);
                w[j] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j] ^ w[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

template <int FAR, bool NT>
void run(const uint32_t* small, uint64_t small_mib, const uint32_t* big, uint32_t* out) {
    const uint64_t sw = small_mib ? small_mib * (1 << 18) / 4 : 1, bw = 512ull * (1 << 20) / 4;  // small_mib in quarter MiB
    const int iters = 100;
    dim3 grid(256 * 8), block(256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<FAR, NT>), grid, block, 0, 0, small, small_mib ? sw : 0, big, bw - 1, 10, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<FAR, NT>), grid, block, 0, 0, small, small_mib ? sw : 0, big, bw - 1, iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double n = (double)grid.x * 256 * iters * 16;
    printf("prefilter %4.2f MiB, far %2d/16 %s: %7.3f ms  %6.1f G queries/s  (%5.1f G far reads/s)\n", small_mib / 4.0, FAR,
           NT ? "nt   " : "plain", ms, n / ms / 1e6, n * FAR / 16 / ms / 1e6);
}

int main() {
    uint32_t *small, *big, *out;
    (void)hipMalloc(&small, 8u << 20); (void)hipMalloc(&big, 512ull << 20); (void)hipMalloc(&out, 64);
    (void)hipMemset(small, 1, 8u << 20); (void)hipMemset(big, 1, 512ull << 20);
    run<16, false>(small, 0, big, out);   // today: every query goes far
    for (uint64_t qmib : {4ull, 8ull, 10ull, 12ull, 14ull, 16ull}) {  // 1, 2, 2.5, 3, 3.5, 4 MiB
        run<0, false>(small, qmib, big, out);
        run<11, false>(small, qmib, big, out);
        run<10, false>(small, qmib, big, out);
        run<9, false>(small, qmib, big, out);
        run<9, true>(small, qmib, big, out);
        run<8, false>(small, qmib, big, out);
    }
    return 0;
}
