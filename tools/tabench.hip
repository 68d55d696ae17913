// tabench.hip — what a random table lookup costs on MI355X by WHERE the table lives and HOW the lanes of one instruction
// spread over cache lines.  Input for the k-mer cover kernel (csrc/score_kmer.hip): its two request classes are 4-byte
// lookups into an L2-resident 2 MiB table (12-mer prefilter) and into a 512 MiB bitmap (one fabric request each).
//   (1) plain / nt / sc1 loads, table 16 KiB .. 2 MiB .. 512 MiB, every lane its own random line
//   (2) G lanes of an instruction share one 128-byte line (G = 1 .. 64), 2 MiB table: is the price per LANE or per LINE?
//   (3) byte loads instead of dword loads
//   (4) LDS: random ds_read_b32 in a 128 KiB table
//   (5) mixed: 16 L2-resident lookups + F far lookups per thread and iteration: do the two classes add up or overlap?
//   (6) waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { PLAIN = 0, NT = 1, SC1 = 2, BYTE = 3 };

template <int FLAVOUR>
__device__ __forceinline__ uint32_t ld(const uint32_t *p) {
    if (FLAVOUR == NT) return __builtin_nontemporal_load(p);
    if (FLAVOUR == SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (FLAVOUR == BYTE) return *reinterpret_cast<const uint8_t *>(p);
    return *p;
}

// every lane group of G lanes shares one random 128-byte line (its lanes read different words of it)
template <int FLAVOUR, int G>
__global__ void __launch_bounds__(256) k_rand(const uint32_t *buf, uint64_t mask_lines, int iters, uint32_t *out) {
    const uint32_t gid = (blockIdx.x * 256u + threadIdx.x) / G;
    uint64_t x = gid * 0x9E3779B97F4A7C15ull + 12345;
    const uint32_t within = (threadIdx.x % G) % 32;  // word inside the line
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t line = (x >> 20) & mask_lines;
            const uint32_t w = G == 1 ? (uint32_t)(x >> 59) : within;
            v[j] = ld<FLAVOUR>(buf + line * 32 + w);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

// 16 L2-resident lookups + F far lookups per iteration
template <int F>
__global__ void __launch_bounds__(256) k_mixed(const uint32_t *small, uint64_t small_mask, const uint32_t *big, uint64_t big_mask,
                                               int iters, uint32_t *out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16], f[F > 0 ? F : 1];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = small[(x >> 20) & small_mask];
        }
#pragma unroll
        for (int j = 0; j < F; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            f[j] = big[(x >> 20) & big_mask];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
#pragma unroll
        for (int j = 0; j < F; ++j) acc ^= f[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

// 16 lookups in a `small` table + F far lookups, far loads non-temporal when NTFAR (does the L2 keep a 3 MiB table next to the far lines?)
template <int F, bool NTFAR>
__global__ void __launch_bounds__(256) k_mixed2(const uint32_t *small, uint32_t small_words, const uint32_t *big, uint64_t big_mask,
                                                int iters, uint32_t *out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16], f[F];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = small[(uint32_t)(((x >> 32) * (uint64_t)small_words) >> 32)];
        }
#pragma unroll
        for (int j = 0; j < F; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            f[j] = NTFAR ? __builtin_nontemporal_load(big + ((x >> 20) & big_mask)) : big[(x >> 20) & big_mask];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
#pragma unroll
        for (int j = 0; j < F; ++j) acc ^= f[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

// SCALAR path: every wave issues 8 s_load_dword to random addresses per iteration (one request per WAVE, not per lane), with
// V vector far lookups per lane beside them: is the scalar data cache's miss path an independent route to far memory?
template <int V>
__global__ void __launch_bounds__(256) k_scalar(const uint32_t *big, uint64_t big_mask, int iters, uint32_t *out) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256u + threadIdx.x) >> 6);
    uint64_t xs = wave * 0x9E3779B97F4A7C15ull + 777;                       // wave-uniform stream (SGPRs)
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;  // per-lane stream
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t f[V > 0 ? V : 1];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            f[j] = big[(x >> 20) & big_mask];
        }
        uint32_t s0, s1, s2, s3, s4, s5, s6, s7;
        const uint32_t *p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xs = xs * 6364136223846793005ull + 1442695040888963407ull;
            p[j] = big + ((xs >> 20) & big_mask);
        }
        asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %9, 0x0\n\ts_load_dword %2, %10, 0x0\n\ts_load_dword %3, %11, 0x0\n\t"
                     "s_load_dword %4, %12, 0x0\n\ts_load_dword %5, %13, 0x0\n\ts_load_dword %6, %14, 0x0\n\ts_load_dword %7, %15, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5), "=&s"(s6), "=&s"(s7)
                     : "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]), "s"(p[4]), "s"(p[5]), "s"(p[6]), "s"(p[7]));
        acc ^= s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
#pragma unroll
        for (int j = 0; j < V; ++j) acc ^= f[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

// far lookups of 32 bytes: two 16-byte loads from one 32-byte aligned entry (W = 2), one 16-byte load (W = 1) or one byte (W = 0)
template <int W>
__global__ void __launch_bounds__(256) k_far_wide(const uint4 *big, uint64_t mask_entries, int iters, uint32_t *out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t e = (x >> 20) & mask_entries;
            if (W == 0) { v[j][0].x = reinterpret_cast<const uint8_t *>(big)[e * 32]; v[j][0].y = v[j][0].z = v[j][0].w = 0; v[j][1] = v[j][0]; }
            else { v[j][0] = big[e * 2]; v[j][1] = W == 2 ? big[e * 2 + 1] : v[j][0]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j][0].x ^ v[j][0].w ^ v[j][1].y ^ v[j][1].z;
    }
    if (acc == 0x12345) out[0] = acc;
}

// far lookups only, F per iteration (same loop shape as k_mixed)
template <int F>
__global__ void __launch_bounds__(256) k_far(const uint32_t *big, uint64_t big_mask, int iters, uint32_t *out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t f[F];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            f[j] = big[(x >> 20) & big_mask];
        }
#pragma unroll
        for (int j = 0; j < F; ++j) acc ^= f[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

// LDS: random 4-byte reads in a table of LDS_WORDS words
template <int LDS_WORDS>
__global__ void __launch_bounds__(256) k_lds(const uint32_t *buf, int iters, uint32_t *out) {
    extern __shared__ uint32_t tab[];
    for (int i = threadIdx.x; i < LDS_WORDS; i += 256) tab[i] = buf[i];
    __syncthreads();
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = tab[(x >> 20) & (LDS_WORDS - 1)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

static hipEvent_t ea, eb;
template <typename L>
static float timed(L launch) {
    launch(4);
    hipDeviceSynchronize();
    hipEventRecord(ea);
    launch(100);
    hipEventRecord(eb);
    hipEventSynchronize(eb);
    float ms;
    hipEventElapsedTime(&ms, ea, eb);
    return ms;
}

int main(int argc, char **argv) {
    const bool only7 = argc > 1 && (argv[1][0] == '7' || argv[1][0] == '8' || argv[1][0] == '9');
    const uint64_t big_bytes = 512ull << 20;
    uint32_t *buf, *out;
    CK(hipMalloc(&buf, big_bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, big_bytes));
    hipEventCreate(&ea);
    hipEventCreate(&eb);
    const int blocks = 256 * 8;
    const double per_iter = (double)blocks * 256 * 16;

    if (!only7) {
    printf("# (1) every lane its own random line: G lookups/s by table size and load flavour (2048 x 256 threads, 16 loads in flight per thread)\n");
    for (uint64_t kib : {16ull, 64ull, 256ull, 1024ull, 2048ull, 4096ull, 65536ull, 524288ull}) {
        const uint64_t mask = kib * 1024 / 128 - 1;
        float a = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 1>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float b = timed([&](int it) { hipLaunchKernelGGL((k_rand<NT, 1>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float c = timed([&](int it) { hipLaunchKernelGGL((k_rand<SC1, 1>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float d = timed([&](int it) { hipLaunchKernelGGL((k_rand<BYTE, 1>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        printf("table %7llu KiB: plain %7.1f   nt %7.1f   sc1 %7.1f   byte %7.1f  G/s\n", (unsigned long long)kib, per_iter * 100 / a / 1e6,
               per_iter * 100 / b / 1e6, per_iter * 100 / c / 1e6, per_iter * 100 / d / 1e6);
    }
    printf("# (2) G lanes of an instruction share one 128-byte line, 2 MiB table (plain loads): G lane-lookups/s\n");
    {
        const uint64_t mask = 2048 * 1024 / 128 - 1;
        float t1 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 1>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float t2 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 2>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float t4 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 4>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float t8 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 8>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float t16 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 16>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float t32 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 32>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        float t64 = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 64>), dim3(blocks), dim3(256), 0, 0, buf, mask, it, out); });
        const float ts[7] = {t1, t2, t4, t8, t16, t32, t64};
        for (int i = 0; i < 7; ++i) printf("G %2d: %7.1f G lane-lookups/s  (%6.1f G lines/s)\n", 1 << i, per_iter * 100 / ts[i] / 1e6, per_iter * 100 / ts[i] / 1e6 / (1 << i));
    }
    printf("# (4) LDS: random ds_read_b32, 128 KiB table per workgroup of 256 threads (1 workgroup per CU) and 32 KiB (4 per CU)\n");
    {
        hipFuncSetAttribute((const void *)k_lds<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        float a = timed([&](int it) { hipLaunchKernelGGL((k_lds<32768>), dim3(blocks), dim3(256), 131072, 0, buf, it * 4, out); });
        float b = timed([&](int it) { hipLaunchKernelGGL((k_lds<8192>), dim3(blocks), dim3(256), 32768, 0, buf, it * 4, out); });
        printf("LDS 128 KiB, 4 waves/CU: %7.1f G/s     LDS 32 KiB, 16 waves/CU: %7.1f G/s\n", per_iter * 400 / a / 1e6, per_iter * 400 / b / 1e6);
    }
    printf("# (5) 16 L2-resident lookups (2 MiB) + F far lookups (512 MiB) per thread and iteration; far alone for the same F\n");
    {
        const uint64_t sm = 2048 * 1024 / 4 - 1, bm = big_bytes / 4 - 1;
        float m0 = timed([&](int it) { hipLaunchKernelGGL((k_mixed<0>), dim3(blocks), dim3(256), 0, 0, buf, sm, buf, bm, it, out); });
        float m1 = timed([&](int it) { hipLaunchKernelGGL((k_mixed<1>), dim3(blocks), dim3(256), 0, 0, buf, sm, buf, bm, it, out); });
        float m2 = timed([&](int it) { hipLaunchKernelGGL((k_mixed<2>), dim3(blocks), dim3(256), 0, 0, buf, sm, buf, bm, it, out); });
        float m3 = timed([&](int it) { hipLaunchKernelGGL((k_mixed<3>), dim3(blocks), dim3(256), 0, 0, buf, sm, buf, bm, it, out); });
        float m4 = timed([&](int it) { hipLaunchKernelGGL((k_mixed<4>), dim3(blocks), dim3(256), 0, 0, buf, sm, buf, bm, it, out); });
        float f1 = timed([&](int it) { hipLaunchKernelGGL((k_far<1>), dim3(blocks), dim3(256), 0, 0, buf, bm, it, out); });
        float f2 = timed([&](int it) { hipLaunchKernelGGL((k_far<2>), dim3(blocks), dim3(256), 0, 0, buf, bm, it, out); });
        float f3 = timed([&](int it) { hipLaunchKernelGGL((k_far<3>), dim3(blocks), dim3(256), 0, 0, buf, bm, it, out); });
        float f4 = timed([&](int it) { hipLaunchKernelGGL((k_far<4>), dim3(blocks), dim3(256), 0, 0, buf, bm, it, out); });
        float f16 = timed([&](int it) { hipLaunchKernelGGL((k_far<16>), dim3(blocks), dim3(256), 0, 0, buf, bm, it, out); });
        const float ms[5] = {m0, m1, m2, m3, m4}, fs[5] = {0, f1, f2, f3, f4};
        for (int i = 0; i < 5; ++i)
            printf("F %d: mixed %8.3f ms   far alone %8.3f ms   L2 alone %8.3f ms   (sum %8.3f)\n", i, ms[i], fs[i], m0, m0 + fs[i]);
        printf("far alone, 16 in flight per thread: %7.1f G/s;  1 in flight: %7.1f G/s\n", per_iter * 100 / f16 / 1e6, per_iter / 16 * 100 / f1 / 1e6);
    }
    }
    printf("# (9) far lookups of one byte / 16 bytes / 32 bytes (two 16-byte loads of one aligned entry), 4 in flight per thread, 512 MiB\n");
    {
        const uint64_t me = big_bytes / 32 - 1;
        float w0 = timed([&](int it) { hipLaunchKernelGGL((k_far_wide<0>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)buf, me, it * 4, out); });
        float w1 = timed([&](int it) { hipLaunchKernelGGL((k_far_wide<1>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)buf, me, it * 4, out); });
        float w2 = timed([&](int it) { hipLaunchKernelGGL((k_far_wide<2>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)buf, me, it * 4, out); });
        const double n = (double)blocks * 256 * 4 * 400;
        printf("byte %7.1f G entries/s   16 B %7.1f G entries/s   32 B (2 loads) %7.1f G entries/s\n", n / w0 / 1e6, n / w1 / 1e6, n / w2 / 1e6);
    }
    printf("# (8) scalar path: 8 s_load_dword per wave and iteration to random addresses in 512 MiB, with V vector far lookups per lane beside them\n");
    {
        const uint64_t bm = big_bytes / 4 - 1;
        const double waves = (double)blocks * 4;
        float s0 = timed([&](int it) { hipLaunchKernelGGL((k_scalar<0>), dim3(blocks), dim3(256), 0, 0, buf, bm, it * 8, out); });
        float s1 = timed([&](int it) { hipLaunchKernelGGL((k_scalar<1>), dim3(blocks), dim3(256), 0, 0, buf, bm, it * 8, out); });
        float s4 = timed([&](int it) { hipLaunchKernelGGL((k_scalar<4>), dim3(blocks), dim3(256), 0, 0, buf, bm, it * 8, out); });
        float f1 = timed([&](int it) { hipLaunchKernelGGL((k_far<1>), dim3(blocks), dim3(256), 0, 0, buf, bm, it * 8, out); });
        float f4 = timed([&](int it) { hipLaunchKernelGGL((k_far<4>), dim3(blocks), dim3(256), 0, 0, buf, bm, it * 8, out); });
        printf("scalar alone: %8.3f ms = %6.2f G scalar requests/s\n", s0, waves * 8 * 800 / s0 / 1e6);
        printf("V=1: scalar+vector %8.3f ms, vector alone %8.3f ms  -> vector %6.1f G/s + scalar %6.2f G/s\n", s1, f1, (double)blocks * 256 * 800 / s1 / 1e6, waves * 8 * 800 / s1 / 1e6);
        printf("V=4: scalar+vector %8.3f ms, vector alone %8.3f ms  -> vector %6.1f G/s + scalar %6.2f G/s\n", s4, f4, (double)blocks * 256 * 4 * 800 / s4 / 1e6, waves * 8 * 800 / s4 / 1e6);
    }
    printf("# (7) 16 lookups in a table of S MiB + 4 far lookups (512 MiB) per iteration, far loads plain / nt: ms per 100 iterations\n");
    {
        const uint64_t bm = big_bytes / 4 - 1;
        for (double mib : {1.0, 2.0, 2.5, 3.0, 3.5, 4.0}) {
            const uint32_t words = (uint32_t)(mib * (1 << 20) / 4);
            float a = timed([&](int it) { hipLaunchKernelGGL((k_mixed2<4, false>), dim3(blocks), dim3(256), 0, 0, buf, words, buf, bm, it, out); });
            float b = timed([&](int it) { hipLaunchKernelGGL((k_mixed2<4, true>), dim3(blocks), dim3(256), 0, 0, buf, words, buf, bm, it, out); });
            float c = timed([&](int it) { hipLaunchKernelGGL((k_mixed2<1, false>), dim3(blocks), dim3(256), 0, 0, buf, words, buf, bm, it, out); });
            printf("table %.1f MiB: F=4 plain %8.3f ms   F=4 nt %8.3f ms   F=1 plain %8.3f ms\n", mib, a, b, c);
        }
    }
    if (only7) return 0;
    printf("# (6) 2 MiB table, plain loads, by resident workgroups (256 threads) per CU\n");
    {
        const uint64_t mask = 2048 * 1024 / 128 - 1;
        for (int wg : {1, 2, 3, 4, 6, 8}) {
            const int b = 256 * wg;
            float a = timed([&](int it) { hipLaunchKernelGGL((k_rand<PLAIN, 1>), dim3(b), dim3(256), 0, 0, buf, mask, it * 8 / wg, out); });
            printf("%d workgroups/CU: %7.1f G/s\n", wg, (double)b * 256 * 16 * (100 * 8 / wg) / a / 1e6);
        }
    }
    return 0;
}
