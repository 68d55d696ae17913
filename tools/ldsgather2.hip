// ldsgather2.hip — (1) rate of ds_read_b64 table gathers on gfx950 with NO other work in the loop (inline asm), by
// address pattern and occupancy; (2) the steady-state scoring loop with the window history held in REGISTERS
// (no LDS ring), plain vs bank-private table layout.  Results go to profiles/r02_microbench.txt.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ldsgather2 tools/ldsgather2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &x) { x = x * 1664525u + 1013904223u; return x >> 8; }

// ------------------------------------------------------------------------------------------------------
// (1) pure gathers
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_gather(int pattern, int iters, unsigned long long *ticks, double *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *t = reinterpret_cast<double *>(smem);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) t[i] = 1.0 / (double)(i + 1);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t x = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u + 12345u;
    uint32_t ctr = x;
    const uint32_t centre = 8 + lcg(ctr) % 18;  // synthetic C2: per-read centre 8..25
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t r = lcg(x);
        uint32_t q;
        {
            int v = (int)centre + (int)(r % 9) - 4 + (int)((r >> 8) % 9) - 4;
            v = v < 1 ? 1 : (v > 60 ? 60 : v);
            q = (uint32_t)v;
        }
        switch (pattern) {
            case 0: a[i] = lane * 8 + i * 512; break;                       // unit stride (the guide's best case)
            case 1: a[i] = i * 8; break;                                    // one address for the whole wave
            case 2: a[i] = (r % 32) * 8; break;                             // 32 entries: duplicates, no bank sharing
            case 3: a[i] = (r % 64) * 8; break;                             // e and e+32 share a bank pair
            case 4: a[i] = (r % 94) * 8; break;                             // full Phred alphabet, uniform
            case 5: a[i] = ((r % 94) * 32 + (lane & 31)) * 8; break;        // bank-private copies: never a conflict
            case 6: a[i] = (q + 33) * 8; break;                             // synthetic C2 distribution, plain table
            case 7: a[i] = (q * 32 + (lane & 31)) * 8; break;               // synthetic C2, bank-private
            case 8: a[i] = ((2 + r % 49) + 33) * 8; break;                  // Q2..Q50 uniform, plain table
            default: a[i] = ((r % 128) * 32 + (lane & 31)) * 8; break;      // 128-entry bank-private (32 KB)
        }
    }
    double acc = 0.0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        double d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11;
        asm volatile(
            "ds_read_b64 %0, %12\n\tds_read_b64 %1, %13\n\tds_read_b64 %2, %14\n\tds_read_b64 %3, %15\n\t"
            "ds_read_b64 %4, %16\n\tds_read_b64 %5, %17\n\tds_read_b64 %6, %18\n\tds_read_b64 %7, %19\n\t"
            "ds_read_b64 %8, %20\n\tds_read_b64 %9, %21\n\tds_read_b64 %10, %22\n\tds_read_b64 %11, %23\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7), "=&v"(d8), "=&v"(d9),
              "=&v"(d10), "=&v"(d11)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
              "v"(a[10]), "v"(a[11]));
        if (it == iters - 1) acc = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + d8 + d9 + d10 + d11;
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc == 1234.5) sink[0] = acc;
}

// ------------------------------------------------------------------------------------------------------
// (2) steady-state scoring loop, history in registers.  R pieces of 16 bytes form a register ring (unrolled, static
// indices); per round 4 new pieces are read from a per-wave LDS slot (which a real kernel fills by LDS-DMA);
// the trailing pieces are t-A-1 and t-A, funnel-shifted by B bytes.  ws = 16*A + B.
// ------------------------------------------------------------------------------------------------------
template <int SEL>
__device__ __forceinline__ uint32_t addr_plain(uint32_t x) {  // byte SEL of x, times 8
    uint32_t r;
    const uint32_t three = 3;
    if (SEL == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(x));
    else if (SEL == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(x));
    else if (SEL == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(x));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(x));
    return r;
}
// bank-private layout: address = byte * 256 + laneoff, laneoff = (lane & 31) * 8 < 256: one v_perm_b32
template <int SEL>
__device__ __forceinline__ uint32_t addr_priv(uint32_t x, uint32_t laneoff) {
    // v_perm_b32 D, S0, S1, sel: bytes 0-3 of S1, 4-7 of S0; selector byte 0x0c = constant 0x00
    constexpr uint32_t sel = 0x0c0c0000u | ((4u + SEL) << 8) | 0u;  // D.b0 = S1.b0 (laneoff), D.b1 = S0.b[SEL], D.b2 = D.b3 = 0
    return __builtin_amdgcn_perm(x, laneoff, sel);
}

// 4 bases of the steady state as ONE asm statement: 12 gathers issued back to back, the four dependent FP64 ops per base
// follow as the data arrives.  (Left to hipcc, the unrolled loop gets its gathers hoisted away from the chain and spills.)
template <int QOFF, int DOFF>
__device__ __forceinline__ void fold4(uint32_t aj0, uint32_t aj1, uint32_t aj2, uint32_t aj3, uint32_t ai0, uint32_t ai1,
                                      uint32_t ai2, uint32_t ai3, double &s, double &w, double &mn) {
    double q0, i0, j0, q1, i1, j1, q2, i2, j2, q3, i3, j3;
    asm volatile(
        "ds_read_b64 %3, %15 offset:%23\n\tds_read_b64 %4, %19 offset:%24\n\tds_read_b64 %5, %15 offset:%24\n\t"
        "ds_read_b64 %6, %16 offset:%23\n\tds_read_b64 %7, %20 offset:%24\n\tds_read_b64 %8, %16 offset:%24\n\t"
        "ds_read_b64 %9, %17 offset:%23\n\tds_read_b64 %10, %21 offset:%24\n\tds_read_b64 %11, %17 offset:%24\n\t"
        "ds_read_b64 %12, %18 offset:%23\n\tds_read_b64 %13, %22 offset:%24\n\tds_read_b64 %14, %18 offset:%24\n\t"
        "s_waitcnt lgkmcnt(11)\n\tv_add_f64 %0, %0, %3\n\t"
        "s_waitcnt lgkmcnt(10)\n\tv_add_f64 %1, %1, -%4\n\t"
        "s_waitcnt lgkmcnt(9)\n\tv_add_f64 %1, %1, %5\n\tv_min_f64 %2, %2, %1\n\t"
        "s_waitcnt lgkmcnt(8)\n\tv_add_f64 %0, %0, %6\n\t"
        "s_waitcnt lgkmcnt(7)\n\tv_add_f64 %1, %1, -%7\n\t"
        "s_waitcnt lgkmcnt(6)\n\tv_add_f64 %1, %1, %8\n\tv_min_f64 %2, %2, %1\n\t"
        "s_waitcnt lgkmcnt(5)\n\tv_add_f64 %0, %0, %9\n\t"
        "s_waitcnt lgkmcnt(4)\n\tv_add_f64 %1, %1, -%10\n\t"
        "s_waitcnt lgkmcnt(3)\n\tv_add_f64 %1, %1, %11\n\tv_min_f64 %2, %2, %1\n\t"
        "s_waitcnt lgkmcnt(2)\n\tv_add_f64 %0, %0, %12\n\t"
        "s_waitcnt lgkmcnt(1)\n\tv_add_f64 %1, %1, -%13\n\t"
        "s_waitcnt lgkmcnt(0)\n\tv_add_f64 %1, %1, %14\n\tv_min_f64 %2, %2, %1"
        : "+v"(s), "+v"(w), "+v"(mn), "=&v"(q0), "=&v"(i0), "=&v"(j0), "=&v"(q1), "=&v"(i1), "=&v"(j1), "=&v"(q2), "=&v"(i2),
          "=&v"(j2), "=&v"(q3), "=&v"(i3), "=&v"(j3)
        : "v"(aj0), "v"(aj1), "v"(aj2), "v"(aj3), "v"(ai0), "v"(ai1), "v"(ai2), "v"(ai3), "i"(QOFF), "i"(DOFF));
}

template <bool PRIV, int BATCH>
__device__ __forceinline__ void body16(const unsigned char *lq, const unsigned char *ld, const uint32_t (&lw)[4],
                                       const uint32_t (&tw)[4], uint32_t laneoff, double &s, double &w, double &mn) {
    constexpr int QOFF = 0, DOFF = PRIV ? 128 * 256 : 264 * 8;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (PRIV)
            fold4<QOFF, DOFF>(addr_priv<0>(lw[d], laneoff), addr_priv<1>(lw[d], laneoff), addr_priv<2>(lw[d], laneoff),
                              addr_priv<3>(lw[d], laneoff), addr_priv<0>(tw[d], laneoff), addr_priv<1>(tw[d], laneoff),
                              addr_priv<2>(tw[d], laneoff), addr_priv<3>(tw[d], laneoff), s, w, mn);
        else
            fold4<QOFF, DOFF>(addr_plain<0>(lw[d]), addr_plain<1>(lw[d]), addr_plain<2>(lw[d]), addr_plain<3>(lw[d]),
                              addr_plain<0>(tw[d]), addr_plain<1>(tw[d]), addr_plain<2>(tw[d]), addr_plain<3>(tw[d]), s, w, mn);
    }
}

template <int WAVES, int A, bool PRIV, int BATCH, int MODE>  // MODE 0 full, 1 no FP64 chain, 2 no gathers
__global__ void __launch_bounds__(WAVES * 64) k_steady(int B, int rounds5, unsigned long long *ticks, double *sink) {
    constexpr int R = ((A + 5 + 3) / 4) * 4;  // ring pieces: multiple of 4, >= A + 5
    constexpr int E = PRIV ? 128 : 264;
    constexpr int TAB = PRIV ? E * 256 : E * 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lq = smem;
    unsigned char *ld = smem + TAB;
    unsigned char *slots = smem + 2 * TAB;
    for (int i = threadIdx.x; i < 2 * TAB / 8; i += WAVES * 64) reinterpret_cast<double *>(smem)[i] = 1.0 / (double)(1 + (i % 300));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *slot = slots + wave * 4096;
    {
        uint32_t x = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u + 12345u;
        uint32_t c = x;
        const uint32_t centre = 8 + lcg(c) % 18;
        for (int i = 0; i < 64; ++i) {
            const uint32_t r = lcg(x);
            int v = (int)centre + (int)(r % 9) - 4 + (int)((r >> 8) % 9) - 4;
            v = v < 1 ? 1 : (v > 60 ? 60 : v);
            slot[lane * 64 + ((i + lane * 16) & 63)] = (unsigned char)(v + 33);  // row per lane, 16-byte pieces rotated by lane
        }
    }
    __syncthreads();
    const uint32_t laneoff = (lane & 31) * 8;
    const unsigned char *my = slot + lane * 64;
    uint32_t ring[R][4];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint4 v = *reinterpret_cast<const uint4 *>(my + (i & 3) * 16);
        ring[i][0] = v.x; ring[i][1] = v.y; ring[i][2] = v.z; ring[i][3] = v.w;
    }
    double s = 0.0, w = 0.5, mn = 0.5;
    const uint32_t bsh = (uint32_t)B & 3u;
    const int csel = (B >> 2) & 3;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < rounds5; ++it) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
            if ((t & 3) == 0) {  // new round: 4 pieces from the slot into the ring
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(my + (((k + (lane >> 2)) & 3) * 16));
                    ring[t + k][0] = v.x; ring[t + k][1] = v.y; ring[t + k][2] = v.z; ring[t + k][3] = v.w;
                }
            }
            // trailing 16 bytes: pieces t-A-1 and t-A of the ring, shifted by 16 - B bytes (B = ws % 16)
            const uint32_t(&p0)[4] = ring[(t + R - A - 1) % R];
            const uint32_t(&p1)[4] = ring[(t + R - A) % R];
            const uint32_t x8[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
            uint32_t tw[4];
            // bytes [16 - B, 32 - B) of x8: dword offset 3 - csel (for B % 4 != 0), byte shift 4 - bsh
#define FUN(D)                                                                                                         \
    _Pragma("unroll") for (int d = 0; d < 4; ++d) tw[d] = __builtin_amdgcn_alignbyte(x8[(D) + d + 1], x8[(D) + d], (4 - bsh) & 3);
            switch (csel) {
                case 0: FUN(3) break;
                case 1: FUN(2) break;
                case 2: FUN(1) break;
                default: FUN(0) break;
            }
#undef FUN
            if (MODE == 2) {
                s += __hiloint2double(ring[t][0], ring[t][1]);
                w += __hiloint2double(tw[0], tw[1]);
                mn = fmin(mn, w);
            } else if (MODE == 1) {
                double s2 = 0, w2 = 0, m2 = 0;
                body16<PRIV, BATCH>(lq, ld, ring[t], tw, laneoff, s2, w2, m2);
                s = s2; w = w2; mn = m2;
            } else {
                body16<PRIV, BATCH>(lq, ld, ring[t], tw, laneoff, s, w, mn);
            }
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (s + w + mn == 1234.5) sink[0] = s;
}

static unsigned long long *d_ticks;
static double *d_sink;

static void run_gather(int pattern, int waves, const char *name) {
    const int iters = 3000;
    const size_t lds = 64 * 1024;
    CK(hipFuncSetAttribute((const void *)k_gather, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_gather, dim3(256), dim3(waves * 64), lds, 0, pattern, 10, d_ticks, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_gather, dim3(256), dim3(waves * 64), lds, 0, pattern, iters, d_ticks, d_sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[256]; CK(hipMemcpy(h, d_ticks, sizeof h, hipMemcpyDeviceToHost));
    double tk = 0; for (int i = 0; i < 256; ++i) tk += (double)h[i]; tk /= 256;
    const double n = (double)iters * 12 * waves;  // wave-gathers per CU
    printf("gather %-34s waves/CU %2d: %7.3f ms  %6.3f ns/gather/CU  %5.2f ticks/gather  (C2 3-gather floor %5.1f ms)\n", name, waves, ms,
           ms * 1e6 / n, tk / n, ms * 1e6 / n * 1e-6 * 3.0 * 1e11 / 64 / 256);
}

template <int WAVES, int A, bool PRIV, int BATCH, int MODE>
static void run_steady(int B, const char *name) {
    constexpr int R = ((A + 5 + 3) / 4) * 4;
    const int rounds5 = 400;
    const size_t lds = (PRIV ? 2 * 128 * 256 : 2 * 264 * 8) + WAVES * 4096;
    auto kern = k_steady<WAVES, A, PRIV, BATCH, MODE>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), lds, 0, B, 4, d_ticks, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), lds, 0, B, rounds5, d_ticks, d_sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)kern));
    const double steps = (double)rounds5 * R * 16 * WAVES;  // 64-base steps per CU
    printf("steady %-28s waves/CU %2d A %2d B %2d batch %2d vgpr %3d: %7.3f ms  %6.3f ns per 64 bases per CU  => C2 %5.1f ms\n", name, WAVES, A, B,
           BATCH, fa.numRegs, ms, ms * 1e6 / steps, ms * 1e6 / steps * 1e-6 * 1e11 / 64 / 256);
}

int main() {
    CK(hipMalloc(&d_ticks, 256 * 8)); CK(hipMalloc(&d_sink, 64));
    const char *names[] = {"unit stride", "single address", "random 32 entries", "random 64 entries", "random 94 entries",
                           "94 entries bank-private", "synthetic C2 plain", "synthetic C2 bank-private", "Q2..Q50 plain",
                           "128 entries bank-private"};
    for (int w : {4, 8, 12, 16}) for (int p = 0; p < 10; ++p) run_gather(p, w, names[p]);
    // steady loop: plain vs bank-private tables, by occupancy
    run_steady<4, 15, false, 4, 0>(10, "plain");
    run_steady<8, 15, false, 4, 0>(10, "plain");
    run_steady<12, 15, false, 4, 0>(10, "plain");
    run_steady<16, 15, false, 4, 0>(10, "plain");
    run_steady<4, 15, true, 4, 0>(10, "private");
    run_steady<8, 15, true, 4, 0>(10, "private");
    run_steady<12, 15, true, 4, 0>(10, "private");
    run_steady<16, 15, true, 4, 0>(10, "private");
    run_steady<12, 15, true, 4, 2>(10, "private, no gathers");
    run_steady<16, 15, true, 4, 2>(10, "private, no gathers");
    return 0;
}
