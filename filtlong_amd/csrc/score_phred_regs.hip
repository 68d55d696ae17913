// score_phred_regs.hip — Phred-only per-read scoring, register-history kernel (the default fast path).
//
// Same arithmetic as score_phred.hip (reference src/read.cpp:35-39, 208-236, 64-73; one lane folds one read strictly left
// to right, tables built with the host libm), different data movement:
//
//   * the last `window_size` bytes of every read live in the lane's REGISTERS (a ring of 16-byte pieces, statically
//     indexed: the loop is unrolled over one ring revolution), not in LDS.  The register file is the largest on-chip
//     store of a CU (512 KiB vs 160 KiB of LDS); with the history out of LDS a wave needs 4 KiB of LDS instead of
//     21 KiB, so 12-16 waves fit a CU instead of 7 — and ds_read_b64 table gathers only reach the LDS's rate
//     (1.0 ns per wave-gather per CU instead of 1.6) from 3-4 waves per SIMD (profiles/r02_microbench.txt).
//   * the plane is still read exactly once: per 128 bases a wave moves one 128-byte line of each of its 64 reads
//     HBM -> LDS with eight `global_load_lds_dwordx4` (no staging VGPRs, no ds_write), into one 8 KiB slot whose rows
//     are rotated so that every lane then pulls its own bytes with conflict-free ds_read_b128.  The next chunk's DMA
//     is issued as soon as the slot has been read, a whole round (64 bases per lane) of compute ahead.
//   * per base: 2 address ops, 3 ds_read_b64 gathers, 3 v_add_f64, 1 v_min_f64 — written as asm statements of
//     4 bases (12 gathers issued back to back, the FP64 chain follows as data arrives); left to hipcc the unrolled
//     loop gets its gathers hoisted away from the chain and spills.
//   * optional bank-private tables (FLX_PHRED_TABLES=private): every table entry is replicated once per bank pair
//     (lane l reads copy l % 32), so a gather never has a bank conflict whatever the quality distribution is
//     (plain tables: entries e and e+32 collide; a wide Phred range costs up to 1.5x per gather).  128 entries per
//     table; a read containing a byte >= 128 (not a FASTQ character) is flagged and re-scored by the direct kernel.
//
// The window size enters as A = ws / 16 (template parameter: it fixes which ring pieces hold the trailing edge)
// and B = ws % 16 (run time: a byte funnel).
#include <algorithm>

#include "flx_internal.h"
#include "score_phred_common.h"

using namespace flx_phred;

#ifndef FLX_REGS_PART
#define FLX_REGS_PART 0
#endif

namespace {

template <bool PRIV>
struct Tab {
    static constexpr int ROW = PRIV ? 256 : 8;              // bytes between consecutive entries
    static constexpr int ENTRIES = PRIV ? 129 : LUT_PAD;    // rows per table (incl. the all-zero "no base" entry)
    static constexpr int BYTES = ROW * ENTRIES;
    static constexpr int QOFF = 0;                           // LDS byte address of the Q table (dynamic LDS starts at 0)
    static constexpr int DOFF = BYTES;                       // ... of the D = Q / ws table
    static constexpr int ZIDX = PRIV ? 128 : 256;            // the entry that holds 0.0 in both tables
    static constexpr int SLOT0 = 2 * BYTES;                  // first DMA slot (multiple of 16)
};

// table address of byte SEL of dword x
template <bool PRIV, int SEL>
__device__ __forceinline__ uint32_t tab_addr(uint32_t x, uint32_t laneoff) {
    if (PRIV) {
        // v_perm_b32: D.b0 = laneoff.b0 (copy select, < 256), D.b1 = x.b[SEL] (entry * 256), D.b2 = D.b3 = 0
        constexpr uint32_t sel = 0x0c0c0000u | ((4u + SEL) << 8);
        return __builtin_amdgcn_perm(x, laneoff, sel);
    }
    return lut_addr(x, SEL);  // byte * 8, one SDWA shift
}

// ---- the FP64 folds as asm statements (see the header comment) ---------------------------------------------
// steady state, 4 bases: s += Q[new]; w -= D[old]; w += D[new]; mn = min(mn, w)     (src/read.cpp:210, 228-231)
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
// positions before the first full window, 4 bases: s += Q[new]
template <int QOFF>
__device__ __forceinline__ void head4(uint32_t aj0, uint32_t aj1, uint32_t aj2, uint32_t aj3, double &s) {
    double q0, q1, q2, q3;
    asm volatile(
        "ds_read_b64 %1, %5 offset:%9\n\tds_read_b64 %2, %6 offset:%9\n\tds_read_b64 %3, %7 offset:%9\n\t"
        "ds_read_b64 %4, %8 offset:%9\n\t"
        "s_waitcnt lgkmcnt(3)\n\tv_add_f64 %0, %0, %1\n\ts_waitcnt lgkmcnt(2)\n\tv_add_f64 %0, %0, %2\n\t"
        "s_waitcnt lgkmcnt(1)\n\tv_add_f64 %0, %0, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 %0, %0, %4"
        : "+v"(s), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
        : "v"(aj0), "v"(aj1), "v"(aj2), "v"(aj3), "i"(QOFF));
}
// single bases, for the one piece that contains position window_size
template <int QOFF>
__device__ __forceinline__ void head1(uint32_t aj, double &s) {
    double q;
    asm volatile("ds_read_b64 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 %0, %0, %1" : "+v"(s), "=&v"(q) : "v"(aj), "i"(QOFF));
}
template <int QOFF, int DOFF>
__device__ __forceinline__ void fold1(uint32_t aj, uint32_t ai, double &s, double &w, double &mn) {
    double q, i, j;
    asm volatile(
        "ds_read_b64 %3, %6 offset:%8\n\tds_read_b64 %4, %7 offset:%9\n\tds_read_b64 %5, %6 offset:%9\n\t"
        "s_waitcnt lgkmcnt(2)\n\tv_add_f64 %0, %0, %3\n\ts_waitcnt lgkmcnt(1)\n\tv_add_f64 %1, %1, -%4\n\t"
        "s_waitcnt lgkmcnt(0)\n\tv_add_f64 %1, %1, %5\n\tv_min_f64 %2, %2, %1"
        : "+v"(s), "+v"(w), "+v"(mn), "=&v"(q), "=&v"(i), "=&v"(j)
        : "v"(aj), "v"(ai), "i"(QOFF), "i"(DOFF));
}

// LDS-DMA: 16 bytes per lane from each lane's own global address to LDS address lds_dst + lane * 16
// (`lds_dst` wave-uniform).  M0 holds the LDS base and is compiler-reserved: set and restored in one statement.
__device__ __forceinline__ void dma16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// bytes [16 - B, 32 - B) of the 32 bytes (p0 : p1), B = ws % 16:  fD = (16 - B) / 4, fsh = (16 - B) % 4
__device__ __forceinline__ void funnel(const uint32_t (&p0)[4], const uint32_t (&p1)[4], int fD, uint32_t fsh, uint32_t (&tw)[4]) {
    const uint32_t x[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#define FLX_FUN(D) _Pragma("unroll") for (int d = 0; d < 4; ++d) tw[d] = __builtin_amdgcn_alignbyte(x[(D) + d + 1], x[(D) + d], fsh);
    switch (fD) {  // wave-uniform and constant for the whole launch
        case 0: FLX_FUN(0) break;
        case 1: FLX_FUN(1) break;
        case 2: FLX_FUN(2) break;
        case 3: FLX_FUN(3) break;
        default: tw[0] = x[4]; tw[1] = x[5]; tw[2] = x[6]; tw[3] = x[7]; break;  // B == 0
    }
#undef FLX_FUN
}

template <int A, bool PRIV, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) flx_score_phred_regs(const PhredArgs a) {
    using T = Tab<PRIV>;
    constexpr int R = ((A + 5 + 3) / 4) * 4;  // ring pieces (16 bytes each): a multiple of the 4 pieces per round, >= A + 5
    constexpr int RR = R / 4;                 // ring rounds = unroll factor of the main loop
    constexpr int H = (A + 1 + 3) / 4;        // prologue rounds: they cover pieces 0..A (everything up to position ws)
    constexpr int SLOT_BYTES = 8192;          // LDS per wave: one 128-byte chunk of each of its 64 reads
    static_assert(H <= RR, "prologue must fit one ring revolution");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // tables (dynamic LDS starts at address 0: this kernel has no static LDS, so the table bases fold into the ds_read
    // offset fields as compile-time constants)
    if (PRIV) {
        for (int i = threadIdx.x; i < 129 * 32; i += WAVES * 64) {
            const int e = i >> 5;
            const double q = e < 128 ? a.lut_q[e] : 0.0, d = e < 128 ? a.lut_d[e] : 0.0;
            *reinterpret_cast<double *>(smem + T::QOFF + i * 8) = q;
            *reinterpret_cast<double *>(smem + T::DOFF + i * 8) = d;
        }
    } else {
        for (int i = threadIdx.x; i < 257; i += WAVES * 64) {
            *reinterpret_cast<double *>(smem + T::QOFF + i * 8) = a.lut_q[i];
            *reinterpret_cast<double *>(smem + T::DOFF + i * 8) = a.lut_d[i];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned char *slot = smem + T::SLOT0 + wave * SLOT_BYTES;
    const uint32_t slot_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(T::SLOT0 + wave * SLOT_BYTES));
    // Slot layout: two halves of 4 KiB, half h holding bytes [64 h, 64 h + 64) of the current 128-byte chunk of every
    // read as a row of 64 bytes (row = lane that owns the read), the four 16-byte pieces of row r stored rotated by
    // r >> 2, so that the ds_read_b128 lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... touch 16 distinct bank
    // quads (MI355X_MICROARCH.md §LDS).  The DMA lands lane-linear (lane l -> slot + 1024 m + 16 l), so the rotation is
    // applied to the SOURCE address: DMA m (0..7), lane l carries piece ((l & 3) - (l >> 4)) & 3 of half m >> 2 of
    // read 16 (m & 3) + (l >> 2).  A chunk is one whole 128-byte line when the read starts 128-byte aligned
    // (flx_plane_layout does that): both halves of a line are requested back to back, so the line crosses the fabric
    // once.  (With 64-byte chunks the second half of a line was requested a whole round later, when the L2 — 4 MiB per
    // XCD for 32 CUs x 768 reads — had usually dropped it: FETCH_SIZE 1.8x the bytes, profiles/r02_*.)
    const unsigned char *my_row = slot + lane * 64;
    const int rot = lane >> 2;
    const uint32_t laneoff = (uint32_t)(lane & 31) * 8u;
    const uint32_t zaddr = PRIV ? (uint32_t)(T::ZIDX * T::ROW) + laneoff : (uint32_t)(T::ZIDX * T::ROW);
    const int ws = a.ws;
    const int B = ws & 15;
    const int fD = (16 - B) >> 2;
    const uint32_t fsh = (uint32_t)(16 - B) & 3u;
    const double ws_d = a.ws_d;

  for (;;) {
    unsigned int group = 0;
    if (lane == 0) group = atomicAdd(a.ticket, 1u);
    group = (unsigned int)__builtin_amdgcn_readfirstlane((int)group);
    if (group >= a.n_groups) break;
    const uint64_t gslot = (uint64_t)group * 64 + lane;
    const bool live = gslot < a.n_reads;
    uint32_t rid = 0;
    int L = 0;
    uint64_t base = 0;
    if (live) {
        rid = a.order ? a.order[gslot] : (uint32_t)gslot;
        L = a.lengths[rid];
        base = a.offsets[rid];
    }
    const int Lmax = wave_max(L);
    const int Lmin = wave_min(L);
    if (Lmax == 0) {
        if (live) finish_read(a, rid, L, 0.0, 0.0);
        continue;
    }
    const int n_rounds = (Lmax + 63) >> 6;

    // DMA map of this lane: for m = 0..7 piece `dq` of half m >> 2 of read 16 (m & 3) + (lane >> 2)
    const int dq = ((lane & 3) - (lane >> 4)) & 3;
    const uint8_t *gsrc[4];
    int lim[4];  // the piece exists at stream offset o (a multiple of 64) iff o < lim
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = m * 16 + (lane >> 2);
        const uint32_t lo = __shfl((uint32_t)base, r, 64);
        const uint32_t hi = __shfl((uint32_t)(base >> 32), r, 64);
        gsrc[m] = a.plane + ((((uint64_t)hi << 32) | lo) + (uint64_t)(dq * 16));
        lim[m] = ((__shfl(L, r, 64) + 15) & ~15) - dq * 16;
    }
    auto issue_dma = [&](int c) {  // chunk c (stream bytes [128 c, 128 c + 128)) -> the slot, whose previous content has been read
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 4; ++m) {  // the two halves of a line back to back
            const int o = c * 128;
            if (o < lim[m]) dma16(gsrc[m] + o, slot_lds + m * 1024);
            if (o + 64 < lim[m]) dma16(gsrc[m] + o + 64, slot_lds + 4096 + m * 1024);
        }
    };

    uint32_t ring[R][4];
#pragma unroll
    for (int i = 0; i < R; ++i) ring[i][0] = ring[i][1] = ring[i][2] = ring[i][3] = 0;
    double s = 0.0, w = 0.0, mn = 0.0;
    uint32_t bad = 0;

    // round r (64 bases): the 4 pieces of half r & 1 of the slot -> ring[t0 .. t0 + 3]; after the second half the slot is
    // free and the next chunk's DMA goes out, a whole round of compute ahead of its first use
    auto load_round = [&](int t0, int r) {
        const int half = r & 1;
        if (half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned char *row = my_row + half * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + (((k + rot) & 3) << 4));
            ring[t0 + k][0] = v.x; ring[t0 + k][1] = v.y; ring[t0 + k][2] = v.z; ring[t0 + k][3] = v.w;
            if (PRIV) bad |= v.x | v.y | v.z | v.w;
        }
        if (half == 1 && r + 1 < n_rounds) issue_dma((r + 1) >> 1);
    };
    // addresses of the 4 bases of dword d; lanes whose read has ended look up the zero entry instead (exact no-op)
    auto addr4 = [&](uint32_t x, bool masked, int rem, int k0, uint32_t (&o)[4]) {
        o[0] = tab_addr<PRIV, 0>(x, laneoff);
        o[1] = tab_addr<PRIV, 1>(x, laneoff);
        o[2] = tab_addr<PRIV, 2>(x, laneoff);
        o[3] = tab_addr<PRIV, 3>(x, laneoff);
        if (__builtin_expect(masked, 0)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (k0 + i) < rem ? o[i] : zaddr;
        }
    };
    auto head_piece = [&](const uint32_t (&lw)[4], int Tp) {  // all 16 positions < window_size
        const bool masked = 16 * (Tp + 1) > Lmin;
        const int rem = L - 16 * Tp;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t aj[4];
            addr4(lw[d], masked, rem, 4 * d, aj);
            head4<T::QOFF>(aj[0], aj[1], aj[2], aj[3], s);
        }
    };
    auto steady_piece = [&](const uint32_t (&lw)[4], const uint32_t (&p0)[4], const uint32_t (&p1)[4], int Tp) {
        const bool masked = 16 * (Tp + 1) > Lmin;
        const int rem = L - 16 * Tp;
        uint32_t tw[4];
        funnel(p0, p1, fD, fsh, tw);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t aj[4], ai[4];
            addr4(lw[d], masked, rem, 4 * d, aj);
            addr4(tw[d], masked, rem, 4 * d, ai);
            fold4<T::QOFF, T::DOFF>(aj[0], aj[1], aj[2], aj[3], ai[0], ai[1], ai[2], ai[3], s, w, mn);
        }
    };
    auto boundary_piece = [&](const uint32_t (&lw)[4], const uint32_t (&p0)[4], const uint32_t (&p1)[4], int Tp) {
        // piece A: positions 16 A + k; k < B belongs to the first window, k == B - 1 completes it (src/read.cpp:219-224)
        const int rem = L - 16 * Tp;
        uint32_t tw[4];
        funnel(p0, p1, fD, fsh, tw);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t aj[4], ai[4];
            addr4(lw[d], true, rem, 4 * d, aj);
            addr4(tw[d], true, rem, 4 * d, ai);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 4 * d + i;
                if (k < B) head1<T::QOFF>(aj[i], s);
                else fold1<T::QOFF, T::DOFF>(aj[i], ai[i], s, w, mn);
                if (k == B - 1) {
                    w = s / ws_d;
                    mn = w;
                }
            }
        }
    };

    issue_dma(0);

    // ---- prologue: rounds 0 .. H-1 hold pieces 0 .. A (static piece numbers) ------------------------------
#pragma unroll
    for (int h = 0; h < H; ++h) {
        if (h >= n_rounds) goto done;
        load_round(4 * h, h);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            constexpr int dummy = 0;
            (void)dummy;
            const int Tp = 4 * h + k;  // compile-time after unrolling
            if (16 * Tp >= Lmax) continue;
            if (Tp < A) {
                head_piece(ring[Tp], Tp);
                if (Tp == A - 1 && B == 0) {  // window_size is a multiple of 16: the first window ends with this piece
                    w = s / ws_d;
                    mn = w;
                }
            } else if (Tp == A) {
                boundary_piece(ring[Tp], ring[(Tp + R - A - 1) % R], ring[(Tp + R - A) % R], Tp);
            } else {
                steady_piece(ring[Tp], ring[(Tp + R - A - 1) % R], ring[(Tp + R - A) % R], Tp);
            }
        }
    }
    // ---- main loop: one ring revolution per iteration, starting at ring round H % RR -------------------------
    for (int rb = H;; rb += RR) {
#pragma unroll
        for (int u = 0; u < RR; ++u) {
            const int r = rb + u;
            if (r >= n_rounds) goto done;
            const int sr = (H + u) % RR;  // ring round (compile-time)
            load_round(4 * sr, r);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * sr + k;  // ring piece (compile-time)
                const int Tp = 4 * r + k;  // stream piece
                if (16 * Tp >= Lmax) break;
                steady_piece(ring[t], ring[(t + R - A - 1) % R], ring[(t + R - A) % R], Tp);
            }
        }
    }
done:
    if (live) {
        finish_read(a, rid, L, s, mn);
        if (PRIV && (bad & 0x80808080u)) a.redo_list[atomicAdd(a.redo_count, 1u)] = rid;
    }
  }
}

#if FLX_REGS_PART == 0
// ---------------------------------------------------------------------------------------------------------------------
// stream kernel: ANY window size.  One lane per read again, but both edges of the window come straight from global memory
// with 16-byte loads per lane: the leading piece [16 t, 16 t + 16) and the aligned piece that completes the trailing edge
// [16 t - ws, 16 t - ws + 16) (its first part is the piece loaded one step earlier).  The trailing stream lags `ws` bytes
// behind the leading one, so it is served by the L2 / Infinity Cache, not by HBM.  64 lanes = 64 lines per load
// instruction: this is bound by the address path (~2.7x slower than the register-history kernel at ws = 250), but it has no
// per-read on-chip state at all, so it replaces the byte-wise direct kernel wherever the LDS ring does not fit
// (ws > ~2000: 283 -> 21 ms per 1e10 bases at ws = 2500, profiles/r02_microbench.txt).  Plain tables, 16 waves per CU.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) flx_score_phred_stream(const PhredArgs a) {
    using T = Tab<false>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 257; i += 1024) {
        *reinterpret_cast<double *>(smem + T::QOFF + i * 8) = a.lut_q[i];
        *reinterpret_cast<double *>(smem + T::DOFF + i * 8) = a.lut_d[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t zaddr = (uint32_t)(T::ZIDX * T::ROW);
    const int ws = a.ws;
    const int A = ws >> 4, B = ws & 15;
    const int fD = (16 - B) >> 2;
    const uint32_t fsh = (uint32_t)(16 - B) & 3u;
    const double ws_d = a.ws_d;
    for (;;) {
        unsigned int group = 0;
        if (lane == 0) group = atomicAdd(a.ticket, 1u);
        group = (unsigned int)__builtin_amdgcn_readfirstlane((int)group);
        if (group >= a.n_groups) break;
        const uint64_t gslot = (uint64_t)group * 64 + lane;
        const bool live = gslot < a.n_reads;
        uint32_t rid = 0;
        int L = 0;
        uint64_t base = 0;
        if (live) {
            rid = a.order ? a.order[gslot] : (uint32_t)gslot;
            L = a.lengths[rid];
            base = a.offsets[rid];
        }
        const int Lmax = wave_max(L);
        if (Lmax == 0) {
            if (live) finish_read(a, rid, L, 0.0, 0.0);
            continue;
        }
        const uint8_t *src = (L > 0) ? a.plane + base : a.plane;
        const int maxoff = max(((L + 15) & ~15) - 16, 0);  // loads past the end of a read are clamped into it (never used: masked)
        auto ld16 = [&](int off) { return *reinterpret_cast<const uint4 *>(src + min(max(off, 0), maxoff)); };
        const int n_pieces = (Lmax + 15) >> 4;
        double s = 0.0, w = 0.0, mn = 0.0;
        uint4 lead = ld16(0), lead_next = ld16(16);
        uint4 tr_prev = make_uint4(0, 0, 0, 0), tr_cur = tr_prev, tr_next = ld16(-16 * A);  // piece -A (clamped to piece 0)
        // aligned trailing pieces: step t needs pieces t-A-1 and t-A of the stream; they exist from t = A on (piece 0)
        for (int t = 0; t < n_pieces; ++t) {
            const uint4 lw4 = lead;
            lead = lead_next;
            lead_next = ld16(16 * (t + 2));
            tr_prev = tr_cur;   // piece t-A-1
            tr_cur = tr_next;   // piece t-A
            tr_next = ld16(16 * (t + 1 - A));  // piece t+1-A (clamped to the read while t+1 < A: unused)
            const uint32_t lw[4] = {lw4.x, lw4.y, lw4.z, lw4.w};
            const int rem = L - 16 * t;
            if (t < A) {  // all 16 positions < window_size: only the running sum
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t aj[4];
                    aj[0] = tab_addr<false, 0>(lw[d], 0); aj[1] = tab_addr<false, 1>(lw[d], 0);
                    aj[2] = tab_addr<false, 2>(lw[d], 0); aj[3] = tab_addr<false, 3>(lw[d], 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) aj[i] = (4 * d + i) < rem ? aj[i] : zaddr;
                    head4<T::QOFF>(aj[0], aj[1], aj[2], aj[3], s);
                }
                if (t == A - 1 && B == 0) {
                    w = s / ws_d;
                    mn = w;
                }
                continue;
            }
            const uint32_t p0[4] = {tr_prev.x, tr_prev.y, tr_prev.z, tr_prev.w};
            const uint32_t p1[4] = {tr_cur.x, tr_cur.y, tr_cur.z, tr_cur.w};
            uint32_t tw[4];
            funnel(p0, p1, fD, fsh, tw);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t aj[4], ai[4];
                aj[0] = tab_addr<false, 0>(lw[d], 0); aj[1] = tab_addr<false, 1>(lw[d], 0);
                aj[2] = tab_addr<false, 2>(lw[d], 0); aj[3] = tab_addr<false, 3>(lw[d], 0);
                ai[0] = tab_addr<false, 0>(tw[d], 0); ai[1] = tab_addr<false, 1>(tw[d], 0);
                ai[2] = tab_addr<false, 2>(tw[d], 0); ai[3] = tab_addr<false, 3>(tw[d], 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool act = (4 * d + i) < rem;
                    aj[i] = act ? aj[i] : zaddr;
                    ai[i] = act ? ai[i] : zaddr;
                }
                if (t > A) {
                    fold4<T::QOFF, T::DOFF>(aj[0], aj[1], aj[2], aj[3], ai[0], ai[1], ai[2], ai[3], s, w, mn);
                } else {  // the piece that holds position window_size (src/read.cpp:219-224)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = 4 * d + i;
                        if (k < B) head1<T::QOFF>(aj[i], s);
                        else fold1<T::QOFF, T::DOFF>(aj[i], ai[i], s, w, mn);
                        if (k == B - 1) {
                            w = s / ws_d;
                            mn = w;
                        }
                    }
                }
            }
        }
        if (live) finish_read(a, rid, L, s, mn);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dual-slot kernel (round 4): ANY window size at one price.  The register-history kernel keeps the last window_size bytes of
// every read on chip, which costs occupancy from ws = 640 on (one wave per SIMD: 0.21-0.23 of the roofline) and stops at 1007;
// the LDS ring behind it falls to 0.16 and the stream kernel to 0.06.  Here NOTHING of the window stays on chip: the trailing
// edge is a second LDS-DMA stream over the same bytes, `ws` behind the leading one (served by the L2 / Infinity Cache, the
// plane still crosses the HBM once).  Per wave: the leading slot of the register-history kernel (one 128-byte chunk of each
// of its 64 reads, 8 KiB) and a ring of four 4-KiB half slots for the trailing stream (64 bytes of every read each): the two
// halves a round of 64 bases reads from, and the next two already on their way.  24 KiB per wave -> 6 waves per CU whatever
// the window.  Same arithmetic, same table gathers, plain tables.
//
// DMA bookkeeping (vmcnt counts DMA instructions in order): every round issues its trailing half FIRST (4 instructions,
// always — a half that does not exist reads offset 0 and is never used), every odd round ends with the next leading chunk
// (8 instructions).  An even round needs the chunk issued at the end of the round before it: all but the 4 youngest
// instructions done.  An odd round needs the half issued two rounds earlier: all but the 8 youngest.
// ---------------------------------------------------------------------------------------------------------------------
// The trailing stream is read ONE aligned half slot (four 16-byte pieces of every read) per round; which of the five trailing
// pieces a round needs come from this half and which from the one before it depends on (ws / 16) mod 4 — a template parameter,
// so that the pieces kept from the previous round are statically indexed registers.  The ring therefore has only TWO half
// slots (the one being read and the one on its way): 16 KiB per wave, 9 waves per CU.  (Round 4's first form read the five
// pieces straight out of a ring of three / four halves: 7 / 6 waves, 4.8 / 5.3 ms per 1e10 bases.)
constexpr int DUAL_WAVES = 9;
template <int AMOD>
__global__ void __launch_bounds__(DUAL_WAVES * 64) flx_score_phred_dual(const PhredArgs a) {
    using T = Tab<false>;
    constexpr int LEAD_BYTES = 8192, TRAIL_BYTES = 8192, WAVE_BYTES = LEAD_BYTES + TRAIL_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 257; i += DUAL_WAVES * 64) {
        *reinterpret_cast<double *>(smem + T::QOFF + i * 8) = a.lut_q[i];
        *reinterpret_cast<double *>(smem + T::DOFF + i * 8) = a.lut_d[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned char *lead_slot = smem + T::SLOT0 + wave * WAVE_BYTES;
    unsigned char *trail_ring = lead_slot + LEAD_BYTES;
    const uint32_t lead_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(T::SLOT0 + wave * WAVE_BYTES));
    const uint32_t trail_lds = lead_lds + LEAD_BYTES;
    const int rot = lane >> 2;
    const uint32_t zaddr = (uint32_t)(T::ZIDX * T::ROW);
    const int ws = a.ws;
    const int A = ws >> 4, B = ws & 15;
    const int A4 = A >> 2;  // round r reads trailing half r - A4 (pieces 4 (r - A4) .. + 3) and issues the next one
    const int fD = (16 - B) >> 2;
    const uint32_t fsh = (uint32_t)(16 - B) & 3u;
    const double ws_d = a.ws_d;

    for (;;) {
        unsigned int group = 0;
        if (lane == 0) group = atomicAdd(a.ticket, 1u);
        group = (unsigned int)__builtin_amdgcn_readfirstlane((int)group);
        if (group >= a.n_groups) break;
        const uint64_t gslot = (uint64_t)group * 64 + lane;
        const bool live = gslot < a.n_reads;
        uint32_t rid = 0;
        int L = 0;
        uint64_t base = 0;
        if (live) {
            rid = a.order ? a.order[gslot] : (uint32_t)gslot;
            L = a.lengths[rid];
            base = a.offsets[rid];
        }
        const int Lmax = wave_max(L);
        const int Lmin = wave_min(L);
        if (Lmax == 0) {
            if (live) finish_read(a, rid, L, 0.0, 0.0);
            continue;
        }
        const int n_rounds = (Lmax + 63) >> 6;
        // DMA map of this lane (as in the register-history kernel): for m = 0..3 piece `dq` of read 16 m + (lane >> 2)
        const int dq = ((lane & 3) - (lane >> 4)) & 3;
        const uint8_t *gsrc[4];
        int lim[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = m * 16 + (lane >> 2);
            const uint32_t lo = __shfl((uint32_t)base, r, 64);
            const uint32_t hi = __shfl((uint32_t)(base >> 32), r, 64);
            const int Lr = __shfl(L, r, 64);
            // a read without bytes (or no read at all) points at the start of the plane: every DMA instruction is issued by every
            // lane, so that the instruction counts behind the vmcnt waits hold
            gsrc[m] = Lr > 0 ? a.plane + ((((uint64_t)hi << 32) | lo) + (uint64_t)(dq * 16)) : a.plane;
            lim[m] = Lr > 0 ? ((Lr + 15) & ~15) - dq * 16 : 0;
        }
        auto lds_idle = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        auto issue_lead = [&](int c) {  // 8 instructions
            lds_idle();
            const int o = c * 128;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                dma16(gsrc[m] + (o < lim[m] ? o : 0), lead_lds + m * 1024);
                dma16(gsrc[m] + (o + 64 < lim[m] ? o + 64 : 0), lead_lds + 4096 + m * 1024);
            }
        };
        auto issue_trail = [&](int h) {  // 4 instructions; h < 0 or behind the reads: a harmless fetch of offset 0
            lds_idle();
            const int o = h * 64;
            const uint32_t dst = trail_lds + (uint32_t)(h & 1) * 4096u;
#pragma unroll
            for (int m = 0; m < 4; ++m) dma16(gsrc[m] + ((o >= 0 && o < lim[m]) ? o : 0), dst + m * 1024);
        };

        double s = 0.0, w = 0.0, mn = 0.0;
        uint32_t prev[4][4];  // the trailing half in front of the round's (pieces 4 (h - 1) .. 4 h - 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) prev[k][0] = prev[k][1] = prev[k][2] = prev[k][3] = 0;
        issue_lead(0);
        if (A4 == 0) issue_trail(0);
        for (int r = 0; r < n_rounds; ++r) {
            const int h = r - A4;
            issue_trail(h + 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything but the half just asked for: this round's chunk and half
            uint32_t lw[4][4], cur[4][4];
            const unsigned char *lrow = lead_slot + (r & 1) * 4096 + lane * 64;
            const unsigned char *trow = trail_ring + (h & 1) * 4096 + lane * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 v = *reinterpret_cast<const uint4 *>(lrow + (((k + rot) & 3) << 4));
                lw[k][0] = v.x; lw[k][1] = v.y; lw[k][2] = v.z; lw[k][3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (h >= 0) v = *reinterpret_cast<const uint4 *>(trow + (((k + rot) & 3) << 4));
                cur[k][0] = v.x; cur[k][1] = v.y; cur[k][2] = v.z; cur[k][3] = v.w;
            }
            if ((r & 1) && r + 1 < n_rounds) issue_lead((r + 1) >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int Tp = 4 * r + k;
                if (16 * Tp >= Lmax) break;
                const bool masked = 16 * (Tp + 1) > Lmin;
                const int rem = L - 16 * Tp;
                if (Tp < A) {  // all 16 positions < window_size: only the running sum
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        uint32_t aj[4];
                        aj[0] = tab_addr<false, 0>(lw[k][d], 0); aj[1] = tab_addr<false, 1>(lw[k][d], 0);
                        aj[2] = tab_addr<false, 2>(lw[k][d], 0); aj[3] = tab_addr<false, 3>(lw[k][d], 0);
                        if (masked) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) aj[i] = (4 * d + i) < rem ? aj[i] : zaddr;
                        }
                        head4<T::QOFF>(aj[0], aj[1], aj[2], aj[3], s);
                    }
                    if (Tp == A - 1 && B == 0) {
                        w = s / ws_d;
                        mn = w;
                    }
                    continue;
                }
                // trailing pieces Tp - A - 1 and Tp - A = pieces 4 h - AMOD - 1 + k and the next: index i = k, k + 1 of the five
                // the round needs; i <= AMOD comes from the previous half (its piece 3 - AMOD + i), the rest from this one
                uint32_t tw[4];
                {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const uint32_t(&p0)[4] = (k <= AMOD) ? prev[3 - AMOD + k] : cur[k - AMOD - 1];
                    const uint32_t(&p1)[4] = (k + 1 <= AMOD) ? prev[3 - AMOD + k + 1] : cur[k - AMOD];
                    funnel(p0, p1, fD, fsh, tw);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t aj[4], ai[4];
                    aj[0] = tab_addr<false, 0>(lw[k][d], 0); aj[1] = tab_addr<false, 1>(lw[k][d], 0);
                    aj[2] = tab_addr<false, 2>(lw[k][d], 0); aj[3] = tab_addr<false, 3>(lw[k][d], 0);
                    ai[0] = tab_addr<false, 0>(tw[d], 0); ai[1] = tab_addr<false, 1>(tw[d], 0);
                    ai[2] = tab_addr<false, 2>(tw[d], 0); ai[3] = tab_addr<false, 3>(tw[d], 0);
                    if (masked || Tp == A) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool act = (4 * d + i) < rem;
                            aj[i] = act ? aj[i] : zaddr;
                            ai[i] = act ? ai[i] : zaddr;
                        }
                    }
                    if (Tp > A) {
                        fold4<T::QOFF, T::DOFF>(aj[0], aj[1], aj[2], aj[3], ai[0], ai[1], ai[2], ai[3], s, w, mn);
                    } else {  // the piece that holds position window_size (src/read.cpp:219-224)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int kk = 4 * d + i;
                            if (kk < B) head1<T::QOFF>(aj[i], s);
                            else fold1<T::QOFF, T::DOFF>(aj[i], ai[i], s, w, mn);
                            if (kk == B - 1) {
                                w = s / ws_d;
                                mn = w;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                prev[k][0] = cur[k][0]; prev[k][1] = cur[k][1]; prev[k][2] = cur[k][2]; prev[k][3] = cur[k][3];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last rounds' fetches must not land in the next group's slots)
        if (live) finish_read(a, rid, L, s, mn);
    }
}

// reads flagged by the bank-private variant (a byte >= 128): exact re-scoring, one lane per read
__global__ void __launch_bounds__(256) flx_score_phred_redo(const PhredArgs a) {
    __shared__ double lq[LUT_PAD];
    __shared__ double ld[LUT_PAD];
    for (int i = threadIdx.x; i < 257; i += 256) {
        lq[i] = a.lut_q[i];
        ld[i] = a.lut_d[i];
    }
    __syncthreads();
    const unsigned int n = *a.redo_count;
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t rid = a.redo_list[i];
        const int L = a.lengths[rid];
        const uint8_t *q = a.plane + a.offsets[rid];
        const int ws = a.ws;
        double s = 0.0, w = 0.0, mn = 0.0;
        const int head = L < ws ? L : ws;
        for (int j = 0; j < head; ++j) s += lq[q[j]];
        if (L > ws) {
            w = s / a.ws_d;
            mn = w;
            for (int j = ws; j < L; ++j) {
                const uint32_t cj = q[j], ci = q[j - ws];
                s += lq[cj];
                w -= ld[ci];
                w += ld[cj];
                if (w < mn) mn = w;
            }
        }
        finish_read(a, rid, L, s, mn);
    }
}

#endif

template <int A, bool PRIV, int WAVES>
int launch_one(flx_ctx *ctx, PhredArgs &a) {
    using T = Tab<PRIV>;
    auto kern = flx_score_phred_regs<A, PRIV, WAVES>;
    // one persistent workgroup per CU; asking for more than half of the LDS keeps a second one off the CU
    const size_t lds = std::max<size_t>((size_t)T::SLOT0 + (size_t)WAVES * 8192, 84 * 1024);
    FLX_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t per_block = (uint64_t)WAVES;
    const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)a.n_groups + per_block - 1) / per_block,
                                                       (uint64_t)ctx->prop.multiProcessorCount);
    ctx->last_phred_kernel = PRIV ? "flx_score_phred_regs_private" : "flx_score_phred_regs";
    flx_time_begin(ctx, ctx->last_phred_kernel);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, ctx->stream, a);
    flx_time_end(ctx);
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}

// waves per CU: 16 (4 per SIMD, <= 128 VGPRs) for rings of up to 12 pieces, 12 (<= 168 VGPRs) up to 20 pieces, 8 (<= 256) up to
// 43 pieces, 4 beyond (one wave per SIMD: the ring spills into the accumulator half of the unified register file, up to
// 256 + 102 registers at ws = 1007); the bank-private tables (66 KB) leave room for 11 slots of 8 KiB
template <int A, bool PRIV>
struct WavesFor {
    static constexpr int plain = A <= 7 ? 16 : A <= 15 ? 12 : 8;
    static constexpr int value = PRIV && plain > 11 ? 11 : plain;
};

template <int A>
int launch_a(flx_ctx *ctx, PhredArgs &a, bool priv) {
    if constexpr (A >= 32) {  // big rings: plain tables only (each instantiation takes ~30 s to compile)
        return launch_one<A, false, WavesFor<A, false>::value>(ctx, a);
    } else {
        return priv ? launch_one<A, true, WavesFor<A, true>::value>(ctx, a) : launch_one<A, false, WavesFor<A, false>::value>(ctx, a);
    }
}

}  // namespace

// The instantiations (window sizes 1..1007, A = ws / 16 = 0..62) are spread over fourteen translation units of this same file
// (-DFLX_REGS_PART=0..13, see the Makefile) so that they compile in parallel.  Rings beyond 36 pieces (A >= 32) need
// -mllvm -unroll-max-upperbound (the prologue loop has an early exit: LLVM unrolls such loops only up to 8 iterations by
// default, and a ring that is not indexed statically everywhere ends up in scratch memory) and larger unroll thresholds.
#define FLX_REGS_CASE(AA) \
    case AA:              \
        *launched = true; \
        return launch_a<AA>(ctx, a, priv);
#define FLX_REGS_CAT2(a, b) a##b
#define FLX_REGS_CAT(a, b) FLX_REGS_CAT2(a, b)
#define FLX_REGS_WIDE_NAME FLX_REGS_CAT(flx_launch_score_phred_regs_part, FLX_REGS_PART)
#if FLX_REGS_PART == 0
int flx_launch_score_phred_regs_part0(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    switch (a.ws / 16) { FLX_REGS_CASE(0) FLX_REGS_CASE(1) FLX_REGS_CASE(2) FLX_REGS_CASE(3) FLX_REGS_CASE(4) FLX_REGS_CASE(5) FLX_REGS_CASE(6) FLX_REGS_CASE(7) default: return FLX_OK; }
}
#elif FLX_REGS_PART == 1
int flx_launch_score_phred_regs_part1(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    switch (a.ws / 16) { FLX_REGS_CASE(8) FLX_REGS_CASE(9) FLX_REGS_CASE(10) FLX_REGS_CASE(11) FLX_REGS_CASE(12) default: return FLX_OK; }
}
#elif FLX_REGS_PART == 2
int flx_launch_score_phred_regs_part2(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    switch (a.ws / 16) { FLX_REGS_CASE(13) FLX_REGS_CASE(14) FLX_REGS_CASE(15) FLX_REGS_CASE(16) FLX_REGS_CASE(17) default: return FLX_OK; }
}
#elif FLX_REGS_PART == 3
int flx_launch_score_phred_regs_part3(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    switch (a.ws / 16) { FLX_REGS_CASE(18) FLX_REGS_CASE(19) FLX_REGS_CASE(20) FLX_REGS_CASE(21) FLX_REGS_CASE(22) default: return FLX_OK; }
}
#elif FLX_REGS_PART == 4
int flx_launch_score_phred_regs_part4(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    switch (a.ws / 16) { FLX_REGS_CASE(23) FLX_REGS_CASE(24) FLX_REGS_CASE(25) FLX_REGS_CASE(26) FLX_REGS_CASE(27) default: return FLX_OK; }
}
#elif FLX_REGS_PART == 5
int flx_launch_score_phred_regs_part5(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    switch (a.ws / 16) { FLX_REGS_CASE(28) FLX_REGS_CASE(29) FLX_REGS_CASE(30) FLX_REGS_CASE(31) default: return FLX_OK; }
}
#else
// parts 6 and 7: A = 32..35 and 36..38: window sizes 512..623 (beyond them a ring leaves room for one wave per SIMD only, and the
// dual-slot kernel is faster)
int FLX_REGS_WIDE_NAME(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched) {
    constexpr int A0 = 32 + 4 * (FLX_REGS_PART - 6);
    switch (a.ws / 16) {
        FLX_REGS_CASE(A0) FLX_REGS_CASE(A0 + 1) FLX_REGS_CASE(A0 + 2)
#if FLX_REGS_PART < 7
        FLX_REGS_CASE(A0 + 3)
#endif
        default: return FLX_OK;
    }
}
#endif
#undef FLX_REGS_CASE

#if FLX_REGS_PART == 0
int flx_launch_score_phred_regs_part1(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched);
int flx_launch_score_phred_regs_part2(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched);
int flx_launch_score_phred_regs_part3(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched);
int flx_launch_score_phred_regs_part4(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched);
int flx_launch_score_phred_regs_part5(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched);
#define FLX_REGS_DECL(P) int flx_launch_score_phred_regs_part##P(flx_ctx *ctx, PhredArgs &a, bool priv, bool *launched);
FLX_REGS_DECL(6) FLX_REGS_DECL(7)
#undef FLX_REGS_DECL

namespace {
// Which table layout?  Plain tables are ~6 % faster when a wavefront's quality values stay within ~32 consecutive table
// entries; bank-private tables cost the same whatever the data is and win (by up to ~11 %) on a wide quality range, where
// plain entries e and e + 32 collide.  16 K sampled bytes give the byte distribution p; C(32,2) * sum over bank pairs of
// p_e * p_e' (e != e', e = e' mod 32) is the expected number of conflicting lane pairs per 32-lane gather group.
__global__ void __launch_bounds__(256) flx_phred_sample(const PhredArgs a, unsigned int *hist) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t z = (t + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    const uint64_t r = z % a.n_reads;
    const int L = a.lengths[r];
    if (L <= 0) return;
    const uint32_t pos = (uint32_t)((z >> 32) % (uint64_t)L);
    atomicAdd(&hist[a.plane[a.offsets[r] + pos]], 1u);
}
}  // namespace

int flx_launch_score_phred_stream(flx_ctx *ctx, PhredArgs a) {
    void *scr;
    FLX_CHECK(flx_scratch(ctx, 64, &scr));
    FLX_HIP(ctx, hipMemsetAsync(scr, 0, 8, ctx->stream));
    a.ticket = (unsigned int *)scr;
    a.n_groups = (unsigned int)((a.n_reads + 63) / 64);
    const size_t lds = 84 * 1024;  // one persistent workgroup of 16 waves per CU
    FLX_HIP(ctx, hipFuncSetAttribute((const void *)flx_score_phred_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)a.n_groups + 15) / 16, (uint64_t)ctx->prop.multiProcessorCount);
    ctx->last_phred_kernel = "flx_score_phred_stream";
    flx_time_begin(ctx, ctx->last_phred_kernel);
    hipLaunchKernelGGL(flx_score_phred_stream, dim3(grid), dim3(1024), lds, ctx->stream, a);
    flx_time_end(ctx);
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}

int flx_launch_score_phred_dual(flx_ctx *ctx, PhredArgs a) {
    void *scr;
    FLX_CHECK(flx_scratch(ctx, 64, &scr));
    FLX_HIP(ctx, hipMemsetAsync(scr, 0, 8, ctx->stream));
    a.ticket = (unsigned int *)scr;
    a.n_groups = (unsigned int)((a.n_reads + 63) / 64);
    const size_t lds = (size_t)Tab<false>::SLOT0 + (size_t)DUAL_WAVES * 16384;  // more than half of the LDS: one workgroup per CU
    const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)a.n_groups + DUAL_WAVES - 1) / DUAL_WAVES, (uint64_t)ctx->prop.multiProcessorCount);
    ctx->last_phred_kernel = "flx_score_phred_dual";
    flx_time_begin(ctx, ctx->last_phred_kernel);
#define FLX_DUAL_LAUNCH(M)                                                                                                              \
    case M:                                                                                                                             \
        FLX_HIP(ctx, hipFuncSetAttribute((const void *)flx_score_phred_dual<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(flx_score_phred_dual<M>, dim3(grid), dim3(DUAL_WAVES * 64), lds, ctx->stream, a);                            \
        break;
    switch ((a.ws >> 4) & 3) {
        FLX_DUAL_LAUNCH(0) FLX_DUAL_LAUNCH(1) FLX_DUAL_LAUNCH(2) FLX_DUAL_LAUNCH(3)
    }
#undef FLX_DUAL_LAUNCH
    flx_time_end(ctx);
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}

int flx_launch_score_phred_regs(flx_ctx *ctx, PhredArgs a, bool *launched) {
    *launched = false;
    const int A = a.ws / 16;
    if (A > 38) return FLX_OK;  // the register-history kernel serves window sizes 1 .. 623 (A = ws / 16 = 0 .. 38); beyond: the dual-slot kernel
    const char *env = getenv("FLX_PHRED_TABLES");  // "plain" | "private" | unset = decide from a sample of the data
    bool priv = env && strcmp(env, "private") == 0;
    // scratch: [0,4) ticket, [4,8) redo count, [64, 1088) sample histogram, [2048, 2048 + 4 n) redo list
    void *scr;
    FLX_CHECK(flx_scratch(ctx, 2048 + a.n_reads * 4, &scr));
    a.ticket = (unsigned int *)scr;
    a.redo_count = (unsigned int *)scr + 1;
    a.redo_list = (uint32_t *)((char *)scr + 2048);
    a.n_groups = (unsigned int)((a.n_reads + 63) / 64);
    FLX_HIP(ctx, hipMemsetAsync(scr, 0, 2048, ctx->stream));
    if (!env || (strcmp(env, "private") != 0 && strcmp(env, "plain") != 0)) {
        unsigned int h[256];
        unsigned int *d_hist = (unsigned int *)((char *)scr + 64);
        hipLaunchKernelGGL(flx_phred_sample, dim3(64), dim3(256), 0, ctx->stream, a, d_hist);
        FLX_HIP(ctx, hipMemcpyAsync(h, d_hist, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double tot = 0.0, conflicts = 0.0;
        for (int i = 0; i < 256; ++i) tot += h[i];
        bool high = false;
        for (int i = 128; i < 256; ++i) high = high || h[i] != 0;
        if (tot > 0) {
            for (int b = 0; b < 32; ++b) {
                double pb = 0.0, sq = 0.0;
                for (int e = b; e < 256; e += 32) {
                    const double pe = h[e] / tot;
                    pb += pe;
                    sq += pe * pe;
                }
                conflicts += pb * pb - sq;
            }
        }
        priv = !high && 496.0 * conflicts > 3.0;  // bytes >= 128 would all go through the redo path: stay plain
    }
    if (A <= 7) FLX_CHECK(flx_launch_score_phred_regs_part0(ctx, a, priv, launched));
    else if (A <= 12) FLX_CHECK(flx_launch_score_phred_regs_part1(ctx, a, priv, launched));
    else if (A <= 17) FLX_CHECK(flx_launch_score_phred_regs_part2(ctx, a, priv, launched));
    else if (A <= 22) FLX_CHECK(flx_launch_score_phred_regs_part3(ctx, a, priv, launched));
    else if (A <= 27) FLX_CHECK(flx_launch_score_phred_regs_part4(ctx, a, priv, launched));
    else if (A <= 31) FLX_CHECK(flx_launch_score_phred_regs_part5(ctx, a, priv, launched));
    else {
        typedef int (*part_fn)(flx_ctx *, PhredArgs &, bool, bool *);
        static const part_fn wide[2] = {flx_launch_score_phred_regs_part6, flx_launch_score_phred_regs_part7};
        priv = false;  // big rings come with plain tables only
        FLX_CHECK(wide[(A - 32) / 4](ctx, a, priv, launched));
    }
    if (*launched && priv) {
        flx_time_begin(ctx, "flx_score_phred_redo");
        hipLaunchKernelGGL(flx_score_phred_redo, dim3(256), dim3(256), 0, ctx->stream, a);
        flx_time_end(ctx);
        FLX_HIP(ctx, hipGetLastError());
    }
    return FLX_OK;
}
#endif
