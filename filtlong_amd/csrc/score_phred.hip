// score_phred.hip — Phred-only per-read scoring on gfx950.
//
// Replaces, for a whole batch of reads, the Phred branch of the reference's Read::Read
// (src/read.cpp:35-39), get_mean_quality (208-213), get_window_quality (216-236) and the hard
// cut-offs (64-73).  Results are bit-identical to the reference: the two FP64 recurrences are
// order-dependent (the drift is part of the result), so each read is folded strictly
// left to right by ONE lane; the parallelism is across reads (64 reads per wavefront).
//
// Data movement (HBM-bound part): the packed quality plane is read exactly once, with 16-byte
// loads.  A wavefront serves 64 reads; per round it fetches CH contiguous bytes of each of its reads
// (lane -> (read, 16-byte piece) mapping, so every load instruction covers 16 reads x 64 B) and
// transposes them through a per-read LDS ring so that each lane can then stream its own read.  The
// ring also keeps the last `window_size` bytes of every read, which is what the trailing edge of the
// sliding window re-reads — the plane is never fetched twice.
//
// Arithmetic: per base one 8-byte LDS lookup of Q[c] = 1 - 10^(-(c-33)/10) and two of D[c] = Q[c]/ws
// (tables built on the host with the host libm; see flx_ctx.hip) and four dependent FP64 ops
// (sum += Q; w -= D_old; w += D_new; min).  No pow, no division, no 8 B/base quality vector.
#include <algorithm>

#include "flx_internal.h"
#include "score_phred_common.h"

using namespace flx_phred;

namespace {

#ifndef FLX_CH
#define FLX_CH 64
#endif
constexpr int CH = FLX_CH;    // bytes staged per read per round
constexpr int PPR = CH / 16;  // 16-byte pieces per read per round (= loads per lane per round)

// ---------------------------------------------------------------------------------------------
// ring kernel: the fast path (window_size small enough for the LDS ring)
//
// Ring row of one read: NS slots of CH bytes (NS = ceil(ws / CH) + 1) followed by a 16-byte mirror of
// the row's first 16 bytes, so that any 20-byte window starting inside the row is contiguous.  The
// row stride is NS*CH + 16 (+16 more when needed to make stride/16 odd: conflict-free b128 rows).
// ---------------------------------------------------------------------------------------------
struct Fold {  // per-lane state of the two recurrences
    double s;   // running sum of Q            (get_mean_quality / first window sum)
    double w;   // current window quality      (src/read.cpp:223-229)
    double mn;  // minimum window quality      (src/read.cpp:230-231)
};

// 16 bases that only feed the running sum (positions < window_size)
__device__ __forceinline__ void head16(const double *lq, const uint32_t (&lw)[4], Fold &f) {
    double qj[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qj[i] = lds_f64(lq, lut_addr(lw[i >> 2], i & 3));
#pragma unroll
    for (int i = 0; i < 16; ++i) f.s += qj[i];
}

// 16 bases in the steady state: every position valid in every lane and >= window_size.
// All table lookups of the piece are issued ahead of the dependent FP64 chain.
//
// Measured on MI355X (tools/gatherbench.hip, tools/ldspeak.hip): a random-address ds_read_b64 costs ~5
// cycles of CU time whatever the occupancy (~100-128 B/clk/CU of gathered data), so three gathers per
// base put the floor of this formulation at ~15 cycles per 64 bases per CU.  Two alternatives were built
// and timed and lost: one 16-byte gather of {Q,D} (ds_read_b128 gathers cost ~10 cycles) and deriving
// D = Q/ws on the VALU with a verified 3-op FMA division (the extra FP64 ops cost more than the gather).
__device__ __forceinline__ void body16(const double *lq, const double *ld, const uint32_t (&lw)[4],
                                       const uint32_t (&tw)[4], Fold &f) {
    uint32_t aj[16], ai[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        aj[i] = lut_addr(lw[i >> 2], i & 3);
        ai[i] = lut_addr(tw[i >> 2], i & 3);
    }
    double qj[16], dj[16], di[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        qj[i] = lds_f64(lq, aj[i]);
        di[i] = lds_f64(ld, ai[i]);
        dj[i] = lds_f64(ld, aj[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        f.s += qj[i];
        f.w -= di[i];            // src/read.cpp:228
        f.w += dj[i];            // src/read.cpp:229
        f.mn = fmin(f.mn, f.w);  // if (w < min) min = w — no NaNs here; the sign of a zero minimum is irrelevant
    }
}

// the 16 trailing bytes that start at ring byte offset `ro` (4-byte granular read + byte funnel)
__device__ __forceinline__ void trail16(const unsigned char *my_row, int ro, uint32_t (&tw)[4]) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(my_row + (ro & ~3));
    const uint32_t x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3], x4 = p[4];
    const uint32_t bsh = (uint32_t)ro & 3u;
    tw[0] = __builtin_amdgcn_alignbyte(x1, x0, bsh);
    tw[1] = __builtin_amdgcn_alignbyte(x2, x1, bsh);
    tw[2] = __builtin_amdgcn_alignbyte(x3, x2, bsh);
    tw[3] = __builtin_amdgcn_alignbyte(x4, x3, bsh);
}

#ifndef FLX_PREFETCH
#define FLX_PREFETCH 3
#endif
constexpr int PF = FLX_PREFETCH;  // rounds of global loads kept in flight per wave (PF * 4 KiB)

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) flx_score_phred_ring(const PhredArgs a) {
    // Static LDS: the two lookup tables.  Static LDS is laid out at compile time, so the table base folds into
    // the ds_read offset field and a lookup costs one address op.  2 * 264 * 8 = 4224 B keeps the dynamic
    // region (the rings) 16-byte aligned.
    __shared__ __attribute__((aligned(16))) double lq[LUT_PAD];
    __shared__ __attribute__((aligned(16))) double ld[LUT_PAD];
    extern __shared__ __attribute__((aligned(16))) unsigned char rings[];

    for (int i = threadIdx.x; i < 257; i += WAVES * 64) {
        lq[i] = a.lut_q[i];
        ld[i] = a.lut_d[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int stride = a.stride;
    const int NS = a.n_slots;
    const int ring_len = NS * CH;  // bytes of ring proper (the mirror lives at [ring_len, ring_len + 16))
    const int ws = a.ws;
    unsigned char *ring = rings + (size_t)wave * 64 * stride;
    unsigned char *my_row = ring + lane * stride;

    // Persistent waves: one workgroup per CU for the whole launch; every wave takes the next group of 64 reads (in
    // processing order, i.e. longest first) from a ticket counter, so a wave that finishes starts its next group at once
    // instead of the CU draining and relaunching a whole workgroup (and the tables are staged once).
  for (;;) {
    unsigned int group = 0;
    if (lane == 0) group = atomicAdd(a.ticket, 1u);
    group = (unsigned int)__builtin_amdgcn_readfirstlane((int)group);
    if (group >= a.n_groups) break;
    const uint64_t slot = (uint64_t)group * 64 + lane;
    const bool live = slot < a.n_reads;
    uint32_t rid = 0;
    int L = 0;
    uint64_t base = 0;
    if (live) {
        rid = a.order ? a.order[slot] : (uint32_t)slot;
        L = a.lengths[rid];
        base = a.offsets[rid];
    }
    const int Lmax = wave_max(L);
    const int Lmin = wave_min(L);
    if (Lmax == 0) {
        if (live) finish_read(a, rid, L, 0.0, 0.0);
        continue;
    }

    // staging map: load m of this lane fetches piece k of read r
    const int k = lane & (PPR - 1);
    uint64_t src_base[PPR];
    int src_l16[PPR];
    int dst_off[PPR];
#pragma unroll
    for (int m = 0; m < PPR; ++m) {
        const int r = m * (64 / PPR) + (lane / PPR);
        const uint32_t lo = __shfl((uint32_t)base, r, 64);
        const uint32_t hi = __shfl((uint32_t)(base >> 32), r, 64);
        src_base[m] = (((uint64_t)hi << 32) | lo) + (uint64_t)(k * 16);
        src_l16[m] = ((__shfl(L, r, 64) + 15) & ~15) - k * 16;  // piece valid at round offset o iff o < src_l16
        dst_off[m] = r * stride + k * 16;
    }

    // Global loads run PF rounds ahead of the compute (PF * 4 KiB in flight per wave): with only ~7 waves per CU
    // the HBM latency has to be covered by depth, not by occupancy.
    uint4 pre[PF][PPR];
    auto issue_loads = [&](int t, uint4 (&buf)[PPR]) {
#pragma unroll
        for (int m = 0; m < PPR; ++m) {
            const int o = t * CH;
            if (o < src_l16[m]) buf[m] = *reinterpret_cast<const uint4 *>(a.plane + src_base[m] + (uint64_t)o);
            else buf[m] = make_uint4(0, 0, 0, 0);
        }
    };
    auto write_ring = [&](int ring_slot, const uint4 (&buf)[PPR]) {
#pragma unroll
        for (int m = 0; m < PPR; ++m) *reinterpret_cast<uint4 *>(ring + dst_off[m] + ring_slot * CH) = buf[m];
        if (ring_slot == 0 && k == 0) {  // mirror of the row's first 16 bytes behind the last slot
#pragma unroll
            for (int m = 0; m < PPR; ++m) *reinterpret_cast<uint4 *>(ring + dst_off[m] + ring_len) = buf[m];
        }
    };
    // ring byte offset of stream position p (wave-uniform, p >= 0); only used to (re)start a trailing stream
    auto ring_off = [&](int p) { return ((p / CH) % NS) * CH + (p % CH); };

    Fold f;
    f.s = 0.0;
    f.w = 0.0;
    f.mn = 0.0;
    const int n_rounds = (Lmax + CH - 1) / CH;

    issue_loads(0, pre[0]);
    write_ring(0, pre[0]);
#pragma unroll
    for (int r = 1; r <= PF; ++r)
        if (r < n_rounds) issue_loads(r, pre[r % PF]);  // round r lives in pre[r % PF]
    __builtin_amdgcn_wave_barrier();

    int lslot = 0;      // ring slot of the current round
    uint4 xc = make_uint4(0, 0, 0, 0);  // last trailing ring piece of the previous steady round
    int tro = -1;       // ring byte offset of the trailing edge of the NEXT body piece; -1 = not tracking
    int tro_pos = -1;   // stream position tro belongs to

    for (int tb = 0; tb < n_rounds; tb += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int t = tb + d;
        if (t >= n_rounds) break;
        const bool more = t + 1 < n_rounds;
        const int r_lo = t * CH, r_hi = r_lo + CH;
        const unsigned char *lead_p = my_row + lslot * CH;

        if (r_hi <= Lmin && r_lo >= ws) {
            // ---------------- whole round in the steady state ----------------
            const bool chained = tro_pos == r_lo - ws;  // the previous round was a steady one too
            if (!chained) tro = ring_off(r_lo - ws);
            uint4 lead[PPR];
#pragma unroll
            for (int kk = 0; kk < PPR; ++kk) lead[kk] = *reinterpret_cast<const uint4 *>(lead_p + kk * 16);
            // Trailing bytes of the round: PPR + 1 ALIGNED 16-byte ring pieces (conflict-free b128 rows), then a
            // funnel shift by the constant (tro & 15).  Dword-granular reads would be 4-way bank conflicted
            // because every row stride is a multiple of 16 bytes.  The last piece of a round is the first of the
            // next one (the ring write in between touches the slot behind it), so a chained round reads only PPR.
            uint32_t x[(PPR + 1) * 4];
            {
                int o = tro & ~15;
                if (chained) {
                    x[0] = xc.x; x[1] = xc.y; x[2] = xc.z; x[3] = xc.w;
                } else {
                    const uint4 v = *reinterpret_cast<const uint4 *>(my_row + o);
                    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
                }
#pragma unroll
                for (int kk = 1; kk <= PPR; ++kk) {
                    o += 16;
                    if (o >= ring_len) o -= ring_len;
                    const uint4 v = *reinterpret_cast<const uint4 *>(my_row + o);
                    x[4 * kk + 0] = v.x; x[4 * kk + 1] = v.y; x[4 * kk + 2] = v.z; x[4 * kk + 3] = v.w;
                }
                xc = make_uint4(x[4 * PPR + 0], x[4 * PPR + 1], x[4 * PPR + 2], x[4 * PPR + 3]);
            }
            uint32_t tw[PPR][4];
            const uint32_t bsh = (uint32_t)tro & 3u;
#define FLX_FUNNEL(D)                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < PPR; ++kk) _Pragma("unroll") for (int d = 0; d < 4; ++d) \
        tw[kk][d] = __builtin_amdgcn_alignbyte(x[4 * kk + (D) + d + 1], x[4 * kk + (D) + d], bsh);
            switch ((tro >> 2) & 3) {  // wave-uniform and constant for the whole launch
                case 0: FLX_FUNNEL(0) break;
                case 1: FLX_FUNNEL(1) break;
                case 2: FLX_FUNNEL(2) break;
                default: FLX_FUNNEL(3) break;
            }
#undef FLX_FUNNEL
            tro += CH;
            if (tro >= ring_len) tro -= ring_len;
            tro_pos = r_hi - ws;
#pragma unroll
            for (int kk = 0; kk < PPR; ++kk) {
                const uint32_t lw[4] = {lead[kk].x, lead[kk].y, lead[kk].z, lead[kk].w};
                body16(lq, ld, lw, tw[kk], f);
            }
        } else if (r_hi <= Lmin && r_hi <= ws) {
            // ---------------- whole round before the first full window ----------------
#pragma unroll
            for (int kk = 0; kk < PPR; ++kk) {
                const uint4 v = *reinterpret_cast<const uint4 *>(lead_p + kk * 16);
                const uint32_t lw[4] = {v.x, v.y, v.z, v.w};
                head16(lq, lw, f);
            }
            if (r_hi == ws) {
                f.w = f.s / a.ws_d;  // src/read.cpp:223
                f.mn = f.w;
            }
        } else {
            // ---------------- mixed round: straddles window_size or the end of some read of this wave --------
#pragma unroll 1
            for (int kk = 0; kk < PPR; ++kk) {
                const int j0 = r_lo + kk * 16;
                if (j0 >= Lmax) break;
                const uint4 lead = *reinterpret_cast<const uint4 *>(lead_p + kk * 16);
                const uint32_t lw[4] = {lead.x, lead.y, lead.z, lead.w};
                if (j0 + 16 <= Lmin && j0 + 16 <= ws) {
                    head16(lq, lw, f);
                    if (j0 + 16 == ws) {
                        f.w = f.s / a.ws_d;
                        f.mn = f.w;
                    }
                } else if (j0 + 16 <= Lmin && j0 >= ws) {
                    uint32_t tw[4];
                    trail16(my_row, ring_off(j0 - ws), tw);
                    body16(lq, ld, lw, tw, f);
                } else {
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) {
                        const int j = j0 + i;
                        if (j >= Lmax) break;
                        const bool act = j < L;
                        const uint32_t cj = act ? byte_of(lead, i) : 256u;  // entry 256 = 0.0: exact no-op
                        f.s += lq[cj];
                        if (j == ws - 1) {
                            f.w = f.s / a.ws_d;
                            f.mn = f.w;
                        }
                        if (j >= ws) {
                            const uint32_t tb = my_row[ring_off(j - ws)];
                            const uint32_t ci = act ? tb : 256u;
                            f.w -= ld[ci];
                            f.w += ld[cj];
                            if (f.w < f.mn) f.mn = f.w;
                        }
                    }
                }
            }
        }

        if (more) {
            lslot = (lslot + 1 == NS) ? 0 : lslot + 1;
            write_ring(lslot, pre[(d + 1) % PF]);                                  // round t + 1
            if (t + 1 + PF < n_rounds) issue_loads(t + 1 + PF, pre[(d + 1) % PF]);  // refill the buffer just drained
        }
        __builtin_amdgcn_wave_barrier();
      }
    }

    if (live) finish_read(a, rid, L, f.s, f.mn);
    __builtin_amdgcn_wave_barrier();  // the ring rows are reused by the next group
  }
}

// ---------------------------------------------------------------------------------------------
// direct kernel: any window size, one lane per read straight from global memory.  Used when the
// window does not fit the LDS ring (window_size > ~2000) and as an independent second
// implementation in the tests.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) flx_score_phred_direct(const PhredArgs a) {
    __shared__ double lq[LUT_PAD];
    __shared__ double ld[LUT_PAD];
    for (int i = threadIdx.x; i < 257; i += 256) {
        lq[i] = a.lut_q[i];
        ld[i] = a.lut_d[i];
    }
    __syncthreads();
    const uint64_t slot = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= a.n_reads) return;
    const uint32_t rid = a.order ? a.order[slot] : (uint32_t)slot;
    const int L = a.lengths[rid];
    const uint8_t *q = a.plane + a.offsets[rid];
    const int ws = a.ws;
    double s = 0.0, w = 0.0, mn = 0.0;
    const int head = L < ws ? L : ws;
    for (int i = 0; i < head; ++i) s += lq[q[i]];
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

}  // namespace

// LDS budget: one workgroup per CU, as many independent waves as fit (each wave owns a ring).
static constexpr size_t kLdsBudget = 160 * 1024;
static constexpr size_t kLutBytes = 2 * LUT_PAD * sizeof(double);

int flx_launch_score_phred(flx_ctx *ctx, const uint8_t *d_plane, uint64_t plane_bytes, const uint64_t *d_offsets,
                           const int32_t *d_lengths, const uint32_t *d_order, uint64_t n_reads,
                           const flx_params *p, flx_score_out_dev out) {
    (void)plane_bytes;
    if (n_reads == 0) return FLX_OK;
    if (p->window_size <= 0) return flx_fail(ctx, FLX_ERR_INVALID, "window_size must be positive");
    FLX_CHECK(flx_ensure_lut_d(ctx, p->window_size));

    PhredArgs a;
    a.plane = d_plane;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.order = d_order;
    a.n_reads = n_reads;
    a.lut_q = ctx->d_lut_q;
    a.lut_d = ctx->d_lut_d;
    a.ws = p->window_size;
    a.ws_d = (double)(size_t)p->window_size;
    {
        volatile double half = 0.5, wsd = a.ws_d;
        a.clamp = half / wsd;
    }
    a.p = *p;
    a.mean_q = out.mean_q;
    a.window_q = out.window_q;
    a.passed = out.passed;
    a.ticket = nullptr;
    a.n_groups = 0;
    a.redo_count = nullptr;
    a.redo_list = nullptr;

    const long long n_slots = ((long long)p->window_size + CH - 1) / CH + 1;
    long long slots16 = n_slots * CH / 16 + 1;  // + the 16-byte mirror behind the last slot
    if ((slots16 & 1) == 0) slots16 += 1;       // odd number of 16-byte slots per row: b128 rows never collide
    const size_t ring_bytes = (size_t)64 * slots16 * 16;

    const uint64_t n_waves = (n_reads + 63) / 64;
    int waves = (int)((kLdsBudget - kLutBytes) / ring_bytes);
    const char *env = getenv("FLX_PHRED_KERNEL");  // test hook: "direct" / "ring" force the older kernels
    const bool force_direct = env && strcmp(env, "direct") == 0;
    const bool force_ring = env && strcmp(env, "ring") == 0;
    const bool force_stream = env && strcmp(env, "stream") == 0;
    const bool force_dual = env && strcmp(env, "dual") == 0;
    if (force_stream) return flx_launch_score_phred_stream(ctx, a);
    if (force_dual) return flx_launch_score_phred_dual(ctx, a);
    if (!force_direct && !force_ring) {  // default: the register-history kernel, where the window size has an instantiation ...
        bool launched = false;
        a.n_slots = 0;
        a.stride = 0;
        FLX_CHECK(flx_launch_score_phred_regs(ctx, a, &launched));
        if (launched) return FLX_OK;
        return flx_launch_score_phred_dual(ctx, a);  // ... and the dual-slot kernel beyond (window sizes from 624 on, any size)
    }

    if (waves >= 1 && !force_direct) {
        if (waves > 7) waves = 7;
        a.n_slots = (int)n_slots;
        a.stride = (int)(slots16 * 16);
        const size_t lds = (size_t)waves * ring_bytes;  // dynamic part; the tables are static LDS
        // persistent workgroups: as many as can be resident (LDS allows floor(budget / per-group) per CU), capped by the work
        const unsigned per_cu = (unsigned)std::max<size_t>(1, kLdsBudget / (lds + kLutBytes));
        const unsigned resident = (unsigned)ctx->prop.multiProcessorCount * per_cu;
        const unsigned grid = (unsigned)std::min<uint64_t>((n_waves + waves - 1) / waves, resident);
        void *scr;
        FLX_CHECK(flx_scratch(ctx, 64, &scr));
        FLX_HIP(ctx, hipMemsetAsync(scr, 0, 4, ctx->stream));
        a.ticket = (unsigned int *)scr;
        a.n_groups = (unsigned int)n_waves;
#define FLX_LAUNCH_RING(W)                                                                                    \
    case W: {                                                                                                 \
        auto kern = flx_score_phred_ring<W>;                                                                  \
        FLX_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                         (int)lds));                                                          \
        ctx->last_phred_kernel = "flx_score_phred_ring";                                                      \
        flx_time_begin(ctx, "flx_score_phred_ring");                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(W * 64), lds, ctx->stream, a);                              \
        flx_time_end(ctx);                                                                                    \
    } break;
        switch (waves) {
            FLX_LAUNCH_RING(1)
            FLX_LAUNCH_RING(2)
            FLX_LAUNCH_RING(3)
            FLX_LAUNCH_RING(4)
            FLX_LAUNCH_RING(5)
            FLX_LAUNCH_RING(6)
            FLX_LAUNCH_RING(7)
        }
#undef FLX_LAUNCH_RING
    } else if (!force_direct) {
        // the window does not fit the LDS ring (ws > ~2000): both window edges streamed from global memory, 16-byte loads
        return flx_launch_score_phred_stream(ctx, a);
    } else {
        a.n_slots = 0;
        a.stride = 0;
        a.ticket = nullptr;
        a.n_groups = 0;
        const unsigned grid = (unsigned)((n_reads + 255) / 256);
        ctx->last_phred_kernel = "flx_score_phred_direct";
        flx_time_begin(ctx, "flx_score_phred_direct");
        hipLaunchKernelGGL(flx_score_phred_direct, dim3(grid), dim3(256), 0, ctx->stream, a);
        flx_time_end(ctx);
    }
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}
