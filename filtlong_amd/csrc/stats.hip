// stats.hip — a20: statistics of the mean qualities, bit-identical to the reference's SERIAL folds
// (src/main.cpp:170-186):
//     for read in reads2: quality_sum += mean_q; track min/max        (strict reads2 order)
//     mean = quality_sum / N
//     for read in reads2: d = mean_q - mean; stdev_sum += d*d         (strict reads2 order)
//     stdev = sqrt(stdev_sum / N)
// FP64 addition is not associative: a tree reduction changes the last bits of the sums and,
// through the z-scores, the normalised quality of every read.  Ten million dependent adds are ~25 ms on
// a host core — as long as the whole scoring kernel — so the folds are reproduced EXACTLY on the device
// with a parallel algorithm:
//
//   While the running sum S stays inside one binade [2^e, 2^(e+1)) every addend x >= 0 is first rounded
//   to that binade's grid u = 2^(e-52); in units of u the fold is then plain INTEGER addition
//   (associative), except for ties (x/u ending in exactly .5), which round to even and therefore depend
//   on the parity of the running integer.  An element is thus a map  m -> m + a[m & 1]  and such maps
//   compose associatively:  (a o b)[p] = a[p] + b[(p + a[p]) & 1].
//
//   1. k_chunk_sums:  approximate sum of every 512-element chunk (any order) + "clean" flag (all finite, >= 0).
//   2. scan:          approximate running sum at every chunk start -> a GUESS of the binade of S there.
//   3. k_chunk_maps:  one wavefront per chunk composes the integer maps of its 512 elements for the guessed
//                     binade (ordered wave reduction).
//   3b. k_batch_maps: the maps of 64 consecutive chunks that assume the same binade are composed once more.
//   4. k_walk:        one wavefront walks the batches (then chunks, then elements) in order carrying the EXACT S.  A chunk's map is used
//                     only if S really is in the guessed binade at the chunk start and stays there (checked
//                     on the exact integers); otherwise the chunk is folded serially, element by element.
//                     So a wrong guess costs time, never correctness.  Binade crossings (~30 per fold),
//                     negative or non-finite values take the serial path.
#include <cmath>

#include "flx_internal.h"
#include "rank_internal.h"

namespace {

constexpr int CHUNK = 512;  // elements per chunk = 64 lanes x 8
constexpr int PER_LANE = CHUNK / 64;

struct Identity {
    double mean;
    __device__ __forceinline__ double operator()(double x) const { return x; }
};
struct SqDev {  // main.cpp:182-185: d = x - mean; d * d   (separate IEEE sub and mul, no FMA)
    double mean;
    __device__ __forceinline__ double operator()(double x) const {
        const double d = x - mean;
        return d * d;
    }
};

struct ChunkMeta {
    long long a0, a1;  // integer increment (units of 2^(e-52)) for entry parity 0 / 1
    int e;             // guessed unbiased binade exponent of the running sum, or INT_MIN: no map
    int pad;
};
constexpr int NO_MAP = -2147483647 - 1;

__device__ __forceinline__ int exponent_of(double v) {  // unbiased exponent of a positive normal double
    return (int)((__double_as_longlong(v) >> 52) & 0x7ff) - 1023;
}

template <typename F>
__global__ void __launch_bounds__(256) k_chunk_sums(uint64_t n, const double *x, F f, double *chunk_sum,
                                                    unsigned char *chunk_clean, double *minmax) {
    // one wave per chunk
    const int lane = threadIdx.x & 63;
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t base = chunk * CHUNK;
    if (base >= n) return;
    double acc = 0.0;
    bool clean = true;
    double lo = 100.0, hi = 0.0;  // main.cpp:170-171 initial values
#pragma unroll
    for (int k = 0; k < PER_LANE; ++k) {
        const uint64_t i = base + (uint64_t)k * 64 + lane;  // coalesced; order is irrelevant here
        if (i < n) {
            const double raw = x[i];
            const double v = f(raw);
            acc += v;
            clean = clean && (v >= 0.0) && (v < __longlong_as_double(0x7ff0000000000000ll));
            if (raw > hi) hi = raw;
            if (raw < lo) lo = raw;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        const double h2 = __shfl_xor(hi, o, 64), l2 = __shfl_xor(lo, o, 64);
        if (h2 > hi) hi = h2;
        if (l2 < lo) lo = l2;
    }
    const bool all_clean = __all(clean);
    if (lane == 0) {
        chunk_sum[chunk] = acc;
        chunk_clean[chunk] = all_clean ? 1 : 0;
        if (minmax) {
            minmax[2 * chunk] = lo;
            minmax[2 * chunk + 1] = hi;
        }
    }
}

// final min / max over the per-chunk values (order free; NaN never wins a comparison, like the reference)
__global__ void __launch_bounds__(256) k_minmax(uint64_t n_chunks, const double *minmax, double *out) {
    double lo = 100.0, hi = 0.0;
    for (uint64_t c = threadIdx.x; c < n_chunks; c += 256) {
        const double l = minmax[2 * c], h = minmax[2 * c + 1];
        if (l < lo) lo = l;
        if (h > hi) hi = h;
    }
    __shared__ double sl[256], sh[256];
    sl[threadIdx.x] = lo;
    sh[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) {
            if (sl[i] < lo) lo = sl[i];
            if (sh[i] > hi) hi = sh[i];
        }
        out[0] = lo;
        out[1] = hi;
    }
}

struct Map2 {
    long long a0, a1;
};
__device__ __forceinline__ Map2 compose(const Map2 &a, const Map2 &b) {  // a first, then b
    Map2 c;
    c.a0 = a.a0 + ((a.a0 & 1) ? b.a1 : b.a0);
    c.a1 = a.a1 + (((1 + a.a1) & 1) ? b.a1 : b.a0);
    return c;
}

// integer map of one addend v >= 0 (finite) for a running sum in binade e; ok = false if v cannot be added
// without leaving the binade
__device__ __forceinline__ Map2 elem_map(double v, int e, bool &ok) {
    Map2 m;
    m.a0 = m.a1 = 0;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if ((bits << 1) == 0) return m;  // +-0
    const int eb = (int)((bits >> 52) & 0x7ff);
    unsigned long long mx = bits & 0x000fffffffffffffull;
    int ex;
    if (eb == 0) ex = -1022;  // subnormal
    else { mx |= 1ull << 52; ex = eb - 1023; }
    const int sh = e - ex;  // v / u = mx >> sh
    if (sh < 0) { ok = false; return m; }
    if (sh == 0) { m.a0 = m.a1 = (long long)mx; return m; }
    if (sh >= 55) return m;
    const unsigned long long f = mx >> sh;
    const unsigned long long rem = mx & ((1ull << sh) - 1ull);
    const unsigned long long half = 1ull << (sh - 1);
    if (rem > half) m.a0 = m.a1 = (long long)(f + 1);
    else if (rem < half) m.a0 = m.a1 = (long long)f;
    else {  // tie: the sum rounds to even
        m.a0 = (long long)(f + (f & 1));
        m.a1 = (long long)(f + ((f + 1) & 1));
    }
    return m;
}

template <typename F>
__global__ void __launch_bounds__(256) k_chunk_maps(uint64_t n, const double *x, F f, const double *chunk_start,
                                                    const double *chunk_sum, const unsigned char *chunk_clean,
                                                    ChunkMeta *meta) {
    const int lane = threadIdx.x & 63;
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t base = chunk * CHUNK;
    if (base >= n) return;
    // guess the binade from the approximate running sum at the chunk's start and end
    const double p0 = chunk_start[chunk] * (1.0 - 1e-9);
    const double p1 = (chunk_start[chunk] + chunk_sum[chunk]) * (1.0 + 1e-9);
    int e = NO_MAP;
    if (chunk_clean[chunk] && p0 > 2.3e-308 && p1 < 1e300 && exponent_of(p0) == exponent_of(p1)) e = exponent_of(p0);
    Map2 m;
    m.a0 = m.a1 = 0;
    bool ok = true;
    if (e != NO_MAP) {
        // lane l owns elements [l*8, l*8+8) so that lane order == element order
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const uint64_t i = base + (uint64_t)lane * PER_LANE + k;
            if (i < n) m = compose(m, elem_map(f(x[i]), e, ok));
        }
        // ordered wave reduction
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            Map2 nb;
            nb.a0 = __shfl_down(m.a0, o, 64);
            nb.a1 = __shfl_down(m.a1, o, 64);
            if ((lane & (2 * o - 1)) == 0) m = compose(m, nb);
        }
        if (!__all(ok)) e = NO_MAP;
    }
    if (lane == 0) {
        ChunkMeta cm;
        cm.a0 = m.a0;
        cm.a1 = m.a1;
        cm.e = e;
        cm.pad = 0;
        meta[chunk] = cm;
    }
}

// second level: one wavefront composes the maps of 64 consecutive chunks when they all assume the same binade
__global__ void __launch_bounds__(256) k_batch_maps(const ChunkMeta *meta, uint64_t n_chunks, ChunkMeta *batch) {
    const int lane = threadIdx.x & 63;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t c0 = b * 64;
    if (c0 >= n_chunks) return;
    ChunkMeta mine;
    mine.a0 = mine.a1 = 0;
    mine.e = NO_MAP;
    if (c0 + lane < n_chunks) mine = meta[c0 + lane];
    const int e0 = __builtin_amdgcn_readfirstlane(mine.e);
    const bool full = c0 + 64 <= n_chunks;
    Map2 m;
    m.a0 = mine.a0;
    m.a1 = mine.a1;
    const bool ok = full && e0 != NO_MAP && __all(mine.e == e0);
    if (ok) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            Map2 nb;
            nb.a0 = __shfl_down(m.a0, o, 64);
            nb.a1 = __shfl_down(m.a1, o, 64);
            if ((lane & (2 * o - 1)) == 0) m = compose(m, nb);
        }
    }
    if (lane == 0) {
        ChunkMeta out;
        out.a0 = m.a0;
        out.a1 = m.a1;
        out.e = ok ? e0 : NO_MAP;
        out.pad = 0;
        batch[b] = out;
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// exact application of an integer map to the running sum: returns true and updates S iff S is in binade e and stays there
__device__ __forceinline__ bool apply_map(double &S, int e, long long a0, long long a1) {
    const unsigned long long sb = (unsigned long long)__double_as_longlong(S);
    const int se = (int)((sb >> 52) & 0x7ff) - 1023;
    if (se != e || (sb >> 63) != 0 || a0 < 0 || a1 < 0) return false;
    const unsigned long long m = (sb & 0x000fffffffffffffull) | (1ull << 52);
    const unsigned long long m2 = m + (unsigned long long)((m & 1) ? a1 : a0);
    if (m2 > (1ull << 53)) return false;  // would leave the binade (reaching 2^(e+1) exactly is still on the grid)
    if (m2 == (1ull << 53)) S = __longlong_as_double((long long)(((unsigned long long)(e + 1 + 1023)) << 52));
    else S = __longlong_as_double((long long)((((unsigned long long)(e + 1023)) << 52) | (m2 & 0x000fffffffffffffull)));
    return true;
}

// Ordered inclusive scan of maps over the lanes of a wavefront: lane l ends with m_0 o m_1 o ... o m_l (m_0 applied first).
__device__ __forceinline__ Map2 wave_scan_maps(Map2 m, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Map2 prev;
        prev.a0 = __shfl_up(m.a0, o, 64);
        prev.a1 = __shfl_up(m.a1, o, 64);
        if (lane >= o) m = compose(prev, m);
    }
    return m;
}

__device__ __forceinline__ bool normal_positive(double S) {
    const unsigned long long sb = (unsigned long long)__double_as_longlong(S);
    const int eb = (int)((sb >> 52) & 0x7ff);
    return (sb >> 63) == 0 && eb >= 1 && eb <= 2046;
}

// Advances S (positive, normal, in binade e) through the maps of lanes start, start + 1, ... for as long as they are usable
// (a map for binade e with non-negative increments) and S stays in the binade, ALL AT ONCE: an ordered scan composes the
// maps, every lane applies its prefix to the entering S, and the first lane whose prefix leaves the binade is found with a
// ballot.  Increments are >= 0, so the mantissa only grows: a prefix that stays inside implies every shorter one does.
// Returns the first lane that was NOT applied (64: all of them were).
__device__ __forceinline__ int wave_apply_prefix(double &S, int e, Map2 m, bool usable, int start, int lane) {
    const bool good = usable && m.a0 >= 0 && m.a1 >= 0;
    const unsigned long long bad = __ballot(lane >= start && !good);
    const int first_bad = bad ? (int)__ffsll((long long)bad) - 1 : 64;
    if (lane < start || lane >= first_bad) m.a0 = m.a1 = 0;
    const Map2 P = wave_scan_maps(m, lane);
    const unsigned long long sb = (unsigned long long)__double_as_longlong(S);
    const unsigned long long ms = (sb & 0x000fffffffffffffull) | (1ull << 52);
    const unsigned long long m2 = ms + (unsigned long long)((ms & 1) ? P.a1 : P.a0);
    const bool inside = m2 <= (1ull << 53);  // reaching 2^(e+1) exactly is still on this binade's grid
    const unsigned long long out = __ballot(lane >= start && lane < first_bad && !inside);
    const int f = out ? (int)__ffsll((long long)out) - 1 : first_bad;
    if (f > start) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(m2 & 0xffffffffull), f - 1);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(m2 >> 32), f - 1);
        const unsigned long long r = ((unsigned long long)hi << 32) | lo;
        if (r == (1ull << 53)) S = __longlong_as_double((long long)(((unsigned long long)(e + 1 + 1023)) << 52));
        else S = __longlong_as_double((long long)((((unsigned long long)(e + 1023)) << 52) | (r & 0x000fffffffffffffull)));
    }
    return f;
}

// single wavefront: the exact fold.  Three levels: batches of 64 chunks -> chunks of 512 elements -> lanes of 8 elements.
// At every level the 64 maps in hand are applied with ONE scan up to the first that does not fit (a binade crossing, a
// wrong guess, a negative or non-finite value); only that one is opened: a batch into its chunks, a chunk into per-lane
// maps computed for the binade S is really in, a lane into 8 serial additions.  A fold has ~25 binade crossings, so
// ~25 descents, each two dependent loads and a handful of scans long.
template <typename F>
__global__ void __launch_bounds__(64) k_walk(uint64_t n, const double *x, F f, const ChunkMeta *meta, const ChunkMeta *batch,
                                             uint64_t n_chunks, double *result, unsigned long long *serial_chunks) {
    const int lane = threadIdx.x;
    double S = 0.0;
    unsigned long long n_serial = 0;
    const uint64_t n_batches = (n_chunks + 63) / 64;
    for (uint64_t bb = 0; bb < n_batches && S == S; bb += 64) {  // NaN is absorbing
        ChunkMeta bm;
        bm.a0 = bm.a1 = 0;
        bm.e = NO_MAP;
        if (bb + lane < n_batches) bm = batch[bb + lane];
        const int bcnt = (int)((n_batches - bb) < 64 ? (n_batches - bb) : 64);
        int bs = 0;
        while (bs < bcnt && S == S) {
            int fb = bs;
            if (normal_positive(S)) {
                const int e = exponent_of(S);
                Map2 m;
                m.a0 = bm.a0;
                m.a1 = bm.a1;
                fb = wave_apply_prefix(S, e, m, lane < bcnt && bm.e == e, bs, lane);
            }
            if (fb >= bcnt) break;
            // batch bb + fb does not apply: open it
            const uint64_t cb = (bb + fb) * 64;
            ChunkMeta mine;
            mine.a0 = mine.a1 = 0;
            mine.e = NO_MAP;
            if (cb + lane < n_chunks) mine = meta[cb + lane];
            const int cnt = (int)((n_chunks - cb) < 64 ? (n_chunks - cb) : 64);
            int cs = 0;
            while (cs < cnt && S == S) {
                int fc = cs;
                if (normal_positive(S)) {
                    const int e = exponent_of(S);
                    Map2 m;
                    m.a0 = mine.a0;
                    m.a1 = mine.a1;
                    fc = wave_apply_prefix(S, e, m, lane < cnt && mine.e == e, cs, lane);
                }
                if (fc >= cnt) break;
                // chunk cb + fc does not apply: open it (lane l holds elements [8 l, 8 l + 8), lane order == element order)
                ++n_serial;
                const uint64_t chunk = cb + fc;
                const uint64_t base = chunk * CHUNK;
                double v[PER_LANE];
#pragma unroll
                for (int j = 0; j < PER_LANE; ++j) {
                    const uint64_t i = base + (uint64_t)lane * PER_LANE + j;
                    v[j] = i < n ? f(x[i]) : 0.0;
                }
                const int m_el = (int)((n - base) < CHUNK ? (n - base) : CHUNK);
                const int n_lanes = (m_el + PER_LANE - 1) / PER_LANE;
                int ls = 0, opened = 0;
                while (ls < n_lanes && S == S) {
                    int fl = ls;
                    // the very first chunk climbs through a binade every few elements, and a chunk that keeps failing is not
                    // worth more scans: fold those lane by lane
                    if (chunk != 0 && opened < 4 && normal_positive(S)) {
                        const int e = exponent_of(S);
                        Map2 m;
                        m.a0 = m.a1 = 0;
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < PER_LANE; ++j) {
                            if (lane * PER_LANE + j < m_el) {
                                const double xv = v[j];
                                if (xv >= 0.0 && xv < __longlong_as_double(0x7ff0000000000000ll)) m = compose(m, elem_map(xv, e, ok));
                                else ok = false;
                            }
                        }
                        fl = wave_apply_prefix(S, e, m, ok && lane < n_lanes, ls, lane);
                    }
                    if (fl >= n_lanes) break;
                    // serial, element by element, exactly like the reference loop
#pragma unroll
                    for (int j = 0; j < PER_LANE; ++j)
                        if (fl * PER_LANE + j < m_el) S += readlane_f64(v[j], fl);
                    ++opened;
                    ls = fl + 1;
                }
                cs = fc + 1;
            }
            bs = fb + 1;
        }
    }
    if (lane == 0) {
        result[0] = S;
        serial_chunks[0] = n_serial;
    }
}

template <typename F>
int exact_fold(flx_ctx *ctx, uint64_t n, const double *x, F f, char *ws, double *d_result, unsigned long long *d_serial,
               double *d_minmax_out) {
    const uint64_t n_chunks = (n + CHUNK - 1) / CHUNK;
    double *chunk_sum = (double *)ws;
    double *chunk_start = chunk_sum + n_chunks;
    double *minmax = chunk_start + n_chunks;
    ChunkMeta *meta = (ChunkMeta *)(minmax + 2 * n_chunks);
    const uint64_t n_batches = (n_chunks + 63) / 64;
    ChunkMeta *batch = meta + n_chunks;
    unsigned char *clean = (unsigned char *)(batch + n_batches);
    char *scan_ws = (char *)(((uintptr_t)(clean + n_chunks) + 255) & ~(uintptr_t)255);
    hipStream_t st = ctx->stream;
    const unsigned nb = (unsigned)((n_chunks + 3) / 4);
    hipLaunchKernelGGL(k_chunk_sums<F>, dim3(nb), dim3(256), 0, st, n, x, f, chunk_sum, clean,
                       d_minmax_out ? minmax : (double *)nullptr);
    if (d_minmax_out) hipLaunchKernelGGL(k_minmax, dim3(1), dim3(256), 0, st, n_chunks, minmax, d_minmax_out);
    FLX_HIP(ctx, hipMemcpyAsync(chunk_start, chunk_sum, n_chunks * 8, hipMemcpyDeviceToDevice, st));
    FLX_CHECK(flx_exclusive_scan_f64_approx(ctx, n_chunks, chunk_start, scan_ws));
    hipLaunchKernelGGL(k_chunk_maps<F>, dim3(nb), dim3(256), 0, st, n, x, f, chunk_start, chunk_sum, clean, meta);
    hipLaunchKernelGGL(k_batch_maps, dim3((unsigned)((n_batches + 3) / 4)), dim3(256), 0, st, meta, n_chunks, batch);
    hipLaunchKernelGGL(k_walk<F>, dim3(1), dim3(64), 0, st, n, x, f, meta, batch, n_chunks, d_result, d_serial);
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}

}  // namespace

int flx_exact_stats(flx_ctx *ctx, uint64_t n, const double *d_mean_q, flx_stats *out) {
    memset(out, 0, sizeof *out);
    if (n == 0) {  // 0/0, like the reference would compute
        volatile double z = 0.0;
        out->min = 100.0; out->max = 0.0; out->sum = 0.0; out->mean = z / z; out->sq_sum = 0.0; out->stdev = sqrt(z / z);
        return FLX_OK;
    }
    const uint64_t n_chunks = (n + CHUNK - 1) / CHUNK;
    const size_t bytes = n_chunks * (8 + 8 + 16 + sizeof(ChunkMeta) + 1) + (n_chunks / 64 + 2) * sizeof(ChunkMeta) + 512 + (n_chunks / 2048 + 4096) * 8 * 2 + 4096;
    void *scr;
    FLX_CHECK(flx_scratch(ctx, bytes, &scr));
    char *ws = (char *)scr;
    double *d_res = (double *)ws;             // [0]=sum/sq_sum, [2..3] = min,max
    unsigned long long *d_serial = (unsigned long long *)(ws + 64);
    ws += 256;

    flx_time_begin(ctx, "flx_rank_stats");
    FLX_CHECK(exact_fold(ctx, n, d_mean_q, Identity{0.0}, ws, d_res, d_serial, d_res + 2));
    double h[4];
    unsigned long long h_serial[2] = {0, 0};
    FLX_HIP(ctx, hipMemcpyAsync(h, d_res, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(&h_serial[0], d_serial, 8, hipMemcpyDeviceToHost, ctx->stream));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->sum = h[0];
    out->min = h[2];
    out->max = h[3];
    {
        volatile double s = out->sum, nn = (double)n;
        out->mean = s / nn;  // main.cpp:181
    }
    FLX_CHECK(exact_fold(ctx, n, d_mean_q, SqDev{out->mean}, ws, d_res, d_serial, (double *)nullptr));
    FLX_HIP(ctx, hipMemcpyAsync(h, d_res, 8, hipMemcpyDeviceToHost, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(&h_serial[1], d_serial, 8, hipMemcpyDeviceToHost, ctx->stream));
    flx_time_end(ctx);
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->sq_sum = h[0];
    {
        volatile double ss = out->sq_sum, nn = (double)n;
        out->stdev = sqrt(ss / nn);  // main.cpp:186 (sqrt is correctly rounded)
    }
    out->serial_chunks = h_serial[0] + h_serial[1];
    return FLX_OK;
}
