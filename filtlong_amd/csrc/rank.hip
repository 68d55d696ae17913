// rank.hip — seam 3: global statistics, normalisation, final score and the --target_bases / --keep_percent cut.
// Replaces reference src/main.cpp:169-261 and Read::set_final_score (src/read.cpp:249-267).
//
// Exactness plan (DESIGN.md "global stage"):
//   * min / max / mean / stdev of the mean qualities: the reference folds them serially in reads2
//     order (main.cpp:173-185, FP non-associative).  flx_exact_stats reproduces those two folds
//     bit-for-bit (see stats.hip).
//   * normalise (main.cpp:202-208): IEEE div/mul/sub only -> identical on the device
//     (-ffp-contract=off, correctly rounded FP64 division).
//   * set_final_score calls glibc pow three times (read.cpp:252,254); glibc pow is not correctly
//     rounded, so the device's pow may differ in the last ulp.  Final scores are therefore used on the
//     device only to ORDER reads; the reads whose device score lies within a relative band of the
//     cut score are re-scored on the host with the host libm ("boundary audit") and the cut is
//     re-decided among them.  NaN scores and exact ties straddling the cut fall back to the reference's
//     own std::sort order (host path, rare; report->exact_fallback = 1).
//
// The cut itself (std::sort by score + serial walk, main.cpp:247-257): the walk keeps a read iff it passed and
// the bases kept before it are < target, so the kept reads are exactly the passed reads up to and including
// the first one (in descending-score order, ties in reads2 order) at which the running total reaches the
// target.  Two device implementations of that:
//   * SELECT (default): MSD radix selection with weights — 8 histogram passes over the order-preserving 64-bit
//     keys (one byte per pass, bins hold summed read lengths) locate the crossing key without moving any data;
//     streaming reads only, which matters when the stage is replicated over 8x the reads after the all-gather.
//   * SORT (FLX_RANK_SORT=1, and the fallback when the audit band is huge): stable LSD radix sort of
//     (key, index) + exclusive scan of the lengths in sorted order + binary search (sort.hip).
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

#include <thread>

#include "flx_internal.h"
#include "rank_internal.h"

namespace {

struct NormArgs {
    double qmean, qstd, zmin, zspan;
    double lw, mw, ww;
};

// a8, src/read.cpp:241-244
__device__ __host__ inline double length_score(int length) {
    const double half = 5000.0;
    return 100.0 * (1.0 + (-half / (length + half)));
}

// order-preserving map double -> uint64 (ascending), NaN sorts above +inf
__device__ __host__ inline uint64_t key_ascending(double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <typename PowFn>
__device__ __host__ inline double final_score_with(PowFn powfn, int length, double mean_raw, double window_raw,
                                                   const NormArgs &s) {
    // main.cpp:203-208
    double ratio = window_raw / mean_raw;
    if (ratio > 1.0) ratio = 1.0;
    const double z = (mean_raw - s.qmean) / s.qstd;
    const double mq = 100.0 * (z - s.zmin) / s.zspan;
    const double wq = mq * ratio;
    // read.cpp:249-267
    const double product = powfn(length_score(length), s.lw) * powfn(mq, s.mw);
    double total = s.lw + s.mw;
    const double gm = powfn(product, 1.0 / total);
    double scale;
    if (mq > 0.0) {
        scale = wq / mq;
        if (1.0 < scale) scale = 1.0;  // std::min(x, 1.0) == (1.0 < x) ? 1.0 : x   (x NaN stays NaN)
    } else
        scale = 1.0;
    total = s.lw + s.mw + s.ww;
    const double wfrac = s.ww / total;
    const double nfrac = 1.0 - wfrac;
    scale = nfrac + (scale * wfrac);
    return gm * scale;
}

// The device score only ORDERS reads (the boundary audit re-scores the band around the cut with the host libm), so the two
// exponents of the default weights need no pow: x^1 is x and x^(1/2) the correctly rounded square root, both within an ulp
// of what glibc's pow returns — far inside the audit band.  The exponent is the same for every read: a uniform branch.
struct DevPow {
    __device__ double operator()(double x, double y) const {
        if (y == 1.0) return x;
        if (y == 0.5) return sqrt(x);
        return pow(x, y);
    }
};

__global__ void k_final_score(uint64_t n, const double *mean_q, const double *window_q, const int32_t *length,
                              NormArgs s, double *final_score, uint64_t *keys, uint32_t *vals,
                              unsigned int *any_nan) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double f = final_score_with(DevPow(), length[i], mean_q[i], window_q[i], s);
    if (any_nan && f != f) atomicOr(any_nan, 1u);
    if (final_score) final_score[i] = f;
    if (keys) {
        keys[i] = ~key_ascending(f);  // descending score == ascending key
        if (vals) vals[i] = (uint32_t)i;
    }
}

// one atomic per workgroup (256 threads): thousands of same-address atomics cost more than the pass over the data
__device__ __forceinline__ void block_add(unsigned long long acc, unsigned long long *out) {
    __shared__ unsigned long long part[4];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(out, t);
    }
}

__global__ void __launch_bounds__(256) k_passed_bases(uint64_t n, const int32_t *length, const uint8_t *passed,
                               unsigned long long *out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (passed[i]) acc += (unsigned long long)length[i];
    block_add(acc, out);
}

// weights in sorted order: bases a read contributes to the walk (main.cpp:251-257)
__global__ void k_cut_weights(uint64_t n, const uint32_t *sorted_idx, const int32_t *length, const uint8_t *passed,
                              int64_t *w, uint8_t *pre_sorted) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = sorted_idx[i];
    const uint8_t ok = passed[r];
    pre_sorted[i] = ok;  // pass flag before the cut, in sorted order (the audit needs it)
    w[i] = ok ? (int64_t)length[r] : 0;
}

// The walk of main.cpp:251-257 keeps a read iff it passed and bases_so_far (before it) < target.  bases_so_far is
// non-decreasing along the sorted order, so the kept reads are exactly the passed reads up to and including the
// LAST passed read whose exclusive prefix is below the target.  Find that position (sequential reads only) ...
__global__ void __launch_bounds__(64) k_cut_find(uint64_t n, const int64_t *excl, const uint8_t *pre_sorted, int64_t target,
                                                  unsigned long long *last_kept_pos_plus1) {
    // single wavefront: binary search for the last position whose exclusive prefix is below the target (excl is
    // non-decreasing), then step back to the nearest read that had passed before the cut.
    const int lane = threadIdx.x;
    uint64_t lo = 0, hi = n;  // first position with excl >= target lies in [lo, hi]
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (excl[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    // positions [0, lo) have excl < target
    unsigned long long found = 0;
    for (uint64_t base = lo; base > 0 && !found;) {
        const uint64_t start = base > 64 ? base - 64 : 0;
        const uint64_t i = start + lane;
        const bool hit = i < base && pre_sorted[i];
        const unsigned long long m = __ballot(hit);
        if (m) found = start + (63 - __clzll(m)) + 1;
        base = start;
    }
    if (lane == 0) *last_kept_pos_plus1 = found;
}

// ... and then mark in READ order, without any scattered access: the sort is stable, so "at or before sorted
// position p" is the same as "(key, read index) <= (key at p, read index at p)".
__global__ void k_cut_mark(uint64_t n, const uint64_t *keys_orig, uint64_t key_star, uint32_t idx_star,
                           uint8_t *passed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys_orig[i];
    const bool before = k < key_star || (k == key_star && (uint32_t)i <= idx_star);
    if (!before) passed[i] = 0;
}


// ---- SELECT path ---------------------------------------------------------------------------------
// state of the radix selection, kept on the device so that the 8 passes need no host round trip
struct SelState {
    unsigned long long prefix;  // key bytes decided so far (most significant first)
    long long remaining;        // bases still to be collected inside the current prefix
    unsigned int nan;           // some final score is NaN
    unsigned int fail;          // ran out of weight (cannot happen when 0 < target < passed_bases)
};

// weight histogram of one key byte: bins[b] += length of every PASSED read whose key agrees with the state's prefix in its
// top `prefix_bytes` bytes and whose next byte is b
__global__ void __launch_bounds__(256) k_select_hist(uint64_t n, const uint64_t *keys, const int32_t *length,
                                                     const uint8_t *passed, const SelState *st, int prefix_bytes,
                                                     unsigned long long *bins) {
    __shared__ unsigned long long h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t prefix = st->prefix;
    const int shift = 56 - 8 * prefix_bytes;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        const bool match = prefix_bytes == 0 || (k >> (shift + 8)) == prefix;
        if (match && passed[i]) {
            const int len = length[i];
            if (len > 0) atomicAdd(&h[(k >> shift) & 0xff], (unsigned long long)len);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
}

// one byte of the crossing key from the (globally summed) histogram of this pass; bins[256] of pass 0 carries the NaN flag
__global__ void __launch_bounds__(256) k_select_decide(const unsigned long long *bins, SelState *st, int pass) {
    __shared__ unsigned long long h[256];
    h[threadIdx.x] = bins[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (pass == 0 && bins[256]) st->nan = 1;
    long long cum = 0;
    const long long remaining = st->remaining;
    int d = 0;
    for (; d < 256; ++d) {
        if (cum + (long long)h[d] >= remaining) break;
        cum += (long long)h[d];
    }
    if (d == 256) {
        st->fail = 1;
        d = 255;
    }
    st->remaining = remaining - cum;
    st->prefix = (st->prefix << 8) | (unsigned long long)d;
}

// boundary-audit record of one read: everything the host needs to re-score it with the host libm
struct BandRec {
    uint64_t key;
    double mean, window;
    uint32_t idx;  // local reads2 index
    int32_t len;
    uint32_t was_passed;
    uint32_t pad;
};
__global__ void __launch_bounds__(256) k_band_gather(unsigned int m, const uint32_t *band_idx, const uint64_t *keys,
                                                     const double *mean, const double *window, const int32_t *length,
                                                     const uint8_t *passed, BandRec *out) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint32_t r = band_idx[i];
    BandRec b;
    b.key = keys[r];
    b.mean = mean[r];
    b.window = window[r];
    b.idx = r;
    b.len = length[r];
    b.was_passed = passed[r];
    b.pad = 0;
    out[i] = b;
}
// the same for a run of SORTED positions [pos0, pos0 + m): read index from the sorted values, pre-cut flag in sorted order
__global__ void __launch_bounds__(256) k_band_gather_sorted(unsigned int m, uint64_t pos0, const uint32_t *sorted_idx,
                                                            const uint64_t *sorted_keys, const double *mean,
                                                            const double *window, const int32_t *length,
                                                            const uint8_t *pre_sorted, BandRec *out) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint32_t r = sorted_idx[pos0 + i];
    BandRec b;
    b.key = sorted_keys[pos0 + i];
    b.mean = mean[r];
    b.window = window[r];
    b.idx = r;
    b.len = length[r];
    b.was_passed = pre_sorted[pos0 + i];
    b.pad = 0;
    out[i] = b;
}
// passed[idx[i]] = val[i]
__global__ void __launch_bounds__(256) k_scatter_flags(unsigned int m, const uint32_t *idx, const uint8_t *val, uint8_t *passed) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) passed[idx[i]] = val[i];
}

// every read whose key lies in [k_lo, k_hi] is appended to the band list; the lengths of passed reads with a
// smaller key (= better score) are summed: that is bases_so_far when the walk enters the band
__global__ void __launch_bounds__(256) k_select_band(uint64_t n, const uint64_t *keys, const int32_t *length,
                                                     const uint8_t *passed, uint64_t k_lo, uint64_t k_hi,
                                                     uint32_t *band_idx, unsigned int *band_n, unsigned int cap,
                                                     unsigned long long *weight_before) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        if (k < k_lo) {
            if (passed[i]) acc += (unsigned long long)length[i];
        } else if (k <= k_hi) {
            const unsigned int at = atomicAdd(band_n, 1u);
            if (at < cap) band_idx[at] = (uint32_t)i;
        }
    }
    block_add(acc, weight_before);
}

// everything at or beyond the band fails; the kept members of the band are switched back on by the host
__global__ void __launch_bounds__(256) k_select_mark(uint64_t n, const uint64_t *keys, uint64_t k_lo, uint8_t *passed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && keys[i] >= k_lo) passed[i] = 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host-exact pieces (used by the boundary audit and the tie fallback)
// ------------------------------------------------------------------------------------------------
struct HostPow {
    double operator()(double x, double y) const {
        static double (*volatile fn)(double, double) = pow;
        return fn(x, y);
    }
};

static double host_final_score(int length, double mean_raw, double window_raw, const NormArgs &s) {
    return final_score_with(HostPow(), length, mean_raw, window_raw, s);
}

static int64_t compute_target(int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                              int64_t total_bases) {
    // main.cpp:229-237
    long long target = target_bases_set ? (long long)target_bases : std::numeric_limits<long long>::max();
    if (keep_percent_set) {
        volatile double kp = keep_percent;
        long long keep = (long long)((kp / 100.0) * total_bases);
        target = std::min(target, keep);
    }
    return target;
}

// Tie fallback: the reference's own order (libstdc++ std::sort on reads2 order, main.cpp:247-248) with
// host-libm scores for every read.  Only reached when equal scores straddle the cut.
static int exact_host_cut(flx_ctx *ctx, uint64_t n, const double *d_mean, const double *d_window,
                          const int32_t *d_length, uint8_t *d_passed, const uint32_t *d_sorted_idx,
                          const uint8_t *d_pre_sorted, const NormArgs &s, int64_t target, flx_cut_report *rep) {
    // d_sorted_idx / d_pre_sorted != NULL: the SORT path already overwrote d_passed; rebuild the pre-cut flags.
    // NULL: d_passed still holds the pre-cut flags (SELECT path).
    std::vector<double> mean(n), window(n), fs(n);
    std::vector<int32_t> len(n);
    std::vector<uint8_t> passed(n);
    if (d_sorted_idx) {
        std::vector<uint8_t> pre(n);
        std::vector<uint32_t> sidx(n);
        FLX_HIP(ctx, hipMemcpy(pre.data(), d_pre_sorted, n, hipMemcpyDeviceToHost));
        FLX_HIP(ctx, hipMemcpy(sidx.data(), d_sorted_idx, n * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) passed[sidx[i]] = pre[i];
    } else {
        FLX_HIP(ctx, hipMemcpy(passed.data(), d_passed, n, hipMemcpyDeviceToHost));
    }
    FLX_HIP(ctx, hipMemcpy(mean.data(), d_mean, n * 8, hipMemcpyDeviceToHost));
    FLX_HIP(ctx, hipMemcpy(window.data(), d_window, n * 8, hipMemcpyDeviceToHost));
    FLX_HIP(ctx, hipMemcpy(len.data(), d_length, n * 4, hipMemcpyDeviceToHost));
    // exact scores with the host libm (three pow calls per read, src/read.cpp:252-254): independent per read -> all host threads
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nt = (unsigned)std::min<uint64_t>(hw, std::max<uint64_t>(1, n / 65536));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                for (uint64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) fs[i] = host_final_score(len[i], mean[i], window[i], s);
            });
        for (auto &th : pool) th.join();
    }
    // The reference sorts its vector<Read*> in reads2 order with libstdc++'s std::sort and a comparator on the scores
    // (src/main.cpp:247-248).  The permutation introsort produces is a function of the comparison OUTCOMES only, not of the
    // element type, so sorting (score, index) records in the same initial order gives the reference's order — without the
    // pointer chase per comparison (10^7 reads: 0.8 s instead of 3.5 s).  No cheaper route exists: the order inside a tie
    // group depends on the whole run of the algorithm (tools/tie_order_probe.cpp: a pre-sorted input, the stable order and its
    // reverse each keep a different member set than the reference in > 99 % of inputs with a tie group at the cut).
    struct Rec { double score; uint32_t idx; };
    std::vector<Rec> order(n);
    for (uint64_t i = 0; i < n; ++i) order[i] = Rec{fs[i], (uint32_t)i};
    std::sort(order.begin(), order.end(), [](const Rec &a, const Rec &b) { return a.score > b.score; });
    long long so_far = 0;
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = order[k].idx;
        if (passed[i] && so_far < target) so_far += len[i];
        else passed[i] = 0;
    }
    FLX_HIP(ctx, hipMemcpy(d_passed, passed.data(), n, hipMemcpyHostToDevice));
    rep->kept_bases = so_far;
    rep->exact_fallback = 1;
    rep->audited = n;
    return FLX_OK;
}

// =================================================================================================
// SELECT path
// =================================================================================================
static const int FLX_SELECT_BAND_TOO_LARGE = -1000;

// One rank's view when the global stage is sharded (flx_rank_and_cut_sharded_dev / flx_rank_and_cut_comm_dev): the
// statistics are global, the final scores / keys / pass flags are those of the local reads2 entries [first, first + n),
// and every quantity the selection needs from the other ranks is a SUM of 64-bit integers.  Two transports:
//   * the context's RCCL communicator (comm.hip): ncclAllReduce on the device buffer, on the context's stream, no host
//     synchronisation — the 8 selection passes then run back to back;
//   * the caller's host callback (tests, torch.distributed/gloo): the buffer goes through the host.
struct Shard {
    flx_allreduce_u64_fn reduce = nullptr;
    void *user = nullptr;
    bool use_comm = false;
    uint64_t first = 0;
    int rank = 0, world = 1;
    bool have_passed_bases = false;  // comm.hip sums them together with the shard sizes
    uint64_t passed_bases = 0;
    bool sharded() const { return reduce != nullptr || use_comm; }
    // sum of a HOST buffer over all ranks
    int sum(flx_ctx *ctx, uint64_t *buf, uint64_t count) const {
        if (!sharded() || count == 0) return FLX_OK;  // (an empty exchange is empty on every rank)
        if (use_comm) return flx_comm_allreduce_u64_host(ctx, buf, count);
        if (reduce(user, buf, count) != 0) return flx_fail(ctx, FLX_ERR_STATE, "all-reduce callback failed");
        return FLX_OK;
    }
    // sum of a DEVICE buffer over all ranks, in stream order
    int sum_dev(flx_ctx *ctx, uint64_t *d_buf, uint64_t count) const {
        if (!sharded() || count == 0) return FLX_OK;
        if (use_comm) return flx_comm_allreduce_u64_dev(ctx, d_buf, count);
        std::vector<uint64_t> h(count);
        FLX_HIP(ctx, hipMemcpyAsync(h.data(), d_buf, count * 8, hipMemcpyDeviceToHost, ctx->stream));
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (reduce(user, h.data(), count) != 0) return flx_fail(ctx, FLX_ERR_STATE, "all-reduce callback failed");
        FLX_HIP(ctx, hipMemcpyAsync(d_buf, h.data(), count * 8, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FLX_OK;
    }
};

static double key_to_score(uint64_t k) {
    uint64_t a = ~k;  // ascending key
    uint64_t b = (a >> 63) ? (a & 0x7fffffffffffffffull) : ~a;
    double v;
    memcpy(&v, &b, 8);
    return v;
}

using TimeScope = flx_time_scope;  // (flx_internal.h)

// ---- the boundary audit, shared by both cut implementations ---------------------------------------------------------
// Candidates = every read whose DEVICE score lies within the band around the crossing score, with its EXACT score (host
// libm).  The walk of main.cpp:251-257 is replayed over them in exact order (descending exact score; equal scores in
// device order).  Returns true when the outcome depends on the order inside a group of EQUAL exact scores — which only the
// reference's own std::sort (unstable, libstdc++ introsort) can settle: the target is reached inside a group with more
// than one passed member and not every order keeps all of them.  (Walking the group in one order and comparing the
// members' decisions is not enough: lengths {10, 20} entering at target - 15 keep both in that order and only the
// second in the other.)
struct Cand { uint64_t idx; uint64_t key; double score; int32_t len; uint8_t was_passed; };
static bool audit_walk(const std::vector<Cand> &cand, long long weight_before, long long target, std::vector<uint8_t> &keep,
                       long long *so_far_out) {
    const size_t m = cand.size();
    std::vector<size_t> ord(m);
    std::iota(ord.begin(), ord.end(), (size_t)0);
    std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
        if (cand[x].score != cand[y].score) return cand[x].score > cand[y].score;
        if (cand[x].key != cand[y].key) return cand[x].key < cand[y].key;
        return cand[x].idx < cand[y].idx;
    });
    keep.assign(m, 0);
    long long so_far = weight_before;
    bool order_dependent = false;
    for (size_t k = 0; k < m;) {
        size_t e = k;
        long long sum = 0, min_len = std::numeric_limits<long long>::max();
        int passed_n = 0;
        while (e < m && cand[ord[e]].score == cand[ord[k]].score) {
            const Cand &c = cand[ord[e]];
            if (c.was_passed) {
                ++passed_n;
                sum += c.len;
                min_len = std::min<long long>(min_len, c.len);
            }
            ++e;
        }
        // before >= target: no member is kept in any order; before + sum - min_len < target: every member is kept in any
        // order (even the shortest one, walked last, still starts below the target); anything in between depends on it
        if (passed_n > 1 && so_far < target && so_far + sum - min_len >= target) order_dependent = true;
        for (size_t i = k; i < e; ++i) {
            const Cand &c = cand[ord[i]];
            if (c.was_passed && so_far < target) {
                so_far += c.len;
                keep[ord[i]] = 1;
            }
        }
        k = e;
    }
    *so_far_out = so_far;
    return order_dependent;
}

static int cut_by_select(flx_ctx *ctx, uint64_t n, const double *mean, const double *window, const int32_t *length,
                         uint8_t *passed, const NormArgs &s, int64_t target, void *d_final_score, flx_cut_report *rep,
                         const Shard &sh = Shard()) {
    hipStream_t st = ctx->stream;
    const unsigned nb = (unsigned)((n + 255) / 256);
    const unsigned cap = 1u << 16;
    const size_t bins_bytes = 8 * 264 * 8;  // 8 passes x (256 bins + flag slot + padding)
    const size_t bytes = n * 8 + bins_bytes + 256 + (size_t)cap * (4 + sizeof(BandRec) + 4 + 1) + 1024;
    void *scr;
    FLX_CHECK(flx_scratch(ctx, bytes, &scr));
    char *p = (char *)scr;
    uint64_t *keys = (uint64_t *)p; p += n * 8;
    unsigned long long *bins = (unsigned long long *)p; p += bins_bytes;  // pass q at bins + 264 q
    SelState *d_state = (SelState *)p; p += 64;
    unsigned long long *d_acc = (unsigned long long *)p; p += 192;       // [0] band count, [1] weight before the band
    BandRec *d_recs = (BandRec *)p; p += (size_t)cap * sizeof(BandRec);
    uint32_t *band_idx = (uint32_t *)p; p += (size_t)cap * 4;
    uint32_t *d_set_idx = (uint32_t *)p; p += (size_t)cap * 4;
    uint8_t *d_set_val = (uint8_t *)p;

    FLX_HIP(ctx, hipMemsetAsync(bins, 0, bins_bytes + 256, st));
    {
        SelState init = {0ull, (long long)target, 0u, 0u};
        FLX_HIP(ctx, hipMemcpyAsync(d_state, &init, sizeof init, hipMemcpyHostToDevice, st));
    }
    {
        TimeScope t(ctx, "flx_rank_final_score");
        if (n)  // the NaN flag is OR-ed into the flag slot of the first histogram and summed over the ranks with it
            hipLaunchKernelGGL(k_final_score, dim3(nb), dim3(256), 0, st, n, mean, window, length, s, (double *)d_final_score, keys,
                               (uint32_t *)nullptr, (unsigned int *)(bins + 256));
    }

    // ---- 8 weighted histogram passes, most significant byte first; the crossing byte is picked on the device ----------
    TimeScope tsel(ctx, "flx_rank_select");
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 2048));
    for (int pass = 0; pass < 8; ++pass) {
        unsigned long long *b = bins + 264 * pass;
        hipLaunchKernelGGL(k_select_hist, dim3(grid), dim3(256), 0, st, n, keys, length, passed, d_state, pass, b);
        FLX_CHECK(sh.sum_dev(ctx, (uint64_t *)b, 257));
        hipLaunchKernelGGL(k_select_decide, dim3(1), dim3(256), 0, st, b, d_state, pass);
    }
    SelState h_state;
    FLX_HIP(ctx, hipMemcpyAsync(&h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    if (h_state.nan) {
        // NaN scores (stdev == 0 -> 0/0, main.cpp:192-206, or 0/0 window ratios): the reference's comparator is
        // inconsistent and its outcome is whatever libstdc++'s introsort does on reads2 order -> host path.
        tsel.end();
        if (sh.sharded()) return FLX_NEED_REPLICATED;
        return exact_host_cut(ctx, n, mean, window, length, passed, nullptr, nullptr, s, target, rep);
    }
    if (h_state.fail)  // cannot happen when 0 < target < passed_bases
        return flx_fail(ctx, FLX_ERR_STATE, "radix select ran out of weight (target %lld)", (long long)target);
    const uint64_t key_star = h_state.prefix;  // key of the read at which the walk reaches the target
    const double sp = key_to_score(key_star);

    // ---- band around the crossing score: everything the reference might order differently -------------------
    const double kBand = 1e-11;  // relative; the device pow is good to a few ulp (1e-16), so this is generous
    const double band = std::fabs(sp) * kBand + 1e-300;
    const uint64_t k_lo = ~key_ascending(sp + band), k_hi = ~key_ascending(sp - band);  // descending keys: lo = best score
    hipLaunchKernelGGL(k_select_band, dim3(grid), dim3(256), 0, st, n, keys, length, passed, k_lo, k_hi, band_idx,
                       (unsigned int *)d_acc, cap, d_acc + 1);
    unsigned long long h_acc[2] = {0, 0};
    FLX_HIP(ctx, hipMemcpyAsync(h_acc, d_acc, 16, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    const unsigned local_n = (unsigned)(h_acc[0] & 0xffffffffull);
    // the band's records in ONE gather kernel + ONE copy; exact scores with the host libm (what the reference computes)
    // (a rank whose own band already exceeds the capacity gathers nothing but still takes part in the sum below: every rank
    // must issue the same exchanges, and the summed sizes then send all of them to the sort path together)
    std::vector<BandRec> recs(local_n <= cap ? local_n : 0);
    if (local_n && local_n <= cap) {
        hipLaunchKernelGGL(k_band_gather, dim3((local_n + 255) / 256), dim3(256), 0, st, local_n, band_idx, keys, mean, window,
                           length, passed, d_recs);
        FLX_HIP(ctx, hipMemcpyAsync(recs.data(), d_recs, (size_t)local_n * sizeof(BandRec), hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipStreamSynchronize(st));
        std::sort(recs.begin(), recs.end(), [](const BandRec &x, const BandRec &y) { return x.idx < y.idx; });
    }
    // five words per candidate: global reads2 index, key, exact score bits, length, pre-cut flag
    auto put = [&](uint64_t *w, const BandRec &b) {
        const double sc = host_final_score(b.len, b.mean, b.window, s);
        w[0] = sh.first + b.idx;
        w[1] = b.key;
        memcpy(&w[2], &sc, 8);
        w[3] = (uint64_t)(uint32_t)b.len;
        w[4] = b.was_passed;
    };
    // ONE sum carries the band sizes of every rank (own slot filled, the rest zero), the weight in front of the band and,
    // in fixed slots of kInline candidates per rank, the candidates themselves — the band is a 1e-11 neighbourhood of the
    // crossing score, normally a handful of reads.  Only a rank with more than kInline members forces a second exchange.
    const unsigned kInline = 32;
    const size_t head = (size_t)sh.world + 1;
    std::vector<uint64_t> counts(head + (sh.sharded() ? (size_t)sh.world * kInline * 5 : 0), 0);
    counts[sh.rank] = local_n;
    counts[sh.world] = h_acc[1];
    if (sh.sharded() && local_n <= kInline)
        for (unsigned i = 0; i < local_n; ++i) put(&counts[head + ((size_t)sh.rank * kInline + i) * 5], recs[i]);
    FLX_CHECK(sh.sum(ctx, counts.data(), counts.size()));
    uint64_t band_total = 0, my_at = 0;
    bool all_inline = sh.sharded();
    for (int r = 0; r < sh.world; ++r) {
        if (r == sh.rank) my_at = band_total;
        band_total += counts[r];
        if (counts[r] > kInline) all_inline = false;
    }
    const long long weight_before = (long long)counts[sh.world];
    if (band_total > cap) {  // huge tie group (e.g. millions of duplicate reads): let the sort path handle it
        tsel.end();
        return sh.sharded() ? FLX_NEED_REPLICATED : FLX_SELECT_BAND_TOO_LARGE;
    }
    const unsigned band_n = (unsigned)band_total;
    std::vector<uint64_t> wire((size_t)band_n * 5, 0);
    if (all_inline) {
        uint64_t at = 0;
        for (int r = 0; r < sh.world; ++r)
            for (uint64_t i = 0; i < counts[r]; ++i, ++at)
                memcpy(&wire[at * 5], &counts[head + ((size_t)r * kInline + i) * 5], 40);
    } else {
        for (unsigned i = 0; i < local_n; ++i) put(&wire[(my_at + i) * 5], recs[i]);
        FLX_CHECK(sh.sum(ctx, wire.data(), wire.size()));
    }
    std::vector<Cand> cand(band_n);
    for (unsigned i = 0; i < band_n; ++i) {
        const uint64_t *w = &wire[(size_t)i * 5];
        Cand &c = cand[i];
        c.idx = w[0];
        c.key = w[1];
        memcpy(&c.score, &w[2], 8);
        c.len = (int32_t)(uint32_t)w[3];
        c.was_passed = (uint8_t)w[4];
    }
    std::vector<uint8_t> keep;
    long long so_far = 0;
    const bool order_dependent = audit_walk(cand, weight_before, target, keep, &so_far);
    tsel.end();
    if (order_dependent) {
        if (sh.sharded()) return FLX_NEED_REPLICATED;
        return exact_host_cut(ctx, n, mean, window, length, passed, nullptr, nullptr, s, target, rep);
    }

    // ---- mark: better than the band -> unchanged; band and worse -> fail; kept band members -> back on ----------
    if (n) hipLaunchKernelGGL(k_select_mark, dim3(nb), dim3(256), 0, st, n, keys, k_lo, passed);
    {   // this rank's members sit at [my_at, my_at + local_n) of the global band: one upload, one scatter kernel
        std::vector<uint32_t> set_idx;
        for (unsigned i = 0; i < local_n; ++i)
            if (keep[my_at + i]) set_idx.push_back((uint32_t)(cand[my_at + i].idx - sh.first));
        if (!set_idx.empty()) {
            const unsigned m = (unsigned)set_idx.size();
            std::vector<uint8_t> ones(m, 1);
            FLX_HIP(ctx, hipMemcpyAsync(d_set_idx, set_idx.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
            FLX_HIP(ctx, hipMemcpyAsync(d_set_val, ones.data(), m, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_scatter_flags, dim3((m + 255) / 256), dim3(256), 0, st, m, d_set_idx, d_set_val, passed);
            FLX_HIP(ctx, hipStreamSynchronize(st));  // the host vectors go out of scope
        }
    }
    FLX_HIP(ctx, hipStreamSynchronize(st));
    rep->kept_bases = so_far;  // "keeping N bp", main.cpp:258
    rep->audited = band_n;
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}

// =================================================================================================
// SORT path: radix sort + exclusive scan + binary search, then the boundary audit on the sorted neighbourhood
// =================================================================================================
static int cut_by_sort(flx_ctx *ctx, uint64_t n, const double *mean, const double *window, const int32_t *length,
                       uint8_t *passed, const NormArgs &s, int64_t target, void *d_final_score, flx_cut_report *rep) {
    hipStream_t st = ctx->stream;
    const unsigned nb = (unsigned)((n + 255) / 256);
    // scratch layout: keys[2][n] u64 | vals[2][n] u32 | weights[n] i64 | excl[n] i64 | sort workspace
    const size_t sort_ws = flx_radix_sort_workspace(n);
    const size_t bytes = n * (24 + 8 + 8 + 8) + ((n + 255) & ~(size_t)255) + 256 + sort_ws;
    void *scr;
    FLX_CHECK(flx_scratch(ctx, bytes, &scr));
    char *p = (char *)scr;
    uint64_t *keys0 = (uint64_t *)p; p += n * 8;
    uint64_t *keys1 = (uint64_t *)p; p += n * 8;
    uint64_t *keys_orig = (uint64_t *)p; p += n * 8;  // keys in read order (the sort ping-pongs keys0/keys1)
    int64_t *wts = (int64_t *)p; p += n * 8;
    int64_t *excl = (int64_t *)p; p += n * 8;
    uint32_t *vals0 = (uint32_t *)p; p += n * 4;
    uint32_t *vals1 = (uint32_t *)p; p += n * 4;
    uint8_t *pre_sorted = (uint8_t *)p; p += (n + 255) & ~(size_t)255;
    unsigned long long *d_acc = (unsigned long long *)p; p += 256;
    void *sort_tmp = p;

    FLX_HIP(ctx, hipMemsetAsync(d_acc, 0, 32, st));
    {
        TimeScope t(ctx, "flx_rank_final_score");
        hipLaunchKernelGGL(k_final_score, dim3(nb), dim3(256), 0, st, n, mean, window, length, s,
                           (double *)d_final_score, keys0, vals0, (unsigned int *)(d_acc + 2));
    }

    FLX_HIP(ctx, hipMemcpyAsync(keys_orig, keys0, n * 8, hipMemcpyDeviceToDevice, st));

    // ---- a24: device radix sort (stable, descending score) -------------------------------------
    uint64_t *skeys = nullptr;
    uint32_t *svals = nullptr;
    FLX_CHECK(flx_radix_sort_pairs(ctx, n, keys0, keys1, vals0, vals1, sort_tmp, sort_ws, &skeys, &svals));

    // ---- a25: cut walk as an exclusive scan ------------------------------------------------------
    TimeScope tcut(ctx, "flx_rank_cut");
    hipLaunchKernelGGL(k_cut_weights, dim3(nb), dim3(256), 0, st, n, svals, length, passed, wts, pre_sorted);
    FLX_CHECK(flx_exclusive_scan_i64(ctx, n, wts, excl, sort_tmp, sort_ws));

    hipLaunchKernelGGL(k_cut_find, dim3(1), dim3(64), 0, st, n, excl, pre_sorted, target, d_acc + 1);
    unsigned long long h_acc[3] = {0, 0, 0};
    FLX_HIP(ctx, hipMemcpyAsync(h_acc, d_acc, 24, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    if (h_acc[1] == 0) {  // nothing can be kept (not reachable when 0 < target < passed_bases); fail everything
        FLX_HIP(ctx, hipMemsetAsync(passed, 0, n, st));
        tcut.end();
        FLX_HIP(ctx, hipStreamSynchronize(st));
        rep->kept_bases = 0;
        return FLX_OK;
    }
    {
        const uint64_t ps = h_acc[1] - 1;
        uint64_t kstar;
        uint32_t istar;
        int64_t ex_w[2];
        FLX_HIP(ctx, hipMemcpyAsync(&kstar, skeys + ps, 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&istar, svals + ps, 4, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&ex_w[0], excl + ps, 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&ex_w[1], wts + ps, 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_cut_mark, dim3(nb), dim3(256), 0, st, n, keys_orig, kstar, istar, passed);
        tcut.end();
        FLX_HIP(ctx, hipStreamSynchronize(st));
        rep->kept_bases = ex_w[0] + ex_w[1];  // "keeping N bp", main.cpp:258
    }
    // Any NaN score (stdev == 0 -> 0/0 at main.cpp:206, or 0/0 window ratios) makes the reference's comparator
    // inconsistent; its outcome is then whatever libstdc++'s introsort does on reads2 order.  Reproduce exactly
    // that on the host instead of guessing.
    if (h_acc[2] & 1ull) return exact_host_cut(ctx, n, mean, window, length, passed, svals, pre_sorted, s, target, rep);

    // ---- boundary audit ------------------------------------------------------------------------
    // p = sorted position of the read that crossed the target.  Every read whose DEVICE score is within
    // a relative band of score[p] could be ordered differently by the reference (host libm pow); re-score
    // those with the host libm and re-decide the cut among them.
    if (h_acc[1] == 0) return FLX_OK;  // nothing kept (cannot happen when target > 0 and passed_bases > target)
    const uint64_t pstar = h_acc[1] - 1;
    const double kBand = 1e-11;  // relative; device pow is good to a few ulp (1e-16), so this is generous
    const uint64_t W = 256;
    uint64_t lo = pstar > W ? pstar - W : 0, hi = std::min<uint64_t>(n, pstar + W + 1);
    for (;;) {
        const uint64_t m = hi - lo;
        std::vector<uint64_t> hk(m);
        std::vector<uint32_t> hv(m);
        std::vector<int64_t> hex(m);
        FLX_HIP(ctx, hipMemcpy(hk.data(), skeys + lo, m * 8, hipMemcpyDeviceToHost));
        FLX_HIP(ctx, hipMemcpy(hv.data(), svals + lo, m * 4, hipMemcpyDeviceToHost));
        FLX_HIP(ctx, hipMemcpy(hex.data(), excl + lo, m * 8, hipMemcpyDeviceToHost));
        auto key_to_score = [](uint64_t k) {
            uint64_t a = ~k;  // ascending key
            uint64_t b = (a >> 63) ? (a & 0x7fffffffffffffffull) : ~a;
            double v;
            memcpy(&v, &b, 8);
            return v;
        };
        const double sp = key_to_score(hk[pstar - lo]);
        if (std::isnan(sp)) {
            // NaN scores (stdev == 0, main.cpp:192-195 then 0/0): every comparison is false, the order is the
            // reference's tie order -> host path.
            return exact_host_cut(ctx, n, mean, window, length, passed, svals, pre_sorted, s, target, rep);
        }
        const double band = std::fabs(sp) * kBand + 1e-300;
        // band limits inside the window
        uint64_t a = pstar - lo, b = pstar - lo;
        while (a > 0 && std::fabs(key_to_score(hk[a - 1]) - sp) <= band) --a;
        while (b + 1 < m && std::fabs(key_to_score(hk[b + 1]) - sp) <= band) ++b;
        const bool open_lo = (a == 0 && lo > 0), open_hi = (b + 1 == m && hi < n);
        if (open_lo || open_hi) {  // band reaches the window edge: widen and retry
            const uint64_t grow = (hi - lo) * 4;
            lo = lo > grow ? lo - grow : 0;
            hi = std::min<uint64_t>(n, hi + grow);
            continue;
        }
        const uint64_t nb_band = b - a + 1;
        rep->audited = nb_band;
        if (nb_band == 1) return FLX_OK;  // only the crossing read itself: nothing can reorder

        // the band's records in ONE gather kernel + ONE copy, re-scored with the host libm
        if (nb_band > 0xffffffffull) return exact_host_cut(ctx, n, mean, window, length, passed, svals, pre_sorted, s, target, rep);
        const unsigned mb = (unsigned)nb_band;
        flx_dbuf d_recs, d_idx, d_val;
        FLX_CHECK(flx_dalloc(ctx, d_recs, (size_t)mb * sizeof(BandRec)));
        hipLaunchKernelGGL(k_band_gather_sorted, dim3((mb + 255) / 256), dim3(256), 0, st, mb, lo + a, svals, skeys, mean, window,
                           length, pre_sorted, d_recs.as<BandRec>());
        std::vector<BandRec> recs(mb);
        FLX_HIP(ctx, hipMemcpyAsync(recs.data(), d_recs.p, (size_t)mb * sizeof(BandRec), hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipStreamSynchronize(st));
        std::vector<Cand> cand(mb);
        for (unsigned i = 0; i < mb; ++i) {
            const BandRec &r = recs[i];
            cand[i].idx = r.idx;
            cand[i].key = r.key;
            cand[i].len = r.len;
            cand[i].was_passed = (uint8_t)r.was_passed;
            cand[i].score = host_final_score(r.len, r.mean, r.window, s);
        }
        // exact order inside the band, walk, and the test whether only the reference's std::sort can decide (audit_walk)
        std::vector<uint8_t> keep;
        long long so_far = 0;
        if (audit_walk(cand, hex[a], target, keep, &so_far))
            return exact_host_cut(ctx, n, mean, window, length, passed, svals, pre_sorted, s, target, rep);
        // patch the flags of the band (one upload, one scatter); reads after it all fail, reads before it are unchanged
        std::vector<uint32_t> idx(mb);
        for (unsigned i = 0; i < mb; ++i) idx[i] = (uint32_t)cand[i].idx;
        FLX_CHECK(flx_dalloc(ctx, d_idx, (size_t)mb * 4));
        FLX_CHECK(flx_dalloc(ctx, d_val, mb));
        FLX_HIP(ctx, hipMemcpyAsync(d_idx.p, idx.data(), (size_t)mb * 4, hipMemcpyHostToDevice, st));
        FLX_HIP(ctx, hipMemcpyAsync(d_val.p, keep.data(), mb, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scatter_flags, dim3((mb + 255) / 256), dim3(256), 0, st, mb, d_idx.as<uint32_t>(), d_val.as<uint8_t>(), passed);
        FLX_HIP(ctx, hipStreamSynchronize(st));
        rep->kept_bases = so_far;
        return FLX_OK;
    }
}


int flx_passed_bases_async(flx_ctx *ctx, uint64_t n, const int32_t *d_length, const uint8_t *d_passed, uint64_t *d_out) {
    if (n == 0) return FLX_OK;
    TimeScope t(ctx, "flx_rank_passed_bases");
    hipLaunchKernelGGL(k_passed_bases, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, n,
                       d_length, d_passed, (unsigned long long *)d_out);
    return FLX_OK;
}

// Global stage on one rank's view: statistics over ALL reads2 entries (mean_all[0, n_total)), everything else on the
// local entries [sh.first, sh.first + n).  Single rank: n == n_total, sh.first == 0, no exchange.
static int rank_and_cut_impl(flx_ctx *ctx, uint64_t n_total, const double *mean_all, uint64_t n, const double *window,
                             const int32_t *length, uint8_t *passed, double lw, double mw, double ww,
                             int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                             int64_t total_bases, void *d_final_score, flx_cut_report *rep, const Shard &sh) {
    const double *mean = mean_all + sh.first;
    hipStream_t st = ctx->stream;
    const bool cutting = target_bases_set || keep_percent_set;

    // ---- a20: statistics (exact serial folds) -------------------------------------------------
    flx_stats stats;
    FLX_CHECK(flx_exact_stats(ctx, n_total, mean_all, &stats));
    NormArgs s;
    s.qmean = stats.mean;
    s.qstd = stats.stdev;
    if (stats.stdev > 0.0) {  // main.cpp:188-195
        s.zmin = (stats.min - stats.mean) / stats.stdev;
        const double zmax = (stats.max - stats.mean) / stats.stdev;
        s.zspan = zmax - s.zmin;
        rep->max_z = zmax;
    } else {
        s.zmin = 1.0;
        s.zspan = 1.0 - 1.0;
        rep->max_z = 1.0;
    }
    s.lw = lw; s.mw = mw; s.ww = ww;
    rep->mean_quality = stats.mean;
    rep->stdev_quality = stats.stdev;
    rep->min_z = s.zmin;

    // ---- early outs that need no sort ----------------------------------------------------------
    int64_t target = 0;
    bool need_sort = false;
    if (cutting && n_total) {
        void *scr;
        FLX_CHECK(flx_scratch(ctx, 64, &scr));
        unsigned long long *d_acc = (unsigned long long *)scr;
        unsigned long long passed_bases = sh.passed_bases;
        if (!sh.have_passed_bases) {
            FLX_HIP(ctx, hipMemsetAsync(d_acc, 0, 8, st));
            FLX_CHECK(flx_passed_bases_async(ctx, n, length, passed, (uint64_t *)d_acc));
            FLX_HIP(ctx, hipMemcpyAsync(&passed_bases, d_acc, 8, hipMemcpyDeviceToHost, st));
            FLX_HIP(ctx, hipStreamSynchronize(st));
            uint64_t pb = passed_bases;
            FLX_CHECK(sh.sum(ctx, &pb, 1));
            passed_bases = pb;
        }
        target = compute_target(target_bases_set, target_bases, keep_percent_set, keep_percent, total_bases);
        rep->target_bases = target;
        if (target >= total_bases) rep->outcome = FLX_CUT_NOT_ENOUGH;
        else if (target >= (int64_t)passed_bases) rep->outcome = FLX_CUT_ALREADY_BELOW;
        else { rep->outcome = FLX_CUT_SORTED; need_sort = true; }
    } else if (cutting) {
        target = compute_target(target_bases_set, target_bases, keep_percent_set, keep_percent, total_bases);
        rep->target_bases = target;
        rep->outcome = target >= total_bases ? FLX_CUT_NOT_ENOUGH : FLX_CUT_ALREADY_BELOW;
    }
    if (n_total == 0) return FLX_OK;

    // ---- a21/a22: normalise + final score (+ keys) --------------------------------------------
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (!need_sort) {
        if (d_final_score && n) {
            TimeScope t(ctx, "flx_rank_final_score");
            hipLaunchKernelGGL(k_final_score, dim3(nb), dim3(256), 0, st, n, mean, window, length, s,
                               (double *)d_final_score, (uint64_t *)nullptr, (uint32_t *)nullptr,
                               (unsigned int *)nullptr);
            t.end();
            FLX_HIP(ctx, hipStreamSynchronize(st));
        }
        return FLX_OK;
    }

    {
        const char *x = getenv("FLX_RANK_EXACT");  // test / bench hook: take the tie fallback (the reference's std::sort on the host)
        if (x && x[0] == '1') {
            if (sh.sharded()) return FLX_NEED_REPLICATED;
            return exact_host_cut(ctx, n, mean, window, length, passed, nullptr, nullptr, s, target, rep);
        }
        const char *e = getenv("FLX_RANK_SORT");  // test hook / fallback selector
        if (e && e[0] == '1') {
            if (sh.sharded()) return FLX_NEED_REPLICATED;  // the sort path wants every record on one device
            return cut_by_sort(ctx, n, mean, window, length, passed, s, target, d_final_score, rep);
        }
    }
    const int rc = cut_by_select(ctx, n, mean, window, length, passed, s, target, d_final_score, rep, sh);
    if (rc == FLX_SELECT_BAND_TOO_LARGE) return cut_by_sort(ctx, n, mean, window, length, passed, s, target, d_final_score, rep);
    return rc;
}

extern "C" int flx_rank_and_cut_dev(flx_ctx *ctx, uint64_t n, const void *d_mean_q, const void *d_window_q,
                                    const void *d_length, void *d_passed, double lw, double mw, double ww,
                                    int target_bases_set, int64_t target_bases, int keep_percent_set,
                                    double keep_percent, int64_t total_bases, void *d_final_score,
                                    flx_cut_report *rep) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!rep) return flx_fail(ctx, FLX_ERR_INVALID, "report must not be NULL");
    memset(rep, 0, sizeof *rep);
    if (n > 0xffffffffull) return flx_fail(ctx, FLX_ERR_INVALID, "at most 2^32-1 reads");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    return rank_and_cut_impl(ctx, n, (const double *)d_mean_q, n, (const double *)d_window_q, (const int32_t *)d_length,
                             (uint8_t *)d_passed, lw, mw, ww, target_bases_set, target_bases, keep_percent_set,
                             keep_percent, total_bases, d_final_score, rep, Shard());
}

extern "C" int flx_rank_and_cut_sharded_dev(flx_ctx *ctx, uint64_t n_total, const void *d_mean_q_all, uint64_t first,
                                            uint64_t n_local, const void *d_window_q, const void *d_length,
                                            void *d_passed, double lw, double mw, double ww, int target_bases_set,
                                            int64_t target_bases, int keep_percent_set, double keep_percent,
                                            int64_t total_bases, void *d_final_score, int rank, int world,
                                            flx_allreduce_u64_fn reduce, void *user, flx_cut_report *rep) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!rep) return flx_fail(ctx, FLX_ERR_INVALID, "report must not be NULL");
    memset(rep, 0, sizeof *rep);
    if (n_total > 0xffffffffull) return flx_fail(ctx, FLX_ERR_INVALID, "at most 2^32-1 reads");
    if (first > n_total || n_local > n_total - first) return flx_fail(ctx, FLX_ERR_INVALID, "shard [first, first + n_local) outside [0, n_total)");
    if (world < 1 || rank < 0 || rank >= world) return flx_fail(ctx, FLX_ERR_INVALID, "bad rank / world");
    if (world > 1 && !reduce) return flx_fail(ctx, FLX_ERR_INVALID, "world > 1 needs an all-reduce callback");
    if (n_total && !d_mean_q_all) return flx_fail(ctx, FLX_ERR_INVALID, "NULL mean array");
    if (n_local && (!d_window_q || !d_length || !d_passed)) return flx_fail(ctx, FLX_ERR_INVALID, "NULL local array");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    Shard sh;
    sh.reduce = reduce; sh.user = user; sh.first = first; sh.rank = rank; sh.world = world;
    return rank_and_cut_impl(ctx, n_total, (const double *)d_mean_q_all, n_local, (const double *)d_window_q,
                             (const int32_t *)d_length, (uint8_t *)d_passed, lw, mw, ww, target_bases_set, target_bases,
                             keep_percent_set, keep_percent, total_bases, d_final_score, rep, sh);
}

int flx_rank_and_cut_sharded_comm(flx_ctx *ctx, uint64_t n_total, const double *d_mean_all, uint64_t first, uint64_t n_local,
                                  const double *d_window, const int32_t *d_length, uint8_t *d_passed, double lw, double mw,
                                  double ww, int target_bases_set, int64_t target_bases, int keep_percent_set,
                                  double keep_percent, int64_t total_bases, void *d_final_score, int rank, int world,
                                  uint64_t passed_bases_all_ranks, flx_cut_report *rep) {
    memset(rep, 0, sizeof *rep);
    Shard sh;
    sh.use_comm = true; sh.first = first; sh.rank = rank; sh.world = world;
    sh.have_passed_bases = true; sh.passed_bases = passed_bases_all_ranks;
    return rank_and_cut_impl(ctx, n_total, d_mean_all, n_local, d_window, d_length, d_passed, lw, mw, ww, target_bases_set,
                             target_bases, keep_percent_set, keep_percent, total_bases, d_final_score, rep, sh);
}

extern "C" int flx_rank_and_cut(flx_ctx *ctx, uint64_t n, const double *mean_q, const double *window_q,
                                const int32_t *length, uint8_t *passed, double lw, double mw, double ww,
                                int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                                int64_t total_bases, double *final_score, flx_cut_report *rep) {
    if (!ctx) return FLX_ERR_INVALID;
    if (n && (!mean_q || !window_q || !length || !passed)) return flx_fail(ctx, FLX_ERR_INVALID, "NULL input array");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    flx_dbuf d_mean, d_win, d_len, d_pass, d_fs;
    FLX_CHECK(flx_dalloc(ctx, d_mean, n * 8));
    FLX_CHECK(flx_dalloc(ctx, d_win, n * 8));
    FLX_CHECK(flx_dalloc(ctx, d_len, n * 4));
    FLX_CHECK(flx_dalloc(ctx, d_pass, n));
    if (final_score) FLX_CHECK(flx_dalloc(ctx, d_fs, n * 8));
    if (n) {
        FLX_HIP(ctx, hipMemcpyAsync(d_mean.p, mean_q, n * 8, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(d_win.p, window_q, n * 8, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(d_len.p, length, n * 4, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(d_pass.p, passed, n, hipMemcpyHostToDevice, ctx->stream));
    }
    FLX_CHECK(flx_rank_and_cut_dev(ctx, n, d_mean.p, d_win.p, d_len.p, d_pass.p, lw, mw, ww, target_bases_set,
                                   target_bases, keep_percent_set, keep_percent, total_bases,
                                   final_score ? d_fs.p : nullptr, rep));
    if (n) {
        FLX_HIP(ctx, hipMemcpyAsync(passed, d_pass.p, n, hipMemcpyDeviceToHost, ctx->stream));
        if (final_score) FLX_HIP(ctx, hipMemcpyAsync(final_score, d_fs.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FLX_OK;
}
