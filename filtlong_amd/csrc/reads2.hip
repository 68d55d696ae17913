// reads2.hip — the reads2 gather between seam 2 and seam 3 (reference src/main.cpp:138-147):
//
//     for (auto read : reads) {
//         if (read->m_child_reads.size() == 0) reads2.push_back(read);
//         else for (auto child : read->m_child_reads) reads2.push_back(child);
//     }
//
// File order, every parent that was trimmed / split replaced IN PLACE by its children.  Everything the global stage
// (flx_rank_and_cut*) reads — mean quality, window quality, length, pass flag — is gathered into reads2 order; two optional
// index arrays tell the caller where every entry came from (the output writer needs them, src/main.cpp:285-309).
//
// One exclusive scan (entries per read: max(1, children)) + one scatter kernel, one lane per read.  Children per read are
// few (C4: ~1 on average), and consecutive lanes write consecutive entries, so the stores coalesce.
#include "flx_internal.h"
#include "rank_internal.h"

namespace {

__global__ void k_reads2_sizes(uint64_t n, const uint64_t *child_off, int64_t *sizes) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        sizes[i] = 0;
        return;
    }
    const uint64_t c = child_off ? child_off[i + 1] - child_off[i] : 0;
    sizes[i] = c ? (int64_t)c : 1;
}

struct Reads2Args {
    uint64_t n;
    const int32_t *lengths;
    const double *mean_q, *window_q;
    const uint8_t *passed;
    const uint64_t *child_off;
    const int32_t *child_ranges;
    const double *child_mean_q, *child_window_q;
    const uint8_t *child_passed;
    const int64_t *start2;  // [n + 1]
    uint64_t capacity;
    double *mean2, *window2;
    int32_t *length2;
    uint8_t *passed2;
    uint32_t *parent2;  // may be NULL
    int64_t *child2;    // may be NULL
};

__global__ void __launch_bounds__(256) k_reads2_scatter(const Reads2Args a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t at = (uint64_t)a.start2[i];
    const uint64_t c0 = a.child_off ? a.child_off[i] : 0, c1 = a.child_off ? a.child_off[i + 1] : 0;
    if (c0 == c1) {
        if (at >= a.capacity) return;
        a.mean2[at] = a.mean_q[i];
        a.window2[at] = a.window_q[i];
        a.length2[at] = a.lengths[i];
        a.passed2[at] = a.passed[i];
        if (a.parent2) a.parent2[at] = (uint32_t)i;
        if (a.child2) a.child2[at] = -1;
        return;
    }
    for (uint64_t k = c0; k < c1; ++k) {
        const uint64_t o = at + (k - c0);
        if (o >= a.capacity) return;
        a.mean2[o] = a.child_mean_q[k];
        a.window2[o] = a.child_window_q[k];
        a.length2[o] = a.child_ranges[2 * k + 1] - a.child_ranges[2 * k];  // Read(child): length = end - start, src/read.cpp:131-137
        a.passed2[o] = a.child_passed[k];
        if (a.parent2) a.parent2[o] = (uint32_t)i;
        if (a.child2) a.child2[o] = (int64_t)k;
    }
}

}  // namespace

extern "C" int flx_reads2_gather_dev(flx_ctx *ctx, uint64_t n_reads, const void *d_lengths, const flx_scores *s,
                                     uint64_t capacity, void *d_mean_q2, void *d_window_q2, void *d_length2,
                                     void *d_passed2, void *d_parent2, void *d_child2, uint64_t *n2_out) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!s || !n2_out) return flx_fail(ctx, FLX_ERR_INVALID, "scores / n2 must not be NULL");
    *n2_out = 0;
    if (n_reads == 0) return FLX_OK;
    if (!d_lengths || !s->mean_q || !s->window_q || !s->passed)
        return flx_fail(ctx, FLX_ERR_INVALID, "lengths and the per-read mean_q / window_q / passed arrays are required");
    if (!d_mean_q2 || !d_window_q2 || !d_length2 || !d_passed2)
        return flx_fail(ctx, FLX_ERR_INVALID, "reads2 output arrays must not be NULL");
    const bool children = s->child_offsets != nullptr && s->n_children > 0;
    if (children && (!s->child_ranges || !s->child_mean_q || !s->child_window_q || !s->child_passed))
        return flx_fail(ctx, FLX_ERR_INVALID, "child arrays are required when there are children");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t scan_ws = flx_radix_sort_workspace(n_reads + 1);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    void *scratch = nullptr;
    FLX_CHECK(flx_scratch(ctx, 2 * up((n_reads + 1) * 8) + up(scan_ws), &scratch));
    int64_t *d_sizes = (int64_t *)scratch;
    int64_t *d_start = (int64_t *)((char *)scratch + up((n_reads + 1) * 8));
    void *d_scanws = (char *)scratch + 2 * up((n_reads + 1) * 8);
    const unsigned nb = (unsigned)((n_reads + 1 + 255) / 256);
    flx_time_begin(ctx, "flx_reads2_gather");
    hipLaunchKernelGGL(k_reads2_sizes, dim3(nb), dim3(256), 0, st, n_reads, children ? s->child_offsets : nullptr, d_sizes);
    flx_time_end(ctx);
    FLX_CHECK(flx_exclusive_scan_i64(ctx, n_reads + 1, d_sizes, d_start, d_scanws, scan_ws));
    int64_t n2 = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&n2, d_start + n_reads, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    *n2_out = (uint64_t)n2;
    if ((uint64_t)n2 > capacity)
        return flx_fail(ctx, FLX_ERR_CAPACITY, "reads2 arrays need room for %lld entries (capacity %llu)", (long long)n2,
                        (unsigned long long)capacity);
    Reads2Args a;
    a.n = n_reads;
    a.lengths = (const int32_t *)d_lengths;
    a.mean_q = s->mean_q;
    a.window_q = s->window_q;
    a.passed = s->passed;
    a.child_off = children ? s->child_offsets : nullptr;
    a.child_ranges = s->child_ranges;
    a.child_mean_q = s->child_mean_q;
    a.child_window_q = s->child_window_q;
    a.child_passed = s->child_passed;
    a.start2 = d_start;
    a.capacity = capacity;
    a.mean2 = (double *)d_mean_q2;
    a.window2 = (double *)d_window_q2;
    a.length2 = (int32_t *)d_length2;
    a.passed2 = (uint8_t *)d_passed2;
    a.parent2 = (uint32_t *)d_parent2;
    a.child2 = (int64_t *)d_child2;
    flx_time_begin(ctx, "flx_reads2_gather");
    hipLaunchKernelGGL(k_reads2_scatter, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, a);
    flx_time_end(ctx);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(st));
    return FLX_OK;
}

// Host-array variant: the arrays are staged through the device, like flx_rank_and_cut.
extern "C" int flx_reads2_gather(flx_ctx *ctx, uint64_t n_reads, const int32_t *lengths, const flx_scores *s, uint64_t capacity,
                                 double *mean_q2, double *window_q2, int32_t *length2, uint8_t *passed2, uint32_t *parent2,
                                 int64_t *child2, uint64_t *n2_out) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!s || !n2_out) return flx_fail(ctx, FLX_ERR_INVALID, "scores / n2 must not be NULL");
    *n2_out = 0;
    if (n_reads == 0) return FLX_OK;
    if (!lengths || !s->mean_q || !s->window_q || !s->passed)
        return flx_fail(ctx, FLX_ERR_INVALID, "lengths and the per-read mean_q / window_q / passed arrays are required");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t nc = s->child_offsets ? s->n_children : 0;
    flx_dbuf d_len, d_mean, d_win, d_pass, d_coff, d_crng, d_cmean, d_cwin, d_cpass, o_mean, o_win, o_len, o_pass, o_par, o_chi;
    auto upl = [&](flx_dbuf &b, const void *src, size_t bytes) -> int {
        FLX_CHECK(flx_dalloc(ctx, b, bytes ? bytes : 16));
        if (bytes) FLX_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return FLX_OK;
    };
    FLX_CHECK(upl(d_len, lengths, n_reads * 4));
    FLX_CHECK(upl(d_mean, s->mean_q, n_reads * 8));
    FLX_CHECK(upl(d_win, s->window_q, n_reads * 8));
    FLX_CHECK(upl(d_pass, s->passed, n_reads));
    flx_scores dev = {};
    dev.mean_q = d_mean.as<double>();
    dev.window_q = d_win.as<double>();
    dev.passed = d_pass.as<uint8_t>();
    if (nc) {
        FLX_CHECK(upl(d_coff, s->child_offsets, (n_reads + 1) * 8));
        FLX_CHECK(upl(d_crng, s->child_ranges, nc * 8));
        FLX_CHECK(upl(d_cmean, s->child_mean_q, nc * 8));
        FLX_CHECK(upl(d_cwin, s->child_window_q, nc * 8));
        FLX_CHECK(upl(d_cpass, s->child_passed, nc));
        dev.child_offsets = d_coff.as<uint64_t>();
        dev.child_ranges = d_crng.as<int32_t>();
        dev.child_mean_q = d_cmean.as<double>();
        dev.child_window_q = d_cwin.as<double>();
        dev.child_passed = d_cpass.as<uint8_t>();
        dev.n_children = nc;
    }
    const uint64_t cap = capacity ? capacity : 1;
    FLX_CHECK(flx_dalloc(ctx, o_mean, cap * 8));
    FLX_CHECK(flx_dalloc(ctx, o_win, cap * 8));
    FLX_CHECK(flx_dalloc(ctx, o_len, cap * 4));
    FLX_CHECK(flx_dalloc(ctx, o_pass, cap));
    if (parent2) FLX_CHECK(flx_dalloc(ctx, o_par, cap * 4));
    if (child2) FLX_CHECK(flx_dalloc(ctx, o_chi, cap * 8));
    FLX_CHECK(flx_reads2_gather_dev(ctx, n_reads, d_len.p, &dev, capacity, o_mean.p, o_win.p, o_len.p, o_pass.p,
                                    parent2 ? o_par.p : nullptr, child2 ? o_chi.p : nullptr, n2_out));
    const uint64_t n2 = *n2_out;
    FLX_HIP(ctx, hipMemcpyAsync(mean_q2, o_mean.p, n2 * 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipMemcpyAsync(window_q2, o_win.p, n2 * 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipMemcpyAsync(length2, o_len.p, n2 * 4, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipMemcpyAsync(passed2, o_pass.p, n2, hipMemcpyDeviceToHost, st));
    if (parent2) FLX_HIP(ctx, hipMemcpyAsync(parent2, o_par.p, n2 * 4, hipMemcpyDeviceToHost, st));
    if (child2) FLX_HIP(ctx, hipMemcpyAsync(child2, o_chi.p, n2 * 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    return FLX_OK;
}
