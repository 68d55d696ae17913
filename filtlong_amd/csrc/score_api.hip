// score_api.hip — C-ABI entry points for seam 2 (per-read scoring; replaces Read::Read,
// reference src/read.cpp:25-144, in batched form).
#include <chrono>

#include "flx_internal.h"
#include "kmerset.h"

namespace {
// FLX_API_TIMING=1: wall-clock of the host entry point's phases on stderr (diagnostics only)
struct PhaseTimer {
    bool on;
    double t0;
    PhaseTimer() : on(getenv("FLX_API_TIMING") != nullptr), t0(now()) {}
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(const char *what) {
        if (!on) return;
        const double t = now();
        fprintf(stderr, "[flx_score_batch] %-22s %8.3f ms\n", what, (t - t0) * 1e3);
        t0 = t;
    }
};
}  // namespace

static int validate_common(flx_ctx *ctx, uint64_t n_reads, uint64_t plane_bytes, const flx_params *params,
                           const flx_scores *out) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!params || !out) return flx_fail(ctx, FLX_ERR_INVALID, "params/out must not be NULL");
    if (n_reads > 0xffffffffull) return flx_fail(ctx, FLX_ERR_INVALID, "at most 2^32-1 reads per batch");
    if (plane_bytes & 15) return flx_fail(ctx, FLX_ERR_INVALID, "plane_bytes must be a multiple of 16");
    if (n_reads && (!out->mean_q || !out->window_q || !out->passed))
        return flx_fail(ctx, FLX_ERR_INVALID, "mean_q/window_q/passed outputs are required");
    if (params->window_size <= 0) return flx_fail(ctx, FLX_ERR_INVALID, "window_size must be positive");
    return FLX_OK;
}

extern "C" int flx_score_batch_dev(flx_ctx *ctx, const flx_kmerset *set, const void *d_plane, uint64_t plane_bytes,
                                   const void *d_offsets, const void *d_lengths, const void *d_order,
                                   uint64_t n_reads, const flx_params *params, flx_scores *out) {
    FLX_CHECK(validate_common(ctx, n_reads, plane_bytes, params, out));
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    const bool kmer_mode = set && flx_kmerset_size(set) > 0;  // Kmers::empty(), src/kmers.h:34
    if (set && !flx_kmerset_is_final(set)) return flx_fail(ctx, FLX_ERR_STATE, "k-mer set is not finalized");
    if (!kmer_mode) {
        flx_score_out_dev o{out->mean_q, out->window_q, out->passed};
        FLX_CHECK(flx_launch_score_phred(ctx, (const uint8_t *)d_plane, plane_bytes, (const uint64_t *)d_offsets,
                                         (const int32_t *)d_lengths, (const uint32_t *)d_order, n_reads, params, o));
        // Phred mode: first/last stay -1 and there are never children (src/read.cpp:33-34,76)
        if (out->first) FLX_HIP(ctx, hipMemsetAsync(out->first, 0xff, n_reads * sizeof(int32_t), ctx->stream));
        if (out->last) FLX_HIP(ctx, hipMemsetAsync(out->last, 0xff, n_reads * sizeof(int32_t), ctx->stream));
        if (out->child_offsets)
            FLX_HIP(ctx, hipMemsetAsync(out->child_offsets, 0, (n_reads + 1) * sizeof(uint64_t), ctx->stream));
        out->n_children = 0;
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FLX_OK;
    }
    return flx_score_kmer_dev(ctx, set, (const uint8_t *)d_plane, plane_bytes, (const uint64_t *)d_offsets,
                              (const int32_t *)d_lengths, (const uint32_t *)d_order, n_reads, params, out);
}

extern "C" int flx_score_batch(flx_ctx *ctx, const flx_kmerset *set, const uint8_t *plane, uint64_t plane_bytes,
                               const uint64_t *offsets, const int32_t *lengths, const uint32_t *order,
                               uint64_t n_reads, const flx_params *params, flx_scores *out) {
    FLX_CHECK(validate_common(ctx, n_reads, plane_bytes, params, out));
    if (n_reads == 0) {
        out->n_children = 0;
        if (out->child_offsets) out->child_offsets[0] = 0;
        return FLX_OK;
    }
    if (!plane || !offsets || !lengths) return flx_fail(ctx, FLX_ERR_INVALID, "plane/offsets/lengths must not be NULL");
    for (uint64_t i = 0; i < n_reads; ++i) {
        if (lengths[i] < 0 || (offsets[i] & 15) ||
            offsets[i] + (((uint64_t)lengths[i] + 15) & ~15ull) > plane_bytes)
            return flx_fail(ctx, FLX_ERR_INVALID, "read %llu: offset must be 16-byte aligned and inside the plane",
                            (unsigned long long)i);
    }
    PhaseTimer pt;
    pt.mark("validate");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    const bool kmer_mode = set && flx_kmerset_size(set) > 0;
    const bool want_children = kmer_mode && (params->trim || params->split_set);
    if (want_children && !out->child_offsets)
        return flx_fail(ctx, FLX_ERR_INVALID, "trim/split requested but child outputs are NULL");

    flx_dbuf d_plane, d_off, d_len, d_ord, d_mean, d_win, d_pass, d_first, d_last;
    flx_dbuf d_coff, d_crng, d_cmean, d_cwin, d_cpass;
    FLX_CHECK(flx_dalloc(ctx, d_plane, plane_bytes));
    FLX_CHECK(flx_dalloc(ctx, d_off, n_reads * 8));
    FLX_CHECK(flx_dalloc(ctx, d_len, n_reads * 4));
    FLX_CHECK(flx_dalloc(ctx, d_mean, n_reads * 8));
    FLX_CHECK(flx_dalloc(ctx, d_win, n_reads * 8));
    FLX_CHECK(flx_dalloc(ctx, d_pass, n_reads));
    pt.mark("device alloc");
    // the one large transfer.  The runtime's pageable path measures 32 GB/s here (2 GB in 62 ms); a hand-rolled pipeline
    // of 8 threads x 2 pinned 8 MiB chunks was slower (105 ms incl. its staging allocation), so it stays a plain copy.
    FLX_HIP(ctx, hipMemcpyAsync(d_plane.p, plane, plane_bytes, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pt.mark("plane upload");
    FLX_HIP(ctx, hipMemcpyAsync(d_off.p, offsets, n_reads * 8, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(d_len.p, lengths, n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    if (order) {
        FLX_CHECK(flx_dalloc(ctx, d_ord, n_reads * 4));
        FLX_HIP(ctx, hipMemcpyAsync(d_ord.p, order, n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    flx_scores dev = {};
    dev.mean_q = d_mean.as<double>();
    dev.window_q = d_win.as<double>();
    dev.passed = d_pass.as<uint8_t>();
    if (out->first && out->last) {
        FLX_CHECK(flx_dalloc(ctx, d_first, n_reads * 4));
        FLX_CHECK(flx_dalloc(ctx, d_last, n_reads * 4));
        dev.first = d_first.as<int32_t>();
        dev.last = d_last.as<int32_t>();
    }
    if (out->child_offsets) {
        FLX_CHECK(flx_dalloc(ctx, d_coff, (n_reads + 1) * 8));
        dev.child_offsets = d_coff.as<uint64_t>();
        dev.child_capacity = out->child_capacity;
        if (out->child_capacity) {
            FLX_CHECK(flx_dalloc(ctx, d_crng, out->child_capacity * 8));
            FLX_CHECK(flx_dalloc(ctx, d_cmean, out->child_capacity * 8));
            FLX_CHECK(flx_dalloc(ctx, d_cwin, out->child_capacity * 8));
            FLX_CHECK(flx_dalloc(ctx, d_cpass, out->child_capacity));
            dev.child_ranges = d_crng.as<int32_t>();
            dev.child_mean_q = d_cmean.as<double>();
            dev.child_window_q = d_cwin.as<double>();
            dev.child_passed = d_cpass.as<uint8_t>();
        }
    }
    // The outputs start out as a pattern no result can be (0xA5...: a huge negative quality, pass flag 165): device memory that
    // comes back from the allocator holds the previous call's results at the same addresses, and a kernel that wrote nothing
    // would otherwise hand back yesterday's right answers (round 5: that is how the fold kernels with both streams from global
    // memory were found dead — their launch had failed silently while the tests compared stale buffers).
    FLX_HIP(ctx, hipMemsetAsync(d_mean.p, 0xA5, n_reads * 8, ctx->stream));
    FLX_HIP(ctx, hipMemsetAsync(d_win.p, 0xA5, n_reads * 8, ctx->stream));
    FLX_HIP(ctx, hipMemsetAsync(d_pass.p, 0xA5, n_reads, ctx->stream));
    pt.mark("small uploads + alloc");
    int rc = flx_score_batch_dev(ctx, set, d_plane.p, plane_bytes, d_off.p, d_len.p, order ? d_ord.p : nullptr,
                                 n_reads, params, &dev);
    pt.mark("kernels");
    if (rc != FLX_OK) {
        out->n_children = dev.n_children;  // on FLX_ERR_CAPACITY this is the required capacity
        return rc;
    }
    FLX_HIP(ctx, hipMemcpyAsync(out->mean_q, dev.mean_q, n_reads * 8, hipMemcpyDeviceToHost, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(out->window_q, dev.window_q, n_reads * 8, hipMemcpyDeviceToHost, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(out->passed, dev.passed, n_reads, hipMemcpyDeviceToHost, ctx->stream));
    if (dev.first) {
        FLX_HIP(ctx, hipMemcpyAsync(out->first, dev.first, n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(out->last, dev.last, n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    out->n_children = dev.n_children;
    if (dev.child_offsets) {
        FLX_HIP(ctx, hipMemcpyAsync(out->child_offsets, dev.child_offsets, (n_reads + 1) * 8, hipMemcpyDeviceToHost,
                                    ctx->stream));
        if (dev.n_children) {
            const uint64_t nc = dev.n_children;
            FLX_HIP(ctx, hipMemcpyAsync(out->child_ranges, dev.child_ranges, nc * 8, hipMemcpyDeviceToHost, ctx->stream));
            FLX_HIP(ctx, hipMemcpyAsync(out->child_mean_q, dev.child_mean_q, nc * 8, hipMemcpyDeviceToHost, ctx->stream));
            FLX_HIP(ctx, hipMemcpyAsync(out->child_window_q, dev.child_window_q, nc * 8, hipMemcpyDeviceToHost, ctx->stream));
            FLX_HIP(ctx, hipMemcpyAsync(out->child_passed, dev.child_passed, nc, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pt.mark("download");
    return FLX_OK;
}
