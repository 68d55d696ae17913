// rank_internal.h — building blocks of the global stage (seam 3), internal to the library.
#pragma once
#include "flx_internal.h"

// a20: min / max / mean / stdev of the mean qualities exactly as the reference's serial FP64 folds
// compute them (src/main.cpp:170-186).
struct flx_stats {
    double min, max, sum, mean, sq_sum, stdev;
    unsigned long long serial_chunks;  // diagnostic: 512-element chunks folded serially (binade crossings etc.)
};
int flx_exact_stats(flx_ctx *ctx, uint64_t n, const double *d_mean_q, flx_stats *out);

// Stable LSD radix sort of (u64 key, u32 value) pairs, ascending by key.  keys0/vals0 hold the input;
// keys1/vals1 are the ping-pong buffers; *sorted_keys / *sorted_vals point at whichever buffer holds
// the result.  Runs on ctx->stream (asynchronous).
size_t flx_radix_sort_workspace(uint64_t n);
int flx_radix_sort_pairs(flx_ctx *ctx, uint64_t n, uint64_t *keys0, uint64_t *keys1, uint32_t *vals0, uint32_t *vals1,
                         void *workspace, size_t workspace_bytes, uint64_t **sorted_keys, uint32_t **sorted_vals);

// out[i] = sum_{j<i} in[j]   (int64, exact).  workspace >= flx_radix_sort_workspace(n) suffices.
int flx_exclusive_scan_i64(flx_ctx *ctx, uint64_t n, const int64_t *in, int64_t *out, void *workspace,
                           size_t workspace_bytes);
// in-place exclusive scan of doubles in tree order (APPROXIMATE sums; only used to guess binades in stats.hip)
int flx_exclusive_scan_f64_approx(flx_ctx *ctx, uint64_t n, double *data, void *workspace);
int flx_exclusive_scan_u32(flx_ctx *ctx, uint64_t n, const uint32_t *in, uint32_t *out, void *workspace,
                           size_t workspace_bytes);

// rank.hip: the sharded global stage with the context's RCCL communicator as transport (called by comm.hip once the mean
// qualities of all ranks are gathered in d_mean_all)
int flx_rank_and_cut_sharded_comm(flx_ctx *ctx, uint64_t n_total, const double *d_mean_all, uint64_t first, uint64_t n_local,
                                  const double *d_window, const int32_t *d_length, uint8_t *d_passed, double lw, double mw,
                                  double ww, int target_bases_set, int64_t target_bases, int keep_percent_set,
                                  double keep_percent, int64_t total_bases, void *d_final_score, int rank, int world,
                                  uint64_t passed_bases_all_ranks, flx_cut_report *rep);
// *d_out += sum of length[i] over passed[i] (main.cpp:222-226), in stream order
int flx_passed_bases_async(flx_ctx *ctx, uint64_t n, const int32_t *d_length, const uint8_t *d_passed, uint64_t *d_out);
