// stats.hip — a20: statistics of the mean qualities, bit-identical to the reference's serial folds
// (src/main.cpp:170-186):
//     for read in reads2: quality_sum += mean_q; track min/max        (strict reads2 order)
//     mean = quality_sum / N
//     for read in reads2: d = mean_q - mean; stdev_sum += d*d         (strict reads2 order)
//     stdev = sqrt(stdev_sum / N)
// FP64 addition is not associative, so a tree reduction gives a different last bit and
// through the z-scores a different normalised quality for every read.
#include <cmath>

#include "flx_internal.h"
#include "rank_internal.h"

int flx_exact_stats(flx_ctx *ctx, uint64_t n, const double *d_mean_q, flx_stats *out) {
    std::vector<double> h(n);
    if (n) {
        FLX_HIP(ctx, hipMemcpyAsync(h.data(), d_mean_q, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    double qmin = 100.0, qmax = 0.0, qsum = 0.0;  // main.cpp:170-172
    for (uint64_t i = 0; i < n; ++i) {
        const double v = h[i];
        qsum += v;
        if (v > qmax) qmax = v;
        if (v < qmin) qmin = v;
    }
    const double qmean = qsum / (double)n;
    double ssum = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        const double d = h[i] - qmean;
        ssum += d * d;
    }
    out->min = qmin;
    out->max = qmax;
    out->sum = qsum;
    out->mean = qmean;
    out->sq_sum = ssum;
    out->stdev = sqrt(ssum / (double)n);
    return FLX_OK;
}
