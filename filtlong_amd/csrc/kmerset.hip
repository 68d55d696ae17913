// kmerset.hip — seam 1 (reference 16-mer set) and k-mer scoring; placeholder until the k-mer path lands.
#include "flx_internal.h"
#include "kmerset.h"

struct flx_kmerset {
    flx_ctx *ctx;
    bool final_;
    uint64_t size;
};

bool flx_kmerset_is_final(const flx_kmerset *set) { return set->final_; }
const uint32_t *flx_kmerset_bitmap(const flx_kmerset *) { return nullptr; }

extern "C" int flx_kmerset_create(flx_ctx *ctx, flx_kmerset **out) {
    if (!ctx || !out) return FLX_ERR_INVALID;
    *out = new flx_kmerset{ctx, false, 0};
    return FLX_OK;
}
extern "C" void flx_kmerset_destroy(flx_kmerset *set) { delete set; }
extern "C" int flx_kmerset_add_assembly(flx_kmerset *set, const uint8_t *, const uint64_t *, const int64_t *, uint64_t) {
    return flx_fail(set->ctx, FLX_ERR_STATE, "k-mer set build not implemented yet");
}
extern "C" int flx_kmerset_add_short_reads(flx_kmerset *set, const uint8_t *, const uint64_t *, const int64_t *, uint64_t) {
    return flx_fail(set->ctx, FLX_ERR_STATE, "k-mer set build not implemented yet");
}
extern "C" int flx_kmerset_finalize(flx_kmerset *set) { set->final_ = true; return FLX_OK; }
extern "C" uint64_t flx_kmerset_size(const flx_kmerset *set) { return set ? set->size : 0; }
extern "C" int flx_kmerset_contains(const flx_kmerset *set, const uint32_t *, uint64_t, uint8_t *) {
    return flx_fail(set->ctx, FLX_ERR_STATE, "k-mer set build not implemented yet");
}
int flx_score_kmer_dev(flx_ctx *ctx, const flx_kmerset *, const uint8_t *, uint64_t, const uint64_t *, const int32_t *,
                       const uint32_t *, uint64_t, const flx_params *, flx_scores *) {
    return flx_fail(ctx, FLX_ERR_STATE, "k-mer scoring not implemented yet");
}
