// kmerset.h — internal interface of the reference 16-mer set (seam 1) and the k-mer scoring path.
#pragma once
#include "flx_internal.h"

bool flx_kmerset_is_final(const flx_kmerset *set);
// exact membership bitmap over all 4^16 16-mers (2^32 bits = 512 MiB), device resident
const uint32_t *flx_kmerset_bitmap(const flx_kmerset *set);

// L2-resident prefilter in front of the bitmap: 2^24 bits (2 MiB), bit flx_prefilter_hash(k) is set for every 16-mer k
// of the set (no false negatives, so "bit clear" answers a query without the 64-byte fabric request a bitmap lookup
// costs).  NULL when the set is so large that the filter would be nearly full.
constexpr int kPrefilterBits = 24;
__host__ __device__ inline uint32_t flx_prefilter_hash(uint32_t kmer) { return (kmer * 0x9E3779B1u) >> (32 - kPrefilterBits); }
const uint32_t *flx_kmerset_prefilter(const flx_kmerset *set);

int flx_score_kmer_dev(flx_ctx *ctx, const flx_kmerset *set, const uint8_t *d_plane, uint64_t plane_bytes,
                       const uint64_t *d_offsets, const int32_t *d_lengths, const uint32_t *d_order,
                       uint64_t n_reads, const flx_params *params, flx_scores *out);
