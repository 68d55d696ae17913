// kmerset.h — internal interface of the reference 16-mer set (seam 1) and the k-mer scoring path.
#pragma once
#include "flx_internal.h"

bool flx_kmerset_is_final(const flx_kmerset *set);
// exact membership bitmap over all 4^16 16-mers (2^32 bits = 512 MiB), device resident
const uint32_t *flx_kmerset_bitmap(const flx_kmerset *set);

// L2-resident prefilter in front of the bitmap: the presence bitmap of all 12-MERS that occur inside a 16-mer of the set
// (4^12 bits = 2 MiB, indexed by the 12-mer's 24-bit value, no hashing).  A 16-mer holds five 12-mers (offsets 0..4); it can
// only be a member if all five are present, so "one of the five bits clear" answers a query without the 64-byte fabric
// request a bitmap lookup costs.  Along a read the 12-mers roll with the 16-mers — ONE L2 lookup per position serves the five
// 16-mers that contain it — and five independent bits leave ~0.45^5 = 2 % false positives for a 5 Mbp genome (a hashed
// one-bit-per-16-mer filter of the same size: 45 %).  No false negatives by construction.  NULL when switched off.
constexpr int kPrefilterBits = 24;
__host__ __device__ inline uint32_t flx_sub12(uint32_t kmer16, int d) { return (kmer16 >> (2 * d)) & 0xFFFFFFu; }  // d = 0: last 12 bases
const uint32_t *flx_kmerset_prefilter(const flx_kmerset *set);

int flx_score_kmer_dev(flx_ctx *ctx, const flx_kmerset *set, const uint8_t *d_plane, uint64_t plane_bytes,
                       const uint64_t *d_offsets, const int32_t *d_lengths, const uint32_t *d_order,
                       uint64_t n_reads, const flx_params *params, flx_scores *out);
