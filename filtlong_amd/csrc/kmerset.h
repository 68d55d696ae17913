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

// ---- pair tables of the wave-level cover kernel (round 3) --------------------------------------------------------------
// A random lookup is priced per distinct cache LINE an instruction touches, not per lane (tools/tabench: 261 G lines/s from
// the L2 however many lanes share a line, 55 G/s beyond it), so both tables answer TWO consecutive positions with one byte:
//
// pre11  — the 12-mer prefilter keyed by the 11-mer C the 12-mers of positions i and i+1 share (x.C ends at i, C.y ends at
//          i+1): bit x of the low nibble = "x.C occurs in a set member", bit 4+y = "C.y occurs".  That is every 12-mer twice
//          (4 MiB); the table is folded onto CANONICAL 11-mers instead — an 11-mer and its reverse complement share a byte,
//          the strand read off the middle base (A/C vs G/T: exactly one of the two strands has it in {A, C}) — which gives
//          2 MiB again, still L2 resident.  Folding ORs the two strands' bits: a superset, and any superset of the present
//          12-mers is a valid filter (no false negatives); the reference inserts both strands of every 16-mer anyway
//          (src/kmers.cpp:109-120), so nothing is lost except around non-ACGT bases.
// exact15 — exact membership keyed by the 15-mer C the 16-mers of positions i and i+1 share: bit x = "x.C is a member",
//          bit 4+y = "C.y is a member" (1 GiB; every member sets two bits).  One 64-byte fabric request answers both.
__host__ __device__ inline uint32_t flx_rc11(uint32_t c) {  // reverse complement of an 11-mer (22 bits, first base on top)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r = __brev(c);
#else
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((c >> i) & 1u) << (31 - i);
#endif
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);  // bits inside every 2-bit group back in order
    return (~r) >> 10;
}
// byte index and bit numbers of the two questions an 11-mer C answers: `even` = bit of x.C, `odd` = bit of C.y
struct flx_pre11_slot { uint32_t index, even_bit, odd_bit; };
__host__ __device__ inline flx_pre11_slot flx_pre11(uint32_t c, uint32_t x, uint32_t y) {
    const uint32_t other = (c >> 11) & 1u;  // middle base (bits 11:10) is G or T: the byte belongs to the other strand
    const uint32_t k = other ? flx_rc11(c) : c;
    flx_pre11_slot s;
    s.index = ((k >> 12) << 11) | (k & 0x7FFu);  // bit 11 of a canonical 11-mer is 0: dropped
    s.even_bit = other ? 7u - x : x;             // RC(x.C) = RC(C).comp(x): a successor of the canonical strand
    s.odd_bit = other ? 3u - y : 4u + y;         // RC(C.y) = comp(y).RC(C): a predecessor
    return s;
}
const uint8_t *flx_kmerset_pre11(const flx_kmerset *set);    // 2 MiB or NULL (no prefilter: saturated, or switched off)
const uint8_t *flx_kmerset_exact15(const flx_kmerset *set);  // 1 GiB

// ---- the assembly as a text: members confirmed along a read's locus (round 4) -------------------------------------------------
// A far request to exact15 answers two positions; 64 bytes of the 2-bit packed assembly answer 256.  When the set was built from
// an assembly (src/kmers.cpp:61-72), finalize keeps the sequences as `text`: for every contig of at least 16 bases its forward
// strand in the forward encoder's codes (src/kmers.cpp:176-196), then its reverse strand in the codes the reverse encoder puts on
// top of its k-mers (src/kmers.cpp:199-219: A 3, C 2, G 1, T 0, anything else 0) in reversed order — so every 16-base window of
// `text` that lies inside one strand copy IS one of the two k-mers src/kmers.cpp:106-121 inserted at that position, non-ACGT quirk
// included.  If 16 consecutive bases of a read carry the codes of such a window, that 16-mer is a member: a locus match is
// SUFFICIENT (not necessary — the 16-mer may occur elsewhere, so windows with a mismatch go through the usual search).
//   text  uint2 per 16 bases: .x = the codes (first base in the top two bits), .y = bit k set when base k of the word is the
//         FIRST base of a strand copy (a window must not run across such a base behind its first); kLocusPad words of padding with
//         every .y bit set in front of the data and at least 68 behind, so that an index clamped to [0, n_alloc) is never a match
//         .y bit 16 + k set when the 13 bases from base k of the word on lie in one strand copy and occur NOWHERE ELSE in the text
//         (U13).  A 16-mer that contains such a 13-mer can only be a member as the window of the text around it — so a window
//         of a read that holds a text-matching unique 13-mer and is not itself a text match is NOT a member, without any lookup:
//         this settles most of the false candidates the 12-mer prefilter lets through next to a mismatch (they share 15, 14 or
//         13 bases with a member, which is exactly why their 12-mers are present)
//   safe1 uint16 per text word (same index as `text`), bit k set when the 16 bases from base k of the word on lie in one strand copy
//         and NONE of the 48 16-mers that differ from them in exactly one base is a member (S1).  A window of a read that differs from
//         the text along the diagonal in exactly one base IS one of those 48: it is not a member, without any lookup — this settles the
//         single-substitution windows whose mismatch sits in the middle (bases 3..12), which U13 cannot reach; at 5 Mbp 89 % of the
//         text's windows are safe.  NULL: not built (FLX_KMER_SAFE1=0, or no memory)
//   seed  open addressing, key = a 16-mer of the text, value = the text position of its first base (the smallest one for a
//         repeated 16-mer), 0xFFFFFFFF = empty; a slot is verified by comparing the text there with the key
constexpr uint32_t kLocusPad = 2;
constexpr uint32_t kLocusEmpty = 0xFFFFFFFFu;
struct flx_locus {
    const uint2 *text;
    uint32_t n_alloc;    // words in `text`, padding included
    uint64_t n_text;     // text positions (bases of both strand copies)
    const uint16_t *safe1;  // n_alloc entries or NULL
    const uint32_t *seed;
    uint32_t seed_mask;  // slots - 1
    int seed_shift;      // 32 - log2(slots)
};
__host__ __device__ inline uint32_t flx_locus_hash(uint32_t kmer, int shift) { return (kmer * 0x9E3779B1u) >> shift; }
#if defined(__HIPCC__)
// the 16 codes of the text from position t on (t + 16 <= n_text is the caller's business: padding follows the data)
__device__ __forceinline__ uint32_t flx_locus_kmer_at(const uint2 *text, uint32_t t) {
    const uint32_t w = (t >> 4) + kLocusPad, s = t & 15u;
    const uint32_t a = text[w].x;
    if (s == 0) return a;
    return __builtin_amdgcn_alignbit(a, text[w + 1].x, 32 - 2 * s);
}
#endif
// pathtext.hip: the same for a set without an assembly (its de Bruijn graph cut into paths)
struct flx_seq_batch {  // sequences on the device as k_add_reference indexes them (those of at least 16 bases)
    const uint8_t *bases;
    const uint64_t *offsets, *pos_base;
    uint64_t n_seqs, n_pos;
};
int flx_build_path_text(flx_ctx *ctx, const uint32_t *present, const uint8_t *exact15, uint64_t n_members, const flx_seq_batch *batches,
                        size_t n_batches, uint32_t **text_out, uint32_t **seed_out, flx_locus *loc);
const flx_locus *flx_kmerset_locus(const flx_kmerset *set);  // NULL: no assembly, too large, or switched off at build time

int flx_score_kmer_dev(flx_ctx *ctx, const flx_kmerset *set, const uint8_t *d_plane, uint64_t plane_bytes,
                       const uint64_t *d_offsets, const int32_t *d_lengths, const uint32_t *d_order,
                       uint64_t n_reads, const flx_params *params, flx_scores *out);
