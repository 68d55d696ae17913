// kmerset.hip — seam 1: the reference 16-mer set on the device.
//
// Replaces the reference's Kmers (src/kmers.cpp:28-172): an std::unordered_set<uint32_t> built from an
// assembly (every 16-mer, both strands, src/kmers.cpp:61-72,137-139) and/or from short reads (a 16-mer
// enters once it has been seen 4 times — or 3 times if the Bloom filter gave a false positive at its first
// sighting, src/kmers.cpp:142-166), queried by exact lookup (is_kmer_present, src/kmers.cpp:170-172).
//
// Device representation: there are only 4^16 = 2^32 possible 16-mers, so an EXACT set is a 512 MiB bitmap
// (one random 4-byte access per query, no probing, no false positives).  The short-read
// multiplicity rule is evaluated with three more bitmaps acting as a saturating counter: each occurrence of
// a 16-mer sets the lowest clear level among seen1, seen2, seen3, present (an atomicOr cascade), so after c
// occurrences exactly min(c,4) levels are set regardless of the order in which the GPU processes them.
//
// The Bloom filter (src/bloom_filter.h, restated bit-exactly: 13 hashes, 1 917 295 480 bits, salts and
// hash_ap's 4-byte branch) only matters for 16-mers seen exactly 3 times whose 13 Bloom bits were all set
// by OTHER 16-mers before their first sighting.  finalize() screens for that with a conservative superset
// test (all 13 bits set by at least two insertions in the final filter) and, for the candidates (expected:
// none — at Filtlong's filter size the probability is ~1e-15 per 16-mer), replays first-sighting times to
// decide exactly (resolve_bloom_candidates).
#include <algorithm>

#include "flx_internal.h"
#include "kmerset.h"

namespace {

constexpr uint64_t kBitmapWords = 1ull << 27;  // 2^32 bits
constexpr uint32_t kBloomHashes = 13;          // bloom_parameters::compute_optimal_parameters (n=1e8, p=1e-4)
constexpr uint64_t kBloomBits = 1917295480ull;
constexpr uint64_t kBloomWords = (kBloomBits + 31) / 32;

// generate_unique_salt (bloom_filter.h:467-529) with seed 0xA5A5A5A5 (kmers.cpp:34): see oracle/flx_oracle.cpp
__constant__ uint32_t c_salts[13] = {0x1B5793D2u, 0x81BDFA38u, 0xEB8E30D5u, 0x45B52496u, 0x85C1FE3Cu, 0x3DACB627u,
                                     0x78776869u, 0x94A40D1Eu, 0x5F9BB638u, 0x40FB59D5u, 0x8174BDB2u, 0x0B466EAAu,
                                     0x209D29A7u};

__device__ __forceinline__ uint32_t bloom_index(uint32_t kmer, int i) {
    uint32_t h = c_salts[i];
    h ^= ~((h << 11) + (kmer ^ (h >> 5)));  // hash_ap, 4-byte key (bloom_filter.h:569-583)
    return h % (uint32_t)kBloomBits;        // compute_indices (bloom_filter.h:461-465)
}

// src/kmers.cpp:176-196 / 199-219: anything that is not ACGTacgt encodes as 0 on BOTH strands
__device__ __forceinline__ uint32_t base_fwd(uint8_t c) {
    switch (c) {
        case 'C': case 'c': return 1u;
        case 'G': case 'g': return 2u;
        case 'T': case 't': return 3u;
        default: return 0u;
    }
}
__device__ __forceinline__ uint32_t base_rev_code(uint8_t c) {  // the 2-bit value placed in the top bits
    switch (c) {
        case 'G': case 'g': return 1u;
        case 'C': case 'c': return 2u;
        case 'A': case 'a': return 3u;
        default: return 0u;
    }
}

__device__ __forceinline__ bool test_bit(const uint32_t *bm, uint32_t k) { return (bm[k >> 5] >> (k & 31)) & 1u; }
__device__ __forceinline__ bool set_bit(uint32_t *bm, uint32_t k) {  // returns the previous value
    const uint32_t m = 1u << (k & 31);
    return (atomicOr(&bm[k >> 5], m) & m) != 0;
}

// One thread per 16-mer START position of the packed reference sequences.  `pos_base[s]` is the number of
// start positions of sequences before s (exclusive scan of max(len-15,0)), so the grid is flat.
template <bool MULTI>
__global__ void __launch_bounds__(256) k_add_reference(const uint8_t *bases, const uint64_t *offsets,
                                                       const int64_t *lengths, const uint64_t *pos_base,
                                                       uint64_t n_seqs, uint64_t n_pos, uint32_t *present,
                                                       uint32_t *seen1, uint32_t *seen2, uint32_t *seen3) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    // find the sequence: largest s with pos_base[s] <= g
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t p = g - pos_base[lo];  // start position inside the sequence
    const uint8_t *s = bases + offsets[lo] + p;
    uint32_t f = 0, r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // starting_kmer_to_bits_forward / _reverse, kmers.cpp:222-239
        const uint8_t c = s[i];
        f = (f << 2) | base_fwd(c);
        r = (r >> 2) | (base_rev_code(c) << 30);
    }
    (void)lengths;
    const uint32_t two[2] = {f, r};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t k = two[q];
        if (!MULTI) {
            if (!test_bit(present, k)) set_bit(present, k);  // add_kmer_require_one_copy
        } else {
            // add_kmer_require_multiple_copies: already in the set (assembly, or promoted) -> nothing to do;
            // otherwise raise the saturating count by one level.
            if (test_bit(present, k)) continue;
            if (!set_bit(seen1, k)) continue;
            if (!set_bit(seen2, k)) continue;
            if (!set_bit(seen3, k)) continue;
            set_bit(present, k);
        }
    }
}

// every member of the exact bitmap marks its five 12-mers in the prefilter
__global__ void __launch_bounds__(256) k_build_prefilter(const uint32_t *bm, uint64_t n_words, uint32_t *pre) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = bm[i];
        while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            const uint32_t k = (uint32_t)(i << 5) | (uint32_t)b;
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                const uint32_t h = flx_sub12(k, d);
                atomicOr(&pre[h >> 5], 1u << (h & 31));
            }
        }
    }
}

// pre11 from the 12-mer presence bits: every present 12-mer Y = x.C = C'.y marks the byte of C (as predecessor x) and of C'
// (as successor y) — kmerset.h
__global__ void __launch_bounds__(256) k_build_pre11(const uint32_t *pre12, uint32_t *pre11_words) {
    const uint32_t n_words = 1u << (kPrefilterBits - 5);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) {
        uint32_t w = pre12[i];
        while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            const uint32_t y12 = (i << 5) | (uint32_t)b;
            const flx_pre11_slot a = flx_pre11(y12 & 0x3FFFFFu, y12 >> 22, 0);  // x.C
            atomicOr(&pre11_words[a.index >> 2], (1u << a.even_bit) << ((a.index & 3u) * 8));
            const flx_pre11_slot c = flx_pre11(y12 >> 2, 0, y12 & 3u);          // C'.y
            atomicOr(&pre11_words[c.index >> 2], (1u << c.odd_bit) << ((c.index & 3u) * 8));
        }
    }
}

// exact15 from the membership bitmap: member K = x.C15 = C15'.y sets bit x of byte C15 and bit 4+y of byte C15'
__global__ void __launch_bounds__(256) k_build_exact15(const uint32_t *bm, uint64_t n_words, uint32_t *ex_words) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = bm[i];
        while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            const uint32_t k = (uint32_t)(i << 5) | (uint32_t)b;
            const uint32_t c = k & 0x3FFFFFFFu, x = k >> 30;
            atomicOr(&ex_words[c >> 2], (1u << x) << ((c & 3u) * 8));
            const uint32_t c2 = k >> 2, y = k & 3u;
            atomicOr(&ex_words[c2 >> 2], (16u << y) << ((c2 & 3u) * 8));
        }
    }
}

__global__ void __launch_bounds__(256) k_popcount(const uint32_t *bm, uint64_t n_words, unsigned long long *out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        acc += __popc(bm[i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// Final Bloom filter with multiplicity: F1 = bit set by >= 1 distinct inserted 16-mer, F2 = by >= 2 insertions.
// Every distinct short-read 16-mer that was not already in the assembly set is assumed to insert (a superset of
// the truth: false-positive 16-mers do not insert), which makes the candidate test below conservative.
__global__ void __launch_bounds__(256) k_bloom_fill(const uint32_t *seen1, uint64_t n_words, uint32_t *F1, uint32_t *F2) {
    for (uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = seen1[wi];
        while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            const uint32_t k = (uint32_t)(wi << 5) | (uint32_t)b;
            for (int i = 0; i < (int)kBloomHashes; ++i) {
                const uint32_t idx = bloom_index(k, i);
                if (set_bit(F1, idx)) set_bit(F2, idx);
            }
        }
    }
}

// Candidates for "Bloom false positive at first sighting": seen exactly 3 times (otherwise the outcome does not
// depend on the filter) and all 13 bits set at least twice in the final filter.
__global__ void __launch_bounds__(256) k_bloom_candidates(const uint32_t *seen1, const uint32_t *seen3,
                                                          const uint32_t *present, uint64_t n_words, const uint32_t *F2,
                                                          uint32_t *cand, unsigned int *n_cand, unsigned int cap,
                                                          int only_count3) {
    for (uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = only_count3 ? (seen3[wi] & ~present[wi]) : seen1[wi];
        while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            const uint32_t k = (uint32_t)(wi << 5) | (uint32_t)b;
            bool all = true;
            for (int i = 0; i < (int)kBloomHashes && all; ++i) all = test_bit(F2, bloom_index(k, i));
            if (all) {
                const unsigned int at = atomicAdd(n_cand, 1u);
                if (at < cap) cand[at] = k;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_contains(const uint32_t *present, const uint32_t *kmers, uint64_t n, uint8_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = test_bit(present, kmers[i]) ? 1 : 0;
}

__global__ void k_set_word_bits(uint32_t *word, uint32_t bits) { *word |= bits; }

__global__ void __launch_bounds__(256) k_set_bits(uint32_t *bm, const uint32_t *kmers, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) set_bit(bm, kmers[i]);
}

// First-sighting times for the exact Bloom replay (only when candidates exist).  Occurrence index of the 16-mer
// starting at position p of sequence s: tau = 2 * (global start position) + strand, which is the order in which
// the reference's loop visits them (forward then reverse at every position, sequences in file order, -1 before -2;
// src/kmers.cpp:106-121).  watch_keys is a small sorted list of 16-mers (candidates) or Bloom bit indices.
__device__ __forceinline__ int find_sorted(const uint32_t *a, int n, uint32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && a[lo] == key) ? lo : -1;
}

__global__ void __launch_bounds__(256) k_first_sightings(const uint8_t *bases, const uint64_t *offsets,
                                                         const uint64_t *pos_base, uint64_t n_seqs, uint64_t n_pos,
                                                         uint64_t tau_base, const uint32_t *asm_present,
                                                         const uint32_t *cand_sorted, int n_cand,
                                                         const uint32_t *bits_sorted, int n_bits,
                                                         unsigned long long *cand_tau, unsigned long long *bit_tau_noncand) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint8_t *s = bases + offsets[lo] + (g - pos_base[lo]);
    uint32_t f = 0, r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint8_t c = s[i];
        f = (f << 2) | base_fwd(c);
        r = (r >> 2) | (base_rev_code(c) << 30);
    }
    const uint32_t two[2] = {f, r};
    for (int q = 0; q < 2; ++q) {
        const uint32_t k = two[q];
        if (test_bit(asm_present, k)) continue;  // never reaches the Bloom filter (kmers.cpp:144-145)
        const unsigned long long tau = tau_base + 2ull * g + (unsigned long long)q;
        const int ci = find_sorted(cand_sorted, n_cand, k);
        if (ci >= 0) {
            atomicMin(&cand_tau[ci], tau);
        } else {  // definitely inserts at its first sighting: earliest time it sets each watched bit
            for (int i = 0; i < (int)kBloomHashes; ++i) {
                const int bi = find_sorted(bits_sorted, n_bits, bloom_index(k, i));
                if (bi >= 0) atomicMin(&bit_tau_noncand[bi], tau);
            }
        }
    }
}

// ---- the assembly as a text + seed table (kmerset.h: flx_locus) ----------------------------------------------------------
// one thread per base of the batch's sequences (those of at least 16 bases; cum[i] = bases of the sequences before i)
__global__ void __launch_bounds__(256) k_locus_text(const uint8_t *bases, const uint64_t *offsets, const uint64_t *pos_base,
                                                    uint64_t n_seqs, uint64_t n_pos, uint64_t text_base, uint32_t *text_words) {
    const uint64_t n_bases = n_pos + 15 * n_seqs;
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_bases) return;
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] + 15 * mid <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t cum = pos_base[lo] + 15 * lo;
    const uint64_t len = (lo + 1 < n_seqs ? pos_base[lo + 1] : n_pos) - pos_base[lo] + 15;
    const uint64_t o = g - cum;
    const uint8_t c = bases[offsets[lo] + o];
    const uint64_t tf = text_base + 2 * cum + o, tr = text_base + 2 * cum + len + (len - 1 - o);
    auto put = [&](uint64_t t, uint32_t code, bool starts_copy) {
        uint32_t *w = text_words + 2 * ((t >> 4) + kLocusPad);
        if (code) atomicOr(w, code << (30 - 2 * (uint32_t)(t & 15)));
        if (starts_copy) atomicOr(w + 1, 1u << (uint32_t)(t & 15));
    };
    put(tf, base_fwd(c), o == 0);
    put(tr, base_rev_code(c), o == len - 1);
}

// U13 (kmerset.h), two passes over every 13-base window inside a strand copy: count its value (saturating at 2: two bitmaps of
// 4^13 bits), then mark the windows whose value was seen once
template <int PASS>
__global__ void __launch_bounds__(256) k_locus_u13(const uint64_t *pos_base, uint64_t n_seqs, uint64_t n_pos, uint64_t text_base,
                                                   uint32_t *text_words, uint32_t *seen1, uint32_t *seen2) {
    const uint64_t n_win = n_pos + 3 * n_seqs;  // 13-windows per strand copy: len - 12 = (len - 15) + 3
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_win) return;
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] + 3 * mid <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t p = g - (pos_base[lo] + 3 * lo);
    const uint64_t cum = pos_base[lo] + 15 * lo;
    const uint64_t len = (lo + 1 < n_seqs ? pos_base[lo + 1] : n_pos) - pos_base[lo] + 15;
    for (int strand = 0; strand < 2; ++strand) {
        const uint64_t t = text_base + 2 * cum + (strand ? len : 0) + p;
        const uint32_t v = flx_locus_kmer_at((const uint2 *)text_words, (uint32_t)t) >> 6;  // the first 13 of the 16 codes from t on
        if (PASS == 0) {
            if (set_bit(seen1, v)) set_bit(seen2, v);
        } else if (!test_bit(seen2, v)) {
            atomicOr(text_words + 2 * ((t >> 4) + kLocusPad) + 1, 0x10000u << (uint32_t)(t & 15));
        }
    }
}

// S1 (kmerset.h), one thread per text position: the 16-window from there on, if it lies in one piece of the text, against the exact
// bitmap with every one of its bases replaced by the three others (48 far lookups; 10^7 windows: ~10 ms)
__global__ void __launch_bounds__(256) k_text_safe1(const uint2 *text, uint64_t n_text, const uint32_t *present, uint32_t *safe_words) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t + 16 > n_text) return;
    {  // no piece starts at t + 1 .. t + 15
        const uint64_t t1 = t + 1;
        const uint64_t w = (t1 >> 4) + kLocusPad;
        const uint32_t s = (uint32_t)(t1 & 15);
        const uint32_t b = ((text[w].y & 0xffffu) >> s) | ((text[w + 1].y & 0xffffu) << (16 - s));
        if (b & 0x7fffu) return;
    }
    const uint32_t k = flx_locus_kmer_at(text, (uint32_t)t);
    uint32_t any = 0;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        uint32_t w[3];
#pragma unroll
        for (uint32_t x = 1; x < 4; ++x) {
            const uint32_t nb = k ^ (x << (2 * j));
            w[x - 1] = (present[nb >> 5] >> (nb & 31u)) & 1u;
        }
        any |= w[0] | w[1] | w[2];
    }
    if (!any) {
        const uint64_t wd = (t >> 4) + kLocusPad;  // (uint16 entry wd = half (wd & 1) of 32-bit word wd >> 1)
        atomicOr(safe_words + (wd >> 1), (1u << (uint32_t)(t & 15)) << (16 * (uint32_t)(wd & 1)));
    }
}

// one thread per 16-mer start of the batch's sequences, both strand copies: the smallest text position of every distinct 16-mer
__global__ void __launch_bounds__(256) k_locus_seed(const uint64_t *pos_base, uint64_t n_seqs, uint64_t n_pos, uint64_t text_base,
                                                    const uint2 *text, uint32_t *seed, uint32_t mask, int shift) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t p = g - pos_base[lo];
    const uint64_t cum = pos_base[lo] + 15 * lo;
    const uint64_t len = (lo + 1 < n_seqs ? pos_base[lo + 1] : n_pos) - pos_base[lo] + 15;
    for (int strand = 0; strand < 2; ++strand) {
        const uint32_t t = (uint32_t)(text_base + 2 * cum + (strand ? len : 0) + p);
        const uint32_t k = flx_locus_kmer_at(text, t);
        uint32_t h = flx_locus_hash(k, shift);
        for (;;) {
            const uint32_t old = atomicCAS(&seed[h], kLocusEmpty, t);
            if (old == kLocusEmpty) break;
            if (flx_locus_kmer_at(text, old) == k) {  // (whoever holds the slot, its text is final: the slot's key cannot change)
                atomicMin(&seed[h], t);
                break;
            }
            h = (h + 1) & mask;
        }
    }
}

}  // namespace

struct flx_kmerset {
    flx_ctx *ctx = nullptr;
    bool final_ = false;
    bool has_short = false;
    uint64_t size = 0;
    uint32_t *present = nullptr;        // 512 MiB
    uint32_t *asm_only = nullptr;       // copy of `present` taken when the first short reads arrive (512 MiB)
    uint32_t *seen1 = nullptr, *seen2 = nullptr, *seen3 = nullptr;
    uint32_t *prefilter = nullptr;      // 2 MiB, built at finalize (kmerset.h)
    uint32_t *pre11 = nullptr;          // 2 MiB: the same filter, two positions per byte (kmerset.h)
    uint32_t *exact15 = nullptr;        // 1 GiB: exact membership, two positions per byte
    // short-read sequences kept on the device until finalize (only replayed if Bloom candidates exist)
    struct Batch {
        uint8_t *bases;
        uint64_t *offsets;
        uint64_t *pos_base;
        uint64_t n_seqs, n_pos;
    };
    std::vector<Batch> short_batches;
    std::vector<Batch> asm_batches;     // the assembly's sequences, kept until finalize builds the locus text from them (kmerset.h)
    uint32_t *locus_text = nullptr, *locus_seed = nullptr, *locus_safe1 = nullptr;
    flx_locus locus;
    bool has_locus = false;
    uint64_t bloom_candidates = 0;
    uint64_t bloom_false_positives = 0;
};

bool flx_kmerset_is_final(const flx_kmerset *set) { return set->final_; }
const uint32_t *flx_kmerset_bitmap(const flx_kmerset *set) { return set->present; }
const uint32_t *flx_kmerset_prefilter(const flx_kmerset *set) { return set->prefilter; }
const uint8_t *flx_kmerset_pre11(const flx_kmerset *set) { return (const uint8_t *)set->pre11; }
const uint8_t *flx_kmerset_exact15(const flx_kmerset *set) { return (const uint8_t *)set->exact15; }
const flx_locus *flx_kmerset_locus(const flx_kmerset *set) { return set->has_locus ? &set->locus : nullptr; }

extern "C" int flx_kmerset_create(flx_ctx *ctx, flx_kmerset **out) {
    if (!ctx || !out) return FLX_ERR_INVALID;
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    flx_kmerset *s = new flx_kmerset();
    s->ctx = ctx;
    hipError_t e = hipMalloc((void **)&s->present, kBitmapWords * 4);
    if (e != hipSuccess) {
        delete s;
        return flx_fail(ctx, FLX_ERR_NOMEM, "k-mer bitmap (512 MiB): %s", hipGetErrorString(e));
    }
    FLX_HIP(ctx, hipMemsetAsync(s->present, 0, kBitmapWords * 4, ctx->stream));
    *out = s;
    return FLX_OK;
}

static void free_batches(std::vector<flx_kmerset::Batch> &batches) {
    for (auto &b : batches) {
        (void)hipFree(b.bases);
        (void)hipFree(b.offsets);
        (void)hipFree(b.pos_base);
    }
    batches.clear();
}
static void free_batches(flx_kmerset *s) { free_batches(s->short_batches); }

extern "C" void flx_kmerset_destroy(flx_kmerset *s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    free_batches(s);
    free_batches(s->asm_batches);
    for (uint32_t *p : {s->present, s->asm_only, s->seen1, s->seen2, s->seen3, s->prefilter, s->pre11, s->exact15, s->locus_text, s->locus_seed, s->locus_safe1})
        if (p) (void)hipFree(p);
    delete s;
}

// uploads the packed sequences and the flat position index; returns device pointers in `b`
static int upload_sequences(flx_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const int64_t *lengths,
                            uint64_t n_seqs, flx_kmerset::Batch &b) {
    std::vector<uint64_t> pos_base(n_seqs + 1);
    uint64_t np = 0, total = 0;
    for (uint64_t i = 0; i < n_seqs; ++i) {
        if (lengths[i] < 0) return flx_fail(ctx, FLX_ERR_INVALID, "negative sequence length");
        pos_base[i] = np;
        if (lengths[i] >= 16) np += (uint64_t)lengths[i] - 15;  // sequences shorter than 16 give no 16-mers (kmers.cpp:99-100)
        total = std::max<uint64_t>(total, offsets[i] + (uint64_t)lengths[i]);
    }
    pos_base[n_seqs] = np;
    // pos_base must be strictly usable by the binary search: sequences without positions share their successor's base,
    // and "largest s with pos_base[s] <= g" then picks the LAST such s, which is the one that owns position g only if
    // it has positions.  Compact the index to sequences that have at least one position.
    std::vector<uint64_t> c_off, c_base;
    for (uint64_t i = 0; i < n_seqs; ++i)
        if (lengths[i] >= 16) {
            c_off.push_back(offsets[i]);
            c_base.push_back(pos_base[i]);
        }
    b.n_seqs = c_off.size();
    b.n_pos = np;
    b.bases = nullptr;
    b.offsets = nullptr;
    b.pos_base = nullptr;
    if (np == 0) return FLX_OK;
    FLX_HIP(ctx, hipMalloc((void **)&b.bases, total + 16));
    FLX_HIP(ctx, hipMalloc((void **)&b.offsets, b.n_seqs * 8));
    FLX_HIP(ctx, hipMalloc((void **)&b.pos_base, b.n_seqs * 8));
    FLX_HIP(ctx, hipMemcpyAsync(b.bases, bases, total, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(b.offsets, c_off.data(), b.n_seqs * 8, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(b.pos_base, c_base.data(), b.n_seqs * 8, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
    return FLX_OK;
}

extern "C" int flx_kmerset_add_assembly(flx_kmerset *s, const uint8_t *bases, const uint64_t *offsets,
                                        const int64_t *lengths, uint64_t n_seqs) {
    if (!s) return FLX_ERR_INVALID;
    flx_ctx *ctx = s->ctx;
    if (s->final_) return flx_fail(ctx, FLX_ERR_STATE, "k-mer set is already finalized");
    if (s->has_short)
        return flx_fail(ctx, FLX_ERR_STATE, "assembly must be added before short reads (the reference hashes the "
                                            "assembly first, src/main.cpp:55-58, and the multi-copy rule depends on it)");
    if (n_seqs == 0) return FLX_OK;
    if (!bases || !offsets || !lengths) return flx_fail(ctx, FLX_ERR_INVALID, "NULL sequence arrays");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    flx_kmerset::Batch b;
    FLX_CHECK(upload_sequences(ctx, bases, offsets, lengths, n_seqs, b));
    if (b.n_pos) {
        flx_time_begin(ctx, "flx_kmerset_add_assembly");
        hipLaunchKernelGGL(k_add_reference<false>, dim3((unsigned)((b.n_pos + 255) / 256)), dim3(256), 0, ctx->stream,
                           b.bases, b.offsets, (const int64_t *)nullptr, b.pos_base, b.n_seqs, b.n_pos, s->present,
                           (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
        flx_time_end(ctx);
        FLX_HIP(ctx, hipGetLastError());
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        s->asm_batches.push_back(b);  // finalize builds the locus text from them (kmerset.h), then they go
    }
    return FLX_OK;
}

extern "C" int flx_kmerset_add_short_reads(flx_kmerset *s, const uint8_t *bases, const uint64_t *offsets,
                                           const int64_t *lengths, uint64_t n_seqs) {
    if (!s) return FLX_ERR_INVALID;
    flx_ctx *ctx = s->ctx;
    if (s->final_) return flx_fail(ctx, FLX_ERR_STATE, "k-mer set is already finalized");
    if (n_seqs == 0) return FLX_OK;
    if (!bases || !offsets || !lengths) return flx_fail(ctx, FLX_ERR_INVALID, "NULL sequence arrays");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    if (!s->has_short) {
        for (uint32_t **p : {&s->seen1, &s->seen2, &s->seen3, &s->asm_only}) {
            hipError_t e = hipMalloc((void **)p, kBitmapWords * 4);
            if (e != hipSuccess) return flx_fail(ctx, FLX_ERR_NOMEM, "k-mer count bitmap (512 MiB): %s", hipGetErrorString(e));
        }
        for (uint32_t *p : {s->seen1, s->seen2, s->seen3}) FLX_HIP(ctx, hipMemsetAsync(p, 0, kBitmapWords * 4, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(s->asm_only, s->present, kBitmapWords * 4, hipMemcpyDeviceToDevice, ctx->stream));
        s->has_short = true;
    }
    flx_kmerset::Batch b;
    FLX_CHECK(upload_sequences(ctx, bases, offsets, lengths, n_seqs, b));
    if (b.n_pos) {
        flx_time_begin(ctx, "flx_kmerset_add_short_reads");
        hipLaunchKernelGGL(k_add_reference<true>, dim3((unsigned)((b.n_pos + 255) / 256)), dim3(256), 0, ctx->stream,
                           b.bases, b.offsets, (const int64_t *)nullptr, b.pos_base, b.n_seqs, b.n_pos, s->present, s->seen1,
                           s->seen2, s->seen3);
        flx_time_end(ctx);
        FLX_HIP(ctx, hipGetLastError());
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        s->short_batches.push_back(b);
    }
    return FLX_OK;
}

// Exact replay of the Bloom filter for the candidate 16-mers (see the file header).  A candidate k with first
// sighting tau(k) is a false positive iff each of its 13 bits was set strictly before tau(k) by a 16-mer that
// really inserted.  Non-candidates always insert (they are not false positives even against the FINAL filter);
// candidates are resolved in tau order, each resolved non-false-positive inserting at its tau.
static int resolve_bloom_candidates(flx_kmerset *s, std::vector<uint32_t> &cand /* sorted */,
                                    std::vector<uint8_t> &is_fp) {
    flx_ctx *ctx = s->ctx;
    const int nc = (int)cand.size();
    // watched Bloom bits
    std::vector<uint32_t> salts = {0x1B5793D2u, 0x81BDFA38u, 0xEB8E30D5u, 0x45B52496u, 0x85C1FE3Cu, 0x3DACB627u, 0x78776869u,
                                   0x94A40D1Eu, 0x5F9BB638u, 0x40FB59D5u, 0x8174BDB2u, 0x0B466EAAu, 0x209D29A7u};
    auto bidx = [&](uint32_t k, int i) {
        uint32_t h = salts[i];
        h ^= ~((h << 11) + (k ^ (h >> 5)));
        return h % (uint32_t)kBloomBits;
    };
    std::vector<uint32_t> bits;
    for (uint32_t k : cand)
        for (int i = 0; i < 13; ++i) bits.push_back(bidx(k, i));
    std::sort(bits.begin(), bits.end());
    bits.erase(std::unique(bits.begin(), bits.end()), bits.end());
    const int nb = (int)bits.size();
    flx_dbuf d_cand, d_bits, d_ctau, d_btau;
    FLX_CHECK(flx_dalloc(ctx, d_cand, nc * 4));
    FLX_CHECK(flx_dalloc(ctx, d_bits, nb * 4));
    FLX_CHECK(flx_dalloc(ctx, d_ctau, nc * 8));
    FLX_CHECK(flx_dalloc(ctx, d_btau, nb * 8));
    FLX_HIP(ctx, hipMemcpy(d_cand.p, cand.data(), nc * 4, hipMemcpyHostToDevice));
    FLX_HIP(ctx, hipMemcpy(d_bits.p, bits.data(), nb * 4, hipMemcpyHostToDevice));
    FLX_HIP(ctx, hipMemset(d_ctau.p, 0xff, nc * 8));
    FLX_HIP(ctx, hipMemset(d_btau.p, 0xff, nb * 8));
    uint64_t tau_base = 0;
    for (auto &b : s->short_batches) {
        if (b.n_pos)
            hipLaunchKernelGGL(k_first_sightings, dim3((unsigned)((b.n_pos + 255) / 256)), dim3(256), 0, ctx->stream, b.bases,
                               b.offsets, b.pos_base, b.n_seqs, b.n_pos, tau_base, s->asm_only, d_cand.as<uint32_t>(), nc,
                               d_bits.as<uint32_t>(), nb, d_ctau.as<unsigned long long>(), d_btau.as<unsigned long long>());
        tau_base += 2 * b.n_pos;
    }
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> ctau(nc), btau(nb);
    FLX_HIP(ctx, hipMemcpy(ctau.data(), d_ctau.p, nc * 8, hipMemcpyDeviceToHost));
    FLX_HIP(ctx, hipMemcpy(btau.data(), d_btau.p, nb * 8, hipMemcpyDeviceToHost));
    std::vector<int> ord(nc);
    for (int i = 0; i < nc; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return ctau[a] < ctau[b]; });
    is_fp.assign(nc, 0);
    for (int oi = 0; oi < nc; ++oi) {
        const int c = ord[oi];
        bool fp = true;
        int bi[13];
        for (int i = 0; i < 13; ++i) {
            bi[i] = (int)(std::lower_bound(bits.begin(), bits.end(), bidx(cand[c], i)) - bits.begin());
            if (!(btau[bi[i]] < ctau[c])) fp = false;
        }
        is_fp[c] = fp ? 1 : 0;
        if (!fp)  // it inserts at its first sighting
            for (int i = 0; i < 13; ++i) btau[bi[i]] = std::min(btau[bi[i]], ctau[c]);
    }
    return FLX_OK;
}

extern "C" int flx_kmerset_finalize(flx_kmerset *s) {
    if (!s) return FLX_ERR_INVALID;
    flx_ctx *ctx = s->ctx;
    if (s->final_) return FLX_OK;
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (s->has_short) {
        // Bloom screening (see file header)
        flx_dbuf F1, F2, d_cand, d_n;
        FLX_CHECK(flx_dalloc(ctx, F1, kBloomWords * 4));
        FLX_CHECK(flx_dalloc(ctx, F2, kBloomWords * 4));
        const unsigned cap = 1u << 20;
        FLX_CHECK(flx_dalloc(ctx, d_cand, (size_t)cap * 4));
        FLX_CHECK(flx_dalloc(ctx, d_n, 16));
        FLX_HIP(ctx, hipMemsetAsync(F1.p, 0, kBloomWords * 4, st));
        FLX_HIP(ctx, hipMemsetAsync(F2.p, 0, kBloomWords * 4, st));
        FLX_HIP(ctx, hipMemsetAsync(d_n.p, 0, 16, st));
        flx_time_begin(ctx, "flx_kmerset_bloom_screen");
        hipLaunchKernelGGL(k_bloom_fill, dim3(8192), dim3(256), 0, st, s->seen1, kBitmapWords, F1.as<uint32_t>(), F2.as<uint32_t>());
        // 1st: is any 16-mer with exactly 3 sightings a candidate?  (only those change the set)
        hipLaunchKernelGGL(k_bloom_candidates, dim3(8192), dim3(256), 0, st, s->seen1, s->seen3, s->present, kBitmapWords,
                           F2.as<uint32_t>(), d_cand.as<uint32_t>(), (unsigned int *)d_n.p, cap, 1);
        flx_time_end(ctx);
        unsigned int n3 = 0;
        FLX_HIP(ctx, hipMemcpyAsync(&n3, d_n.p, 4, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipStreamSynchronize(st));
        s->bloom_candidates = n3;
        if (n3 > 0) {
            // Exact replay needs EVERY candidate non-inserter (any count), because a false-positive 16-mer does not
            // insert its bits and that can change the filter other 16-mers see.
            FLX_HIP(ctx, hipMemsetAsync(d_n.p, 0, 16, st));
            hipLaunchKernelGGL(k_bloom_candidates, dim3(8192), dim3(256), 0, st, s->seen1, s->seen3, s->present, kBitmapWords,
                               F2.as<uint32_t>(), d_cand.as<uint32_t>(), (unsigned int *)d_n.p, cap, 0);
            unsigned int nall = 0;
            FLX_HIP(ctx, hipMemcpyAsync(&nall, d_n.p, 4, hipMemcpyDeviceToHost, st));
            FLX_HIP(ctx, hipStreamSynchronize(st));
            if (nall > cap) return flx_fail(ctx, FLX_ERR_CAPACITY, "too many Bloom false-positive candidates (%u)", nall);
            std::vector<uint32_t> cand(nall);
            FLX_HIP(ctx, hipMemcpy(cand.data(), d_cand.p, (size_t)nall * 4, hipMemcpyDeviceToHost));
            std::sort(cand.begin(), cand.end());
            std::vector<uint8_t> is_fp;
            FLX_CHECK(resolve_bloom_candidates(s, cand, is_fp));
            // a false positive starts counting at 2 (kmers.cpp:152-155): 3 sightings are enough
            std::vector<uint32_t> promote;
            std::vector<uint32_t> h3(nall);
            {
                // which candidates have exactly 3 sightings: seen3 set, present clear
                flx_dbuf d_q, d_o3, d_op;
                FLX_CHECK(flx_dalloc(ctx, d_q, (size_t)nall * 4));
                FLX_CHECK(flx_dalloc(ctx, d_o3, nall));
                FLX_CHECK(flx_dalloc(ctx, d_op, nall));
                FLX_HIP(ctx, hipMemcpy(d_q.p, cand.data(), (size_t)nall * 4, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_contains, dim3((nall + 255) / 256), dim3(256), 0, st, s->seen3, d_q.as<uint32_t>(), (uint64_t)nall, d_o3.as<uint8_t>());
                hipLaunchKernelGGL(k_contains, dim3((nall + 255) / 256), dim3(256), 0, st, s->present, d_q.as<uint32_t>(), (uint64_t)nall, d_op.as<uint8_t>());
                std::vector<uint8_t> o3(nall), op(nall);
                FLX_HIP(ctx, hipStreamSynchronize(st));
                FLX_HIP(ctx, hipMemcpy(o3.data(), d_o3.p, nall, hipMemcpyDeviceToHost));
                FLX_HIP(ctx, hipMemcpy(op.data(), d_op.p, nall, hipMemcpyDeviceToHost));
                for (unsigned i = 0; i < nall; ++i)
                    if (is_fp[i]) {
                        ++s->bloom_false_positives;
                        if (o3[i] && !op[i]) promote.push_back(cand[i]);
                    }
            }
            if (!promote.empty()) {
                flx_dbuf d_p;
                FLX_CHECK(flx_dalloc(ctx, d_p, promote.size() * 4));
                FLX_HIP(ctx, hipMemcpy(d_p.p, promote.data(), promote.size() * 4, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_set_bits, dim3((unsigned)((promote.size() + 255) / 256)), dim3(256), 0, st, s->present,
                                   d_p.as<uint32_t>(), (uint64_t)promote.size());
                FLX_HIP(ctx, hipStreamSynchronize(st));
            }
        }
        // (the sequences stay until the locus text is built, below: their 17-mers are its witnesses)
        for (uint32_t **p : {&s->seen1, &s->seen2, &s->seen3, &s->asm_only}) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    }
    // size
    void *scr;
    FLX_CHECK(flx_scratch(ctx, 64, &scr));
    FLX_HIP(ctx, hipMemsetAsync(scr, 0, 8, st));
    hipLaunchKernelGGL(k_popcount, dim3(4096), dim3(256), 0, st, s->present, kBitmapWords, (unsigned long long *)scr);
    unsigned long long sz = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&sz, scr, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    s->size = sz;
    // prefilter: worth its L2 footprint while the 12-mer table stays sparse (a genome beyond ~20 Mbp saturates it)
    const char *pf_env = getenv("FLX_KMER_PREFILTER");  // "0" disables (A/B measurements)
    const bool pf_off = pf_env && pf_env[0] == '0';
    if (!pf_off && sz > 0 && sz < (3ull << (kPrefilterBits - 1))) {
        const size_t pf_bytes = (size_t)1 << (kPrefilterBits - 3);
        FLX_HIP(ctx, hipMalloc((void **)&s->prefilter, pf_bytes));
        FLX_HIP(ctx, hipMemsetAsync(s->prefilter, 0, pf_bytes, st));
        hipLaunchKernelGGL(k_build_prefilter, dim3(8192), dim3(256), 0, st, s->present, kBitmapWords, s->prefilter);
        FLX_HIP(ctx, hipMalloc((void **)&s->pre11, (size_t)2 << 20));
        FLX_HIP(ctx, hipMemsetAsync(s->pre11, 0, (size_t)2 << 20, st));
        hipLaunchKernelGGL(k_build_pre11, dim3(2048), dim3(256), 0, st, s->prefilter, s->pre11);
        FLX_HIP(ctx, hipStreamSynchronize(st));
    }
    if (sz > 0) {  // the pair form of the exact bitmap (what the cover kernel asks)
        // (no room for it: the set works without — scoring then takes the kernel that asks the 512 MiB bitmap, score_kmer.hip: k_kmer_cover)
        const char *pt = getenv("FLX_KMER_PAIRTABLE");  // "0": as if the allocation had failed (tests)
        hipError_t e = (pt && pt[0] == '0') ? hipErrorOutOfMemory : hipMalloc((void **)&s->exact15, (size_t)1 << 30);
        if (e != hipSuccess) {
            s->exact15 = nullptr;
            (void)hipGetLastError();
        } else {
            FLX_HIP(ctx, hipMemsetAsync(s->exact15, 0, (size_t)1 << 30, st));
            hipLaunchKernelGGL(k_build_exact15, dim3(8192), dim3(256), 0, st, s->present, kBitmapWords, s->exact15);
            FLX_HIP(ctx, hipStreamSynchronize(st));
        }
    }
    // the assembly as a text + seed table (kmerset.h).  Worth its memory while the 16-mer space is sparse: up to 2^28 text
    // positions (a 128 Mbp assembly; the seed table is then 4 GiB); FLX_KMER_LOCUS_BUILD=0 leaves it out.
    {
        uint64_t n_text = 0, n_windows = 0;
        for (auto &b : s->asm_batches) {
            n_text += 2 * (b.n_pos + 15 * b.n_seqs);
            n_windows += 2 * b.n_pos;
        }
        const char *lb = getenv("FLX_KMER_LOCUS_BUILD");
        if (s->has_short && sz > 0 && s->exact15 && !(lb && lb[0] == '0')) {
            // a set with short reads in it: the members themselves as a text (pathtext.hip) — every member is a window of it, so
            // U13 holds there too; an assembly underneath is part of the same graph
            std::vector<flx_seq_batch> wb;
            for (auto *list : {&s->asm_batches, &s->short_batches})
                for (auto &b : *list) wb.push_back({b.bases, b.offsets, b.pos_base, b.n_seqs, b.n_pos});
            // (the text is optional — include/filtlong_hip.h: "when that memory cannot be had the set works without" — so a build
            // that fails on its transient memory, ~12 GB for the C4 set, leaves a set without a text, not a failed finalize)
            // ONLY a failure for want of memory: a kernel fault or any other HIP error in the build is an error of finalize (it would
            // otherwise surface later in an unrelated call — hipGetLastError does not clear a sticky fault; advisor, round 5)
            const int text_rc = flx_build_path_text(ctx, s->present, (const uint8_t *)s->exact15, sz, wb.data(), wb.size(), &s->locus_text, &s->locus_seed, &s->locus);
            if (text_rc != FLX_OK) {
                if (s->locus_text) (void)hipFree(s->locus_text);
                if (s->locus_seed) (void)hipFree(s->locus_seed);
                s->locus_text = s->locus_seed = nullptr;
                const bool no_memory = text_rc == FLX_ERR_NOMEM || (text_rc == FLX_ERR_HIP && ctx->err.find(hipGetErrorString(hipErrorOutOfMemory)) != std::string::npos);
                if (!no_memory) return text_rc;
                (void)hipGetLastError();
                ctx->err.clear();  // (the set works without a text: nothing failed)
            }
            s->has_locus = s->locus_text != nullptr;
        } else if (!s->has_short && n_windows > 0 && n_text <= (1ull << 28) && !(lb && lb[0] == '0')) {
            // (!has_short: with short reads in the set — and no room for the pair table, or the text switched off above — the assembly's
            // text would NOT hold every member as a window, which is what U13 / S1 rest on: no text then, and no memory spent on one)
            const uint64_t n_words = (n_text + 15) / 16;
            const uint64_t n_alloc = n_words + kLocusPad + 68;
            int bits = 10;
            while ((1ull << bits) < n_windows * 5 / 2) ++bits;
            const uint64_t slots = 1ull << bits;
            hipError_t e1 = hipMalloc((void **)&s->locus_text, n_alloc * 8);
            hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **)&s->locus_seed, slots * 4) : e1;
            // the two counting planes of U13 as well, BEFORE anything is built: a failure here is "no text", like the two above
            uint32_t *u13_planes = nullptr;
            const size_t plane = (size_t)1 << (26 - 3);
            hipError_t e3 = e2 == hipSuccess ? hipMalloc((void **)&u13_planes, 2 * plane) : e2;
            if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) {
                struct PlaneGuard { uint32_t *p; ~PlaneGuard() { if (p) (void)hipFree(p); } } plane_guard{u13_planes};
                FLX_HIP(ctx, hipMemsetAsync(s->locus_text, 0, n_alloc * 8, st));
                // padding: no window may start or end there
                std::vector<uint32_t> pad_front(2 * kLocusPad), pad_back(2 * 68);
                for (size_t i = 0; i < pad_front.size(); i += 2) { pad_front[i] = 0; pad_front[i + 1] = 0xffffu; }
                for (size_t i = 0; i < pad_back.size(); i += 2) { pad_back[i] = 0; pad_back[i + 1] = 0xffffu; }
                FLX_HIP(ctx, hipMemcpyAsync(s->locus_text, pad_front.data(), pad_front.size() * 4, hipMemcpyHostToDevice, st));
                FLX_HIP(ctx, hipMemcpyAsync(s->locus_text + 2 * (kLocusPad + n_words), pad_back.data(), pad_back.size() * 4, hipMemcpyHostToDevice, st));
                FLX_HIP(ctx, hipMemsetAsync(s->locus_seed, 0xff, slots * 4, st));
                flx_time_begin(ctx, "flx_kmerset_locus_build");
                uint64_t tb = 0;
                for (auto &b : s->asm_batches) {
                    const uint64_t nb_bases = b.n_pos + 15 * b.n_seqs;
                    if (nb_bases)
                        hipLaunchKernelGGL(k_locus_text, dim3((unsigned)((nb_bases + 255) / 256)), dim3(256), 0, st, b.bases, b.offsets, b.pos_base,
                                           b.n_seqs, b.n_pos, tb, s->locus_text);
                    tb += 2 * nb_bases;
                }
                // the last word's bases behind the text must not look like the start of anything: a copy "starts" at n_text
                if (n_text % 16) {
                    const uint32_t tail_bits = 0xffffu & ~((1u << (n_text % 16)) - 1u);
                    hipLaunchKernelGGL(k_set_word_bits, dim3(1), dim3(1), 0, st, s->locus_text + 2 * (kLocusPad + n_words - 1) + 1, tail_bits);
                }
                {  // U13: the 13-mers that occur once
                    FLX_HIP(ctx, hipMemsetAsync(u13_planes, 0, 2 * plane, st));
                    uint32_t *seen1 = u13_planes, *seen2 = seen1 + plane / 4;
                    for (int pass = 0; pass < 2; ++pass) {
                        tb = 0;
                        for (auto &b : s->asm_batches) {
                            const uint64_t n_win = b.n_pos + 3 * b.n_seqs;
                            if (n_win && pass == 0)
                                hipLaunchKernelGGL(k_locus_u13<0>, dim3((unsigned)((n_win + 255) / 256)), dim3(256), 0, st, b.pos_base, b.n_seqs,
                                                   b.n_pos, tb, s->locus_text, seen1, seen2);
                            else if (n_win)
                                hipLaunchKernelGGL(k_locus_u13<1>, dim3((unsigned)((n_win + 255) / 256)), dim3(256), 0, st, b.pos_base, b.n_seqs,
                                                   b.n_pos, tb, s->locus_text, seen1, seen2);
                            tb += 2 * (b.n_pos + 15 * b.n_seqs);
                        }
                    }
                    FLX_HIP(ctx, hipStreamSynchronize(st));  // (the counters go out of scope)
                }
                tb = 0;
                for (auto &b : s->asm_batches) {
                    if (b.n_pos)
                        hipLaunchKernelGGL(k_locus_seed, dim3((unsigned)((b.n_pos + 255) / 256)), dim3(256), 0, st, b.pos_base, b.n_seqs, b.n_pos, tb,
                                           (const uint2 *)s->locus_text, s->locus_seed, (uint32_t)(slots - 1), 32 - bits);
                    tb += 2 * (b.n_pos + 15 * b.n_seqs);
                }
                flx_time_end(ctx);
                FLX_HIP(ctx, hipGetLastError());
                FLX_HIP(ctx, hipStreamSynchronize(st));
                s->locus.text = (const uint2 *)s->locus_text;
                s->locus.n_alloc = (uint32_t)n_alloc;
                s->locus.n_text = n_text;
                s->locus.seed = s->locus_seed;
                s->locus.seed_mask = (uint32_t)(slots - 1);
                s->locus.seed_shift = 32 - bits;
                s->has_locus = true;
            } else {  // not enough memory for it: the scoring path works without
                if (s->locus_text) (void)hipFree(s->locus_text);
                if (s->locus_seed) (void)hipFree(s->locus_seed);
                if (u13_planes) (void)hipFree(u13_planes);
                s->locus_text = s->locus_seed = nullptr;
                (void)hipGetLastError();
            }
        }
        free_batches(s->asm_batches);
        free_batches(s);
        s->locus.safe1 = nullptr;
        const char *s1 = getenv("FLX_KMER_SAFE1");  // "0": without S1 (tests, A/B)
        if (s->has_locus && !(s1 && s1[0] == '0')) {
            const size_t bytes = (((size_t)s->locus.n_alloc + 1) / 2) * 4;
            if (hipMalloc((void **)&s->locus_safe1, bytes) == hipSuccess) {
                FLX_HIP(ctx, hipMemsetAsync(s->locus_safe1, 0, bytes, st));
                flx_time_begin(ctx, "flx_kmerset_locus_build");
                hipLaunchKernelGGL(k_text_safe1, dim3((unsigned)((s->locus.n_text + 255) / 256)), dim3(256), 0, st, s->locus.text, s->locus.n_text,
                                   s->present, s->locus_safe1);
                flx_time_end(ctx);
                FLX_HIP(ctx, hipGetLastError());
                FLX_HIP(ctx, hipStreamSynchronize(st));
                s->locus.safe1 = (const uint16_t *)s->locus_safe1;
            } else {
                s->locus_safe1 = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    s->final_ = true;
    return FLX_OK;
}

extern "C" uint64_t flx_kmerset_size(const flx_kmerset *s) { return s ? s->size : 0; }

extern "C" int flx_kmerset_contains(const flx_kmerset *s, const uint32_t *kmers, uint64_t n, uint8_t *present) {
    if (!s) return FLX_ERR_INVALID;
    flx_ctx *ctx = s->ctx;
    if (!s->final_) return flx_fail(ctx, FLX_ERR_STATE, "k-mer set is not finalized");
    if (n == 0) return FLX_OK;
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    flx_dbuf d_k, d_o;
    FLX_CHECK(flx_dalloc(ctx, d_k, n * 4));
    FLX_CHECK(flx_dalloc(ctx, d_o, n));
    FLX_HIP(ctx, hipMemcpyAsync(d_k.p, kmers, n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_contains, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, s->present, d_k.as<uint32_t>(), n,
                       d_o.as<uint8_t>());
    FLX_HIP(ctx, hipMemcpyAsync(present, d_o.p, n, hipMemcpyDeviceToHost, ctx->stream));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLX_OK;
}
