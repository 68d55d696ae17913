// pathtext.hip — a 16-mer set without an assembly (short reads, src/kmers.cpp:142-166) as a TEXT for the locus path of the cover
// kernel (kmerset.h: flx_locus; score_kmer.hip: k_kmer_cover_w<.., LOCUS>).
//
// The locus path needs a text in which every 16-base window inside one piece is a member, and — for the refutation by unique
// 13-mers — in which every member IS such a window.  An assembly is that text for its own 16-mers.  For any other set the
// members are the edges of a de Bruijn graph over 15-mers (member x.C enters node C, member C.y leaves it; exact15 holds both
// nibbles per node), and a decomposition of the edges into PATHS gives the pieces: at every node the i-th entering member (in
// the order of its first base) is continued by the i-th leaving member (in the order of its last base), entering members
// without a partner end their path, leaving members without one start a path.  Where a node has more than one entering or
// leaving member the SEQUENCES the set was built from say which belong together: every 17-mer x.C.y of theirs whose two
// 16-mers are members is a witness for "x.C is continued by C.y" (kept in a small hash table of the branching nodes only), and
// witnessed pairs are matched first — then a path follows the genome through a repeated 15-mer and only ends where a whole
// 16-mer repeats (a wrong or missing witness costs speed, never exactness: any pairing gives a valid text).  Every member lies on exactly one path (or on
// a cycle: every member of a cycle becomes a path of its own), a path of m members is a piece of m + 15 bases, and along a
// genome the pairing is wrong only where a 15-mer repeats — about every 200 bases of a 5 Mbp genome, where the cover kernel
// seeds again inside the span.  Everything runs on the device: ranks from a popcount index of the bitmap, predecessor links,
// pointer jumping to (head, distance), lengths, offsets by an exclusive scan, the text, U13, the seed table.
//
// Round 4, second form (flx_build_path_text, first choice): the same construction one order higher.  At order 16 a piece ends
// wherever a 16-mer of the genome repeats (every ~580 bases of a read through a 5 Mbp genome even with the witnesses); the
// sequences the set was built from are longer than that.  So the graph is built over their 24-MERS (48 bits; a 23-mer of a 5 Mbp
// genome practically never repeats): every 24-mer of both strands whose nine 16-mers are all members, sorted and made unique
// (radix sort), linked by binary search, cut into paths exactly like the 16-mers above — a piece of m 24-mers is a text of m + 23
// bases in which every 16-base window is a member by construction.  Members that no such 24-mer holds (the ends of the
// sequences, 16-mers that only just made the count) become pieces of their own, so that every member is a window of the text
// (U13).  U13 and the seeds are then taken from the text itself, one thread per text position.
#include <cstring>
#include <vector>

#include "flx_internal.h"
#include "kmerset.h"
#include "rank_internal.h"

namespace {

__device__ __forceinline__ bool pt_test_bit(const uint32_t *bm, uint32_t k) { return (bm[k >> 5] >> (k & 31)) & 1u; }
__device__ __forceinline__ bool pt_set_bit(uint32_t *bm, uint32_t k) {
    const uint32_t m = 1u << (k & 31);
    return (atomicOr(&bm[k >> 5], m) & m) != 0;
}

// members per 256 bits of the bitmap (2^24 blocks)
__global__ void __launch_bounds__(256) k_pt_count256(const uint32_t *bm, int64_t *cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 a = reinterpret_cast<const uint4 *>(bm)[2 * (size_t)i], b = reinterpret_cast<const uint4 *>(bm)[2 * (size_t)i + 1];
    cnt[i] = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
}

// index of member v among the members in increasing order
__device__ __forceinline__ uint32_t pt_rank(const uint32_t *bm, const int64_t *pre256, uint32_t v) {
    const uint32_t w = v >> 5;
    uint32_t r = (uint32_t)pre256[v >> 8];
    for (uint32_t k = w & ~7u; k < w; ++k) r += __popc(bm[k]);
    return r + __popc(bm[w] & ((1u << (v & 31)) - 1u));
}

__global__ void __launch_bounds__(256) k_pt_members(const uint32_t *bm, const int64_t *pre256, uint32_t *members) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t at = (uint32_t)pre256[i];
    for (uint32_t k = 0; k < 8; ++k) {
        uint32_t w = bm[8 * (size_t)i + k];
        while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            members[at++] = ((8u * i + k) << 5) | (uint32_t)b;
        }
    }
}

// ---- witnesses at branching nodes: bit 4 x + y of a node's entry = "a sequence holds x.C.y" ----
struct PtWitness {
    uint32_t *keys;  // node + 1, 0 = empty
    uint32_t *vals;
    uint32_t mask;
};
__device__ __forceinline__ uint32_t pt_node_hash(uint32_t node, uint32_t mask) { return ((node * 0x9E3779B1u) >> 7) & mask; }
__device__ __forceinline__ bool pt_branching(uint32_t byte) { return (byte & 15u) && (byte >> 4) && (__popc(byte & 15u) > 1 || __popc(byte >> 4) > 1); }

__global__ void __launch_bounds__(256) k_pt_witness(const uint8_t *bases, const uint64_t *offsets, const uint64_t *pos_base, uint64_t n_seqs,
                                                    uint64_t n_pos, const uint8_t *exact15, PtWitness w) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t p = g - pos_base[lo];
    const uint64_t len = (lo + 1 < n_seqs ? pos_base[lo + 1] : n_pos) - pos_base[lo] + 15;
    if (p + 17 > len) return;
    const uint8_t *sq = bases + offsets[lo] + p;
    // the 17 codes of both strands (src/kmers.cpp:176-219): forward as they come, reverse from the other end in the reverse encoder's codes
    uint64_t f = 0, r = 0;
    for (int j = 0; j < 17; ++j) {
        uint32_t cf = 0, cr = 0;
        switch (sq[j]) {
            case 'A': case 'a': cr = 3; break;
            case 'C': case 'c': cf = 1; cr = 2; break;
            case 'G': case 'g': cf = 2; cr = 1; break;
            case 'T': case 't': cf = 3; break;
            default: break;
        }
        f = (f << 2) | cf;
        r |= (uint64_t)cr << (2 * j);
    }
    const uint64_t two[2] = {f, r};
    for (int q = 0; q < 2; ++q) {
        const uint64_t v = two[q];  // 34 bits: x . C (30 bits) . y
        const uint32_t x = (uint32_t)(v >> 32) & 3u, node = (uint32_t)(v >> 2) & 0x3FFFFFFFu, y = (uint32_t)v & 3u;
        const uint32_t byte = exact15[node];
        if (!pt_branching(byte) || !((byte >> x) & 1u) || !((byte >> (4 + y)) & 1u)) continue;
        uint32_t h = pt_node_hash(node, w.mask);
        for (int probe = 0; probe < 64; ++probe) {
            const uint32_t old = atomicCAS(&w.keys[h], 0u, node + 1u);
            if (old == 0u || old == node + 1u) {
                atomicOr(&w.vals[h], 1u << (4 * x + y));
                break;
            }
            h = (h + 1) & w.mask;
        }
    }
}

// Which entering member (first base x) is continued by the leaving member with last base y at this node: witnessed pairs first
// (entering members in the order of x, each takes the smallest witnessed partner still free), then the rest in order.  4 = none.
__device__ __forceinline__ uint32_t pt_partner_of_out(uint32_t byte, uint32_t wit, uint32_t y) {
    const uint32_t in_mask = byte & 15u, out_mask = byte >> 4;
    uint32_t taken_out = 0, matched_in = 0, answer = 4;
    for (uint32_t x = 0; x < 4; ++x) {
        if (!((in_mask >> x) & 1u)) continue;
        const uint32_t c = out_mask & ((wit >> (4 * x)) & 15u) & ~taken_out;
        if (!c) continue;
        const uint32_t yy = (uint32_t)(__ffs(c) - 1);
        taken_out |= 1u << yy;
        matched_in |= 1u << x;
        if (yy == y) answer = x;
    }
    if ((taken_out >> y) & 1u) return answer;
    uint32_t free_in = in_mask & ~matched_in, free_out = out_mask & ~taken_out;
    const uint32_t rank = __popc(free_out & ((1u << y) - 1u));
    if (rank >= (uint32_t)__popc(free_in)) return 4;
    for (uint32_t k = 0; k < rank; ++k) free_in &= free_in - 1;
    return (uint32_t)(__ffs(free_in) - 1);
}

// predecessor on the path of every member (itself: the member starts a path)
__global__ void __launch_bounds__(256) k_pt_pred(const uint32_t *bm, const int64_t *pre256, const uint8_t *exact15, const uint32_t *members,
                                                 uint32_t n, PtWitness w, uint32_t *link, uint32_t *dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = members[i];
    const uint32_t node = e >> 2, y = e & 3u;  // e leaves the node of its first 15 bases with its last base y
    const uint32_t byte = exact15[node];
    uint32_t wit = 0;
    if (pt_branching(byte)) {
        uint32_t h = pt_node_hash(node, w.mask);
        for (int probe = 0; probe < 64; ++probe) {
            const uint32_t k = w.keys[h];
            if (k == 0u) break;
            if (k == node + 1u) { wit = w.vals[h]; break; }
            h = (h + 1) & w.mask;
        }
    }
    const uint32_t x = pt_partner_of_out(byte, wit, y);
    uint32_t p = i;
    if (x < 4) {
        const uint32_t pe = (x << 30) | node;  // the entering member x.node
        if (pe != e) p = pt_rank(bm, pre256, pe);
    }
    link[i] = p;
    dist[i] = p == i ? 0u : 1u;
}

__global__ void __launch_bounds__(256) k_pt_jump(uint32_t n, const uint32_t *link_in, const uint32_t *dist_in, uint32_t *link_out, uint32_t *dist_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = link_in[i];
    dist_out[i] = dist_in[i] + dist_in[p];
    link_out[i] = link_in[p];
}

// after the jumps link[i] is the head of i's path — unless i lies on a cycle, where no member points at itself: such members become
// paths of their own.  Then the longest distance per head.
__global__ void __launch_bounds__(256) k_pt_heads(uint32_t n, uint32_t *link, uint32_t *dist, uint32_t *len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = link[i];
    if (link[h] != h) {  // (reads its neighbours' links of the jump phase: the rewrite goes to `len`'s kernel below, not here)
        dist[i] = 0x80000000u;  // marked; resolved in k_pt_lengths
    }
}
__global__ void __launch_bounds__(256) k_pt_lengths(uint32_t n, uint32_t *link, uint32_t *dist, uint32_t *len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (dist[i] & 0x80000000u) {
        link[i] = i;
        dist[i] = 0;
    }
    atomicMax(&len[link[i]], dist[i] + 1u);
}
__global__ void __launch_bounds__(256) k_pt_piece_bases(uint32_t n, const uint32_t *link, const uint32_t *len, int64_t *bases) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bases[i] = link[i] == i ? (int64_t)len[i] + 15 : 0;
}

__device__ __forceinline__ void pt_put(uint32_t *text_words, uint64_t t, uint32_t code, bool starts_piece) {
    uint32_t *w = text_words + 2 * ((t >> 4) + kLocusPad);
    if (code) atomicOr(w, code << (30 - 2 * (uint32_t)(t & 15)));
    if (starts_piece) atomicOr(w + 1, 1u << (uint32_t)(t & 15));
}

__global__ void __launch_bounds__(256) k_pt_text(uint32_t n, const uint32_t *members, const uint32_t *link, const uint32_t *dist, const int64_t *off,
                                                 uint32_t *text_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = members[i];
    const uint64_t o = (uint64_t)off[link[i]];
    pt_put(text_words, o + 15 + dist[i], e & 3u, false);
    if (dist[i] == 0) {
        for (int j = 0; j < 15; ++j) pt_put(text_words, o + j, (e >> (30 - 2 * j)) & 3u, j == 0);
    }
}

// U13 (kmerset.h): every member's first 13 bases, and behind the last member of a piece the three 13-mers that follow
template <int PASS>
__global__ void __launch_bounds__(256) k_pt_u13(uint32_t n, const uint32_t *members, const uint32_t *link, const uint32_t *dist, const uint32_t *len,
                                                const int64_t *off, uint32_t *text_words, uint32_t *seen1, uint32_t *seen2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = members[i], h = link[i];
    const uint64_t t = (uint64_t)off[h] + dist[i];
    const int extra = dist[i] + 1 == len[h] ? 3 : 0;
    for (int k = 0; k <= extra; ++k) {
        const uint32_t v = (e >> (6 - 2 * k)) & 0x3FFFFFFu;
        if (PASS == 0) {
            if (pt_set_bit(seen1, v)) pt_set_bit(seen2, v);
        } else if (!pt_test_bit(seen2, v)) {
            atomicOr(text_words + 2 * (((t + k) >> 4) + kLocusPad) + 1, 0x10000u << (uint32_t)((t + k) & 15));
        }
    }
}

__global__ void __launch_bounds__(256) k_pt_seed(uint32_t n, const uint32_t *members, const uint32_t *link, const uint32_t *dist, const int64_t *off,
                                                 const uint2 *text, uint32_t *seed, uint32_t mask, int shift) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = members[i];
    const uint32_t t = (uint32_t)((uint64_t)off[link[i]] + dist[i]);
    uint32_t h = flx_locus_hash(k, shift);
    for (;;) {
        const uint32_t old = atomicCAS(&seed[h], kLocusEmpty, t);
        if (old == kLocusEmpty) break;
        if (flx_locus_kmer_at(text, old) == k) {  // (cannot happen: every member is the window of one text position)
            atomicMin(&seed[h], t);
            break;
        }
        h = (h + 1) & mask;
    }
}

__global__ void k_pt_set_word_bits(uint32_t *word, uint32_t bits) { *word |= bits; }

// ---------------------------------------------------------------------------------------------------------------------
// order 24
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint64_t kNo24 = ~0ull;

// both strands' 24-mers at every start position of the sequences (key = 48 bits of codes, or kNo24 when one of its nine 16-mers is no member)
__global__ void __launch_bounds__(256) k24_emit(const uint8_t *bases, const uint64_t *offsets, const uint64_t *pos_base, uint64_t n_seqs, uint64_t n_pos,
                                                const uint32_t *present, uint64_t *keys) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    uint64_t lo = 0, hi = n_seqs;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos_base[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t p = g - pos_base[lo];
    const uint64_t len = (lo + 1 < n_seqs ? pos_base[lo + 1] : n_pos) - pos_base[lo] + 15;
    uint64_t kf = kNo24, kr = kNo24;
    if (p + 24 <= len) {
        const uint8_t *sq = bases + offsets[lo] + p;
        uint64_t f = 0, r = 0;
        for (int j = 0; j < 24; ++j) {
            uint32_t cf = 0, cr = 0;
            switch (sq[j]) {
                case 'A': case 'a': cr = 3; break;
                case 'C': case 'c': cf = 1; cr = 2; break;
                case 'G': case 'g': cf = 2; cr = 1; break;
                case 'T': case 't': cf = 3; break;
                default: break;
            }
            f = (f << 2) | cf;
            r |= (uint64_t)cr << (2 * j);
        }
        bool okf = true, okr = true;
        for (int w = 0; w < 9; ++w) {
            okf = okf && pt_test_bit(present, (uint32_t)(f >> (2 * w)));
            okr = okr && pt_test_bit(present, (uint32_t)(r >> (2 * w)));
        }
        if (okf) kf = f;
        if (okr) kr = r;
    }
    keys[2 * g] = kf;
    keys[2 * g + 1] = kr;
}

__global__ void __launch_bounds__(256) k24_flag(uint64_t n, const uint64_t *keys, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (keys[i] != kNo24 && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k24_compact(uint64_t n, const uint64_t *keys, const uint32_t *flag, const uint32_t *at, uint64_t *edges) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) edges[at[i]] = keys[i];
}

__device__ __forceinline__ uint32_t k24_lower_bound(const uint64_t *e, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (e[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// predecessor of every 24-mer on its path: entering and leaving edges of the node of its first 23 bases paired in order
__global__ void __launch_bounds__(256) k24_pred(const uint64_t *edges, uint32_t n, uint32_t *link, uint32_t *dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = edges[i];
    const uint64_t node = e >> 2;  // 46 bits
    const uint32_t first_out = k24_lower_bound(edges, n, node << 2);
    const uint32_t out_rank = i - first_out;  // the leaving edges of a node are neighbours in the sorted array
    uint32_t in_idx[4], n_in = 0;
    for (uint64_t x = 0; x < 4; ++x) {
        const uint64_t key = (x << 46) | node;
        const uint32_t at = k24_lower_bound(edges, n, key);
        if (at < n && edges[at] == key) in_idx[n_in++] = at;
    }
    uint32_t p = i;
    if (out_rank < n_in && in_idx[out_rank] != i) p = in_idx[out_rank];
    link[i] = p;
    dist[i] = p == i ? 0u : 1u;
}

__global__ void __launch_bounds__(256) k24_piece_bases(uint32_t n, const uint32_t *link, const uint32_t *len, int64_t *bases) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bases[i] = link[i] == i ? (int64_t)len[i] + 23 : 0;
}

__global__ void __launch_bounds__(256) k24_cover(uint32_t n, const uint64_t *edges, uint32_t *covered) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = edges[i];
    for (int w = 0; w < 9; ++w) pt_set_bit(covered, (uint32_t)(e >> (2 * w)));
}
__global__ void __launch_bounds__(256) k24_leftover(const uint32_t *present, uint32_t *covered, uint64_t n_words) {  // covered := present & ~covered
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) covered[i] = present[i] & ~covered[i];
}

__global__ void __launch_bounds__(256) k24_text(uint32_t n, const uint64_t *edges, const uint32_t *link, const uint32_t *dist, const int64_t *off,
                                                uint32_t *text_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = edges[i];
    const uint64_t o = (uint64_t)off[link[i]];
    pt_put(text_words, o + 23 + dist[i], (uint32_t)(e & 3u), false);
    if (dist[i] == 0) {
        for (int j = 0; j < 23; ++j) pt_put(text_words, o + j, (uint32_t)((e >> (46 - 2 * j)) & 3u), j == 0);
    }
}
__global__ void __launch_bounds__(256) k24_text_leftover(uint32_t n, const uint32_t *members, uint64_t text_base, uint32_t *text_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = members[i];
    for (int j = 0; j < 16; ++j) pt_put(text_words, text_base + 16ull * i + j, (e >> (30 - 2 * j)) & 3u, j == 0);
}

// ---- U13 and seeds from the text itself: one thread per text position ----
// no piece starts at t + 1 .. t + len - 1 and the window ends inside the text
__device__ __forceinline__ bool pt_in_piece(const uint2 *text, uint64_t n_text, uint64_t t, int len) {
    if (t + len > n_text) return false;
    const uint64_t t1 = t + 1;
    const uint64_t w = (t1 >> 4) + kLocusPad;
    const uint32_t s = (uint32_t)(t1 & 15);
    const uint32_t b = ((text[w].y & 0xffffu) >> s) | ((text[w + 1].y & 0xffffu) << (16 - s));
    return (b & ((1u << (len - 1)) - 1u)) == 0;
}
template <int PASS>
__global__ void __launch_bounds__(256) k_text_u13(uint32_t *text_words, uint64_t n_text, uint32_t *seen1, uint32_t *seen2) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_text || !pt_in_piece((const uint2 *)text_words, n_text, t, 13)) return;
    const uint32_t v = flx_locus_kmer_at((const uint2 *)text_words, (uint32_t)t) >> 6;
    if (PASS == 0) {
        if (pt_set_bit(seen1, v)) pt_set_bit(seen2, v);
    } else if (!pt_test_bit(seen2, v)) {
        atomicOr(text_words + 2 * ((t >> 4) + kLocusPad) + 1, 0x10000u << (uint32_t)(t & 15));
    }
}
__global__ void __launch_bounds__(256) k_text_seed(const uint2 *text, uint64_t n_text, uint32_t *seed, uint32_t mask, int shift) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_text || !pt_in_piece(text, n_text, t, 16)) return;
    const uint32_t k = flx_locus_kmer_at(text, (uint32_t)t);
    uint32_t h = flx_locus_hash(k, shift);
    for (;;) {
        const uint32_t old = atomicCAS(&seed[h], kLocusEmpty, (uint32_t)t);
        if (old == kLocusEmpty) break;
        if (flx_locus_kmer_at(text, old) == k) {  // the same 16-mer at another place of the text: the smallest position stays
            atomicMin(&seed[h], (uint32_t)t);
            break;
        }
        h = (h + 1) & mask;
    }
}

}  // namespace

// Order 24 (see the file header).  *text_out stays nullptr when this form cannot be built (no sequences, too many, no memory):
// the caller then takes the order-16 form below.
static int build_path_text_k24(flx_ctx *ctx, const uint32_t *present, uint64_t n_members, const flx_seq_batch *batches, size_t n_batches,
                               uint32_t **text_out, uint32_t **seed_out, flx_locus *loc) {
    hipStream_t st = ctx->stream;
    uint64_t n_pos = 0;
    for (size_t b = 0; b < n_batches; ++b) n_pos += batches[b].n_pos;
    if (n_pos == 0 || 2 * n_pos >= (1ull << 32)) return FLX_OK;
    const uint64_t nk = 2 * n_pos;
    auto alloc = [&](flx_dbuf &b, size_t bytes) { return hipMalloc(&b.p, bytes) == hipSuccess; };
    flx_dbuf k0, k1, v0, v1, ws, flag, at;
    const size_t ws_bytes = flx_radix_sort_workspace(nk + 1);
    if (!alloc(k0, nk * 8) || !alloc(k1, nk * 8) || !alloc(v0, nk * 4) || !alloc(v1, nk * 4) || !alloc(ws, ws_bytes) || !alloc(flag, (nk + 1) * 4) ||
        !alloc(at, (nk + 1) * 4)) {
        (void)hipGetLastError();
        return FLX_OK;
    }
    flx_time_begin(ctx, "flx_kmerset_locus_build");
    uint64_t done = 0;
    for (size_t b = 0; b < n_batches; ++b) {
        if (!batches[b].n_pos) continue;
        hipLaunchKernelGGL(k24_emit, dim3((unsigned)((batches[b].n_pos + 255) / 256)), dim3(256), 0, st, batches[b].bases, batches[b].offsets, batches[b].pos_base,
                           batches[b].n_seqs, batches[b].n_pos, present, k0.as<uint64_t>() + 2 * done);
        done += batches[b].n_pos;
    }
    uint64_t *sk = nullptr;
    uint32_t *sv = nullptr;
    FLX_CHECK(flx_radix_sort_pairs(ctx, nk, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), ws.p, ws_bytes, &sk, &sv));
    FLX_HIP(ctx, hipMemsetAsync(flag.p, 0, (nk + 1) * 4, st));
    hipLaunchKernelGGL(k24_flag, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, nk, sk, flag.as<uint32_t>());
    FLX_CHECK(flx_exclusive_scan_u32(ctx, nk + 1, flag.as<uint32_t>(), at.as<uint32_t>(), ws.p, ws_bytes));
    uint32_t m = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&m, at.as<uint32_t>() + nk, 4, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    if (m == 0 || m > (1u << 27)) {
        flx_time_end(ctx);
        return FLX_OK;
    }
    uint64_t *edges = sk == k0.as<uint64_t>() ? k1.as<uint64_t>() : k0.as<uint64_t>();  // the buffer the sort did not end in
    hipLaunchKernelGGL(k24_compact, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, nk, sk, flag.as<uint32_t>(), at.as<uint32_t>(), edges);
    const uint32_t nb = (m + 255) / 256;
    // paths (the same jumps as at order 16)
    flx_dbuf d_link0, d_link1, d_dist0, d_dist1, d_len, d_bases, d_off, d_cov, d_cnt, d_pre, d_left, ws2;
    const uint64_t n_blocks = 1ull << 24;
    const size_t ws2_bytes = std::max(flx_radix_sort_workspace(n_blocks + 1), flx_radix_sort_workspace((uint64_t)m + 1));
    if (!alloc(d_link0, (size_t)m * 4) || !alloc(d_link1, (size_t)m * 4) || !alloc(d_dist0, (size_t)m * 4) || !alloc(d_dist1, (size_t)m * 4) || !alloc(d_len, (size_t)m * 4) ||
        !alloc(d_bases, ((size_t)m + 1) * 8) || !alloc(d_off, ((size_t)m + 1) * 8) || !alloc(d_cov, (size_t)1 << 29) || !alloc(d_cnt, (n_blocks + 1) * 8) ||
        !alloc(d_pre, (n_blocks + 1) * 8) || !alloc(ws2, ws2_bytes)) {
        (void)hipGetLastError();
        flx_time_end(ctx);
        return FLX_OK;
    }
    hipLaunchKernelGGL(k24_pred, dim3(nb), dim3(256), 0, st, edges, m, d_link0.as<uint32_t>(), d_dist0.as<uint32_t>());
    uint32_t *link = d_link0.as<uint32_t>(), *link2 = d_link1.as<uint32_t>(), *dist = d_dist0.as<uint32_t>(), *dist2 = d_dist1.as<uint32_t>();
    int rounds = 1;
    while ((1ull << rounds) < (uint64_t)m + 1) ++rounds;
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(k_pt_jump, dim3(nb), dim3(256), 0, st, m, link, dist, link2, dist2);
        std::swap(link, link2);
        std::swap(dist, dist2);
    }
    FLX_HIP(ctx, hipMemsetAsync(d_len.p, 0, (size_t)m * 4, st));
    hipLaunchKernelGGL(k_pt_heads, dim3(nb), dim3(256), 0, st, m, link, dist, d_len.as<uint32_t>());
    hipLaunchKernelGGL(k_pt_lengths, dim3(nb), dim3(256), 0, st, m, link, dist, d_len.as<uint32_t>());
    FLX_HIP(ctx, hipMemsetAsync(d_bases.p, 0, ((size_t)m + 1) * 8, st));
    hipLaunchKernelGGL(k24_piece_bases, dim3(nb), dim3(256), 0, st, m, link, d_len.as<uint32_t>(), d_bases.as<int64_t>());
    FLX_CHECK(flx_exclusive_scan_i64(ctx, (uint64_t)m + 1, d_bases.as<int64_t>(), d_off.as<int64_t>(), ws2.p, ws2_bytes));
    // members no 24-mer holds
    FLX_HIP(ctx, hipMemsetAsync(d_cov.p, 0, (size_t)1 << 29, st));
    hipLaunchKernelGGL(k24_cover, dim3(nb), dim3(256), 0, st, m, edges, d_cov.as<uint32_t>());
    hipLaunchKernelGGL(k24_leftover, dim3(8192), dim3(256), 0, st, present, d_cov.as<uint32_t>(), (uint64_t)1 << 27);
    FLX_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, (n_blocks + 1) * 8, st));
    hipLaunchKernelGGL(k_pt_count256, dim3((unsigned)(n_blocks / 256)), dim3(256), 0, st, d_cov.as<uint32_t>(), d_cnt.as<int64_t>());
    FLX_CHECK(flx_exclusive_scan_i64(ctx, n_blocks + 1, d_cnt.as<int64_t>(), d_pre.as<int64_t>(), ws2.p, ws2_bytes));
    int64_t path_bases = 0, n_left = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&path_bases, d_off.as<int64_t>() + m, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipMemcpyAsync(&n_left, d_pre.as<int64_t>() + n_blocks, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t n_text = (uint64_t)path_bases + 16ull * (uint64_t)n_left;
    if (n_text == 0 || n_text > (1ull << 28) || (uint64_t)n_left > n_members) {
        flx_time_end(ctx);
        return FLX_OK;
    }
    if (n_left > 0 && !alloc(d_left, (size_t)n_left * 4)) {
        (void)hipGetLastError();
        flx_time_end(ctx);
        return FLX_OK;
    }
    const uint64_t n_words = (n_text + 15) / 16;
    const uint64_t n_alloc = n_words + kLocusPad + 68;
    int bits = 10;
    while ((1ull << bits) < n_members * 5 / 2) ++bits;
    const uint64_t slots = 1ull << bits;
    uint32_t *text = nullptr, *seed = nullptr;
    flx_dbuf seen;
    const size_t plane = (size_t)1 << (26 - 3);
    if (hipMalloc((void **)&text, n_alloc * 8) != hipSuccess || hipMalloc((void **)&seed, slots * 4) != hipSuccess || !alloc(seen, 2 * plane)) {
        if (text) (void)hipFree(text);
        if (seed) (void)hipFree(seed);
        (void)hipGetLastError();
        flx_time_end(ctx);
        return FLX_OK;
    }
    FLX_HIP(ctx, hipMemsetAsync(text, 0, n_alloc * 8, st));
    std::vector<uint32_t> pad_front(2 * kLocusPad), pad_back(2 * 68);
    for (size_t i = 0; i < pad_front.size(); i += 2) { pad_front[i] = 0; pad_front[i + 1] = 0xffffu; }
    for (size_t i = 0; i < pad_back.size(); i += 2) { pad_back[i] = 0; pad_back[i + 1] = 0xffffu; }
    FLX_HIP(ctx, hipMemcpyAsync(text, pad_front.data(), pad_front.size() * 4, hipMemcpyHostToDevice, st));
    FLX_HIP(ctx, hipMemcpyAsync(text + 2 * (kLocusPad + n_words), pad_back.data(), pad_back.size() * 4, hipMemcpyHostToDevice, st));
    FLX_HIP(ctx, hipMemsetAsync(seed, 0xff, slots * 4, st));
    FLX_HIP(ctx, hipMemsetAsync(seen.p, 0, 2 * plane, st));
    hipLaunchKernelGGL(k24_text, dim3(nb), dim3(256), 0, st, m, edges, link, dist, d_off.as<int64_t>(), text);
    if (n_left > 0) {
        hipLaunchKernelGGL(k_pt_members, dim3((unsigned)(n_blocks / 256)), dim3(256), 0, st, d_cov.as<uint32_t>(), d_pre.as<int64_t>(), d_left.as<uint32_t>());
        hipLaunchKernelGGL(k24_text_leftover, dim3((unsigned)((n_left + 255) / 256)), dim3(256), 0, st, (uint32_t)n_left, d_left.as<uint32_t>(), (uint64_t)path_bases, text);
    }
    if (n_text % 16)
        hipLaunchKernelGGL(k_pt_set_word_bits, dim3(1), dim3(1), 0, st, text + 2 * (kLocusPad + n_words - 1) + 1, 0xffffu & ~((1u << (n_text % 16)) - 1u));
    uint32_t *seen1 = seen.as<uint32_t>(), *seen2 = seen1 + plane / 4;
    const unsigned tb = (unsigned)((n_text + 255) / 256);
    hipLaunchKernelGGL(k_text_u13<0>, dim3(tb), dim3(256), 0, st, text, n_text, seen1, seen2);
    hipLaunchKernelGGL(k_text_u13<1>, dim3(tb), dim3(256), 0, st, text, n_text, seen1, seen2);
    hipLaunchKernelGGL(k_text_seed, dim3(tb), dim3(256), 0, st, (const uint2 *)text, n_text, seed, (uint32_t)(slots - 1), 32 - bits);
    flx_time_end(ctx);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(st));
    loc->text = (const uint2 *)text;
    loc->n_alloc = (uint32_t)n_alloc;
    loc->n_text = n_text;
    loc->seed = seed;
    loc->seed_mask = (uint32_t)(slots - 1);
    loc->seed_shift = 32 - bits;
    *text_out = text;
    *seed_out = seed;
    return FLX_OK;
}

// Builds text + seed table for the members of `present` (n_members of them; exact15 is their pair table).  On success the caller
// owns *text_out / *seed_out (hipFree) and `loc` describes them; returns FLX_OK with *text_out == nullptr when the set is too
// large for it or the device memory is not there (the scoring path works without).
int flx_build_path_text(flx_ctx *ctx, const uint32_t *present, const uint8_t *exact15, uint64_t n_members, const flx_seq_batch *batches,
                        size_t n_batches, uint32_t **text_out, uint32_t **seed_out, flx_locus *loc) {
    *text_out = nullptr;
    *seed_out = nullptr;
    if (n_members == 0 || n_members > (1ull << 27)) return FLX_OK;
    {
        const char *order = getenv("FLX_KMER_TEXT_ORDER");  // "16": the order-16 form only (second implementation; tests, A/B)
        if (!(order && strcmp(order, "16") == 0)) {
            FLX_CHECK(build_path_text_k24(ctx, present, n_members, batches, n_batches, text_out, seed_out, loc));
            if (*text_out) return FLX_OK;
        }
    }
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)n_members;
    const uint32_t nb = (n + 255) / 256;
    const uint64_t n_blocks = 1ull << 24;
    flx_dbuf d_cnt, d_pre, d_members, d_link0, d_link1, d_dist0, d_dist1, d_len, d_bases, d_off, d_ws, d_wkeys, d_wvals;
    const uint32_t wit_slots = 1u << 22;  // (a 5 Mbp genome has ~1e5 branching 15-mers; what does not fit is paired by order)
    const size_t ws_bytes = std::max(flx_radix_sort_workspace(n_blocks + 1), flx_radix_sort_workspace((uint64_t)n + 1));
    auto alloc = [&](flx_dbuf &b, size_t bytes) { return hipMalloc(&b.p, bytes) == hipSuccess; };
    if (!alloc(d_cnt, (n_blocks + 1) * 8) || !alloc(d_pre, (n_blocks + 1) * 8) || !alloc(d_members, (size_t)n * 4) || !alloc(d_link0, (size_t)n * 4) ||
        !alloc(d_link1, (size_t)n * 4) || !alloc(d_dist0, (size_t)n * 4) || !alloc(d_dist1, (size_t)n * 4) || !alloc(d_len, (size_t)n * 4) ||
        !alloc(d_bases, ((size_t)n + 1) * 8) || !alloc(d_off, ((size_t)n + 1) * 8) || !alloc(d_ws, ws_bytes) || !alloc(d_wkeys, (size_t)wit_slots * 4) ||
        !alloc(d_wvals, (size_t)wit_slots * 4)) {
        (void)hipGetLastError();
        return FLX_OK;
    }
    flx_time_begin(ctx, "flx_kmerset_locus_build");
    FLX_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, (n_blocks + 1) * 8, st));
    hipLaunchKernelGGL(k_pt_count256, dim3((unsigned)(n_blocks / 256)), dim3(256), 0, st, present, d_cnt.as<int64_t>());
    FLX_CHECK(flx_exclusive_scan_i64(ctx, n_blocks + 1, d_cnt.as<int64_t>(), d_pre.as<int64_t>(), d_ws.p, ws_bytes));
    hipLaunchKernelGGL(k_pt_members, dim3((unsigned)(n_blocks / 256)), dim3(256), 0, st, present, d_pre.as<int64_t>(), d_members.as<uint32_t>());
    PtWitness wit;
    wit.keys = d_wkeys.as<uint32_t>();
    wit.vals = d_wvals.as<uint32_t>();
    wit.mask = wit_slots - 1;
    FLX_HIP(ctx, hipMemsetAsync(wit.keys, 0, (size_t)wit_slots * 4, st));
    FLX_HIP(ctx, hipMemsetAsync(wit.vals, 0, (size_t)wit_slots * 4, st));
    for (size_t b = 0; b < n_batches; ++b)
        if (batches[b].n_pos)
            hipLaunchKernelGGL(k_pt_witness, dim3((unsigned)((batches[b].n_pos + 255) / 256)), dim3(256), 0, st, batches[b].bases, batches[b].offsets,
                               batches[b].pos_base, batches[b].n_seqs, batches[b].n_pos, exact15, wit);
    hipLaunchKernelGGL(k_pt_pred, dim3(nb), dim3(256), 0, st, present, d_pre.as<int64_t>(), exact15, d_members.as<uint32_t>(), n, wit, d_link0.as<uint32_t>(),
                       d_dist0.as<uint32_t>());
    uint32_t *link = d_link0.as<uint32_t>(), *link2 = d_link1.as<uint32_t>(), *dist = d_dist0.as<uint32_t>(), *dist2 = d_dist1.as<uint32_t>();
    int rounds = 1;
    while ((1ull << rounds) < (uint64_t)n + 1) ++rounds;
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(k_pt_jump, dim3(nb), dim3(256), 0, st, n, link, dist, link2, dist2);
        std::swap(link, link2);
        std::swap(dist, dist2);
    }
    FLX_HIP(ctx, hipMemsetAsync(d_len.p, 0, (size_t)n * 4, st));
    hipLaunchKernelGGL(k_pt_heads, dim3(nb), dim3(256), 0, st, n, link, dist, d_len.as<uint32_t>());
    hipLaunchKernelGGL(k_pt_lengths, dim3(nb), dim3(256), 0, st, n, link, dist, d_len.as<uint32_t>());
    FLX_HIP(ctx, hipMemsetAsync(d_bases.p, 0, ((size_t)n + 1) * 8, st));
    hipLaunchKernelGGL(k_pt_piece_bases, dim3(nb), dim3(256), 0, st, n, link, d_len.as<uint32_t>(), d_bases.as<int64_t>());
    FLX_CHECK(flx_exclusive_scan_i64(ctx, (uint64_t)n + 1, d_bases.as<int64_t>(), d_off.as<int64_t>(), d_ws.p, ws_bytes));
    int64_t n_text = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&n_text, d_off.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    if (n_text <= 0 || (uint64_t)n_text > (1ull << 28)) {
        flx_time_end(ctx);
        return FLX_OK;
    }
    const uint64_t n_words = ((uint64_t)n_text + 15) / 16;
    const uint64_t n_alloc = n_words + kLocusPad + 68;
    int bits = 10;
    while ((1ull << bits) < (uint64_t)n * 5 / 2) ++bits;
    const uint64_t slots = 1ull << bits;
    uint32_t *text = nullptr, *seed = nullptr;
    flx_dbuf seen;
    const size_t plane = (size_t)1 << (26 - 3);
    if (hipMalloc((void **)&text, n_alloc * 8) != hipSuccess || hipMalloc((void **)&seed, slots * 4) != hipSuccess || !alloc(seen, 2 * plane)) {
        if (text) (void)hipFree(text);
        if (seed) (void)hipFree(seed);
        (void)hipGetLastError();
        flx_time_end(ctx);
        return FLX_OK;
    }
    FLX_HIP(ctx, hipMemsetAsync(text, 0, n_alloc * 8, st));
    std::vector<uint32_t> pad_front(2 * kLocusPad), pad_back(2 * 68);
    for (size_t i = 0; i < pad_front.size(); i += 2) { pad_front[i] = 0; pad_front[i + 1] = 0xffffu; }
    for (size_t i = 0; i < pad_back.size(); i += 2) { pad_back[i] = 0; pad_back[i + 1] = 0xffffu; }
    FLX_HIP(ctx, hipMemcpyAsync(text, pad_front.data(), pad_front.size() * 4, hipMemcpyHostToDevice, st));
    FLX_HIP(ctx, hipMemcpyAsync(text + 2 * (kLocusPad + n_words), pad_back.data(), pad_back.size() * 4, hipMemcpyHostToDevice, st));
    FLX_HIP(ctx, hipMemsetAsync(seed, 0xff, slots * 4, st));
    FLX_HIP(ctx, hipMemsetAsync(seen.p, 0, 2 * plane, st));
    hipLaunchKernelGGL(k_pt_text, dim3(nb), dim3(256), 0, st, n, d_members.as<uint32_t>(), link, dist, d_off.as<int64_t>(), text);
    if (n_text % 16)
        hipLaunchKernelGGL(k_pt_set_word_bits, dim3(1), dim3(1), 0, st, text + 2 * (kLocusPad + n_words - 1) + 1, 0xffffu & ~((1u << (n_text % 16)) - 1u));
    uint32_t *seen1 = seen.as<uint32_t>(), *seen2 = seen1 + plane / 4;
    hipLaunchKernelGGL(k_pt_u13<0>, dim3(nb), dim3(256), 0, st, n, d_members.as<uint32_t>(), link, dist, d_len.as<uint32_t>(), d_off.as<int64_t>(), text, seen1, seen2);
    hipLaunchKernelGGL(k_pt_u13<1>, dim3(nb), dim3(256), 0, st, n, d_members.as<uint32_t>(), link, dist, d_len.as<uint32_t>(), d_off.as<int64_t>(), text, seen1, seen2);
    hipLaunchKernelGGL(k_pt_seed, dim3(nb), dim3(256), 0, st, n, d_members.as<uint32_t>(), link, dist, d_off.as<int64_t>(), (const uint2 *)text, seed,
                       (uint32_t)(slots - 1), 32 - bits);
    flx_time_end(ctx);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(st));
    loc->text = (const uint2 *)text;
    loc->n_alloc = (uint32_t)n_alloc;
    loc->n_text = (uint64_t)n_text;
    loc->seed = seed;
    loc->seed_mask = (uint32_t)(slots - 1);
    loc->seed_shift = 32 - bits;
    *text_out = text;
    *seed_out = seed;
    return FLX_OK;
}
