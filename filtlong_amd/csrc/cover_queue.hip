// cover_queue.hip — the coverage kernel of k-mer mode for a set with a text (round 6): k_kmer_cover_q.
//
// Reference semantics: the k-mer branch of Read::Read, src/read.cpp:43-58 — a rolling 2-bit 16-mer, one set lookup per position,
// bases i-15..i marked on a hit; here: seq plane -> coverage bit plane + covered count / first / last covered base per read
// (src/read.cpp:75-84).  Same outputs, bit for bit, as k_kmer_cover_w (score_kmer.hip), which stays as the second implementation
// (FLX_KMER_COVER=w) and is what a set without a text runs.
//
// Why a second structure.  k_kmer_cover_w is bound by its vector instructions (0.67 per position, profiles/r05_kmer_issue_mix_c3.txt):
// every wave runs the prefilter rounds and the search for all 64 lanes of every span, although the text settles most of them.
// Counted on C3 (tools/exp/cover_stats.py, profiles/r06_cover_lane_stats.txt): of a span's 64 lanes 32.5 have any question left
// behind the text comparison, 17 take part in the second prefilter round, and a search step serves 5 lanes — in 92 % of the spans
// at least one lane needs each block, so no wave-uniform branch skips them.
//
// So the span loop is cut in two PHASES inside one kernel, with a queue in LDS between them:
//   phase A (every lane of every span): base codes, the text along the diagonal (known members, U13 / S1 refutations, 12-mers the text
//           vouches for), seeds.  A lane whose 17-position window leaves no candidate outside its confirmed members is done — its hits
//           are the known ones.  Every other lane APPENDS its piece to the wave's queue: the 32 bases it needs, what the text knows
//           about its 16 windows and about the 12-mers in front of it, whether the left neighbour's last 16-mer is a known member,
//           and its position.  An entry holds everything phase B needs: nothing there looks at a lane that is not in the queue.
//   phase B (when 64 entries have gathered, or the oldest waits too long): prefilter in two rounds over the TEN pairs that hold the
//           12-mers of the piece's own 16 windows (positions -4 .. 15; k_kmer_cover_w let the left neighbour fetch the first two
//           pairs), candidates, the outermost-member search against exact15 — the same policy as before, but all 64 lanes carry a
//           piece.  Pieces that are neighbours in the read are neighbours in the queue (the append keeps the order): where both are
//           queued the left one fetches the two pairs they share for both, and its answer still spares the bottom-up search.
//   The hits of a span wait in a ring in LDS (kRing spans) until its queued pieces are through; then the 4-step OR-dilation, the
//   counts and the row words as before.
// A global queue between two kernels (the form the round-5 review sketched) was costed first: 32.5 entries of 20 bytes per span are
// 1.3 kB of extra HBM traffic per 1 kB of plane, and the cover stage sits 15 % above the floor its requests set —
// the queue must not leave the chip.
//
// The kernel exists twice (template parameter INDELS).  <false> serves every read and hands over, by a mark, the reads whose diagonal
// moves by a base or three again and again — insertions and deletions —; <true> runs on the marked reads afterwards and lets every lane
// SEARCH a diagonal of its own (lane_diagonals below).  What that gave, what it did not, and the forms that were measured and
// rejected (a second queue for the search, the lane-diagonal code inside the one kernel): DESIGN.md §4.3, profiles/r06_microbench.txt.
#include "flx_internal.h"
#include "kmerset.h"
#include "cover_common.h"

namespace {

#ifndef FLX_COVER_RING
#define FLX_COVER_RING 8
#endif
#ifndef FLX_COVER_LAG
#define FLX_COVER_LAG 4
#endif
constexpr int kRing = FLX_COVER_RING;  // spans whose hits a wave keeps in LDS (a power of two)
constexpr int kQueue = 128;            // queue slots per wave (a power of two, >= 63 + 64)
constexpr int kLag = FLX_COVER_LAG;    // a queued piece is served at the latest when the wave is this many spans ahead of it (<= kRing - 3)
static_assert((kRing & (kRing - 1)) == 0 && kLag >= 1 && kLag <= kRing - 3, "ring / lag");

struct WaveLds {
    uint16_t ring[kRing][64];  // hits of span s at ring[s % kRing]: the known ones from phase A, replaced by phase B's for queued pieces
    uint32_t q[5][kQueue];     // hi, lo, known | refuted << 16, 12-mers the text vouches for (positions -4 .. 15), id | left known << 31
    // the text of the current span, staged for lanes that follow a diagonal of their own (reads with insertions / deletions, below):
    // words W0 - 3 .. W0 + 65 at slots 0 .. 68, W0 = lane 0's word on the wave's diagonal
    uint32_t tx[72], ty[72];
    uint16_t tss[72];
};

// everything the lanes of a wave wrote to LDS is visible to its other lanes (a wave's LDS operations execute in order; this keeps
// the compiler from moving accesses across and waits for the outstanding ones)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- reads with insertions and deletions: a diagonal per lane (phase A of k_kmer_cover_q, below) ---------------------------------
// Stages the span's text in LDS, searches every lane's own diagonal among the 33 shifts -16 .. +16 around the wave's, and compares
// the lane's 32-base window with the text on its own diagonal and on its left neighbour's.  Returns what that adds to the lane's
// known members / refutations / text-vouched 12-mers, the lane's shift and whether the search found one.
struct LaneDiag {
    uint32_t known_refuted;  // known | refuted << 16
    uint32_t text12;
    int dl;
    int matched;
};
typedef __attribute__((address_space(3))) WaveLds *LdsPtr;
__device__ __attribute__((noinline)) LaneDiag lane_diagonals(LdsPtr Sp, int lane, int e, uint32_t lo, uint32_t hi, uint32_t valid16, uint32_t twx,
                                                             uint32_t twy, uint32_t ts, uint32_t xwx, uint32_t xwy, uint32_t xs, int c_dl) {
    auto &S = *Sp;
    uint32_t known = 0, refuted = 0, text12 = 0;
    wave_lds_sync();
    S.tx[lane + 3] = twx;
    S.ty[lane + 3] = twy;
    S.tss[lane + 3] = (uint16_t)ts;
    if (lane < 5) {
        const int slot_x = lane < 3 ? lane : lane + 64;
        S.tx[slot_x] = xwx;
        S.ty[slot_x] = xwy;
        S.tss[slot_x] = (uint16_t)xs;
    }
    wave_lds_sync();
    // the search: the text's 48 bases around my 16, aligned to them
    int dl = 0;
    bool matched;
    {
        const uint32_t x0 = S.tx[lane + 1], x1 = S.tx[lane + 2], x2 = S.tx[lane + 3], x3 = S.tx[lane + 4];
        const uint32_t am = __builtin_amdgcn_alignbit(x0, x1, 2 * (15 - e));  // the 16 bases in front of mine on the wave's diagonal
        const uint32_t a0 = __builtin_amdgcn_alignbit(x1, x2, 2 * (15 - e));  // mine
        const uint32_t ap = __builtin_amdgcn_alignbit(x2, x3, 2 * (15 - e));  // the 16 behind
        uint32_t best = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 33; ++r) {  // shift 0 first, then -1, +1, -2, ..: the smallest key wins, ties go to the nearer shift
            const int sh = (r & 1) ? -((r + 1) >> 1) : (r >> 1);
            const uint32_t t = sh == -16 ? am : sh < 0 ? __builtin_amdgcn_alignbit(am, a0, -2 * sh) : sh == 0 ? a0 : __builtin_amdgcn_alignbit(a0, ap, 32 - 2 * sh);
            best = min(best, (uint32_t)__popc(lo ^ t) * 64u + (uint32_t)r);
        }
        // (6 of 32 bits: three or four substitutions; 16 random bases come that close at one of 33 shifts once in 80 lanes)
        matched = (valid16 >> 15) != 0 && (best >> 6) <= 6u;
        const int r = (int)(best & 63u);
        const int mine = matched ? ((r & 1) ? -((r + 1) >> 1) : (r >> 1)) : 0x7f;
        const int left = (int)flx_from_left((uint32_t)mine, (uint32_t)c_dl);
        // (a lane without a diagonal of its own looks where its left neighbour does.  Looking further left — up to four lanes — changes
        // nothing, counted: the 6.4 lanes per span that still find a text window with the exact table are pairs of neighbours with an
        // indel each, whose common stretch lies on a diagonal neither of them has most of its bases on)
        dl = matched ? mine : (left != 0x7f ? left : 0);
    }
    // my 32-base window against the text on the diagonal `diag + d`: adds to known / refuted / text12
    auto compare_at = [&](int d) {
        d = max(-16, min(16, d));
        const int ed = e + d;                 // -16 .. 31
        const int el = ed & 15;               // index of my last base in its word
        const int j2 = lane + 3 + (ed >> 4);  // the slot of that word
        const uint32_t w0x = S.tx[j2 - 2], w1x = S.tx[j2 - 1], w2x = S.tx[j2];
        const uint32_t w0y = S.ty[j2 - 2], w1y = S.ty[j2 - 1], w2y = S.ty[j2];
        const uint32_t w0s = S.tss[j2 - 2], w1s = S.tss[j2 - 1], w2s = S.tss[j2];
        const uint32_t t_own = __builtin_amdgcn_alignbit(w1x, w2x, 2 * (15 - el));
        const uint32_t t_hi = __builtin_amdgcn_alignbit(w0x, w1x, 2 * (15 - el));
        // per-base flags of the window's 32 bases: bit i = base i (base 0 = the first of the 16 in front of mine)
        auto flags32 = [&](uint32_t f0, uint32_t f1, uint32_t f2) -> uint32_t {
            const uint64_t f = (uint64_t)((f0 & 0xffffu) | (f1 << 16)) | ((uint64_t)(f2 & 0xffffu) << 32);
            return (uint32_t)(f >> (el + 1));
        };
        const uint32_t b32 = flags32(w0y, w1y, w2y);                    // base i is the first of a piece
        const uint32_t u32 = flags32(w0y >> 16, w1y >> 16, w2y >> 16);  // a unique 13-mer starts at base i
        const uint32_t s32 = flags32(w0s, w1s, w2s);                    // the text's 16 bases from base i on are S1
        auto mismatches = [](uint32_t x) -> uint32_t {  // bit j: base j of the 16 differs
            uint32_t m = (x | (x >> 1)) & 0x55555555u;
            m = (m | (m >> 1)) & 0x33333333u;
            m = (m | (m >> 2)) & 0x0f0f0f0fu;
            m = (m | (m >> 4)) & 0x00ff00ffu;
            m = (m | (m >> 8)) & 0xffffu;
            return __brev(m) >> 16;
        };
        const uint32_t z = ~(mismatches(hi ^ t_hi) | (mismatches(lo ^ t_own) << 16));  // bit i: base i of the window matches
        uint32_t r = z & (z >> 1);
        r &= r >> 2;
        r &= r >> 4;
        r &= r >> 8;  // bit i: bases i .. i + 15 match
        uint32_t q = ~b32 >> 1;  // bit i: no piece starts at base i + 1
        q &= q >> 1;
        q &= q >> 2;
        q &= q >> 4;
        q &= q >> 7;  // bit i: none at i + 1 .. i + 15 — the 16 bases from i on lie in one piece of the text
        {
            uint32_t m12 = z & (z >> 1);
            m12 &= m12 >> 2;
            m12 &= m12 >> 4;
            m12 &= m12 >> 4;  // bit i: bases i .. i + 11 match
            uint32_t q12 = ~b32 >> 1;
            q12 &= q12 >> 1;
            q12 &= q12 >> 2;
            q12 &= q12 >> 4;
            q12 &= q12 >> 3;  // bit i: no piece starts at i + 1 .. i + 11
            text12 |= ((m12 & q12) >> 5) & 0xffffu;
        }
        r &= q;
        known |= (r >> 1) & valid16;
        uint32_t g = z & (z >> 1);
        g &= g >> 2;
        g &= g >> 4;
        g &= g >> 5;  // bit i: bases i .. i + 12 match the text
        g &= u32;     // ... and that 13-mer occurs nowhere else
        g |= g >> 1;
        g |= g >> 2;
        uint32_t one = ~z, two;  // S1: exactly one of the 16 bases from i on differs (saturating two-bit counter per window)
        two = one & (one >> 1);
        one ^= one >> 1;
        {
            const uint32_t t2 = two | (two >> 2) | (one & (one >> 2));
            one = (one ^ (one >> 2)) & ~t2;
            two = t2;
        }
        {
            const uint32_t t2 = two | (two >> 4) | (one & (one >> 4));
            one = (one ^ (one >> 4)) & ~t2;
            two = t2;
        }
        {
            const uint32_t t2 = two | (two >> 8) | (one & (one >> 8));
            one = (one ^ (one >> 8)) & ~t2;
        }
        one &= q & s32;
        refuted |= (((g & ~r) | one) >> 1) & valid16;
    };
    compare_at(dl);
    const int dleft = (int)flx_from_left((uint32_t)dl, (uint32_t)c_dl);
    if (__any(dleft != dl)) compare_at(dleft);  // windows that begin in front of an indel lie on the left neighbour's diagonal
    LaneDiag out;
    out.known_refuted = known | (refuted << 16);
    out.text12 = text12;
    out.dl = dl;
    out.matched = matched ? 1 : 0;
    return out;
}

// INDELS = false: the kernel every read goes through.  A read whose diagonal jumps by a few bases — a seed found within 64 bases of
// the diagonal the wave already had: an insertion or a deletion, not junk (whose seeds fail) and not the same locus behind
// substitutions (whose seed finds the same diagonal) — is handed over: the wave drops it and appends it to a list (a.redo).  INDELS = true runs on that list afterwards and lets every lane follow a diagonal of its own (lane_diagonals
// above).  Two kernels of one source because the lane-diagonal code inside the span loop cost the loop 21 spilled vector registers
// and C3 a quarter of its speed (a cold call instead: a fifth — measured, profiles/r06_microbench.txt).
template <bool HAS_PREFILTER, bool INDELS>
__global__ void __launch_bounds__(FLX_COVER_THREADS) FLX_COVER_OCC k_kmer_cover_q(const CoverArgs a) {
    __shared__ WaveLds lds_all[FLX_COVER_THREADS / 64];
    WaveLds &S = lds_all[threadIdx.x >> 6];
    const uint8_t *plane = a.plane;
    const uint64_t *offsets = a.offsets;
    const int32_t *lengths = a.lengths;
    const uint32_t *order = a.order;
    const uint64_t n_reads = a.n_reads;
    uint32_t *cov = a.cov;
    const uint64_t *cov_off = a.cov_off;
    int32_t *count = a.count, *first = a.first, *last = a.last;
    const uint32_t loc_n_alloc = a.loc.n_alloc, loc_seed_mask = a.loc.seed_mask;
    const int loc_seed_shift = a.loc.seed_shift;
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    // INDELS: the wave takes 64 slots of the processing order at a time and serves the ones the first kernel has marked
    for (uint64_t slot_0 = wave0 * (INDELS ? 64 : 1); slot_0 < n_reads; slot_0 += n_waves * (INDELS ? 64 : 1)) {
      unsigned long long marked = 1ull;
      if (INDELS) marked = __ballot(slot_0 + (uint64_t)lane < n_reads && a.redo[slot_0 + (uint64_t)lane] != 0);
      for (; marked; marked &= marked - 1) {
        const uint64_t slot_r = slot_0 + (INDELS ? (uint64_t)(__ffsll((long long)marked) - 1) : 0ull);
        const uint32_t rid = __builtin_amdgcn_readfirstlane(order ? order[slot_r] : (uint32_t)slot_r);
        const int L = __builtin_amdgcn_readfirstlane(lengths[rid]);
        const uint8_t *seq = plane + offsets[rid];
        uint32_t *row = cov + (cov_off[rid] >> 2);
        const int row_words = (((L + 7) / 8 + 15) & ~15) >> 2;
        const int n_spans = (L + 1023) >> 10;
        int cnt = 0, fst = 0x7fffffff, lst = -1;
        // carried from lane 63 of the previous span (wave-uniform)
        uint32_t c_lo = 0, c_known15 = 0, c_t12 = 0;
        // the diagonal (wave-uniform), the text of this span / the next one, lane 63's carries of the text comparison
        long long diag = 0;
        bool have_diag = false, carry_ok = false;
        uint32_t c_mb = 0xffffffffu /* mismatches | piece starts << 16 of lane 63 */, c_us = 0 /* its U13 | S1 << 16 */, c_twx = 0, c_twy = 0xffffu;
        uint2 tw = make_uint2(0, 0xffffu), tw_next = make_uint2(0, 0xffffu);
        uint32_t ts = 0, ts_next = 0, c_ts = 0;  // S1 bits of those text words (kmerset.h: safe1); lane 63's for the next span
        const bool has_s1 = a.loc.safe1 != nullptr;
        // a read with insertions / deletions: every lane follows a diagonal of its own (below); c_dl = lane 63's shift against the
        // wave's diagonal as the next span sees it
        bool indel_mode = false;
        int c_dl = 0;
        bool handed_over = false;  // INDELS = false: a seed was found within 64 bases of the diagonal the wave had
        // the queue: q_head = slot of the oldest entry, q_n = entries; fin = the next span to turn into coverage
        int q_head = 0, q_n = 0, fin = 0;

        // the text word that holds the LAST base of the lane's 16 at this diagonal, for the lane whose 16 bases start at `base` +
        // 16 * lane (index clamped into the padded array).  The diagonal and `base` are wave-uniform: the 64-bit part of the index
        // is scalar work, a lane adds its number and clamps
        auto word_index = [&](long long dg, int base) -> uint32_t {
            long long u = ((dg + base + 15) >> 4) + (long long)kLocusPad;  // (16 * lane + c) >> 4 == lane + (c >> 4)
            u = u < -64 ? -64 : (u > (long long)loc_n_alloc ? (long long)loc_n_alloc : u);
            const int w = (int)u + lane;
            return (uint32_t)max(0, min(w, (int)loc_n_alloc - 1));
        };
        auto text_word = [&](long long dg, int base) -> uint2 {
            // (a 32-bit byte offset on a scalar base: one address register — the text has at most 2^28 positions, 2^27 bytes)
            const uint64_t tv = *(FLX_GLOBAL_PTR(uint64_t))(FLX_KARG_PTR(uint8_t, loc.text) + (uint32_t)(word_index(dg, base) * 8u));
            return make_uint2((uint32_t)tv, (uint32_t)(tv >> 32));
        };
        auto safe_word = [&](long long dg, int base) -> uint32_t {  // the S1 bits of that word
            return has_s1 ? (uint32_t)*(FLX_GLOBAL_PTR(uint16_t))(FLX_KARG_PTR(uint8_t, loc.safe1) + (uint32_t)(word_index(dg, base) * 2u)) : 0u;
        };

        // the five words around the span's 64 that lanes on a shifted diagonal reach into: lanes 0..2 fetch W0 - 3 .. W0 - 1, lanes 3 and 4
        // W0 + 64 and W0 + 65 (the others fetch a clamped word nobody uses)
        auto edge_index = [&](long long dg, int base) -> uint32_t {
            long long u = ((dg + base + 15) >> 4) + (long long)kLocusPad;
            u = u < -64 ? -64 : (u > (long long)loc_n_alloc ? (long long)loc_n_alloc : u);
            const int w = (int)u + (lane < 3 ? lane - 3 : lane + 61);
            return (uint32_t)max(0, min(w, (int)loc_n_alloc - 1));
        };

        // hits of span sp (complete in the ring) -> coverage bits, counts, row words
        auto finalize = [&](int sp) {
            const int p0 = (sp << 10) + lane * 16;
            const uint32_t h = S.ring[sp & (kRing - 1)][lane];
            const uint32_t right_of_63 = sp + 1 < n_spans ? (uint32_t)S.ring[(sp + 1) & (kRing - 1)][0] : 0u;
            const uint32_t next = flx_from_right(h, right_of_63);
            uint32_t x = h | (next << 16);
            x |= x >> 1;
            x |= x >> 2;
            x |= x >> 4;
            x |= x >> 8;  // bit j = OR of hit bits j .. j+15: base p0+j lies in a member 16-mer (src/read.cpp:53-54)
            uint32_t c16 = x & 0xffffu;
            if (((sp + 1) << 10) > L) {  // (wave-uniform: only the read's last span has positions to cut off)
                if (p0 >= L) c16 = 0;
                else if (p0 + 16 > L) c16 &= (1u << (L - p0)) - 1u;
            }
            cnt += __popc(c16);
            if (c16) {
                fst = min(fst, p0 + (__ffs(c16) - 1));
                lst = max(lst, p0 + (32 - __clz(c16)));
            }
            const uint32_t up = flx_from_right(c16, 0u);  // (only the even lanes write: lane 63's is never used)
            const int word = p0 >> 5;
            if ((lane & 1) == 0 && word < row_words) __builtin_nontemporal_store(c16 | (up << 16), &row[(uint32_t)word]);
        };

        // ---- phase B: the first n (<= 64) entries of the queue, one per lane ----
        auto serve = [&](int n) {
            wave_lds_sync();
            const bool act = lane < n;
            const uint32_t qs = (uint32_t)(q_head + lane) & (kQueue - 1);
            uint32_t hi = 0, lo = 0, kr = 0, t20 = 0, idw = 0x7fffffffu;
            if (act) {
                hi = S.q[0][qs];
                lo = S.q[1][qs];
                kr = S.q[2][qs];
                t20 = S.q[3][qs];
                idw = S.q[4][qs];
            }
            const uint32_t known = kr & 0xffffu, refuted = kr >> 16;
            const uint32_t id = idw & 0x7fffffffu, lk = idw >> 31;  // piece number in the read (16 positions each); left neighbour's last 16-mer known
            // positions p0 + j that end a 12-mer / a 16-mer inside the read (all of them except in the read's first and last piece)
            uint32_t valid16 = act ? 0xffffu : 0u, valid12 = valid16;
            if (__any(act && (id == 0 || (int)(id << 4) + 16 > L))) {
                const int p0 = (int)(id << 4);
                if (act) {
                    if (p0 < 11) valid12 &= ~((1u << (11 - p0)) - 1u);
                    if (p0 < 15) valid16 &= ~((1u << (15 - p0)) - 1u);
                    if (p0 + 16 > L) {
                        valid12 &= (1u << (L - p0)) - 1u;
                        valid16 &= (1u << (L - p0)) - 1u;
                    }
                }
            }
            // ---- 12-mer prefilter over the 20 positions -4 .. 15 (bit i <-> position i - 4): pair k = positions 2k - 4, 2k - 3;
            // x.C.y = the 13 bases ending at position 2k - 3.  Present without a lookup: what the text vouches for, and the 12-mers
            // ending at [lowest known member - 4, highest]: those inside a member are present, the others only make candidates
            // between two confirmed members, which are never asked. ----
            const bool adj = act && flx_from_left(id, 0x7ffffff0u) + 1u == id;  // the entry in front of this one is its left neighbour in the read
            const uint32_t V20 = (valid12 << 4) | ((act && id != 0) ? 0xFu : 0u);
            uint32_t P = 0xFFFFFu;
            if (HAS_PREFILTER) {
                uint32_t need20 = V20 & ~t20;
                if (known) need20 &= ~(((32u << (31 - __clz(known))) - 1u) & ~((1u << (__ffs(known) - 1)) - 1u));
                // In TWO rounds: a 16-mer is out as soon as ONE of its five 12-mers is absent, and what is left to look up holds a
                // mismatch against the text, so it is absent more often than not.  Round 1 fetches the even pairs where needed; every
                // 16-mer holds two or three of their positions, so most are out after it.  Round 2 fetches an odd pair only if one of
                // the 16-mers that hold its 12-mers is still alive under the assumption that every 12-mer not yet seen is present.
                // (the reverse complement of the whole 32-base window once: the canonical form of every pair's 11-mer is then one
                // funnel shift, and a byte read for the other strand is looked at bit-reversed — kmerset.h, flx_pre11)
                auto rc32 = [](uint32_t w) {
                    const uint32_t r = __brev(w);
                    return ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));
                };
                const uint64_t hl64 = ((uint64_t)hi << 32) | lo;
                const uint64_t r64 = ((uint64_t)rc32(lo) << 32) | rc32(hi);
                auto fetch = [&](uint32_t want, int parity) -> uint32_t {  // actual bits of the pairs k = parity, parity + 2, .. that hold a wanted position; 1 elsewhere
                    uint32_t byte[5], got = parity ? 0x33333u : 0xCCCCCu;
                    FLX_GLOBAL_PTR(uint8_t) pre11 = FLX_KARG_PTR(uint8_t, pre11);
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int k = 2 * j + parity;
                        const uint32_t x13 = (uint32_t)(hl64 >> (36 - 4 * k));  // x.C.y, the 13 bases ending at position 2k - 3
                        const uint32_t c = (x13 >> 2) & 0x3FFFFFu, rc = (uint32_t)(r64 >> (4 + 4 * k)) & 0x3FFFFFu;
                        const uint32_t kk = (x13 & 0x2000u) ? rc : c;  // the middle base of C is G or T: the byte belongs to the other strand
                        const uint32_t index = ((kk >> 12) << 11) | (kk & 0x7FFu);
                        byte[j] = ((want >> (2 * k)) & 3u) ? pre11[index] : 0xffu;
                    }
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int k = 2 * j + parity;
                        const uint32_t x13 = (uint32_t)(hl64 >> (36 - 4 * k));
                        const uint32_t b = (x13 & 0x2000u) ? (__brev(byte[j]) >> 24) : byte[j];
                        const uint32_t two = ((b >> ((x13 >> 24) & 3u)) & 1u) | (((b >> (4u + (x13 & 3u))) & 1u) << 1);
                        got |= two << (2 * k);  // (a pair that was not fetched holds 0xff: both present)
                    }
                    return got;
                };
                auto dep5 = [](uint32_t alive16) -> uint32_t {  // bit i: one of the windows ending at positions i - 4 .. i holds ... i.e. the 12-mer ending at position i - 4 lies in an alive window
                    const uint32_t x = alive16 << 4;
                    uint32_t d = x | (x >> 1);
                    d |= d >> 2;
                    return d | (x >> 4);
                };
                // Where the piece in front of this one is the entry in front of it (`adj`), the positions -4 .. -1 are that piece's 12 .. 15:
                // it fetches them (what this one wants of them travels left, the bits come back), so that a pair shared by two queued
                // neighbours is looked up once (measured: a tenth of the kernel's L2 requests were such doubles)
                const uint32_t radj = flx_from_right(adj ? 1u : 0u, 0u);  // the entry behind this one is its right neighbour in the read
                {
                    // round 1 leaves out the pairs whose 12-mers only lie in 16-mers the text has refuted (U13, S1)
                    const uint32_t alive = valid16 & (~refuted | known);
                    const uint32_t w1 = need20 & 0x33333u & dep5(alive);
                    const uint32_t from_r = flx_from_right(w1 & 0x3u, 0u);
                    const uint32_t mine = (adj ? (w1 & ~0x3u) : w1) | (radj ? (from_r << 16) : 0u);
                    if (__any(mine != 0)) P = fetch(mine, 0);
                    const uint32_t from_l = flx_from_left(P >> 16, 0xFu);
                    if (adj) P = (P & ~0x3u) | (from_l & 0x3u);
                }
                if (__any((need20 & 0xCCCCCu) != 0)) {
                    const uint32_t v1 = P & V20;
                    uint32_t alive = v1 & (v1 >> 1);
                    alive &= alive >> 2;
                    alive &= v1 >> 4;  // bit j: the five 12-mers of the window ending at position j are present so far
                    alive &= valid16 & ~refuted;
                    const uint32_t w2 = need20 & 0xCCCCCu & dep5(alive);
                    const uint32_t from_r = flx_from_right(w2 & 0xCu, 0u);
                    const uint32_t mine = (adj ? (w2 & ~0xCu) : w2) | (radj ? (from_r << 16) : 0u);
                    if (__any(mine != 0)) P &= fetch(mine, 1);
                    const uint32_t from_l = flx_from_left(P >> 16, 0xFu);
                    if (adj) P = (P & ~0xCu) | (from_l & 0xCu);
                }
            }
            P &= V20;
            uint32_t cand = P & (P >> 1);
            cand &= cand >> 2;
            cand &= P >> 4;  // all five 12-mers present
            cand &= valid16 & (~refuted | known);  // (a member known on one diagonal cannot be refuted on another — its 13-mers then occur twice in the text — but nothing is lost by saying so)

            // ---- exact membership: one byte of exact15 answers the pair of positions (a, a + 1), any a in 0..14 — the 15 bases
            // ending at a are the byte's index, the base before them picks the bit of position a, the base after them the bit of
            // a + 1.  A question from ABOVE (top-down search) takes the pair that ENDS at the asked position, one from below the
            // pair that starts there: either way the request also settles the next candidate in the direction of the search. ----
            uint32_t hits = known & cand, probed = known | (~cand & 0xffffu);
            auto probe = [&](int top, int bot) {  // positions asked from above / from below, -1 = none
                const int a0 = top > 0 ? top - 1 : 0, a1 = bot < 14 ? bot : 14;
                uint32_t g0 = 0, g1 = 0;
                FLX_GLOBAL_PTR(uint8_t) exact15 = FLX_KARG_PTR(uint8_t, exact15);
                if (top >= 0) g0 = exact15[__builtin_amdgcn_alignbit(hi, lo, 30 - 2 * a0) & 0x3FFFFFFFu];
                if (bot >= 0) g1 = exact15[__builtin_amdgcn_alignbit(hi, lo, 30 - 2 * a1) & 0x3FFFFFFFu];
                if (top >= 0) {
                    const uint32_t x = (hi >> (28 - 2 * a0)) & 3u, y = (lo >> (28 - 2 * a0)) & 3u;
                    hits |= (((g0 >> x) & 1u) | (((g0 >> (4 + y)) & 1u) << 1)) << a0;
                    probed |= 3u << a0;
                }
                if (bot >= 0) {
                    const uint32_t x = (hi >> (28 - 2 * a1)) & 3u, y = (lo >> (28 - 2 * a1)) & 3u;
                    hits |= (((g1 >> x) & 1u) | (((g1 >> (4 + y)) & 1u) << 1)) << a1;
                    probed |= 3u << a1;
                }
                hits &= cand;  // positions outside the read hold no 16-mer
            };
            // One step of the search in the piece's window of 17 positions (bit 0 = the left neighbour's last position, bit j + 1 =
            // position j): the highest open candidate above the confirmed members and the lowest one below them.
            auto next_asks = [&](uint32_t left_member, int &top, int &bot) -> bool {
                const uint32_t H = (hits << 1) | left_member;
                const uint32_t open = (cand & ~probed) << 1;
                uint32_t above = open, below = open;
                if (H) {
                    above = open & ~((2u << (31 - __clz(H))) - 1u);
                    below = open & ((H & (0u - H)) - 1u);
                }
                top = above ? 30 - __clz(above) : -1;  // position = bit - 1
                bot = below ? __ffs(below) - 2 : -1;
                if (bot >= 0 && bot + 1 >= top && top >= 0) bot = -1;  // the two questions meet: the pair that ends at `top` answers both
                return (top & bot) != -1;
            };
            // the left neighbour's last position: a known member (exact, from phase A), or — where the left neighbour is the entry in
            // front of this one — what its own search finds.  First step on a BET: a candidate there is taken for a member (it is the
            // top of that piece's search, so its answer arrives with this round's), corrected right after.
            int top, bot;
            {
                const uint32_t lcand = flx_from_left(cand >> 15, 0u);
                const bool need = next_asks(lk | (adj ? lcand : 0u), top, bot);
                if (__any(need)) probe(top, bot);
            }
            const uint32_t lh = flx_from_left(hits >> 15, 0u);
            const uint32_t lhit = lk | (adj ? lh : 0u);
            for (;;) {
                const bool need = next_asks(lhit, top, bot);
                if (!__any(need)) break;
                probe(top, bot);
            }
            if (act) S.ring[(id >> 6) & (kRing - 1)][id & 63u] = (uint16_t)hits;
            q_head = (q_head + n) & (kQueue - 1);
            q_n -= n;
            wave_lds_sync();
        };

        uint4 raw = make_uint4(0, 0, 0, 0);
        // (offsets as unsigned 32-bit values: a uniform base plus a 32-bit lane offset is one address register, not two)
        if (lane * 16 < L) raw = flx_plane16(seq + (uint32_t)(lane * 16));  // rows are 16-byte aligned and padded
        for (int sp = 0; sp < n_spans; ++sp) {
            const int p0 = (sp << 10) + lane * 16;
            uint4 raw_next = make_uint4(0, 0, 0, 0);
            if (p0 + 1024 < L) raw_next = flx_plane16(seq + (uint32_t)(p0 + 1024));
            if (have_diag && !indel_mode && sp + 1 < n_spans) {  // (indel mode: the diagonal moves at the end of the span, the words are fetched there)
                tw_next = text_word(diag, (sp << 10) + 1024);
                ts_next = safe_word(diag, (sp << 10) + 1024);
            }
            // 2 bits per base, earliest base on top: lo = my 16 bases, hi = the 16 before them
            const uint32_t lo = (codes4(raw.x) << 24) | (codes4(raw.y) << 16) | (codes4(raw.z) << 8) | codes4(raw.w);
            const uint32_t hi = flx_from_left(lo, c_lo);
            // positions p0 + j that end a 16-mer inside the read
            uint32_t valid16 = 0;
            if (sp > 0 && ((sp + 1) << 10) <= L) {  // (wave-uniform: a span inside the read has every position, no lane computes masks)
                valid16 = 0xffffu;
            } else if (p0 < L) {
                valid16 = 0xffffu;
                if (p0 < 15) valid16 &= ~((1u << (15 - p0)) - 1u);
                if (p0 + 16 > L) valid16 &= (1u << (L - p0)) - 1u;
            }
            // ---- members known from the text along the diagonal ----
            uint32_t known = 0, refuted = 0;  // refuted: not a text match, but holds a text-matching 13-mer that occurs nowhere else (U13), or is one substitution away from a text window without such members (S1)
            uint32_t text12 = 0;              // bit j: the 12 bases ending at my position j match the text inside one piece: that 12-mer IS present
            {
                int mm_cnt = 0;        // my bases that differ from the text on the wave's diagonal (last comparison)
                uint32_t t_diag = 0;   // the text's 16 bases under mine on that diagonal
                // my 16 bases against the text along `diag` (tw = the word that holds the last of them): adds to known / refuted
                auto compare = [&]() {
                    const int e = (int)((diag + 15) & 15);  // index of my last base in my word (p0 is a multiple of 16: the same for every lane)
                    // lane 0's left word: the carry, or behind a new seed a load (wave-uniform choice; only lane 0's copy is used)
                    const uint2 tw0 = carry_ok ? make_uint2(c_twx, c_twy) : text_word(diag, (sp << 10) - 16);
                    uint2 twl;
                    twl.x = flx_from_left(tw.x, tw0.x);
                    twl.y = flx_from_left(tw.y, tw0.y);
                    const uint32_t tsl = flx_from_left(ts, carry_ok ? c_ts : 0u);  // (not worth a load: lane 0 behind a new seed refutes its own window only, below)
                    const uint32_t t_own = __builtin_amdgcn_alignbit(twl.x, tw.x, 2 * (15 - e));
                    // (piece starts are rare — two per contig — and a span whose text words hold none, nor the carried word of the lane in
                    // front, needs none of the masks that keep a window inside one piece: 25 of the comparison's ~100 instructions)
                    const uint32_t mb0 = carry_ok ? c_mb : 0xffffffffu, us0 = carry_ok ? c_us : 0u;
                    const bool any_start = (mb0 >> 16) != 0 || __any(((tw.y | twl.y) & 0xffffu) != 0);
                    uint32_t b_own = 0;  // bit j: my base j is the first of a piece
                    if (any_start) b_own = (((twl.y & 0xffffu) >> (e + 1)) | (tw.y << (15 - e))) & 0xffffu;
                    const uint32_t u_own = (((twl.y >> 16) >> (e + 1)) | ((tw.y >> 16) << (15 - e))) & 0xffffu;  // bit j: a unique 13-mer starts at my base j
                    const uint32_t s_own = ((tsl >> (e + 1)) | (ts << (15 - e))) & 0xffffu;  // bit j: the text's 16 bases from my base j on are S1
                    t_diag = t_own;
                    const uint32_t x = lo ^ t_own;
                    uint32_t m = (x | (x >> 1)) & 0x55555555u;  // even bit 2k: the base k places from the END differs
                    m = (m | (m >> 1)) & 0x33333333u;
                    m = (m | (m >> 2)) & 0x0f0f0f0fu;
                    m = (m | (m >> 4)) & 0x00ff00ffu;
                    m = (m | (m >> 8)) & 0xffffu;
                    const uint32_t mml = __brev(m) >> 16;  // bit j: my base j differs from the text
                    mm_cnt = __popc(mml);
                    const uint32_t mmh = flx_from_left(mml, mb0 & 0xffffu);
                    uint32_t bh = 0;
                    if (any_start) bh = flx_from_left(b_own, mb0 >> 16);
                    const uint32_t uh = flx_from_left(u_own, us0 & 0xffffu), sh = flx_from_left(s_own, us0 >> 16);
                    const uint32_t z = ~(mmh | (mml << 16));  // bit i: base i of the window [p0 - 16, p0 + 16) matches
                    uint32_t r = z & (z >> 1);
                    r &= r >> 2;
                    r &= r >> 4;
                    r &= r >> 8;  // bit i: bases i .. i + 15 match
                    uint32_t q = 0xffffffffu, q12 = 0xffffffffu;  // (no piece start in sight: every window lies inside one piece)
                    if (any_start) {
                        q = ~(bh | (b_own << 16)) >> 1;  // bit i: no piece starts at base i + 1
                        q &= q >> 1;
                        q &= q >> 2;
                        q &= q >> 4;
                        q &= q >> 7;  // bit i: none at i + 1 .. i + 15 — the 16 bases from i on lie in one piece of the text
                        q12 = ~(bh | (b_own << 16)) >> 1;
                        q12 &= q12 >> 1;
                        q12 &= q12 >> 2;
                        q12 &= q12 >> 4;
                        q12 &= q12 >> 3;  // bit i: no piece starts at i + 1 .. i + 11 (a piece has at least 16 bases: the 12-mer lies in one of its 16-mers)
                    }
                    {
                        uint32_t m12 = z & (z >> 1);
                        m12 &= m12 >> 2;
                        m12 &= m12 >> 4;
                        m12 &= m12 >> 4;  // bit i: bases i .. i + 11 match
                        text12 |= ((m12 & q12) >> 5) & 0xffffu;  // the 12-mer ending at my position j starts at base j + 5
                    }
                    r &= q;
                    known |= (r >> 1) & valid16;  // the 16-mer ending at my position j starts at base j + 1 of the window
                    uint32_t g = z & (z >> 1);
                    g &= g >> 2;
                    g &= g >> 4;
                    g &= g >> 5;  // bit i: bases i .. i + 12 match the text
                    g &= uh | (u_own << 16);  // ... and that 13-mer occurs nowhere else (U13 is only set inside one piece)
                    g |= g >> 1;
                    g |= g >> 2;  // bit i: such a 13-mer starts at base i, i + 1, i + 2 or i + 3: inside the 16 bases from i on
                    // S1: exactly ONE of the 16 bases from i on differs from the text, and no 16-mer one base away from the text's is a
                    // member (counted with a saturating two-bit counter per window: `one` = exactly one mismatch, `two` = more)
                    uint32_t one = ~z, two;
                    two = one & (one >> 1);
                    one ^= one >> 1;
                    {
                        const uint32_t t2 = two | (two >> 2) | (one & (one >> 2));
                        one = (one ^ (one >> 2)) & ~t2;
                        two = t2;
                    }
                    {
                        const uint32_t t2 = two | (two >> 4) | (one & (one >> 4));
                        one = (one ^ (one >> 4)) & ~t2;
                        two = t2;
                    }
                    {
                        const uint32_t t2 = two | (two >> 8) | (one & (one >> 8));
                        one = (one ^ (one >> 8)) & ~t2;
                    }
                    one &= q & (sh | (s_own << 16));
                    uint32_t rf = (((g & ~r) | one) >> 1) & valid16;
                    // (lane 0 behind a new seed knows nothing about the 16 bases in front of it — taken for mismatches above, which is
                    // safe for `known` and would be wrong here: only the window made of its own 16 bases can be refuted)
                    if (lane == 0 && !carry_ok) rf &= 0x8000u;
                    refuted |= rf;
                    c_us = __builtin_amdgcn_readlane(u_own | (s_own << 16), 63);
                    c_ts = __builtin_amdgcn_readlane(ts, 63);
                    c_mb = __builtin_amdgcn_readlane(mml | (b_own << 16), 63);
                    c_twx = __builtin_amdgcn_readlane(tw.x, 63);
                    c_twy = __builtin_amdgcn_readlane(tw.y, 63);
                    carry_ok = true;
                };
                // The carried diagonal is tested for nothing.  Then, while at least three lanes behind the last lane with a known
                // member hold 16-mers nothing is known about (junk, an indel, the end of a piece of the text, the wrong copy of a
                // repeat), two of them look their own 16 bases up in the seed table — eight lanes spread over the span when nothing
                // is known at all; a seed on another diagonal is compared in turn, what it confirms adds to what is known.
                const unsigned long long whole = __ballot((valid16 >> 15) != 0);  // lanes that hold a whole 16-mer of the read
                bool again = have_diag;
                for (int seeds_left = FLX_LOCUS_SEEDS; !indel_mode;) {
                    if (again) compare();
                    // (a lane is ON the diagonal when it holds a known member — or when its 16 bases differ from the text in fewer than 6
                    // places: substitutions, not a wrong diagonal.  Round 5 looked at known members only, and a read with 10 % substitutions
                    // re-seeded in every span to find the diagonal it already had: a third of the kernel's seed lookups)
                    const unsigned long long kn = __ballot(known != 0 || (again && mm_cnt < 6 && (valid16 >> 15) != 0));
                    const unsigned long long tail = kn ? whole & ~((2ull << (63 - __clzll(kn))) - 1ull) : whole;
                    if (seeds_left-- == 0 || __popcll(tail) < FLX_LOCUS_TAIL) break;
                    // (a read that has nothing to do with the text — a contaminant — fails every attempt: a read that has found no
                    // diagonal in its first two spans tries again only in every fourth.  Ten seed lanes per span were 40 % of such a
                    // read's far requests.  No counter of failures: one more scalar carried through the span loop cost C3 2 %, measured)
                    if (!have_diag && sp >= 2 && (sp & 3) != 0) break;
                    bool tries;
                    if (kn) {
                        const unsigned long long t1 = tail & (tail - 1), t2 = t1 & (t1 - 1);  // without its first lane / first two lanes
                        tries = lane == __ffsll(t1) - 1 || lane == __ffsll(t2) - 1;
                    } else {
                        // (two lanes at the first attempt, eight from the second on: half the reads' first seeds are exact)
                        tries = (lane & (seeds_left == FLX_LOCUS_SEEDS - 1 ? 31 : 7)) == 3 && ((whole >> lane) & 1ull);
                    }
                    uint32_t tpos = kLocusEmpty;
                    if (tries) {
                        uint32_t h = flx_locus_hash(lo, loc_seed_shift);
                        FLX_GLOBAL_PTR(uint32_t) seed_tab = FLX_KARG_PTR(uint32_t, loc.seed);
                        FLX_GLOBAL_PTR(uint32_t) seed_text = FLX_KARG_PTR(uint32_t, loc.text);  // (.x of text word i at dword 2 i)
#pragma unroll 1
                        for (int probe_no = 0; probe_no < 4; ++probe_no) {
                            const uint32_t v = seed_tab[h];
                            if (v == kLocusEmpty) break;
                            {  // (flx_locus_kmer_at, kmerset.h, on the global-space pointer)
                                const uint32_t tw_i = (v >> 4) + kLocusPad, ts_i = v & 15u;
                                const uint32_t t0 = seed_text[2 * tw_i];
                                const uint32_t at = ts_i == 0 ? t0 : __builtin_amdgcn_alignbit(t0, seed_text[2 * tw_i + 2], 32 - 2 * ts_i);
                                if (at == lo) { tpos = v; break; }
                            }
                            h = (h + 1) & loc_seed_mask;
                        }
                    }
                    const unsigned long long found = __ballot(tpos != kLocusEmpty);
                    if (!found) {
                        if (!kn && seeds_left == FLX_LOCUS_SEEDS - 1) {  // the two first lanes held no exact 16-mer: eight more
                            again = false;
                            continue;
                        }
                        break;
                    }
                    const int src = __ffsll(found) - 1;
                    const long long nd = (long long)__builtin_amdgcn_readlane(tpos, src) - (long long)((sp << 10) + src * 16);
                    if (have_diag && nd == diag) break;  // the same locus: what is missing are mismatches, not the diagonal
                    if (!INDELS && have_diag && nd - diag >= -64 && nd - diag <= 64) {
                        handed_over = true;  // the diagonal has moved by a few bases: an insertion or a deletion, more will follow
                        break;
                    }
                    diag = nd;
                    have_diag = true;
                    carry_ok = false;
                    tw = text_word(diag, sp << 10);
                    ts = safe_word(diag, sp << 10);
                    if (sp + 1 < n_spans) {
                        tw_next = text_word(diag, (sp << 10) + 1024);
                        ts_next = safe_word(diag, (sp << 10) + 1024);
                    }
                    again = true;
                }

                // INDELS = false: is this a read whose diagonal has just moved by a base or three?  Lanes that look nothing like the text
                // on the wave's diagonal but like it one, two or three bases beside it (<= 4 differing bits of 32; the neighbours' aligned
                // text by DPP) say so — without waiting for a seed to confirm it, which half the time takes a span of prefilter and exact
                // table for nothing (a sixth of such a read's time went there).  Junk matches at no shift.
                if (!INDELS && have_diag && !handed_over && __popcll(__ballot((valid16 >> 15) != 0 && mm_cnt >= 6)) >= 2) {
                    const uint32_t tl = flx_from_left(t_diag, 0u), tr = flx_from_right(t_diag, 0u);
                    uint32_t best = 32;
#pragma unroll
                    for (int k = 1; k <= 3; ++k) {
                        best = min(best, (uint32_t)__popc(lo ^ __builtin_amdgcn_alignbit(tl, t_diag, 2 * k)));
                        best = min(best, (uint32_t)__popc(lo ^ __builtin_amdgcn_alignbit(t_diag, tr, 32 - 2 * k)));
                    }
                    if (__popcll(__ballot((valid16 >> 15) != 0 && mm_cnt >= 6 && best <= 4u)) >= 2) handed_over = true;
                }

                // ---- reads with insertions and deletions: a diagonal per lane (round 6; the round-5 review's item 2).  One diagonal
                // per wave, re-seeded four times per span, follows substitutions and the odd junk block; a nanopore read has an indel
                // every 20-50 bases, each one moves the diagonal by a base or three, and behind the first of a span everything went
                // through the prefilter and the exact table (3x the time, measured: profiles/r06_microbench.txt).  So where lanes that
                // hold a whole 16-mer look nothing like the text on the wave's diagonal (>= 6 of 16 bases differ: a wrong diagonal, not
                // substitutions), every lane SEARCHES its own: its 16 bases against the text at the 33 shifts -16 .. +16 around the
                // wave's diagonal (three funnel shifts align the four words around it, then one funnel shift, one XOR and one bit
                // count per shift: no request), takes the closest (<= 8 differing bits) and compares its 32-base window with the text
                // there and on its left neighbour's diagonal — known members, U13, S1 and text-vouched 12-mers exactly as on the wave's
                // diagonal: every one of those statements is about the TEXT at that place and true whatever the read's real locus is.
                // The text words come out of LDS (staged once per span) because a lane's window then starts in any of five words.  The
                // wave's diagonal moves on with the last lane that found one. ----
                const bool was_indel_mode = indel_mode;
                bool lane_diag = INDELS && indel_mode;
                if (INDELS && !indel_mode && have_diag) lane_diag = __popcll(__ballot(((valid16 >> 15) != 0) && mm_cnt >= 6)) >= 4;
                if (INDELS && lane_diag && have_diag) {
                    const int e = (int)((diag + 15) & 15);
                    const uint32_t xi = edge_index(diag, sp << 10);
                    uint2 xw = make_uint2(0, 0xffffu);
                    uint32_t xs = 0;
                    if (lane < 5) {
                        const uint64_t tv = *(FLX_GLOBAL_PTR(uint64_t))(FLX_KARG_PTR(uint8_t, loc.text) + (uint32_t)(xi * 8u));
                        xw = make_uint2((uint32_t)tv, (uint32_t)(tv >> 32));
                        if (has_s1) xs = (uint32_t)*(FLX_GLOBAL_PTR(uint16_t))(FLX_KARG_PTR(uint8_t, loc.safe1) + (uint32_t)(xi * 2u));
                    }
                    // (a function of its own, not inlined: inside the span loop its registers cost the loop 21 spilled vector registers
                    // and C3 a quarter of its speed — measured; the call is on the cold side of a wave-uniform branch)
                    const LaneDiag ld = lane_diagonals((LdsPtr)&S, lane, e, lo, hi, valid16, tw.x, tw.y, ts, xw.x, xw.y, xs, c_dl);
                    known |= ld.known_refuted & 0xffffu;
                    refuted |= ld.known_refuted >> 16;
                    text12 |= ld.text12;
                    const int dl = ld.dl;
                    const bool matched = ld.matched != 0;
                    // the wave's diagonal follows the last lane that found one; a span in which (next to) none did goes back to the seeds
                    const unsigned long long got = __ballot(matched);
                    const bool shifted = __popcll(__ballot(matched && dl != 0)) >= 4;
                    if (!indel_mode && shifted) indel_mode = true;
                    if (indel_mode && __popcll(got) < 4) indel_mode = false;
                    int step = 0;
                    if (indel_mode && got) step = (int)__builtin_amdgcn_readlane((uint32_t)dl, 63 - __clzll(got));
                    c_dl = max(-16, min(16, (int)__builtin_amdgcn_readlane((uint32_t)dl, 63) - step));
                    carry_ok = false;  // (the carries of the wave-diagonal comparison are not kept up here)
                    if (was_indel_mode || indel_mode) {  // (the span's own prefetch was left out, or the diagonal has moved)
                        diag += step;
                        if (sp + 1 < n_spans) {
                            tw_next = text_word(diag, (sp << 10) + 1024);
                            ts_next = safe_word(diag, (sp << 10) + 1024);
                        }
                    }
                } else {
                    c_dl = 0;
                }
            }

            if (!INDELS && handed_over) break;  // (wave-uniform: the read goes to the other kernel, whatever was written for it is written again)

            // ---- phase A ends: the hits the text knows, and the pieces that have a question left ----
            // The piece's window of 17 positions (bit 0 = the left neighbour's last position): a candidate is only ever asked when it
            // lies above the highest or below the lowest confirmed member (two members inside the window are at most 16 apart: between
            // them every base is covered).  No possible member outside the known ones' span: nothing to ask, the hits are the known ones.
            const uint32_t lk = flx_from_left(known >> 15, c_known15);
            const uint32_t t20 = (text12 << 4) | flx_from_left(text12 >> 12, c_t12);
            bool need;
            {
                const uint32_t open = (valid16 & ~refuted & ~known) << 1;
                const uint32_t H = (known << 1) | lk;
                uint32_t outside = open;
                if (H) outside = open & (~((2u << (31 - __clz(H))) - 1u) | ((H & (0u - H)) - 1u));
                need = outside != 0;
            }
            S.ring[sp & (kRing - 1)][lane] = (uint16_t)known;
            const unsigned long long nb = __ballot(need);
            if (nb) {
                if (need) {
                    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(nb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nb, 0u));
                    const uint32_t qs = (uint32_t)(q_head + q_n + (int)at) & (kQueue - 1);
                    S.q[0][qs] = hi;
                    S.q[1][qs] = lo;
                    S.q[2][qs] = known | (refuted << 16);
                    S.q[3][qs] = t20;
                    S.q[4][qs] = (uint32_t)((sp << 6) + lane) | (lk << 31);
                }
                q_n += __popcll(nb);
            }
            c_lo = __builtin_amdgcn_readlane(lo, 63);
            c_known15 = __builtin_amdgcn_readlane(known >> 15, 63);
            c_t12 = __builtin_amdgcn_readlane(text12 >> 12, 63);
            raw = raw_next;
            tw = tw_next;
            ts = ts_next;

            // ---- phase B where it is due: full batches, and whatever waits while the wave has moved kLag spans on ----
            while (q_n >= 64) serve(64);
            if (q_n > 0) {
                wave_lds_sync();
                const int head_span = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)S.q[4][q_head]) & 0x7fffffffu) >> 6;
                if (head_span + kLag <= sp) serve(q_n);
            }
            // ---- coverage of the spans whose pieces are through and whose right neighbour's first piece is ----
            {
                int done = sp + 1;  // the first span that still has a piece in the queue
                if (q_n > 0) {
                    wave_lds_sync();
                    done = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)S.q[4][q_head]) & 0x7fffffffu) >> 6;
                } else {
                    wave_lds_sync();
                }
                while (fin + 1 < done) finalize(fin++);
            }
        }
        if (!INDELS && handed_over) {
            if (lane == 0) a.redo[slot_r] = 1;  // (a mark per slot, not a list: a million appends to one counter took 10 ms)
            wave_lds_sync();
            continue;
        }
        while (q_n > 0) serve(q_n < 64 ? q_n : 64);
        wave_lds_sync();
        while (fin < n_spans) finalize(fin++);
        for (int wd = n_spans * 32 + lane; wd < row_words; wd += 64) row[wd] = 0;  // (only L == 0 leaves words unwritten)
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o, 64);
            fst = min(fst, __shfl_xor(fst, o, 64));
            lst = max(lst, __shfl_xor(lst, o, 64));
        }
        if (lane == 0) {
            count[rid] = cnt;
            first[rid] = cnt ? fst : -1;  // m_first_base_in_kmer / m_last_base_in_kmer, src/read.cpp:75-84
            last[rid] = cnt ? lst : -1;
        }
        wave_lds_sync();  // (the next read's first span must not overtake this read's last ring reads)
      }
    }
}

}  // namespace

int flx_cover_queue_launch(flx_ctx *ctx, const CoverArgs &args, bool has_prefilter, unsigned grid, bool every_read_to_second) {
    if (!args.redo) return flx_fail(ctx, FLX_ERR_INVALID, "cover kernel: no room for the marks of reads with insertions / deletions");
    FLX_HIP(ctx, hipMemsetAsync(args.redo, every_read_to_second ? 1 : 0, args.n_reads, ctx->stream));
    // every read; then the reads the first kernel handed over (marked on the device: no host round trip, a grid of resident
    // workgroups walks the marks 64 at a time).  every_read_to_second (FLX_KMER_COVER=q2; tests): every read is marked beforehand and the
    // first kernel does not run — the second one must give the same bits on ANY read, not only on those that are handed to it
    const unsigned grid2 = std::min(grid, 8u * 256u);
    if (has_prefilter) {
        if (!every_read_to_second) hipLaunchKernelGGL((k_kmer_cover_q<true, false>), dim3(grid), dim3(FLX_COVER_THREADS), 0, ctx->stream, args);
        hipLaunchKernelGGL((k_kmer_cover_q<true, true>), dim3(grid2), dim3(FLX_COVER_THREADS), 0, ctx->stream, args);
    } else {
        if (!every_read_to_second) hipLaunchKernelGGL((k_kmer_cover_q<false, false>), dim3(grid), dim3(FLX_COVER_THREADS), 0, ctx->stream, args);
        hipLaunchKernelGGL((k_kmer_cover_q<false, true>), dim3(grid2), dim3(FLX_COVER_THREADS), 0, ctx->stream, args);
    }
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}
