// score_kmer.hip — k-mer mode per-read scoring on gfx950.
//
// Replaces the k-mer branch of the reference's Read::Read (src/read.cpp:43-58: rolling 2-bit 16-mer, one
// set lookup per position, mark bases i-15..i on a hit), first/last covered base (75-84), bad ranges /
// trim / split -> child ranges (86-130) and the scoring of every child read (131-141), plus the shared
// mean / window / cut-off code (208-236, 64-73).
//
// Two kernels:
//   k_kmer_cover  (position-parallel)  seq plane -> coverage bit plane (1 bit per base) + per-read covered
//                 count / first / last.  One workgroup per read; a thread builds the 16 rolling 16-mers ending
//                 at its 16 positions, issues its 16 bitmap lookups back to back, and turns hits into
//                 coverage with a 4-step OR-dilation over its own and its right neighbour's hit mask.
//                 The qualities are 0.0 / 1.0, so the reference's serial FP64 sum is an exact integer:
//                 mean = 100 * popcount / L needs no serial pass.
//   k_kmer_fold   (read-serial)  one lane per read walks its coverage bits and replays get_window_quality
//                 bit-exactly (w -= q[i]/ws; w += q[j]/ws with q/ws in {0, fl(1/ws)}: the drift is real,
//                 SURVEY §8c test_trim_3 = 0x1.5ffffffffffffp+6), and — under --trim/--split — finds the bad
//                 zero-runs, the child ranges, and runs the same recurrence for every child on the fly.
//                 Children never have grandchildren (their coverage is a slice of the parent's),
//                 so a child's mean/window/cut-offs come from the parent's bits without new lookups.
#include "flx_internal.h"
#include "kmerset.h"
#include "cover_common.h"
#include "rank_internal.h"

#ifndef FLX_FARFIRST_LANES
#define FLX_FARFIRST_LANES 16  // settled lanes of a span from which the next span asks the exact table first (score_kmer.hip, below)
#endif
#ifndef FLX_FARFIRST_LANES_LOCUS
#define FLX_FARFIRST_LANES_LOCUS 65  // the same with a text: never (65 > 64) — the text settles the clean lanes, the mode only costs instructions (profiles/r04_microbench.txt)
#endif

namespace {

// ---------------------------------------------------------------------------------------------------
// coverage
// ---------------------------------------------------------------------------------------------------
// One workgroup per read, COVER_THREADS * 16 positions per iteration.  Two instantiations share the batch: 256 threads
// (spans of 4096 positions) for reads longer than kCoverShort, one wavefront (spans of 1024) for the short ones, which
// would leave most of a 256-thread workgroup idle (500 bp reads: 206 -> 98 ms per 1e10 bases, 2 kbp: 83 -> 71; profiles/r02_microbench.txt).
constexpr int kCoverShort = 3072;

template <int COVER_THREADS>
__global__ void __launch_bounds__(COVER_THREADS) k_kmer_cover(const uint8_t *plane, const uint64_t *offsets,
                                                              const int32_t *lengths, const uint32_t *order,
                                                              uint64_t n_reads, const uint32_t *bitmap,
                                                              const uint32_t *prefilter, uint32_t *cov, const uint64_t *cov_off, int32_t *count,
                                                              int32_t *first, int32_t *last) {
    constexpr int COVER_SPAN = COVER_THREADS * 16;  // positions per workgroup iteration
    __shared__ uint32_t sh_hits[COVER_THREADS];
    __shared__ uint16_t sh_p12[COVER_THREADS];
    __shared__ uint8_t sh_anchor[COVER_THREADS];
    __shared__ uint32_t sh_carry;
    __shared__ int sh_cnt[COVER_THREADS / 64], sh_first[COVER_THREADS / 64], sh_last[COVER_THREADS / 64];
    for (uint64_t slot = blockIdx.x; slot < n_reads; slot += gridDim.x) {
        const uint32_t rid = order ? order[slot] : (uint32_t)slot;
        const int L = lengths[rid];
        if ((L > kCoverShort) != (COVER_THREADS > 64)) continue;  // the other instantiation's read (uniform for the workgroup)
        const uint8_t *seq = plane + offsets[rid];
        uint32_t *row = cov + (cov_off[rid] >> 2);
        const int row_words = (((L + 7) / 8 + 15) & ~15) >> 2;
        const int t = threadIdx.x;
        int cnt = 0, fst = 0x7fffffff, lst = -1;
        if (t == 0) sh_carry = 0;
        __syncthreads();
        const int n_spans = (L + COVER_SPAN - 1) / COVER_SPAN;
        for (int sp = n_spans - 1; sp >= 0; --sp) {  // descending: the right neighbour's hits are already known
            const int p0 = sp * COVER_SPAN + t * 16;
            const bool active = p0 < L && L >= 16;
            uint32_t hits = 0;
            uint32_t kmers[16];
            uint32_t kleft[5] = {0, 0, 0, 0, 0};  // the 16-mers ending at p0-5 .. p0-1 (their low 24 bits are the 12-mers there)
            uint32_t p12 = 0;                     // bit j: the 12-mer ending at p0 + j occurs in the set
            uint32_t p12_left = 0;                // the same for p0-5 .. p0-1 (only the first thread of a span looks them up itself)
            if (active) {
                // bases [p0-16, p0+16): both loads are 16-byte aligned (read starts are)
                uint4 a = make_uint4(0, 0, 0, 0);
                if (p0 > 0) a = *reinterpret_cast<const uint4 *>(seq + p0 - 16);
                const uint4 b = *reinterpret_cast<const uint4 *>(seq + p0);
                // 2 bits per base: hi = bases p0-16 .. p0-1, lo = bases p0 .. p0+15 (earliest base in the top bits); the 16-mer
                // ending at p0 + j is a 32-bit window of hi:lo
                const uint32_t hi = (codes4(a.x) << 24) | (codes4(a.y) << 16) | (codes4(a.z) << 8) | codes4(a.w);
                const uint32_t lo = (codes4(b.x) << 24) | (codes4(b.y) << 16) | (codes4(b.z) << 8) | codes4(b.w);
#pragma unroll
                for (int j = 0; j < 15; ++j) kmers[j] = __builtin_amdgcn_alignbit(hi, lo, 2 * (15 - j));
                kmers[15] = lo;
#pragma unroll
                for (int j = 0; j < 5; ++j) kleft[j] = hi >> (2 * (4 - j));  // low 24 bits: the 12-mer ending at p0 - 5 + j
                // 12-mer prefilter (kmerset.h): one L2 lookup per position, all 16 in flight
                if (prefilter) {
                    uint32_t pw[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) pw[j] = prefilter[flx_sub12(kmers[j], 0) >> 5];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int i = p0 + j;
                        if (i >= 11 && i < L) p12 |= ((pw[j] >> (kmers[j] & 31)) & 1u) << j;
                    }
                    if (t == 0 && p0 > 0) {
#pragma unroll
                        for (int j = 0; j < 5; ++j)
                            if (p0 - 5 + j >= 11) p12_left |= ((prefilter[flx_sub12(kleft[j], 0) >> 5] >> (kleft[j] & 31)) & 1u) << j;
                    }
                } else {
                    p12 = 0xffffu;
                    p12_left = 0x1fu;
                }
            }
            sh_p12[t] = (uint16_t)p12;
            __syncthreads();
            // Candidates: bit j = the 16-mer ending at p0 + j may be present (all five of its 12-mers are, and it lies in the read).
            // Lookups: a base is covered iff ANY 16-mer over it is present, so of a run of consecutive candidates only
            // the END points matter as long as they are present — their spans [j-15, j] overlap inside a thread's 16
            // positions and cover everything the run's other members could.  Per thread: probe the first and last
            // candidate of every run (the first one is skipped when the run continues from the left neighbour and that
            // neighbour's last probe hit); only if a probe MISSES (a filter false positive, ~2 %) are the run's other
            // members looked up too.  Coverage is identical to looking all of them up; far requests per position drop
            // from one per present 16-mer to ~2 per clean stretch.
            uint32_t cand = 0, probed = 0;
            bool left_run = false;  // the candidate run at bit 0 continues from the left neighbour's bit 15
            if (active) {
                if (t > 0) p12_left = prefilter ? (uint32_t)(sh_p12[t - 1] >> 11) : 0x1fu;
                const uint32_t m = (p12_left >> 1) | (p12 << 4);  // bit i: the 12-mer ending at p0 - 4 + i
                cand = m & (m >> 1) & (m >> 2) & (m >> 3) & (m >> 4) & 0xffffu;
                uint32_t valid = 0xffffu;  // 16-mers end at positions 15 .. L-1
                if (p0 < 15) valid &= ~((1u << (15 - p0)) - 1u);
                if (p0 + 16 > L) valid &= (1u << (L - p0)) - 1u;
                cand &= valid;
                left_run = t > 0 && p12_left == 0x1fu && p0 - 1 >= 15;  // the left neighbour's bit 15 is a candidate
                const uint32_t starts = cand & ~(cand << 1), ends = cand & ~(cand >> 1);
                probed = ends | (left_run ? starts & ~1u : starts);
                uint32_t words[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) words[j] = ((probed >> j) & 1u) ? bitmap[kmers[j] >> 5] : 0u;  // independent, in flight
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if ((probed >> j) & 1u) hits |= ((words[j] >> (kmers[j] & 31)) & 1u) << j;
            }
            sh_anchor[t] = (uint8_t)((hits >> 15) & 1u);
            __syncthreads();
            if (active) {
                if ((cand & 1u) && !(probed & 1u)) {  // run continuing from the left: its start is needed only if the neighbour's end missed
                    if (!sh_anchor[t - 1]) {
                        probed |= 1u;
                        hits |= (bitmap[kmers[0] >> 5] >> (kmers[0] & 31)) & 1u;
                    }
                }
                // A probe that missed (typically a false candidate right behind a clean stretch: the 12-mers it shares with the
                // stretch are genuine, so the filter passes it with probability ~0.45): what is still needed for an exact answer
                // are the run members OUTSIDE the span of its confirmed ones (or all of it, if none is confirmed yet).  Round 2
                // asks for those within 4 positions of a missed probe (a false extension is rarely longer), round 3 for the rest.
                auto still_needed = [&]() -> uint32_t {
                    uint32_t up = hits, dn = hits, m = cand;  // flood the confirmed bits along their candidate runs
                    up |= (up << 1) & m; dn |= (dn >> 1) & m;
                    uint32_t mu = m & (m << 1), md = m & (m >> 1);
                    up |= (up << 2) & mu; dn |= (dn >> 2) & md;
                    mu &= mu << 2; md &= md >> 2;
                    up |= (up << 4) & mu; dn |= (dn >> 4) & md;
                    mu &= mu << 4; md &= md >> 4;
                    up |= (up << 8) & mu; dn |= (dn >> 8) & md;
                    const uint32_t inside = up & dn;  // between the lowest and the highest confirmed member of a run
                    return cand & ~inside & ~probed & 0xffffu;
                };
                if (probed & ~hits) {
                    const uint32_t miss = probed & ~hits;
                    uint32_t near = (miss << 1) | (miss << 2) | (miss << 3) | (miss << 4) | (miss >> 1) | (miss >> 2) | (miss >> 3) | (miss >> 4);
                    for (int round = 0; round < 2; ++round) {
                        const uint32_t ask = still_needed() & (round == 0 ? near : 0xffffu);
                        if (ask) {
                            uint32_t words[16];
#pragma unroll
                            for (int j = 0; j < 16; ++j) words[j] = ((ask >> j) & 1u) ? bitmap[kmers[j] >> 5] : 0u;
#pragma unroll
                            for (int j = 0; j < 16; ++j)
                                if ((ask >> j) & 1u) hits |= ((words[j] >> (kmers[j] & 31)) & 1u) << j;
                            probed |= ask;
                        }
                    }
                }
            }
            sh_hits[t] = hits;
            __syncthreads();
            const uint32_t next = (t + 1 < COVER_THREADS) ? sh_hits[t + 1] : sh_carry;
            uint32_t x = hits | (next << 16);
            x |= x >> 1;
            x |= x >> 2;
            x |= x >> 4;
            x |= x >> 8;  // bit j = OR of hit bits j .. j+15: base p0+j lies in a present 16-mer
            uint32_t c16 = x & 0xffffu;
            if (p0 >= L) c16 = 0;
            else if (p0 + 16 > L) c16 &= (1u << (L - p0)) - 1u;
            cnt += __popc(c16);
            if (c16) {
                fst = min(fst, p0 + (__ffs(c16) - 1));
                lst = max(lst, p0 + (32 - __clz(c16)));
            }
            const uint32_t hi = __shfl_down(c16, 1, 64);
            const int word = p0 >> 5;
            if ((t & 1) == 0 && word < row_words) row[word] = c16 | (hi << 16);
            __syncthreads();
            if (t == 0) sh_carry = hits;
            __syncthreads();
        }
        // zero the padding words beyond the spans (rows are padded to 16 bytes)
        for (int wd = n_spans * (COVER_SPAN / 32) + t; wd < row_words; wd += COVER_THREADS) row[wd] = 0;
        // block reduction of count / first / last
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o, 64);
            fst = min(fst, __shfl_xor(fst, o, 64));
            lst = max(lst, __shfl_xor(lst, o, 64));
        }
        if ((t & 63) == 0) {
            sh_cnt[t >> 6] = cnt;
            sh_first[t >> 6] = fst;
            sh_last[t >> 6] = lst;
        }
        __syncthreads();
        if (t == 0) {
            int c = 0, f = 0x7fffffff, l = -1;
            for (int wv = 0; wv < COVER_THREADS / 64; ++wv) {
                c += sh_cnt[wv];
                f = min(f, sh_first[wv]);
                l = max(l, sh_last[wv]);
            }
            count[rid] = c;
            first[rid] = c ? f : -1;  // m_first_base_in_kmer / m_last_base_in_kmer, src/read.cpp:75-84
            last[rid] = c ? l : -1;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// coverage, wave level (round 3) — the kernel that runs; k_kmer_cover above stays as the second implementation
// (FLX_KMER_COVER=v2, cross-checked in the tests)
// ---------------------------------------------------------------------------------------------------
// One WAVEFRONT owns a read and walks it left to right in spans of 1024 positions; a lane owns 16 consecutive positions
// (one 16-byte load, prefetched one span ahead).  No LDS, no barrier: the left neighbour's bases / 12-mer bits / last
// candidate travel by __shfl_up, lane 63's by a wave-uniform carry into the next span, and the coverage of a span is
// written one span late, when the hits of its right neighbour are known.
//
// What a lookup costs (tools/tabench, profiles/r03_microbench.txt): one distinct cache LINE per instruction — 261 G/s
// from the L2, 55 G/s beyond it, the two classes add up — so the kernel is built around lines, not lanes:
//   * prefilter: ONE byte of `pre11` answers the 12-mers of two consecutive positions (kmerset.h): 8 loads per lane;
//   * exact membership: ONE byte of `exact15` answers the 16-mers of two consecutive positions: a far request per PAIR;
//   * which pairs are asked: a base is covered iff ANY 16-mer over it is a member, and two confirmed members inside a
//     lane's window of 17 positions (its own 16 + the left neighbour's last) are at most 16 apart, so together they cover
//     everything a candidate between them could.  Only the OUTERMOST members matter: search the candidates from the top
//     down until the first member, and from the bottom up (not at all if the left neighbour's last position is a member),
//     one pair per side and round; a clean stretch costs one request per 16 positions, a false candidate costs one only
//     when it lies outside the confirmed span.  Rounds repeat until no lane of the wave has an open question.
//
// LOCUS (round 4, assembly references — kmerset.h: flx_locus): the wave carries a DIAGONAL, the text position its read's base 0
// would have in the assembly.  Every lane compares its 16 bases with the text along the diagonal (one coalesced 8-byte load per
// lane and span, prefetched a span ahead); 16 matching bases inside one strand copy ARE a member, without any request — `known`.
// The known members enter the search as confirmed hits: a lane whose own last 16-mer and whose left neighbour's are known is
// settled before anything is asked, a lane with a mismatch looks up only the prefilter pairs outside [lowest hit - 4, highest
// hit] and asks the exact table only outside its confirmed span, exactly as before (a 16-mer with a mismatch against the locus
// may still occur elsewhere).  The diagonal comes from the seed table: eight lanes look their own 16 bases up (one far request
// each) when the wave has none or the last 16 lanes of the previous span matched nowhere; a seed that fails leaves the old
// diagonal in place as a hypothesis that costs nothing to test.
template <bool HAS_PREFILTER, bool LOCUS>
__global__ void __launch_bounds__(FLX_COVER_THREADS) FLX_COVER_OCC k_kmer_cover_w(const CoverArgs a) {
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
    for (uint64_t slot = wave0; slot < n_reads; slot += n_waves) {
        const uint32_t rid = __builtin_amdgcn_readfirstlane(order ? order[slot] : (uint32_t)slot);
        const int L = __builtin_amdgcn_readfirstlane(lengths[rid]);
        const uint8_t *seq = plane + offsets[rid];
        uint32_t *row = cov + (cov_off[rid] >> 2);
        const int row_words = (((L + 7) / 8 + 15) & ~15) >> 2;
        const int n_spans = (L + 1023) >> 10;
        int cnt = 0, fst = 0x7fffffff, lst = -1;
        // carried from lane 63 of the previous span (wave-uniform)
        uint32_t c_lo = 0, c_p12 = 0, c_cand15 = 0, c_hit15 = 0;
        uint32_t prev_hits = 0;  // hits of the previous span, waiting for their right neighbour's
        bool far_first = false;  // this span asks the exact table BEFORE the prefilter (decided by the previous span, below)
        // LOCUS: the diagonal (wave-uniform), whether a seed is due, the text of this span / the next one, lane 63's carry
        long long diag = 0;
        bool have_diag = false, carry_ok = false;
        uint32_t c_mb = 0xffffffffu /* mismatches | piece starts << 16 of lane 63 */, c_us = 0 /* its U13 | S1 << 16 */, c_known15 = 0, c_twx = 0, c_twy = 0xffffu;
        uint2 tw = make_uint2(0, 0xffffu), tw_next = make_uint2(0, 0xffffu);
        uint32_t ts = 0, ts_next = 0, c_ts = 0;  // S1 bits of those text words (kmerset.h: safe1); lane 63's for the next span
        const bool has_s1 = a.loc.safe1 != nullptr;
        // the text word that holds the LAST base of the lane's 16 at this diagonal, for the lane whose 16 bases start at `base` +
        // 16 * lane (index clamped into the padded array).  The diagonal and `base` are wave-uniform: the 64-bit part of the index
        // is scalar work, a lane adds its number and clamps (the cover kernel is bound by its vector instructions)
        auto word_index = [&](long long dg, int base) -> uint32_t {
            long long u = ((dg + base + 15) >> 4) + (long long)kLocusPad;  // (16 * lane + c) >> 4 == lane + (c >> 4)
            u = u < -64 ? -64 : (u > (long long)loc_n_alloc ? (long long)loc_n_alloc : u);
            const int w = (int)u + lane;
            return (uint32_t)max(0, min(w, (int)loc_n_alloc - 1));
        };
        auto text_word = [&](long long dg, int base) -> uint2 {
            // (a 32-bit byte offset on a scalar base: one address register — the text has at most 2^28 positions, 2^27 bytes)
            const uint64_t tv = *(FLX_GLOBAL_PTR(uint64_t))(FLX_KARG_PTR(uint8_t, loc.text) + (uint32_t)(word_index(dg, base) * 8u));
            return make_uint2((uint32_t)tv, (uint32_t)(tv >> 32));  // (non-temporal here is slower: 14.2 vs 13.8 ms per 1e10 — a text word is used again by the next span's lane 0 and by reads of the same locus)
        };
        auto safe_word = [&](long long dg, int base) -> uint32_t {  // the S1 bits of that word
            return has_s1 ? (uint32_t)*(FLX_GLOBAL_PTR(uint16_t))(FLX_KARG_PTR(uint8_t, loc.safe1) + (uint32_t)(word_index(dg, base) * 2u)) : 0u;
        };

        auto finalize = [&](int sp, uint32_t h, uint32_t right_of_63) {  // hits of span sp -> coverage bits, counts, row words
            const int p0 = (sp << 10) + lane * 16;
            const uint32_t next = flx_from_right(h, right_of_63);
            uint32_t x = h | (next << 16);
            x |= x >> 1;
            x |= x >> 2;
            x |= x >> 4;
            x |= x >> 8;  // bit j = OR of hit bits j .. j+15: base p0+j lies in a member 16-mer (src/read.cpp:53-54)
            uint32_t c16 = x & 0xffffu;
            if (((sp + 1) << 10) > L)  // (wave-uniform: only the read's last span has positions to cut off)
            {
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

        uint4 raw = make_uint4(0, 0, 0, 0);
        // (offsets as unsigned 32-bit values: a uniform base plus a 32-bit lane offset is one address register, not two)
        if (lane * 16 < L) raw = flx_plane16(seq + (uint32_t)(lane * 16));  // rows are 16-byte aligned and padded
        for (int sp = 0; sp < n_spans; ++sp) {
            const int p0 = (sp << 10) + lane * 16;
            uint4 raw_next = make_uint4(0, 0, 0, 0);
            if (p0 + 1024 < L) raw_next = flx_plane16(seq + (uint32_t)(p0 + 1024));
            if (LOCUS && have_diag && sp + 1 < n_spans) {
                tw_next = text_word(diag, (sp << 10) + 1024);
                ts_next = safe_word(diag, (sp << 10) + 1024);
            }
            // 2 bits per base, earliest base on top: lo = my 16 bases, hi = the 16 before them
            const uint32_t lo = (codes4(raw.x) << 24) | (codes4(raw.y) << 16) | (codes4(raw.z) << 8) | codes4(raw.w);
            const uint32_t hi = flx_from_left(lo, c_lo);
            // positions p0 + j that end a 12-mer / a 16-mer inside the read
            uint32_t valid12 = 0, valid16 = 0;
            if (sp > 0 && ((sp + 1) << 10) <= L) {  // (wave-uniform: a span inside the read has every position, no lane computes masks)
                valid12 = valid16 = 0xffffu;
            } else
            if (p0 < L) {
                valid12 = valid16 = 0xffffu;
                if (p0 < 11) valid12 &= ~((1u << (11 - p0)) - 1u);
                if (p0 < 15) valid16 &= ~((1u << (15 - p0)) - 1u);
                if (p0 + 16 > L) {
                    valid12 &= (1u << (L - p0)) - 1u;
                    valid16 &= (1u << (L - p0)) - 1u;
                }
            }
            // ---- LOCUS: members known from the text along the diagonal ----
            uint32_t known = 0, refuted = 0;  // refuted: not a text match, but holds a text-matching 13-mer that occurs nowhere else
            uint32_t text12 = 0;              // bit j: the 12 bases ending at my position j match the text inside one piece: that 12-mer IS present
            if (LOCUS) {
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
                    const uint32_t b_own = (((twl.y & 0xffffu) >> (e + 1)) | (tw.y << (15 - e))) & 0xffffu;  // bit j: my base j is the first of a piece
                    const uint32_t u_own = (((twl.y >> 16) >> (e + 1)) | ((tw.y >> 16) << (15 - e))) & 0xffffu;  // bit j: a unique 13-mer starts at my base j
                    const uint32_t s_own = ((tsl >> (e + 1)) | (ts << (15 - e))) & 0xffffu;  // bit j: the text's 16 bases from my base j on are S1
                    const uint32_t x = lo ^ t_own;
                    uint32_t m = (x | (x >> 1)) & 0x55555555u;  // even bit 2k: the base k places from the END differs
                    m = (m | (m >> 1)) & 0x33333333u;
                    m = (m | (m >> 2)) & 0x0f0f0f0fu;
                    m = (m | (m >> 4)) & 0x00ff00ffu;
                    m = (m | (m >> 8)) & 0xffffu;
                    const uint32_t mml = __brev(m) >> 16;  // bit j: my base j differs from the text
                    const uint32_t mb0 = carry_ok ? c_mb : 0xffffffffu, us0 = carry_ok ? c_us : 0u;
                    const uint32_t mmh = flx_from_left(mml, mb0 & 0xffffu), bh = flx_from_left(b_own, mb0 >> 16);
                    const uint32_t uh = flx_from_left(u_own, us0 & 0xffffu), sh = flx_from_left(s_own, us0 >> 16);
                    const uint32_t z = ~(mmh | (mml << 16));  // bit i: base i of the window [p0 - 16, p0 + 16) matches
                    uint32_t r = z & (z >> 1);
                    r &= r >> 2;
                    r &= r >> 4;
                    r &= r >> 8;  // bit i: bases i .. i + 15 match
                    uint32_t q = ~(bh | (b_own << 16)) >> 1;  // bit i: no piece starts at base i + 1
                    q &= q >> 1;
                    q &= q >> 2;
                    q &= q >> 4;
                    q &= q >> 7;  // bit i: none at i + 1 .. i + 15 — the 16 bases from i on lie in one piece of the text
                    {
                        uint32_t m12 = z & (z >> 1);
                        m12 &= m12 >> 2;
                        m12 &= m12 >> 4;
                        m12 &= m12 >> 4;  // bit i: bases i .. i + 11 match
                        uint32_t q12 = ~(bh | (b_own << 16)) >> 1;
                        q12 &= q12 >> 1;
                        q12 &= q12 >> 2;
                        q12 &= q12 >> 4;
                        q12 &= q12 >> 3;  // bit i: no piece starts at i + 1 .. i + 11 (a piece has at least 16 bases: the 12-mer lies in one of its 16-mers)
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
                for (int seeds_left = FLX_LOCUS_SEEDS;;) {
                    if (again) compare();
                    const unsigned long long kn = __ballot(known != 0);
                    const unsigned long long tail = kn ? whole & ~((2ull << (63 - __clzll(kn))) - 1ull) : whole;
                    if (seeds_left-- == 0 || __popcll(tail) < FLX_LOCUS_TAIL) break;
                    bool tries;
                    if (kn) {
                        const unsigned long long t1 = tail & (tail - 1), t2 = t1 & (t1 - 1);  // without its first lane / first two lanes
                        tries = lane == __ffsll(t1) - 1 || lane == __ffsll(t2) - 1;
                    } else {
                        tries = (lane & 7) == 3 && ((whole >> lane) & 1ull);
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
                    if (!found) break;
                    const int src = __ffsll(found) - 1;
                    const long long nd = (long long)__builtin_amdgcn_readlane(tpos, src) - (long long)((sp << 10) + src * 16);
                    if (have_diag && nd == diag) break;  // the same locus: what is missing are mismatches, not the diagonal
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
            }

            // ---- exact membership: one byte of exact15 answers the pair of positions (a, a + 1), any a in 0..14 — the 15 bases
            // ending at a are the byte's index, the base before them picks the bit of position a, the base after them the bit of
            // a + 1.  A question from ABOVE (top-down search) takes the pair that ENDS at the asked position, one from below the
            // pair that starts there: either way the request also settles the next candidate in the direction of the search. ----
            uint32_t hits = known, probed = known;
            auto probe = [&](int top, int bot, uint32_t keep) {  // positions asked from above / from below, -1 = none
                const int a0 = top > 0 ? top - 1 : 0, a1 = bot < 14 ? bot : 14;
                uint32_t g0 = 0, g1 = 0;
                // plain byte loads: non-temporal ones measured 8 % slower here (43.3 vs 40.1 ms per 1e10 positions), 4-byte loads
                // 6 % slower — although a microbenchmark that mixes table and far lookups in one burst prefers nt (tools/tabench (7))
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
                hits &= keep;  // positions outside the read hold no 16-mer
            };

            // ---- far first (clean stretches): where most lanes of the previous span had their own last 16-mer AND their left
            // neighbour's confirmed, ask for the last 16-mer of every lane before anything else.  A lane whose own and whose left
            // neighbour's are members is SETTLED: its 16 bases lie in its own last 16-mer, the 15 before them in the neighbour's,
            // so none of its other 16-mers can add coverage and its eight prefilter lines are never fetched.  The request is the
            // one a clean lane needs anyway; it is wasted only on a lane whose last pair holds no candidate. ----
            bool settled = false;
            uint32_t ltop = 0;  // far first: is the left neighbour's last 16-mer a member
            if (far_first) {
                const bool ask15 = (valid16 >> 15) != 0 && ((known | refuted) >> 15) == 0;
                if (__any(ask15)) probe(ask15 ? 15 : -1, -1, valid16);
                ltop = flx_from_left(hits >> 15, c_hit15);
                settled = (hits >> 15) && ltop;
            } else if (LOCUS) {
                ltop = flx_from_left(known >> 15, c_hit15);
                settled = (known >> 15) && ltop;
            }
            // LOCUS: the 12-mers ending at [lowest hit - 4, highest hit] need no lookup — those inside a member are present, the
            // others only make candidates between two confirmed members, which are never asked
            uint32_t need12 = 0xffffu;
            if (LOCUS && hits) {
                const int a = __ffs(hits) - 1, b = 31 - __clz(hits);
                const int from = a > 4 ? a - 4 : 0;
                need12 = ~(((2u << b) - 1u) & ~((1u << from) - 1u)) & 0xffffu;
            }
            if (LOCUS) need12 &= ~text12;  // (12-mers that match the text between two mismatches less than 16 apart: present without a lookup)

            // ---- 12-mer prefilter: pair m = positions p0 + 2m, p0 + 2m + 1; x.C.y = the 13 bases ending at p0 + 2m + 1 ----
            uint32_t p12 = 0xffffu;  // (a settled lane: every 12-mer of its last 16-mer is present, the others are not needed)
            if (HAS_PREFILTER && !LOCUS && !settled) {
                uint32_t byte[8], sel[8];
                FLX_GLOBAL_PTR(uint8_t) pre11 = FLX_KARG_PTR(uint8_t, pre11);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const uint32_t a = __builtin_amdgcn_alignbit(hi, lo, 28 - 4 * m);
                    const flx_pre11_slot q = flx_pre11((a >> 2) & 0x3FFFFFu, (a >> 24) & 3u, a & 3u);
                    byte[m] = pre11[q.index];
                    sel[m] = q.even_bit | (q.odd_bit << 8);
                }
                p12 = 0;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    p12 |= (((byte[m] >> (sel[m] & 0xffu)) & 1u) | (((byte[m] >> (sel[m] >> 8)) & 1u) << 1)) << (2 * m);
            }
            if (HAS_PREFILTER && LOCUS) {
                // In TWO rounds: a 16-mer is out as soon as ONE of its five 12-mers is absent, and what is left to look up holds a
                // mismatch against the text, so it is absent more often than not.  Round 1 fetches the even pairs (positions 0 1,
                // 4 5, 8 9, 12 13) where needed; every 16-mer holds two or three of those positions, so most are out after it.
                // Round 2 fetches an odd pair only if one of the six 16-mers that hold its 12-mers — five of them may be the right
                // neighbour's — is still alive under the assumption that every 12-mer not yet seen is present.  A pair that is
                // skipped keeps that assumption: it only concerns 16-mers that are out anyway.
                // (the reverse complement of the whole 32-base window once: the canonical form of every pair's 11-mer is then one
                // funnel shift, and a byte read for the other strand is looked at bit-reversed — bit 7 - x is bit x, bit 3 - y is bit
                // 4 + y — instead of with two selected bit numbers: kmerset.h, flx_pre11, in 19 instead of 33 instructions per pair)
                auto rc32 = [](uint32_t w) {
                    const uint32_t r = __brev(w);
                    return ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));
                };
                const uint64_t r64 = ((uint64_t)rc32(lo) << 32) | rc32(hi);
                auto fetch = [&](uint32_t want, int parity) -> uint32_t {  // actual bits of the pairs m = parity, parity + 2, .. that hold a wanted position; 1 elsewhere
                    uint32_t byte[4], got = parity ? 0x3333u : 0xCCCCu;
                    FLX_GLOBAL_PTR(uint8_t) pre11 = FLX_KARG_PTR(uint8_t, pre11);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = 2 * k + parity;
                        const uint32_t a = __builtin_amdgcn_alignbit(hi, lo, 28 - 4 * m);  // x.C.y, the 13 bases ending at position 2m + 1
                        const uint32_t c = (a >> 2) & 0x3FFFFFu, rc = (uint32_t)(r64 >> (12 + 4 * m)) & 0x3FFFFFu;
                        const uint32_t kk = (a & 0x2000u) ? rc : c;  // the middle base of C is G or T: the byte belongs to the other strand
                        const uint32_t index = ((kk >> 12) << 11) | (kk & 0x7FFu);
                        byte[k] = ((want >> (2 * m)) & 3u) ? pre11[index] : 0xffu;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = 2 * k + parity;
                        const uint32_t a = __builtin_amdgcn_alignbit(hi, lo, 28 - 4 * m);
                        const uint32_t b = (a & 0x2000u) ? (__brev(byte[k]) >> 24) : byte[k];
                        const uint32_t two = ((b >> ((a >> 24) & 3u)) & 1u) | (((b >> (4u + (a & 3u))) & 1u) << 1);
                        got |= two << (2 * m);  // (a pair that was not fetched holds 0xff: both present)
                    }
                    return got;
                };
                const uint32_t want = settled ? 0u : (need12 & valid12);
                {
                    // round 1 as well leaves out the pairs whose 12-mers only lie in 16-mers the text has refuted (U13, S1)
                    uint32_t alive = settled ? 0u : (valid16 & (~refuted | known));
                    const uint32_t right = flx_from_right(alive, 0xffffu);
                    uint32_t dep = alive | (right << 16);
                    dep |= dep >> 1;
                    dep |= dep >> 2;
                    dep |= dep >> 1;
                    const uint32_t w1 = want & 0x3333u & dep;
                    p12 = __any(w1 != 0) ? fetch(w1, 0) : 0xffffu;  // (a span the text settles: nothing is computed for it)
                }
                if (__any((want & 0xCCCCu) != 0)) {
                    const uint32_t v1 = p12 & valid12;
                    const uint32_t l1 = flx_from_left(v1 >> 11, c_p12);
                    const uint32_t m1 = (l1 >> 1) | (v1 << 4);
                    uint32_t alive = m1 & (m1 >> 1) & (m1 >> 2) & (m1 >> 3) & (m1 >> 4) & valid16 & ~refuted;
                    if (settled) alive = 0;  // (a settled lane asks nothing; its neighbours' 16-mers that reach into it count below)
                    const uint32_t right = flx_from_right(alive, 0xffffu);  // (lane 63: the next span's first lane is not known yet)
                    uint32_t dep = alive | (right << 16);
                    dep |= dep >> 1;
                    dep |= dep >> 2;
                    dep |= dep >> 1;  // bit q: one of the 16-mers ending at q .. q + 4 (those that hold the 12-mer ending at q) is alive
                    const uint32_t w2 = want & 0xCCCCu & dep;
                    if (__any(w2 != 0)) p12 &= fetch(w2, 1);
                }
            }
            p12 &= valid12;
            uint32_t p12_left = flx_from_left(p12 >> 11, c_p12);  // the left lane's 12-mers ending at its positions 11..15 = mine at -5..-1
            if (!HAS_PREFILTER) p12_left = 0x1fu;
            const uint32_t m12 = (p12_left >> 1) | (p12 << 4);  // bit i: the 12-mer ending at p0 - 4 + i
            uint32_t cand = m12 & (m12 >> 1) & (m12 >> 2) & (m12 >> 3) & (m12 >> 4) & valid16;  // all five 12-mers present
            if (LOCUS) cand &= ~refuted | known;  // (a member known on one diagonal cannot be refuted on another — its 13-mers then occur twice in the text — but nothing is lost by saying so)
            if (settled) cand = hits;  // nothing open: the confirmed members are all this lane contributes
            const uint32_t lcand = flx_from_left(cand >> 15, c_cand15);
            hits &= cand;  // (a member is always a candidate)
            probed |= ~cand & 0xffffu;

            // One step of the search in the lane's window of 17 positions (bit 0 = the left neighbour's last position, bit j + 1 =
            // my position j): the highest open candidate above the confirmed members and the lowest one below them.
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
            uint32_t lhit;
            int top, bot;
            if (far_first) {
                lhit = ltop & lcand;  // already exact
            } else {
                // first step on a BET: a candidate at the left neighbour's last position is taken for a member (it is the top of
                // that lane's search, so its answer arrives with this round's), corrected right after
                const bool need = next_asks(lcand, top, bot);
                if (__any(need)) probe(top, bot, cand);
                lhit = flx_from_left(hits >> 15, c_hit15) & lcand;
            }
            for (;;) {
                const bool need = next_asks(lhit, top, bot);
                if (!__any(need)) break;
                probe(top, bot, cand);
            }

            // far first for the next span?  Per lane the skipped prefilter lines are worth 8 x 3.8 ps, a wasted request 18 ps
            // (tools/tabench): worth it from about half the lanes settled.
            if (LOCUS) {  // lanes the text settles anyway do not count: they ask nothing either way
                const uint32_t lknown = flx_from_left(known >> 15, c_known15);
                far_first = __popcll(__ballot((hits >> 15) && lhit && !((known >> 15) && lknown))) >= FLX_FARFIRST_LANES_LOCUS;
                c_known15 = __builtin_amdgcn_readlane(known >> 15, 63);
            } else {
                far_first = __popcll(__ballot((hits >> 15) && lhit)) >= FLX_FARFIRST_LANES;
            }

            // ---- coverage of the previous span (its lane 63 needed my lane 0's hits), then carry ----
            if (sp > 0) finalize(sp - 1, prev_hits, __builtin_amdgcn_readfirstlane(hits));
            prev_hits = hits;
            c_lo = __builtin_amdgcn_readlane(lo, 63);
            c_p12 = __builtin_amdgcn_readlane(p12 >> 11, 63);
            c_cand15 = __builtin_amdgcn_readlane(cand >> 15, 63);
            c_hit15 = __builtin_amdgcn_readlane(hits >> 15, 63);
            raw = raw_next;
            if (LOCUS) {
                tw = tw_next;
                ts = ts_next;
            }
        }
        if (n_spans > 0) finalize(n_spans - 1, prev_hits, 0u);
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
    }
}

// ---------------------------------------------------------------------------------------------------
// serial fold over the coverage bits
// ---------------------------------------------------------------------------------------------------
#ifndef FLX_FOLD_FMA
#define FLX_FOLD_FMA 1
#endif
// ---- the window recurrence on the integer grid (round 5) ------------------------------------------------------------------------
// src/read.cpp:228-231 with qualities 0.0 / 1.0 is  w = fl(fl(w - tb * d) + lb * d),  d = fl(1 / ws), and the drift of those
// roundings is part of the result.  But WHERE w rounds is known: on the grid of its binade.  Let d_E be d rounded to the grid of
// binade E (2^(E-52)); a GROUP is a run of binades on which d_E is the same real number d* (and no binade rounds d on a tie).  While
//   * w stays strictly above the bottom of its group (and above 4 d: a step that takes a base out and puts one in dips by d, and the
//     way back is only exact from at most one binade down), and
//   * w does not reach a binade above the one it was in when the regime began (there w's own low bits would be rounded away),
// every step is EXACT:  w = w_b + c * d*  with c the number of covered bases that entered the window minus those that left — all
// operands are multiples of the regime's grid, nothing rounds.  A word of 32 positions then is three small integers — the total,
// the lowest and the highest prefix of its +-1 walk (a table over the (new, old) nibble pairs in LDS) — two compares and two adds;
// the minimum of w over the word is w_b + (lowest prefix) * d*.  A word that leaves the regime is replayed in floating point, FOR
// THAT LANE (the others keep their integer step), and the regime begins again from the value it ends on.  ws = 250 (the default):
// d* = d + 4 ulp on every binade from 2^-4 up, so one regime holds while 16 of the 250 bases are covered and w has been as high
// before; on the synthetic reads 0.7 % of a lane's words are replayed (the way into and out of a junk block, the first time a read's
// window fills up).  tools/sim_fold_grid.cpp: the same regime logic on the host against the plain recurrence, 180 000 bit streams x
// 6000 window sizes, bit for bit — and the tests hold this kernel against the FP kernel (FLX_KMER_FOLD_GRID=0) on every read and child.
struct GridTab {
    enum { kMax = 26 };
    double dstar[kMax];  // per binade of w (biased exponent e0 + i): d on that binade's grid; 0 = no regime there (a tie, or outside)
    double lv[kMax];     // the value w must stay strictly above: max(bottom of the binade's group, 4 d)
    int top[kMax];       // biased exponent of the highest binade of the binade's group
    int e0, n;
};

constexpr int kInlineChildren = 8;
struct FoldArgs {
    GridTab gt;
    const uint32_t *cov;
    const uint64_t *cov_off;
    const int32_t *lengths;
    const uint32_t *order;
    uint64_t n_reads;
    const int32_t *count;
    const int32_t *first;
    const int32_t *last;
    int ws;
    int ring_words;  // RING kernels: words per lane in the LDS ring (a power of two)
    int events;      // FLX_KMER_FOLD_EVENTS=1: the steady state walks the positions where the window's edges differ (measured: not faster)
    int grid;        // 1: the steady state runs on the integer grid (GridTab below; FLX_KMER_FOLD_GRID=0 and windows without a wide group: 0)
    double ws_d;
    double delta;  // fl(1.0 / ws): the value of q/ws for a covered base (src/read.cpp:228-229)
    double clamp;  // 0.5 / ws
    flx_params p;
    double *mean_q;
    double *window_q;
    uint8_t *passed;
    // MODE 3 leaves the first kInlineChildren ranges of every read here ([n_reads][kInlineChildren][2], or NULL): they are moved to
    // their places in the CSR once the counts have been scanned, and the ranges pass (MODE 5) only runs for a batch in which some
    // read has more (round 5: MODE 5 walked every row again for ~1 child per read — 7.7 of C4's 33 ms of folds)
    int32_t *inline_ranges;
    // MODE 5 / 6: one lane per child
    uint32_t *child_parent;         // [n_children] read index of every child (written by MODE 5, read by MODE 6)
    const uint32_t *child_order;    // [n_children] children by descending length (MODE 6)
    uint64_t n_children;
    // children
    uint32_t *n_child;              // [n] (count pass)
    const uint64_t *child_offsets;  // [n+1] (emit pass)
    int32_t *child_ranges;
    double *child_mean_q;
    double *child_window_q;
    uint8_t *child_passed;
};

__device__ __forceinline__ uint8_t cutoffs(const flx_params &p, int L, double mean, double window) {
    bool ok = true;  // src/read.cpp:64-73
    if (p.min_length_set && L < p.min_length) ok = false;
    else if (p.max_length_set && L > p.max_length) ok = false;
    else if (p.min_mean_q_set && mean < p.min_mean_q) ok = false;
    else if (p.min_window_q_set && window < p.min_window_q) ok = false;
    return ok ? 1 : 0;
}

struct Win {  // one sliding-window recurrence (parent or current child)
    int cnt;    // covered bases so far
    double w;   // window quality
    double mn;  // its minimum
};

__device__ __forceinline__ double window_result(const FoldArgs &a, int len, int cnt, double mn) {
    const double mean = 100.0 * (double)cnt / (double)len;
    if (len <= a.ws) return mean;  // src/read.cpp:217-218
    if (mn < a.clamp) mn = 0.0;
    return 100.0 * mn;
}

// MODE 0: parent only (no --trim/--split).  MODE 1: parent + count children, bit by bit.  MODE 2: emit children.
// MODE 4: emit children with the zero-run events found at word level and a branch-light bit loop (same condition as 3).
// MODE 3: parent + count children at WORD level — a zero run can only be a bad range if it starts at position 0, reaches
// the end of the read, or is at least --split long; with --split >= 32 (or unset) every such run crosses a 32-bit word
// boundary, so the runs that lie inside one word never matter and the parent keeps MODE 0's branch-free steady state.
// MODE 5: the word-level events of MODE 3 once more, without any floating point: writes every child's (start, end) and its
// read's index at the child's place in the CSR.  MODE 6: ONE LANE PER CHILD, children in descending order of length — a child
// is a read of its own (src/read.cpp:131-137: Read(child name, seq + start, ...)), so its lane runs MODE 0's branch-free
// recurrence on the parent's coverage bits [start, end) (the row words funnel-shifted by start mod 32) and writes the child's
// mean / window / pass flag.  5 + 6 replace MODE 4, whose 32 predicated positions per word carry the event machinery through
// every bit (67 of the 98 ms per 10^11 positions of C4's folds).
#ifndef FLX_FOLD_WALK_COPIES
#define FLX_FOLD_WALK_COPIES 1  // copies of the integer-grid fold's walk table in LDS (a power of two; 1, 8, 16 measured: no difference — the gathers are not what bounds the kernel)
#endif
#ifdef FLX_FOLD_WAVES_PER_EU  // (A/B builds: the register budget of the fold kernels)
#define FLX_FOLD_OCC __attribute__((amdgpu_waves_per_eu(FLX_FOLD_WAVES_PER_EU)))
#else
#define FLX_FOLD_OCC
#endif
template <int MODE, bool RING, bool GRID = false>
__global__ void __launch_bounds__(256) FLX_FOLD_OCC k_kmer_fold(const FoldArgs a) {
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = slot < (MODE == 6 ? a.n_children : a.n_reads);
    uint32_t rid = 0;  // MODE 6: the child's index
    int L = 0;
    const uint32_t *row = a.cov;
    int row_words_left = 0;       // MODE 6: words of the parent's row from `row` on
    uint32_t bit_off = 0;         // MODE 6: the child starts at bit `bit_off` (0..127) of row[0]
    if (live && MODE != 6) {
        rid = a.order ? a.order[slot] : (uint32_t)slot;
        L = a.lengths[rid];
        row = a.cov + (a.cov_off[rid] >> 2);
    }
    if (live && MODE == 6) {
        rid = a.child_order[slot];
        const uint32_t parent = a.child_parent[rid];
        const int start = a.child_ranges[2 * (size_t)rid], end = a.child_ranges[2 * (size_t)rid + 1];
        L = end - start;
        const int base_word = (start >> 7) << 2;  // 16-byte aligned piece of the row the child starts in
        row = a.cov + (a.cov_off[parent] >> 2) + base_word;
        row_words_left = ((a.lengths[parent] + 31) >> 5) - base_word;
        bit_off = (uint32_t)(start - 32 * base_word);
    }
    int Lmax = L;
    for (int o = 32; o > 0; o >>= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o, 64));
    const int ws = a.ws;
    const double delta = a.delta;

    Win P = {0, 0.0, 0.0};
    // child machinery
    Win C = {0, 0.0, 0.0};
    Win S = {0, 0.0, 0.0};  // snapshot of C at the start of the current zero run
    int cs = 0;             // start of the current child candidate
    int zs = -1;            // start of the current zero run (-1: none)
    bool any_bad = false;
    uint32_t nchild = 0;
    const uint64_t cbase = ((MODE == 2 || MODE == 4 || MODE == 5) && live) ? a.child_offsets[rid] : 0;
    const bool split_set = a.p.split_set != 0;
    const bool trim = a.p.trim != 0;
    const int split = a.p.split;

    auto emit_child = [&](int start, int end, const Win &st) {
        if (end <= start) return;
        if (MODE == 5) {
            const uint64_t at = cbase + nchild;
            a.child_ranges[2 * at] = start;
            a.child_ranges[2 * at + 1] = end;
            a.child_parent[at] = rid;
        }
        if (MODE == 3 && a.inline_ranges && nchild < (uint32_t)kInlineChildren) {
            int32_t *slot = a.inline_ranges + ((size_t)rid * kInlineChildren + nchild) * 2;
            slot[0] = start;
            slot[1] = end;
        }
        if (MODE == 2 || MODE == 4) {
            const int len = end - start;
            const double mean = 100.0 * (double)st.cnt / (double)len;
            const double window = window_result(a, len, st.cnt, st.mn);
            const uint64_t at = cbase + nchild;
            a.child_ranges[2 * at] = start;
            a.child_ranges[2 * at + 1] = end;
            a.child_mean_q[at] = mean;
            a.child_window_q[at] = window;
            a.child_passed[at] = cutoffs(a.p, len, mean, window);
        }
        ++nchild;
    };

    // Two word streams over the read's coverage row — the leading edge (position j) and the trailing edge (position
    // j - ws).  Each lane walks its own row: 64 lanes = 64 distinct lines per load instruction, and the rows of all resident
    // lanes do not fit L1 / L2 together, so a line is gone again before the lane comes back to it — every load of a new piece
    // is a far request (55 G/s, profiles/r03_microbench.txt).
    //   RING (default): the row is read ONCE, 64 bytes per lane at a time (four 16-byte loads issued back to back to one
    //   half line, a block ahead of their use), and parked in a per-lane ring of words in LDS (word k of lane l at
    //   ((k mod R) * 64 + l): every access of a wave is conflict free and touches only the lane's own words, so no barrier).
    //   Both edges then come out of the ring with ds_read_b32: one far request per 512 positions instead of two per 128,
    //   which had made the folds request bound (round 2: 28 of the 36 ms per 10^11 positions).
    //   !RING: both streams straight from global memory in 16-byte blocks (windows too long for the ring).
    const int n_words = MODE == 6 ? row_words_left : (L + 31) >> 5;  // words of the row that exist behind `row`
    const int o5 = (int)(bit_off >> 5);                                // MODE 6: the child's word k = row words k + o5, k + o5 + 1 ...
    const unsigned o = bit_off & 31u;                                  // ... shifted right by o bits
    auto ldq = [&](int b) -> uint4 {
        return (b * 4 < n_words) ? *reinterpret_cast<const uint4 *>(row + 4 * (size_t)b) : make_uint4(0u, 0u, 0u, 0u);
    };
    struct WStream { uint4 cur, nxt; int blk; };
    auto advance = [&](WStream &st, int b) {  // streams only move forward, one block at a time
        if (b != st.blk) {
            st.cur = st.nxt;
            st.blk = b;
            st.nxt = ldq(b + 1);
        }
    };
    auto word = [&](const WStream &st, int wi) -> uint32_t {  // wi inside block st.blk or st.blk + 1
        // (selects, no reference to one of the two blocks: a reference makes the compiler keep the stream in scratch memory)
        const bool cur = (wi >> 2) == st.blk;
        const int c = wi & 3;
        const uint32_t x = cur ? st.cur.x : st.nxt.x, y = cur ? st.cur.y : st.nxt.y, z = cur ? st.cur.z : st.nxt.z, w = cur ? st.cur.w : st.nxt.w;
        const uint32_t v = c == 0 ? x : c == 1 ? y : c == 2 ? z : w;
        return wi < n_words ? v : 0u;  // the padding of the last block is not coverage
    };
    WStream lead = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), 0};
    if (!RING) lead = {ldq(0), ldq(1), 0};
    WStream trail = lead;
    extern __shared__ uint32_t fold_ring[];
    const int R = a.ring_words;  // power of two >= 16 + ceil(ws / 32) + 2
    // GRID: in front of the rings the table of the +-1 walk of four positions, indexed by (new nibble << 4 | old nibble): two dwords,
    // {lowest prefix, -(highest prefix)} and {total, -total} as pairs of 16-bit integers (packed adds and minima carry both at once)
    // ... and behind it the grid table itself (GridTab: d* and the lower bound per binade, 4 dwords per entry): a regime begins in
    // the middle of the steady state, and a load from the kernel's arguments there costs the whole wave a trip to memory
    // (kWalkCopies copies of the walk table, entry idx of copy c at (idx * copies + c): a lane reads copy lane % copies, so that
    // lanes with different nibble pairs rarely meet in one bank — with one copy the 64 lanes of a gather share 32 bank pairs)
    constexpr int kWalkCopies = FLX_FOLD_WALK_COPIES;
    constexpr int kGridTabAt = 512 * kWalkCopies;  // dword index of the grid table
    constexpr int kWalkWords = GRID ? kGridTabAt + 8 * 32 : 0;
    if (GRID) {
        for (int i = threadIdx.x; i < GridTab::kMax; i += blockDim.x) {
            fold_ring[kGridTabAt + 8 * i + 0] = (uint32_t)__double2loint(a.gt.dstar[i]);
            fold_ring[kGridTabAt + 8 * i + 1] = (uint32_t)__double2hiint(a.gt.dstar[i]);
            fold_ring[kGridTabAt + 8 * i + 2] = (uint32_t)__double2loint(a.gt.lv[i]);
            fold_ring[kGridTabAt + 8 * i + 3] = (uint32_t)__double2hiint(a.gt.lv[i]);
            fold_ring[kGridTabAt + 8 * i + 4] = (uint32_t)a.gt.top[i];
        }
        for (int idx = threadIdx.x; idx < 256; idx += blockDim.x) {
            int t = 0, mp = 0, xp = 0;
            for (int i = 0; i < 4; ++i) {
                t += ((idx >> (4 + i)) & 1) - ((idx >> i) & 1);
                mp = min(mp, t);
                xp = max(xp, t);
            }
            for (int c = 0; c < kWalkCopies; ++c) {
                fold_ring[2 * (idx * kWalkCopies + c)] = ((uint32_t)mp & 0xffffu) | ((uint32_t)(-xp) << 16);
                fold_ring[2 * (idx * kWalkCopies + c) + 1] = ((uint32_t)t & 0xffffu) | ((uint32_t)(-t) << 16);
            }
        }
        __syncthreads();
    }
    uint32_t *ring = fold_ring + kWalkWords + (size_t)(threadIdx.x >> 6) * (size_t)R * 64 + (threadIdx.x & 63);
    uint4 nq[4];       // the block after the newest one in the ring
    int have_blk = -1;  // newest block in the ring (wave-uniform: every lane is at the same position)
    if (RING) {
#pragma unroll
        for (int q = 0; q < 4; ++q) nq[q] = ldq(q);
    }
    auto ring_fill = [&](int wi) {  // word wi (and everything up to the end of its block) into the ring; wi only moves forward
        const int b = wi >> 4;
        if (b > have_blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = (b * 16 + 4 * q) & (R - 1);
                ring[(k + 0) * 64] = nq[q].x; ring[(k + 1) * 64] = nq[q].y; ring[(k + 2) * 64] = nq[q].z; ring[(k + 3) * 64] = nq[q].w;
            }
            have_blk = b;
#pragma unroll
            for (int q = 0; q < 4; ++q) nq[q] = ldq((b + 1) * 4 + q);
        }
    };
    auto ring_word = [&](int wi) -> uint32_t { return wi < n_words ? ring[(wi & (R - 1)) * 64] : 0u; };
    // MODE 6: bit p of the child is bit p + bit_off of the row, so 32 child positions from position p on are row words
    // (p + bit_off) / 32 and the next one, funnel-shifted by (p + bit_off) mod 32
    auto lead_bits = [&](int p) -> uint32_t {  // p a multiple of 32, moving forward
        const int w = (p >> 5) + o5;
        uint32_t lo, hi;
        if (RING) { ring_fill(w + 1); lo = ring_word(w); hi = ring_word(w + 1); }
        else { advance(lead, w >> 2); lo = word(lead, w); hi = word(lead, w + 1); }
        return __builtin_amdgcn_alignbit(hi, lo, o);
    };
    auto trail_bits = [&](int p) -> uint32_t {  // any p >= 0 behind the leading edge, moving forward
        const int t = p + (int)bit_off, w = t >> 5;
        uint32_t lo, hi;
        if (RING) { lo = ring_word(w); hi = ring_word(w + 1); }
        else { advance(trail, w >> 2); lo = word(trail, w); hi = word(trail, w + 1); }
        return __builtin_amdgcn_alignbit(hi, lo, (unsigned)(t & 31));
    };
    auto lead_word = [&](int wi) -> uint32_t {  // the word holding the leading edge
        if (MODE == 6) return lead_bits(wi << 5);
        if (RING) { ring_fill(wi); return ring_word(wi); }
        advance(lead, wi >> 2);
        return word(lead, wi);
    };
    auto trail_word = [&](int wi) -> uint32_t {  // words of the trailing edge: never ahead of the leading one
        if (MODE == 6) return trail_bits(wi << 5);
        if (RING) return ring_word(wi);
        advance(trail, wi >> 2);
        return word(trail, wi);
    };
    uint32_t lead_w = 0, trail_w = 0;
    int Lmin = live ? L : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) Lmin = min(Lmin, __shfl_xor(Lmin, o, 64));
    const unsigned int d_lo = (unsigned int)(__double_as_longlong(delta) & 0xffffffffll);
    const unsigned int d_hi = (unsigned int)(__double_as_longlong(delta) >> 32);
    // GRID: the regime of this lane's parent window (w = g_wb + g_c * g_ds while it holds; lowest c so far in g_cmin)
    double g_wb = 0.0, g_ds = 0.0;
    int g_c = 0, g_cmin = 0, g_lo = 0x7fffffff, g_hi = (int)0x80000000;
    bool grid_on = false;  // wave-uniform: the steady state has begun (and not ended)
    auto grid_flush = [&]() {  // the regime's state as the recurrence's: exact, every operand is a multiple of the regime's grid
        P.mn = fmin(P.mn, fma((double)g_cmin, g_ds, g_wb));
        P.w = fma((double)g_c, g_ds, g_wb);
    };
    auto grid_begin = [&]() {  // a regime from P.w on (tools/sim_fold_grid.cpp: begin_regime — the same arithmetic)
        g_wb = P.w;
        g_ds = 0.0;
        g_c = 0;
        g_cmin = 0;
        g_lo = 0x7fffffff;
        g_hi = (int)0x80000000;
        const int eb = (__double2hiint(P.w) >> 20) & 0x7ff;
        const int idx = eb - a.gt.e0;
        if (P.w > 0.0 && idx >= 0 && idx < a.gt.n) {
            const uint4 e = *reinterpret_cast<const uint4 *>(fold_ring + kGridTabAt + 8 * idx);
            const double ds = __hiloint2double((int)e.y, (int)e.x), lv = __hiloint2double((int)e.w, (int)e.z);
            if (ds > 0.0) {
                // The top: w must not reach a binade on whose grid w_b does NOT lie (its low bits would be rounded away there).  w_b
                // lies on the grid of binade eb + z, z = the trailing zero bits of its mantissa — a window that has once been full
                // (w = 1.0) stays on the grid of [1, 2) whatever is subtracted, d* is a multiple of it — up to the top of the group.
                const uint32_t m_lo = (uint32_t)__double2loint(P.w), m_hi = ((uint32_t)__double2hiint(P.w) & 0xfffffu) | 0x100000u;
                const int z = m_lo ? __ffs((int)m_lo) - 1 : 32 + (__ffs((int)m_hi) - 1);
                const int gb = min(eb + z, (int)fold_ring[kGridTabAt + 8 * idx + 4]);
                const double uv = __hiloint2double((gb + 1) << 20, 0);
                // smallest c with w + c d* > lv: the estimate's floor is the answer or up to two below it; the values decide
                int k0 = (int)floor((lv - P.w) * a.ws_d);
                if (fma((double)k0, ds, P.w) <= lv) ++k0;
                if (fma((double)k0, ds, P.w) <= lv) ++k0;
                // largest c with w + c d* < uv
                int k1 = (int)ceil((uv - P.w) * a.ws_d);
                if (fma((double)k1, ds, P.w) >= uv) --k1;
                if (fma((double)k1, ds, P.w) >= uv) --k1;
                g_ds = ds;
                g_lo = k0;
                g_hi = k1;
            }
        }
    };
    for (int j0 = 0; j0 < Lmax; j0 += 32) {
      lead_w = lead_word(j0 >> 5);
      if (MODE == 4) {
          // ---- children only: events per word, then 32 predicated positions ----
          int ev_end = -1, ev_zs = 0, snap = -1;
          if (j0 < L) {
              const int v = min(32, L - j0);
              const uint32_t w = v < 32 ? (lead_w & ((1u << v) - 1u)) : lead_w;
              if (j0 == 0 && !(w & 1u)) zs = 0;  // S is still the initial (empty) state
              if (w != 0) {
                  if (zs >= 0) {
                      const int f = __ffs(w) - 1;
                      const bool bad = (split_set && j0 + f - zs >= split) || (trim && zs == 0);
                      if (bad) { ev_end = f; ev_zs = zs; }
                      zs = -1;
                  }
                  const int top = 32 - __clz(w);
                  if (top < v) { zs = j0 + top; snap = top; }
              } else if (zs < 0) {
                  zs = j0;
                  snap = 0;
              }
          }
          // trailing window of the 32 positions (positions before the read count as uncovered; they are never used,
          // a child's trailing edge lies inside the child)
          const int tj0 = j0 - ws;
          uint32_t tw = 0;
          if (tj0 > -32) {
              const int twi = tj0 >> 5, sh = tj0 & 31;  // twi == -1 for the word that straddles position 0
              const uint32_t lo = twi >= 0 ? trail_word(twi) : 0u;
              tw = sh ? __builtin_amdgcn_alignbit(trail_word(twi + 1), lo, (unsigned)sh) : lo;
          }
#pragma unroll
          for (int i = 0; i < 32; ++i) {
              const int j = j0 + i;
              if (i == ev_end) {  // a bad range ended here: the child [cs, ev_zs) is complete, a new one starts at j
                  any_bad = true;
                  emit_child(cs, ev_zs, S);
                  cs = j;
                  C.cnt = 0;
                  C.w = 0.0;
                  C.mn = 0.0;
              }
              if (i == snap) {  // state of the current child at the start of a zero run that may turn out bad
                  S.cnt = C.cnt;
                  S.mn = C.mn;
              }
              const bool act = j < L;
              const int k = j - cs;
              const int ml = __builtin_amdgcn_sbfe((int)lead_w, i, 1);  // 0 or -1 (bits beyond L are 0)
              const int mt = __builtin_amdgcn_sbfe((int)tw, i, 1);
              C.cnt -= ml;
              if (act && k == ws - 1) {
                  C.w = (double)C.cnt / a.ws_d;
                  C.mn = C.w;
              }
              const bool steady = act && k >= ws;
              const int ms = steady ? -1 : 0;
              const double dl = __hiloint2double((int)(d_hi & (unsigned)(ml & ms)), (int)(d_lo & (unsigned)(ml & ms)));
              const double dt = __hiloint2double((int)(d_hi & (unsigned)(mt & ms)), (int)(d_lo & (unsigned)(mt & ms)));
              C.w -= dt;  // exact no-ops outside the steady state
              C.w += dl;
              const double m2 = fmin(C.mn, C.w);
              C.mn = steady ? m2 : C.mn;
          }
          continue;
      }
      if ((MODE == 3 || MODE == 5) && j0 < L) {
          const int v = min(32, L - j0);  // valid bits of this word
          const uint32_t w = v < 32 ? (lead_w & ((1u << v) - 1u)) : lead_w;
          if (j0 == 0 && !(w & 1u)) zs = 0;  // the read starts inside a zero run
          if (w != 0) {
              if (zs >= 0) {  // the run [zs, j) that reached this word ends at its first covered base
                  const int j = j0 + __ffs(w) - 1;
                  const bool bad = (split_set && j - zs >= split) || (trim && zs == 0);
                  if (bad) {
                      any_bad = true;
                      emit_child(cs, zs, S);  // (counts it; MODE 5 also writes its range)
                      cs = j;
                  }
                  zs = -1;
              }
              const int top = 32 - __clz(w);  // one past the last covered base of the word
              if (top < v) zs = j0 + top;     // the word ends inside a new zero run
          } else if (zs < 0) {
              zs = j0;
          }
      }
      if (MODE == 5) continue;
      if ((MODE == 0 || MODE == 3 || MODE == 6) && j0 + 32 <= ws - 1 && j0 + 32 <= Lmin) {
          // ---- the head, a word at a time: every position of the word lies in front of the first full window (j < ws - 1) and inside
          // every read of the wave — the recurrence has not begun, only the covered bases are counted (src/read.cpp:221-225).  (The
          // per-bit loop below spent ~20 instructions on each of these positions: a fifth of a 10 kbp read's fold once the steady state
          // ran on the integer grid.)
          P.cnt += __popc(lead_w);
          continue;
      }
      if ((MODE == 0 || MODE == 3 || MODE == 6) && j0 >= ws && j0 + 32 <= Lmin) {
          // ---- steady state, one window per lane: 32 positions, every lane active, no per-bit control flow ----
          uint32_t tw;
          if (MODE == 6) {
              tw = trail_bits(j0 - ws);
          } else {
              const int tj0 = j0 - ws, sh = tj0 & 31, twi = tj0 >> 5;
              const uint32_t t0 = trail_word(twi);
              tw = sh ? __builtin_amdgcn_alignbit(trail_word(twi + 1), t0, (unsigned)sh) : t0;
          }
          P.cnt += __popc(lead_w);
          if (GRID) {
              if (!grid_on) {
                  grid_begin();
                  grid_on = true;
              }
              typedef short s16x2 __attribute__((ext_vector_type(2)));
              // the (new, old) nibble pairs of the word: byte k of `even` = nibbles 2k, of `odd` = nibbles 2k + 1
              const uint32_t odd = (lead_w & 0xF0F0F0F0u) | ((tw >> 4) & 0x0F0F0F0Fu);
              const uint32_t even = ((lead_w << 4) & 0xF0F0F0F0u) | (tw & 0x0F0F0F0Fu);
              const uint2 *walk = reinterpret_cast<const uint2 *>(fold_ring) + (threadIdx.x & (kWalkCopies - 1));
              s16x2 run = {0, 0}, ext = {0, 0};  // {prefix, -prefix} so far; {lowest prefix, -(highest prefix)}
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                  const uint32_t idx = ((k & 1 ? odd : even) >> (8 * (k >> 1))) & 0xffu;
                  const uint2 e = walk[idx * kWalkCopies];
                  ext = __builtin_elementwise_min(ext, run + __builtin_bit_cast(s16x2, e.x));
                  run = run + __builtin_bit_cast(s16x2, e.y);
              }
              const int mp = ext.x, xp = -(int)ext.y, t = run.x;
              // not a no-op (a word of zeros on both edges changes nothing in any regime) and outside the regime: this lane's word in FP
#ifdef FLX_FOLD_GRID_NOSLOW  // (timing experiment only: wrong results)
              const bool slow = false;
#else
              const bool slow = (lead_w | tw) != 0u && !(g_c + mp >= g_lo && g_c + xp <= g_hi);
#endif
              if (!__any(slow)) {
                  g_cmin = min(g_cmin, g_c + mp);
                  g_c += t;
                  continue;
              }
              if (!slow) {
                  g_cmin = min(g_cmin, g_c + mp);
                  g_c += t;
                  lead_w = tw = 0;  // (32 exact no-ops below)
              } else {
                  grid_flush();
              }
              // (four rounds of eight steps, not 32 unrolled: unrolled, the compiler converts all 64 bits to doubles ahead of the chain
              // and the kernel needs 122 registers, or spills)
#pragma unroll 1
              for (int i0 = 0; i0 < 32; i0 += 8) {
#pragma unroll
                  for (int i = 0; i < 8; ++i) {
                      const double lb = (double)__builtin_amdgcn_ubfe(lead_w, i0 + i, 1);
                      const double tb = (double)__builtin_amdgcn_ubfe(tw, i0 + i, 1);
                      P.w = fma(tb, -delta, P.w);
                      P.w = fma(lb, delta, P.w);
                      P.mn = fmin(P.mn, P.w);
                  }
              }
              if (slow) grid_begin();
              continue;
          }
          if (a.events) {
              // Round-3 review, item 5: only the positions where the two edges DIFFER change w for certain (one exact step each);
              // where both are 0 nothing happens, and where both are 1 the step is fl(fl(w - d) + d), which is w itself unless the
              // subtraction leaves w's binade — checked once per stretch of such positions, with the 32-step loop below as the
              // fallback for a word where it fails.  Lanes diverge (a wave runs as many rounds as its busiest lane has events).
              const double w0 = P.w, mn0 = P.mn;
              const uint32_t both = lead_w & tw;
              uint32_t ev = lead_w ^ tw, handled = 0;
              bool slow = false;
              for (;;) {
                  const int i = ev ? __ffs(ev) - 1 : 32;
                  const uint32_t upto = i == 32 ? 0xffffffffu : ((1u << i) - 1u);
                  if (both & upto & ~handled) {
                      double t = P.w - delta;
                      t = t + delta;
                      if (t != P.w) { slow = true; break; }
                  }
                  if (i == 32) break;
                  if ((lead_w >> i) & 1u) {
                      P.w = P.w + delta;
                  } else {
                      P.w = P.w - delta;
                      P.mn = fmin(P.mn, P.w);
                  }
                  handled = upto | (1u << i);
                  ev &= ev - 1;
              }
              if (!__any(slow)) continue;
              if (!slow) {
                  lead_w = tw = 0;  // (this lane is done with the word: 32 exact no-ops below)
              } else {
                  P.w = w0;
                  P.mn = mn0;
              }
          }
#pragma unroll
          for (int i = 0; i < 32; ++i) {
#if FLX_FOLD_FMA
              // w - q[j-ws]/ws and + q[j]/ws with q in {0.0, 1.0} (src/read.cpp:228-229) as fma(bit, -+delta, w): the product is exact
              // (0 or delta), so the one rounding of the fma is the rounding of the reference's subtraction / addition, and
              // adding a zero product leaves w as it is.  7 VALU instructions per position instead of 9 — the folds are VALU bound.
              const double lb = (double)__builtin_amdgcn_ubfe(lead_w, i, 1);
              const double tb = (double)__builtin_amdgcn_ubfe(tw, i, 1);
              P.w = fma(tb, -delta, P.w);
              P.w = fma(lb, delta, P.w);
#else
              const int ml = __builtin_amdgcn_sbfe((int)lead_w, i, 1);  // 0 or -1
              const int mt = __builtin_amdgcn_sbfe((int)tw, i, 1);
              const double dl = __hiloint2double((int)(d_hi & (unsigned)ml), (int)(d_lo & (unsigned)ml));
              const double dt = __hiloint2double((int)(d_hi & (unsigned)mt), (int)(d_lo & (unsigned)mt));
              P.w -= dt;
              P.w += dl;
#endif
              P.mn = fmin(P.mn, P.w);
          }
          continue;
      }
      if (GRID && grid_on) {  // the steady state is over (the shortest read of the wave ends inside this word): back to the recurrence's own state
          grid_flush();
          grid_on = false;
          g_ds = 0.0;
          g_c = g_cmin = 0;
          g_wb = P.w;
      }
      for (int jj = 0; jj < 32; ++jj) {
        const int j = j0 + jj;
        if (j >= Lmax) break;
        const int tj = j - ws;
        if (tj >= 0 && ((tj & 31) == 0 || jj == 0)) {
            trail_w = trail_word(tj >> 5);
        }
        const bool act = j < L;
        const uint32_t b = act ? ((lead_w >> (j & 31)) & 1u) : 0u;
        const uint32_t tb = (act && tj >= 0) ? ((trail_w >> (tj & 31)) & 1u) : 0u;
        const double dl = b ? delta : 0.0;
        const double dt = tb ? delta : 0.0;

        if (act) {
            // ---- parent window (src/read.cpp:216-236) ----
            P.cnt += (int)b;
            if (j == ws - 1) {
                P.w = (double)P.cnt / a.ws_d;
                P.mn = P.w;
            } else if (j >= ws) {
                P.w -= dt;
                P.w += dl;
                if (P.w < P.mn) P.mn = P.w;
            }
            if (MODE == 1 || MODE == 2) {
                // ---- zero runs -> bad ranges -> children (src/read.cpp:89-141) ----
                if (b == 0 && zs < 0) {
                    zs = j;
                    S = C;
                }
                if (b == 1 && zs >= 0) {  // the run [zs, j) has ended
                    const bool bad = (split_set && j - zs >= split) || (trim && zs == 0);
                    if (bad) {
                        any_bad = true;
                        emit_child(cs, zs, S);
                        cs = j;
                        C.cnt = 0;
                        C.w = 0.0;
                        C.mn = 0.0;
                    }
                    zs = -1;
                }
                const int k = j - cs;  // position inside the current child
                C.cnt += (int)b;
                if (k == ws - 1) {
                    C.w = (double)C.cnt / a.ws_d;
                    C.mn = C.w;
                } else if (k >= ws) {
                    C.w -= dt;
                    C.w += dl;
                    if (C.w < C.mn) C.mn = C.w;
                }
            }
        }
      }
    }
    if (GRID && grid_on) grid_flush();
    if (!live) return;

    if (MODE == 6) {  // the child's own scores (src/read.cpp:131-137 -> the Read constructor's folds on the child's slice)
        const double mean = 100.0 * (double)P.cnt / (double)L;
        const double window = window_result(a, L, P.cnt, P.mn);
        a.child_mean_q[rid] = mean;
        a.child_window_q[rid] = window;
        a.child_passed[rid] = cutoffs(a.p, L, mean, window);
        return;
    }
    if (MODE != 0) {
        int end = L;
        if (zs >= 0) {  // the read ends inside a zero run [zs, L)
            const bool bad = (split_set && L - zs >= split) || (trim && zs > 0);
            if (bad) {
                any_bad = true;
                end = zs;
                C = S;
            }
        }
        if (any_bad) emit_child(cs, end, C);
        else nchild = 0;
    }
    if (MODE == 1 || MODE == 3) a.n_child[rid] = nchild;
    if (MODE != 2 && MODE != 4 && MODE != 5) {
        const double mean = 100.0 * (double)a.count[rid] / (double)L;  // exact: the qualities are 0.0 / 1.0
        const double window = window_result(a, L, P.cnt, P.mn);
        a.mean_q[rid] = mean;
        a.window_q[rid] = window;
        a.passed[rid] = cutoffs(a.p, L, mean, window);
    }
}

__global__ void k_cov_row_bytes(uint64_t n, const int32_t *lengths, int64_t *row_bytes) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row_bytes[i] = (int64_t)((((uint64_t)lengths[i] + 7) / 8 + 15) & ~15ull);
}

__global__ void k_widen_u32_i64(uint64_t n, const uint32_t *in, int64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// the ranges MODE 3 left inline -> their places in the CSR (+ every child's read); counts the reads that have more than fit inline
__global__ void __launch_bounds__(256) k_children_from_inline(uint64_t n, const uint32_t *n_child, const uint64_t *child_offsets, const int32_t *inline_ranges,
                                                              int32_t *child_ranges, uint32_t *child_parent, unsigned int *overflow) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t nc = n_child[i];
    if (nc == 0) return;
    if (nc > (uint32_t)kInlineChildren) {
        atomicAdd(overflow, 1u);
        return;
    }
    const uint64_t at = child_offsets[i];
    const int32_t *src = inline_ranges + (size_t)i * kInlineChildren * 2;
    for (uint32_t k = 0; k < nc; ++k) {
        child_ranges[2 * (at + k)] = src[2 * k];
        child_ranges[2 * (at + k) + 1] = src[2 * k + 1];
        child_parent[at + k] = (uint32_t)i;
    }
}

// sort key of a child: longest first (the lanes of a wave then run the same number of steps)
__global__ void k_child_keys(uint64_t n, const int32_t *ranges, uint64_t *keys, uint32_t *vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        keys[i] = (uint64_t)(0x7fffffffu - (uint32_t)(ranges[2 * i + 1] - ranges[2 * i]));
        vals[i] = (uint32_t)i;
    }
}

}  // namespace

// the fold kernels: with the LDS ring when it fits (R words per lane; 4 waves per workgroup up to R = 64, one wave up to R = 512),
// else (windows beyond ~15 000 positions) both streams from global memory
// The grid table of a window size (GridTab, above): per binade of w the step d rounded to that binade's grid, the groups of binades
// that share it, and whether the integer-grid steady state pays — the group that holds [1, 2) must reach down to 2^-3 at least (ws =
// 250: 2^-4; ws = 1000: d rounds differently on either side of 1.0, where the window of a clean read sits — the FP kernel then).
static bool build_grid_table(int ws, GridTab &g) {
    memset(&g, 0, sizeof g);
    if (ws < 8 || ws > (1 << 20)) return false;
    volatile double one = 1.0, wsd = (double)ws;
    const double delta = one / wsd;
    uint64_t bits;
    memcpy(&bits, &delta, 8);
    const int e_d = (int)((bits >> 52) & 0x7ff) - 1023;
    const uint64_t M = (bits & ((1ull << 52) - 1)) | (1ull << 52);  // delta = M * 2^(e_d - 52)
    const int e_min = e_d - 2, e_max = 1;
    const int n = e_max - e_min + 1;
    if (n > GridTab::kMax) return false;
    bool tie[GridTab::kMax];
    for (int i = 0; i < n; ++i) {
        const int shift = (e_min + i) - e_d;  // the grid of binade E is 2^shift ulps of delta
        tie[i] = false;
        if (shift <= 0) {
            g.dstar[i] = delta;
        } else {
            const uint64_t rem = M & ((1ull << shift) - 1), half = 1ull << (shift - 1);
            tie[i] = rem == half;
            g.dstar[i] = ldexp((double)((M >> shift) + (rem > half ? 1 : 0)), shift + e_d - 52);
        }
    }
    bool pays = false;
    for (int i = 0; i < n;) {  // groups: maximal runs of binades without a tie that share d*
        if (tie[i]) { g.dstar[i] = 0.0; g.lv[i] = 0.0; ++i; continue; }
        int j = i;
        while (j + 1 < n && !tie[j + 1] && g.dstar[j + 1] == g.dstar[i]) ++j;
        const double bottom = std::max(ldexp(1.0, e_min + i), 4.0 * delta);
        for (int k = i; k <= j; ++k) {
            g.lv[k] = bottom;
            g.top[k] = 1023 + e_min + j;
        }
        if (e_min + i <= -3 && e_min + j >= 0) pays = true;
        i = j + 1;
    }
    g.e0 = 1023 + e_min;
    g.n = n;
    return pays;
}

template <int MODE>
static int launch_fold(flx_ctx *ctx, FoldArgs &a) {
    int R = 32;
    while (R < (MODE == 6 ? 24 : 18) + (a.ws + 31) / 32) R *= 2;  // MODE 6 starts up to 4 words into its first block, and reads one word further
    const char *env = getenv("FLX_KMER_FOLD_STREAMS");  // "global": the round-2 data path (second implementation, tests)
    const bool ring = R <= 512 && !(env && strcmp(env, "global") == 0);
    const unsigned threads = (!ring || R <= 64) ? 256u : 64u;
    const unsigned nb = (unsigned)(((MODE == 6 ? a.n_children : a.n_reads) + threads - 1) / threads);
    a.ring_words = R;
    if (ring && a.grid && !a.events && (MODE == 0 || MODE == 3 || MODE == 6)) {  // the steady state on the integer grid (GridTab)
        constexpr int M = (MODE == 0 || MODE == 3 || MODE == 6) ? MODE : 0;
        ctx->last_kmer_fold_grid = true;
        const size_t lds = (size_t)(threads / 64) * (size_t)R * 256 + 2048 * FLX_FOLD_WALK_COPIES + 1024;
        FLX_HIP(ctx, hipFuncSetAttribute((const void *)k_kmer_fold<M, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_kmer_fold<M, true, true>), dim3(nb), dim3(threads), lds, ctx->stream, a);
    } else if (ring) {
        const size_t lds = (size_t)(threads / 64) * (size_t)R * 256;
        FLX_HIP(ctx, hipFuncSetAttribute((const void *)k_kmer_fold<MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_kmer_fold<MODE, true>), dim3(nb), dim3(threads), lds, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((k_kmer_fold<MODE, false>), dim3(nb), dim3(threads), 0, ctx->stream, a);
    }
    FLX_HIP(ctx, hipGetLastError());  // (a launch that fails must not pass for a kernel that wrote nothing)
    return FLX_OK;
}

int flx_score_kmer_dev(flx_ctx *ctx, const flx_kmerset *set, const uint8_t *d_plane, uint64_t plane_bytes,
                       const uint64_t *d_offsets, const int32_t *d_lengths, const uint32_t *d_order,
                       uint64_t n_reads, const flx_params *params, flx_scores *out) {
    (void)plane_bytes;
    out->n_children = 0;
    if (n_reads == 0) return FLX_OK;
    hipStream_t st = ctx->stream;
    const bool want_children = params->trim || params->split_set;
    if (want_children && !out->child_offsets)
        return flx_fail(ctx, FLX_ERR_INVALID, "trim/split requested but child_offsets is NULL");
    const unsigned nb = (unsigned)((n_reads + 255) / 256);

    // ---- coverage plane layout: row i = ceil(L/8) bytes rounded up to 16, rows packed by an exclusive scan ----
    // All device memory of this path comes from the context's two grow-only workspaces: nothing is allocated or freed per
    // batch once they have reached their size.
    const size_t scan_ws = flx_radix_sort_workspace(n_reads + 1);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const char *fold_env0 = getenv("FLX_KMER_FOLD");
    const bool inline_children = want_children && !(params->split_set && params->split < 32) && !fold_env0;  // (the one-lane-per-child path)
    const size_t small_bytes = 2 * up((n_reads + 1) * 8) + 3 * up(n_reads * 4) + up((n_reads + 1) * 4) + up(scan_ws) + up(n_reads) +
                               (inline_children ? up(n_reads * (size_t)kInlineChildren * 8) + up(64) : 0);
    void *small = nullptr;
    FLX_CHECK(flx_workspace(ctx, 0, small_bytes, &small));
    char *wp = (char *)small;
    auto carve = [&](size_t bytes) { void *q = wp; wp += up(bytes); return q; };
    int64_t *d_rowb = (int64_t *)carve((n_reads + 1) * 8);
    int64_t *d_covoff = (int64_t *)carve((n_reads + 1) * 8);
    int32_t *d_cnt = (int32_t *)carve(n_reads * 4);
    int32_t *d_first_tmp = (int32_t *)carve(n_reads * 4);
    int32_t *d_last_tmp = (int32_t *)carve(n_reads * 4);
    uint32_t *d_nchild = (uint32_t *)carve((n_reads + 1) * 4);
    void *d_scanws = carve(scan_ws);
    uint8_t *d_redo = (uint8_t *)carve(n_reads);  // cover_queue.hip: marks of the reads handed to the kernel with a diagonal per lane
    int32_t *d_inline = inline_children ? (int32_t *)carve(n_reads * (size_t)kInlineChildren * 8) : nullptr;
    unsigned int *d_overflow = inline_children ? (unsigned int *)carve(64) : nullptr;
    int32_t *first = out->first ? out->first : d_first_tmp, *last = out->last ? out->last : d_last_tmp;
    FLX_HIP(ctx, hipMemsetAsync(d_rowb, 0, (n_reads + 1) * 8, st));
    hipLaunchKernelGGL(k_cov_row_bytes, dim3(nb), dim3(256), 0, st, n_reads, d_lengths, d_rowb);
    FLX_CHECK(flx_exclusive_scan_i64(ctx, n_reads + 1, d_rowb, d_covoff, d_scanws, scan_ws));
    int64_t cov_bytes = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&cov_bytes, d_covoff + n_reads, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    void *d_cov = nullptr;
    FLX_CHECK(flx_workspace(ctx, 1, (size_t)cov_bytes + 64, &d_cov));

    // ---- kernel 1: lookups -> coverage bits ----
    {
        const unsigned grid = (unsigned)std::min<uint64_t>(n_reads, 1u << 20);
        // FLX_KMER_COVER: "v2" = round 2's workgroup-per-read kernel, "w" = the wave-level kernel of rounds 3-5 for every set (second
        // and third implementation; default: k_kmer_cover_q, cover_queue.hip, for a set with a text, k_kmer_cover_w for one without),
        // "q2" = k_kmer_cover_q with EVERY read in its second launch (a diagonal per lane: tests)
        const char *cover_env = getenv("FLX_KMER_COVER");
        const bool old_cover = (cover_env && strcmp(cover_env, "v2") == 0) || !flx_kmerset_exact15(set);  // (no pair table: finalize found no room for it)
        const bool wave_cover = cover_env && strcmp(cover_env, "w") == 0;
        const bool second_only = cover_env && strcmp(cover_env, "q2") == 0;  // every read through the kernel with a diagonal per lane (tests)
        flx_time_scope tc(ctx, "flx_score_kmer_cover");
        ctx->last_kmer_locus = false;
        ctx->last_kmer_cover = "v2";
        ctx->last_kmer_redo = nullptr;
        ctx->last_kmer_redo_n = 0;
        if (!old_cover) {
            const unsigned wgrid = (unsigned)std::min<uint64_t>((n_reads + FLX_COVER_THREADS / 64 - 1) / (FLX_COVER_THREADS / 64), 1u << 22);
            const char *locus_env = getenv("FLX_KMER_LOCUS");  // "0": without the assembly text (the round-3 kernel; tests, A/B)
            const flx_locus *lp = (locus_env && locus_env[0] == '0') ? nullptr : flx_kmerset_locus(set);
            flx_locus none;
            memset(&none, 0, sizeof none);
            ctx->last_kmer_locus = lp != nullptr;
            const uint8_t *pre11 = flx_kmerset_pre11(set);
            CoverArgs ca = {d_plane, d_offsets, d_lengths, d_order, n_reads, flx_kmerset_exact15(set), pre11, lp ? *lp : none, (uint32_t *)d_cov, (const uint64_t *)d_covoff, d_cnt, first, last, d_redo};
            ctx->last_kmer_cover = (lp && !wave_cover) ? (second_only ? "q2" : "q") : "w";
            if (lp && !wave_cover) {
                ctx->last_kmer_redo = d_redo;
                ctx->last_kmer_redo_n = n_reads;
                const int rc = flx_cover_queue_launch(ctx, ca, pre11 != nullptr, wgrid, second_only);
                if (rc != FLX_OK) return rc;
            } else if (pre11 && lp)
                hipLaunchKernelGGL((k_kmer_cover_w<true, true>), dim3(wgrid), dim3(FLX_COVER_THREADS), 0, st, ca);
            else if (pre11)
                hipLaunchKernelGGL((k_kmer_cover_w<true, false>), dim3(wgrid), dim3(FLX_COVER_THREADS), 0, st, ca);
            else if (lp)
                hipLaunchKernelGGL((k_kmer_cover_w<false, true>), dim3(wgrid), dim3(FLX_COVER_THREADS), 0, st, ca);
            else
                hipLaunchKernelGGL((k_kmer_cover_w<false, false>), dim3(wgrid), dim3(FLX_COVER_THREADS), 0, st, ca);
        } else {
        hipLaunchKernelGGL(k_kmer_cover<256>, dim3(grid), dim3(256), 0, st, d_plane, d_offsets, d_lengths, d_order, n_reads,
                           flx_kmerset_bitmap(set), flx_kmerset_prefilter(set), (uint32_t *)d_cov, (const uint64_t *)d_covoff, d_cnt, first,
                           last);
        hipLaunchKernelGGL(k_kmer_cover<64>, dim3(grid), dim3(64), 0, st, d_plane, d_offsets, d_lengths, d_order, n_reads,
                           flx_kmerset_bitmap(set), flx_kmerset_prefilter(set), (uint32_t *)d_cov, (const uint64_t *)d_covoff, d_cnt, first,
                           last);
        }
    }

    // ---- kernel 2: serial fold ----
    FoldArgs a;
    a.cov = (uint32_t *)d_cov;
    a.cov_off = (const uint64_t *)d_covoff;
    a.lengths = d_lengths;
    a.order = d_order;
    a.n_reads = n_reads;
    a.count = d_cnt;
    a.first = first;
    a.last = last;
    a.ws = params->window_size;
    a.ws_d = (double)(size_t)params->window_size;
    {
        volatile double one = 1.0, half = 0.5, wsd = a.ws_d;
        a.delta = one / wsd;
        a.clamp = half / wsd;
    }
    a.p = *params;
    {
        const char *ev_env = getenv("FLX_KMER_FOLD_EVENTS");
        a.events = ev_env && ev_env[0] == '1';
        const char *grid_env = getenv("FLX_KMER_FOLD_GRID");  // "0": the floating-point steady state (the second implementation; tests, A/B)
        a.grid = build_grid_table(params->window_size, a.gt) && !(grid_env && grid_env[0] == '0');
        ctx->last_kmer_fold_grid = false;  // (launch_fold says so when a grid kernel really runs)
    }
    a.mean_q = out->mean_q;
    a.window_q = out->window_q;
    a.passed = out->passed;
    a.n_child = nullptr;
    a.inline_ranges = d_inline;
    a.child_parent = nullptr;
    a.child_order = nullptr;
    a.n_children = 0;
    a.child_offsets = nullptr;
    a.child_ranges = out->child_ranges;
    a.child_mean_q = out->child_mean_q;
    a.child_window_q = out->child_window_q;
    a.child_passed = out->child_passed;

    if (!want_children) {
        {
            flx_time_scope tf(ctx, "flx_score_kmer_fold");
            FLX_CHECK(launch_fold<0>(ctx, a));
        }
        if (out->child_offsets) FLX_HIP(ctx, hipMemsetAsync(out->child_offsets, 0, (n_reads + 1) * 8, st));
        FLX_HIP(ctx, hipGetLastError());
        FLX_HIP(ctx, hipStreamSynchronize(st));
        return FLX_OK;
    }

    FLX_HIP(ctx, hipMemsetAsync(d_nchild, 0, (n_reads + 1) * 4, st));
    a.n_child = d_nchild;
    // FLX_KMER_FOLD=bits forces the bit-level passes (tests compare the two implementations on every read)
    const char *fold_env = getenv("FLX_KMER_FOLD");
    const bool bit_level = (params->split_set && params->split < 32) || (fold_env && strcmp(fold_env, "bits") == 0);
    {
        flx_time_scope tf(ctx, "flx_score_kmer_fold");
        if (bit_level)
            FLX_CHECK(launch_fold<1>(ctx, a));  // runs inside one word can be bad ranges
        else
            FLX_CHECK(launch_fold<3>(ctx, a));
    }
    // child_offsets = exclusive scan of the counts (n + 1 entries; the last one is the total)
    hipLaunchKernelGGL(k_widen_u32_i64, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, st, n_reads + 1,
                       d_nchild, d_rowb);
    FLX_CHECK(flx_exclusive_scan_i64(ctx, n_reads + 1, d_rowb, (int64_t *)out->child_offsets, d_scanws, scan_ws));
    int64_t total_children = 0;
    FLX_HIP(ctx, hipMemcpyAsync(&total_children, (int64_t *)out->child_offsets + n_reads, 8, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    out->n_children = (uint64_t)total_children;
    if ((uint64_t)total_children > out->child_capacity)
        return flx_fail(ctx, FLX_ERR_CAPACITY, "child outputs need room for %lld children (capacity %llu)",
                        (long long)total_children, (unsigned long long)out->child_capacity);
    if (total_children > 0) {
        a.child_offsets = out->child_offsets;
        flx_time_scope tf(ctx, "flx_score_kmer_fold");  // (a scope: the checks below may return)
        // FLX_KMER_FOLD=words: the children inside their read's lane (MODE 4, second implementation of the word-level path)
        const bool per_child = !bit_level && !(fold_env && strcmp(fold_env, "words") == 0);
        if (bit_level) {
            FLX_CHECK(launch_fold<2>(ctx, a));
        } else if (!per_child) {
            FLX_CHECK(launch_fold<4>(ctx, a));
        } else {
            // ranges (word-level events only) -> children by descending length -> one lane per child
            const uint64_t nc = (uint64_t)total_children;
            const size_t sort_ws = flx_radix_sort_workspace(nc);
            void *cw = nullptr;
            FLX_CHECK(flx_workspace(ctx, 2, 2 * up(nc * 8) + 3 * up(nc * 4) + up(sort_ws), &cw));
            wp = (char *)cw;
            uint64_t *keys0 = (uint64_t *)carve(nc * 8), *keys1 = (uint64_t *)carve(nc * 8);
            uint32_t *vals0 = (uint32_t *)carve(nc * 4), *vals1 = (uint32_t *)carve(nc * 4);
            a.child_parent = (uint32_t *)carve(nc * 4);
            void *d_sortws = carve(sort_ws);
            a.n_children = nc;
            unsigned int overflow = 1;
            if (d_inline) {  // the ranges MODE 3 left inline go to their places; a read with more than fit sends the batch through MODE 5
                FLX_HIP(ctx, hipMemsetAsync(d_overflow, 0, 4, st));
                hipLaunchKernelGGL(k_children_from_inline, dim3(nb), dim3(256), 0, st, n_reads, (const uint32_t *)d_nchild, (const uint64_t *)out->child_offsets,
                                   (const int32_t *)d_inline, out->child_ranges, a.child_parent, d_overflow);
                FLX_HIP(ctx, hipMemcpyAsync(&overflow, d_overflow, 4, hipMemcpyDeviceToHost, st));
                FLX_HIP(ctx, hipStreamSynchronize(st));
            }
            if (overflow) FLX_CHECK(launch_fold<5>(ctx, a));
            tf.end();  // (the sort times its own passes)
            hipLaunchKernelGGL(k_child_keys, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, nc, out->child_ranges, keys0, vals0);
            uint64_t *skeys = nullptr;
            uint32_t *svals = nullptr;
            FLX_CHECK(flx_radix_sort_pairs(ctx, nc, keys0, keys1, vals0, vals1, d_sortws, sort_ws, &skeys, &svals));
            a.child_order = svals;
            flx_time_scope tf6(ctx, "flx_score_kmer_fold");
            FLX_CHECK(launch_fold<6>(ctx, a));
        }
    }
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(st));
    return FLX_OK;
}
