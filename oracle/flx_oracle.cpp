// flx_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A from-scratch CPU restatement of Filtlong's per-read scoring hot path and its
// global rank/cut stage.  Nothing under filtlong_amd/ (the product) may include,
// link, dlopen or execute this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py use it, and only as the checker.
//
// Parity status: PINNED.  Every function below is checked in tests/ against
//   (1) the reference's own objects linked into oracle/_ref/ref_probe (bit-level,
//       hex floats) on the reference's fixtures and on seeded synthetic reads,
//   (2) the committed golden vectors under tests/golden/ (made by
//       tests/golden/make_golden.py from ref_probe / the reference binary),
//   (3) the reference's known answers recorded in SURVEY.md §8(c).
//
// It is written in C-style C++ (extern "C", plain arrays) rather than C for one
// reason: the tie order of the reference's final sort is *defined* by libstdc++'s
// std::sort (reference src/main.cpp:247-248), so the oracle must call the same
// library routine.  libm pow/sqrt come from the same glibc as the reference.
// Build flags mirror the reference Makefile:11-20 (-std=c++11 -O3, no -march,
// ISO mode => no FP contraction).
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference).

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "flx_oracle.h"

// ---------------------------------------------------------------------------
// a1  Phred char -> quality      (src/read.cpp:270-273)
// ---------------------------------------------------------------------------
extern "C" double flo_qscore_to_quality(int c_signed_char) {
    // `char` is signed on this ABI: bytes >= 128 give negative q.
    int q = c_signed_char - 33;
    return 1.0 - pow(10.0, -q / 10.0);
}

extern "C" void flo_phred_lut(double *lut256) {
    for (int b = 0; b < 256; ++b) lut256[b] = flo_qscore_to_quality((int)(signed char)(unsigned char)b);
}

// ---------------------------------------------------------------------------
// a6  mean quality               (src/read.cpp:208-213)
//     strictly left-to-right FP64 sum; (100.0*sum)/double(n); n == 0 -> NaN
// ---------------------------------------------------------------------------
extern "C" double flo_mean_quality(const double *q, uint64_t n) {
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) acc += q[i];
    return 100.0 * acc / (double)n;
}

// ---------------------------------------------------------------------------
// a7  sliding-window minimum     (src/read.cpp:216-236)
// ---------------------------------------------------------------------------
extern "C" double flo_window_quality(const double *q, uint64_t n, uint64_t ws) {
    if (n <= ws) return flo_mean_quality(q, n);
    double acc = 0.0;
    for (uint64_t i = 0; i < ws; ++i) acc += q[i];
    double w = acc / (double)ws;
    double lowest = w;
    for (uint64_t j = ws; j < n; ++j) {
        w -= q[j - ws] / (double)ws;  // two divisions per step, strict order
        w += q[j] / (double)ws;
        if (w < lowest) lowest = w;
    }
    if (lowest < 0.5 / (double)ws) lowest = 0.0;  // drift clamp, read.cpp:233-234
    return 100.0 * lowest;
}

// ---------------------------------------------------------------------------
// a8  length score               (src/read.cpp:241-244), half score at 5 kbp
// ---------------------------------------------------------------------------
extern "C" double flo_length_score(int length) {
    const double half = 5000.0;
    return 100.0 * (1.0 + (-half / (length + half)));
}

// ---------------------------------------------------------------------------
// a4  2-bit encoders             (src/kmers.cpp:176-239)
//     forward: A0 C1 G2 T3, anything else 0
//     reverse: complement in the top two bits, anything else 0 (so 'N' acts as
//     'A' forward but as 'T' on the reverse strand — asymmetric, kept)
// ---------------------------------------------------------------------------
extern "C" uint32_t flo_base_fwd(int base) {
    switch (base) {
        case 'C': case 'c': return 1u;
        case 'G': case 'g': return 2u;
        case 'T': case 't': return 3u;
        default: return 0u;
    }
}

extern "C" uint32_t flo_base_rev(int base) {
    switch (base) {
        case 'G': case 'g': return 1u << 30;
        case 'C': case 'c': return 2u << 30;
        case 'A': case 'a': return 3u << 30;
        default: return 0u;
    }
}

extern "C" uint32_t flo_start_kmer_fwd(const char *s) {
    uint32_t k = 0;
    for (int i = 0; i < 16; ++i) k = (k << 2) | flo_base_fwd(s[i]);
    return k;
}

extern "C" uint32_t flo_start_kmer_rev(const char *s) {
    uint32_t k = 0;
    for (int i = 0; i < 16; ++i) k = (k >> 2) | flo_base_rev(s[i]);
    return k;
}

// ---------------------------------------------------------------------------
// a17-a19  Bloom filter (Arash Partow "Open Bloom Filter", vendored in the
//     reference as src/bloom_filter.h).  Only the members the short-read set
//     build reaches are restated:
//       compute_optimal_parameters  bloom_filter.h:108-160
//       ctor + generate_unique_salt bloom_filter.h:183-195, 467-529
//       hash_ap (4-byte key branch) bloom_filter.h:551-608 (569-583)
//       compute_indices/insert/contains  461-465, 260-280, 303-325
// ---------------------------------------------------------------------------
extern "C" void flo_bloom_parameters(uint64_t n_projected, double fp_prob, uint32_t *n_hashes, uint64_t *table_bits) {
    double best_m = std::numeric_limits<double>::infinity();
    double best_k = 0.0;
    for (double k = 1.0; k < 1000.0; k += 1.0) {
        double m = (-k * (double)n_projected) / std::log(1.0 - std::pow(fp_prob, 1.0 / k));
        if (m < best_m) { best_m = m; best_k = k; }
    }
    uint32_t nh = (uint32_t)best_k;
    uint64_t bits = (uint64_t)best_m;
    if (bits % 8) bits += 8 - bits % 8;
    // clamp to the library's [min,max] (bloom_filter.h:44-60 defaults: hashes 1..UINT_MAX, size 1..ULLONG_MAX)
    if (nh < 1) nh = 1;
    if (bits < 1) bits = 1;
    *n_hashes = nh;
    *table_bits = bits;
}

static const uint32_t kPredefSalt13[13] = {  // first 13 of bloom_filter.h:478-511
    0xAAAAAAAAu, 0x55555555u, 0x33333333u, 0xCCCCCCCCu, 0x66666666u, 0x99999999u, 0xB5B5B5B5u,
    0x4B4B4B4Bu, 0xAA55AA55u, 0x55335533u, 0x33CC33CCu, 0xCC66CC66u, 0x66996699u};

extern "C" void flo_bloom_salts(uint64_t user_seed, uint32_t n_hashes, uint32_t *salts) {
    // n_hashes must be <= 13 here (Filtlong's parameters give exactly 13).
    uint64_t random_seed = user_seed * 0xA5A5A5A5ULL + 1ULL;  // bloom_filter.h:186
    for (uint32_t i = 0; i < n_hashes; ++i) salts[i] = kPredefSalt13[i];
    for (uint32_t i = 0; i < n_hashes; ++i)  // in place, so later salts see updated earlier ones
        salts[i] = salts[i] * salts[(i + 3) % n_hashes] + (uint32_t)random_seed;
}

extern "C" uint32_t flo_bloom_hash(uint32_t key, uint32_t salt) {
    // 4-byte key: one pass through the `remaining_length >= 4` branch with loop == 0.
    uint32_t h = salt;
    h ^= ~((h << 11) + (key ^ (h >> 5)));
    return h;
}

struct flo_bloom {
    uint32_t n_hashes;
    uint64_t table_bits;
    uint32_t salts[16];
    std::vector<uint8_t> table;
    bool contains(uint32_t key) const {
        for (uint32_t i = 0; i < n_hashes; ++i) {
            uint64_t idx = flo_bloom_hash(key, salts[i]) % table_bits;
            if (!(table[idx >> 3] & (1u << (idx & 7)))) return false;
        }
        return true;
    }
    void insert(uint32_t key) {
        for (uint32_t i = 0; i < n_hashes; ++i) {
            uint64_t idx = flo_bloom_hash(key, salts[i]) % table_bits;
            table[idx >> 3] |= (uint8_t)(1u << (idx & 7));
        }
    }
};

// ---------------------------------------------------------------------------
// a13-a16  reference 16-mer set   (src/kmers.cpp:28-42, 75-134, 137-166, 170-172)
// ---------------------------------------------------------------------------
struct flo_kmerset {
    std::unordered_set<uint32_t> present;
    std::unordered_map<uint32_t, int> counts;
    flo_bloom *bloom;
    int required_copies;
    uint64_t bloom_false_positives;  // diagnostic: 2nd-sighting decisions made at a first sighting
};

extern "C" flo_kmerset *flo_kmerset_new(void) {
    flo_kmerset *s = new flo_kmerset();
    s->bloom = nullptr;  // the reference allocates 240 MB eagerly (kmers.cpp:28-42); we defer to first use
    s->required_copies = 4;
    s->bloom_false_positives = 0;
    return s;
}

extern "C" void flo_kmerset_free(flo_kmerset *s) {
    if (!s) return;
    delete s->bloom;
    delete s;
}

extern "C" uint64_t flo_kmerset_size(const flo_kmerset *s) { return s->present.size(); }
extern "C" int flo_kmerset_contains(const flo_kmerset *s, uint32_t kmer) { return s->present.count(kmer) ? 1 : 0; }

extern "C" uint64_t flo_kmerset_dump(const flo_kmerset *s, uint32_t *out, uint64_t cap) {
    uint64_t n = 0;
    for (uint32_t k : s->present) {
        if (n < cap) out[n] = k;
        ++n;
    }
    if (n <= cap) std::sort(out, out + n);
    return n;
}

static void ensure_bloom(flo_kmerset *s) {
    if (s->bloom) return;
    flo_bloom *b = new flo_bloom();
    flo_bloom_parameters(100000000ULL, 0.0001, &b->n_hashes, &b->table_bits);  // kmers.cpp:32-36
    flo_bloom_salts(0xA5A5A5A5ULL, b->n_hashes, b->salts);
    b->table.assign(b->table_bits / 8, 0);
    s->bloom = b;
}

static inline void add_one_copy(flo_kmerset *s, uint32_t k) { s->present.insert(k); }  // kmers.cpp:137-139

static inline void add_multi_copy(flo_kmerset *s, uint32_t k) {  // kmers.cpp:142-166
    if (s->present.count(k)) return;
    if (!s->bloom->contains(k)) {
        s->bloom->insert(k);
        return;
    }
    auto it = s->counts.find(k);
    if (it == s->counts.end()) {
        s->counts[k] = 2;
        return;
    }
    if (++(it->second) >= s->required_copies) {
        s->present.insert(k);
        s->counts.erase(it);
    }
}

// One reference sequence (the body of the while loop at kmers.cpp:89-128).
// multi_copy = 0: assembly rule; 1: short-read (>= 4 copies) rule.
extern "C" void flo_kmerset_add_sequence(flo_kmerset *s, const char *seq, uint64_t len, int multi_copy) {
    if (len < 16) return;  // kmers.cpp:99-100
    if (multi_copy) ensure_bloom(s);
    uint32_t f = flo_start_kmer_fwd(seq);
    uint32_t r = flo_start_kmer_rev(seq);
    for (uint64_t i = 15;; ) {
        if (multi_copy) { add_multi_copy(s, f); add_multi_copy(s, r); }
        else { add_one_copy(s, f); add_one_copy(s, r); }
        if (++i >= len) break;
        f = (f << 2) | flo_base_fwd(seq[i]);
        r = (r >> 2) | flo_base_rev(seq[i]);
    }
}

// ---------------------------------------------------------------------------
// a2/a3/a9-a12  per-read scoring  (src/read.cpp:25-144)
// ---------------------------------------------------------------------------
static void score_slice(const flo_kmerset *set, const char *seq, const char *qual, int length, const flo_params *p,
                        std::vector<double> &q, flo_read_result *out) {
    const bool kmer_mode = set && !set->present.empty();  // Kmers::empty(), kmers.h:34
    q.assign((size_t)length, 0.0);
    if (!kmer_mode) {  // read.cpp:35-39
        for (int i = 0; i < length; ++i) q[i] = flo_qscore_to_quality((int)(signed char)qual[i]);
    } else if (length >= 16) {  // read.cpp:43-58
        uint32_t k = flo_start_kmer_fwd(seq);
        for (int i = 15; i < length; ++i) {
            if (i > 15) k = (k << 2) | flo_base_fwd(seq[i]);
            if (set->present.count(k))
                for (int j = i - 15; j <= i; ++j) q[j] = 1.0;
        }
    }
    out->length = length;
    out->mean_q = flo_mean_quality(q.data(), (uint64_t)length);
    out->window_q = flo_window_quality(q.data(), (uint64_t)length, (uint64_t)p->window_size);
    out->length_score = flo_length_score(length);

    int ok = 1;  // read.cpp:64-73 (else-if chain; order matters only for NaN, which compares false anyway)
    if (p->min_length_set && length < p->min_length) ok = 0;
    else if (p->max_length_set && length > p->max_length) ok = 0;
    else if (p->min_mean_q_set && out->mean_q < p->min_mean_q) ok = 0;
    else if (p->min_window_q_set && out->window_q < p->min_window_q) ok = 0;
    out->passed = ok;

    out->first = -1;  // read.cpp:75-84
    out->last = -1;
    if (kmer_mode) {
        for (int i = 0; i < length; ++i)
            if (q[i] != 0) {
                if (out->first == -1) out->first = i;
                out->last = i + 1;
            }
    }
}

extern "C" int flo_score_read(const flo_kmerset *set, const char *seq, const char *qual, int length,
                              const flo_params *p, flo_read_result *out, int32_t *bad_ranges, int32_t *child_ranges,
                              flo_read_result *children, int cap) {
    std::vector<double> q;
    score_slice(set, seq, qual, length, p, q, out);
    out->n_bad = 0;
    out->n_child = 0;
    const bool kmer_mode = set && !set->present.empty();
    if (!kmer_mode || !(p->trim || p->split_set)) return 0;

    std::vector<std::pair<int, int>> bad;
    if (p->split_set) {  // read.cpp:89-103: maximal zero runs of length >= split
        int i = 0;
        while (i < length) {
            if (q[i] == 0.0) {
                int s = i;
                while (i < length && q[i] == 0.0) ++i;
                if (i - s >= p->split) bad.push_back(std::make_pair(s, i));
            } else
                ++i;
        }
    }
    if (p->trim) {  // read.cpp:106-117
        if (out->first > 0) {
            std::pair<int, int> head(0, out->first);
            if (bad.empty() || bad.front() != head) bad.insert(bad.begin(), head);
        }
        if (out->last != -1 && out->last < length) {
            std::pair<int, int> tail(out->last, length);
            if (bad.empty() || bad.back() != tail) bad.push_back(tail);
        }
    }
    std::vector<std::pair<int, int>> kids;
    if (!bad.empty()) {  // read.cpp:119-130: complement of the bad ranges, empty pieces dropped
        int s = 0;
        for (auto &b : bad) {
            if (b.first - s > 0) kids.push_back(std::make_pair(s, b.first));
            s = b.second;
        }
        if (length - s > 0) kids.push_back(std::make_pair(s, length));
    }
    out->n_bad = (int)bad.size();
    out->n_child = (int)kids.size();
    if ((int)bad.size() > cap || (int)kids.size() > cap) return -1;
    for (size_t i = 0; i < bad.size(); ++i) { bad_ranges[2 * i] = bad[i].first; bad_ranges[2 * i + 1] = bad[i].second; }
    // read.cpp:131-141: each child is a full Read on (seq+start, qual+start, len) with the same set/params.
    // Children never produce grandchildren in practice but the reference *does* run the whole
    // constructor on them; the child fields the rest of the program reads are the ones filled here.
    flo_params child_p = *p;
    for (size_t i = 0; i < kids.size(); ++i) {
        child_ranges[2 * i] = kids[i].first;
        child_ranges[2 * i + 1] = kids[i].second;
        std::vector<double> cq;
        score_slice(set, seq + kids[i].first, qual ? qual + kids[i].first : nullptr, kids[i].second - kids[i].first,
                    &child_p, cq, &children[i]);
        children[i].n_bad = 0;
        children[i].n_child = 0;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// the same constructor over a packed batch, on several host threads (whole-population parity at BASELINE size:
// tests/test_gpu_fullsize.py re-scores 10^5..10^6 reads of the device's own plane).  Every read goes through
// flo_score_read above; the threads only share the read-only set.  Children come back as a CSR in read order.
// Returns the number of children, or -1 if a read has more than `cap_per_read` children / the CSR more than child_cap.
// ---------------------------------------------------------------------------
extern "C" int64_t flo_score_plane_mt(const flo_kmerset *set, int kmer_plane, const uint8_t *plane, const uint64_t *offsets,
                                      const int32_t *lengths, uint64_t n, const flo_params *p, int n_threads, double *mean_q,
                                      double *window_q, uint8_t *passed, int32_t *first, int32_t *last,
                                      uint64_t *child_offsets, uint64_t child_cap, int32_t *child_ranges, double *child_mean_q,
                                      double *child_window_q, uint8_t *child_passed) {
    struct Kid { uint64_t read; int32_t s, e; double mean, window; uint8_t passed; };
    if (n_threads < 1) n_threads = 1;
    std::vector<std::vector<Kid>> kids((size_t)n_threads);
    std::vector<uint32_t> n_kids((size_t)n + 1, 0u);
    std::atomic<uint64_t> next(0);
    std::atomic<int> failed(0);
    const int cap = 4096;
    auto work = [&](int t) {
        std::vector<int32_t> bad(2 * cap), rng(2 * cap);
        std::vector<flo_read_result> ch((size_t)cap);
        for (;;) {
            const uint64_t b = next.fetch_add(64);
            if (b >= n) break;
            for (uint64_t i = b; i < std::min<uint64_t>(b + 64, n); ++i) {
                flo_read_result r;
                const char *bytes = (const char *)plane + offsets[i];
                if (flo_score_read(set, kmer_plane ? bytes : nullptr, kmer_plane ? nullptr : bytes, lengths[i], p, &r, bad.data(),
                                   rng.data(), ch.data(), cap) != 0) {
                    failed = 1;
                    continue;
                }
                mean_q[i] = r.mean_q;
                window_q[i] = r.window_q;
                passed[i] = (uint8_t)r.passed;
                if (first) first[i] = r.first;
                if (last) last[i] = r.last;
                n_kids[i] = (uint32_t)r.n_child;
                for (int k = 0; k < r.n_child; ++k)
                    kids[(size_t)t].push_back(Kid{i, rng[2 * k], rng[2 * k + 1], ch[k].mean_q, ch[k].window_q, (uint8_t)ch[k].passed});
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (failed) return -1;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (child_offsets) child_offsets[i] = total;
        total += n_kids[i];
    }
    if (child_offsets) child_offsets[n] = total;
    if (total == 0) return 0;
    if (!child_offsets || total > child_cap) return -1;
    std::vector<uint32_t> fill((size_t)n, 0u);  // a read's children were pushed in order by ONE thread
    for (auto &v : kids)
        for (auto &k : v) {
            const uint64_t at = child_offsets[k.read] + fill[k.read]++;
            child_ranges[2 * at] = k.s;
            child_ranges[2 * at + 1] = k.e;
            child_mean_q[at] = k.mean;
            child_window_q[at] = k.window;
            child_passed[at] = k.passed;
        }
    return (int64_t)total;
}

// ---------------------------------------------------------------------------
// a22  final score               (src/read.cpp:249-267)
// ---------------------------------------------------------------------------
extern "C" double flo_final_score(double length_score, double mean_q, double window_q, double lw, double mw,
                                  double ww) {
    double product = pow(length_score, lw) * pow(mean_q, mw);
    double total = lw + mw;
    double score = pow(product, 1.0 / total);
    double scale;
    if (mean_q > 0.0)
        scale = std::min(window_q / mean_q, 1.0);
    else
        scale = 1.0;
    total = lw + mw + ww;
    double wfrac = ww / total;
    double nfrac = 1.0 - wfrac;
    scale = nfrac + (scale * wfrac);
    return score * scale;
}

// ---------------------------------------------------------------------------
// a19b  reads2 gather             (src/main.cpp:138-147)
//   for (auto read : reads) { if (read->m_child_reads.size() == 0) reads2.push_back(read);
//                             else for (auto child : read->m_child_reads) reads2.push_back(child); }
//   a child's length is end - start of its range (src/read.cpp:131-137)
// ---------------------------------------------------------------------------
extern "C" uint64_t flo_reads2_gather(uint64_t n, const int32_t *length, const double *mean_q, const double *window_q,
                                      const uint8_t *passed, const uint64_t *child_offsets, const int32_t *child_ranges,
                                      const double *child_mean_q, const double *child_window_q,
                                      const uint8_t *child_passed, double *mean_q2, double *window_q2, int32_t *length2,
                                      uint8_t *passed2, uint32_t *parent2, int64_t *child2) {
    uint64_t at = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t a = child_offsets ? child_offsets[i] : 0, b = child_offsets ? child_offsets[i + 1] : 0;
        if (a == b) {
            mean_q2[at] = mean_q[i]; window_q2[at] = window_q[i]; length2[at] = length[i]; passed2[at] = passed[i];
            if (parent2) parent2[at] = (uint32_t)i;
            if (child2) child2[at] = -1;
            ++at;
        } else {
            for (uint64_t k = a; k < b; ++k) {
                mean_q2[at] = child_mean_q[k]; window_q2[at] = child_window_q[k];
                length2[at] = child_ranges[2 * k + 1] - child_ranges[2 * k]; passed2[at] = child_passed[k];
                if (parent2) parent2[at] = (uint32_t)i;
                if (child2) child2[at] = (int64_t)k;
                ++at;
            }
        }
    }
    return at;
}

// ---------------------------------------------------------------------------
// a20-a25  global stage           (src/main.cpp:169-261)
//   arrays are in reads2 order (file order, children in place of their parents,
//   main.cpp:138-147).  mean_q / window_q are overwritten with the normalised
//   values exactly as the reference overwrites the Read fields (main.cpp:207-208).
// ---------------------------------------------------------------------------
extern "C" int flo_rank_and_cut(uint64_t n, double *mean_q, double *window_q, const int32_t *length, uint8_t *passed,
                                double lw, double mw, double ww, int target_bases_set, int64_t target_bases_arg,
                                int keep_percent_set, double keep_percent, int64_t total_bases, double *final_score,
                                flo_cut_report *rep) {
    // main.cpp:170-196 — serial folds in reads2 order
    double qmin = 100.0, qmax = 0.0, qsum = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        qsum += mean_q[i];
        if (mean_q[i] > qmax) qmax = mean_q[i];
        if (mean_q[i] < qmin) qmin = mean_q[i];
    }
    double qmean = qsum / (double)n;
    double ssum = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        double d = mean_q[i] - qmean;
        ssum += d * d;
    }
    double qstd = sqrt(ssum / (double)n);
    double zmin, zmax;
    if (qstd > 0.0) {
        zmin = (qmin - qmean) / qstd;
        zmax = (qmax - qmean) / qstd;
    } else {
        zmin = 1.0;
        zmax = 1.0;
    }
    double zspan = zmax - zmin;
    rep->mean_quality = qmean;
    rep->stdev_quality = qstd;
    rep->min_z = zmin;
    rep->max_z = zmax;

    // main.cpp:202-212
    for (uint64_t i = 0; i < n; ++i) {
        double ratio = window_q[i] / mean_q[i];
        if (ratio > 1.0) ratio = 1.0;
        double z = (mean_q[i] - qmean) / qstd;
        mean_q[i] = 100.0 * (z - zmin) / zspan;
        window_q[i] = mean_q[i] * ratio;
        final_score[i] = flo_final_score(flo_length_score(length[i]), mean_q[i], window_q[i], lw, mw, ww);
    }

    rep->outcome = FLO_CUT_NONE;
    rep->target_bases = 0;
    rep->kept_bases = 0;
    if (!(target_bases_set || keep_percent_set)) return 0;

    // main.cpp:218-244
    long long passed_bases = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (passed[i]) passed_bases += length[i];
    long long target = target_bases_set ? (long long)target_bases_arg : std::numeric_limits<long long>::max();
    if (keep_percent_set) {
        long long keep = (long long)((keep_percent / 100.0) * total_bases);
        target = std::min(target, keep);
    }
    rep->target_bases = target;
    if (target >= total_bases) { rep->outcome = FLO_CUT_NOT_ENOUGH; return 0; }
    if (target >= passed_bases) { rep->outcome = FLO_CUT_ALREADY_BELOW; return 0; }

    // main.cpp:247-257 — std::sort (unstable, libstdc++ introsort) on the reads2 order, then the walk
    std::vector<uint64_t> order(n);
    for (uint64_t i = 0; i < n; ++i) order[i] = i;
    const double *fs = final_score;
    std::sort(order.begin(), order.end(), [fs](uint64_t a, uint64_t b) { return fs[a] > fs[b]; });
    long long so_far = 0;
    for (uint64_t k = 0; k < n; ++k) {
        uint64_t i = order[k];
        if (passed[i] && so_far < target)
            so_far += length[i];
        else
            passed[i] = 0;
    }
    rep->kept_bases = so_far;
    rep->outcome = FLO_CUT_SORTED;
    return 0;
}

// ---------------------------------------------------------------------------
// synthetic generator (shared definition, oracle/synth.h) exported for tests
// ---------------------------------------------------------------------------
#include "synth.h"

extern "C" uint64_t flo_synth_mix(uint64_t seed, uint64_t stream, uint64_t read, uint64_t pos) {
    return flx_mix(seed, stream, read, pos);
}

extern "C" void flo_synth_qual(uint64_t seed, uint64_t read, uint64_t length, uint8_t *out) {
    int mu = flx_synth_mu(seed, read);
    for (uint64_t i = 0; i < length; ++i) out[i] = flx_synth_qual(seed, read, i, mu);
}

extern "C" void flo_synth_bases(uint64_t seed, uint64_t stream, uint64_t read, uint64_t start, uint64_t length,
                                uint8_t *out) {
    for (uint64_t i = 0; i < length; ++i) out[i] = flx_synth_base(seed, stream, read, start + i);
}

extern "C" void flo_synth_seq(uint64_t seed, int profile, uint64_t read, int length, const uint8_t *ref, uint64_t ref_len, uint8_t *out) {
    flx_synth_seq_read(seed, profile, read, length, ref, ref_len, out);
}

// ---------------------------------------------------------------------------
// bench support ("port" CPU baseline): score + rank n synthetic Phred-only reads
// with the restatement above, in memory, timed.  Used by bench.py only when the
// real reference harness (oracle/_ref/ref_bench) is not available.
// ---------------------------------------------------------------------------
#include <chrono>

extern "C" int flo_bench_phred(uint64_t n, uint64_t seed, uint64_t first_read, int64_t target_bases,
                               double *score_s, double *rank_s, int64_t *total_bases_out, int64_t *kept_out) {
    using clk = std::chrono::steady_clock;
    std::vector<int> len(n);
    int64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        double g = 0.0;
        for (int j = 0; j < 4; ++j) {
            uint64_t h = flx_mix(seed, FLX_STREAM_LEN, first_read + i, j);
            g += -log(((double)(h >> 11) + 0.5) / 9007199254740992.0);
        }
        long long L = llround(2500.0 * g);
        len[i] = (int)std::min<long long>(std::max<long long>(L, 200), 200000);
        total += len[i];
    }
    std::vector<char> plane((size_t)total + 1);
    {
        size_t off = 0;
        for (uint64_t i = 0; i < n; ++i) {
            flo_synth_qual(seed, first_read + i, (uint64_t)len[i], (uint8_t *)&plane[off]);
            off += len[i];
        }
    }
    flo_params p;
    memset(&p, 0, sizeof p);
    p.window_size = 250;
    std::vector<double> mq(n), wq(n), fs(n);
    std::vector<uint8_t> passed(n);
    auto t0 = clk::now();
    {
        size_t off = 0;
        std::vector<double> q;
        flo_read_result r;
        for (uint64_t i = 0; i < n; ++i) {
            score_slice(nullptr, &plane[off], &plane[off], len[i], &p, q, &r);
            mq[i] = r.mean_q; wq[i] = r.window_q; passed[i] = (uint8_t)r.passed;
            off += len[i];
        }
    }
    auto t1 = clk::now();
    flo_cut_report rep;
    flo_rank_and_cut(n, mq.data(), wq.data(), len.data(), passed.data(), 1.0, 1.0, 1.0, 1, target_bases, 0, 0.0, total,
                     fs.data(), &rep);
    auto t2 = clk::now();
    *score_s = std::chrono::duration<double>(t1 - t0).count();
    *rank_s = std::chrono::duration<double>(t2 - t1).count();
    *total_bases_out = total;
    *kept_out = rep.kept_bases;
    return 0;
}
