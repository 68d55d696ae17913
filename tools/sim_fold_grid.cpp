// sim_fold_grid.cpp — the integer-grid window fold, on the host, before it was built for the GPU.
//
// Reference semantics: src/read.cpp:216-236 with qualities 0.0 / 1.0:  w0 = cnt / ws;  per step  w -= q[j-ws]/ws; w += q[j]/ws;
// mn = min(mn, w).  Claim: while w stays inside a GROUP of binades in which fl(1/ws) rounds to the same real number d* on every
// binade's grid (no ties), and does not reach a binade above the highest one it has been in since the regime began, every step is
// exact:  w = w_b + (c - c_b) * d*  — so a word of 32 positions is (total, min prefix, max prefix) of the +-1 walk of its bits.
// Everything else (a new record-high binade, leaving the group, ties, w = 0) replays the word in floating point.
//   (1) exactness: random bit streams x many window sizes, the regime fold against the plain FP fold, bit for bit;
//   (2) how often a wave of 64 lanes has to take the slow path on C3-like coverage (substitution rates 0..12 %, junk blocks).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct Binade { double dstar; bool tie; int group; };  // per unbiased exponent E of w: the step on that binade's grid
struct Table {
    int e_min, e_max;  // exponents covered: e_min .. e_max
    std::vector<Binade> b;
    std::vector<int> glo, ghi;  // per group: lowest / highest exponent
    double delta;
};

static Table make_table(int ws) {
    Table t;
    volatile double one = 1.0, wsd = (double)ws;
    t.delta = one / wsd;
    uint64_t bits;
    memcpy(&bits, &t.delta, 8);
    const int e_d = (int)((bits >> 52) & 0x7ff) - 1023;
    const uint64_t M = (bits & ((1ull << 52) - 1)) | (1ull << 52);  // delta = M * 2^(e_d - 52)
    t.e_min = e_d - 2;
    t.e_max = 1;
    for (int E = t.e_min; E <= t.e_max; ++E) {
        Binade x;
        const int shift = E - e_d;  // grid of binade E = 2^(E-52) = 2^shift units of delta's ulp
        x.tie = false;
        if (shift <= 0) x.dstar = t.delta;
        else if (shift >= 53) { x.dstar = 0; x.tie = true; }  // (not reached: w <= 2)
        else {
            const uint64_t rem = M & ((1ull << shift) - 1), half = 1ull << (shift - 1);
            x.tie = rem == half;
            const uint64_t q = (M >> shift) + (rem > half ? 1 : 0);
            x.dstar = ldexp((double)q, shift + e_d - 52);  // q < 2^54: exact
        }
        x.group = -1;
        t.b.push_back(x);
    }
    for (size_t i = 0; i < t.b.size(); ++i) {
        if (t.b[i].tie) continue;
        if (i > 0 && !t.b[i - 1].tie && t.b[i - 1].dstar == t.b[i].dstar) { t.b[i].group = t.b[i - 1].group; t.ghi[t.b[i].group] = t.e_min + (int)i; }
        else { t.b[i].group = (int)t.glo.size(); t.glo.push_back(t.e_min + (int)i); t.ghi.push_back(t.e_min + (int)i); }
    }
    return t;
}

// what the kernel does when a regime begins (score_kmer.hip: begin_regime) — same arithmetic, same corrections
struct Regime { bool valid; double wb, dstar; long lo_c, hi_c; };
static long g_reason[4];  // slow words by reason: 0 no regime, 1 below, 2 above
static int g_nibble = 0;    // 1: a slow word is replayed from its first nibble that leaves the regime (costed in round 6, NOT built: below); 0: whole, what the kernel does
static long g_fp_steps = 0; // positions replayed in floating point
static int g_ctz_rule = 1;  // the top of a regime: the highest binade on whose grid w_b already lies (0: the binade w_b is in)
static Regime begin_regime(const Table &t, double w, int ws) {
    Regime r{false, w, 0, 1, -1};
    if (!(w > 0)) return r;
    int E;
    frexp(w, &E);
    E -= 1;  // w in [2^E, 2^(E+1))
    if (E < t.e_min || E > t.e_max) return r;
    const Binade &b = t.b[(size_t)(E - t.e_min)];
    if (b.group < 0) return r;
    // the bottom of the group, and never closer to 0 than 4 delta: a (1,1) step dips by delta and must stay within ONE binade of w
    // the top: w must not reach a binade on whose grid w_b does NOT lie (there its low bits would be rounded away).  w_b lies on the
    // grid of binade E + z, z = trailing zero bits of its mantissa (a window that has been full, w = 1.0, stays on the grid of [1, 2)
    // whatever is subtracted: d* is a multiple of that grid) — up to the top of the group
    int G = E;
    if (g_ctz_rule) {
        uint64_t bits;
        memcpy(&bits, &w, 8);
        const uint64_t m = (bits & ((1ull << 52) - 1)) | (1ull << 52);
        G = std::min(E + (int)__builtin_ctzll(m), t.ghi[(size_t)b.group]);
    }
    const double Lv = std::max(ldexp(1.0, t.glo[(size_t)b.group]), 4.0 * t.delta), Uv = ldexp(1.0, G + 1);
    const double ds = b.dstar, wsd = (double)ws;
    // smallest k with w + k d* > Lv: the estimate (Lv - w) * ws is within 1e-9 of (Lv - w) / d*, so its floor is the answer or
    // one or two below it; the values themselves decide (w + k d* is exact: a multiple of w's grid)
    // (strictly above Lv: a step that lands exactly ON the bottom of the group's lowest binade has its true value, w - delta, a
    // hair below it when delta > d* — in the binade underneath, which rounds on its own, finer grid)
    long k0 = (long)floor((Lv - w) * wsd);
    if (fma((double)k0, ds, w) <= Lv) ++k0;
    if (fma((double)k0, ds, w) <= Lv) ++k0;
    // largest k with w + k d* < Uv: the ceiling of the estimate is the answer or one or two above it
    long k1 = (long)ceil((Uv - w) * wsd);
    if (fma((double)k1, ds, w) >= Uv) --k1;
    if (fma((double)k1, ds, w) >= Uv) --k1;
    r.valid = true;
    r.dstar = ds;
    r.lo_c = k0;
    r.hi_c = k1;
    return r;
}

struct Fold { double w, mn; };
// plain FP fold of positions [ws, L) given bits
static Fold fold_fp(const std::vector<uint8_t> &q, int ws) {
    const int L = (int)q.size();
    volatile double one = 1.0, wsd = (double)ws;
    const double d = one / wsd;
    long cnt = 0;
    for (int i = 0; i < ws; ++i) cnt += q[i];
    double w = (double)cnt / wsd, mn = w;
    for (int j = ws; j < L; ++j) {
        w -= q[j - ws] ? d : 0.0;
        w += q[j] ? d : 0.0;
        if (w < mn) mn = w;
    }
    return {w, mn};
}

// regime fold; `slow` gets the word indices that took the FP path
static Fold fold_grid(const std::vector<uint8_t> &q, int ws, const Table &t, std::vector<int> *slow) {
    const int L = (int)q.size();
    volatile double one = 1.0, wsd = (double)ws;
    const double d = one / wsd;
    long cnt = 0;
    for (int i = 0; i < ws; ++i) cnt += q[i];
    double w = (double)cnt / wsd, mn = w;
    int j = ws;
    // words aligned to 32 from the first multiple of 32 >= ws, like the kernel (the head runs bit by bit)
    for (; j < L && (j & 31); ++j) {
        w -= q[j - ws] ? d : 0.0;
        w += q[j] ? d : 0.0;
        if (w < mn) mn = w;
    }
    Regime r = begin_regime(t, w, ws);
    long c = 0, cmin = 0;
    auto flush = [&]() {
        if (r.valid) {
            const double wn = fma((double)c, r.dstar, r.wb);
            const double m2 = fma((double)cmin, r.dstar, r.wb);
            if (m2 < mn) mn = m2;
            w = wn;
        }
    };
    for (; j + 32 <= L; j += 32) {
        int tt = 0, mp = 0, xp = 0;
        bool any = false;
        for (int i = 0; i < 32; ++i) {
            const int x = (int)q[j + i] - (int)q[j + i - ws];
            any = any || q[j + i] || q[j + i - ws];
            tt += x;
            mp = std::min(mp, tt);
            xp = std::max(xp, tt);
        }
        if (!any) continue;  // (0,0) steps change nothing in any regime
        if (r.valid && c + mp >= r.lo_c && c + xp <= r.hi_c) {
            cmin = std::min(cmin, c + mp);
            c += tt;
            continue;
        }
        // Round 6, costed and NOT built (g_nibble = 1 runs it here): replay a slow word only from the first NIBBLE (4 positions: the
        // granularity of the kernel's walk table) whose prefix leaves the regime — the steps in front of it are still exact on the grid, the
        // same criterion on a shorter walk.  Exact (the cases below run both forms), but 26 of the 32 positions are still replayed on the
        // C3-like reads (half the slow words have no regime at all — w in the tie binade or at 0 — and start at position 0), and finding the
        // nibble costs the kernel a second walk through the table: a loss.
        int first = 0;  // first position replayed in floating point
        if (g_nibble && r.valid) {
            int pre = 0, lo = 0, hi = 0;  // the walk through the nibbles that stay inside
            for (int k = 0; k < 8; ++k) {
                int p2 = pre, lo2 = lo, hi2 = hi;
                for (int i = 4 * k; i < 4 * k + 4; ++i) {
                    p2 += (int)q[j + i] - (int)q[j + i - ws];
                    lo2 = std::min(lo2, p2);
                    hi2 = std::max(hi2, p2);
                }
                if (!(c + lo2 >= r.lo_c && c + hi2 <= r.hi_c)) break;
                pre = p2; lo = lo2; hi = hi2;
                first = 4 * k + 4;
            }
            cmin = std::min(cmin, c + lo);
            c += pre;
        }
        g_fp_steps += 32 - first;
        flush();
        if (slow) slow->push_back(j >> 5);
        ++g_reason[!r.valid ? 0 : (c + mp < r.lo_c ? 1 : 2)];
        for (int i = first; i < 32; ++i) {
            w -= q[j + i - ws] ? d : 0.0;
            w += q[j + i] ? d : 0.0;
            if (w < mn) mn = w;
        }
        r = begin_regime(t, w, ws);
        c = 0;
        cmin = 0;
    }
    flush();
    r.valid = false;
    for (; j < L; ++j) {
        w -= q[j - ws] ? d : 0.0;
        w += q[j] ? d : 0.0;
        if (w < mn) mn = w;
    }
    return {w, mn};
}

static std::vector<uint8_t> c3_like(std::mt19937_64 &g, int L) {
    std::vector<uint8_t> sub((size_t)L, 0), cov((size_t)L, 0);
    const int erate = (int)(g() % 13);
    for (int i = 0; i < L; ++i) sub[i] = (int)(g() % 100) < erate && (g() & 3) != 0;  // (a substitution by the same base changes nothing)
    int js = -1, je = -1;
    if (L > 3000 && g() % 10 < 3) { js = 500 + (int)(g() % (uint64_t)(L - 2000)); je = std::min(js + 800, L); }
    for (int i = 0; i < L; ++i) if (i >= js && i < je) sub[i] = (g() & 3) != 0;
    int run = 0;
    for (int i = 0; i < L; ++i) {
        run = sub[i] ? 0 : run + 1;
        if (run >= 16) for (int k = i - 15; k <= i; ++k) cov[k] = 1;
    }
    return cov;
}

int main(int argc, char **argv) {
    std::mt19937_64 g(12345);
    // (1) exactness
    long cases = 0, bad = 0;
    const int n_ws = argc > 1 ? atoi(argv[1]) : 400;
    for (int k = 0; k < n_ws; ++k) {
        const int ws = k < 40 ? (int[]){250, 100, 1, 2, 3, 5, 7, 16, 32, 33, 64, 127, 128, 129, 255, 256, 257, 500, 512, 1000, 1024, 2047, 2048, 4096, 9999, 10000, 31, 63, 65, 96, 97, 200, 300, 333, 400, 600, 750, 800, 900, 1500}[k] : 1 + (int)(g() % 3000);
        const Table t = make_table(ws);
        for (int rep = 0; rep < 30; ++rep) {
            const int L = ws + 64 + (int)(g() % 20000);
            std::vector<uint8_t> q((size_t)L);
            const int style = rep % 6;
            if (style == 0) q = c3_like(g, L);
            else if (style == 1) { const int p = (int)(g() % 100); for (auto &x : q) x = (int)(g() % 100) < p; }
            else if (style == 2) { int run = 0, v = 1; for (auto &x : q) { if (run-- <= 0) { v ^= 1; run = (int)(g() % (uint64_t)(2 * ws + 2)); } x = (uint8_t)v; } }
            else if (style == 3) { for (int i = 0; i < L; ++i) q[i] = (i / (1 + rep)) & 1; }
            else if (style == 4) { const int p = 45 + (int)(g() % 10); int run = 0, v = 0; for (auto &x : q) { if (run-- <= 0) { v = (int)(g() % 100) < p; run = (int)(g() % 40); } x = (uint8_t)v; } }
            else { for (auto &x : q) x = 1; for (int z = 0; z < 5; ++z) { const int a = (int)(g() % (uint64_t)L); for (int i = a; i < std::min(L, a + (int)(g() % 600)); ++i) q[i] = 0; } }
            const Fold a = fold_fp(q, ws);
            ++cases;
            for (int nib = 0; nib < 2; ++nib) {  // whole-word replay (the kernel) and replay from the first nibble outside (costed, not built)
                g_nibble = nib;
                const Fold b = fold_grid(q, ws, t, nullptr);
                if (memcmp(&a.w, &b.w, 8) || memcmp(&a.mn, &b.mn, 8)) {
                    if (++bad < 10) printf("MISMATCH ws %d style %d L %d nibble %d: w %a vs %a  mn %a vs %a\n", ws, style, L, nib, a.w, b.w, a.mn, b.mn);
                }
            }
            g_nibble = 0;
        }
    }
    printf("exactness: %ld cases, %ld mismatches\n", cases, bad);
    // the groups of the default window
    for (int ws : {250, 100, 1000, 500}) {
        const Table t = make_table(ws);
        printf("ws %d: groups", ws);
        for (size_t i = 0; i < t.glo.size(); ++i) printf(" [2^%d, 2^%d)", t.glo[i], t.ghi[i] + 1);
        printf("   ties at:");
        for (size_t i = 0; i < t.b.size(); ++i) if (t.b[i].tie) printf(" 2^%d", t.e_min + (int)i);
        printf("\n");
    }
    g_reason[0] = g_reason[1] = g_reason[2] = 0;
    g_fp_steps = 0;
    g_nibble = argc > 2 ? atoi(argv[2]) : 0;  // (second argument 1: the statistics of the nibble form)
    // (2) slow words per wave, C3-like reads (gamma lengths, sorted descending, 64 per wave)
    for (int rule = 0; rule < 2; ++rule)
    for (int ws : {250, 100, 500}) {
        g_ctz_rule = rule;
        printf("top of a regime %s: ", rule ? "by the grid w_b lies on" : "the binade w_b is in");
        const Table t = make_table(ws);
        const int n = 64 * 60;
        std::vector<int> len((size_t)n);
        std::gamma_distribution<double> gd(4.0, 2500.0);
        for (auto &x : len) x = std::max(ws + 100, std::min(200000, (int)gd(g)));
        std::sort(len.begin(), len.end(), std::greater<int>());
        long wave_words = 0, wave_slow = 0, lane_words = 0, lane_slow = 0;
        for (int w0 = 0; w0 < n; w0 += 64) {
            std::vector<std::vector<int>> slow(64);
            int maxw = 0;
            for (int l = 0; l < 64; ++l) {
                const std::vector<uint8_t> q = c3_like(g, len[(size_t)(w0 + l)]);
                fold_grid(q, ws, t, &slow[(size_t)l]);
                lane_slow += (long)slow[(size_t)l].size();
                lane_words += (len[(size_t)(w0 + l)] - ws) / 32;
                maxw = std::max(maxw, len[(size_t)(w0 + l)] / 32 + 1);
            }
            std::vector<char> any((size_t)maxw + 1, 0);
            for (auto &v : slow) for (int x : v) any[(size_t)x] = 1;
            for (char x : any) wave_slow += x;
            wave_words += (len[(size_t)w0] - ws) / 32;
        }
        printf("ws %d: lane-words %ld, slow %ld (%.3f %%); wave-words %ld, with a slow lane %ld (%.1f %%); reasons: no regime %ld, below %ld, above %ld; positions replayed per slow lane-word %.1f\n", ws, lane_words, lane_slow,
               100.0 * lane_slow / lane_words, wave_words, wave_slow, 100.0 * wave_slow / wave_words, g_reason[0], g_reason[1], g_reason[2], lane_slow ? (double)g_fp_steps / lane_slow : 0.0);
        g_fp_steps = 0;
        g_reason[0] = g_reason[1] = g_reason[2] = 0;
    }
    return bad != 0;
}
