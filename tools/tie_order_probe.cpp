// tie_order_probe — does libstdc++ std::sort leave a group of EQUAL keys in an order that can be predicted without running it
// on the full input?  (the exact-tie fallback of csrc/rank.hip runs the reference's own std::sort over all
// reads2 entries, src/main.cpp:247-248.)  Three candidate shortcuts are compared with the real thing on inputs with ties:
//   (a) std::sort of the array already in (score desc, index asc) order — what a stable device sort produces;
//   (b) stable order (index ascending inside a tie group);   (c) reverse stable order.
// Prints how often the tie group straddling a cut keeps a different member set.   g++ -O2 -std=c++11 tools/tie_order_probe.cpp
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
int main() {
    std::mt19937_64 rng(7);
    int trials = 0, diff_a = 0, diff_b = 0, diff_c = 0;
    for (int t = 0; t < 400; ++t) {
        const int n = 1000 + (int)(rng() % 50000);
        const int levels = 3 + (int)(rng() % 200);  // few distinct scores: big tie groups
        std::vector<double> sc(n);
        std::vector<int> len(n);
        for (int i = 0; i < n; ++i) { sc[i] = (double)(rng() % levels); len[i] = 100 + (int)(rng() % 900); }
        long long total = std::accumulate(len.begin(), len.end(), 0ll), target = total / 2;
        auto cmp = [&](uint32_t a, uint32_t b) { return sc[a] > sc[b]; };
        auto walk = [&](const std::vector<uint32_t> &ord) {
            std::vector<uint8_t> keep(n, 0);
            long long so = 0;
            for (uint32_t i : ord) if (so < target) { so += len[i]; keep[i] = 1; }
            return keep;
        };
        std::vector<uint32_t> ref(n), a(n), b(n), c(n);
        std::iota(ref.begin(), ref.end(), 0u);
        b = ref;
        std::sort(ref.begin(), ref.end(), cmp);              // the reference: introsort on reads2 order
        std::stable_sort(b.begin(), b.end(), cmp);           // (b)
        a = b; std::sort(a.begin(), a.end(), cmp);           // (a) std::sort of the pre-sorted order
        c = b;
        for (size_t i = 0; i < c.size();) { size_t e = i; while (e < c.size() && sc[c[e]] == sc[c[i]]) ++e; std::reverse(c.begin() + i, c.begin() + e); i = e; }
        const auto k = walk(ref);
        ++trials;
        diff_a += walk(a) != k; diff_b += walk(b) != k; diff_c += walk(c) != k;
    }
    printf("%d inputs with tie groups straddling the cut: kept set differs from std::sort(reads2 order) for\n"
           "  (a) std::sort of the pre-sorted order: %d   (b) stable order: %d   (c) reverse stable order: %d\n", trials, diff_a, diff_b, diff_c);
    return 0;
}
