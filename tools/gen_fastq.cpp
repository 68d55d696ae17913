// gen_fastq — writes the synthetic Phred-only workload of SURVEY §8(d) as a FASTQ file, fast (all host threads): the same
// lengths and quality bytes as filtlong_amd/synth.py / oracle/synth.h, sequence = "ACGT" repeated (never looked at in Phred
// mode).  BENCH / TEST INFRASTRUCTURE.   usage: gen_fastq <n_reads> <out.fastq> [first_read]
// build: g++ -O2 -std=c++17 -pthread -Ioracle -o tools/gen_fastq tools/gen_fastq.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "synth.h"

static int length_of(uint64_t seed, uint64_t read) {
    double g = 0.0;
    for (int j = 0; j < 4; ++j) {
        uint64_t h = flx_mix(seed, FLX_STREAM_LEN, read, j);
        double u = ((double)(h >> 11) + 0.5) / 9007199254740992.0;
        g += -log(u);
    }
    long long L = llround(2500.0 * g);
    if (L < 200) L = 200;
    if (L > 200000) L = 200000;
    return (int)L;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: gen_fastq n_reads out.fastq [first_read]\n"); return 2; }
    const long long n = atoll(argv[1]);
    const uint64_t first = argc > 3 ? strtoull(argv[3], 0, 10) : 0;
    const uint64_t seed = FLX_SYNTH_SEED;
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 1;
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    const long long batch = 4096;
    long long bases = 0;
    for (long long b0 = 0; b0 < n; b0 += (long long)T * batch) {
        std::vector<std::string> part(T);
        std::vector<long long> pb(T, 0);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                const long long lo = b0 + (long long)t * batch, hi = std::min(n, lo + batch);
                std::string &s = part[t];
                for (long long i = lo; i < hi; ++i) {
                    const uint64_t r = first + (uint64_t)i;
                    const int L = length_of(seed, r);
                    pb[t] += L;
                    s += "@r" + std::to_string(r) + "\n";
                    const size_t at = s.size();
                    s.resize(at + (size_t)L);
                    for (int p = 0; p < L; ++p) s[at + p] = "ACGT"[p & 3];
                    s += "\n+\n";
                    const size_t q = s.size();
                    s.resize(q + (size_t)L);
                    const int mu = flx_synth_mu(seed, r);
                    for (int p = 0; p < L; ++p) s[q + p] = (char)flx_synth_qual(seed, r, (uint64_t)p, mu);
                    s += "\n";
                }
            });
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < T; ++t) {
            fwrite(part[t].data(), 1, part[t].size(), f);
            bases += pb[t];
        }
    }
    fclose(f);
    printf("%lld\n", bases);
    return 0;
}
