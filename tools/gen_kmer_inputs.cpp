// gen_kmer_inputs — the k-mer-mode inputs of SURVEY §8(d) as files, on all host threads: the 5 Mbp reference (FASTA), long reads
// drawn from it (FASTQ, definition of filtlong_amd/synth.py: seq_read — substitutions by an integer threshold, a junk block in 30 %
// of the reads), and error-free short-read pairs of 100 bp (C4: -1 a forward substring, -2 the reverse complement of the substring
// 350 bp downstream).  BENCH / TEST INFRASTRUCTURE.
//   usage: gen_kmer_inputs <dir> <n_long_reads> <n_pairs> [ref_len=5000000]   -> dir/ref.fasta reads.fastq sr_1.fastq sr_2.fastq
//   prints: long-read bases, short-read bases
// build: g++ -O2 -std=c++17 -pthread -Ioracle -o tools/gen_kmer_inputs tools/gen_kmer_inputs.cpp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
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
    return (int)std::min<long long>(200000, std::max<long long>(200, L));
}

// items [0, n) in batches on all threads, written in order
static void write_parallel(const std::string &path, long long n, long long batch, const std::function<void(long long, long long, std::string &)> &make) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(1); }
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    for (long long b0 = 0; b0 < n; b0 += (long long)T * batch) {
        std::vector<std::string> part(T);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                const long long lo = b0 + (long long)t * batch, hi = std::min(n, lo + batch);
                if (lo < hi) make(lo, hi, part[t]);
            });
        for (auto &x : th) x.join();
        for (auto &s : part) fwrite(s.data(), 1, s.size(), f);
    }
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: gen_kmer_inputs dir n_long_reads n_pairs [ref_len]\n"); return 2; }
    const std::string dir = argv[1];
    const long long n_reads = atoll(argv[2]), n_pairs = atoll(argv[3]);
    const uint64_t ref_len = argc > 4 ? strtoull(argv[4], 0, 10) : 5000000ull;
    const uint64_t seed = FLX_SYNTH_SEED;
    std::vector<uint8_t> ref(ref_len);
    for (uint64_t i = 0; i < ref_len; ++i) ref[i] = flx_synth_base(seed, FLX_STREAM_REF, 0, i);
    {
        FILE *f = fopen((dir + "/ref.fasta").c_str(), "wb");
        if (!f) return 1;
        fputs(">ref\n", f);
        fwrite(ref.data(), 1, ref.size(), f);
        fputc('\n', f);
        fclose(f);
    }
    std::vector<long long> bases_of(1, 0);
    long long long_bases = 0;
    for (long long i = 0; i < n_reads; ++i) long_bases += length_of(seed, (uint64_t)i);
    write_parallel(dir + "/reads.fastq", n_reads, 256, [&](long long lo, long long hi, std::string &s) {
        for (long long i = lo; i < hi; ++i) {
            const uint64_t r = (uint64_t)i;
            const int L = length_of(seed, r);
            const uint64_t start = ref_len > (uint64_t)L ? flx_mix(seed, FLX_STREAM_START, r, 0) % (ref_len - (uint64_t)L) : 0;
            const unsigned erate = (unsigned)(flx_mix(seed, FLX_STREAM_ERATE, r, 0) % 13);
            s += "@r" + std::to_string(r) + "\n";
            const size_t at = s.size();
            s.resize(at + (size_t)L);
            for (int p = 0; p < L; ++p) {
                const uint64_t h = flx_mix(seed, FLX_STREAM_SUB, r, (uint64_t)p >> 2);
                const uint32_t f = (uint32_t)(h >> (16 * (p & 3))) & 0xffffu;
                s[at + p] = ((f & 0xff) % 100 < erate) ? "ACGT"[(f >> 8) & 3] : (char)ref[(start + (uint64_t)p) % ref_len];
            }
            if (L > 3000 && flx_mix(seed, FLX_STREAM_JUNK, r, 0) % 10 < 3) {
                const int js = 500 + (int)(flx_mix(seed, FLX_STREAM_JUNK, r, 1) % (uint64_t)(L - 2000));
                const int je = std::min(js + 800, L);
                for (int p = js; p < je; ++p) s[at + p] = (char)flx_synth_base(seed, FLX_STREAM_BASE, r, (uint64_t)p);
            }
            s += "\n+\n";
            s.append((size_t)L, 'I');
            s += "\n";
        }
    });
    static const char comp[256] = {0};
    (void)comp;
    auto pair_start = [&](long long k) { return flx_mix(seed, FLX_STREAM_START, (uint64_t)k + (1ull << 40), 0) % (ref_len - 450); };
    write_parallel(dir + "/sr_1.fastq", n_pairs, 65536, [&](long long lo, long long hi, std::string &s) {
        const std::string q(100, 'I');
        for (long long k = lo; k < hi; ++k) {
            const uint64_t st = pair_start(k);
            s += "@p" + std::to_string(k) + "/1\n";
            s.append((const char *)ref.data() + st, 100);
            s += "\n+\n" + q + "\n";
        }
    });
    write_parallel(dir + "/sr_2.fastq", n_pairs, 65536, [&](long long lo, long long hi, std::string &s) {
        const std::string q(100, 'I');
        for (long long k = lo; k < hi; ++k) {
            const uint64_t st = pair_start(k) + 350;
            s += "@p" + std::to_string(k) + "/2\n";
            const size_t at = s.size();
            s.resize(at + 100);
            for (int p = 0; p < 100; ++p) {
                const char c = (char)ref[st + 99 - (uint64_t)p];
                s[at + p] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
            }
            s += "\n+\n" + q + "\n";
        }
    });
    printf("%lld %lld\n", long_bases, 200 * n_pairs);
    return 0;
}
