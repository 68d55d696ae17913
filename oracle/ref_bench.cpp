// ref_bench — CPU BASELINE HARNESS.  TEST / BENCH INFRASTRUCTURE ONLY.
//
// Times the *reference's own compiled objects* (read.o kmers.o arguments.o misc.o built from
// /root/reference/src by oracle/Makefile) on the synthetic Phred-only workload of SURVEY.md
// §8(d), in memory, so that the number is like-for-like with the GPU kernel window
// ("scored + sorted", no FASTQ parse, no stdout):
//   score : one `Read::Read(...)` per read            (reference src/read.cpp:25-144)
//   rank  : statistics + normalise + Read::set_final_score + std::sort + cut walk
//           (reference src/main.cpp:169-261; those loops are inline in main(), so this
//           harness drives the reference's Read::set_final_score and libstdc++'s std::sort
//           with the same comparator over the same Read* vector)
// The reference is single-threaded, so this is a 1-core number.
//
// usage: ref_bench <n_reads> <fixed_len|0> <target_bases> [seed] [first_read_index]
//        ref_bench kmer <reads.fastq> <target_bases> [filtlong options: -a ref.fasta | -1 r1.fq -2 r2.fq, --trim, --split N]
//            k-mer mode (BASELINE configs[2]/[3]): the reads are loaded into memory first (not timed); timed separately:
//            the reference's set build (Kmers::add_assembly_fasta / add_read_fastqs), one Read::Read per read (incl. its
//            children), and the rank stage over reads2 (children in place of their parents, src/main.cpp:138-147).
// prints one JSON line on stdout.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "read.h"
#include "kmers.h"
#include "arguments.h"
#include "synth.h"

static double now_s() {
    using namespace std::chrono;
    return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

static int kmer_mode(int argc, char **argv) {
    std::string reads_path = argv[2];
    long long target_arg = atoll(argv[3]);
    std::string t = std::to_string(target_arg);
    std::vector<const char *> fargv = {"filtlong"};
    for (int i = 4; i < argc; ++i) fargv.push_back(argv[i]);
    fargv.push_back("--target_bases");
    fargv.push_back(t.c_str());
    fargv.push_back(reads_path.c_str());
    Arguments args((int)fargv.size(), (char **)fargv.data());
    if (args.parsing_result != GOOD) return 2;

    // load the reads (4-line FASTQ written by bench.py); not timed
    std::vector<std::string> names, seqs, quals;
    {
        FILE *f = fopen(reads_path.c_str(), "r");
        if (!f) return 2;
        char *line = nullptr;
        size_t cap = 0;
        ssize_t m;
        int k = 0;
        while ((m = getline(&line, &cap, f)) > 0) {
            while (m > 0 && (line[m - 1] == '\n' || line[m - 1] == '\r')) --m;
            std::string v(line, (size_t)m);
            if (k == 0) names.push_back(v.substr(1));
            else if (k == 1) seqs.push_back(v);
            else if (k == 3) quals.push_back(v);
            k = (k + 1) & 3;
        }
        free(line);
        fclose(f);
    }
    const long long n = (long long)seqs.size();
    long long total_bases = 0;
    for (auto &q : seqs) total_bases += (long long)q.size();

    double t0 = now_s();
    Kmers kmers;
    if (args.assembly_set) kmers.add_assembly_fasta(args.assembly);
    if (args.short_reads.size() > 0) kmers.add_read_fastqs(args.short_reads);
    double t1 = now_s();

    std::vector<Read *> reads;
    reads.reserve(n);
    for (long long i = 0; i < n; ++i)
        reads.push_back(new Read(names[i], &seqs[i][0], &quals[i][0], (int)seqs[i].size(), &kmers, &args));
    double t2 = now_s();

    std::vector<Read *> reads2;  // src/main.cpp:138-147
    long long children = 0;
    for (auto r : reads) {
        if (r->m_child_reads.size() > 0) {
            for (auto c : r->m_child_reads) reads2.push_back(c);
            children += (long long)r->m_child_reads.size();
        } else reads2.push_back(r);
    }
    double qmin = 100.0, qmax = 0.0, qsum = 0.0;
    for (auto r : reads2) {
        qsum += r->m_mean_quality;
        qmax = std::max(qmax, r->m_mean_quality);
        qmin = std::min(qmin, r->m_mean_quality);
    }
    double qmean = qsum / reads2.size(), ssum = 0.0;
    for (auto r : reads2) {
        double d = r->m_mean_quality - qmean;
        ssum += d * d;
    }
    double qstd = sqrt(ssum / reads2.size());
    double zmin = qstd > 0 ? (qmin - qmean) / qstd : 1.0, zmax = qstd > 0 ? (qmax - qmean) / qstd : 1.0;
    for (auto r : reads2) {
        double ratio = r->m_window_quality / r->m_mean_quality;
        if (ratio > 1.0) ratio = 1.0;
        double z = (r->m_mean_quality - qmean) / qstd;
        r->m_mean_quality = 100.0 * (z - zmin) / (zmax - zmin);
        r->m_window_quality = r->m_mean_quality * ratio;
        r->set_final_score(args.length_weight, args.mean_q_weight, args.window_q_weight);
    }
    long long kept = 0, kept_reads = 0;
    if (target_arg < total_bases) {
        std::sort(reads2.begin(), reads2.end(),
                  [](const Read *a, const Read *b) { return a->m_final_score > b->m_final_score; });
        for (auto r : reads2) {
            if (r->m_passed && kept < target_arg) { kept += r->m_length; ++kept_reads; }
            else r->m_passed = false;
        }
    }
    double t3 = now_s();
    printf("{\"reads\": %lld, \"bases\": %lld, \"set_build_s\": %.6f, \"score_s\": %.6f, \"rank_s\": %.6f, "
           "\"mbases_per_s\": %.3f, \"children\": %lld, \"kept_bases\": %lld, \"kept_reads\": %lld}\n",
           n, total_bases, t1 - t0, t2 - t1, t3 - t2, total_bases / ((t2 - t1) + (t3 - t2)) / 1e6, children, kept, kept_reads);
    for (auto r : reads) delete r;
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && std::string(argv[1]) == "kmer") return kmer_mode(argc, argv);
    if (argc < 4) {
        fprintf(stderr, "usage: ref_bench n_reads fixed_len|0 target_bases [seed] [first_read]\n");
        return 2;
    }
    long long n = atoll(argv[1]);
    int fixed_len = atoi(argv[2]);
    long long target_arg = atoll(argv[3]);
    uint64_t seed = argc > 4 ? strtoull(argv[4], 0, 10) : FLX_SYNTH_SEED;
    uint64_t first = argc > 5 ? strtoull(argv[5], 0, 10) : 0;

    // the reference's Arguments needs an existing input path and one threshold
    std::string t = std::to_string(target_arg);
    const char *fargv[] = {"filtlong", "--target_bases", t.c_str(), argv[0]};
    Arguments args(4, (char **)fargv);
    if (args.parsing_result != GOOD) return 2;
    Kmers kmers;  // empty => Phred mode (src/read.cpp:35)

    // lengths: gamma(k=4), mean 10 kbp, clamp [200, 200000] unless fixed
    std::vector<int> len(n);
    long long total_bases = 0;
    for (long long i = 0; i < n; ++i) {
        if (fixed_len > 0) len[i] = fixed_len;
        else {
            double g = 0.0;
            for (int j = 0; j < 4; ++j) {
                uint64_t h = flx_mix(seed, FLX_STREAM_LEN, first + i, j);
                double u = ((double)(h >> 11) + 0.5) / 9007199254740992.0;
                g += -log(u);
            }
            long long L = llround(2500.0 * g);
            if (L < 200) L = 200;
            if (L > 200000) L = 200000;
            len[i] = (int)L;
        }
        total_bases += len[i];
    }

    // generate all quality strings up front (not timed); seq is never touched in Phred mode
    std::vector<char> plane((size_t)total_bases + 1);
    {
        size_t off = 0;
        for (long long i = 0; i < n; ++i) {
            int mu = flx_synth_mu(seed, first + i);
            for (int p = 0; p < len[i]; ++p) plane[off + p] = (char)flx_synth_qual(seed, first + i, p, mu);
            off += len[i];
        }
    }

    std::vector<Read *> reads;
    reads.reserve(n);
    double t0 = now_s();
    {
        size_t off = 0;
        for (long long i = 0; i < n; ++i) {
            reads.push_back(new Read("r" + std::to_string(i), &plane[off], &plane[off], len[i], &kmers, &args));
            off += len[i];
        }
    }
    double t1 = now_s();

    // ---- rank stage (drives the reference's set_final_score + std::sort) ----
    double qmin = 100.0, qmax = 0.0, qsum = 0.0;
    for (auto r : reads) {
        qsum += r->m_mean_quality;
        qmax = std::max(qmax, r->m_mean_quality);
        qmin = std::min(qmin, r->m_mean_quality);
    }
    double qmean = qsum / reads.size(), ssum = 0.0;
    for (auto r : reads) {
        double d = r->m_mean_quality - qmean;
        ssum += d * d;
    }
    double qstd = sqrt(ssum / reads.size());
    double zmin = qstd > 0 ? (qmin - qmean) / qstd : 1.0, zmax = qstd > 0 ? (qmax - qmean) / qstd : 1.0;
    for (auto r : reads) {
        double ratio = r->m_window_quality / r->m_mean_quality;
        if (ratio > 1.0) ratio = 1.0;
        double z = (r->m_mean_quality - qmean) / qstd;
        r->m_mean_quality = 100.0 * (z - zmin) / (zmax - zmin);
        r->m_window_quality = r->m_mean_quality * ratio;
        r->set_final_score(args.length_weight, args.mean_q_weight, args.window_q_weight);
    }
    long long kept = 0, kept_reads = 0;
    if (target_arg < total_bases) {
        std::sort(reads.begin(), reads.end(),
                  [](const Read *a, const Read *b) { return a->m_final_score > b->m_final_score; });
        for (auto r : reads) {
            if (r->m_passed && kept < target_arg) { kept += r->m_length; ++kept_reads; }
            else r->m_passed = false;
        }
    }
    double t2 = now_s();

    printf("{\"reads\": %lld, \"bases\": %lld, \"score_s\": %.6f, \"rank_s\": %.6f, \"total_s\": %.6f, "
           "\"mbases_per_s\": %.3f, \"kept_bases\": %lld, \"kept_reads\": %lld}\n",
           n, total_bases, t1 - t0, t2 - t1, t2 - t0, total_bases / (t2 - t0) / 1e6, kept, kept_reads);
    for (auto r : reads) delete r;
    return 0;
}
