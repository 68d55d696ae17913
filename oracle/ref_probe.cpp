// ref_probe — TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// A small harness of our own that links against the *reference's own objects*
// (read.o kmers.o arguments.o misc.o compiled from /root/reference/src where they
// lie; see oracle/Makefile) and dumps every public per-read field of the
// reference's `Read` (src/read.h:40-56) as hex floats, so the CPU restatement in
// oracle/flx_oracle.cpp and the HIP kernels can be pinned bit-for-bit against the
// real implementation. Output goes to stdout, one line per read / child read.
//
// usage: ref_probe <reads.bin> <kmer_queries.bin|-> -- <filtlong args ...>
//   reads.bin : u64 N, then per read { u32 name_len, name, u32 L, seq[L], u8 has_qual, qual[L] }
//   kmer_queries.bin : u64 M, then M u32 16-mers; prints "K <hex> <0|1>" per query
//                      (Kmers::is_kmer_present, src/kmers.cpp:170-172)
//   filtlong args: passed verbatim to the reference's Arguments (src/arguments.cpp:124)
//
// line format:
//   R <name> <length> <length_score %a> <mean %a> <window %a> <passed> <first> <last> B<n> s-e ... C<n> s-e ...
//   children follow their parent, prefixed with "c " instead of "R ".

#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "read.h"
#include "kmers.h"
#include "arguments.h"

static void dump_read(const Read *r, const char *tag) {
    printf("%s %s %d %a %a %a %d %d %d B%zu", tag, r->m_name.c_str(), r->m_length, r->m_length_score,
           r->m_mean_quality, r->m_window_quality, r->m_passed ? 1 : 0, r->m_first_base_in_kmer,
           r->m_last_base_in_kmer, r->m_bad_ranges.size());
    for (auto &p : r->m_bad_ranges) printf(" %d-%d", p.first, p.second);
    printf(" C%zu", r->m_child_read_ranges.size());
    for (auto &p : r->m_child_read_ranges) printf(" %d-%d", p.first, p.second);
    printf("\n");
    for (auto c : r->m_child_reads) dump_read(c, "c");
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: ref_probe reads.bin queries.bin|- -- <filtlong args>\n");
        return 2;
    }
    const char *reads_path = argv[1];
    const char *query_path = argv[2];
    int sep = 3;
    if (strcmp(argv[sep], "--") != 0) { fprintf(stderr, "expected --\n"); return 2; }
    std::vector<char *> fargv;
    fargv.push_back((char *)"filtlong");
    for (int i = sep + 1; i < argc; ++i) fargv.push_back(argv[i]);
    Arguments args((int)fargv.size(), fargv.data());
    if (args.parsing_result != GOOD) { fprintf(stderr, "ref_probe: bad filtlong args\n"); return 2; }

    Kmers kmers;
    if (args.assembly_set) kmers.add_assembly_fasta(args.assembly);
    if (args.short_reads.size() > 0) kmers.add_read_fastqs(args.short_reads);
    printf("E %d\n", kmers.empty() ? 1 : 0);

    if (strcmp(query_path, "-") != 0) {
        FILE *q = fopen(query_path, "rb");
        if (!q) { perror("queries"); return 2; }
        uint64_t m = 0;
        if (fread(&m, 8, 1, q) != 1) return 2;
        std::vector<uint32_t> ks(m);
        if (m && fread(ks.data(), 4, m, q) != m) return 2;
        fclose(q);
        for (uint64_t i = 0; i < m; ++i) printf("K %08x %d\n", ks[i], kmers.is_kmer_present(ks[i]) ? 1 : 0);
    }

    FILE *f = fopen(reads_path, "rb");
    if (!f) { perror("reads"); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) return 2;
    std::vector<char> seq, qual;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t nl = 0, L = 0;
        uint8_t hq = 0;
        if (fread(&nl, 4, 1, f) != 1) return 2;
        std::string name(nl, ' ');
        if (nl && fread(&name[0], 1, nl, f) != nl) return 2;
        if (fread(&L, 4, 1, f) != 1) return 2;
        seq.assign(L + 1, 0);
        if (L && fread(seq.data(), 1, L, f) != L) return 2;
        if (fread(&hq, 1, 1, f) != 1) return 2;
        qual.assign(L + 1, 0);
        if (hq && L && fread(qual.data(), 1, L, f) != L) return 2;
        Read *r = new Read(name, seq.data(), qual.data(), (int)L, &kmers, &args);
        dump_read(r, "R");
        delete r;
    }
    fclose(f);
    return 0;
}
