// kseq_probe.cpp — TEST INFRASTRUCTURE: what the reference's own reader makes of a file.
//
// Our harness around the reference's vendored kseq.h (compiled where it lies: -I$(REF)/src, nothing copied) over zlib's gzread,
// exactly as src/main.cpp:70-72 and src/kmers.cpp:90-91 set it up (KSEQ_INIT(gzFile, gzread), gzopen(path, "r")).  Prints one
// line in the format of the drop-in's FLX_CLI_PARSE_ONLY mode: record count, the first negative return of kseq_read
// (-1 end of file, -2 truncated quality, -3 stream error), the name at a -2, and an FNV-1a digest of every field of every record.
// tests/test_cli_damaged_gzip.py compares the drop-in's three ingest paths with it on damaged gzip files.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "kseq.h"
KSEQ_INIT(gzFile, gzread)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: kseq_probe FILE\n"); return 2; }
    gzFile fp = gzopen(argv[1], "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    kseq_t *seq = kseq_init(fp);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const char *p, size_t n) {
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        h ^= 0xff; h *= 1099511628211ull;
    };
    unsigned long long records = 0;
    long long l;
    for (;;) {
        l = kseq_read(seq);
        if (l < 0) break;
        mix(seq->name.s, seq->name.l);
        mix(seq->comment.s, seq->comment.l);
        mix(seq->seq.s, seq->seq.l);
        mix(seq->qual.s, seq->qual.l);
        h ^= (uint64_t)(seq->is_fastq != 0); h *= 1099511628211ull;
        ++records;
    }
    printf("records %llu status %lld bad %s parallel 0 digest %llu\n", records, l, l == -2 && seq->name.s ? seq->name.s : "", (unsigned long long)h);
    kseq_destroy(seq);
    gzclose(fp);
    return 0;
}
