// reference.h — the reference files of -a / -1 / -2 as a STREAM into the device-side 16-mer set (src/kmers.cpp:75-134).
// Included by main.cpp only.
//
// The reference reads a record at a time through kseq and hashes it on the spot: constant memory, and a progress line
// "\r  file (N bp)" whenever 483 611 more bases have been hashed (src/kmers.cpp:123-126), once more at the end, then "\n".
// Here: a regular file — plain or gzip — goes through the block-wise reader (gzblocks.h: a block is inflated, the records that
// are complete inside it are parsed, the unfinished tail moves on), the sequences of at least 16 bases are packed back to back
// and handed to flx_kmerset_add_assembly / flx_kmerset_add_short_reads in batches of at most FLX_CLI_REF_BATCH_BYTES (256 MiB);
// the C ABI takes any number of such calls, in file order (include/filtlong_hip.h).  Host memory is O(block + batch) whatever the
// size of the file (round 4 held three copies of every sequence).  The progress lines are printed record by record with the
// reference's rule, so stderr is the reference's byte for byte.  Pipes and empty files keep the in-memory reader (fastx.h: Input).
#pragma once
#include "fastx.h"
#include "gzblocks.h"

static void print_hash_progress(const std::string &filename, long long base_count) {  // src/kmers.cpp (print_hash_progress)
    std::cerr << "\r  " << filename << " (" << int_to_string(base_count) << " bp)";
}

// hashes one reference file into sets[0 .. n_sets); returns the number of sequences (those shorter than 16 bases count too,
// src/kmers.cpp:96-100); ok = false: the library refused a batch (flx_last_error says why)
static int hash_reference(const std::string &filename, flx_kmerset *const *sets, int n_sets, bool short_reads, bool &ok) {
    int n = 0;
    long long bases = 0, last_progress = 0;
    size_t batch_bytes = (size_t)256 << 20;
    if (const char *e = getenv("FLX_CLI_REF_BATCH_BYTES")) batch_bytes = std::max<size_t>(16, (size_t)atoll(e));  // tests force many batches
    std::vector<uint8_t> pack;
    std::vector<uint64_t> offsets;
    std::vector<int64_t> lengths;
    ok = true;
    auto flush = [&]() {
        if (offsets.empty() || !ok) return;
        for (int k = 0; k < n_sets && ok; ++k) {
            const int rc = short_reads ? flx_kmerset_add_short_reads(sets[k], pack.data(), offsets.data(), lengths.data(), offsets.size())
                                       : flx_kmerset_add_assembly(sets[k], pack.data(), offsets.data(), lengths.data(), offsets.size());
            ok = rc == FLX_OK;
        }
        pack.clear();
        offsets.clear();
        lengths.clear();
    };
    auto take = [&](const Record &r) {  // one pass of the loop of src/kmers.cpp:91-127
        ++n;
        if (r.seq.size() < 16) return;
        bases += (long long)r.seq.size();
        if (pack.size() + r.seq.size() > batch_bytes) flush();
        offsets.push_back(pack.size());
        lengths.push_back((int64_t)r.seq.size());
        pack.insert(pack.end(), (const uint8_t *)r.seq.p, (const uint8_t *)r.seq.p + r.seq.size());
        if (bases - last_progress >= 483611) {
            last_progress = bases;
            print_hash_progress(filename, bases);
        }
    };
    BlockReader blocks;
    if (!getenv("FLX_CLI_NO_STREAM") && blocks.open(filename, false)) {
        Parsed batch;
        while (ok && blocks.next(batch)) {  // any error ends the loop silently, like `while ((l = kseq_read(seq)) >= 0)`
            for (const Record &r : batch.recs) take(r);
            if (batch.status <= -2) break;
        }
    } else {
        Input data;
        std::deque<std::string> arena;
        if (data.open(filename)) {
            Parser p(data, arena);
            Record r;
            while (ok && p.next(r) >= 0) {
                take(r);
                arena.clear();  // (the record's multi-line fields have been copied)
            }
        }
    }
    flush();
    print_hash_progress(filename, bases);
    std::cerr << "\n";
    return n;
}
