// parse_only.h — the parsers without a GPU: a digest of every field of the input (the CPU tests compare the sequential parser, the
// concurrent one, the block-wise reader and the ranks' shares on generated odd files).  Included by main.cpp only.
#pragma once
#include "fastx.h"
#include "gzblocks.h"

// FLX_CLI_PARSE_ONLY=seq|par|blk|unit|ranks:W: parse the input, print a digest of every field and exit (no GPU needed).  The CPU tests
// compare the sequential parser with the concurrent one and with the block-wise reader on generated odd files.
static int parse_only(const std::string &path, const char *mode) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const View &v) {
        for (size_t i = 0; i < v.n; ++i) { h ^= (unsigned char)v.p[i]; h *= 1099511628211ull; }
        h ^= 0xff; h *= 1099511628211ull;
    };
    auto mix_all = [&](const Parsed &pd) {
        for (const Record &r : pd.recs) { mix(r.name); mix(r.comment); mix(r.seq); mix(r.qual); h ^= r.is_fastq; h *= 1099511628211ull; }
    };
    Parsed parsed;
    bool par = false;
    size_t n_records = 0;
    if (mode[0] == 'b' || mode[0] == 'u') {  // blocks: the streaming reader, FLX_CLI_BLOCK_BYTES per block
        BlockReader rd;
        if (!rd.open(path, true)) { std::cerr << "Error reading " << path << "\n"; return 1; }
        UnitIndex idx;
        const uint64_t h_blocks_start = h;
        while (rd.next(parsed)) {
            mix_all(parsed);
            for (const Record &r : parsed.recs) idx.note_record(rd.points, rd.offset_of(r.name.p - 1), n_records++);
            if (parsed.status <= -2) break;
        }
        if (rd.io_error) { std::cerr << "Error reading " << path << "\n"; return 1; }
        std::cerr << "inflate: " << rd.z.parallel_bytes() << " of " << rd.z.total_out() << " bytes from the parallel path (" << rd.z.zlib_tail_bytes() << " by zlib behind the marker decoder), " << rd.z.rounds()
                  << " round(s), " << rd.z.dropped_chunks() << " chunk(s) dropped\n";
        if (mode[0] == 'u' && parsed.status > -2) {
            // units: every piece between two access points (FLX_CLI_SPAN_BYTES apart) inflated and parsed on its own, on
            // several threads, as the output pass does; digest of the pieces in order
            idx.finish(rd.points, rd.end_offset(), n_records);
            std::vector<Parsed> got(idx.units());
            std::vector<std::vector<char>> texts(idx.units());
            std::vector<int> ok(idx.units(), 1);
            parallel_for(idx.units(), [&](size_t j) {
                if (idx.start[j + 1] == idx.start[j]) return;
                if (!inflate_range(rd.file, rd.points[j], idx.start[j], idx.start[j + 1], texts[j])) { ok[j] = 0; return; }
                Input view;
                view.p = texts[j].data();
                view.n = texts[j].size();
                parse_sequential(view, got[j]);
                if (got[j].recs.size() != idx.first_rec[j + 1] - idx.first_rec[j]) ok[j] = 0;
            });
            h = h_blocks_start;
            for (size_t j = 0; j < idx.units(); ++j) {
                if (!ok[j]) { std::cerr << "unit " << j << " failed\n"; return 1; }
                mix_all(got[j]);
            }
            std::cerr << "units " << idx.units() << " points " << rd.points.size() << "\n";
        }
        std::cout << "records " << n_records << " status " << parsed.status << " bad " << parsed.bad.name << " parallel 0 digest " << h << "\n";
        return 0;
    }
    Input data;
    if (!data.open(path)) { std::cerr << "Error reading " << path << "\n"; return 1; }
    if (strncmp(mode, "ranks:", 6) == 0) {
        // ranks:W — every rank's share of the file (parse_rank_range), one after the other; accepted only if EVERY rank accepts its
        // share, as the command line decides it from a sum over the ranks — else the whole file, as every rank would parse it then
        const int world = std::max(1, atoi(mode + 6));
        std::vector<Parsed> share((size_t)world);
        bool all = data.map != nullptr;
        for (int r = 0; r < world && all; ++r) all = parse_rank_range(data, r, world, share[(size_t)r]);
        size_t n = 0;
        if (all) {
            for (const Parsed &sh : share) { mix_all(sh); n += sh.recs.size(); }
        } else {
            parse_all(data, parsed);
            mix_all(parsed);
            n = parsed.recs.size();
        }
        std::cout << "records " << n << " status " << parsed.status << " bad " << parsed.bad.name << " parallel " << (all ? 1 : 0) << " digest " << h << "\n";
        return 0;
    }
    if (mode[0] == 's') parse_sequential(data, parsed);
    else par = parse_all(data, parsed);
    mix_all(parsed);
    std::cout << "records " << parsed.recs.size() << " status " << parsed.status << " bad " << parsed.bad.name << " parallel "
              << (par ? 1 : 0) << " digest " << h << "\n";
    return 0;
}

