// filtlong-amd — C++ host of the MI355X-native Filtlong hot path: same command line, same stdout FASTQ/FASTA,
// same stderr lines and exit codes as the reference binary, with the per-read scoring and the global rank/cut
// done on the GPU through the C ABI (include/filtlong_hip.h).
//
// What this file mirrors of the reference (rrwick/Filtlong v0.3.1, paths relative to its root):
//   arguments       src/arguments.cpp:28-393  flags, unit suffixes, validation order and messages
//   input parsing   src/kseq.h:176-224         FASTA/FASTQ records, multi-line, "\r\n", gz via zlib
//   orchestration   src/main.cpp:37-321        sections printed to stderr, reads2 gather, output order
//   formatting      src/misc.cpp:24-49         2-decimal doubles, locale-grouped integers
// Differences by design: a plain file is mapped and parsed once (the reference parses it twice, main.cpp:70-127 and
// 264-313), a gzip file is streamed block by block (gzblocks.h); scoring is batched and streamed (flx_pipeline_*) instead of
// one Read object per record; ranks (one process per GPU) share the global stage through the library's communicator.
#include <zlib.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <deque>
#include <string_view>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <locale>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/filtlong_hip.h"

#include "format.h"
#include "args.h"
#include "fastx.h"
#include "gzblocks.h"
#include "parse_only.h"

// ------------------------------------------------------------------------------------------------ helpers
#include <chrono>
static int g_rank = 0, g_world = 1;          // multi-GPU: one process per GPU (RANK / WORLD_SIZE, or forked by --gpus N)
static std::string g_part_prefix;             // where the ranks leave their parts of the output
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool g_timing = false;
static double g_t0 = 0;
static void stage(const char *what) {  // FLX_CLI_TIMING=1: per-stage wall clock + resident memory on stderr (not part of the reference surface)
    if (!g_timing) return;
    const double t = now_s();
    long anon_kb = 0, file_kb = 0, hwm_kb = 0;
    if (FILE *f = fopen("/proc/self/status", "r")) {  // RssAnon = what the process owns; RssFile = resident pages of the mapped input
        char line[256];
        while (fgets(line, sizeof line, f)) {
            sscanf(line, "RssAnon: %ld kB", &anon_kb);
            sscanf(line, "RssFile: %ld kB", &file_kb);
            sscanf(line, "VmHWM: %ld kB", &hwm_kb);
        }
        fclose(f);
    }
    fprintf(stderr, "[timing] %-30s %8.3f s   RssAnon %7ld MiB  RssFile %7ld MiB  VmHWM %7ld MiB\n", what, t - g_t0, anon_kb >> 10,
            file_kb >> 10, hwm_kb >> 10);
    g_t0 = now_s();
}

static int fail_flx(flx_ctx *ctx, const char *what) {
    std::cerr << "Error: " << what << ": " << flx_last_error(ctx) << "\n";
    return 1;
}

#include "reference.h"
#include "output.h"
#include "ranks.h"

// The FLX_CLI_* environment variables this binary reads (test hooks and tuning knobs, README.md); any other FLX_CLI_* name is an
// error — a mistyped switch must not be ignored silently.  (The library checks the rest of the FLX_* names: flx_ctx_create.)
static int check_cli_environment() {
    static const char *const known[] = {
        "FLX_CLI_BLOCK_BYTES", "FLX_CLI_BLOCK_MB", "FLX_CLI_CHUNK_BYTES", "FLX_CLI_CHUNK_MB", "FLX_CLI_CLEAN_EXIT", "FLX_CLI_FORCE_STREAM",
        "FLX_CLI_FAIL_WRITE_RANK", "FLX_CLI_INFLATE_THREADS", "FLX_CLI_NO_STREAM", "FLX_CLI_ORDERED_OUTPUT", "FLX_CLI_PARALLEL_PARSE_MIN", "FLX_CLI_PARSE_ONLY",
        "FLX_CLI_PINFLATE", "FLX_CLI_PINFLATE_AHEAD_MB", "FLX_CLI_PINFLATE_CHUNK", "FLX_CLI_PINFLATE_MIN", "FLX_CLI_PINFLATE_TIMING",
        "FLX_CLI_RANK_RANGES", "FLX_CLI_RANK_STREAM", "FLX_CLI_REF_BATCH_BYTES", "FLX_CLI_SPAN_BYTES", "FLX_CLI_THREADS", "FLX_CLI_TIMING",
    };
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "FLX_CLI_", 8) != 0) continue;
        const char *eq = strchr(*e, '=');
        const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
        bool ok = false;
        for (const char *k : known) ok = ok || name == k;
        if (!ok) {
            std::cerr << "Error: unknown environment variable " << name << " (the FLX_CLI_* switches are listed in README.md)\n";
            return 1;
        }
    }
    return 0;
}

int main(int argc, char **argv) {
    Args args;
    const ParsingResult pr = parse_args(argc, argv, args);
    if (pr == BAD) return 1;
    if (pr == HELP) return 0;
    if (pr == VERSION) { std::cout << "Filtlong v" << PROGRAM_VERSION << "\n"; return 0; }
    if (const int bad_env = check_cli_environment()) return bad_env;
    if (const char *po = getenv("FLX_CLI_PARSE_ONLY")) return parse_only(args.input_reads, po);

    // ---- ranks: one process per GPU (north_star / SURVEY §8e; ranks.h) — under a launcher, or forked here by --gpus N ----
    std::string id_file;
    int id_pipe = -1;  // --gpus: the read end of this rank's pipe from rank 0
    if (const int rc = start_ranks(args, id_file, id_pipe); rc >= 0) return rc;
    JobGuard job_guard;  // rank 0 of --gpus: whatever way main() is left, no child and no part file stays behind
    if (g_rank < 0 || g_rank >= g_world) { std::cerr << "Error: RANK " << g_rank << " outside WORLD_SIZE " << g_world << "\n"; return 1; }
    if (args.gpus > 1 && g_world > 1 && id_file.empty()) g_shared_out = dup(1);  // (forked ranks share the job's stdout: see the output pass)
    if (g_rank > 0) {  // rank 0 speaks for the job
        if (!freopen("/dev/null", "w", stderr)) return 1;
        if (!freopen("/dev/null", "w", stdout)) return 1;
    }

    std::cerr << "\n";
    g_timing = getenv("FLX_CLI_TIMING") != nullptr;
    g_t0 = now_s();
    if (g_timing) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "[timing] main() reached at wall clock %.3f\n", ts.tv_sec % 100000 + ts.tv_nsec * 1e-9);
    }
    // The context (HIP runtime start-up, ~0.1 s) is created on a second thread while this one maps, parses and checks the input,
    // when nothing needs it before the scoring: one rank, no reference 16-mers to build.
    flx_ctx *ctx = nullptr;
    int ctx_rc = FLX_OK;
    std::thread ctx_thread;
    struct ThreadJoiner {
        std::thread &t;
        ~ThreadJoiner() { if (t.joinable()) t.join(); }
    } ctx_joiner{ctx_thread};
    {
        const char *dev = getenv("FLX_DEVICE");
        int ordinal = dev ? atoi(dev) : 0;
        if (!dev && g_world > 1) ordinal = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : g_rank;
        if (g_world == 1 && !args.assembly_set && args.short_reads.empty() && !g_timing) {
            ctx_thread = std::thread([&ctx, &ctx_rc, ordinal] { ctx_rc = flx_ctx_create(ordinal, &ctx); });
        } else if (flx_ctx_create(ordinal, &ctx) != FLX_OK) {
            std::cerr << "Error: " << flx_last_error(nullptr) << "\n";
            return 1;
        }
    }
    auto ctx_ready = [&]() -> bool {  // before the first use of `ctx`
        if (ctx_thread.joinable()) ctx_thread.join();
        if (ctx_rc != FLX_OK) {
            std::cerr << "Error: " << flx_last_error(nullptr) << "\n";
            ctx_rc = FLX_OK;  // reported once
            return false;
        }
        return ctx != nullptr;
    };
    if (g_world > 1)
        if (const int rc = exchange_communicator_id(ctx, id_file, id_pipe); rc >= 0) return rc;

    stage("context");
    // ---- reference 16-mers (src/main.cpp:51-59, src/kmers.cpp:50-72) --------------------------------------
    flx_kmerset *kmers = nullptr;
    bool kmers_empty = true;
    if (args.assembly_set || !args.short_reads.empty()) {
        if (flx_kmerset_create(ctx, &kmers) != FLX_OK) return fail_flx(ctx, "k-mer set");
        if (args.assembly_set) {
            std::cerr << "Hashing 16-mers from assembly\n";
            std::cerr << "  " << args.assembly << "\n";
            // the reference prints the set's size after the assembly alone (src/kmers.cpp:60-72); with short reads to follow that
            // takes a second set, which is fed the same batches and dropped once it has been counted
            flx_kmerset *alone = nullptr;
            if (!args.short_reads.empty() && flx_kmerset_create(ctx, &alone) != FLX_OK) return fail_flx(ctx, "k-mer set");
            flx_kmerset *both[2] = {kmers, alone};
            bool ok = true;
            const int count = hash_reference(args.assembly, both, alone ? 2 : 1, false, ok);
            if (!ok) return fail_flx(ctx, "assembly");
            flx_kmerset *counted = alone ? alone : kmers;
            if (flx_kmerset_finalize(counted) != FLX_OK) return fail_flx(ctx, alone ? "assembly" : "k-mer set");
            std::cerr << "  " << int_to_string(count) << " " << (count == 1 ? "contig" : "contigs") << ", "
                      << int_to_string((long long)flx_kmerset_size(counted)) << " 16-mers\n\n";
            if (alone) flx_kmerset_destroy(alone);
        }
        if (!args.short_reads.empty()) {
            std::cerr << "Hashing 16-mers from short reads\n";
            int count = 0;
            for (auto &f : args.short_reads) {
                bool ok = true;
                count += hash_reference(f, &kmers, 1, true, ok);
                if (!ok) return fail_flx(ctx, "short reads");
            }
            if (flx_kmerset_finalize(kmers) != FLX_OK) return fail_flx(ctx, "k-mer set");
            std::cerr << "  " << int_to_string(count) << " reads, " << int_to_string((long long)flx_kmerset_size(kmers)) << " 16-mers\n\n";
        }
        kmers_empty = flx_kmerset_size(kmers) == 0;
    }

    stage("reference 16-mers");
    // ---- pass 1: parse, checks (src/main.cpp:63-130) -----------------------------------------------------
    if (!args.verbose) std::cerr << "Scoring long reads\n";
    // Two kinds of input.  A plain file is mapped and parsed in one piece (one batch): its record views stay valid, so the
    // output pass needs no second parse, and every rank can index it.  A gzip file is STREAMED on one GPU: a block is
    // inflated, its complete records are checked, packed and submitted, and the block's memory is reused; the output pass
    // inflates the file a second time, like the reference's pass 2 (src/main.cpp:263-313).  Pipes cannot be read twice and
    // several ranks need the record count before they score: both are inflated into memory.
    const int world = g_world, rank = g_rank;
    Input data;
    BlockReader blocks;
    bool streamed = false;
    {
        const int fd = ::open(args.input_reads.c_str(), O_RDONLY);
        if (fd < 0) { std::cerr << "Error reading " << args.input_reads << "\n"; return 1; }
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
        const bool gz = regular && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        ::close(fd);
        // Several ranks (round 5): a gzip input is streamed by every rank as well — rank 0 counts the records in a pass of its own, the
        // count fixes every rank's contiguous share, then every rank streams the file, runs the checks of src/main.cpp:84-117 over
        // ALL records (they are per-record facts: every rank finds the same error at the same record) and packs and scores only its
        // share.  Up to round 4 every rank inflated the whole file into its memory.  (--verbose keeps that path: on an error it
        // scores the reads in front of it, all of them on rank 0.)
        // By default for compressed files of 1 GiB and more: below that the whole text fits every rank's memory easily and one pass is
        // quicker than two (measured with 8 ranks on 0.37 GB of gzip: 4.95 s streamed, 4.63 s in memory, and no smaller resident set —
        // the HIP runtime, the pinned slots and the inflater's buffers are 3.5 GB per rank either way; profiles/r05_gz_ranks.log).
        // FLX_CLI_RANK_STREAM=1: always (tests), =0: never.
        const char *rs_env = getenv("FLX_CLI_RANK_STREAM");
        const bool rank_stream = !args.verbose && (rs_env ? rs_env[0] != '0' : (gz && st.st_size >= ((off_t)1 << 30)));
        streamed = regular && (world == 1 || rank_stream) && !getenv("FLX_CLI_NO_STREAM") && (gz || getenv("FLX_CLI_FORCE_STREAM"));
    }
    if (streamed ? !blocks.open(args.input_reads, true) : !data.open(args.input_reads)) { std::cerr << "Error reading " << args.input_reads << "\n"; return 1; }
    stage("read input file");
    uint64_t streamed_records_counted = UINT64_MAX;  // streamed input, several ranks: what rank 0's count pass saw
    uint64_t share_lo = 0, share_n = UINT64_MAX;  // streamed input, several ranks: this rank's records [share_lo, share_lo + share_n) of file order
    if (streamed && world > 1) {
        uint64_t n_all[2] = {0, 0};  // records, and whether the count pass could not open the file (every rank must learn that: advisor, round 5)
        if (rank == 0) {  // (a damaged stream counts the records in front of the damage: every rank ends there in its own pass)
            BlockReader counter;
            Parsed b;
            if (counter.open(args.input_reads, false)) {
                while (counter.next(b)) {
                    n_all[0] += b.recs.size();
                    if (b.status <= -2) break;
                }
            } else {
                n_all[1] = 1;
            }
        }
        if (!ctx_ready()) return 1;
        if (flx_comm_sum_u64(ctx, n_all, 2) != FLX_OK) return fail_flx(ctx, "exchange");
        if (n_all[1]) { std::cerr << "Error reading " << args.input_reads << "\n"; return 1; }
        streamed_records_counted = n_all[0];
        share_lo = n_all[0] / (uint64_t)world * (uint64_t)rank + std::min<uint64_t>((uint64_t)rank, n_all[0] % (uint64_t)world);
        share_n = n_all[0] / (uint64_t)world + ((uint64_t)rank < n_all[0] % (uint64_t)world ? 1 : 0);
        stage("count pass (rank 0)");
    }
    auto in_share = [&](uint64_t rec) { return rec >= share_lo && rec - share_lo < share_n; };

    flx_params prm;
    memset(&prm, 0, sizeof prm);
    prm.window_size = args.window_size;  // (already narrowed to the reference's int by parse_args)
    prm.min_length_set = args.min_length_set; prm.min_length = args.min_length;
    prm.max_length_set = args.max_length_set; prm.max_length = args.max_length;
    prm.min_mean_q_set = args.min_mean_q_set; prm.min_mean_q = args.min_mean_q;
    prm.min_window_q_set = args.min_window_q_set; prm.min_window_q = args.min_window_q;
    prm.trim = args.trim; prm.split_set = args.split_set; prm.split = args.split;
    // two pinned staging slots of this size: pinning costs ~0.2 s per GiB and again when unpinned, so the slots are kept at
    // 256 MiB (profiles/r03_e2e.txt: 1 GiB slots cost a 2 GB input 0.45 s of 1.3 s)
    uint64_t chunk_bytes = streamed ? std::max<uint64_t>(4096, BlockReader::block_bytes()) : 256ull << 20, chunk_reads = 4u << 20;
    if (const char *e = getenv("FLX_CLI_CHUNK_MB")) chunk_bytes = std::max<uint64_t>(1, (uint64_t)atoll(e)) << 20;
    if (const char *e = getenv("FLX_CLI_CHUNK_BYTES")) chunk_bytes = std::max<uint64_t>(4096, (uint64_t)atoll(e));  // tests force many chunks

    // what survives pass 1 for this rank's records: lengths and names (views into the mapped input, or copies when streaming)
    Parsed kept;                          // the single batch of a mapped / in-memory input
    std::vector<int32_t> lengths;
    std::vector<std::string_view> names;
    std::deque<std::string> name_arena;
    uint64_t lo_rec = 0;                  // first record of this rank's contiguous block of file order
    long long total_bases = 0, last_progress = 0;
    bool any_fasta = false, any_fastq = false;
    std::unordered_set<std::string_view> seen_names;
    uint64_t n_records = 0, n_chunks = 0, n_batches = 0;
    UnitIndex units;                      // streamed input: the record-aligned pieces the output pass inflates concurrently
    flx_pipeline *pipe = nullptr;
    struct PipeGuard {  // an error return must not leave the worker thread running into the runtime's teardown
        flx_pipeline *&p;
        ~PipeGuard() { if (p) { flx_pipeline_destroy(p); p = nullptr; } }
    } pipe_guard{pipe};
    std::vector<uint64_t> offsets;

    // Pack records [lo, lo + cnt) of a batch chunk by chunk into the pipeline's pinned staging buffers (two slots: the GPU
    // copies and scores chunk k while the host threads pack chunk k+1); only per-read scalars survive a chunk.
    auto score_records = [&](const std::vector<Record> &recs, uint64_t lo, uint64_t cnt) -> int {
        if (!ctx_ready()) return 1;
        const uint64_t base = lengths.size();
        int32_t longest = 0;
        for (uint64_t i = 0; i < cnt; ++i) {
            if (recs[lo + i].seq.size() > (size_t)INT32_MAX) {  // the reference holds a read's length in an int as well (src/main.cpp:108)
                std::cerr << "\nError: read " << recs[lo + i].name.sv() << " is longer than 2^31-1 bases\n";
                return 1;
            }
            lengths.push_back((int32_t)recs[lo + i].seq.size());
            longest = std::max(longest, lengths.back());
        }
        const uint64_t need = (((uint64_t)longest + 15) & ~15ull) + 256;  // a read is never split over chunks
        if (!pipe) {
            if (!streamed) {  // everything is known: no larger slots than this rank's reads need
                uint64_t total = 4096;
                for (uint64_t i = 0; i < cnt; ++i) total += (((uint64_t)lengths[base + i] + 15) & ~15ull) + (lengths[base + i] >= 1024 ? 128 : 0);
                chunk_bytes = std::min(chunk_bytes, total);
                // small inputs: at least ~8 chunks, so that copy and scoring overlap the packing (but not below 64 MiB)
                if (!getenv("FLX_CLI_CHUNK_MB") && !getenv("FLX_CLI_CHUNK_BYTES"))
                    chunk_bytes = std::min(chunk_bytes, std::max<uint64_t>(total / 8, 64ull << 20));
                chunk_reads = std::min<uint64_t>(chunk_reads, std::max<uint64_t>(1, cnt));
            }
            chunk_bytes = std::max(chunk_bytes, need);
            if (flx_pipeline_create(ctx, kmers_empty ? nullptr : kmers, &prm, chunk_bytes, chunk_reads, &pipe) != FLX_OK) return fail_flx(ctx, "pipeline");
        } else if (need > chunk_bytes) {
            chunk_bytes = need;
            if (flx_pipeline_reserve(pipe, chunk_bytes, chunk_reads) != FLX_OK) return fail_flx(ctx, "pipeline");
        }
        const int32_t *len = lengths.data() + base;
        for (uint64_t at = 0; at < cnt;) {
            // the next chunk: as many records as fit the slot (flx_plane_layout's rule: 16-byte slots, 128-byte starts for long reads)
            uint64_t end = at, bytes = 0;
            while (end < cnt && end - at < chunk_reads) {
                uint64_t off = bytes;
                if (len[end] >= 1024) off = (off + 127u) & ~(uint64_t)127u;
                const uint64_t nb = off + (((uint64_t)len[end] + 15u) & ~(uint64_t)15u);
                if (nb > chunk_bytes && end > at) break;
                bytes = nb;
                ++end;
            }
            const uint64_t m = end - at;
            offsets.assign(m, 0);
            uint64_t plane_bytes = 0;
            flx_plane_layout(len + at, m, offsets.data(), &plane_bytes);
            uint8_t *plane = nullptr;
            if (flx_pipeline_next_buffer(pipe, &plane, nullptr, nullptr) != FLX_OK) return fail_flx(ctx, "scoring");
            const size_t parts = std::min<uint64_t>(m, (uint64_t)host_threads() * 8);
            parallel_for(parts, [&](size_t k) {  // byte-balanced slices of the chunk's reads
                const uint64_t lo_b = plane_bytes / parts * k, hi_b = k + 1 == parts ? plane_bytes : plane_bytes / parts * (k + 1);
                const uint64_t first = std::lower_bound(offsets.begin(), offsets.end(), lo_b) - offsets.begin();
                const uint64_t last = k + 1 == parts ? m : std::lower_bound(offsets.begin(), offsets.end(), hi_b) - offsets.begin();
                for (uint64_t i = first; i < last; ++i) {
                    const Record &r = recs[lo + at + i];
                    const View &src = kmers_empty ? r.qual : r.seq;  // Phred mode reads qual, k-mer mode reads seq
                    if (!src.empty()) memcpy(plane + offsets[i], src.p, src.size());
                    const uint64_t tail = offsets[i] + src.size();  // the staging buffer is reused: clear the padding behind the read
                    const uint64_t next = i + 1 < m ? offsets[i + 1] : plane_bytes;
                    if (next > tail) memset(plane + tail, 0, next - tail);
                }
            });
            if (flx_pipeline_submit(pipe, plane_bytes, offsets.data(), len + at, m) != FLX_OK) return fail_flx(ctx, "scoring");
            at = end;
            ++n_chunks;
        }
        return 0;
    };

    // Read::print_verbose_read_info for reads [0, n_first) (src/read.cpp:169-194), in file order like the pass-1 loop (main.cpp:110-111)
    auto print_read_blocks = [&](std::ostream &os, const flx_scores &res, uint64_t n_first) {
        const double *mean_q = res.mean_q, *window_q = res.window_q, *c_mean = res.child_mean_q, *c_window = res.child_window_q;
        const int32_t *c_ranges = res.child_ranges;
        const uint64_t *child_off = res.child_offsets;
        const uint64_t n = n_first;
        for (uint64_t i = 0; i < n; ++i) {
            const std::string_view rname = names[i];
            os << "\n" << rname << "\n";
            os << "            length = " << pad(std::to_string(lengths[i]), 11) << "mean quality = " << double_to_string(mean_q[i])
                      << "      window quality = " << double_to_string(window_q[i]) << "\n";
            const uint64_t a = child_off[i], b = child_off[i + 1];
            // m_bad_ranges (read.cpp:86-117): disjoint, non-adjacent and sorted, so with children they are exactly the gaps the
            // children leave in [0, L); without children the only possibility is the whole read (no covered base at all and
            // at least --split long)
            std::vector<std::pair<int, int>> bad;
            if (a != b) {
                int from = 0;
                for (uint64_t k = a; k < b; ++k) {
                    if (c_ranges[2 * k] > from) bad.push_back({from, c_ranges[2 * k]});
                    from = c_ranges[2 * k + 1];
                }
                if (from < lengths[i]) bad.push_back({from, lengths[i]});
            } else if (!kmers_empty && args.split_set && res.first[i] == -1 && lengths[i] > 0 && lengths[i] >= args.split) {
                bad.push_back({0, lengths[i]});
            }
            if (!bad.empty()) {
                os << "        bad ranges = ";
                for (size_t k = 0; k < bad.size(); ++k) os << bad[k].first << "-" << bad[k].second << (k + 1 < bad.size() ? ", " : "");
                os << "\n";
            }
            if (a != b) {
                os << "      child ranges = ";
                for (uint64_t k = a; k < b; ++k) os << c_ranges[2 * k] << "-" << c_ranges[2 * k + 1] << (k + 1 < b ? ", " : "");
                os << "\n";
                for (uint64_t k = a; k < b; ++k) {
                    os << "\n" << rname << "_" << c_ranges[2 * k] + 1 << "-" << c_ranges[2 * k + 1] << "\n";
                    os << "            length = " << pad(std::to_string(c_ranges[2 * k + 1] - c_ranges[2 * k]), 11) << "mean quality = "
                              << double_to_string(c_mean[k]) << "      window quality = " << double_to_string(c_window[k]) << "\n";
                }
            }
        }
    };
    // --verbose on an ERROR path: the reference scores and prints every read inside its pass-1 loop, so the blocks of the reads in
    // front of the failing record (for a duplicate name: that record's too) are on stderr before the error line
    // (src/main.cpp:108-117).  Scoring is batched here: the reads read so far are scored now, then their blocks printed.
    auto verbose_before_error = [&](const std::vector<Record> &recs, uint64_t k, bool streamed_names) -> void {
        if (!args.verbose || g_rank > 0) return;  // (several ranks: rank 0 alone scores the reads in front of the error — no exchange is involved)
        if (!streamed_names) for (uint64_t i = 0; i < k; ++i) names.push_back(recs[i].name.sv());
        if (score_records(recs, 0, k) != 0) return;
        if (!pipe && flx_pipeline_create(ctx, kmers_empty ? nullptr : kmers, &prm, chunk_bytes, chunk_reads, &pipe) != FLX_OK) return;
        flx_scores res;
        uint64_t n_scored = 0;
        if (flx_pipeline_finish(pipe, &res, &n_scored) != FLX_OK || n_scored != lengths.size()) return;
        print_read_blocks(std::cerr, res, n_scored);
    };

    // A record that is a header and nothing else (no sequence, no '+' line; kseq returns it with length 0, src/kseq.h:206-213) prints
    // oddly in the reference's FASTQ output: src/main.cpp:279 sends the C string seq->qual.s, and kseq has only reset that buffer's
    // LENGTH — the quality string of the last record in front of it that had a '+' line comes out again.  With no such record the
    // pointer is null, std::cout goes bad and nothing at all is written from there on.  Both are reproduced: the string to print
    // is noted here, in file order; the stream's death is decided in the output pass (only a record that passes prints).
    std::unordered_map<uint64_t, std::string> stale_qual;  // header-only record -> the quality string the reference prints for it
    std::unordered_set<uint64_t> null_qual;                 // header-only records in front of the first '+' line
    bool have_plus = false;
    View last_plus_qual;            // quality of the last record with a '+' line in the current batch ...
    std::string last_plus_stash;    // ... or, from an earlier block of a streamed input, a copy of it
    bool last_plus_in_batch = false;

    // ---- several ranks, a mapped file: every rank indexes only ITS byte range (round-3 review, item 8a; the default since round 5,
    // after the ranks fuzz, the damaged-input fuzz and the CLI's multi-rank tests had been through it; FLX_CLI_RANK_RANGES=0 switches it off).  parse_rank_range gives the share; what the loop below checks record by record
    // over the whole file becomes three facts about the shares and two exchanges:
    //   * every share is accepted and made of ordinary records of ONE kind (FASTQ with as many qualities as bases, or FASTA in k-mer
    //     mode), none empty, none longer than an int: a sum of flags.  Anything else — an error to report in file order, records
    //     whose output depends on the ones in front of them (the header-only records below) — and EVERY rank parses the whole file
    //     as before: the odd cases keep the code that is checked against the reference, and they are cheap or fatal anyway;
    //   * no name occurs twice: the 64-bit hashes of all names, gathered (a sum into disjoint slots) and sorted on every rank; two
    //     equal hashes — a duplicate or a collision — send every rank to the whole file as well;
    //   * the progress lines of src/main.cpp:119-127 depend on every read's length in file order: gathered with the hashes, rank 0
    //     replays them.
    // 0: not taken (parse the whole file), 1: `mine` holds this rank's records and the totals are set, -1: the exchange failed.
    bool ranged = false;
    const char *rr_env = getenv("FLX_CLI_RANK_RANGES");  // "0": every rank parses the whole file (round 4's default; tests, A/B)
    const bool rank_ranges = world > 1 && !streamed && data.map != nullptr && !(rr_env && rr_env[0] == '0');
    auto index_rank_range = [&](Parsed &mine) -> int {
        bool ok = parse_rank_range(data, rank, world, mine);
        bool fa = false, fq = false;
        uint64_t bases = 0;
        if (ok)
            for (const Record &r : mine.recs) {
                const bool fasta_format = r.qual.empty() && !r.seq.empty() && !r.is_fastq;
                const bool fastq_format = r.is_fastq && !r.seq.empty() && r.qual.size() == r.seq.size();
                if ((!fasta_format && !fastq_format) || r.seq.size() > (size_t)INT32_MAX) { ok = false; break; }
                fa = fa || fasta_format;
                fq = fq || fastq_format;
                bases += r.seq.size();
            }
        const uint64_t n_mine = ok ? mine.recs.size() : 0;
        std::vector<uint64_t> v(3 + 2 * (size_t)world, 0);
        v[0] = ok; v[1] = ok && fa; v[2] = ok && fq;
        v[3 + (size_t)rank] = n_mine;
        v[3 + (size_t)world + (size_t)rank] = ok ? bases : 0;
        if (flx_comm_sum_u64(ctx, v.data(), v.size()) != FLX_OK) return -1;
        if (v[0] != (uint64_t)world || (v[1] && v[2]) || (v[1] && kmers_empty)) return 0;
        uint64_t n_all = 0, lo = 0, bases_all = 0;
        for (int r = 0; r < world; ++r) {
            if (r == rank) lo = n_all;
            n_all += v[3 + (size_t)r];
            bases_all += v[3 + (size_t)world + (size_t)r];
        }
        // names and lengths of every read, in file order: hashes in [0, n_all), lengths two to a word behind them
        std::vector<uint64_t> w(n_all + (n_all + 1) / 2, 0);
        parallel_for(std::min<size_t>(std::max<size_t>(1, n_mine), 64), [&](size_t k) {
            const size_t parts = std::min<size_t>(std::max<size_t>(1, n_mine), 64);
            for (size_t i = n_mine * k / parts; i < n_mine * (k + 1) / parts; ++i) {
                uint64_t h = 1469598103934665603ull;
                const View &nm = mine.recs[i].name;
                for (size_t q = 0; q < nm.n; ++q) { h ^= (unsigned char)nm.p[q]; h *= 1099511628211ull; }
                w[lo + i] = h ^ (h >> 29);
            }
        });
        for (uint64_t i = 0; i < n_mine; ++i)  // (serial: two neighbours share a word)
            w[n_all + ((lo + i) >> 1)] |= (uint64_t)(uint32_t)mine.recs[i].seq.size() << (32 * ((lo + i) & 1));
        if (flx_comm_sum_u64(ctx, w.data(), w.size()) != FLX_OK) return -1;
        {  // two equal hashes anywhere?  1024 buckets by the top bits, sorted concurrently
            const size_t NB = 1024;
            std::vector<size_t> at(NB + 1, 0);
            for (uint64_t i = 0; i < n_all; ++i) ++at[(w[i] >> 54) + 1];
            for (size_t b = 0; b < NB; ++b) at[b + 1] += at[b];
            std::vector<uint64_t> sorted(n_all);
            {
                std::vector<size_t> cur(at.begin(), at.end() - 1);
                for (uint64_t i = 0; i < n_all; ++i) sorted[cur[w[i] >> 54]++] = w[i];
            }
            std::vector<char> twice(NB, 0);
            parallel_for(NB, [&](size_t b) {
                std::sort(sorted.begin() + (ptrdiff_t)at[b], sorted.begin() + (ptrdiff_t)at[b + 1]);
                for (size_t i = at[b] + 1; i < at[b + 1]; ++i)
                    if (sorted[i] == sorted[i - 1]) { twice[b] = 1; break; }
            });
            for (char t : twice)
                if (t) return 0;
        }
        n_records = n_all;
        total_bases = (long long)bases_all;
        any_fasta = v[1] != 0;
        any_fastq = v[2] != 0;
        if (rank == 0 && !args.verbose) {  // the progress lines, as the loop below prints them read by read
            long long tb = 0, lp = 0;
            for (uint64_t i = 0; i < n_all; ++i) {
                tb += (long long)((w[n_all + (i >> 1)] >> (32 * (i & 1))) & 0xffffffffull);
                if (tb - lp >= 483611) {
                    lp = tb;
                    std::cerr << "\r  " << int_to_string((long long)(i + 1)) << " reads (" << int_to_string(tb) << " bp)";
                }
            }
        }
        return 1;
    };

    // An error of the INPUT is found by every rank at the same record (all of them index the whole file): rank 0 reports it and
    // ends the job; the others leave quietly with status 0, so that the watchdog does not take their exit for a rank that died
    // while rank 0 is still scoring and printing the --verbose blocks in front of the error.
    const int input_error_rc = g_rank > 0 ? 0 : 1;
    for (;;) {
        Parsed batch_store;
        Parsed &batch = streamed ? batch_store : kept;
        if (streamed) {
            if (!blocks.next(batch)) {
                if (blocks.io_error) { std::cerr << "Error reading " << args.input_reads << "\n"; return input_error_rc; }
                break;
            }
        } else {
            if (n_batches > 0) break;
            const int took = rank_ranges ? index_rank_range(batch) : 0;
            if (took < 0) return fail_flx(ctx, "exchange");
            ranged = took > 0;
            if (!ranged) {
                batch = Parsed();
                parse_all(data, batch);
            }
            stage("parse");
        }
        ++n_batches;
        const std::vector<Record> &recs = batch.recs;
        if (ranged) {  // every check of the loop below has been made for the whole file (index_rank_range): this rank's records are its share
            lo_rec = 0;
            names.reserve(recs.size());
            for (const Record &r : recs) names.push_back(r.name.sv());
            if (g_timing) fprintf(stderr, "[timing] rank ranges: %llu of %llu records indexed here\n", (unsigned long long)recs.size(), (unsigned long long)n_records);
            if (const int rc = score_records(recs, 0, recs.size())) return rc;
            continue;
        }
        // the per-record checks of src/main.cpp:84-117, in file order, then the parser's own end status
        // Duplicate names (src/main.cpp:113-117: the first record whose name an earlier record has).  Streamed input: a set, block
        // after block.  Mapped input: every thread owns the names whose hash falls into its share, walks the records in file order and
        // stops at the first name it has seen before; the smallest such record over all threads is the reference's.
        uint64_t dup_at = UINT64_MAX;
        if (!streamed && recs.size() > 1) {
            const size_t nrec = recs.size();
            std::vector<uint64_t> name_hash(nrec);
            const size_t hparts = std::min<size_t>(nrec, 64);
            parallel_for(hparts, [&](size_t k) {
                for (size_t i = nrec * k / hparts; i < nrec * (k + 1) / hparts; ++i) {
                    uint64_t h = 1469598103934665603ull;
                    const View &v = recs[i].name;
                    for (size_t q = 0; q < v.n; ++q) { h ^= (unsigned char)v.p[q]; h *= 1099511628211ull; }
                    name_hash[i] = h ^ (h >> 29);
                }
            });
            const size_t owners = std::max<size_t>(1, std::min<size_t>(host_threads(), nrec / 4096));
            std::vector<uint64_t> first_dup(owners, UINT64_MAX);
            parallel_for(owners, [&](size_t t) {
                std::unordered_set<std::string_view> mine;
                mine.reserve(nrec / owners * 2 + 16);
                for (size_t i = 0; i < nrec; ++i)
                    if (name_hash[i] % owners == t && !mine.insert(recs[i].name.sv()).second) { first_dup[t] = i; return; }
            });
            for (uint64_t d : first_dup) dup_at = std::min(dup_at, d);
        }
        for (const Record &r : recs) {
            total_bases += (long long)r.seq.size();
            const bool fasta_format = r.qual.empty() && !r.seq.empty();
            const bool fastq_format = !r.qual.empty() && !r.seq.empty() && r.qual.size() == r.seq.size();
            any_fasta = any_fasta || fasta_format;
            any_fastq = any_fastq || fastq_format;
            if (any_fasta && any_fastq) {
                verbose_before_error(recs, (uint64_t)(&r - recs.data()), streamed);
                std::cerr << "\n\n" << "Error: could not parse input reads" << "\n";
                std::cerr << "  problem occurred at read " << r.name << "\n";
                return input_error_rc;
            }
            if (fasta_format && kmers_empty) {
                verbose_before_error(recs, (uint64_t)(&r - recs.data()), streamed);
                std::cerr << "\n\n" << "Error: FASTA input not supported without an external reference" << "\n";
                return input_error_rc;
            }
            std::string_view name = r.name.sv();
            if (streamed) {  // the block's memory is reused: keep a copy
                name_arena.emplace_back(name);
                name = name_arena.back();
            }
            if (streamed ? !seen_names.insert(name).second : (uint64_t)(&r - recs.data()) == dup_at) {
                if (streamed && in_share(n_records)) names.push_back(name);  // the duplicate itself is scored and printed before the check (main.cpp:108-113)
                verbose_before_error(recs, (uint64_t)(&r - recs.data()) + 1, streamed);
                std::cerr << "Error: duplicate read name: " << r.name << "\n";
                return input_error_rc;
            }
            if (streamed) {
                if (in_share(n_records)) names.push_back(name);
                units.note_record(blocks.points, blocks.offset_of(r.name.p - 1), n_records);
            }
            if (r.is_fastq) {
                have_plus = true;
                last_plus_qual = r.qual;
                last_plus_in_batch = true;
            } else if (r.seq.empty()) {
                if (!have_plus) null_qual.insert(n_records);
                else stale_qual.emplace(n_records, last_plus_in_batch ? std::string(last_plus_qual.p, last_plus_qual.n) : last_plus_stash);
            }
            ++n_records;
            if (total_bases - last_progress >= 483611) {
                last_progress = total_bases;
                if (!args.verbose) std::cerr << "\r  " << int_to_string((long long)n_records) << " reads (" << int_to_string(total_bases) << " bp)";
            }
        }
        if (streamed && last_plus_in_batch) {  // the block's memory goes away
            last_plus_stash.assign(last_plus_qual.p, last_plus_qual.n);
            last_plus_in_batch = false;
        }
        if (batch.status == -2) {
            verbose_before_error(recs, recs.size(), streamed);
            std::cerr << "Error: incorrect FASTQ format for read " << batch.bad.name << "\n";
            return input_error_rc;
        }
        if (batch.status == -3) {  // a damaged gzip stream: kseq's error state behind the bytes gzread delivered (src/main.cpp:85-88)
            verbose_before_error(recs, recs.size(), streamed);
            std::cerr << "Error reading " << args.input_reads << "\n";
            return input_error_rc;
        }
        if (!streamed) stage("record checks");
        // this rank's share of the batch: everything when streaming (one rank), else a contiguous block of file order by count
        uint64_t lo = 0, cnt = recs.size();
        if (streamed && world > 1) {  // the part of this block that lies in the rank's share (n_records has moved behind the block)
            const uint64_t first = n_records - recs.size(), last = n_records;
            const uint64_t a0 = std::max(first, share_lo), a1 = std::min(last, share_lo + share_n);
            lo = a0 < a1 ? a0 - first : 0;
            cnt = a0 < a1 ? a1 - a0 : 0;
            lo_rec = share_lo;
        }
        if (!streamed) {
            const uint64_t n_all = recs.size();
            lo = n_all / (uint64_t)world * (uint64_t)rank + std::min<uint64_t>((uint64_t)rank, n_all % (uint64_t)world);
            cnt = n_all / (uint64_t)world + ((uint64_t)rank < n_all % (uint64_t)world ? 1 : 0);
            lo_rec = lo;
            names.reserve(cnt);
            for (uint64_t i = 0; i < cnt; ++i) names.push_back(recs[lo + i].name.sv());
        }
        if (const int rc = score_records(recs, lo, cnt)) return rc;
    }
    { std::unordered_set<std::string_view>().swap(seen_names); }
    if (streamed) units.finish(blocks.points, blocks.end_offset(), n_records);
    if (streamed_records_counted != UINT64_MAX && n_records != streamed_records_counted) {
        // the shares were cut from rank 0's count: a file that changed between the two passes would give shares that do not match the records seen
        std::cerr << "Error: " << args.input_reads << " changed while it was read (" << streamed_records_counted << " records counted, " << n_records << " read)\n";
        return 1;
    }
    if (!args.verbose) std::cerr << "\r  " << int_to_string((long long)n_records) << " reads (" << int_to_string(total_bases) << " bp)";
    if (!args.verbose) std::cerr << "\n";  // verbose: after the per-read blocks, as in main.cpp:110-129
    const bool fasta_output = any_fasta, fastq_output = any_fastq;
    const uint64_t n = lengths.size();
    if (!ctx_ready()) return 1;
    if (!pipe && flx_pipeline_create(ctx, kmers_empty ? nullptr : kmers, &prm, chunk_bytes, chunk_reads, &pipe) != FLX_OK) return fail_flx(ctx, "pipeline");
    flx_scores res;
    uint64_t n_scored = 0;
    if (flx_pipeline_finish(pipe, &res, &n_scored) != FLX_OK || n_scored != n) return fail_flx(ctx, "scoring");
    const int32_t *c_ranges = res.child_ranges;
    if (g_timing) fprintf(stderr, "[timing] %llu chunk(s) of <= %llu MiB\n", (unsigned long long)n_chunks, (unsigned long long)(chunk_bytes >> 20));

    stage("pack + H2D + score (streamed)");
    // ---- reads2: children replace their parents in place (src/main.cpp:138-147) -----------------------------
    struct Out { uint64_t rec; int start, end; bool child; std::string name; };
    std::vector<Out> reads2;
    std::vector<double> r2_mean, r2_window;
    std::vector<int32_t> r2_len;
    std::vector<uint8_t> r2_pass;
    {
        // the gather itself is the library's (flx_reads2_gather): values in reads2 order + where every entry came from
        const uint64_t cap2 = n + res.n_children;
        r2_mean.resize(cap2); r2_window.resize(cap2); r2_len.resize(cap2); r2_pass.resize(cap2);
        std::vector<uint32_t> parent2(cap2);
        std::vector<int64_t> child2(cap2);
        uint64_t n2_gathered = 0;
        if (flx_reads2_gather(ctx, n, lengths.data(), &res, cap2, r2_mean.data(), r2_window.data(), r2_len.data(), r2_pass.data(),
                              parent2.data(), child2.data(), &n2_gathered) != FLX_OK)
            return fail_flx(ctx, "reads2 gather");
        r2_mean.resize(n2_gathered); r2_window.resize(n2_gathered); r2_len.resize(n2_gathered); r2_pass.resize(n2_gathered);
        reads2.reserve(n2_gathered);
        for (uint64_t j = 0; j < n2_gathered; ++j) {
            const uint64_t i = parent2[j];
            if (child2[j] < 0) {
                reads2.push_back({lo_rec + i, 0, lengths[i], false, std::string(names[i])});
            } else {
                const int s0 = c_ranges[2 * child2[j]], e0 = c_ranges[2 * child2[j] + 1];
                reads2.push_back({lo_rec + i, s0, e0, true, std::string(names[i]) + "_" + std::to_string(s0 + 1) + "-" + std::to_string(e0)});  // read.cpp:135-136
            }
        }
    }
    size_t longest_name = 0;
    for (auto &o : reads2) longest_name = std::max(longest_name, o.name.size());

    // Read::print_verbose_read_info, src/read.cpp:169-194, in file order like the pass-1 loop (main.cpp:110-111).  Several ranks:
    // rank r > 0 leaves the blocks of its reads in a file of the job's private directory; rank 0 prints its own and, behind the
    // exchange of the totals below (every rank has written its file when that returns), the others' in rank = file order.
    auto verbose_part = [&](const char *kind, int r) { return g_part_prefix + "." + kind + std::to_string(r); };
    auto print_verbose_parts = [&](const char *kind) -> bool {
        std::vector<char> vbuf(1 << 20);
        for (int r = 1; r < world; ++r) {
            const std::string pth = verbose_part(kind, r);
            FILE *f = fopen(pth.c_str(), "rb");
            if (!f) { std::cerr << "Error: cannot read " << pth << "\n"; return false; }
            size_t got;
            while ((got = fread(vbuf.data(), 1, vbuf.size(), f)) > 0) std::cerr.write(vbuf.data(), (std::streamsize)got);
            fclose(f);
            unlink(pth.c_str());
        }
        return true;
    };
    if (args.verbose) {
        if (rank == 0) {
            print_read_blocks(std::cerr, res, n);
        } else {
            std::ofstream f(verbose_part("vblocks", rank), std::ios::binary);
            print_read_blocks(f, res, n);
            f.close();
            if (!f) return fail_flx(ctx, "verbose part");
        }
        if (world == 1) std::cerr << "\n";  // the line main.cpp:129 prints after the loop
    }

    // totals over all ranks (one rank: the local values)
    uint64_t n2_local = reads2.size();
    std::vector<uint64_t> n2_of((size_t)world, 0);
    long long after_local = 0;
    for (auto v : r2_len) after_local += v;
    uint64_t n2_total = n2_local;
    long long after_total = after_local;
    if (world > 1) {
        std::vector<uint64_t> sums(2 * (size_t)world + 1, 0);
        sums[rank] = n2_local;
        sums[world] = (uint64_t)after_local;
        sums[(size_t)world + 1 + rank] = longest_name;  // (the --verbose table pads every name to the longest of ALL reads2, main.cpp:199-201)
        if (flx_comm_sum_u64(ctx, sums.data(), sums.size()) != FLX_OK) return fail_flx(ctx, "exchange");
        n2_total = 0;
        for (int r = 0; r < world; ++r) { n2_of[r] = sums[r]; n2_total += sums[r]; longest_name = std::max<size_t>(longest_name, sums[(size_t)world + 1 + r]); }
        after_total = (long long)sums[world];
        if (args.verbose && rank == 0) {
            if (!print_verbose_parts("vblocks")) return 1;
            std::cerr << "\n";
        }
    }
    if ((args.trim || args.split_set) && rank == 0) {  // src/main.cpp:155-166
        if (args.trim && args.split_set) std::cerr << "  after trimming and splitting: ";
        else if (args.trim) std::cerr << "  after trimming: ";
        else std::cerr << "  after splitting: ";
        std::cerr << int_to_string((long long)n2_total) << " reads (" << int_to_string(after_total) << " bp)\n";
    }
    if (rank == 0) std::cerr << "\n";

    // ---- global stage (src/main.cpp:169-261) ---------------------------------------------------------------
    const uint64_t n2 = n2_local;
    flx_cut_report rep;
    memset(&rep, 0, sizeof rep);
    const bool cutting = args.target_bases_set || args.keep_percent_set;
    if (n2_total > 0 || cutting) {
        const int rc = world > 1
            ? flx_rank_and_cut_comm(ctx, n2, r2_mean.data(), r2_window.data(), r2_len.data(), r2_pass.data(), args.length_weight,
                                    args.mean_q_weight, args.window_q_weight, args.target_bases_set, args.target_bases,
                                    args.keep_percent_set, args.keep_percent, total_bases, nullptr, &rep)
            : flx_rank_and_cut(ctx, n2, r2_mean.data(), r2_window.data(), r2_len.data(), r2_pass.data(), args.length_weight,
                               args.mean_q_weight, args.window_q_weight, args.target_bases_set, args.target_bases,
                               args.keep_percent_set, args.keep_percent, total_bases, nullptr, &rep);
        if (rc != FLX_OK) return fail_flx(ctx, "rank and cut");
    }
    if (args.verbose) {  // src/main.cpp:199-214: the table shows the NORMALISED qualities and the final score, host libm like the reference
        std::ofstream table_file;
        if (rank > 0) table_file.open(verbose_part("vtable", rank), std::ios::binary);
        std::ostream &tos = rank > 0 ? (std::ostream &)table_file : (std::ostream &)std::cerr;
        if (rank == 0)
            std::cerr << "\n\n" << "Read name" << "\t" << "Length score" << "\t" << "Mean quality score" << "\t" << "Window quality score"
                      << "\t" << "Final score" << "\n";
        const double zspan = rep.max_z - rep.min_z;
        double (*volatile powfn)(double, double) = pow;
        for (uint64_t i = 0; i < n2; ++i) {
            double ratio = r2_window[i] / r2_mean[i];  // main.cpp:203-208
            if (ratio > 1.0) ratio = 1.0;
            const double z = (r2_mean[i] - rep.mean_quality) / rep.stdev_quality;
            const double mq = 100.0 * (z - rep.min_z) / zspan;
            const double wq = mq * ratio;
            const double lscore = 100.0 * (1.0 + (-5000.0 / (r2_len[i] + 5000.0)));
            // Read::set_final_score, read.cpp:249-267
            const double product = powfn(lscore, args.length_weight) * powfn(mq, args.mean_q_weight);
            const double gm = powfn(product, 1.0 / (args.length_weight + args.mean_q_weight));
            double scale = 1.0;
            if (mq > 0.0) scale = std::min(wq / mq, 1.0);
            const double wfrac = args.window_q_weight / (args.length_weight + args.mean_q_weight + args.window_q_weight);
            const double fs = gm * ((1.0 - wfrac) + (scale * wfrac));
            tos << pad(reads2[i].name, longest_name) << "\t" << double_to_string(lscore) << "\t" << double_to_string(mq) << "\t"
                << double_to_string(wq) << "\t" << double_to_string(fs) << "\n";
        }
        if (world > 1) {  // every rank's rows are in its file when this exchange returns; rank 0 prints them in rank = file order
            if (rank > 0) { table_file.close(); if (!table_file) return fail_flx(ctx, "verbose part"); }
            uint64_t one = 1;
            if (flx_comm_sum_u64(ctx, &one, 1) != FLX_OK) return fail_flx(ctx, "exchange");
            if (rank == 0 && !print_verbose_parts("vtable")) return 1;
        }
        if (rank == 0) std::cerr << "\n";
    }
    if (cutting && rank == 0) {
        std::cerr << "Filtering long reads\n";
        std::cerr << "  target: " << int_to_string(rep.target_bases) << " bp\n";
        if (rep.outcome == FLX_CUT_NOT_ENOUGH) std::cerr << "  not enough reads to reach target\n";
        else if (rep.outcome == FLX_CUT_ALREADY_BELOW) std::cerr << "  reads already fall below target after filtering\n";
        else std::cerr << "  keeping " << int_to_string(rep.kept_bases) << " bp\n";
        std::cerr << "\n";
    }

    stage("rank and cut");
    // ---- output in input order (src/main.cpp:263-313) -----------------------------------------------------
    // One rank: straight to stdout.  Several ranks: every rank writes the passed records of its own block to a part file,
    // rank 0 streams the parts to stdout in rank (= file) order.
    if (rank == 0) std::cerr << "Outputting passed long reads\n";
    // (forked ranks whose common stdout is a regular file write their records straight into it, each at its own offset: below)
    FILE *sink = stdout;
    std::string part_path;
    bool shared_file = false, shared_skip = false;  // several ranks, one output file / an earlier rank's output "died": nothing of this rank's follows
    off_t shared_base = -1, shared_end = -1;
    auto open_part = [&]() -> bool {
        part_path = g_part_prefix + ".part" + std::to_string(rank);
        sink = fopen(part_path.c_str(), "wb");
        if (!sink) { std::cerr << "Error: cannot write " << part_path << "\n"; return false; }
        return true;
    };
    // the first passing header-only record without a quality string to repeat: the reference's std::cout dies behind its "+" line
    uint64_t dies_at = UINT64_MAX;
    if (fastq_output && !null_qual.empty())
        for (uint64_t i = 0; i < n2 && dies_at == UINT64_MAX; ++i)
            if (r2_pass[i] && !reads2[i].child && null_qual.count(reads2[i].rec)) dies_at = i;
    auto repeated_qual = [&](uint64_t i) -> const std::string * {
        if (stale_qual.empty() || reads2[i].child) return nullptr;
        const auto it = stale_qual.find(reads2[i].rec);
        return it == stale_qual.end() ? nullptr : &it->second;
    };
    auto emit = [&](std::string &out, uint64_t i, const Record &r) {  // output read i of reads2, cut out of its record
        if (!r2_pass[i] || i > dies_at) return;
        const Out &o = reads2[i];
        if (o.child && o.end - o.start <= 0) return;
        out += fasta_output ? '>' : '@';
        out += o.name;
        if (!r.comment.empty()) { out += ' '; out.append(r.comment.p, r.comment.n); }
        out += '\n';
        out.append(r.seq.p + o.start, (size_t)(o.end - o.start));
        out += '\n';
        if (fastq_output) {
            out += "+\n";
            if (i == dies_at) return;
            if (const std::string *q = repeated_qual(i)) out += *q;
            else out.append(r.qual.p + o.start, (size_t)(o.end - o.start));
            out += '\n';
        }
    };
    bool pieces_ok = true;
    if (!streamed) {
        // The passed records are cut out of the mapped input by several threads, ~16 MiB of output per piece.  Every piece's
        // place in the output is known beforehand, so when the sink is a regular file each thread writes its pieces itself
        // (pwrite at the piece's offset); a pipe or terminal gets the pieces in order from this thread.
        auto out_bytes = [&](uint64_t i) -> uint64_t {
            if (!r2_pass[i]) return 0;
            const Out &o = reads2[i];
            if (o.child && o.end - o.start <= 0) return 0;
            const Record &r = kept.recs[o.rec];
            const uint64_t L = (uint64_t)(o.end - o.start);
            if (i > dies_at) return 0;
            uint64_t b = 1 + o.name.size() + (r.comment.empty() ? 0 : 1 + r.comment.n) + 1 + L + 1;
            if (fastq_output) {
                b += 2;
                if (i == dies_at) return b;
                const std::string *q = repeated_qual(i);
                b += (q ? q->size() : L) + 1;
            }
            return b;
        };
        std::vector<uint64_t> piece_first{0}, piece_at{0};  // reads2 range and byte offset of every piece
        {
            uint64_t bytes = 0, in_piece = 0;
            for (uint64_t i = 0; i < n2; ++i) {
                const uint64_t b = out_bytes(i);
                bytes += b;
                in_piece += b;
                if (in_piece >= (16u << 20) && i + 1 < n2) { piece_first.push_back(i + 1); piece_at.push_back(bytes); in_piece = 0; }
            }
            piece_first.push_back(n2);
            piece_at.push_back(bytes);
        }
        const size_t n_pieces = piece_first.size() - 1;
        // A passed read whose record in the input already HAS the bytes of its output record — one header line "@name" or
        // "@name comment" with a single blank, one sequence line, a bare "+" line, one quality line, LF ends — is not formatted at
        // all: its bytes go from the mapping to the file in one pwritev, neighbours in the input merged into one range.  Anything
        // else (children, CRLF, wrapped lines, "+name", tabs) is formatted into a side buffer as before; the byte count per read is
        // the same either way, so the pieces keep their precomputed offsets.
        const char *map_lo = data.data(), *map_hi = data.data() + data.size();
        auto verbatim = [&](uint64_t i, const char *&from, size_t &len) -> bool {
            const Out &o = reads2[i];
            if (o.child || i >= dies_at) return false;
            const Record &r = kept.recs[o.rec];
            const char *h = r.name.p - 1;
            if (h < map_lo || r.name.p + r.name.n >= map_hi || *h != (fasta_output ? '>' : '@')) return false;
            const char *nl = r.name.p + r.name.n;  // the byte behind the name
            if (!r.comment.empty()) {
                if (*nl != ' ' || r.comment.p != nl + 1 || r.comment.p + r.comment.n >= map_hi) return false;
                nl = r.comment.p + r.comment.n;
            }
            if (*nl != '\n' || r.seq.p != nl + 1 || r.seq.p + r.seq.n >= map_hi || r.seq.p[r.seq.n] != '\n') return false;
            const char *end = r.seq.p + r.seq.n + 1;
            if (fastq_output) {
                if (end + 2 > map_hi || end[0] != '+' || end[1] != '\n' || r.qual.p != end + 2 || r.qual.n != r.seq.n ||
                    r.qual.p + r.qual.n >= map_hi || r.qual.p[r.qual.n] != '\n') return false;
                end = r.qual.p + r.qual.n + 1;
            }
            from = h;
            len = (size_t)(end - h);
            return true;
        };
        if (world > 1) {
            // Round-3 review, item 8: ONE output file.  The ranks forked by --gpus N share the job's stdout; when that is a regular file
            // (not in append mode) every rank writes its passed records at its own offset — the sum of the bytes of the ranks in front
            // of it, one exchange — with the same pwrite / pwritev pieces a single rank uses, and nothing is written twice.  A pipe, a
            // terminal, or ranks under a launcher (no common stdout): part files that rank 0 streams out in order, as before.
            std::vector<uint64_t> v(2 + 2 * (size_t)world, 0);
            if (rank == 0 && g_shared_out >= 0 && !getenv("FLX_CLI_ORDERED_OUTPUT")) {
                fflush(stdout);
                struct stat st;
                const int fl = fcntl(g_shared_out, F_GETFL);
                const off_t at = lseek(g_shared_out, 0, SEEK_CUR);
                if (fstat(g_shared_out, &st) == 0 && S_ISREG(st.st_mode) && fl >= 0 && !(fl & O_APPEND) && at >= 0) { v[0] = 1; v[1] = (uint64_t)at; }
            }
            v[2 + (size_t)rank] = piece_at.back();
            v[2 + (size_t)world + (size_t)rank] = dies_at != UINT64_MAX;
            if (flx_comm_sum_u64(ctx, v.data(), v.size()) != FLX_OK) return fail_flx(ctx, "exchange");
            if (v[0] && g_shared_out >= 0) {
                shared_file = true;
                uint64_t before = 0, total = 0;
                bool dead = false;
                for (int r = 0; r < world; ++r) {
                    if (r == rank) { before = total; shared_skip = dead; }
                    if (!dead) total += v[2 + (size_t)r];
                    dead = dead || v[2 + (size_t)world + (size_t)r] != 0;
                }
                shared_base = (off_t)(v[1] + before);
                shared_end = (off_t)(v[1] + total);
                sink = fdopen(dup(g_shared_out), "wb");
                if (!sink) { std::cerr << "Error: cannot write the output\n"; return 1; }
            } else if (!open_part()) {
                return 1;
            }
        }
        const int out_fd = fileno(sink);
        const char *fail_env = getenv("FLX_CLI_FAIL_WRITE_RANK");  // tests: this rank's writes fail (a full disk under one rank's share of the file)
        const bool fail_writes = fail_env && atoi(fail_env) == rank;
        const bool ok = shared_skip || write_pieces(n_pieces, [&](size_t j, std::string &buf) {
            if (fail_writes) return false;
            if (!g_direct_pieces) {  // a pipe / terminal / append-mode file: the caller writes the formatted piece in order
                for (uint64_t i = piece_first[j]; i < piece_first[j + 1]; ++i) emit(buf, i, kept.recs[reads2[i].rec]);
                return true;
            }
            // regular file: this thread writes the piece itself, ranges of the mapping and formatted records interleaved
            std::vector<struct iovec> iov;
            std::deque<std::string> side;
            bool last_is_map = false;
            off_t at = g_direct_base + (off_t)piece_at[j];
            auto flush = [&]() -> bool {
                size_t k = 0;
                while (k < iov.size()) {
                    const int cnt = (int)std::min<size_t>(iov.size() - k, 512);
                    ssize_t w = pwritev(out_fd, iov.data() + k, cnt, at);
                    if (w <= 0) return false;
                    at += w;
                    while (w > 0 && k < iov.size()) {  // a short write: drop what went out
                        if ((size_t)w >= iov[k].iov_len) { w -= (ssize_t)iov[k].iov_len; ++k; }
                        else { iov[k].iov_base = (char *)iov[k].iov_base + w; iov[k].iov_len -= (size_t)w; w = 0; }
                    }
                }
                iov.clear();
                side.clear();
                return true;
            };
            for (uint64_t i = piece_first[j]; i < piece_first[j + 1]; ++i) {
                if (!r2_pass[i]) continue;
                const char *from = nullptr;
                size_t len = 0;
                if (verbatim(i, from, len)) {
                    if (last_is_map && (const char *)iov.back().iov_base + iov.back().iov_len == from) iov.back().iov_len += len;  // neighbours in the input
                    else iov.push_back({(void *)from, len});
                    last_is_map = true;
                } else {
                    side.emplace_back();
                    emit(side.back(), i, kept.recs[reads2[i].rec]);
                    if (!side.back().empty()) {
                        iov.push_back({(void *)side.back().data(), side.back().size()});
                        last_is_map = false;
                    }
                }
                if (iov.size() >= 4096) {
                    if (!flush()) return false;
                    last_is_map = false;
                }
            }
            if (!flush()) return false;
            buf.clear();
            return at == g_direct_base + (off_t)piece_at[j + 1];
        }, sink, &piece_at, shared_file ? shared_base : (off_t)-1);
        // (several ranks: a rank that could not write — a full disk under its pwrite — still goes to the exchange below, where
        // every rank learns of it and rank 0 says why; leaving here would strand the others in that exchange)
        if (!ok && world == 1) { std::cerr << "Error: could not write the output\n"; return 1; }
        pieces_ok = ok;
    } else {
        // Second pass over the compressed input (src/main.cpp:263-313 re-reads the file too), but not front to back on one
        // thread: pass 1 left access points in the deflate stream, the pieces between them (whole records, ~32 MiB of text)
        // are inflated and parsed concurrently and written in order.  Pieces without a passing read are not inflated at all.
        if (world > 1 && !open_part()) return 1;  // (several ranks: every rank's records to its part file, rank 0 streams the parts out in order below)
        const size_t n_units = units.units();
        std::vector<uint64_t> r2_at(n_units + 1, n2);  // first reads2 entry of every unit (reads2 is in record order)
        {
            uint64_t cur = 0;
            for (size_t j = 0; j < n_units; ++j) {
                while (cur < n2 && reads2[cur].rec < units.first_rec[j]) ++cur;
                r2_at[j] = cur;
            }
        }
        const bool ok = write_pieces(n_units, [&](size_t j, std::string &buf) {
            bool any = false;
            for (uint64_t i = r2_at[j]; i < r2_at[j + 1] && !any; ++i) any = r2_pass[i] != 0;
            if (!any) return true;
            std::vector<char> text;
            Parsed got;
            if (!inflate_range(blocks.file, blocks.points[j], units.start[j], units.start[j + 1], text)) return false;
            Input view;
            view.p = text.data();
            view.n = text.size();
            parse_sequential(view, got);
            if (got.recs.size() != units.first_rec[j + 1] - units.first_rec[j]) return false;
            uint64_t cur = r2_at[j];
            for (size_t k = 0; k < got.recs.size(); ++k) {
                const uint64_t rec = units.first_rec[j] + k;
                const Record &r = got.recs[k];
                if (rec < lo_rec || rec - lo_rec >= n) continue;  // (several ranks: a unit at the edge of the share holds other ranks' records too)
                if (r.name.sv() != names[rec - lo_rec] || (int32_t)r.seq.size() != lengths[rec - lo_rec]) return false;
                for (; cur < r2_at[j + 1] && reads2[cur].rec == rec; ++cur) emit(buf, cur, r);
            }
            return true;
        }, sink, nullptr);
        if (!ok && world == 1) { std::cerr << "Error: " << args.input_reads << " could not be read a second time (did it change?)\n"; return 1; }
        pieces_ok = ok;  // (several ranks: to the exchange below, like a failed write)
    }
    // a sink that did not take everything (disk full, the reader of a pipe gone while SIGPIPE is ignored) ends the job with status 1
    const bool sink_ok = pieces_ok && fflush(sink) == 0 && !ferror(sink);
    if (world == 1 && !sink_ok) { std::cerr << "Error: could not write the output\n"; return 1; }
    if (world > 1) {
        fclose(sink);
        // every part is complete before rank 0 reads it; a rank whose output "died" (above) ends the whole output
        std::vector<uint64_t> done((size_t)world + 1, 0);
        done[0] = sink_ok;
        done[(size_t)rank + 1] = dies_at != UINT64_MAX;
        if (flx_comm_sum_u64(ctx, done.data(), done.size()) != FLX_OK) return fail_flx(ctx, "exchange");
        if (done[0] != (uint64_t)world) {  // some rank could not write its share: every rank leaves, rank 0 says why
            if (rank == 0) {
                for (int r = 0; r < world && !shared_file; ++r) unlink((g_part_prefix + ".part" + std::to_string(r)).c_str());
                std::cerr << "Error: could not write the output\n";
            }
            return rank == 0 ? 1 : 0;
        }
        if (shared_file) {  // everything is in the file already (every rank has written when the exchange returns): the position behind it
            if (rank == 0 && lseek(g_shared_out, shared_end, SEEK_SET) < 0) { std::cerr << "Error: could not write the output\n"; return 1; }
        } else if (rank == 0) {
            std::vector<char> buf(1 << 22);
            bool dead = false;
            for (int r = 0; r < world; ++r) {
                if (dead) { unlink((g_part_prefix + ".part" + std::to_string(r)).c_str()); continue; }
                dead = done[(size_t)r + 1] != 0;
                const std::string pth = g_part_prefix + ".part" + std::to_string(r);
                FILE *f = fopen(pth.c_str(), "rb");
                if (!f) { std::cerr << "Error: cannot read " << pth << "\n"; return 1; }
                size_t got;
                bool wrote = true;
                while (wrote && (got = fread(buf.data(), 1, buf.size(), f)) > 0) wrote = fwrite(buf.data(), 1, got, stdout) == got;
                fclose(f);
                unlink(pth.c_str());
                if (!wrote) {
                    for (int q = r + 1; q < world; ++q) unlink((g_part_prefix + ".part" + std::to_string(q)).c_str());
                    std::cerr << "Error: could not write the output\n";
                    return 1;
                }
            }
            if (fflush(stdout) != 0) { std::cerr << "Error: could not write the output\n"; return 1; }
        }
    }
    stage("output");

    // The output is complete.  Unpinning the staging buffers, shutting the HIP runtime down and unmapping the input is work the
    // kernel does faster when the process simply ends (0.5-0.9 s of 1.3-2.7 s on 2-10 GB inputs): flush and leave, unless a
    // clean teardown is asked for (FLX_CLI_CLEAN_EXIT=1, the timing report, or ranks to reap).
    const bool clean_exit = getenv("FLX_CLI_CLEAN_EXIT") != nullptr || g_timing;
    if (clean_exit) {
        flx_pipeline_destroy(pipe);
        pipe = nullptr;
        stage("pipeline teardown");
        if (kmers) flx_kmerset_destroy(kmers);
        flx_ctx_destroy(ctx);
        stage("context teardown");
    }
    if (!g_job.finish()) { std::cerr << "Error: a rank failed\n"; return 1; }
    if (g_timing) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "[timing] main() returns at wall clock %.3f\n", ts.tv_sec % 100000 + ts.tv_nsec * 1e-9);
    }
    if (rank == 0) std::cerr << "\n";
    const bool flushed = fflush(stdout) == 0 && !ferror(stdout);
    fflush(stderr);
    if (!clean_exit) _exit(flushed ? 0 : 1);
    return flushed ? 0 : 1;
}

