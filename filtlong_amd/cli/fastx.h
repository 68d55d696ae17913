// fastx.h — FASTA/FASTQ records over an addressable input: the kseq-compatible parser (reference src/kseq.h:176-224), the
// whole-file parse (sequential, and concurrent with verification) and the host-thread helper.  Included by main.cpp only.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "pinflate.h"

// ------------------------------------------------------------------------------------------------ FASTA/FASTQ
// In-memory parser with the record grammar of klib's kseq (src/kseq.h:176-224): records start at the next '>' or
// '@'; the name ends at the first whitespace, the rest of the header line is the comment; sequence lines run until a
// line whose first character is '>', '+' or '@'; after '+' the quality is read line by line until it is at least as
// long as the sequence; a trailing '\r' is dropped from every line; empty lines are skipped.
struct View {  // a piece of the input buffer (or of the side arena for multi-line records); never owns memory
    const char *p = nullptr;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    std::string str() const { return std::string(p ? p : "", n); }
    std::string_view sv() const { return std::string_view(p ? p : "", n); }
};
static std::ostream &operator<<(std::ostream &os, const View &v) { return os.write(v.p ? v.p : "", (std::streamsize)v.n); }

struct Record {
    View name, comment, seq, qual;
    bool is_fastq = false;
};

// ---- host threads for the two byte-moving stages (page-in of the input, packing the read plane) ------------------------
static unsigned host_threads() {
    const char *e = getenv("FLX_CLI_THREADS");
    if (e && atoi(e) > 0) return (unsigned)atoi(e);
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hw ? hw : 1u));
}

template <class F>
static void parallel_for(size_t n_parts, F &&body) {  // body(part) for part in [0, n_parts), on up to host_threads() threads
    const unsigned t = (unsigned)std::min<size_t>(host_threads(), n_parts);
    if (t <= 1) { for (size_t i = 0; i < n_parts; ++i) body(i); return; }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < t; ++k)
        th.emplace_back([&, k] { for (size_t i = k; i < n_parts; i += t) body(i); });
    for (auto &x : th) x.join();
}

// The whole input, addressable.  Plain files are mapped (no copy; pages are faulted in by several threads); gzip and
// pipes are inflated / read into memory through zlib, as the reference's kseq does (src/kseq.h:87-110).
struct Input {
    const char *p = nullptr;
    size_t n = 0;
    void *map = nullptr;
    size_t map_len = 0;
    std::string owned;
    // The bytes end where zlib's gzread reported a DATA ERROR to the reference's parser: kseq is then in its error state
    // (src/kseq.h:47,71-76,105-108: every further read returns -3) — not at the end of the file.  The Parser below turns that into
    // what kseq_read returns from there (a truncated stream is simply shorter: gzread delivers what it could decode, then EOF).
    bool stream_error = false;
    Input() = default;
    Input(const Input &) = delete;
    Input &operator=(const Input &) = delete;
    ~Input() { if (map) munmap(map, map_len); }
    const char *data() const { return p; }
    size_t size() const { return n; }

    bool open(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
        const bool gz = regular && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (regular && !gz) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                ::close(fd);
                map = m; map_len = (size_t)st.st_size;
                p = (const char *)m; n = map_len;
                madvise(m, map_len, MADV_WILLNEED);
                const size_t part = 64u << 20;
                const size_t parts = (n + part - 1) / part;
                std::vector<unsigned> sink(parts, 0);
                parallel_for(parts, [&](size_t i) {  // touch one byte per page so the page-table fill runs on all threads
                    unsigned acc = 0;
                    const size_t end = std::min(n, (i + 1) * part);
                    for (size_t at = i * part; at < end; at += 4096) acc += (unsigned char)p[at];
                    sink[i] = acc;
                });
                return true;
            }
        }
        if (gz) {  // a gzip file (reads taken into memory, references): block-parallel inflate of the mapped file
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                bool ok = false;
                {
                    ParallelInflate z;
                    if (z.open((const unsigned char *)m, (size_t)st.st_size, true, ParallelInflate::default_threads(host_threads()))) {
                        size_t have = 0;
                        owned.resize(std::max<size_t>((size_t)1 << 22, (size_t)st.st_size * 4));
                        while (!z.eof() && !z.error()) {
                            if (have == owned.size()) owned.resize(owned.size() * 2);
                            have += z.read(&owned[have], owned.size() - have);
                        }
                        ok = true;
                        if (z.error()) {  // a damaged file: the bytes gzread would have delivered, then kseq's error state
                            have = (size_t)std::min<uint64_t>(have, z.deliverable());
                            stream_error = true;
                        }
                        owned.resize(have);
                    }
                }
                munmap(m, (size_t)st.st_size);
                if (ok) {
                    ::close(fd);
                    p = owned.data(); n = owned.size();
                    return true;
                }
                owned.clear();
            }
        }
        ::close(fd);
        // pipes and whatever could not be mapped: through gzread itself, in kseq's own calls (16 KiB, zlib's default buffer —
        // src/kseq.h:234), so that a damaged stream ends for the parser exactly where it ends for the reference's
        gzFile fp = gzopen(path.c_str(), "r");  // transparent for uncompressed streams too
        if (!fp) return false;
        char buf[16384];
        for (;;) {
            const int got = gzread(fp, buf, (unsigned)sizeof buf);
            if (got < 0) { stream_error = true; break; }
            if (got == 0) break;
            owned.append(buf, (size_t)got);
        }
        gzclose(fp);
        p = owned.data(); n = owned.size();
        return true;
    }
};

// Record views point into the buffer; only multi-line sequences / qualities are copied (into `arena`).
struct Parser {
    const Input &d;
    std::deque<std::string> &arena;
    size_t pos = 0;
    int last_char = 0;
    Parser(const Input &data, std::deque<std::string> &side) : d(data), arena(side) {}
    int getc() { return pos < d.size() ? (unsigned char)d.p[pos++] : -1; }
    // rest of the current line as a view [from, end-of-line), consuming the newline; false at EOF
    bool rest_of_line(size_t from, View &v) {
        if (from > d.size()) return false;
        const void *nlp = pos < d.size() ? memchr(d.data() + pos, '\n', d.size() - pos) : nullptr;
        const size_t end = nlp ? (size_t)((const char *)nlp - d.data()) : d.size();
        v.p = d.data() + from;
        v.n = end - from;
        pos = nlp ? end + 1 : d.size();
        return true;
    }
    // kseq's ks_getuntil2(KS_SEP_LINE, append): append the rest of the line to the accumulating field `v`
    // (first line: a view; later lines: copied into the arena); a trailing '\r' is dropped when the field has > 1 chars
    void append_line(View &v, bool &owned, size_t first_from, bool first_consumed) {
        View line;
        if (v.n == 0 && !owned) {
            rest_of_line(first_from, v);
        } else {
            if (!owned) {
                arena.emplace_back(v.p, v.n);
                owned = true;
            }
            rest_of_line(first_from, line);
            arena.back().append(line.p, line.n);
            v.p = arena.back().data();
            v.n = arena.back().size();
        }
        // in kseq's error state ks_getuntil2 leaves with -3 in front of the strip (src/kseq.h:105-108): a line the error cut off keeps its '\r'
        if (d.stream_error && pos >= d.size() && (d.size() == 0 || d.p[d.size() - 1] != '\n')) return;
        // kseq strips the '\r' at the end of ks_getuntil2 — which returns before that (-1) when the stream is at its end right
        // behind the line's first character, the one kseq_read consumed itself with ks_getc (src/kseq.h:141,188-190): a last
        // line that is a bare '\r' without a newline keeps it
        if (first_consumed && first_from + 1 >= d.size()) return;
        if (v.n > 1 && v.p[v.n - 1] == '\r') {
            --v.n;
            if (owned) arena.back().pop_back();
        }
    }
    // Position of the header character ('@' / '>') of the record next() would return, or size() if there is none:
    // performs the same skip next() starts with, without consuming the header.
    size_t peek_header() {
        if (last_char != 0) return pos - 1;
        while (pos < d.size() && d.p[pos] != '>' && d.p[pos] != '@') ++pos;
        return pos;
    }
    // returns length >= 0, -1 at EOF, -2 on truncated / mismatching quality, -3 in the stream's error state   (src/kseq.h:176-224)
    //
    // d.stream_error: where the bytes end, gzread returned -1 and every ks_getc / ks_getuntil2 from there on returns -3 instead of
    // "end of file" (src/kseq.h:71-76,105-108).  kseq_read only LOOKS at that in three places — the search for the header and the
    // name return it, the '+' line compares with -1 only — so a record the error cuts off ends as: header search / name: -3;
    // sequence: a FASTA-like record of what was read; behind the '+': -2 unless nothing is missing (or the record is empty).
    long long next(Record &r) {
        int c;
        const long long at_end = d.stream_error ? -3 : -1;
        if (last_char == 0) {
            while ((c = getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) return at_end;
            last_char = c;
        }
        r = Record();
        if (pos >= d.size()) return at_end;
        size_t e = pos;
        while (e < d.size() && !isspace((unsigned char)d.p[e])) ++e;  // the name ends at the first whitespace
        r.name.p = d.data() + pos;
        r.name.n = e - pos;
        if (e >= d.size() && d.stream_error) return -3;  // ks_getuntil ran into the error before it found the delimiter
        c = e < d.size() ? (unsigned char)d.p[e] : -1;
        pos = e < d.size() ? e + 1 : e;
        if (c != '\n' && c >= 0) {
            rest_of_line(pos, r.comment);
            const bool cut_off = d.stream_error && pos >= d.size() && d.p[d.size() - 1] != '\n';  // (-3 leaves ks_getuntil2 in front of the strip)
            if (!cut_off && r.comment.n > 1 && r.comment.p[r.comment.n - 1] == '\r') --r.comment.n;
        }
        bool seq_owned = false, qual_owned = false;
        while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            append_line(r.seq, seq_owned, pos - 1, true);  // the line starts at the character just consumed
        }
        if (c == '>' || c == '@') last_char = c;
        r.is_fastq = (c == '+');
        if (!r.is_fastq) { if (c < 0) last_char = 0; return (long long)r.seq.size(); }
        while ((c = getc()) >= 0 && c != '\n') {}
        if (c == -1 && !d.stream_error) return -2;  // (ks_getc gives -3 in the error state, and kseq_read only compares with -1)
        for (;;) {
            if (pos >= d.size()) break;
            append_line(r.qual, qual_owned, pos, false);
            if (r.qual.size() >= r.seq.size()) break;
        }
        last_char = 0;
        if (r.seq.size() != r.qual.size()) return -2;
        return (long long)r.seq.size();
    }
};

// ---- whole-file parse -------------------------------------------------------------------------------------------------
struct Parsed {
    std::vector<Record> recs;
    std::deque<std::deque<std::string>> arenas;  // owners of the multi-line fields the records point into
    long long status = -1;                       // -1: clean EOF, -2: the record `bad` is truncated / mismatching, -3: the stream's error state
    Record bad;
};

static void parse_sequential(const Input &d, Parsed &out) {
    out.arenas.emplace_back();
    Parser p(d, out.arenas.back());
    Record r;
    for (;;) {
        const long long l = p.next(r);
        if (l < 0) { out.status = l; if (l == -2) out.bad = r; return; }
        out.recs.push_back(r);
    }
}

// A position that is certainly the start of a record.  FASTQ input (`fastq`: the file's first record is one): the beginning
// of a line starting with '@' (or '>') whose line + 2 starts with '+' and whose lines + 1 and + 3 have equal lengths — a
// quality line can start with '@' or '>' (Phred 31 / 29) too, but then the line after it is a header or a sequence, not '+'.
// FASTA input: a line starting with '>' (sequence lines never start with it).  Returns d.size() if none is found before `limit`.
static size_t find_record_start(const Input &d, size_t from, size_t limit, bool fastq) {
    const char *b = d.p;
    const size_t n = d.n;
    auto line_end = [&](size_t at) -> size_t { return at >= n ? n : (size_t)(std::find(b + at, b + n, '\n') - b); };
    size_t at = from;
    if (at > 0) at = line_end(at - 1) + 1;  // first line start >= from
    while (at < limit && at < n) {
        const size_t e0 = line_end(at);
        if (b[at] == '>' && !fastq) return at;
        if ((b[at] == '@' || b[at] == '>') && fastq && e0 < n) {
            const size_t s1 = e0 + 1, e1 = line_end(s1);
            const size_t s2 = e1 + 1;
            if (e1 < n && s2 < n && b[s2] == '+') {
                const size_t e2 = line_end(s2);
                const size_t s3 = e2 + 1, e3 = line_end(s3);
                if (e2 < n && e3 - s3 == e1 - s1 && e1 > s1) return at;
            }
        }
        at = e0 + 1;
    }
    return n;
}

// Chunks of the file parsed concurrently.  Chunk k starts at a certain record start S_k and stops when the next record
// would start at or after S_{k+1}; the result is accepted only if every chunk stopped EXACTLY at S_{k+1} between two
// records — then the concatenation is what the sequential parser produces (its state there is just "between records").
// Anything else (odd formats, an error inside a chunk) returns false and the caller parses sequentially.
// [from, to): `from` is 0 or a certain record start, `to` a certain record start or the end of the data (the whole file: 0, d.n);
// `fastq` as find_record_start wants it (the kind of the FILE's first record).
static bool parse_parallel_range(const Input &d, size_t from, size_t to, bool fastq, size_t min_bytes, Parsed &out) {
    const unsigned t = host_threads();
    const size_t len = to - from;
    if (t < 2 || len < min_bytes || len < t || d.stream_error) return false;  // (a damaged stream: the sequential parser knows how kseq ends)
    std::vector<size_t> start(t + 1, to);
    start[0] = from;
    for (unsigned k = 1; k < t; ++k) {
        start[k] = find_record_start(d, from + len / t * k, from + len / t * (k + 1), fastq);
        if (start[k] >= to || start[k] <= start[k - 1]) return false;
    }
    struct Chunk { std::vector<Record> recs; std::deque<std::string> arena; bool ok = false; };
    std::vector<Chunk> chunks(t);
    parallel_for(t, [&](size_t k) {
        Chunk &c = chunks[k];
        Parser p(d, c.arena);
        p.pos = start[k];
        const size_t stop = start[k + 1];
        Record r;
        for (;;) {
            const size_t h = p.peek_header();
            if (h >= stop) { c.ok = (h == stop); return; }
            const long long l = p.next(r);
            if (l < 0) { c.ok = false; return; }  // EOF inside a chunk that should end at a record start, or a bad record
            c.recs.push_back(r);
        }
    });
    size_t total = 0;
    for (auto &c : chunks) {
        if (!c.ok) return false;
        total += c.recs.size();
    }
    out.recs.reserve(total);
    for (auto &c : chunks) {
        out.recs.insert(out.recs.end(), c.recs.begin(), c.recs.end());
        out.arenas.push_back(std::move(c.arena));
    }
    out.status = -1;
    return true;
}

static bool first_record_is_fastq(const Input &d) {  // the kind of the first record decides which lines can start a record
    size_t h0 = 0;
    while (h0 < d.n && d.p[h0] != '>' && d.p[h0] != '@') ++h0;
    return h0 < d.n && d.p[h0] == '@';
}

static bool parse_parallel(const Input &d, Parsed &out) {
    size_t min_bytes = 32u << 20;
    if (const char *e = getenv("FLX_CLI_PARALLEL_PARSE_MIN")) min_bytes = (size_t)atoll(e);  // tests force it on small files
    if (d.n < min_bytes) return false;
    return parse_parallel_range(d, 0, d.n, first_record_is_fastq(d), min_bytes, out);
}

// One rank's share of a mapped file (review of round 3, item 8: a rank that parses only its byte range).  Rank r of `world` owns
// the records that START in [S_r, S_{r+1}), S_0 = 0, S_world = the end, S_k = the first certain record start at or behind byte
// k * n / world — every rank computes its two borders itself, they are a function of the file.  Accepted (true) only if the range
// ends exactly at S_{r+1} between two records and nothing in it is out of the ordinary (a parse error, the end of the data inside a
// record): then the ranks' shares, in rank order, are what one sequential parse of the file gives.  On false the caller falls back
// to parsing the whole file — on EVERY rank, which is why the decision is taken from a sum over the ranks.
static bool parse_rank_range(const Input &d, int rank, int world, Parsed &out) {
    if (world < 1 || rank < 0 || rank >= world || d.stream_error || d.n < (size_t)world) return false;
    const bool fastq = first_record_is_fastq(d);
    auto border = [&](int k) -> size_t {
        if (k <= 0) return 0;
        if (k >= world) return d.n;
        return find_record_start(d, d.n / (size_t)world * (size_t)k, d.n / (size_t)world * (size_t)(k + 1), fastq);
    };
    const size_t from = border(rank), to = border(rank + 1);
    if (from >= d.n || to <= from) return false;  // (no certain record start inside a slice: a file of very long or unusual records)
    if (rank > 0 && border(rank - 1) >= from) return false;
    size_t min_bytes = 32u << 20;
    if (const char *e = getenv("FLX_CLI_PARALLEL_PARSE_MIN")) min_bytes = (size_t)atoll(e);
    if (to - from >= min_bytes && parse_parallel_range(d, from, to, fastq, min_bytes, out)) return true;
    out = Parsed();
    out.arenas.emplace_back();
    Parser p(d, out.arenas.back());
    p.pos = from;
    Record r;
    for (;;) {
        const size_t h = p.peek_header();
        if (h >= to) { out.status = -1; return h == to; }
        const long long l = p.next(r);
        if (l < 0) return false;
        out.recs.push_back(r);
    }
}

static bool parse_all(const Input &d, Parsed &out) {  // true: the concurrent parse was accepted
    if (parse_parallel(d, out)) return true;
    out = Parsed();
    parse_sequential(d, out);
    return false;
}
