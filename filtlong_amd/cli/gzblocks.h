// gzblocks.h — gzip input without holding it in memory: block-wise inflate + parse (pass 1, leaves access points) and the
// record-aligned pieces the output pass inflates concurrently from those points.  Included by main.cpp only.
#pragma once
#include "fastx.h"
#include "inflate_stream.h"
#include "pinflate.h"

// ---- block-wise parse of a compressed input -----------------------------------------------------------------------------
// A gzip file cannot be mapped, and inflating all of it costs its uncompressed size in memory (the reference never holds
// more than one record, src/kseq.h:87-110).  BlockReader inflates a block at a time and parses the records that are complete
// inside it with the same Parser; the unfinished tail moves to the front of the next block.  A record is complete when the
// parser stopped BEFORE the end of the buffer: it then never saw the end, so more data behind it cannot change the record.
// Views of a batch are valid until the next call.  A record larger than the block doubles the buffer.
struct MappedFile {  // read-only mapping of a regular file (the compressed input)
    const unsigned char *p = nullptr;
    size_t n = 0;
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile() { if (p) munmap((void *)p, n); }
    bool open(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) { ::close(fd); return false; }
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return false;
        p = (const unsigned char *)m;
        n = (size_t)st.st_size;
        madvise(m, n, MADV_SEQUENTIAL);
        return true;
    }
    bool gz() const { return n >= 2 && p[0] == 0x1f && p[1] == 0x8b; }
};

struct BlockReader {
    static constexpr size_t kHistory = 32768;  // output kept in front of the write position: the window of an access point
    // A damaged gzip stream ends for the reference's parser up to 32 KiB BEFORE the byte at which zlib notices the damage (gzread
    // loses the whole 16 KiB call, inflate_stream.h: gzread_delivered_before_error).  So the last 32 KiB inflated are held back
    // from the parser until more has been inflated behind them or the stream has ended: no record is ever handed out that the
    // reference's kseq would not have seen whole.
    static constexpr size_t kHoldback = 32768;
    MappedFile file;
    ParallelInflate z;  // pass 1 of a large gzip input on all host threads; zlib for everything else
    std::vector<char> buf;
    size_t have = 0;        // valid bytes in buf
    size_t view_from = 0;   // the parse window is [view_from, have); bytes before it are history
    size_t carry_from = 0;  // where the next window starts (set by next())
    uint64_t buf_offset = 0;  // uncompressed offset of buf[0]
    bool done = false, io_error = false;
    bool stream_error = false;  // the stream ended in a data error: the last batch was parsed with kseq's error state behind its bytes
    std::vector<GzPoint> points;  // access points for a concurrent second pass (the first is the beginning of the file)
    uint64_t span = 0;
    BlockReader() = default;
    BlockReader(const BlockReader &) = delete;
    BlockReader &operator=(const BlockReader &) = delete;
    static size_t block_bytes() {
        if (const char *e = getenv("FLX_CLI_BLOCK_BYTES")) return std::max<size_t>(64, (size_t)atoll(e));  // tests force tiny blocks
        if (const char *e = getenv("FLX_CLI_BLOCK_MB")) return std::max<size_t>(1, (size_t)atoll(e)) << 20;
        return (size_t)256 << 20;
    }
    static uint64_t point_span() {  // uncompressed bytes between access points = the work unit of the output pass
        if (const char *e = getenv("FLX_CLI_SPAN_BYTES")) return std::max<uint64_t>(1, (uint64_t)atoll(e));
        return (uint64_t)32 << 20;
    }
    bool open(const std::string &path, bool want_points) {
        if (!file.open(path) || !z.open(file.p, file.n, file.gz(), ParallelInflate::default_threads(host_threads()))) return false;
        holdback = file.gz() ? kHoldback : 0;
        buf.resize(block_bytes() + kHistory + holdback);
        span = want_points ? point_span() : 0;
        points.clear();
        if (want_points) points.emplace_back();
        return true;
    }
    // uncompressed offset of a byte of the current batch
    uint64_t offset_of(const char *p) const { return buf_offset + (uint64_t)(p - buf.data()); }
    uint64_t end_offset() const { return buf_offset + have; }
    bool next(Parsed &out) {  // false: nothing left (or io_error)
        out = Parsed();
        if (done) return false;
        if (carry_from > 0) {  // drop what has been parsed, keep the history in front of the unfinished tail
            const size_t drop = carry_from > kHistory ? carry_from - kHistory : 0;
            if (drop > 0) {
                memmove(buf.data(), buf.data() + drop, have - drop);
                have -= drop;
                buf_offset += drop;
            }
            view_from = carry_from - drop;
            carry_from = 0;
        }
        for (;;) {
            if (!z.eof() && !z.error() && have < buf.size())
                have += z.read(buf.data() + have, buf.size() - have, span ? &points : nullptr, span);
            if (!file.gz() && buf_offset > dropped_ + ((uint64_t)64 << 20)) {
                // a plain file: the pages of the mapping behind the window are not looked at again by this pass — given back, so
                // that the resident set of a pass over a 20 GB file stays at the block's size (a later pass faults them in again
                // from the page cache)
                const uint64_t upto = buf_offset & ~(uint64_t)((2u << 20) - 1);
                if (upto > dropped_) madvise((void *)(file.p + dropped_), (size_t)(upto - dropped_), MADV_DONTNEED);
                dropped_ = upto;
            }
            bool eof = z.eof();
            size_t released = eof ? have : (have > holdback ? have - holdback : 0);
            if (z.error()) {  // the bytes gzread would have delivered, then kseq's error state (fastx.h: Input::stream_error)
                const uint64_t d = z.deliverable();
                released = d > buf_offset ? (size_t)std::min<uint64_t>(have, d - buf_offset) : 0;
                have = std::max(released, view_from);
                stream_error = true;
                eof = true;
            }
            released = std::max(released, view_from);
            view.p = buf.data() + view_from;
            view.n = released - view_from;
            view.stream_error = stream_error;
            out.arenas.emplace_back();
            Parser ps(view, out.arenas.back());
            Record r;
            size_t consumed = view.n;
            for (;;) {
                const size_t header = ps.peek_header();
                const long long len = ps.next(r);
                if (!eof && ps.pos >= view.n) { consumed = std::min(header, view.n); break; }  // ran into the end of the block: unfinished
                if (len == -1) break;
                if (len == -3) { out.status = -3; done = true; break; }
                if (len == -2) { out.status = -2; out.bad = r; done = true; break; }
                out.recs.push_back(r);
            }
            if (out.recs.empty() && !done && !eof && consumed == 0) {  // one record fills the whole block
                buf.resize((buf.size() - kHistory - holdback) * 2 + kHistory + holdback);
                out = Parsed();
                continue;
            }
            carry_from = view_from + consumed;
            if (eof) done = true;
            return true;
        }
    }

private:
    uint64_t dropped_ = 0;  // bytes of a plain file's mapping given back so far (a multiple of 2 MiB)
    size_t holdback = 0;
    Input view;  // non-owning window on buf
};

// The work units of the output pass over a streamed input: unit j is the text from the first record that starts at or
// after access point j up to the first record of unit j + 1, so every unit is a whole number of records and can be
// inflated (from its point) and parsed on its own.
struct UnitIndex {
    std::vector<uint64_t> start, first_rec;  // per access point, plus one closing entry (total size, record count)
    void note_record(const std::vector<GzPoint> &points, uint64_t header_offset, uint64_t rec) {
        while (start.size() < points.size() && points[start.size()].out <= header_offset) {
            start.push_back(header_offset);
            first_rec.push_back(rec);
        }
    }
    void finish(const std::vector<GzPoint> &points, uint64_t total_bytes, uint64_t n_records) {
        while (start.size() < points.size() + 1) {
            start.push_back(total_bytes);
            first_rec.push_back(n_records);
        }
    }
    size_t units() const { return start.empty() ? 0 : start.size() - 1; }
};

// bytes [from, to) of the uncompressed stream, inflated from an access point at or before `from`
static bool inflate_range(const MappedFile &file, const GzPoint &pt, uint64_t from, uint64_t to, std::vector<char> &text) {
    InflateStream z;
    if (pt.out > from || !z.open_at(file.p, file.n, file.gz(), pt)) return false;
    std::vector<char> skip(std::min<uint64_t>(from - pt.out, 1u << 20));
    for (uint64_t left = from - pt.out; left > 0;) {
        const size_t got = z.read(skip.data(), (size_t)std::min<uint64_t>(left, skip.size()));
        if (got == 0) return false;
        left -= got;
    }
    text.resize((size_t)(to - from));
    return z.read(text.data(), text.size()) == text.size() && !z.error();
}
