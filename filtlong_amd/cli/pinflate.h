// pinflate.h — block-parallel inflate of a gzip member: pass 1 of a compressed input on all host threads.
//
// The reference reads a gzip input through zlib's gzread, one thread (src/main.cpp:70 via src/kseq.h:87-110); pass 1 of the
// streamed reader here did the same (inflate_stream.h) and was the longest stage of a compressed run.  A deflate stream can only be
// decoded from a block boundary and with the 32 KiB of output in front of it — neither is known in the middle of a file.  The
// technique below is the one published with pugz / rapidgzip:
//   1. the compressed bytes are cut into chunks; each thread SEARCHES its chunk, bit by bit, for the start of a dynamic-Huffman
//      block: a header whose three code sets are complete and not over-subscribed, followed by a block that decodes to text;
//   2. it decodes from there with an UNKNOWN window: 16-bit symbols, a back-reference that reaches in front of the chunk becomes a
//      marker "byte w of the window" (256 + w) and is copied around like a literal;
//   3. a chunk stops at the block boundary where the next chunk's decoding began.  The decoding of chunk i runs through every true
//      block boundary, so a chunk whose start it meets exactly was started at a true boundary and its symbols are the true ones; a
//      start it runs past was a false positive and that chunk's work is dropped;
//   4. along the chain of chunks the windows are resolved one after the other (32 KiB each), then all markers are replaced and the
//      CRC-32 of every chunk computed on all threads; the member's CRC-32 and size are checked against the trailer.
// Every chain boundary is an access point (GzPoint) for the concurrent output pass, exactly as the serial reader leaves them.
// Whatever this decoder cannot do or does not like (a chunk that fails, a stream it finds corrupt, a second member, a small file)
// goes to zlib from the last boundary: zlib has the last word on every byte that is not plainly decodable.
// The deflate format itself is RFC 1951; nothing here is taken from zlib's or the reference's sources.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "inflate_stream.h"

namespace pinflate {

struct Entry {
    uint16_t val;  // literal byte / base of a length or distance / offset of a secondary table
    uint8_t bits;  // bits this entry consumes
    uint8_t op;
};
enum : uint8_t { OP_LIT = 0, OP_EOB = 1, OP_LINK = 2, OP_BAD = 3, OP_BASE = 16 /* + number of extra bits */ };

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
                                       6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// LSB-first bit reader over the mapped file
struct Bits {
    const uint8_t *base = nullptr, *end = nullptr, *p = nullptr;
    uint64_t buf = 0;
    int cnt = 0;  // valid bits in buf; negative: read past the end of the data
    void init(const uint8_t *data, size_t size) { base = data; end = data + size; }
    void seek(uint64_t bitpos) {
        p = base + (bitpos >> 3);
        buf = 0;
        cnt = 0;
        if (p > end) { p = end; cnt = -1; return; }
        refill();
        drop((int)(bitpos & 7));
    }
    uint64_t pos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
    inline void refill() {  // at least 56 valid bits afterwards (fewer only at the end of the data)
        if (p + 8 <= end) {
            uint64_t v;
            memcpy(&v, p, 8);
            buf |= v << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline uint32_t get(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

// One canonical Huffman code as a two-level table: `root` bits index the primary table, longer codes continue in a secondary table
// of 2^(max - root) entries behind it.
struct Huff {
    enum Kind { CODES, LENS, DISTS };
    std::vector<Entry> t;
    int root = 0, sub = 0;
    uint32_t root_mask = 0, sub_mask = 0;

    // false: over-subscribed, or incomplete where the format does not allow it (RFC 1951 3.2.7; a single one-bit code is the exception
    // decoders accept for length and distance sets)
    bool build(const uint8_t *lens, int n, int root_bits, Kind kind) {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) ++count[lens[i]];
        int max = 15;
        while (max >= 1 && count[max] == 0) --max;
        root = root_bits;
        root_mask = (1u << root) - 1;
        const Entry bad = {0, 1, OP_BAD};
        if (max == 0) {  // no code at all: every symbol is an error (a distance set may be empty when the block has no match)
            sub = 0; sub_mask = 0;
            t.assign((size_t)1 << root, bad);
            return true;
        }
        int left = 1;
        for (int len = 1; len <= 15; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;
        }
        if (left > 0 && (kind == CODES || max != 1)) return false;
        sub = max > root ? max - root : 0;
        sub_mask = (1u << sub) - 1;
        t.assign((size_t)1 << root, bad);
        uint32_t next_code[16];
        uint32_t code = 0;
        count[0] = 0;
        for (int len = 1; len <= 15; ++len) {
            code = (code + (uint32_t)count[len - 1]) << 1;
            next_code[len] = code;
        }
        for (int sym = 0; sym < n; ++sym) {
            const int len = lens[sym];
            if (!len) continue;
            const uint32_t c = next_code[len]++;
            uint32_t rev = 0;
            for (int b = 0; b < len; ++b) rev |= ((c >> b) & 1u) << (len - 1 - b);
            Entry e;
            if (kind == CODES) {
                e = {(uint16_t)sym, 0, OP_LIT};
            } else if (kind == LENS) {
                if (sym < 256) e = {(uint16_t)sym, 0, OP_LIT};
                else if (sym == 256) e = {0, 0, OP_EOB};
                else if (sym < 286) e = {kLenBase[sym - 257], 0, (uint8_t)(OP_BASE + kLenExtra[sym - 257])};
                else e = {0, 0, OP_BAD};
            } else {
                if (sym < 30) e = {kDistBase[sym], 0, (uint8_t)(OP_BASE + kDistExtra[sym])};
                else e = {0, 0, OP_BAD};
            }
            if (len <= root) {
                e.bits = (uint8_t)len;
                for (uint32_t k = rev; k < (1u << root); k += 1u << len) t[k] = e;
            } else {
                const uint32_t prefix = rev & root_mask;
                if (t[prefix].op != OP_LINK) {
                    const size_t at = t.size();
                    if (at > 60000) return false;
                    t.resize(at + ((size_t)1 << sub), bad);
                    t[prefix] = {(uint16_t)at, (uint8_t)root, OP_LINK};
                }
                e.bits = (uint8_t)(len - root);
                const size_t at = t[prefix].val;
                for (uint32_t k = rev >> root; k < (1u << sub); k += 1u << (len - root)) t[at + k] = e;
            }
        }
        return true;
    }
    inline Entry decode(Bits &in) const {  // needs max <= 15 valid bits
        Entry e = t[in.buf & root_mask];
        if (e.op == OP_LINK) {
            in.drop(root);
            e = t[e.val + (in.buf & sub_mask)];
        }
        in.drop(e.bits);
        return e;
    }
};

static inline bool text_byte(unsigned c) { return c >= 32 ? c != 127 : (c == 9 || c == 10 || c == 13); }

enum Status { OK = 0, FAIL = 1, LIMIT = 2 };

// decoder of one thread: tables, the fixed code, the symbol output
struct Decoder {
    Bits in;
    Huff lit, dist, cl, fixed_lit, fixed_dist;
    bool have_fixed = false;
    std::vector<uint16_t> out;  // symbols: < 256 a byte, else 256 + index into the 32 KiB in front of the chunk
    size_t n = 0;               // symbols produced
    size_t limit = 0;           // give up beyond this many symbols

    void ensure_fixed() {
        if (have_fixed) return;
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        fixed_lit.build(l, 288, 10, Huff::LENS);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        fixed_dist.build(d, 32, 8, Huff::DISTS);
        have_fixed = true;
    }

    // the header of a dynamic block (after its three header bits): builds lit / dist
    bool dynamic_header() {
        in.refill();
        const int nlen = (int)in.get(5) + 257, ndist = (int)in.get(5) + 1, ncode = (int)in.get(4) + 4;
        if (nlen > 286 || ndist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cll[19] = {0};
        for (int i = 0; i < ncode; ++i) {
            if ((i & 7) == 0) in.refill();
            cll[order[i]] = (uint8_t)in.get(3);
        }
        if (in.cnt < 0) return false;
        if (!cl.build(cll, 19, 7, Huff::CODES)) return false;
        uint8_t lens[286 + 30];
        int i = 0;
        while (i < nlen + ndist) {
            in.refill();
            const Entry e = cl.decode(in);
            if (e.op != OP_LIT) return false;
            const int sym = e.val;
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            int rep;
            uint8_t prev = 0;
            if (sym == 16) {
                if (i == 0) return false;
                prev = lens[i - 1];
                rep = 3 + (int)in.get(2);
            } else if (sym == 17) {
                rep = 3 + (int)in.get(3);
            } else {
                rep = 11 + (int)in.get(7);
            }
            if (i + rep > nlen + ndist) return false;
            while (rep--) lens[i++] = prev;
            if (in.cnt < 0) return false;
        }
        if (in.cnt < 0) return false;
        if (lens[256] == 0) return false;  // no end-of-block code
        return lit.build(lens, nlen, 10, Huff::LENS) && dist.build(lens + nlen, ndist, 8, Huff::DISTS);
    }

    // the symbols of one compressed block with the code sets l / d, up to its end-of-block code (bit reader and output position
    // in locals: the loop keeps them in registers)
    Status block(const Huff &l, const Huff &d, bool check_text) { return check_text ? block_t<true>(l, d) : block_t<false>(l, d); }

    template <bool TEXT>
    Status block_t(const Huff &l, const Huff &d) {
        const Entry *const lt = l.t.data(), *const dt = d.t.data();
        const uint64_t lrm = l.root_mask, lsm = l.sub_mask, drm = d.root_mask, dsm = d.sub_mask;
        const int lroot = l.root, droot = d.root;
        uint64_t buf = in.buf;
        int cnt = in.cnt;
        const uint8_t *p = in.p;
        const uint8_t *const end = in.end;
        uint16_t *base = out.data();
        size_t cap = out.size(), m = n;
        Status result = FAIL;
#define FLX_PINF_REFILL()                                                              \
        do {                                                                           \
            if (p + 8 <= end) {                                                        \
                uint64_t v_;                                                           \
                memcpy(&v_, p, 8);                                                     \
                buf |= v_ << cnt;                                                      \
                p += (63 - cnt) >> 3;                                                  \
                cnt |= 56;                                                             \
            } else {                                                                   \
                while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; } \
            }                                                                          \
        } while (0)
#define FLX_PINF_LIT(e)                                                                \
        do {                                                                           \
            e = lt[buf & lrm];                                                         \
            if (e.op == OP_LINK) { buf >>= lroot; cnt -= lroot; e = lt[e.val + (buf & lsm)]; } \
            buf >>= e.bits; cnt -= e.bits;                                             \
        } while (0)
        for (;;) {
            if (m + 600 > cap) {
                if (m + 600 > limit) { result = LIMIT; break; }
                out.resize(std::min(limit, std::max<size_t>(cap * 2, (size_t)1 << 16)));
                base = out.data();
                cap = out.size();
            }
            FLX_PINF_REFILL();
            Entry e;
            FLX_PINF_LIT(e);
            if (e.op == OP_LIT) {  // the refill covers three codes of 15 bits: most symbols of a text are literals
                if (TEXT && !text_byte(e.val)) break;
                base[m++] = e.val;
                FLX_PINF_LIT(e);
                if (e.op == OP_LIT) {
                    if (TEXT && !text_byte(e.val)) break;
                    base[m++] = e.val;
                    FLX_PINF_LIT(e);
                    if (e.op == OP_LIT) {
                        if (TEXT && !text_byte(e.val)) break;
                        base[m++] = e.val;
                        if (cnt < 0) break;
                        continue;
                    }
                }
                FLX_PINF_REFILL();  // (a length + distance needs up to 48 bits)
            }
            if (e.op == OP_EOB) { result = cnt < 0 ? FAIL : OK; break; }
            if (e.op < OP_BASE) break;
            const int lx = e.op - OP_BASE;
            const unsigned len = e.val + (unsigned)(buf & ((1u << lx) - 1));
            buf >>= lx; cnt -= lx;
            Entry de = dt[buf & drm];
            if (de.op == OP_LINK) { buf >>= droot; cnt -= droot; de = dt[de.val + (buf & dsm)]; }
            buf >>= de.bits; cnt -= de.bits;
            if (de.op < OP_BASE) break;
            const int dx = de.op - OP_BASE;
            const size_t dd = (size_t)de.val + (size_t)(buf & ((1u << dx) - 1));
            buf >>= dx; cnt -= dx;
            if (cnt < 0) break;
            uint16_t *o = base + m;
            if (dd <= m) {
                const uint16_t *s = o - dd;
                if (dd >= len) memcpy(o, s, (size_t)len * 2);
                else for (unsigned k = 0; k < len; ++k) o[k] = s[k];
            } else {  // reaches in front of the chunk: markers for the part that does
                const size_t before = dd - m;  // symbols of the match that lie in front of the chunk (if it is that long)
                const unsigned nb = (unsigned)std::min<size_t>(before, len);
                for (unsigned k = 0; k < nb; ++k) o[k] = (uint16_t)(256 + 32768 - before + k);
                for (unsigned k = nb; k < len; ++k) o[k] = base[k - nb];
            }
            m += len;
        }
#undef FLX_PINF_REFILL
#undef FLX_PINF_LIT
        in.buf = buf; in.cnt = cnt; in.p = p;
        n = m;
        return result;
    }

    // one block from its three header bits on; *final_block = BFINAL
    Status any_block(bool *final_block, bool check_text) {
        in.refill();
        if (in.cnt < 3) return FAIL;
        *final_block = in.get(1) != 0;
        const unsigned type = in.get(2);
        if (type == 2) {
            if (!dynamic_header()) return FAIL;
            return block(lit, dist, check_text);
        }
        if (type == 1) {
            ensure_fixed();
            return block(fixed_lit, fixed_dist, check_text);
        }
        if (type != 0) return FAIL;
        in.drop(in.cnt & 7);  // stored: to the byte boundary
        in.refill();
        const unsigned len = in.get(16), nlen = in.get(16);
        if (in.cnt < 0 || (len ^ nlen) != 0xffffu) return FAIL;
        if (n + len + 600 > out.size()) {
            if (n + len + 600 > limit) return LIMIT;
            out.resize(std::min(limit, std::max<size_t>(out.size() * 2, n + len + 600)));
        }
        for (unsigned k = 0; k < len; ++k) {
            if ((k & 3) == 0) in.refill();
            const unsigned c = in.get(8);
            if (check_text && !text_byte(c)) return FAIL;
            out[n++] = (uint16_t)c;
        }
        return in.cnt < 0 ? FAIL : OK;
    }
};

// first position in [from, to) that looks like the start of a non-final dynamic block whose content is text and which is followed by
// another well-formed block header
static bool find_block_start(Decoder &d, uint64_t from, uint64_t to, uint64_t *start) {
    Decoder &probe = d;
    for (uint64_t c = from; c < to; ++c) {
        probe.in.seek(c);
        if (probe.in.cnt < 17) return false;
        const uint32_t h = probe.in.peek(13);
        // BFINAL = 0, BTYPE = 2, HLIT <= 29 (286 length codes), HDIST <= 29
        if ((h & 7) != 4 || ((h >> 3) & 31) > 29 || ((h >> 8) & 31) > 29) continue;
        probe.n = 0;
        bool fin = false;
        if (probe.any_block(&fin, true) != OK || fin || probe.n == 0) continue;
        // the block after it: a header that holds together
        probe.in.refill();
        if (probe.in.cnt < 3) continue;
        probe.in.drop(1);
        const unsigned type = probe.in.get(2);
        if (type == 3) continue;
        if (type == 2 && !probe.dynamic_header()) continue;
        if (type == 0) {
            probe.in.drop(probe.in.cnt & 7);
            probe.in.refill();
            const unsigned len = probe.in.get(16), nlen = probe.in.get(16);
            if (probe.in.cnt < 0 || (len ^ nlen) != 0xffffu) continue;
        }
        *start = c;
        return true;
    }
    return false;
}

// false: a worker threw (an allocation that failed): the caller leaves the job to zlib instead of ending the process
template <class F>
static bool run_parallel(size_t n, unsigned threads, F &&body) {
    std::atomic<bool> ok{true};
    auto guarded = [&](size_t i) {
        try { body(i); } catch (...) { ok = false; }
    };
    if (n <= 1 || threads <= 1) { for (size_t i = 0; i < n; ++i) guarded(i); return ok; }
    std::vector<std::thread> th;
    const unsigned t = (unsigned)std::min<size_t>(threads, n);
    for (unsigned k = 0; k < t; ++k) th.emplace_back([&, k] { for (size_t i = k; i < n; i += t) guarded(i); });
    for (auto &x : th) x.join();
    return ok;
}

}  // namespace pinflate

// Same interface as InflateStream (which it falls back to and hands over to): sequential reads of the uncompressed bytes.
class ParallelInflate {
public:
    static size_t min_bytes() {  // smaller files are not worth the threads
        if (const char *e = getenv("FLX_CLI_PINFLATE_MIN")) return (size_t)atoll(e);
        return (size_t)8 << 20;
    }
    static unsigned default_threads(unsigned host_threads) {  // decoding scales further than the host's memory-bound stages
        if (const char *e = getenv("FLX_CLI_INFLATE_THREADS")) return (unsigned)std::max(1, atoi(e));
        if (getenv("FLX_CLI_THREADS")) return host_threads;
        return std::max(host_threads, std::min(32u, std::thread::hardware_concurrency()));
    }
    static size_t chunk_bytes() {  // compressed bytes per chunk (tests force tiny chunks)
        if (const char *e = getenv("FLX_CLI_PINFLATE_CHUNK")) return std::max<size_t>(64, (size_t)atoll(e));
        return (size_t)2 << 20;
    }

    ParallelInflate() = default;
    ParallelInflate(const ParallelInflate &) = delete;
    ParallelInflate &operator=(const ParallelInflate &) = delete;
    ~ParallelInflate() { shutdown(); }

    bool open(const unsigned char *data, size_t size, bool gz, unsigned threads) {
        shutdown();
        data_ = data; size_ = size; threads_ = std::max(1u, threads);
        serial_mode_ = true; done_ = false; error_ = false;
        delivered_ = 0; queue_.clear(); pending_.clear();
        shared_q_.clear(); shared_pts_.clear(); queued_bytes_ = 0; finished_ = false; stop_ = false;
        stage_q_.clear(); stage_pts_.clear(); p_finished_ = false; p_error_ = false; p_handover_set_ = false; p_deliverable_ = 0;
        stream_out_ = 0; member_base_ = 0; bgzf_mode_ = false; bgzf_at_ = 0;
        parallel_bytes_ = 0; zlib_tail_bytes_ = 0; rounds_ = 0; dropped_chunks_ = 0;
        const char *off = getenv("FLX_CLI_PINFLATE");  // 0: zlib only; "nozlib": every chunk to its end with the marker decoder (tests)
        zlib_tails_ = !(off && strcmp(off, "nozlib") == 0);
        if (const char *e = getenv("FLX_CLI_PINFLATE_AHEAD_MB")) max_ahead_ = (size_t)std::max(1, atoi(e)) << 20;
        if (gz && threads_ >= 2 && size >= min_bytes() && !(off && off[0] == '0')) {
            decoders_.resize(threads_);
            last_point_out_ = 0;
            begin_member(0);
            if (p_error_) { error_ = true; return false; }
            if (p_handover_set_) return take_over();  // nothing for the parallel paths: zlib from the first byte
            serial_mode_ = false;
            producer_ = std::thread([this] { produce(); });  // decodes ahead of read(), at most max_ahead_ bytes
            return true;
        }
        return serial_.open(data, size, gz);
    }
    bool eof() const { return queue_.empty() && (serial_mode_ ? serial_.eof() : done_); }
    bool error() const { return error_ || (serial_mode_ && serial_.error()); }
    // after error(): how many bytes of the stream zlib's gzread would have handed to the reference's parser before its error
    // state (inflate_stream.h: gzread_delivered_before_error) — up to 32 KiB fewer than read() has produced.  A stream that is
    // merely TRUNCATED is no error: everything decodable comes out, then eof(), like gzread.
    uint64_t deliverable() const { return serial_mode_ && serial_.error() ? serial_.deliverable() : p_deliverable_; }
    uint64_t total_out() const { return delivered_; }
    // diagnostics: bytes that came out of the parallel path (of those: from zlib running behind the marker decoder), rounds, chunks whose
    // work was dropped
    uint64_t parallel_bytes() const { return parallel_bytes_; }
    uint64_t zlib_tail_bytes() const { return zlib_tail_bytes_; }
    unsigned rounds() const { return rounds_; }
    unsigned dropped_chunks() const { return dropped_chunks_; }

    // like InflateStream::read: up to cap bytes to dst (the caller keeps the previous 32 KiB in front of dst + n); access points at
    // least `span` output bytes apart are appended to *points
    size_t read(char *dst, size_t cap, std::vector<GzPoint> *points = nullptr, uint64_t span = 0) {
        size_t produced = 0;
        while (produced < cap && !error_) {
            if (queue_.empty() && !serial_mode_ && !done_) {  // what the producer has ready; wait for it if that is nothing
                bool fin;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return !shared_q_.empty() || finished_; });
                    for (Piece &pc : shared_q_) queue_.push_back(std::move(pc));
                    for (GzPoint &pt : shared_pts_) pending_.push_back(std::move(pt));
                    shared_q_.clear();
                    shared_pts_.clear();
                    queued_bytes_ = 0;
                    fin = finished_;
                }
                cv_.notify_all();
                if (queue_.empty() && fin) {  // the producer has left: the end, an error, or zlib's turn
                    if (producer_.joinable()) producer_.join();
                    if (p_error_) error_ = true;
                    else if (p_handover_set_) take_over();
                    else done_ = true;
                }
                continue;
            }
            if (!queue_.empty()) {  // whole chunks to their places on all threads
                struct Copy { const char *from; char *to; size_t n; };
                std::vector<Copy> plan;
                size_t at = produced;
                for (const Piece &pc : queue_) {
                    const size_t m = std::min(cap - at, pc.n - pc.pos);
                    if (m == 0) break;
                    for (size_t o = 0; o < m; o += (size_t)4 << 20)
                        plan.push_back({pc.bytes.data() + pc.pos + o, dst + at + o, std::min<size_t>(m - o, (size_t)4 << 20)});
                    at += m;
                    if (m < pc.n - pc.pos) break;
                }
                pinflate::run_parallel(plan.size(), threads_, [&](size_t k) { memcpy(plan[k].to, plan[k].from, plan[k].n); });
                size_t left = at - produced;
                produced = at;
                delivered_ += left;
                while (left > 0) {
                    Piece &pc = queue_.front();
                    const size_t m = std::min(left, pc.n - pc.pos);
                    pc.pos += m;
                    left -= m;
                    if (pc.pos >= pc.n) {
                        {
                            std::lock_guard<std::mutex> g(pool_mutex_);
                            if (pool_.size() < 3 * (size_t)threads_) pool_.push_back(std::move(pc.bytes));
                        }
                        queue_.pop_front();
                    }
                }
                while (!pending_.empty() && pending_.front().out <= delivered_) {
                    if (points && pending_.front().out - last_point_out_ >= span) {
                        last_point_out_ = pending_.front().out;
                        points->push_back(std::move(pending_.front()));
                    }
                    pending_.pop_front();
                }
                continue;
            }
            if (serial_mode_) {
                const size_t m = serial_.read(dst + produced, cap - produced, points, span);
                produced += m; delivered_ += m;
                break;  // (less than asked for only at the end or on an error)
            }
            break;  // done_
        }
        return produced;
    }

private:
    struct Piece {
        std::vector<char> bytes;
        size_t n = 0, pos = 0;
    };
    struct Chunk {
        uint64_t begin = 0, end = 0;  // the bits this chunk searches
        uint64_t start = 0;           // where its decoding began
        bool found = false;
        pinflate::Status status = pinflate::FAIL;
        uint64_t stop = 0;   // block boundary it stopped at
        int arrived = -1;    // the chunk whose start that is (-1: the end of the round)
        bool final_block = false;
        std::vector<uint16_t> sym;  // the head of the chunk as symbols (markers possible) ...
        size_t n_sym = 0;
        std::vector<char> bytes;    // ... and all of it as bytes: [0, n_sym) filled in once the window is known, the rest by zlib
        size_t n = 0;
        uint64_t out = 0;    // member offset of its first byte
        uLong crc = 0;
        bool bad_marker = false;
        std::string window;  // the (up to) 32 KiB in front of it
    };

    // RFC 1952 header at byte `at`; *bgzf_size: the member's total size if it carries the BGZF subfield (SAM spec 4.1), else 0
    bool gzip_header(size_t at, size_t *deflate_at, size_t *bgzf_size) const {
        *bgzf_size = 0;
        if (at + 18 + 8 > size_ || data_[at] != 0x1f || data_[at + 1] != 0x8b || data_[at + 2] != 8) return false;
        const unsigned flg = data_[at + 3];
        if (flg & 0xe0) return false;
        size_t p = at + 10;
        if (flg & 4) {
            if (p + 2 > size_) return false;
            const size_t xlen = (size_t)data_[p] | (size_t)data_[p + 1] << 8;
            p += 2;
            if (p + xlen > size_) return false;
            for (size_t q = p; q + 4 <= p + xlen;) {
                const size_t slen = (size_t)data_[q + 2] | (size_t)data_[q + 3] << 8;
                if (data_[q] == 'B' && data_[q + 1] == 'C' && slen == 2 && q + 6 <= p + xlen)
                    *bgzf_size = ((size_t)data_[q + 4] | (size_t)data_[q + 5] << 8) + 1;
                q += 4 + slen;
            }
            p += xlen;
        }
        for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
            if (flg & f) {
                while (p < size_ && data_[p]) ++p;
                ++p;
            }
        if (flg & 2) p += 2;
        if (p + 8 >= size_) return false;
        *deflate_at = p;
        return true;
    }

    // a member starts at byte `at` (stream offset delivered so far + queued = stream_out_): BGZF blocks are inflated side by side, a
    // large plain member goes to the chunked decoder, anything else to zlib for the rest of the file
    void begin_member(size_t at) {
        size_t hdr = 0, bsize = 0;
        bgzf_at_ = 0;
        bgzf_mode_ = false;
        if (gzip_header(at, &hdr, &bsize)) {
            if (bsize > 0 && at + bsize <= size_) {
                bgzf_mode_ = true;
                bgzf_at_ = at;
                return;
            }
            if (size_ - at >= min_bytes() || at == 0) {
                chain_bit_ = (uint64_t)hdr * 8;
                member_base_ = stream_out_;
                member_out_ = 0;
                crc_ = crc32(0L, Z_NULL, 0);
                window_.clear();
                return;
            }
        }
        GzPoint pt;
        pt.in = at;
        pt.out = stream_out_;
        pt.raw = false;
        hand_over(pt);
    }

    // producer: zlib reads on from this point (done by the reading thread once it has taken everything decoded before it)
    void hand_over(const GzPoint &pt) {
        p_handover_ = pt;
        p_handover_set_ = true;
        p_finished_ = true;
    }
    // reader: the serial stream takes over where the producer stopped
    bool take_over() {
        serial_mode_ = true;
        const GzPoint &pt = p_handover_;
        const bool ok = (pt.in == 0 && !pt.raw) ? serial_.open(data_, size_, true) : serial_.open_at(data_, size_, true, pt);
        if (!ok) error_ = true;
        return ok;
    }
    // the producer thread: round after round, published to the reader, never more than max_ahead_ bytes in front of it
    void produce() {
        for (;;) {
            if (!p_finished_) {
                // a round that cannot get its memory (std::bad_alloc out of a vector) must not end the process: what it staged is
                // dropped, the few words of state it advanced are put back, and zlib reads on alone from where the round began
                const uint64_t chain_bit0 = chain_bit_, member_out0 = member_out_, stream_out0 = stream_out_;
                const uLong crc0 = crc_;
                const size_t bgzf_at0 = bgzf_at_;
                const bool bgzf0 = bgzf_mode_;
                const std::string window0 = window_;
                const size_t staged_q = stage_q_.size(), staged_p = stage_pts_.size();
                bool threw = false;
                try {
                    if (bgzf_mode_) bgzf_round();
                    else if (!round()) threw = true;
                } catch (...) { threw = true; }
                if (threw) {
                    stage_q_.resize(staged_q);
                    stage_pts_.resize(staged_p);
                    chain_bit_ = chain_bit0; member_out_ = member_out0; stream_out_ = stream_out0; crc_ = crc0; window_ = window0;
                    p_error_ = false;
                    if (bgzf0) {
                        GzPoint pt;
                        pt.in = bgzf_at0;
                        pt.out = stream_out0;
                        pt.raw = false;
                        bgzf_mode_ = false;
                        hand_over(pt);
                    } else {
                        hand_over(point_at(chain_bit0, member_out0, window0, true));
                    }
                }
            }
            std::unique_lock<std::mutex> lk(mu_);
            for (Piece &pc : stage_q_) { queued_bytes_ += pc.n; shared_q_.push_back(std::move(pc)); }
            for (GzPoint &pt : stage_pts_) shared_pts_.push_back(std::move(pt));
            stage_q_.clear();
            stage_pts_.clear();
            if (p_finished_ || stop_) { finished_ = true; lk.unlock(); cv_.notify_all(); return; }
            cv_.notify_all();
            cv_.wait(lk, [&] { return queued_bytes_ <= max_ahead_ || stop_; });
            if (stop_) { finished_ = true; lk.unlock(); cv_.notify_all(); return; }
        }
    }
    void shutdown() {
        if (producer_.joinable()) {
            { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
            cv_.notify_all();
            producer_.join();
        }
    }

    // BGZF: up to threads x 128 blocks, every thread a run of neighbours through zlib (which checks each block's CRC-32 and size)
    void bgzf_round() {
        struct Member { size_t at, size; uint32_t isize; };
        std::vector<Member> mem;
        size_t at = bgzf_at_;
        bool more = true, forged = false;
        while (mem.size() < (size_t)threads_ * 128) {
            size_t hdr = 0, bsize = 0;
            if (at + 2 > size_ || data_[at] != 0x1f || data_[at + 1] != 0x8b) { more = false; break; }  // the end of the file (or not a member)
            if (!gzip_header(at, &hdr, &bsize) || bsize == 0 || at + bsize > size_ || bsize < hdr - at + 8) break;  // not BGZF: decided below
            const size_t t = at + bsize - 4;
            const uint32_t isize = (uint32_t)data_[t] | (uint32_t)data_[t + 1] << 8 | (uint32_t)data_[t + 2] << 16 | (uint32_t)data_[t + 3] << 24;
            if (isize > 65536) { forged = true; break; }  // a BGZF block holds at most 64 KiB (SAM spec 4.1): not one, or damaged — zlib's
            mem.push_back({at, bsize, isize});
            at += bsize;
        }
        if (mem.empty()) {
            if (forged) {
                GzPoint pt;
                pt.in = at;
                pt.out = stream_out_;
                pt.raw = false;
                bgzf_mode_ = false;
                hand_over(pt);
                return;
            }
            if (!more) { p_finished_ = true; return; }
            begin_member(at);  // a member of another kind
            if (bgzf_mode_) { p_error_ = true; p_finished_ = true; }  // (cannot happen: the loop above would have taken it)
            return;
        }
        const size_t runs = std::min<size_t>(threads_, mem.size());
        std::vector<Piece> pieces(runs);
        std::vector<int> ok(runs, 1);
        std::vector<size_t> first(runs + 1);
        for (size_t r = 0; r <= runs; ++r) first[r] = mem.size() * r / runs;
        const bool all_ran = pinflate::run_parallel(runs, threads_, [&](size_t r) {
            size_t total = 0;
            for (size_t k = first[r]; k < first[r + 1]; ++k) total += mem[k].isize;
            Piece &pc = pieces[r];
            pc.bytes = take_buffer();
            if (pc.bytes.size() < total) pc.bytes.resize(total);
            size_t have = 0;
            z_stream z;
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, 31) != Z_OK) { ok[r] = 0; return; }
            for (size_t k = first[r]; k < first[r + 1] && ok[r]; ++k) {
                z.next_in = (Bytef *)data_ + mem[k].at; z.avail_in = (uInt)mem[k].size;
                z.next_out = (Bytef *)pc.bytes.data() + have; z.avail_out = (uInt)mem[k].isize;
                const int ret = inflate(&z, Z_FINISH);
                if (ret != Z_STREAM_END || z.avail_in != 0 || z.avail_out != 0) ok[r] = 0;
                have += mem[k].isize;
                if (inflateReset(&z) != Z_OK) ok[r] = 0;
            }
            inflateEnd(&z);
            pc.n = total;
        });
        if (!all_ran) ok.assign(runs, 0);  // a worker could not get its memory: zlib reads on alone from the first run
        ++rounds_;
        for (size_t r = 0; r < runs; ++r) {
            if (!ok[r]) {  // zlib reads this run again, alone, and says what is wrong with it
                GzPoint pt;
                pt.in = mem[first[r]].at;
                pt.out = stream_out_;
                pt.raw = false;
                bgzf_mode_ = false;
                hand_over(pt);
                return;
            }
            if (stream_out_ > 0) {
                GzPoint pt;
                pt.in = mem[first[r]].at;
                pt.out = stream_out_;
                pt.raw = false;
                stage_pts_.push_back(std::move(pt));
            }
            stream_out_ += pieces[r].n;
            parallel_bytes_ += pieces[r].n;
            if (pieces[r].n > 0) stage_q_.push_back(std::move(pieces[r]));
        }
        bgzf_at_ = at;
    }

    // check: the point is a hand-over to zlib inside the member — crc_ covers the member's bytes in front of it, and whoever
    // inflates on verifies the trailer (an access point of the output pass re-reads validated bytes: no check)
    GzPoint point_at(uint64_t bit, uint64_t out, const std::string &window, bool check = false) const {
        GzPoint pt;
        pt.raw = true;
        pt.out = member_base_ + out;
        pt.in = (bit + 7) >> 3;
        pt.bits = (int)((8 - (bit & 7)) & 7);
        pt.window = window;
        pt.check = check;
        pt.member_base = member_base_;
        pt.crc = crc_;
        return pt;
    }

    std::vector<char> take_buffer() {
        std::lock_guard<std::mutex> g(pool_mutex_);
        if (pool_.empty()) return std::vector<char>();
        std::vector<char> v = std::move(pool_.back());
        pool_.pop_back();
        return v;
    }

    // Chunk c from block boundary `bit` on with zlib: the 32 KiB in front of it are known bytes (the last 32768 symbols are literals).
    // Stops like the marker decoder: at a later chunk's start, at the first boundary behind the round, at the end of the member.
    void zlib_tail(Chunk &c, uint64_t bit, const std::vector<Chunk> &ch, size_t j, uint64_t round_end, size_t limit) {
        using namespace pinflate;
        const size_t n = ch.size();
        char dict[32768];
        for (size_t q = 0; q < 32768; ++q) dict[q] = (char)c.sym[c.n_sym - 32768 + q];
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { c.status = FAIL; return; }
        const uint64_t in0 = (bit + 7) >> 3;
        const int bits = (int)((8 - (bit & 7)) & 7);
        bool ok = true;
        if (bits > 0) ok = inflatePrime(&z, bits, data_[in0 - 1] >> (8 - bits)) == Z_OK;
        ok = ok && inflateSetDictionary(&z, (const Bytef *)dict, 32768) == Z_OK;
        uint64_t in_pos = in0;
        size_t have = c.n_sym;
        c.status = FAIL;
        while (ok) {
            if (c.bytes.size() < have + ((size_t)1 << 16)) {
                if (have > limit) { c.status = LIMIT; break; }
                c.bytes.resize(std::max<size_t>(c.bytes.size() * 2, have + ((size_t)1 << 20)));
            }
            const size_t out_room = std::min<size_t>(c.bytes.size() - have, (size_t)1 << 30);
            const size_t in_room = (size_t)std::min<uint64_t>(size_ - in_pos, (uint64_t)1 << 30);
            z.next_out = (Bytef *)c.bytes.data() + have; z.avail_out = (uInt)out_room;
            z.next_in = (Bytef *)data_ + in_pos; z.avail_in = (uInt)in_room;
            const int ret = inflate(&z, Z_BLOCK);
            const size_t got = out_room - z.avail_out;
            have += got;
            in_pos += in_room - z.avail_in;
            if (ret == Z_STREAM_END) { c.status = OK; c.final_block = true; c.stop = in_pos * 8; break; }
            if (ret != Z_OK && ret != Z_BUF_ERROR) break;
            if (got == 0 && in_room == z.avail_in && (ret == Z_BUF_ERROR || in_pos >= size_)) break;  // truncated
            if ((z.data_type & 128) && !(z.data_type & 64)) {  // at a block boundary, not in the last block
                const uint64_t p = in_pos * 8 - (uint64_t)(z.data_type & 7);
                while (j < n && (!ch[j].found || ch[j].start < p)) ++j;
                if (j < n && ch[j].start == p) { c.status = OK; c.stop = p; c.arrived = (int)j; break; }
                if (p >= round_end) { c.status = OK; c.stop = p; break; }
            }
        }
        inflateEnd(&z);
        c.n = have;
    }

    // one round: up to `threads` chunks searched, decoded, chained, resolved; their bytes are queued for read()
    // (false: a worker ran out of memory — nothing of this round counts)
    bool round() {
        using namespace pinflate;
        ++rounds_;
        static const bool timing = getenv("FLX_CLI_PINFLATE_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        const uint64_t data_end = (uint64_t)(size_ - 8) * 8;  // the trailer is not deflate data
        const size_t cb = chunk_bytes();
        const uint64_t base_byte = chain_bit_ >> 3;
        const uint64_t left_bytes = (data_end >> 3) > base_byte ? (data_end >> 3) - base_byte : 0;
        const size_t n = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(threads_, (left_bytes + cb - 1) / cb));
        std::vector<Chunk> ch(n);
        const uint64_t round_end = std::min<uint64_t>(data_end, (base_byte + (uint64_t)n * cb) * 8);
        for (size_t i = 0; i < n; ++i) {
            ch[i].begin = i == 0 ? chain_bit_ : std::min<uint64_t>(round_end, (base_byte + (uint64_t)i * cb) * 8);
            ch[i].end = i + 1 < n ? std::min<uint64_t>(round_end, (base_byte + (uint64_t)(i + 1) * cb) * 8) : round_end;
        }
        ch[0].found = true;
        ch[0].start = chain_bit_;
        const size_t limit = std::max<size_t>(cb * 24, (size_t)1 << 20);  // a text compresses 3-5x; beyond 24x zlib does it alone
        bool all_ran = run_parallel(n, threads_, [&](size_t i) {
            Decoder &d = decoders_[i];
            d.in.init(data_, size_);
            if (i == 0) return;
            d.limit = (size_t)1 << 22;
            if (d.out.size() < ((size_t)1 << 16)) d.out.resize((size_t)1 << 16);
            ch[i].found = ch[i].begin < ch[i].end && find_block_start(d, ch[i].begin, ch[i].end, &ch[i].start);
        });
        const double t1 = now();
        if (!all_ran) return false;
        all_ran = run_parallel(n, threads_, [&](size_t i) {
            Chunk &c = ch[i];
            if (!c.found) return;
            Decoder &d = decoders_[i];
            d.limit = limit;
            d.n = 0;
            d.in.seek(c.start);
            size_t j = i + 1;
            size_t scanned = 0, clean_from = 0;  // clean_from: one past the last marker among the symbols scanned
            uint64_t hand_over = 0;
            bool handed = false;
            for (;;) {
                bool fin = false;
                const Status st = d.any_block(&fin, false);
                if (st != OK) { c.status = st; break; }
                if (fin) { c.status = OK; c.final_block = true; c.stop = (d.in.pos() + 7) & ~7ull; break; }
                const uint64_t p = d.in.pos();
                while (j < n && (!ch[j].found || ch[j].start < p)) ++j;  // starts it ran past were not block boundaries
                if (j < n && ch[j].start == p) { c.status = OK; c.stop = p; c.arrived = (int)j; break; }
                if (p >= round_end) { c.status = OK; c.stop = p; break; }
                if (zlib_tails_) {  // once 32 KiB without a marker are behind us, the rest does not depend on the unknown window: zlib's
                    for (const uint16_t *s = d.out.data(); scanned < d.n; ++scanned)
                        if (s[scanned] >= 256) clean_from = scanned + 1;
                    if (d.n - clean_from >= 32768) { handed = true; hand_over = p; break; }
                }
            }
            c.n_sym = c.n = d.n;
            c.sym.swap(d.out);  // (handed back below)
            c.bytes = take_buffer();
            if (handed) {
                zlib_tail(c, hand_over, ch, j, round_end, limit);
            } else if (c.bytes.size() < c.n) {
                c.bytes.resize(c.n);
            }
        });
        const double t2 = now();
        if (!all_ran) return false;
        // the chain from chunk 0; windows one after the other
        std::vector<size_t> chain;
        size_t i = 0;
        size_t broken = (size_t)-1;
        uint64_t out = member_out_;
        std::string window = window_;
        for (;;) {
            Chunk &c = ch[i];
            c.out = out;
            c.window = window;
            if (c.status != OK) { broken = i; break; }
            if (!next_window(c, &window)) { broken = i; break; }
            chain.push_back(i);
            out += c.n;
            if (c.final_block || c.arrived < 0) break;
            i = (size_t)c.arrived;
        }
        for (size_t k = 0; k < n; ++k) dropped_chunks_ += ch[k].found && std::find(chain.begin(), chain.end(), k) == chain.end() && k != broken;
        // markers -> bytes, CRC-32 per chunk
        const double t3 = now();
        all_ran = run_parallel(chain.size(), threads_, [&](size_t k) {
            Chunk &c = ch[chain[k]];
            char *dst = c.bytes.data();
            const uint16_t *s = c.sym.data();
            const size_t wn = c.window.size();
            const unsigned char *w = (const unsigned char *)c.window.data();
            for (size_t q = 0; q < c.n_sym; ++q) {
                const unsigned v = s[q];
                if (v < 256) { dst[q] = (char)v; continue; }
                const size_t back = 32768 - (v - 256);  // bytes in front of the chunk
                if (back > wn) { c.bad_marker = true; dst[q] = 0; continue; }
                dst[q] = (char)w[wn - back];
            }
            uLong crc = crc32(0L, Z_NULL, 0);
            for (size_t at = 0; at < c.n; at += (size_t)1 << 30)
                crc = crc32(crc, (const Bytef *)dst + at, (uInt)std::min<size_t>(c.n - at, (size_t)1 << 30));
            c.crc = crc;
        });
        if (!all_ran) return false;
        if (timing) {
            size_t tail = 0, all = 0;
            for (size_t k : chain) { tail += ch[k].n - ch[k].n_sym; all += ch[k].n; }
            fprintf(stderr, "[pinflate] round %u: search %.3f  decode %.3f  chain %.3f  resolve+crc %.3f s, %zu of %zu chunks chained, %zu bytes (%zu by zlib)\n",
                    rounds_.load(), t1 - t0, t2 - t1, t3 - t2, now() - t3, chain.size(), n, all, tail);
        }
        // the symbol buffers go back to the decoders
        for (size_t k = 0; k < n; ++k)
            if (ch[k].sym.size() > decoders_[k].out.size()) decoders_[k].out.swap(ch[k].sym);
        size_t good = chain.size();
        for (size_t k = 0; k < chain.size(); ++k)
            if (ch[chain[k]].bad_marker) { good = k; break; }  // a distance too far back: zlib will say so
        if (good < chain.size()) {
            broken = chain[good];
            chain.resize(good);
        }
        for (size_t k = 0; k < chain.size(); ++k) {
            Chunk &c = ch[chain[k]];
            crc_ = crc32_combine(crc_, c.crc, (z_off_t)c.n);
            if (member_base_ + c.out > 0) stage_pts_.push_back(point_at(c.start, c.out, c.window));
            parallel_bytes_ += c.n;
            stream_out_ += c.n;
            zlib_tail_bytes_ += c.n - c.n_sym;
            if (c.n > 0) {
                Piece pc;
                pc.bytes = std::move(c.bytes);
                pc.n = c.n;
                stage_q_.push_back(std::move(pc));
            }
        }
        if (broken != (size_t)-1) {  // zlib continues from the start of the chunk that did not work out
            const Chunk &c = ch[broken];
            hand_over(point_at(c.start, c.out, c.window, true));
            return true;
        }
        const Chunk &last = ch[chain.back()];
        member_out_ = out;
        window_ = window;
        chain_bit_ = last.stop;
        if (last.final_block) finish_member(last.stop);
        return true;
    }

    // the 32 KiB behind chunk c from the 32 KiB in front of it, its symbols and its bytes; false: a marker points in front of the data
    static bool next_window(const Chunk &c, std::string *window) {
        const std::string &w = c.window;
        const size_t keep = 32768;
        const size_t n_tail = c.n - c.n_sym;                       // bytes zlib wrote
        const size_t from_tail = std::min(n_tail, keep);
        const size_t from_sym = std::min(c.n_sym, keep - from_tail);
        const size_t from_w = std::min(w.size(), keep - from_tail - from_sym);
        std::string nw;
        nw.reserve(keep);
        nw.append(w, w.size() - from_w, from_w);
        for (size_t q = c.n_sym - from_sym; q < c.n_sym; ++q) {
            const unsigned v = c.sym[q];
            if (v < 256) { nw.push_back((char)v); continue; }
            const size_t back = 32768 - (v - 256);
            if (back > w.size()) return false;
            nw.push_back(w[w.size() - back]);
        }
        nw.append(c.bytes.data() + c.n - from_tail, from_tail);
        window->swap(nw);
        return true;
    }

    void finish_member(uint64_t end_bit) {
        p_finished_ = true;
        const size_t t = (size_t)((end_bit + 7) >> 3);
        if (t + 8 > size_) return;  // the trailer is cut off: to gzread that is the end of the file, everything decoded is delivered
        auto le32 = [&](size_t p) { return (uint32_t)data_[p] | (uint32_t)data_[p + 1] << 8 | (uint32_t)data_[p + 2] << 16 | (uint32_t)data_[p + 3] << 24; };
        if (le32(t) != (uint32_t)crc_ || le32(t + 4) != (uint32_t)member_out_) {  // zlib: "incorrect data check" / "incorrect length check"
            p_error_ = true;
            p_deliverable_ = gzread_delivered_before_error(member_base_, member_out_, false);
            return;
        }
        const size_t next = t + 8;
        if (next + 2 <= size_ && data_[next] == 0x1f && data_[next + 1] == 0x8b) {  // like gzread: members follow each other
            p_finished_ = false;
            begin_member(next);
        }
    }

    const unsigned char *data_ = nullptr;
    size_t size_ = 0;
    unsigned threads_ = 1;
    InflateStream serial_;
    // the reading thread's side
    bool serial_mode_ = true, done_ = false, error_ = false, zlib_tails_ = true;
    uint64_t delivered_ = 0;   // bytes handed to the caller
    uint64_t chain_bit_ = 0;   // where the next round starts
    uint64_t member_out_ = 0;  // bytes of the current member decoded so far
    uint64_t member_base_ = 0; // stream offset of the current member's first byte
    uint64_t stream_out_ = 0;  // bytes decoded so far (delivered or queued)
    bool bgzf_mode_ = false;
    size_t bgzf_at_ = 0;       // the next BGZF block
    uLong crc_ = 0;
    std::string window_;       // the last 32 KiB decoded
    uint64_t last_point_out_ = 0;
    std::deque<Piece> queue_;  // decoded chunks, in order, taken from the producer and not yet read
    std::deque<GzPoint> pending_;
    std::vector<std::vector<char>> pool_;  // byte buffers to use again (both threads)
    std::mutex pool_mutex_;
    // between the two threads
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Piece> shared_q_;
    std::deque<GzPoint> shared_pts_;
    size_t queued_bytes_ = 0, max_ahead_ = (size_t)256 << 20;
    bool finished_ = false, stop_ = false;
    std::thread producer_;
    // the producer's side: what a round made, and how it ended
    std::deque<Piece> stage_q_;
    std::deque<GzPoint> stage_pts_;
    bool p_finished_ = false, p_error_ = false, p_handover_set_ = false;
    uint64_t p_deliverable_ = 0;  // with p_error_: the bytes gzread would have delivered
    GzPoint p_handover_;
    std::vector<pinflate::Decoder> decoders_;
    std::atomic<uint64_t> parallel_bytes_{0}, zlib_tail_bytes_{0};
    std::atomic<unsigned> rounds_{0}, dropped_chunks_{0};
};
