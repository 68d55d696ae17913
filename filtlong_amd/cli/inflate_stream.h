// inflate_stream.h — sequential inflate of a mapped gzip file that leaves ACCESS POINTS behind, and inflate from such a point.
//
// The reference reads its (usually gzip-compressed) input twice through zlib's gzread, one thread each time
// (src/main.cpp:70 and :263 via src/kseq.h:87-110).  A deflate stream can only be entered at a block boundary, and only with the
// 32 KiB of output that precede it; pass 1 here notes such boundaries (compressed bit position, uncompressed offset, window)
// every `span` bytes of output, so that the output pass can inflate the pieces between them on several threads.  The
// technique is the one of zlib's examples/zran.c (inflate with Z_BLOCK, inflatePrime + inflateSetDictionary to resume);
// concatenated gzip members are followed as gzread does.  Plain (uncompressed) data is served by the same interface
// with points at multiples of `span`, for tests that force this path.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

struct GzPoint {
    uint64_t in = 0;    // compressed offset of the first whole byte of the block
    uint64_t out = 0;   // uncompressed offset of the block's first byte
    int bits = 0;       // > 0: the block starts `bits` bits before `in`, in the top bits of byte in - 1
    bool raw = false;   // false: a gzip header starts at `in` (the beginning of the file)
    std::string window; // the up to 32 KiB of output before `out`
    // raw points inside a member whose trailer is still to be CHECKED by whoever inflates on from here (a hand-over from the
    // block-parallel decoder): the member's first byte is stream offset `member_base`, and `crc` is the CRC-32 of its bytes
    // [member_base, out).  Access points of the output pass re-read bytes pass 1 has validated: check = false.
    bool check = false;
    uint64_t member_base = 0;
    unsigned long crc = 0;
};

// What zlib's gzread hands to kseq when a gzip stream is DAMAGED (the reference: src/kseq.h:71-76,98-108 over gzread, 16 KiB per
// call).  gzread inflates 16 KiB at a time (gz_decomp: straight into the caller's buffer, or into its own 16 KiB buffer after a
// member boundary) and a data error makes the whole CALL fail: the bytes of that call are lost and kseq enters its error state
// at the last call boundary.  `member_base`: stream offset of the failing member's first byte, `produced`: bytes of that member
// inflate had written when it reported the error, `late`: the error is one inflate only notices when it WRITES ("invalid distance
// too far back" is checked behind the test for a full buffer; every other error while the next symbol is parsed, buffer full or
// not).  Returns the number of bytes kseq gets before its error state.  A TRUNCATED stream is no error to gzread: everything
// decodable is delivered, then the end of the file.
static inline uint64_t gzread_delivered_before_error(uint64_t member_base, uint64_t produced, bool late) {
    const uint64_t call = 16384;
    uint64_t j = produced == 0 ? 0 : (produced - 1) / call;       // the gz_decomp call of this member that fails
    if (late && produced > 0 && produced % call == 0) j = produced / call;
    return (member_base + j * call) / call * call;                // the gzread call that needed it starts here
}

class InflateStream {
public:
    InflateStream() = default;
    InflateStream(const InflateStream &) = delete;
    InflateStream &operator=(const InflateStream &) = delete;
    ~InflateStream() { close(); }

    // the whole compressed file, mapped; gz = it starts with the gzip magic
    bool open(const unsigned char *data, size_t size, bool gz) {
        GzPoint start;
        return open_at(data, size, gz, start);
    }
    bool open_at(const unsigned char *data, size_t size, bool gz, const GzPoint &pt) {
        close();
        data_ = data; size_ = size; gz_ = gz;
        in_pos_ = pt.in; total_out_ = pt.out; last_point_out_ = pt.out;
        eof_ = false; error_ = false; truncated_ = false; deliverable_ = 0;
        member_start_ = pt.raw && pt.check ? pt.member_base : pt.out;
        check_ = pt.raw && pt.check;
        crc_ = pt.raw && pt.check ? pt.crc : crc32(0L, Z_NULL, 0);
        if (!gz_) { eof_ = in_pos_ >= size_; return true; }
        memset(&z_, 0, sizeof z_);
        raw_ = pt.raw;
        if (inflateInit2(&z_, raw_ ? -15 : 47) != Z_OK) { error_ = true; return false; }
        live_ = true;
        if (raw_) {
            if (pt.bits > 0) {
                if (pt.in == 0 || inflatePrime(&z_, pt.bits, data_[pt.in - 1] >> (8 - pt.bits)) != Z_OK) { error_ = true; return false; }
            }
            if (!pt.window.empty() &&
                inflateSetDictionary(&z_, (const Bytef *)pt.window.data(), (uInt)pt.window.size()) != Z_OK) { error_ = true; return false; }
        }
        return true;
    }
    void close() {
        if (live_) inflateEnd(&z_);
        live_ = false;
    }
    bool eof() const { return eof_; }
    bool error() const { return error_; }          // a data error (not a truncation: that is the end of the file, like gzread's)
    bool truncated() const { return truncated_; }  // the compressed data ended early; everything decodable has been produced
    uint64_t deliverable() const { return deliverable_; }  // after error(): the bytes gzread would have handed out (see above)
    uint64_t total_out() const { return total_out_; }

    // Produces up to `cap` bytes at dst; returns how many (less than cap only at the end of the data or on error).
    // points != nullptr: the caller keeps the previous min(32 KiB, total_out) bytes of output directly in front of dst + n
    // for every n (one contiguous buffer); block boundaries at least `span` output bytes apart are appended to *points.
    size_t read(char *dst, size_t cap, std::vector<GzPoint> *points = nullptr, uint64_t span = 0) {
        size_t produced = 0;
        while (produced < cap && !eof_ && !error_) {
            if (!gz_) {
                uint64_t want = std::min<uint64_t>(cap - produced, size_ - in_pos_);
                if (points) want = std::min<uint64_t>(want, last_point_out_ + span - total_out_);  // stop at the next multiple
                memcpy(dst + produced, data_ + in_pos_, (size_t)want);
                produced += (size_t)want; in_pos_ += want; total_out_ += want;
                if (in_pos_ >= size_) { eof_ = true; break; }
                if (points && total_out_ - last_point_out_ >= span) {
                    GzPoint p;
                    p.in = in_pos_; p.out = total_out_; p.raw = true;
                    points->push_back(std::move(p));
                    last_point_out_ = total_out_;
                }
                continue;
            }
            const size_t out_room = std::min<size_t>(cap - produced, (size_t)1 << 30);
            const size_t in_room = (size_t)std::min<uint64_t>(size_ - in_pos_, (uint64_t)1 << 30);
            z_.next_out = (Bytef *)dst + produced; z_.avail_out = (uInt)out_room;
            z_.next_in = (Bytef *)data_ + in_pos_; z_.avail_in = (uInt)in_room;
            const int ret = inflate(&z_, points ? Z_BLOCK : Z_NO_FLUSH);
            const size_t got = out_room - z_.avail_out;
            if (check_ && got > 0) crc_ = crc32(crc_, (const Bytef *)dst + produced, (uInt)got);
            produced += got; total_out_ += got;
            in_pos_ += in_room - z_.avail_in;
            if (ret == Z_STREAM_END) {
                if (raw_) {  // raw inflate leaves the member's crc32 + isize: checked here when the point asks for it
                    if (check_) {
                        if (in_pos_ + 8 > size_) { truncated_ = true; eof_ = true; break; }  // (gzread: unexpected end of file)
                        auto le32 = [&](uint64_t p) { return (uint32_t)data_[p] | (uint32_t)data_[p + 1] << 8 | (uint32_t)data_[p + 2] << 16 | (uint32_t)data_[p + 3] << 24; };
                        if (le32(in_pos_) != (uint32_t)crc_ || le32(in_pos_ + 4) != (uint32_t)(total_out_ - member_start_)) { fail(false); break; }
                    }
                    in_pos_ = std::min<uint64_t>(size_, in_pos_ + 8);
                }
                if (in_pos_ + 2 <= size_ && data_[in_pos_] == 0x1f && data_[in_pos_ + 1] == 0x8b) {  // next member, like gzread
                    if (inflateReset2(&z_, 47) != Z_OK) { fail(false); break; }
                    raw_ = false; check_ = false;
                    member_start_ = total_out_;
                    continue;
                }
                eof_ = true;
                break;
            }
            if (ret != Z_OK && ret != Z_BUF_ERROR) { fail(z_.msg && strstr(z_.msg, "too far back") != nullptr); break; }
            if (got == 0 && in_room == z_.avail_in && (ret == Z_BUF_ERROR || in_pos_ >= size_)) { truncated_ = true; eof_ = true; break; }
            if (points && (z_.data_type & 128) && !(z_.data_type & 64) && total_out_ - last_point_out_ >= span) {
                GzPoint p;
                p.in = in_pos_; p.out = total_out_; p.bits = z_.data_type & 7; p.raw = true;
                const size_t w = (size_t)std::min<uint64_t>(32768, total_out_);
                p.window.assign(dst + produced - w, w);
                points->push_back(std::move(p));
                last_point_out_ = total_out_;
            }
        }
        return produced;
    }

private:
    void fail(bool late) {
        error_ = true;
        deliverable_ = gzread_delivered_before_error(member_start_, total_out_ - member_start_, late);
    }
    const unsigned char *data_ = nullptr;
    size_t size_ = 0;
    bool gz_ = false, raw_ = false, live_ = false, eof_ = true, error_ = false, truncated_ = false, check_ = false;
    uint64_t in_pos_ = 0, total_out_ = 0, last_point_out_ = 0;
    uint64_t member_start_ = 0;   // stream offset of the current member's first byte
    uint64_t deliverable_ = 0;
    unsigned long crc_ = 0;        // running CRC-32 of the current member (only kept when a hand-over point asked for the check)
    z_stream z_;
};
