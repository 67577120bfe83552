// What the REFERENCE does with a .gz file, step by step -- so that a damaged, padded or empty file ends the way it
// does there: the same exception class with the same message, or the same matrix.
//
// The reference reads a .gz through gzip.open(fqfile, 'rt') and `for line in fqcon` (tagdigger_fun.py:240-243,
// :250), leaves the loop at maxreads (:272-273), and lets whatever gzip raises propagate.  gzip is CPython's
// Lib/gzip.py (3.10: _GzipReader.read / _read_gzip_header / _read_eof / _read_exact, _PaddedFile) over
// zlib.decompressobj(-15).decompress(buf, max_length) (Modules/zlibmodule.c): restated here call for call on the same
// zlib, because WHICH exception comes out -- EOFError for a stream that ends early, gzip.BadGzipFile for a member that
// fails its check or bytes that are no gzip header, zlib.error for invalid DEFLATE data -- and WHETHER one comes out at
// all depends on how far the reader has got when the loop stops asking:
//   * the text layer (io.TextIOWrapper, newline=None) asks for one chunk of 8192 bytes at a time, and only when the
//     text it holds has no complete line;
//   * a request is one _GzipReader.read(8192): 8192 compressed bytes (what the last call left unconsumed first) go
//     through ONE inflate() with room for 8192 bytes; a call that meets invalid data raises and its output is lost;
//     a member's CRC-32 and ISIZE are checked, zero padding is skipped and the next header is read by the first
//     request AFTER the one that delivered the member's last bytes; a stream that just stops raises EOFError only
//     when a request finds no input at all.
// The fast decoders of this library (fast_inflate.hpp, par_inflate.hpp, gpu_inflate.hpp) only have to notice THAT
// something is wrong with a file; td_count_file / td_gunzip_file then ask gz_verdict() what the reference would
// have done, and where it would have returned a matrix they count through PyGzipReader itself.  Never on the
// path of a good file.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace tdhost {

enum { GZ_OK = 0, GZ_EOF = 1, GZ_BADFILE = 2, GZ_ZLIB = 3 };   // -> EOFError, gzip.BadGzipFile, zlib.error

class PyGzipReader {
  public:
    static constexpr size_t CHUNK = 8192;                      // io.DEFAULT_BUFFER_SIZE: every request and every input piece
    int kind = GZ_OK;
    std::string message;

    PyGzipReader(const uint8_t *data, size_t n) : p_(data), n_(n) {
        memset(&zs_, 0, sizeof(zs_));
        ok_ = inflateInit2(&zs_, -15) == Z_OK;
    }
    ~PyGzipReader() { if (ok_) inflateEnd(&zs_); }
    PyGzipReader(const PyGzipReader &) = delete;
    PyGzipReader &operator=(const PyGzipReader &) = delete;

    // _GzipReader.read(size): 1..size bytes, 0 at the end of the data, -1: raised (kind, message)
    long read(uint8_t *dst, size_t size) {
        if (kind != GZ_OK || !ok_) { if (!ok_ && kind == GZ_OK) raise(GZ_ZLIB, "Error -2 while preparing to decompress data: inconsistent stream state"); return -1; }
        if (size == 0) return 0;
        for (;;) {
            if (eof_) {                                        // the member's trailer, then on to the next member
                if (!read_eof()) return -1;
                new_member_ = true;
                inflateReset2(&zs_, -15);
                eof_ = false;
            }
            if (new_member_) {
                crc_ = 0; stream_size_ = 0;                    // _init_read
                const int r = read_header();
                if (r < 0) return -1;
                if (r == 0) return 0;
                new_member_ = false;
            }
            const size_t k = n_ - cur_ < CHUNK ? n_ - cur_ : CHUNK;
            zs_.next_in = const_cast<Bytef *>(p_ + cur_); zs_.avail_in = (uInt)k;
            cur_ += k;
            zs_.next_out = dst; zs_.avail_out = (uInt)size;
            const int err = inflate(&zs_, Z_SYNC_FLUSH);
            if (err != Z_OK && err != Z_BUF_ERROR && err != Z_STREAM_END) {
                char m[320];
                const char *zmsg = zs_.msg ? zs_.msg : err == Z_DATA_ERROR ? "invalid input data" : err == Z_STREAM_ERROR ? "inconsistent stream state" : nullptr;
                if (zmsg) snprintf(m, sizeof(m), "Error %d while decompressing data: %.200s", err, zmsg);
                else snprintf(m, sizeof(m), "Error %d while decompressing data", err);
                raise(GZ_ZLIB, m);
                return -1;
            }
            cur_ -= zs_.avail_in;                              // (unconsumed_tail / unused_data: prepended to the file again)
            if (err == Z_STREAM_END) eof_ = true;
            const size_t made = size - zs_.avail_out;
            if (made) {
                crc_ = (uint32_t)crc32(crc_, dst, (uInt)made);
                stream_size_ += (uint32_t)made;
                out_total_ += made;
                return (long)made;
            }
            if (k == 0) { raise(GZ_EOF, EOF_MSG); return -1; }
        }
    }
    uint64_t delivered() const { return out_total_; }

  private:
    static constexpr const char *EOF_MSG = "Compressed file ended before the end-of-stream marker was reached";
    const uint8_t *p_; size_t n_, cur_ = 0;
    z_stream zs_;
    bool ok_ = false, eof_ = false, new_member_ = true;
    uint32_t crc_ = 0, stream_size_ = 0;
    uint64_t out_total_ = 0;

    void raise(int k, const std::string &m) { kind = k; message = m; }
    bool read_exact(size_t n, const uint8_t **q) {
        if (n_ - cur_ < n) { cur_ = n_; raise(GZ_EOF, EOF_MSG); return false; }
        *q = p_ + cur_; cur_ += n;
        return true;
    }
    // repr() of a bytes object
    static std::string py_repr(const uint8_t *b, size_t n) {
        bool has_sq = false, has_dq = false;
        for (size_t i = 0; i < n; i++) { has_sq |= b[i] == '\''; has_dq |= b[i] == '"'; }
        const char quote = has_sq && !has_dq ? '"' : '\'';
        std::string s = "b";
        s += quote;
        for (size_t i = 0; i < n; i++) {
            const uint8_t c = b[i];
            char t[8];
            if (c == (uint8_t)quote || c == '\\') { s += '\\'; s += (char)c; }
            else if (c == '\t') s += "\\t";
            else if (c == '\n') s += "\\n";
            else if (c == '\r') s += "\\r";
            else if (c < 0x20 || c >= 0x7f) { snprintf(t, sizeof(t), "\\x%02x", c); s += t; }
            else s += (char)c;
        }
        s += quote;
        return s;
    }
    int read_header() {                                        // _read_gzip_header: 1 a member follows, 0 the file is through
        const size_t have = n_ - cur_ < 2 ? n_ - cur_ : 2;
        if (have == 0) return 0;
        const uint8_t *magic = p_ + cur_;
        cur_ += have;
        if (have != 2 || magic[0] != 0x1f || magic[1] != 0x8b) { raise(GZ_BADFILE, "Not a gzipped file (" + py_repr(magic, have) + ")"); return -1; }
        const uint8_t *q;
        if (!read_exact(8, &q)) return -1;
        const uint32_t method = q[0], flag = q[1];
        if (method != 8) { raise(GZ_BADFILE, "Unknown compression method"); return -1; }
        if (flag & 4) {
            if (!read_exact(2, &q)) return -1;
            const size_t xlen = (size_t)q[0] | ((size_t)q[1] << 8);
            if (!read_exact(xlen, &q)) return -1;
        }
        for (uint32_t f = 8; f <= 16; f <<= 1)                 // FNAME, FCOMMENT: up to a NUL or the end of the file
            if (flag & f) while (cur_ < n_) { if (p_[cur_++] == 0) break; }
        if (flag & 2) { if (!read_exact(2, &q)) return -1; }
        return 1;
    }
    bool read_eof() {                                          // _read_eof
        const uint8_t *q;
        if (!read_exact(8, &q)) return false;
        uint32_t want_crc, isize;
        memcpy(&want_crc, q, 4); memcpy(&isize, q + 4, 4);
        if (want_crc != crc_) {
            char m[96];
            snprintf(m, sizeof(m), "CRC check failed 0x%x != 0x%x", want_crc, crc_);
            raise(GZ_BADFILE, m);
            return false;
        }
        if (isize != stream_size_) { raise(GZ_BADFILE, "Incorrect length of data produced"); return false; }
        while (cur_ < n_ && p_[cur_] == 0) cur_++;              // zero padding
        return true;
    }
};

// What `for line in gzip.open(path, 'rt')` left at read number max_reads (reference :250, :272-273) meets in this
// file: kind GZ_OK -- the loop ends (at the bound or at the end of the file) without an exception -- or the
// exception's class and message.
struct GzVerdict { int kind = GZ_OK; std::string message; };
// `need`: how many complete lines the loop takes out of the text layer before it breaks (~0: it runs to the end of the file)
inline GzVerdict gz_verdict_lines(const uint8_t *data, size_t n, uint64_t need) {
    GzVerdict v;
    PyGzipReader r(data, n);
    uint8_t buf[PyGzipReader::CHUNK];
    uint64_t lines = 0;
    bool pending_cr = false;           // IncrementalNewlineDecoder: a '\r' that ends a chunk waits for the next one
    while (lines < need) {
        const long got = r.read(buf, sizeof(buf));
        if (got < 0) { v.kind = r.kind; v.message = r.message; return v; }
        if (got == 0) break;
        long i = 0;
        if (pending_cr) { lines++; pending_cr = false; if (buf[0] == '\n') i = 1; }
        for (; i < got; i++) {
            const uint8_t c = buf[i];
            if (c == '\n') lines++;
            else if (c == '\r') {
                if (i + 1 == got) pending_cr = true;
                else { lines++; if (buf[i + 1] == '\n') i++; }
            }
        }
    }
    return v;
}

// find_tags_fastq's loop breaks inside the body for line 4 (r - 1) + 1, the sequence line of read number r = max_reads
inline GzVerdict gz_verdict(const uint8_t *data, size_t n, uint64_t max_reads) {
    if (max_reads == 0) max_reads = 1;
    return gz_verdict_lines(data, n, max_reads >= (1ull << 60) ? ~0ull : 4 * (max_reads - 1) + 2);
}

}  // namespace tdhost
