// A DEFLATE / gzip decoder for the host side of libtagdig (no GPU code), written for throughput on
// FASTQ: 64-bit bit buffer refilled eight bytes at a time, two-level Huffman tables looked up with
// the stream's low bits, matches copied eight bytes at a time.  RFC 1951 / RFC 1952: stored, fixed
// and dynamic blocks; FEXTRA / FNAME / FCOMMENT / FHCRC headers; any number of members; the CRC-32
// and length of every member are checked.  The whole compressed file is mapped into memory (the
// caller guarantees FastInflate::PAD readable bytes behind it); output is produced on demand into
// the caller's buffers.
#pragma once
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

namespace tdhost {

class FastInflate {
  public:
    static constexpr size_t PAD = 64;          // readable bytes the caller guarantees behind the input

    void open(const uint8_t *data, size_t n) {
        in_ = in_begin_ = data; in_end_ = data + n;
        bitbuf_ = 0; bitcnt_ = 0;
        state_ = S_MEMBER; failed_ = false;
        if (buf_.size() != HIST + CAP + SLACK) buf_.assign(HIST + CAP + SLACK, 0);     // (kept across open() calls)
        obuf_ = buf_.data();
        wpos_ = rpos_ = crc_pos_ = HIST;
        crc_ = 0; isize_ = 0;
    }
    const char *error() const { return err_; }

    // up to `want` decompressed bytes into dst; 0 at the end of the stream, < 0 on error
    long read(uint8_t *dst, size_t want) {
        size_t done = 0;
        while (done < want) {
            if (rpos_ == wpos_) {
                if (failed_) return -1;
                if (state_ == S_END) break;
                if (wpos_ >= HIST + CAP) next_staging();       // full and drained
                const size_t before = wpos_;
                if (!produce()) { failed_ = true; if (rpos_ == wpos_) return -1; }
                isize_ += (uint32_t)(wpos_ - before);
                if (member_done_) { if (!finish_member()) { failed_ = true; return -1; } }
                continue;
            }
            const size_t n = std::min(want - done, wpos_ - rpos_);
            memcpy(dst + done, obuf_ + rpos_, n);
            rpos_ += n; done += n;
        }
        return (long)done;
    }

    // ---- pieces the chunk-parallel decoder (par_inflate.hpp) shares
    // Huffman table entry: bits 0..7 code length (total bits to drop), 8..15 extra bits (or, for a
    // pointer, the subtable's index width), 16..30 value (literal, base length, base distance or
    // subtable start), flags in the top bits.
    // literal/length table: F_LIT, F_SUB, F_EOB above a 13-bit value; distance table: D_SUB above a 15-bit value.
    // F_LIT2: the primary index decodes TWO literals at once (first in bits 16..23, second in bits 8..15,
    // both code lengths summed in bits 0..7) -- FASTQ bases have 2-3 bit codes
    static constexpr uint32_t F_LIT = 1u << 31, F_SUB = 1u << 30, F_EOB = 1u << 29, F_LIT2 = 1u << 28, D_SUB = 1u << 31;
    static constexpr int LTB = 11, DTB = 8;
    static constexpr size_t LIT_ENTRIES = (1 << LTB) + 288 * 16, DIST_ENTRIES = (1 << DTB) + 32 * 128;
    static const uint16_t *length_base() {
        static const uint16_t t[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        return t;
    }
    static const uint8_t *length_extra() {
        static const uint8_t t[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        return t;
    }
    static const uint16_t *dist_base() {
        static const uint16_t t[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        return t;
    }
    static const uint8_t *dist_extra() {
        static const uint8_t t[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        return t;
    }

  private:
    // ---- staging: [HIST bytes of history | CAP bytes being produced | SLACK for over-long copies]
    static constexpr size_t HIST = 32768, CAP = 1 << 20, SLACK = 320;
    std::vector<uint8_t> buf_;
    uint8_t *obuf_ = nullptr;
    size_t wpos_ = 0, rpos_ = 0;
    size_t crc_pos_ = 0;                       // staging offset up to which this member's CRC has been taken
    size_t valid_from_ = 0;                    // staging offset of the oldest byte a match may copy from (this member's output)
    // ---- input
    const uint8_t *in_ = nullptr, *in_end_ = nullptr, *in_begin_ = nullptr;
    uint64_t bitbuf_ = 0;
    uint32_t bitcnt_ = 0;
    // ---- state
    enum { S_MEMBER, S_BLOCK, S_STORED, S_HUFF, S_END } state_ = S_MEMBER;
    bool final_ = false, member_done_ = false, failed_ = false;
    uint32_t stored_left_ = 0;
    uint32_t crc_ = 0, isize_ = 0;
    const char *err_ = "";
    // ---- Huffman tables of the current block
    uint32_t lit_[LIT_ENTRIES];
    uint32_t dist_[DIST_ENTRIES];

    bool fail(const char *m) { err_ = m; return false; }

    // the running CRC-32 of the current member up to staging offset wpos_
    uint32_t crc_now() {
        crc_ = crc32_update(crc_, obuf_ + crc_pos_, wpos_ - crc_pos_);
        crc_pos_ = wpos_;
        return crc_;
    }
    // the staging buffer is full and has been read out: keep the last 32 KiB as history
    void next_staging() {
        crc_now();
        memmove(obuf_, obuf_ + wpos_ - HIST, HIST);
        const size_t shift = wpos_ - HIST;
        valid_from_ = valid_from_ > shift ? valid_from_ - shift : 0;
        wpos_ = rpos_ = crc_pos_ = HIST;
    }

    // ---- bit buffer: the stream's next bits in the low end of a 64-bit word
    static uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }     // (little-endian hosts only)
    void refill() {                                               // afterwards 56..63 bits are available
        bitbuf_ |= load64(in_) << bitcnt_;
        in_ += (63 - bitcnt_) >> 3;
        bitcnt_ |= 56;
    }
    bool overrun() const { return in_ > in_end_ + 16; }           // read well into the padding: the stream is truncated
    uint32_t bits(uint32_t n) {                                   // n <= 32, after a refill
        const uint32_t v = (uint32_t)(bitbuf_ & ((1ull << n) - 1));
        bitbuf_ >>= n; bitcnt_ -= n;
        return v;
    }
    void byte_align() {                                           // give whole unread bytes back to the input
        in_ -= bitcnt_ >> 3;
        bitbuf_ = 0; bitcnt_ = 0;
    }

    // ---- CRC-32 (IEEE, reflected) by carry-less multiplication where the CPU has it (about 6 GB/s
    // against zlib's 1 GB/s table walk): four 128-bit lanes folded 512 bits at a time, folded into
    // one, reduced to 64 and then 32 bits (Barrett).  Constants: x^(512+-32), x^(128+-32), x^64 mod P,
    // P and its inverse, bit-reflected.  len >= 64 and a multiple of 16; state without the final inversion.
    __attribute__((target("pclmul,sse4.1")))
    static uint32_t crc32_clmul(uint32_t crc, const uint8_t *buf, size_t len) {
        const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
        const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
        const __m128i k5k0 = _mm_set_epi64x(0x0000000000, 0x0163cd6124);
        const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
        __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
        x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
        x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
        x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
        x0 = k1k2;
        buf += 64; len -= 64;
        while (len >= 64) {
            x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
            x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
            x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
            x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
            y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
            y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
            x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
            x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
            buf += 64; len -= 64;
        }
        x0 = k3k4;
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
        while (len >= 16) {
            x2 = _mm_loadu_si128((const __m128i *)buf);
            x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
            buf += 16; len -= 16;
        }
        x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
        x3 = _mm_setr_epi32(~0, 0, ~0, 0);
        x1 = _mm_srli_si128(x1, 8);
        x1 = _mm_xor_si128(x1, x2);
        x0 = k5k0;
        x2 = _mm_srli_si128(x1, 4);
        x1 = _mm_and_si128(x1, x3);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_xor_si128(x1, x2);
        x0 = poly;
        x2 = _mm_and_si128(x1, x3);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
        x2 = _mm_and_si128(x2, x3);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x1 = _mm_xor_si128(x1, x2);
        return (uint32_t)_mm_extract_epi32(x1, 1);
    }
  public:
    // CRC-32 of A || B from CRC(A), CRC(B) and B's length: CRC(A) * x^(8 len) mod P, plus CRC(B) -- the
    // power by square and multiply over a table of x^(2^k) mod P (bit-reflected polynomials, x^0 = bit 31).
    // (zlib 1.2.11's crc32_combine squares 32 x 32 bit matrices per call: tens of microseconds.)
    static uint32_t poly_mul(uint32_t a, uint32_t b) {
        uint32_t p = 0;
        for (uint32_t m = 1u << 31; m; m >>= 1) {
            if (a & m) p ^= b;
            b = b & 1 ? (b >> 1) ^ 0xEDB88320u : b >> 1;
        }
        return p;
    }
    static uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
        struct Powers { uint32_t t[64]; Powers() { uint32_t p = 1u << 30; t[0] = p; for (int k = 1; k < 64; k++) t[k] = p = poly_mul(p, p); } };
        static const Powers pw;                                   // t[k] = x^(2^k) mod P
        // crc_a * x^(8 len_b): one multiplication by x^(8 * 2^k) for every bit k of len_b, each a constant -- linear in crc_a, so
        // four look-ups in tables of 256 entries, made when a k is first met.  (A device-decoded segment of 1 GiB joins 86 000 blocks
        // of 64 KiB and 8 000 of odd lengths: with bit-by-bit multiplications that was 19 ms of host time a segment, nothing running.)
        struct Shift { std::once_flag once; uint32_t t[4][256]; };
        static Shift sh[61];
        uint32_t v = crc_a;
        uint64_t n = len_b;
        for (int k = 0; n && k < 61; n >>= 1, k++) {
            if (!(n & 1)) continue;
            Shift &s = sh[k];
            std::call_once(s.once, [&s, k]() {
                const uint32_t f = pw.t[k + 3];
                for (int j = 0; j < 4; j++) for (uint32_t b = 0; b < 256; b++) s.t[j][b] = poly_mul(f, b << (8 * j));
            });
            v = s.t[0][v & 255u] ^ s.t[1][(v >> 8) & 255u] ^ s.t[2][(v >> 16) & 255u] ^ s.t[3][v >> 24];
        }
        return v ^ crc_b;
    }
    // zlib's convention (inverted going in and coming out)
    static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
        static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
        if (clmul && n >= 64) {
            const size_t m = n & ~(size_t)15;
            crc = ~crc32_clmul(~crc, p, m);
            p += m; n -= m;
        }
        while (n) {
            const uInt k = (uInt)std::min<size_t>(n, 1u << 30);
            crc = (uint32_t)::crc32(crc, p, k);
            p += k; n -= k;
        }
        return crc;
    }

  private:

    // ---- gzip framing
  public:
    // the member header at p (first: the file's first member): 1 and *body (the first byte of the deflate stream); 0 where
    // the file ends -- at its end, behind the zero bytes gzip.open skips after a member; -1 and *err for anything else (the
    // caller refuses the file, and gz_pyrules.hpp says how the reference ends on it: EOFError for a header cut short,
    // gzip.BadGzipFile for bytes that are no header).  Reserved flag bits are ignored, as Lib/gzip.py ignores them.
    static int parse_member_header(const uint8_t *p, const uint8_t *end, const uint8_t **body, const char **err, bool first = false) {
        if (!first) while (p < end && *p == 0) p++;
        if (p >= end) return 0;
        if (end - p < 10) { *err = "truncated gzip header (or bytes that are none) behind a member"; return -1; }
        if (p[0] != 0x1f || p[1] != 0x8b) { *err = "bytes that are no gzip header behind a member"; return -1; }
        if (p[2] != 8) { *err = "unknown compression method in a gzip header"; return -1; }
        const uint32_t flg = p[3];
        p += 10;
        if (flg & 4) {
            if (end - p < 2) { *err = "truncated gzip header"; return -1; }
            const size_t xlen = (size_t)p[0] | ((size_t)p[1] << 8);
            p += 2;
            if ((size_t)(end - p) < xlen) { *err = "truncated gzip header"; return -1; }
            p += xlen;
        }
        if (flg & 8) { while (p < end && *p) p++; if (p < end) p++; }
        if (flg & 16) { while (p < end && *p) p++; if (p < end) p++; }
        if (flg & 2) { if (end - p < 2) { *err = "truncated gzip header"; return -1; } p += 2; }
        if (p >= end) { *err = "truncated gzip header"; return -1; }
        *body = p;
        return 1;
    }

  private:
    bool member_header() {
        byte_align();
        const uint8_t *body = nullptr;
        const int r = parse_member_header(in_, in_end_, &body, &err_, in_ == in_begin_);
        if (r < 0) return false;
        if (r == 0) { state_ = S_END; return true; }
        in_ = body;
        crc_ = 0; isize_ = 0; member_done_ = false;
        crc_pos_ = wpos_;
        valid_from_ = wpos_;                                         // (a member cannot reach into the one before it)
        state_ = S_BLOCK;
        return true;
    }
    bool finish_member() {
        byte_align();
        if (in_ + 8 > in_end_) return fail("truncated gzip member (no CRC / length)");
        uint32_t want_crc, want_len;
        memcpy(&want_crc, in_, 4); memcpy(&want_len, in_ + 4, 4);
        in_ += 8;
        if (want_crc != crc_now()) return fail("gzip member fails its CRC-32 check");
        if (want_len != isize_) return fail("gzip member fails its length check");
        member_done_ = false;
        state_ = S_MEMBER;
        return true;
    }

  public:
    // ---- table construction from code lengths (canonical Huffman, RFC 1951 3.2.2); `complete`
    // (optional) reports whether the code uses its whole code space
    template <typename Make>
    static bool build(uint32_t *table, int tb, const uint8_t *lens, int nsym, uint32_t sub_flag, Make &&make,
                      bool *complete = nullptr) {
        int count[16] = {0};
        for (int s = 0; s < nsym; s++) count[lens[s]]++;
        count[0] = 0;
        int maxlen = 0;
        uint32_t code = 0, next[16];
        long left = 1;
        for (int l = 1; l <= 15; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;                            // over-subscribed
            code = (code + count[l - 1]) << 1;
            next[l] = code;
            if (count[l]) maxlen = l;
        }
        if (complete) *complete = left == 0;
        const uint32_t psize = 1u << tb;
        for (uint32_t i = 0; i < psize; i++) table[i] = 0;          // 0 = invalid code
        uint32_t used = psize;
        const int sb = maxlen > tb ? maxlen - tb : 0;               // every subtable is indexed by sb bits
        for (int s = 0; s < nsym; s++) {
            const int l = lens[s];
            if (!l) continue;
            uint32_t c = next[l]++, rev = 0;
            for (int k = 0; k < l; k++) { rev = (rev << 1) | (c & 1); c >>= 1; }
            const uint32_t e = make(s, l);
            if (l <= tb) {
                for (uint32_t i = rev; i < psize; i += 1u << l) table[i] = e;
            } else {
                const uint32_t prefix = rev & (psize - 1);
                if (!(table[prefix] & sub_flag)) {
                    table[prefix] = sub_flag | ((used & 0x1FFFu) << 16) | ((uint32_t)sb << 8) | (uint32_t)tb;
                    for (uint32_t i = 0; i < (1u << sb); i++) table[used + i] = 0;
                    used += 1u << sb;
                }
                const uint32_t base = (table[prefix] >> 16) & 0x1FFFu;
                for (uint32_t i = rev >> tb; i < (1u << sb); i += 1u << (l - tb)) table[base + i] = e;
            }
        }
        return true;
    }
    // both tables of a block from its code lengths; false: over-subscribed.  pair_literals: see F_LIT2
    static bool build_block_tables(uint32_t *lit, uint32_t *dist, const uint8_t *ll, int nlit, const uint8_t *dl, int ndist,
                                   bool pair_literals, bool *lit_complete = nullptr, bool *dist_complete = nullptr) {
        const uint16_t *lbase = length_base(), *dbase = dist_base();
        const uint8_t *lext = length_extra(), *dext = dist_extra();
        if (!build(lit, LTB, ll, nlit, F_SUB, [&](int s, int l) -> uint32_t {
                if (s < 256) return F_LIT | ((uint32_t)s << 16) | (uint32_t)l;
                if (s == 256) return F_EOB | (uint32_t)l;
                if (s > 285) return 0;                              // (never valid in a stream)
                return ((uint32_t)lbase[s - 257] << 16) | ((uint32_t)lext[s - 257] << 8) | (uint32_t)l;
            }, lit_complete)) return false;
        if (!build(dist, DTB, dl, ndist, D_SUB, [&](int s, int l) -> uint32_t {
                if (s > 29) return 0;
                return ((uint32_t)dbase[s] << 16) | ((uint32_t)dext[s] << 8) | (uint32_t)l;
            }, dist_complete)) return false;
        if (!pair_literals) return true;
        // pair up literals: where the bits left in a primary index after one literal decode a second
        // literal completely, the entry yields both
        uint32_t single[1u << LTB];
        memcpy(single, lit, sizeof(single));
        for (uint32_t i = 0; i < (1u << LTB); i++) {
            const uint32_t e = single[i];
            if (!(e & F_LIT)) continue;
            const uint32_t l1 = e & 0xFF;
            const uint32_t e2 = single[i >> l1];          // (index: the stream's next LTB - l1 bits, zeros above them)
            const uint32_t l2 = e2 & 0xFF;
            if ((e2 & F_LIT) && l1 + l2 <= (uint32_t)LTB)  // decided by those bits alone
                lit[i] = F_LIT | F_LIT2 | (e & 0x00FF0000u) | (((e2 >> 16) & 0xFFu) << 8) | (l1 + l2);
        }
        return true;
    }

  private:
    bool build_tables(const uint8_t *ll, int nlit, const uint8_t *dl, int ndist) {
        if (!build_block_tables(lit_, dist_, ll, nlit, dl, ndist, true)) return fail("over-subscribed Huffman code");
        return true;
    }

    bool block_header() {
        if (overrun()) return fail("truncated deflate stream");
        refill();
        final_ = bits(1) != 0;
        const uint32_t type = bits(2);
        if (type == 0) {
            byte_align();
            if (in_ + 4 > in_end_) return fail("truncated stored block");
            const uint32_t len = in_[0] | (in_[1] << 8), nlen = in_[2] | (in_[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) return fail("stored block length check failed");
            in_ += 4;
            stored_left_ = len;
            state_ = S_STORED;
            return true;
        }
        if (type == 1) {
            uint8_t ll[288], dl[30];
            for (int i = 0; i < 144; i++) ll[i] = 8;
            for (int i = 144; i < 256; i++) ll[i] = 9;
            for (int i = 256; i < 280; i++) ll[i] = 7;
            for (int i = 280; i < 288; i++) ll[i] = 8;
            for (int i = 0; i < 30; i++) dl[i] = 5;
            if (!build_tables(ll, 288, dl, 30)) return false;
            state_ = S_HUFF;
            return true;
        }
        if (type == 3) return fail("reserved deflate block type");
        const uint32_t nlit = bits(5) + 257, ndist = bits(5) + 1, nclen = bits(4) + 4;
        if (nlit > 286 || ndist > 30) return fail("too many length or distance codes");
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        refill();
        for (uint32_t i = 0; i < nclen; i++) { if (bitcnt_ < 8) refill(); cl[order[i]] = (uint8_t)bits(3); }
        uint32_t cltab[128 + 19 * 1];
        if (!build(cltab, 7, cl, 19, F_SUB, [&](int s, int l) -> uint32_t { return ((uint32_t)s << 16) | (uint32_t)l; }))
            return fail("over-subscribed code-length code");
        uint8_t lens[286 + 30];
        uint32_t i = 0;
        while (i < nlit + ndist) {
            refill();
            if (overrun()) return fail("truncated deflate stream");
            const uint32_t e = cltab[bitbuf_ & 127];
            if (!(e & 0xFF)) return fail("invalid code-length code");
            bits(e & 0xFF);
            const uint32_t s = e >> 16;
            if (s < 16) { lens[i++] = (uint8_t)s; continue; }
            uint32_t rep, val = 0;
            if (s == 16) { if (i == 0) return fail("repeat with no previous length"); val = lens[i - 1]; rep = 3 + bits(2); }
            else if (s == 17) rep = 3 + bits(3);
            else rep = 11 + bits(7);
            if (i + rep > nlit + ndist) return fail("code lengths run past the end");
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) return fail("no end-of-block code");
        if (!build_tables(lens, (int)nlit, lens + nlit, (int)ndist)) return false;
        state_ = S_HUFF;
        return true;
    }

    // decodes until the staging buffer is (nearly) full, the block ends or the member ends
    bool huffman() {
        uint8_t *out = obuf_ + wpos_;
        uint8_t *const out_stop = obuf_ + HIST + CAP;     // (SLACK bytes follow)
        const uint8_t *const begin = obuf_ + valid_from_;
        const uint64_t lmask = (1u << LTB) - 1, dmask = (1u << DTB) - 1;
        bool ok = true;
        for (;;) {
            if (out >= out_stop) break;                           // staging full: come back after a flush
            if (in_ > in_end_) { if (overrun()) { ok = fail("truncated deflate stream"); break; } }
            refill();
            uint32_t e = lit_[bitbuf_ & lmask];
            if (e & F_SUB) e = lit_[((e >> 16) & 0x1FFFu) + (uint32_t)((bitbuf_ >> LTB) & ((1u << ((e >> 8) & 0xFF)) - 1))];
            if (e & F_LIT) {
                // literals: one or two per lookup, up to three lookups on what the refill left
                // (3 x 11 bits at most after the first, which may have come through a subtable: 15)
                bitbuf_ >>= (e & 0xFF); bitcnt_ -= (e & 0xFF);
                out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 8);
                out += 1 + ((e >> 28) & 1u);
                uint32_t e2 = lit_[bitbuf_ & lmask];
                if (e2 & F_LIT) {
                    bitbuf_ >>= (e2 & 0xFF); bitcnt_ -= (e2 & 0xFF);
                    out[0] = (uint8_t)(e2 >> 16); out[1] = (uint8_t)(e2 >> 8);
                    out += 1 + ((e2 >> 28) & 1u);
                    e2 = lit_[bitbuf_ & lmask];
                    if (e2 & F_LIT) {
                        bitbuf_ >>= (e2 & 0xFF); bitcnt_ -= (e2 & 0xFF);
                        out[0] = (uint8_t)(e2 >> 16); out[1] = (uint8_t)(e2 >> 8);
                        out += 1 + ((e2 >> 28) & 1u);
                    }
                }
                continue;
            }
            const uint32_t clen = e & 0xFF;
            if (clen == 0) { ok = fail("invalid literal/length code"); break; }
            bitbuf_ >>= clen; bitcnt_ -= clen;
            if (e & F_EOB) {
                state_ = S_BLOCK;
                if (final_) member_done_ = true;
                break;
            }
            const uint32_t lext = (e >> 8) & 0xFF;
            uint32_t length = ((e >> 16) & 0x1FF) + (uint32_t)(bitbuf_ & ((1u << lext) - 1));
            bitbuf_ >>= lext; bitcnt_ -= lext;
            if (bitcnt_ < 32) refill();
            uint32_t d = dist_[bitbuf_ & dmask];
            if (d & D_SUB) d = dist_[((d >> 16) & 0x1FFFu) + (uint32_t)((bitbuf_ >> DTB) & ((1u << ((d >> 8) & 0xFF)) - 1))];
            const uint32_t dlen = d & 0xFF;
            if (dlen == 0) { ok = fail("invalid distance code"); break; }
            bitbuf_ >>= dlen; bitcnt_ -= dlen;
            const uint32_t dext = (d >> 8) & 0xFF;
            const uint32_t distance = ((d >> 16) & 0x7FFF) + (uint32_t)(bitbuf_ & ((1u << dext) - 1));
            bitbuf_ >>= dext; bitcnt_ -= dext;
            if (distance > (size_t)(out - begin)) { ok = fail("distance reaches before the start of the output"); break; }
            const uint8_t *src = out - distance;
            uint8_t *const end = out + length;
            if (distance >= 16) {
                do { memcpy(out, src, 16); out += 16; src += 16; } while (out < end);
            } else if (distance >= 8) {
                do { memcpy(out, src, 8); out += 8; src += 8; } while (out < end);
            } else if (distance == 1) {
                const uint64_t v = 0x0101010101010101ull * *src;
                do { memcpy(out, &v, 8); out += 8; } while (out < end);
            } else {
                // a short period: lay the pattern down twice by bytes, then copy from far enough back
                // (a multiple of the period, at least 8) eight bytes at a time
                uint8_t *q = out;
                const uint32_t head = distance * ((7 + distance) / distance);       // >= 8, multiple of the period
                for (uint32_t k = 0; k < head && q < end; k++) *q++ = *src++;
                if (q < end) {
                    const uint8_t *s2 = q - head;
                    do { memcpy(q, s2, 8); q += 8; s2 += 8; } while (q < end);
                }
            }
            out = end;
        }
        wpos_ = (size_t)(out - obuf_);
        return ok;
    }

    bool stored() {
        const size_t room = HIST + CAP - wpos_;
        size_t n = std::min<size_t>(stored_left_, room);
        if ((size_t)(in_end_ - in_) < n) return fail("truncated stored block");
        memcpy(obuf_ + wpos_, in_, n);
        in_ += n; wpos_ += n; stored_left_ -= (uint32_t)n;
        if (stored_left_ == 0) {
            state_ = S_BLOCK;
            if (final_) member_done_ = true;
        }
        return true;
    }

    // one step of the state machine; true unless the stream is bad
    bool produce() {
        for (;;) {
            switch (state_) {
            case S_MEMBER: if (!member_header()) return false; if (state_ == S_END) return true; break;
            case S_BLOCK: if (member_done_) return true; if (!block_header()) return false; break;
            case S_STORED: if (!stored()) return false; return true;
            case S_HUFF: if (!huffman()) return false; return true;
            case S_END: return true;
            }
        }
    }
};

}  // namespace tdhost
