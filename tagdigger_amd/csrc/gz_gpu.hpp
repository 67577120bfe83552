// Ordinary (single-stream) gzip decoded ON THE GPU: what the reference reads through gzip.open (tagdigger_fun.py:240-243).
// The host threads of par_inflate.hpp bound that tier (16 cores decode ~3 GB/s of compressed data into symbols, and the
// symbols are twice the size of the text on their way over PCIe); here only the compressed bytes go to the device:
//   1. k_gz_find     one wave per territory of the compressed file (128 KiB): the first bit position where a block
//                    starts that only a compressor would write -- non-final, dynamic Huffman, all three codes complete, an
//                    end code present.  64 positions a step: a lane each for the header's fixed fields and the code-length
//                    code's Kraft sum; the few that pass are looked at by the whole wave, which builds the tables as the
//                    decoder does (option gz_gpu_verify: it also decodes the block and asks for a header behind it).
//   2. k_gz_tokens   one wave per chunk (from one found start to the next): the Huffman decoding.  WHERE a code starts is
//                    serial, WHAT would start at a bit is not: every lane decodes the token that would begin at its bit of the
//                    next 64 (its three words of input by ds_bpermute, the fields by 32-bit funnel shifts, two table look-ups
//                    for all lanes at once: 10-bit literal/length and 8-bit distance tables of 32-bit entries, 6 KiB of LDS
//                    per wave, built by the wave's lanes), the wave follows the chain from bit 0 by lane number (six scalar
//                    instructions a token), the lanes that were real store their tokens.  Block headers and the rare
//                    longer codes are read by wave-uniform scalar code; the compressed bytes sit in registers, 64 words each.
//                    Output: TOKENS (a literal, or length + distance) -- the decoder never reads what it has decoded, so no
//                    memory latency sits in its loop.  A chunk ends at the first block boundary at or past its successor's
//                    start.
//   3. the host chains the chunks (each must begin exactly where its predecessor ended: the result never depends on
//      what step 1 found; the stretch behind a false start is decoded a second time), and sums their sizes;
//   4. k_gz_lz       one wave per chunk, 64 tokens at a time: positions by a wave scan, literals stored, copies made lane
//                    by lane in rounds (a copy waits for the copies in front of it that it reads from); a copy that reaches
//                    before its chunk gives MARKERS (0x8000 | position in the unknown 32 KiB before the chunk), as in
//                    par_inflate.hpp.  Output: 16-bit symbols;
//   5. k_gz_windows  every chunk's window = its predecessor's last 32 KiB with the markers in them replaced from the
//                    predecessor's window: a chain, walked in segments of 32 chunks (maps from the identity, one workgroup
//                    applying them in order, the segments again from their real windows);
//   6. gz_resolve.hpp's k_gz_resolve / k_gz_crc turn symbols into bytes and take the CRC-32 (as for the host decoder's
//      symbols), and the count kernels run over the bytes where they lie.
// The host side (tagdig.hip GzGpuStream) takes a file through these in SEGMENTS of 1 GiB of compressed bytes -- a segment's
// first chunk starts where the segment before ended, the window, the unfinished line and the member's CRC-32 are carried,
// the next segment's bytes are uploaded meanwhile -- and member by member.  Anything unusual in the first segment -- small
// members, a chunk that does not chain, tables zlib would refuse, an overflowing token buffer -- sends the file to the host
// decoder (count_gzip_dev) before anything has been counted; later, and for a failed CRC or length check, to the reference's
// reading rules (gz_pyrules.hpp).
// Measured (16 M reads, 662 MB of gzip, 427 M tokens of which 0.7 % need the scalar code): upload 13 ms, k_gz_find 7.4,
// k_gz_tokens 17.8, k_gz_lz 20.3, windows 2, k_gz_resolve 3, k_gz_crc 6: 72 ms, 222 M reads/s (16 host threads: 75-79 M).
// The decoder is bound by the instructions it issues (five waves a SIMD, no memory wait in its loop): what made it faster
// was fewer of them -- DESIGN 5. has the steps from 122 ms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdgz2 {

constexpr int LROOT = 10, DROOT = 8, PROOT = 7;
constexpr uint32_t WAVES = 4;                         // waves (streams) per workgroup
constexpr uint32_t WINDOW = 32768;
constexpr uint64_t NONE = ~0ull;

// A table entry (32 bits): bits 0-3 the code's length (0: no code of this index -- E_NONE -- or a longer one -- E_SLOW);
// bits 4-7 the number of extra bits; bits 8-23 the value: a literal's byte, a copy's base length, a distance's base;
// F_COPY / F_END / F_BAD: a length symbol, the end-of-block symbol, a symbol DEFLATE does not define (286, 287; 30, 31)
// F_RARE: what the 64-positions-a-step decoder leaves to the scalar code -- set in E_NONE, E_SLOW and with F_BAD
constexpr uint32_t F_COPY = 1u << 24, F_END = 1u << 25, F_BAD = 1u << 26, F_RARE = 1u << 27, E_NONE = F_RARE, E_SLOW = 0x10u | F_RARE;
struct __attribute__((aligned(16))) WaveMem {         // one wave's tables (LDS)
    uint32_t ltab[1 << LROOT];                        // literal/length
    uint32_t dtab[1 << DROOT];                        // distance (and the code-length code while a header is read)
    uint16_t lsym[288], dsym[32];                     // symbols in canonical order (codes longer than the index)
    uint16_t lcnt[16], dcnt[16];                      // codes per length
    uint8_t lens[320];                                // code lengths: literal/length [0, 288), distance [288, 320)
};

struct Chunk {
    uint64_t start_bit;       // where its first block header begins
    uint64_t stop_bit;        // it ends at the first block boundary at or past this (its successor's start)
    uint64_t tok_off;         // its tokens in the token buffer
    uint32_t tok_cap, pad;
};
enum { S_NONE = 0, S_BOUNDARY = 1, S_FINAL = 2, S_ERR = 3, S_TOKCAP = 4, S_UNUSUAL = 5 };
struct ChunkOut {
    uint64_t end_bit;         // the bit behind its last block
    uint64_t out_len;         // bytes it inflates to
    uint32_t ntok, status;
    uint32_t nslow, pad;      // (statistics: tokens the scalar code decoded)
};

// ---------------------------------------------------------------- the bit reader: wave-uniform state, the compressed bytes in
// two registers (this lane's word of the current block of 64 words, and of the next)
struct Bits {
    const uint32_t *base;
    uint64_t nwords;          // words that may be loaded (zero padding behind the file included)
    uint64_t w0;              // the word wpos counts from (a multiple of 64)
    uint32_t wpos;            // the next word to go into the bit buffer
    uint32_t bc;
    uint64_t bb;
    uint32_t vin, vnext;
};
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t bits_block(const Bits &b, uint32_t blk, int lane) {
    const uint64_t i = b.w0 + (uint64_t)blk * 64u + (uint32_t)lane;
    return i < b.nwords ? b.base[i] : 0u;
}
__device__ __forceinline__ void refill(Bits &b, int lane) {
    if (b.bc <= 32u) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)b.vin, (int)(b.wpos & 63u));
        b.bb |= (uint64_t)w << b.bc;
        b.bc += 32u;
        b.wpos++;
        if (__builtin_expect((b.wpos & 63u) == 0u, 0)) {
            b.vin = b.vnext;
            b.vnext = bits_block(b, (b.wpos >> 6) + 1u, lane);
        }
    }
}
__device__ __forceinline__ void drop(Bits &b, uint32_t n) { b.bb >>= n; b.bc -= n; }
__device__ __forceinline__ uint32_t take(Bits &b, uint32_t n) {
    const uint32_t v = (uint32_t)b.bb & ((1u << n) - 1u);
    drop(b, n);
    return v;
}
__device__ __forceinline__ uint64_t bitpos(const Bits &b) { return (b.w0 + b.wpos) * 32u - b.bc; }
__device__ __forceinline__ void bits_init(Bits &b, uint64_t pos, int lane) {
    b.w0 = (pos >> 5) & ~(uint64_t)63;
    b.wpos = (uint32_t)(pos >> 5) & 63u;
    b.vin = bits_block(b, 0u, lane);
    b.vnext = bits_block(b, 1u, lane);
    b.bb = 0; b.bc = 0;
    refill(b, lane);
    drop(b, (uint32_t)pos & 31u);
    refill(b, lane);
}

// ---------------------------------------------------------------- canonical Huffman tables from n code lengths in LDS (n <= 320),
// built by the wave: tab[1 << root], sym[] (symbols in canonical order), cnt[16].  Returns 0 for a complete code, 1 for an
// incomplete one (ncodes, maxlen say how incomplete), -1 for an over-subscribed one.
// length and distance of a copy from their symbols: base value and number of extra bits (RFC 1951 3.2.5, by formula)
__device__ __forceinline__ void length_code(uint32_t sy, uint32_t &base, uint32_t &ext) {
    const uint32_t li = sy - 257u;
    ext = li < 8u ? 0u : (li - 4u) >> 2;
    base = li < 8u ? li + 3u : ((4u + (li & 3u)) << ext) + 3u;
    if (li == 28u) { ext = 0; base = 258u; }
}
__device__ __forceinline__ void distance_code(uint32_t ds, uint32_t &base, uint32_t &ext) {
    ext = ds < 4u ? 0u : (ds - 2u) >> 1;
    base = ds < 4u ? ds + 1u : ((2u + (ds & 1u)) << ext) + 1u;
}
// what a table says about symbol i (without the code's length): KIND 0 literal/length, 1 distance, 2 the code-length code
template <int KIND> __device__ __forceinline__ uint32_t entry_of(uint32_t i) {
    if (KIND == 2) return i << 8;
    if (KIND == 1) {
        if (i > 29u) return F_BAD | F_RARE;
        uint32_t base, ext;
        distance_code(i, base, ext);
        return (base << 8) | (ext << 4);
    }
    if (i < 256u) return i << 8;
    if (i == 256u) return F_END;
    if (i > 285u) return F_BAD | F_RARE;
    uint32_t base, ext;
    length_code(i, base, ext);
    return F_COPY | (base << 8) | (ext << 4);
}
template <int KIND>
__device__ __forceinline__ int build(const uint8_t *lens, uint32_t n, uint32_t *tab, uint32_t root, uint16_t *sym, uint16_t *cnt, int lane,
                                     uint32_t &ncodes, uint32_t &maxlen) {
    uint32_t c[16];
#pragma unroll
    for (int l = 0; l < 16; l++) c[l] = 0;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t li = i < n ? lens[i] : 0u;
#pragma unroll
        for (int l = 1; l < 16; l++) c[l] += (uint32_t)__builtin_popcountll(__ballot(li == (uint32_t)l));
    }
    int left = 1;
    ncodes = 0; maxlen = 0;
    bool over = false;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        left = 2 * left - (int)c[l];
        over |= left < 0;
        ncodes += c[l];
        if (c[l]) maxlen = (uint32_t)l;
    }
    if (over) return -1;
    for (uint32_t k = (uint32_t)lane; k < (1u << root); k += 64u) tab[k] = E_NONE;
#pragma unroll
    for (int l = 0; l < 16; l++) cnt[l] = (uint16_t)c[l];              // (every lane the same value)
    uint32_t nc[16], of[16];
    nc[1] = 0; of[1] = 0;
#pragma unroll
    for (int l = 1; l < 15; l++) { nc[l + 1] = (nc[l] + c[l]) << 1; of[l + 1] = of[l] + c[l]; }
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t li = i < n ? lens[i] : 0u;
        uint32_t mycode = 0, mypos = 0;
#pragma unroll
        for (int l = 1; l < 16; l++) {
            const uint64_t m = __ballot(li == (uint32_t)l);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (li == (uint32_t)l) { mycode = nc[l] + below; mypos = of[l] + below; }
            const uint32_t k = (uint32_t)__builtin_popcountll(m);
            nc[l] += k; of[l] += k;
        }
        if (li) {
            sym[mypos] = (uint16_t)i;
            const uint32_t rv = __builtin_bitreverse32(mycode) >> (32u - li);
            if (li <= root) {
                const uint32_t e = entry_of<KIND>(i) | li;
                for (uint32_t k = rv; k < (1u << root); k += 1u << li) tab[k] = e;
            } else {
                tab[rv & ((1u << root) - 1u)] = E_SLOW;
            }
        }
    }
    return left == 0 ? 0 : 1;
}

// a symbol whose code is longer than the table's index: bit by bit against the canonical arrays (RFC 1951 3.2.2); the
// code's bits are dropped; returns its entry (length bits 0) or E_NONE
template <int KIND>
__device__ __forceinline__ uint32_t slow_entry(Bits &b, const uint16_t *sym, const uint16_t *cnt) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l < 16; l++) {
        code |= (uint32_t)(b.bb >> (l - 1u)) & 1u;
        const uint32_t c = rfl(cnt[l]);
        if (code < first + c) { drop(b, l); return entry_of<KIND>(rfl(sym[index + (code - first)])) | 1u; }
        index += c; first += c;
        first <<= 1; code <<= 1;
    }
    return E_NONE;
}
// the next symbol's entry, its code's bits dropped (at least 15 bits are in the buffer); E_NONE: no such code
template <int KIND>
__device__ __forceinline__ uint32_t next_entry(Bits &b, const uint32_t *tab, uint32_t root, const uint16_t *sym, const uint16_t *cnt) {
    const uint32_t e = rfl(tab[(uint32_t)b.bb & ((1u << root) - 1u)]);
    if (__builtin_expect((e & 15u) == 0u, 0)) return e == E_SLOW ? slow_entry<KIND>(b, sym, cnt) : E_NONE;
    drop(b, e & 15u);
    return e;
}

__constant__ const uint8_t PRECODE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// One block header.  Returns 2: Huffman tables built; 0: a stored block (`stored` bytes follow at the byte boundary the
// reader stands on); -1: invalid (what zlib refuses); -2: valid for zlib perhaps, but not decoded here (an incomplete
// literal/length code).  strict (the block search): a dynamic block with all three codes complete, or -1.
__device__ __attribute__((noinline)) int read_header_impl(Bits &b, WaveMem &m, bool strict, uint32_t &final, uint32_t &stored, int lane) {
    refill(b, lane);
    final = take(b, 1);
    const uint32_t type = take(b, 2);
    if (type == 3u) return -1;
    if (strict && (type != 2u || final)) return -1;
    if (type == 0u) {
        drop(b, b.bc & 7u);
        refill(b, lane);
        const uint32_t len = take(b, 16);
        refill(b, lane);
        const uint32_t nlen = take(b, 16);
        if ((len ^ nlen) != 0xFFFFu) return -1;
        stored = len;
        return 0;
    }
    uint32_t nlit = 288, ndist = 30;
    if (type == 1u) {
        for (uint32_t i = (uint32_t)lane; i < 320u; i += 64u)
            m.lens[i] = (uint8_t)(i < 144u ? 8 : i < 256u ? 9 : i < 280u ? 7 : i < 288u ? 8 : i < 318u ? 5 : 0);
    } else {
        nlit = take(b, 5) + 257u; ndist = take(b, 5) + 1u;
        const uint32_t ncl = take(b, 4) + 4u;
        if (nlit > 286u || ndist > 30u) return -1;
        uint8_t *pl = reinterpret_cast<uint8_t *>(m.lsym);              // the code-length code's lengths (19)
        if (lane < 19) pl[lane] = 0;
        for (uint32_t i = 0; i < ncl; i++) {
            refill(b, lane);
            pl[PRECODE_ORDER[i]] = (uint8_t)take(b, 3);
        }
        uint32_t pn, pmax;
        if (build<2>(pl, 19, m.dtab, PROOT, m.dsym, m.dcnt, lane, pn, pmax) != 0) return -1;   // (zlib: must be complete)
        const uint32_t total = nlit + ndist;
        uint32_t i = 0, prev = 0;
        while (i < total) {
            refill(b, lane);
            const uint32_t ce = next_entry<2>(b, m.dtab, PROOT, m.dsym, m.dcnt);
            if (ce == E_NONE) return -1;
            const int c = (int)(ce >> 8);
            if (c < 16) { m.lens[i++] = (uint8_t)c; prev = (uint32_t)c; continue; }
            uint32_t val = 0, rep;
            if (c == 16) { if (i == 0) return -1; val = prev; rep = 3u + take(b, 2); }
            else if (c == 17) { rep = 3u + take(b, 3); }
            else { rep = 11u + take(b, 7); }
            if (i + rep > total) return -1;
            for (uint32_t k = (uint32_t)lane; k < rep; k += 64u) m.lens[i + k] = (uint8_t)val;
            i += rep; prev = val;
        }
        // the distance lengths follow the literal/length ones: to their place (one instruction reads them all, the next writes)
        const uint32_t dl = (uint32_t)lane < ndist ? m.lens[nlit + (uint32_t)lane] : 0u;
        if (lane < 32) m.lens[288 + lane] = (uint8_t)dl;
        for (uint32_t k = nlit + (uint32_t)lane; k < 288u; k += 64u) m.lens[k] = 0;
        if (rfl(m.lens[256]) == 0u) return -1;                           // no end-of-block code
    }
    uint32_t ln, lmax, dn, dmax;
    const int rl = build<0>(m.lens, 288, m.ltab, LROOT, m.lsym, m.lcnt, lane, ln, lmax);
    const int rd = build<1>(m.lens + 288, 32, m.dtab, DROOT, m.dsym, m.dcnt, lane, dn, dmax);
    if (rl < 0 || rd < 0) return -1;
    if (strict) return rl == 0 && rd == 0 ? 2 : -1;
    if (rl == 1) return lmax == 1u ? -2 : -1;                            // (zlib allows an incomplete code of one 1-bit symbol only)
    if (rd == 1 && type == 2u && !(dn == 0u || dmax == 1u)) return -1;   // (... the fixed block's 30 of 32 five-bit codes are its own)
    return 2;
}

// (a real call, once a block: the decoding loop's registers are not the header's -- and the reader goes in and out by value,
// so that it never lives in memory)
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }
__device__ __forceinline__ int read_header(Bits &b, WaveMem &m, bool strict, uint32_t &final, uint32_t &stored, int lane) {
    Bits t = b;
    uint32_t f = 0, st = 0;
    const int r = read_header_impl(t, m, strict, f, st, lane);
    b.w0 = rfl64(t.w0); b.wpos = rfl(t.wpos); b.bc = rfl(t.bc); b.bb = rfl64(t.bb);
    b.vin = t.vin; b.vnext = t.vnext;
    final = rfl(f); stored = rfl(st);
    return (int)rfl((uint32_t)r);
}

// ---------------------------------------------------------------- a block's symbols, 64 bit positions at a time
// A Huffman stream is serial -- where a code starts is known only when the one before it has been read -- but WHAT would
// start at a given bit is not: every lane decodes the token that would begin at its bit of the next 64 (a literal, the end
// code, or length + extra bits + distance + extra bits: two table look-ups, all 64 lanes' at once), and the wave then only
// follows the chain from the first bit -- token and size by lane number out of two registers, no table, no memory.  What a
// look-up cannot settle (a code longer than the table's index) is decoded by the scalar code when the chain comes to it.
// pos: in, the bit behind the block's header; out, the bit behind its end code.  sink(mask, tokens): the lanes of `mask` hold
// the next tokens, in lane order; false stops the decoding.  Returns S_NONE at the end code, S_ERR for invalid data, S_TOKCAP
// for the sink's stop.
template <class F>
__device__ __forceinline__ uint32_t decode_block(const Bits &rd, uint64_t &pos, uint64_t in_bits, const WaveMem &m, int lane, F &&sink, uint32_t *nslow = nullptr) {
    Bits t = rd;
    t.w0 = (pos >> 5) & ~(uint64_t)63;
    uint32_t cur = (uint32_t)(pos - (t.w0 << 5));
    // three blocks of 64 words in registers: the two a step reads from, and the one behind them -- loaded a block ahead and waited
    // for only where it moves up.  (With two, the step's first read of vnext made the compiler wait for the memory counter in
    // EVERY step, and on this hardware that counter holds the stores as well: each step waited for the token store of the
    // step before, a round trip to the L2.)
    uint32_t vin = bits_block(t, 0u, lane), vnext = bits_block(t, 1u, lane);
    asm volatile("" : "+v"(vin), "+v"(vnext));
    uint32_t vnn = bits_block(t, 2u, lane);
    for (;;) {
        if (cur >= 2048u) {
            t.w0 += 64u; cur -= 2048u;
            vin = vnext; vnext = vnn;
            asm volatile("" : "+v"(vnext));
            vnn = bits_block(t, 2u, lane);
        }
        if ((t.w0 << 5) + cur > in_bits) return S_ERR;
        const uint32_t k0 = cur >> 5, off = cur & 31u;
        // this lane's 64 bits from its position on: three words of the input, the first one word (off + lane) / 32 of the step's
        const uint32_t bp = off + (uint32_t)lane, q = bp >> 5, r = bp & 31u;
        uint32_t lo, mid, hi;
        if (__builtin_expect(k0 + 4u < 64u, 1)) {
            // fifteen steps in sixteen all of them lie in the first register: three ds_bpermute (a lane each its own index) -- the five
            // words by readlane into scalar registers and six selects a lane were a fifth of the step's vector instructions
            const int a = (int)((k0 + q) << 2);
            lo = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)vin);
            mid = (uint32_t)__builtin_amdgcn_ds_bpermute(a + 4, (int)vin);
            hi = (uint32_t)__builtin_amdgcn_ds_bpermute(a + 8, (int)vin);
        } else {
            uint32_t W[5];
#pragma unroll
            for (uint32_t i = 0; i < 5u; i++) {
                const uint32_t k = k0 + i;
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)vin, (int)(k & 63u)), c = (uint32_t)__builtin_amdgcn_readlane((int)vnext, (int)(k & 63u));
                W[i] = k < 64u ? a : c;
            }
            lo = q == 0u ? W[0] : q == 1u ? W[1] : W[2]; mid = q == 0u ? W[1] : q == 1u ? W[2] : W[3]; hi = q == 0u ? W[2] : q == 1u ? W[3] : W[4];
        }
        // (x1:x0, and every field out of them by one 32-bit funnel shift: the fields begin at bit 0, l1 <= 15, l1 + ext <= 20 and
        // l1 + ext + l2 -- a copy whose distance bits would begin past bit 31 is the scalar code's; 64-bit shifts are slower)
        const uint32_t x0 = __builtin_amdgcn_alignbit(mid, lo, r), x1 = __builtin_amdgcn_alignbit(hi, mid, r);
        const uint32_t e = m.ltab[x0 & ((1u << LROOT) - 1u)];
        const uint32_t l1 = e & 15u, ext = (e >> 4) & 15u;
        const uint32_t len = ((e >> 8) & 0xFFFFu) + (__builtin_amdgcn_alignbit(x1, x0, l1) & ((1u << ext) - 1u));
        const uint32_t s2 = l1 + ext;
        const uint32_t d = m.dtab[__builtin_amdgcn_alignbit(x1, x0, s2) & ((1u << DROOT) - 1u)];
        const uint32_t l2 = d & 15u, dext = (d >> 4) & 15u;
        const uint32_t s3 = s2 + l2;
        const uint32_t dist = ((d >> 8) & 0xFFFFu) + (__builtin_amdgcn_alignbit(x1, x0, s3) & ((1u << dext) - 1u));
        const bool copy = (e & F_COPY) != 0;
        // vinfo: the token's size in bits | 0x200 the end code | 0x800 for the scalar code (a longer code, none, or an undefined symbol)
        uint32_t flags = ((e | (copy ? d : 0u)) >> 16) & ((F_END | F_RARE) >> 16);
        if (copy && s3 > 31u) flags |= F_RARE >> 16;
        uint32_t vtok = copy ? 0x80000000u | (len << 15) | (dist - 1u) : (e >> 8) & 0xFFu;
        uint32_t vinfo = (copy ? s3 + dext : s2) | flags;
        // the chain from bit 0: which lanes' tokens are real.  The common way is six scalar instructions a token: the jump to the next
        // token's lane; an end code or a token for the scalar code jumps out of the window and is looked at behind the loop.
        const uint32_t vjump = flags ? 64u : (vinfo & 0xFFu);
        uint32_t o = 0;
        uint64_t onpath = 0;
        bool ended = false;
        for (;;) {
            uint32_t last, j;
            do {
                asm("s_bitset1_b64 %0, %1" : "+s"(onpath) : "s"(o));            // onpath |= 1 << o
                j = (uint32_t)__builtin_amdgcn_readlane((int)vjump, (int)o);
                o += j;
            } while (o < 64u);
            last = o - j;
            uint32_t info = (uint32_t)__builtin_amdgcn_readlane((int)vinfo, (int)last);
            if (__builtin_expect(info < 0x200u, 1)) break;
            onpath &= ~(1ull << last);
            o = last;
            if (info >= 0x300u) {
                // the scalar decoder on this one token: 64 bits from its position are enough (15 + 5 + 15 + 13)
                if (nslow) (*nslow)++;
                Bits sbits = t;
                sbits.bb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)x1, (int)o) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)x0, (int)o);      // (its lane's)
                sbits.bc = 64u;
                uint32_t tk = 0;
                // (the tables through a pointer the compiler cannot see through: it had hoisted the loads of the canonical arrays -- six LDS
                // reads and their unpacking -- out of this rare branch into every step of the loop)
                const WaveMem *mm = &m;
                asm volatile("" : "+v"(mm));
                const uint32_t se = next_entry<0>(sbits, mm->ltab, LROOT, mm->lsym, mm->lcnt);
                if (se == E_NONE || (se & F_BAD)) return S_ERR;
                if (se & F_END) { info = (64u - sbits.bc) | (2u << 8); }
                else if (!(se & F_COPY)) { tk = (se >> 8) & 0xFFu; info = 64u - sbits.bc; }
                else {
                    const uint32_t slen = ((se >> 8) & 0xFFFFu) + take(sbits, (se >> 4) & 15u);
                    const uint32_t sd = next_entry<1>(sbits, mm->dtab, DROOT, mm->dsym, mm->dcnt);
                    if (sd == E_NONE || (sd & F_BAD)) return S_ERR;
                    const uint32_t sdist = ((sd >> 8) & 0xFFFFu) + take(sbits, (sd >> 4) & 15u);
                    tk = 0x80000000u | (slen << 15) | (sdist - 1u);
                    info = (64u - sbits.bc) | (1u << 8);
                }
                if ((uint32_t)lane == o) vtok = tk;
            }
            if ((info >> 8) == 2u) { ended = true; o += info & 0xFFu; break; }
            onpath |= 1ull << o;
            o += info & 0xFFu;
            if (o >= 64u) break;
        }
        // the real tokens, in order, to the sink (false: it has no room, or has seen enough)
        if (onpath && !sink(onpath, vtok)) return S_TOKCAP;
        cur += o;
        if (ended) { pos = (t.w0 << 5) + cur; return S_NONE; }
    }
}

// ---------------------------------------------------------------- 1. the block search
// found[t] (t >= 1): the first block start in territory t, or NONE
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_gz_find(const uint8_t *in, uint64_t in_bits, uint64_t nwords, uint64_t first_bit,
                                                        uint64_t terr_bits, uint32_t nterr, uint64_t *found, uint32_t verify, uint32_t t_first) {
    __shared__ WaveMem mem[WAVES];
    __shared__ uint64_t queue[WAVES][128];
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint32_t t = blockIdx.x * WAVES + wave + t_first;              // (t_first 1: territory 0 begins with the known start)
    if (t >= nterr) return;
    WaveMem &m = mem[wave];
    uint64_t lo = (uint64_t)t * terr_bits;
    const uint64_t hi = lo + terr_bits < in_bits ? lo + terr_bits : in_bits;
    if (lo <= first_bit) lo = first_bit + 1u;
    uint64_t result = NONE;
    Bits b;
    b.base = reinterpret_cast<const uint32_t *>(in); b.nwords = nwords;
    // the 17 + 3 x 19 bits of a header's fixed fields and code-length code from bit p on
    auto window = [&](uint64_t p, uint64_t &x0, uint64_t &x1) {
        uint64_t w0, w1;
        __builtin_memcpy(&w0, in + (p >> 3), 8);
        __builtin_memcpy(&w1, in + (p >> 3) + 8, 8);
        const uint32_t sh = (uint32_t)p & 7u;
        x0 = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
        x1 = w1 >> sh;
    };
    // the whole wave on one position: the header (and with `verify` the block to its end code and the header behind it)
    auto validate = [&](uint64_t pc) -> bool {
        bits_init(b, pc, lane);
        uint32_t final, stored;
        if (read_header(b, m, true, final, stored, lane) != 2) return false;
        if (lane == 0) atomicAdd((unsigned long long *)(found + nterr), 1ull);          // (statistics: headers that pass)
        // verify: the block is decoded to its end code, and another header must follow -- else the headers' checks are
        // trusted (they left no false start in 10^9 positions of the measured file), and a false start costs the chunk in
        // front of it a second decoding when the host chains the chunks
        if (verify) {
            uint64_t at = bitpos(b);
            uint32_t nsym = 0;
            const uint32_t dr = decode_block(b, at, in_bits, m, lane, [&](uint64_t mask, uint32_t) { nsym += (uint32_t)__builtin_popcountll(mask); return nsym < (1u << 21); });
            if (dr != S_NONE || at + 3u > in_bits) return false;
            bits_init(b, at, lane);
            refill(b, lane);
            const uint32_t nx = (uint32_t)b.bb;
            const uint32_t ntype = (nx >> 1) & 3u;
            if (ntype == 3u) return false;
            if (ntype == 0u) {
                drop(b, 3);
                drop(b, b.bc & 7u);
                refill(b, lane);
                const uint32_t len = take(b, 16);
                refill(b, lane);
                const uint32_t nlen = take(b, 16);
                if ((len ^ nlen) != 0xFFFFu) return false;
            } else if (ntype == 2u) {
                if (((nx >> 3) & 31u) > 29u || ((nx >> 8) & 31u) > 29u) return false;
            }
        }
        if (lane == 0) atomicAdd((unsigned long long *)(found + nterr + 1), 1ull);      // (... of them block starts)
        return true;
    };
    // Two sieves.  The first (three header bits, two 5-bit counts: one position in nine passes) runs on 64 positions a step; what
    // passes waits in a queue until 64 are together for the second (the code-length code's Kraft sum: nineteen fields) -- run on
    // the few lanes of every step it was two thirds of the search's instructions.
    uint64_t *q = queue[wave];
    uint32_t nq = 0;
    auto drain = [&](bool all) {
        while (result == NONE && (nq >= 64u || (all && nq))) {
            const uint32_t n = nq < 64u ? nq : 64u;
            const uint64_t pq = (uint32_t)lane < n ? q[lane] : 0ull;
            bool pass = false;
            if ((uint32_t)lane < n) {
                uint64_t x0, x1;
                window(pq, x0, x1);
                const uint32_t ncl = (((uint32_t)x0 >> 13) & 15u) + 4u;
                const uint64_t y = (x0 >> 17) | (x1 << 47);
                uint32_t kraft = 0;
#pragma unroll
                for (uint32_t i = 0; i < 19u; i++) {
                    const uint32_t l = (uint32_t)(y >> (3u * i)) & 7u;
                    kraft += i < ncl && l ? 128u >> l : 0u;
                }
                pass = kraft == 128u;
            }
            uint64_t pm = __ballot(pass);
            while (pm && result == NONE) {                              // (in the order of the positions: the first start counts)
                const int l = (int)__builtin_ctzll(pm);
                pm &= pm - 1u;
                const uint64_t pc = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pq >> 32), l) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pq, l);
                if (validate(pc)) result = pc;
            }
            const bool more = (uint32_t)lane + n < nq;
            const uint64_t rest = more ? q[(uint32_t)lane + n] : 0ull;
            if (more) q[lane] = rest;
            nq -= n;
        }
    };
    for (uint64_t p0 = lo; p0 < hi && result == NONE; p0 += 64u) {
        const uint64_t p = p0 + (uint32_t)lane;
        uint64_t x0, x1;
        window(p, x0, x1);
        const bool cand = p < hi && ((uint32_t)x0 & 7u) == 4u && (((uint32_t)x0 >> 3) & 31u) <= 29u && (((uint32_t)x0 >> 8) & 31u) <= 29u;
        const uint64_t cm = __ballot(cand);
        if (cand) q[nq + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u))] = p;
        nq += (uint32_t)__builtin_popcountll(cm);
        drain(false);
    }
    drain(true);
    if (lane == 0) found[t] = result;
}

// ---------------------------------------------------------------- 2. Huffman decoding into tokens
// a token: a literal (its byte), or 0x80000000 | length << 15 | (distance - 1)
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_gz_tokens(const uint8_t *in, uint64_t in_bits, uint64_t nwords, const Chunk *chunks, uint32_t nchunks,
                                                          uint32_t *tok, ChunkOut *out) {
    __shared__ WaveMem mem[WAVES];
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint32_t ci = blockIdx.x * WAVES + wave;
    if (ci >= nchunks) return;
    WaveMem &m = mem[wave];
    const uint64_t start = chunks[ci].start_bit, stop = chunks[ci].stop_bit;
    uint32_t *tk = tok + chunks[ci].tok_off;
    const uint32_t cap = chunks[ci].tok_cap;
    Bits b;
    b.base = reinterpret_cast<const uint32_t *>(in); b.nwords = nwords;
    bits_init(b, start, lane);
    uint32_t ntok = 0, status = S_NONE, nslow = 0;
    uint64_t out_len = 0, end_bit = 0;
    uint64_t vlen = 0;                                                     // this lane's share of the decoded tokens' bytes
    for (;;) {
        const uint64_t pos = bitpos(b);
        if (pos >= stop) { status = S_BOUNDARY; end_bit = pos; break; }
        if (pos + 3u > in_bits) { status = S_ERR; break; }
        uint32_t final = 0, stored = 0;
        const int h = read_header(b, m, false, final, stored, lane);
        if (h < 0) { status = h == -2 ? S_UNUSUAL : S_ERR; break; }
        if (h == 0) {
            // a stored block: its bytes are literals (64 a step); the reader starts again behind them
            const uint64_t at = bitpos(b) >> 3;
            if ((at + stored) * 8u > in_bits) { status = S_ERR; break; }
            if ((uint64_t)ntok + stored + 192u > cap) { status = S_TOKCAP; break; }
            for (uint32_t i = (uint32_t)lane; i < stored; i += 64u) tk[ntok + i] = in[at + i];
            ntok += stored; out_len += stored;
            bits_init(b, (at + stored) * 8u, lane);
        } else {
            uint64_t at = bitpos(b);
            const uint32_t dr = decode_block(b, at, in_bits, m, lane, [&](uint64_t mask, uint32_t t) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if ((mask >> lane) & 1ull) {
                    tk[ntok + rank] = t;
                    vlen += t >> 31 ? (t >> 15) & 0x1FFu : 1u;
                }
                ntok += (uint32_t)__builtin_popcountll(mask);
                return ntok + 128u <= cap;
            }, &nslow);
            if (dr != S_NONE && status == S_NONE) status = dr;
            if (status == S_NONE) bits_init(b, at, lane);
            if (status != S_NONE) break;
        }
        if (final) { status = S_FINAL; end_bit = bitpos(b); break; }
    }
    {
        uint64_t v = vlen;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, d, 64);
        out_len += v;
    }
    if (lane == 0) {
        ChunkOut o;
        o.end_bit = end_bit; o.out_len = out_len; o.ntok = ntok; o.status = status; o.nslow = nslow; o.pad = 0;
        out[ci] = o;
    }
}

// ---------------------------------------------------------------- 4. tokens -> 16-bit symbols
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ void wave_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__global__ __launch_bounds__(64 * WAVES) void k_gz_lz(const uint32_t *tok, const Chunk *chunks, const ChunkOut *res, const uint64_t *sym_off,
                                                      uint32_t nchunks, uint16_t *sym) {
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint32_t ci = blockIdx.x * WAVES + wave;
    if (ci >= nchunks) return;
    const uint32_t ntok = res[ci].ntok;
    const uint32_t *tk = tok + chunks[ci].tok_off;
    uint16_t *out = sym + sym_off[ci];
    int64_t base = 0;
    uint32_t tnext = (uint32_t)lane < ntok ? tk[lane] : 0u;
    for (uint32_t g = 0; g < ntok; g += 64u) {
        const bool valid = g + (uint32_t)lane < ntok;
        const uint32_t t = tnext;
        // the next 64 tokens are asked for now: their way from memory lies behind this group's first wait for its stores
        tnext = g + 64u + (uint32_t)lane < ntok ? tk[g + 64u + (uint32_t)lane] : 0u;
        const bool match = valid && (t >> 31);
        const uint32_t L = match ? (t >> 15) & 0x1FFu : (valid ? 1u : 0u);
        const uint32_t dist = (t & 0x7FFFu) + 1u;
        const uint32_t incl = wave_scan_incl(L, lane);
        const int64_t pos = base + (int64_t)(incl - L);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (valid && !match) out[pos] = (uint16_t)t;
        const int64_t src = pos - (int64_t)dist;
        const uint32_t need = L < dist ? L : dist;
        bool pending = match;
        // everything in front of F is in place: in the first round what the groups before have written -- the copies that read only
        // that go with the literals, one wait for both --, then everything in front of the first copy still pending
        int64_t F = base;
        for (;;) {
            const bool ready = pending && src + (int64_t)need <= F;
            // long copies (48 symbols and more, source and destination apart) by the whole wave: four symbols a lane, up to four
            // copies' loads in flight before their stores -- one lane alone takes a memory round trip per sixteen symbols
            const bool coop = ready && L >= 48u && L <= dist && src >= 0;
            const uint32_t cdone = coop ? (L & ~3u) < 256u ? (L & ~3u) : 256u : 0u;      // what the wave copies of this lane's copy
            for (uint64_t cm = __ballot(coop); cm;) {
                int64_t cs[4], cp[4];
                uint32_t cn[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    cn[q] = 0; cs[q] = 0; cp[q] = 0;
                    if (cm) {
                        const int l = (int)__builtin_ctzll(cm);
                        cm &= cm - 1u;
                        cs[q] = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)src >> 32), l) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)src, l));
                        cp[q] = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)pos >> 32), l) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)pos, l));
                        cn[q] = (uint32_t)__builtin_amdgcn_readlane((int)cdone, l);
                    }
                }
                uint64_t cv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) if (4u * (uint32_t)lane < cn[q]) __builtin_memcpy(&cv[q], out + cs[q] + 4 * lane, 8);
#pragma unroll
                for (int q = 0; q < 4; q++) if (4u * (uint32_t)lane < cn[q]) __builtin_memcpy(out + cp[q] + 4 * lane, &cv[q], 8);
            }
            if (ready) {
                if (dist == 1u && src >= 0 && L > 1u) {
                    // a run of one symbol (the commonest overlap): one load, stores only
                    const uint64_t y = out[src];
                    const uint64_t v4 = y | (y << 16) | (y << 32) | (y << 48);
                    uint16_t *dp = out + pos;
                    uint32_t j = 0;
                    for (; j + 4u <= L; j += 4u) __builtin_memcpy(dp + j, &v4, 8);
                    for (; j < L; j++) dp[j] = (uint16_t)y;
                } else if (L <= dist && src >= 0) {
                    // the common copy -- source and destination apart, nothing from before the chunk: four symbols (8 bytes, at any
                    // 2-byte boundary) a load, sixteen symbols in flight before the first store (a load that had to wait for the
                    // store before it would take a memory round trip per piece)
                    const uint16_t *sp = out + src;
                    uint16_t *dp = out + pos;
                    uint32_t j = cdone;
                    for (; j + 16u <= L; j += 16u) {
                        uint64_t v[4];
#pragma unroll
                        for (uint32_t i = 0; i < 4u; i++) __builtin_memcpy(&v[i], sp + j + 4u * i, 8);
#pragma unroll
                        for (uint32_t i = 0; i < 4u; i++) __builtin_memcpy(dp + j + 4u * i, &v[i], 8);
                    }
                    uint64_t v[4];
                    const uint32_t rest = L - j;                           // 0..15
#pragma unroll
                    for (uint32_t i = 0; i < 4u; i++) if (4u * i < rest) __builtin_memcpy(&v[i], sp + j + 4u * i, 8);     // (reads up to 3 symbols past the source: inside the buffer)
#pragma unroll
                    for (uint32_t i = 0; i < 4u; i++) {
                        if (4u * i + 4u <= rest) __builtin_memcpy(dp + j + 4u * i, &v[i], 8);
                        else if (4u * i < rest) {
                            for (uint32_t q = 0; q < rest - 4u * i; q++) dp[j + 4u * i + q] = (uint16_t)(v[i] >> (16u * q));
                        }
                    }
                } else {
                    // a run (the source overlaps the destination: every symbol comes from [src, src + dist)), or symbols from before
                    // the chunk (markers): symbol by symbol, sixteen loads, then sixteen stores
                    uint32_t k = 0;
                    for (uint32_t j = 0; j < L; j += 16u) {
                        uint16_t v[16];
#pragma unroll
                        for (uint32_t i = 0; i < 16u; i++) {
                            const int64_t s = src + (int64_t)k;
                            v[i] = j + i >= L ? (uint16_t)0 : s < 0 ? (uint16_t)(0x8000u | (uint32_t)((int64_t)WINDOW + s)) : out[s];
                            if (++k == dist) k = 0;
                        }
#pragma unroll
                        for (uint32_t i = 0; i < 16u; i++)
                            if (j + i < L) out[pos + (int64_t)(j + i)] = v[i];
                    }
                }
            }
            wave_mem_fence();
            pending = pending && !ready;
            const uint64_t pm = __ballot(pending);
            if (!pm) break;
            const int first = (int)__builtin_ctzll(pm);
            const uint32_t flo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)pos, first);
            const uint32_t fhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)pos >> 32), first);
            F = (int64_t)(((uint64_t)fhi << 32) | flo);
        }
        base += total;
    }
}

// ---------------------------------------------------------------- 5. the chunks' windows
// A chunk's window is its predecessor's last 32 KiB with the markers in them replaced from the predecessor's window: a chain
// through all chunks -- 3.5 us a link on one CU, 18 ms for 5 000 chunks.  So the chain is walked in SEGMENTS of chunks, three
// launches:
//   k_gz_windows<uint16_t>  every segment by itself (a workgroup each), from the identity: what each place of the window behind
//                           the segment holds -- a byte, or a marker into the window in FRONT of the segment (a map);
//   k_gz_seg_windows        one workgroup, segment by segment: the window in front of each, through the maps;
//   k_gz_windows<uint8_t>   every segment by itself again, from its real window: each chunk's window to d_win.
// T: what a place of the window holds (uint8_t a byte; uint16_t a symbol).  LDS: two windows of T.
template <typename T>
__global__ __launch_bounds__(1024) void k_gz_windows(const uint16_t *sym, const uint64_t *sym_off, const ChunkOut *res, uint32_t nchunks, uint32_t seg_len,
                                                     const T *seg_in, T *seg_out, uint8_t *d_win) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    T *W0 = reinterpret_cast<T *>(lds_raw);
    constexpr uint32_t PER = WINDOW / 4096u;                             // groups of four consecutive symbols per thread
    constexpr bool BYTES = sizeof(T) == 1;
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * seg_len, c1 = c0 + seg_len < nchunks ? c0 + seg_len : nchunks;
    if (c0 >= nchunks) return;
    if (seg_in) {
        const T *src = seg_in + (size_t)blockIdx.x * WINDOW;
        for (uint32_t p = tid; p < WINDOW; p += 1024u) W0[p] = src[p];
    } else {
        for (uint32_t p = tid; p < WINDOW; p += 1024u) W0[p] = (T)(0x8000u | p);
    }
    // (off: where the chunk's last 32 KiB of symbols begin -- read a step ahead of the loads that need it; NONE: a short chunk)
    auto tail_at = [&](uint32_t c) -> uint64_t {
        if (c >= c1) return NONE;
        const uint64_t n = res[c].out_len;
        return n >= WINDOW ? sym_off[c] + n - WINDOW : NONE;
    };
    // a chunk's last 32 KiB of symbols, this thread's eight groups of four (8 bytes a load, at any 2-byte boundary); a chunk
    // shorter than that: markers into what is left of the window before it
    auto tail = [&](uint32_t c, uint64_t off, uint64_t (&v)[PER]) {
        if (off != NONE) {
            const uint16_t *t0 = sym + off;
#pragma unroll
            for (uint32_t i = 0; i < PER; i++) __builtin_memcpy(&v[i], t0 + (i * 1024u + tid) * 4u, 8);
        } else {
            const uint64_t n = res[c].out_len;
            const uint16_t *s = sym + sym_off[c];
#pragma unroll
            for (uint32_t i = 0; i < PER; i++) {
                uint64_t x = 0;
                for (uint32_t q = 0; q < 4u; q++) {
                    const uint32_t p = (i * 1024u + tid) * 4u + q;
                    const int64_t o = (int64_t)n - (int64_t)WINDOW + (int64_t)p;
                    const uint16_t y = o >= 0 ? s[o] : (uint16_t)(0x8000u | (uint32_t)((int64_t)p + (int64_t)n));   // (p + n < 32768: a place in the old window)
                    x |= (uint64_t)y << (16u * q);
                }
                v[i] = x;
            }
        }
    };
    // one chunk: its window out, the window behind it from its symbols `v` -- while the symbols of the chunk after next come (two
    // sets of registers taking turns: a set is not touched between its loads and its turn)
    uint32_t cur = 0;
    uint64_t off_next = NONE;                                            // tail_at(chunk after next), known a step early
    auto step = [&](uint32_t c, uint64_t (&v)[PER]) {
        const uint64_t off_use = off_next;
        off_next = tail_at(c + 3u);
        const T *Wc = W0 + cur * WINDOW;
        T *Wn = W0 + (cur ^ 1u) * WINDOW;
        if (BYTES && d_win) {
            uint8_t *wout = d_win + (size_t)c * WINDOW;
            for (uint32_t p = tid * 16u; p < WINDOW; p += 1024u * 16u)
                *reinterpret_cast<uint4 *>(wout + p) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(Wc) + p);
        }
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) {
            const uint64_t x = v[i];
            const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
            const uint32_t at = (i * 1024u + tid) * 4u;
            if (BYTES) {
                uint32_t y = __builtin_amdgcn_perm(xh, xl, 0x06040200u);  // the four symbols' low bytes
                if ((xl | xh) & 0x80008000u) {                            // (markers are the exception: only they read the old window --
                    uint32_t wb[4];                                       // four reads in flight, then four selects)
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++) wb[q] = (uint32_t)Wc[(uint32_t)(x >> (16u * q)) & 0x7FFFu];
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++)
                        if ((x >> (16u * q)) & 0x8000u) y = (y & ~(0xFFu << (8u * q))) | (wb[q] << (8u * q));
                }
                *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(Wn) + at) = y;
            } else {
                uint64_t y = x;
                if ((xl | xh) & 0x80008000u) {
                    uint32_t wb[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++) wb[q] = (uint32_t)Wc[(uint32_t)(x >> (16u * q)) & 0x7FFFu];
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++)
                        if ((x >> (16u * q)) & 0x8000u) y = (y & ~(0xFFFFull << (16u * q))) | ((uint64_t)wb[q] << (16u * q));
                }
                *reinterpret_cast<uint64_t *>(reinterpret_cast<uint16_t *>(Wn) + at) = y;
            }
        }
        if (c + 2u < c1) tail(c + 2u, off_use, v);
        // (the workgroup's LDS writes, not its loads from memory: __syncthreads would wait for those too)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur ^= 1u;
    };
    uint64_t va[PER], vb[PER];
    tail(c0, tail_at(c0), va);
    if (c0 + 1u < c1) tail(c0 + 1u, tail_at(c0 + 1u), vb);
    off_next = tail_at(c0 + 2u);
    __syncthreads();
    for (uint32_t c = c0; c < c1; c += 2u) {
        step(c, va);
        if (c + 1u < c1) step(c + 1u, vb);
    }
    if (seg_out) {
        T *dst = seg_out + (size_t)blockIdx.x * WINDOW;
        const T *Wc = W0 + cur * WINDOW;
        for (uint32_t p = tid; p < WINDOW; p += 1024u) dst[p] = Wc[p];
    }
}

// the window in front of every segment (seg_win + 32768 s), through the segments' maps, in order; carry: in, the window in front
// of the first chunk; out, the window behind the last
__global__ __launch_bounds__(1024) void k_gz_seg_windows(const uint16_t *maps, uint32_t nseg, uint8_t *seg_win, uint8_t *carry) {
    __shared__ uint8_t W[2][WINDOW];
    const uint32_t tid = threadIdx.x;
    for (uint32_t p = tid; p < WINDOW; p += 1024u) W[0][p] = carry[p];
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t sgm = 0; sgm < nseg; sgm++) {
        const uint16_t *mp = maps + (size_t)sgm * WINDOW;
        uint8_t *wout = seg_win + (size_t)sgm * WINDOW;
        for (uint32_t p = tid; p < WINDOW; p += 1024u) {
            wout[p] = W[cur][p];
            const uint32_t y = mp[p];
            W[cur ^ 1u][p] = (y & 0x8000u) ? W[cur][y & 0x7FFFu] : (uint8_t)y;
        }
        __syncthreads();
        cur ^= 1u;
    }
    for (uint32_t p = tid; p < WINDOW; p += 1024u) carry[p] = W[cur][p];
}

// the segments' maps composed, in order, into ONE map: what each place of the window behind the last segment holds in terms of
// the window in front of the first (a rank's share of a file that is decoded by several devices: multi.count_file_sharded)
__global__ __launch_bounds__(1024) void k_gz_compose_maps(const uint16_t *maps, uint32_t nseg, uint16_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint16_t *W = reinterpret_cast<uint16_t *>(lds_raw);                 // two windows of symbols
    const uint32_t tid = threadIdx.x;
    for (uint32_t p = tid; p < WINDOW; p += 1024u) W[p] = (uint16_t)(0x8000u | p);
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t sgm = 0; sgm < nseg; sgm++) {
        const uint16_t *mp = maps + (size_t)sgm * WINDOW;
        const uint16_t *Wc = W + cur * WINDOW;
        uint16_t *Wn = W + (cur ^ 1u) * WINDOW;
        for (uint32_t p = tid; p < WINDOW; p += 1024u) {
            const uint32_t y = mp[p];
            Wn[p] = (y & 0x8000u) ? Wc[y & 0x7FFFu] : (uint16_t)y;
        }
        __syncthreads();
        cur ^= 1u;
    }
    for (uint32_t p = tid; p < WINDOW; p += 1024u) out[p] = W[cur * WINDOW + p];
}

}  // namespace tdgz2
