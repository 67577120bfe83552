// A raw-DEFLATE (RFC 1951) decoder written to run ONE STREAM PER LANE on the GPU -- the BGZF members of a
// .gz FASTQ (bgzip: independent gzip members of at most 64 KiB), thousands at once -- and, from the same source,
// on the host (td_inflate_raw_host: how it is tested here against zlib without a GPU).  SURVEY 8f-2's second
// option; the reference reads .gz input through gzip.open (tagdigger_fun.py:240-243).
//
// Per stream: an 8-bit primary table for literal/length codes and a 7-bit one for distance codes (768 bytes:
// LDS on the device), the canonical-code arrays for the rare longer codes and the code lengths while a block's
// tables are built (global scratch, 1 KiB per stream), a 64-bit bit buffer refilled four bytes at a time.  The
// decoder is a small state machine -- one block header, one symbol, or eight bytes of a copy per step -- so that
// the lanes of a wave, each in its own stream, stay busy whatever their neighbours are doing; every step makes
// progress or fails, and the number of steps is bounded by the stream's sizes.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TDI_FN __host__ __device__ __forceinline__
#else
#define TDI_FN inline
#endif

namespace tdinf {

constexpr int LBITS = 8, DBITS = 7;
constexpr int TABLE_U16 = (1 << LBITS) + (1 << DBITS);        // primary tables per stream (uint16 entries)
constexpr int SCRATCH_BYTES = 1024;                           // per stream: lengths[320] | lsym u16[288] | dsym u16[32] | lcnt u16[16] | dcnt u16[16]
constexpr uint16_t E_SLOW = 0xFFFFu;                          // primary entry: code longer than the index

enum { ST_HEADER = 0, ST_SYMBOL = 1, ST_COPY = 2, ST_STORED = 3, ST_DONE = 4, ST_ERROR = 5 };
enum { ERR_NONE = 0, ERR_BTYPE = 1, ERR_STORED = 2, ERR_LENGTHS = 3, ERR_CODE = 4, ERR_DIST = 5, ERR_OVERRUN = 6, ERR_INPUT = 7,
       ERR_SIZE = 8, ERR_STEPS = 9 };

struct Stream {
    const uint8_t *in;      // compressed bytes (readable up to in_len + 8)
    uint32_t in_len;
    uint8_t *out;
    uint32_t out_len;       // exact size the stream must inflate to (the member's ISIZE)
    uint16_t *tab;          // TABLE_U16 entries, entry e at tab[e * tstride] (device: the 64 streams of a wave interleave
    uint32_t tstride;       //   their tables in LDS, entry by entry, so that equal indices fall into different banks)
    uint8_t *scratch;       // SCRATCH_BYTES
    // state
    uint64_t bb;
    uint32_t bc, ipos, opos;
    uint32_t state, err, last_block;
    uint32_t copy_len, copy_dist;
    uint64_t pat;           // COPY with a distance below 8: the eight bytes written last (the next eight follow from them)
    uint32_t ahead;         // the input word after the bit buffer's, loaded one refill early (its latency runs under the decoding)
};

TDI_FN uint32_t load32(const uint8_t *p) {                    // (any alignment: one load on the device)
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
TDI_FN uint64_t load64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
TDI_FN void store64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
// ipos = the input consumed into the bit buffer; `ahead` holds in[ipos .. ipos + 4).  No load ever reaches beyond
// in[in_len + 8): a position past the payload's end (a truncated or hostile stream: a dynamic block's header alone
// consumes up to ~570 bytes between two of step()'s checks) re-reads the last allowed word, and step() then fails
// the stream with ERR_INPUT.
TDI_FN uint32_t load_ahead(const Stream &s) {
    const uint32_t lim = s.in_len + 4u;
    return load32(s.in + (s.ipos < lim ? s.ipos : lim));
}
TDI_FN void refill(Stream &s) {
    if (s.bc <= 32) {
        s.bb |= (uint64_t)s.ahead << s.bc;
        s.ipos += 4; s.bc += 32;
        s.ahead = load_ahead(s);
    }
}
TDI_FN uint32_t peek(const Stream &s, uint32_t n) { return (uint32_t)s.bb & ((1u << n) - 1u); }
TDI_FN void drop(Stream &s, uint32_t n) { s.bb >>= n; s.bc -= n; }
TDI_FN uint32_t take(Stream &s, uint32_t n) { const uint32_t v = peek(s, n); drop(s, n); return v; }
TDI_FN uint32_t rev(uint32_t code, uint32_t len) {            // the low `len` bits of code, reversed
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Canonical Huffman tables from `n` code lengths: primary[1 << bits] (entry = symbol << 4 | length, E_SLOW for the
// prefixes of longer codes, 0 = no code), sym[] = symbols in canonical order, cnt[len] = codes of that length.
// Returns false for an over-subscribed set (incomplete sets are allowed, as zlib allows a single distance code).
TDI_FN bool build(const uint8_t *lens, uint32_t n, uint16_t *primary, uint32_t ts, uint32_t bits, uint16_t *sym, uint16_t *cnt) {
    for (uint32_t l = 0; l < 16; l++) cnt[l] = 0;
    for (uint32_t i = 0; i < n; i++) cnt[lens[i]]++;
    cnt[0] = 0;
    int32_t left = 1;
    for (uint32_t l = 1; l < 16; l++) { left <<= 1; left -= (int32_t)cnt[l]; if (left < 0) return false; }
    uint32_t offs[16], next[16];
    offs[1] = 0; next[1] = 0;
    for (uint32_t l = 1; l < 15; l++) { offs[l + 1] = offs[l] + cnt[l]; next[l + 1] = (next[l] + cnt[l]) << 1; }
    for (uint32_t i = 0; i < (1u << bits); i++) primary[i * ts] = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = lens[i];
        if (!l) continue;
        sym[offs[l]++] = (uint16_t)i;
        const uint32_t code = next[l]++;
        if (l <= bits) {
            const uint32_t r = rev(code, l);
            for (uint32_t k = r; k < (1u << bits); k += 1u << l) primary[k * ts] = (uint16_t)((i << 4) | l);
        } else {
            primary[rev(code >> (l - bits), bits) * ts] = E_SLOW;
        }
    }
    return true;
}
// a symbol whose code is longer than the primary index: bit by bit against the canonical arrays
TDI_FN int32_t slow_symbol(Stream &s, const uint16_t *sym, const uint16_t *cnt) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l < 16; l++) {
        code |= (uint32_t)(s.bb >> (l - 1)) & 1u;
        const uint32_t c = cnt[l];
        if (code < first + c) { drop(s, l); return (int32_t)sym[index + (code - first)]; }
        index += c; first += c;
        first <<= 1; code <<= 1;
    }
    return -1;
}
TDI_FN int32_t symbol(Stream &s, const uint16_t *primary, uint32_t bits, const uint16_t *sym, const uint16_t *cnt) {
    const uint16_t e = primary[peek(s, bits) * s.tstride];
    if (e == E_SLOW) return slow_symbol(s, sym, cnt);
    if (e == 0) return -1;
    drop(s, e & 15u);
    return (int32_t)(e >> 4);
}

// one block header: stored -> ST_STORED; fixed / dynamic -> tables built, ST_SYMBOL
TDI_FN void header(Stream &s) {
    uint8_t *lens = s.scratch;                                        // [320]
    uint16_t *lsym = reinterpret_cast<uint16_t *>(s.scratch + 320);   // [288]
    uint16_t *dsym = lsym + 288;                                      // [32]
    uint16_t *lcnt = dsym + 32, *dcnt = lcnt + 16;
    uint16_t *ltab = s.tab, *dtab = s.tab + (1 << LBITS) * s.tstride;
    refill(s);
    s.last_block = take(s, 1);
    const uint32_t type = take(s, 2);
    if (type == 0) {
        drop(s, s.bc & 7u);                                           // to the byte boundary
        refill(s);
        const uint32_t len = take(s, 16), nlen = take(s, 16);
        if ((len ^ nlen) != 0xFFFFu) { s.state = ST_ERROR; s.err = ERR_STORED; return; }
        // the bytes still in the bit buffer go back to the input
        s.ipos -= s.bc >> 3; s.bb = 0; s.bc = 0;
        s.ahead = load_ahead(s);
        s.copy_len = len;
        s.state = len ? ST_STORED : (s.last_block ? ST_DONE : ST_HEADER);
        return;
    }
    if (type == 3) { s.state = ST_ERROR; s.err = ERR_BTYPE; return; }
    uint32_t nlit, ndist;
    if (type == 1) {
        nlit = 288; ndist = 30;
        for (uint32_t i = 0; i < 144; i++) lens[i] = 8;
        for (uint32_t i = 144; i < 256; i++) lens[i] = 9;
        for (uint32_t i = 256; i < 280; i++) lens[i] = 7;
        for (uint32_t i = 280; i < 288; i++) lens[i] = 8;
        for (uint32_t i = 0; i < 30; i++) lens[288 + i] = 5;
    } else {
        nlit = take(s, 5) + 257; ndist = take(s, 5) + 1;
        const uint32_t ncl = take(s, 4) + 4;
        if (nlit > 286 || ndist > 30) { s.state = ST_ERROR; s.err = ERR_LENGTHS; return; }
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19];
        for (uint32_t i = 0; i < 19; i++) cl[i] = 0;
        for (uint32_t i = 0; i < ncl; i++) { refill(s); cl[order[i]] = (uint8_t)take(s, 3); }
        // the code-length code: a 7-bit table in the distance table's place (built afterwards)
        uint16_t csym[19], ccnt[16];
        if (!build(cl, 19, dtab, s.tstride, 7, csym, ccnt)) { s.state = ST_ERROR; s.err = ERR_LENGTHS; return; }
        uint32_t i = 0;
        while (i < nlit + ndist) {
            refill(s);
            const int32_t c = symbol(s, dtab, 7, csym, ccnt);
            if (c < 0) { s.state = ST_ERROR; s.err = ERR_LENGTHS; return; }
            if (c < 16) { lens[i++] = (uint8_t)c; continue; }
            uint32_t val = 0, rep;
            if (c == 16) { if (i == 0) { s.state = ST_ERROR; s.err = ERR_LENGTHS; return; } val = lens[i - 1]; rep = 3 + take(s, 2); }
            else if (c == 17) rep = 3 + take(s, 3);
            else rep = 11 + take(s, 7);
            if (i + rep > nlit + ndist) { s.state = ST_ERROR; s.err = ERR_LENGTHS; return; }
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) { s.state = ST_ERROR; s.err = ERR_LENGTHS; return; }       // no end-of-block code
        // distance lengths follow the literal/length ones: move them to their place
        for (uint32_t k = ndist; k-- > 0;) lens[288 + k] = lens[nlit + k];           // (backwards: the ranges overlap)
        for (uint32_t k = nlit; k < 288; k++) lens[k] = 0;
    }
    if (!build(lens, nlit, ltab, s.tstride, LBITS, lsym, lcnt) || !build(lens + 288, ndist, dtab, s.tstride, DBITS, dsym, dcnt)) {
        s.state = ST_ERROR; s.err = ERR_LENGTHS; return;
    }
    s.state = ST_SYMBOL;
}

// one literal, or one length + distance pair (the copy itself is the COPY state's)
TDI_FN void step_symbol(Stream &s) {
    const uint16_t *lsym = reinterpret_cast<const uint16_t *>(s.scratch + 320), *dsym = lsym + 288;
    const uint16_t *lcnt = dsym + 32, *dcnt = lcnt + 16;
    refill(s);
    const int32_t sy = symbol(s, s.tab, LBITS, lsym, lcnt);
    if (sy < 0) { s.state = ST_ERROR; s.err = ERR_CODE; return; }
    if (sy < 256) {
        if (s.opos >= s.out_len) { s.state = ST_ERROR; s.err = ERR_OVERRUN; return; }
        s.out[s.opos++] = (uint8_t)sy;
        return;
    }
    if (sy == 256) { s.state = s.last_block ? ST_DONE : ST_HEADER; return; }
    if (sy > 285) { s.state = ST_ERROR; s.err = ERR_CODE; return; }
    const uint32_t li = (uint32_t)sy - 257;
    // length: base and extra bits by formula (no tables: they would live in constant memory per lane)
    uint32_t lext = li < 8 ? 0u : (li - 4u) >> 2, lbase = li < 8 ? li + 3u : ((4u + (li & 3u)) << lext) + 3u;
    if (li == 28) { lext = 0; lbase = 258; }
    const uint32_t length = lbase + take(s, lext);
    refill(s);
    const int32_t ds = symbol(s, s.tab + (1 << LBITS) * s.tstride, DBITS, dsym, dcnt);
    if (ds < 0 || ds > 29) { s.state = ST_ERROR; s.err = ERR_DIST; return; }
    const uint32_t dext = ds < 4 ? 0u : ((uint32_t)ds - 2u) >> 1, dbase = ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << dext) + 1u;
    refill(s);
    const uint32_t dist = dbase + take(s, dext);
    if (dist > s.opos) { s.state = ST_ERROR; s.err = ERR_DIST; return; }
    if (s.opos + length > s.out_len) { s.state = ST_ERROR; s.err = ERR_OVERRUN; return; }
    s.copy_len = length; s.copy_dist = dist;
    if (dist < 8u) {
        // the eight bytes "before" the copy's first byte, as if the pattern had been running: byte k of pat = out[opos - 8 + k]
        // for the last `dist` of them, the others by periodicity
        const uint64_t last = load64(s.out + s.opos - dist) ;              // (its first `dist` bytes are what counts)
        uint64_t pat = 0;
        for (uint32_t k = 0; k < 8u; k++) {
            const uint32_t from = (k + 8u * dist - 8u) % dist;             // out[opos - 8 + k] == out[opos - dist + ((k - 8) mod dist)]
            pat |= ((last >> (8u * from)) & 0xFFull) << (8u * k);
        }
        s.pat = pat;
    }
    s.state = ST_COPY;
}

TDI_FN void step(Stream &s) {
    if (s.ipos > s.in_len + 8u) { s.state = ST_ERROR; s.err = ERR_INPUT; return; }
    switch (s.state) {
    case ST_HEADER: header(s); break;
    case ST_SYMBOL: step_symbol(s); break;
    case ST_COPY: {
        const uint32_t n = s.copy_len < 8u ? s.copy_len : 8u;
        uint8_t *o = s.out + s.opos;
        uint64_t v;
        if (s.copy_dist >= 8u) {
            v = load64(o - s.copy_dist);                                   // source and destination do not overlap within 8 bytes
        } else {
            // byte k of the next eight = the byte `dist` before it: from the previous eight, or from these
            const uint32_t d = s.copy_dist;
            v = 0;
            for (uint32_t k = 0; k < 8u; k++) {
                const uint64_t b = k < d ? (s.pat >> (8u * (8u - d + k))) & 0xFFull : (v >> (8u * (k - d))) & 0xFFull;
                v |= b << (8u * k);
            }
            s.pat = v;
        }
        if (n == 8u) store64(o, v);
        else for (uint32_t k = 0; k < n; k++) o[k] = (uint8_t)(v >> (8u * k));     // (exact: the next member's output follows)
        s.opos += n; s.copy_len -= n;
        if (s.copy_len == 0) s.state = ST_SYMBOL;
        break;
    }
    case ST_STORED: {
        uint32_t n = s.copy_len < 8u ? s.copy_len : 8u;
        if (s.opos + n > s.out_len || s.ipos + n > s.in_len) { s.state = ST_ERROR; s.err = ERR_OVERRUN; return; }
        s.copy_len -= n;
        while (n--) s.out[s.opos++] = s.in[s.ipos++];
        if (s.copy_len == 0) { s.ahead = load_ahead(s); s.state = s.last_block ? ST_DONE : ST_HEADER; }
        break;
    }
    default: break;
    }
}

// the whole stream; 0 or an ERR_ code.  Every step consumes input or produces output (or ends the stream) except a
// block header, and a block holds at least its end code: 4 (in + out) + 64 steps are more than any valid stream takes.
TDI_FN uint32_t run(Stream &s) {
    s.bb = 0; s.bc = 0; s.ipos = 0; s.opos = 0; s.state = ST_HEADER; s.err = ERR_NONE; s.last_block = 0;
    s.copy_len = 0; s.copy_dist = 0; s.pat = 0;
    s.ahead = load32(s.in);
    uint64_t budget = 4ull * ((uint64_t)s.in_len + s.out_len) + 64;
    while (s.state < ST_DONE) {
        if (budget-- == 0) { s.state = ST_ERROR; s.err = ERR_STEPS; break; }
        step(s);
    }
    if (s.state == ST_DONE && s.opos != s.out_len) { s.state = ST_ERROR; s.err = ERR_SIZE; }
    return s.state == ST_DONE ? (uint32_t)ERR_NONE : s.err;
}


#ifdef __HIPCC__
// One BGZF member per lane, one wave per workgroup (64 members; LDS: their 64 x 768 bytes of primary tables).
struct Member {
    uint64_t in_off;        // the member's deflate payload in the batch's compressed bytes
    uint64_t out_off;       // where it inflates to in the batch's output
    uint32_t in_len, out_len;
    uint32_t crc;           // CRC-32 of the inflated bytes (the member's trailer)
    uint32_t pad;
};
// CRC-32 of a member's output, four bytes a step (slicing-by-4: T[k][b] = the CRC of byte b followed by k zero
// bytes; the four tables, 4 KiB, sit in LDS beside the Huffman tables)
__device__ __forceinline__ uint32_t crc32_lds(const uint8_t *p, uint32_t n, const uint32_t *T) {
    uint32_t c = 0xFFFFFFFFu, i = 0;
    for (; i + 4 <= n; i += 4) {
        c ^= load32(p + i);
        c = T[768 + (c & 0xFFu)] ^ T[512 + ((c >> 8) & 0xFFu)] ^ T[256 + ((c >> 16) & 0xFFu)] ^ T[c >> 24];
    }
    for (; i < n; i++) c = T[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return ~c;
}
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *in, uint8_t *out, const Member *mem, uint32_t n, uint8_t *scratch,
                                                     uint32_t *status, const uint32_t *crc_tables, uint32_t check_crc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *L_crc = reinterpret_cast<uint32_t *>(lds + 64 * TABLE_U16 * 2);
    const uint32_t lane = threadIdx.x, i = blockIdx.x * 64u + lane;
    if (check_crc) {
        for (uint32_t k = lane; k < 1024u; k += 64u) L_crc[k] = crc_tables[k];
        __syncthreads();
    }
    if (i >= n) return;
    const Member m = mem[i];
    Stream s{};
    s.in = in + m.in_off; s.in_len = m.in_len; s.out = out + m.out_off; s.out_len = m.out_len;
    s.tab = reinterpret_cast<uint16_t *>(lds) + lane; s.tstride = 64;
    s.scratch = scratch + (size_t)i * SCRATCH_BYTES;
    uint32_t rc = run(s);
    if (rc == ERR_NONE && check_crc && crc32_lds(s.out, m.out_len, L_crc) != m.crc) rc = 100;
    status[i] = rc;
}
#endif

}  // namespace tdinf
