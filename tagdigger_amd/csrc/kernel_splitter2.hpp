// k_split2: the splitter's per-read branch (reference tagdigger_fun.py:1251-1283 and the loop at
// :1328-1363) on the tile machinery of k_fast2 -- the tile's raw bytes staged in LDS, terminator masks and per-wave
// lists of line starts built without a workgroup barrier, then ONE lane per sequence line (full lanes: the j-th
// sequence line of the tile goes to thread j), which reads its line from LDS in unaligned 16-byte pieces:
//   * barcode + cut site: the first 32 bases packed (pack16_ascii) and looked up in the LDS directory;
//   * the first full restriction site behind them (str.find, :1263-1268): per piece one 16-bit mask per base
//     letter, kept with the previous piece's in a 32-bit history word; a site ends where the shifted masks of its
//     letters all have a bit -- sixteen text positions per handful of instructions instead of one;
//   * else the adapter that runs off the read's end (:1269-1280), compared backwards from the read's last byte.
// Line numbers are exact (the caller's prefix of terminators per tile), so no vote and no fix-up pass.  What does
// not fit the fast form -- the buffer's first and last tiles, tiles holding bytes >= 0x80 or more line starts than a
// wave's list holds, a line that ends behind the staged window -- goes through k_split's per-thread walk over global
// memory (split_line), line by line or tile by tile.
#pragma once
#include "kernel_fast2.hpp"
#include "kernel_splitter.hpp"

namespace tdk {

constexpr uint32_t SPLIT2_HALO = 512;       // bytes staged behind the tile (multiple of 64)

// One sequence line whose bytes are raw[s0 .. nx - 1) (nx: start of the next line; all inside the staged window).
#ifdef TD_PHASE_PROF
#define TD_LSTAMP(i) do { if (pacc) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long now_ = __builtin_amdgcn_s_memtime(); pacc[i] += now_ - *plast; *plast = now_; } } while (0)
#else
#define TD_LSTAMP(i) do {} while (0)
#endif
__device__ __forceinline__ int2 split_line_lds(const SplitParams &p, const unsigned long long *L_bval, const uint32_t *L_bmeta,
                                               const uint16_t *L_bdir, const uint8_t *raw, uint32_t s0, uint32_t nx,
                                               unsigned long long *pacc = nullptr, unsigned long long *plast = nullptr) {
    TD_LSTAMP(6);    // line selection (list reads) before the call
    // the terminator (one byte, or "\r\n") off the end, then line.strip(): blanks off both ends
    uint32_t s = s0, e = nx - 1u;
    if (e > s && raw[e] == 0x0Au && raw[e - 1] == 0x0Du) e--;
    while (s < e && is_blank(raw[s])) s++;
    while (e > s && is_blank(raw[e - 1])) e--;
    const uint32_t len = e - s;
    const uint8_t *src = raw + s;
    TD_LSTAMP(7);    // terminator + strip (dependent byte reads)

    // ---- barcode + cut site: the first (up to 32) valid bases, packed like the index
    const uint4 q0 = lds_read16(src), q1 = lds_read16(src + 16);
    const uint2 c0 = pack16_ascii(q0.x, q0.y, q0.z, q0.w), c1 = pack16_ascii(q1.x, q1.y, q1.z, q1.w);
    const unsigned long long Kfull = ((unsigned long long)c0.x << 32) | c1.x;
    const uint32_t invalid = (c0.y & 0xFFFFu) | (c1.y << 16);
    uint32_t nvalid = invalid ? (uint32_t)__builtin_ctz(invalid) : 32u;
    if (nvalid > len) nvalid = len;
    const unsigned long long K = nvalid >= 32 ? Kfull : nvalid == 0 ? 0ull : Kfull & (~0ull << (64 - 2 * nvalid));
    uint32_t ci = L_bdir[(uint32_t)(K >> (64 - 2 * BDIR_BASES))];
    uint32_t meta = 0;
    bool hit = false;
    if (ci != 0xFFFFu) {
        for (;;) {
            const uint32_t m = L_bmeta[ci];
            const uint32_t l = m & 63u;
            if (l <= nvalid && ((K ^ L_bval[ci]) >> (64u - 2u * l)) == 0) { meta = m; hit = true; break; }
            if (m & BMETA_LAST) break;
            ci++;
        }
    }
    TD_LSTAMP(8);    // pack 2 pieces + barcode walk
    if (!hit) return make_int2(-1, 999);
    const uint32_t bar = meta >> 16;
    const uint32_t start = ((meta >> 6) & 63u) + p.cutlen;           // searchstart = len(barcode) + len(cutsite)
    // The adapter entries this read could end with -- those of its barcode that end with its last THREE characters: a
    // group of eight compact entries (64 bytes) found by address alone, so it is requested NOW and arrives under the
    // search for the restriction sites.  (A second- or third-last byte that is no base leaves the shorter entries,
    // which every group of their characters holds; round 2 grouped by two characters: up to sixteen entries of 16 bytes,
    // fetched four at a time, one dependent round trip after the other.)
    const uint32_t lastc = len ? (uint32_t)raw[e - 1] & 0xDFu : 0u;
    const uint32_t lcode = (lastc >> 1) & 3u;
    const bool tail_is_base = len != 0 && lastc == ((0x47544341u >> (8 * lcode)) & 0xFFu) && !(p.dbg & 128u);
    const uint32_t prevc = len >= 2u ? (uint32_t)raw[e - 2] & 0xDFu : 0x41u;
    const uint32_t thirdc = len >= 3u ? (uint32_t)raw[e - 3] & 0xDFu : 0x41u;
    const uint4 *grp = reinterpret_cast<const uint4 *>(p.entries8 + (size_t)(64u * bar + 16u * lcode + 4u * ((prevc >> 1) & 3u) + ((thirdc >> 1) & 3u)) * 8u);
    uint4 ent[4];
#pragma unroll
    for (int u = 0; u < 4; u++) ent[u] = tail_is_base ? grp[u] : make_uint4(0u, 0u, 0u, 0u);      // two entries each: {key, meta}

    // ---- first full restriction site at or after `start` (str.find)
    uint32_t rs0 = 0xFFFFFFFFu, rs1 = 0xFFFFFFFFu;
    if (start <= len && !(p.dbg & 64u)) {
        if (p.site0_len == 0) rs0 = start;                           // (an empty site is found at `start` itself)
        if (p.site1_len == 0) rs1 = start;
        // SP[site][letter]: bit `back` set where the site's character `back` places before its last is that letter (uniform)
        uint32_t SP[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
        for (int which = 0; which < 2; which++) {
            const uint32_t L = which ? p.site1_len : p.site0_len;
            const unsigned long long site = which ? p.site1 : p.site0;          // last character in the low byte
            for (uint32_t back = 0; back < L; back++) {
                const uint32_t code = ((uint32_t)(site >> (8u * back)) >> 1) & 3u;   // A C T G = 0 1 2 3
#pragma unroll
                for (int c = 0; c < 4; c++) SP[which][c] |= code == (uint32_t)c ? 1u << back : 0u;
            }
        }
        // Round 3: the letter masks of 128 text positions at a time (eight pieces, bit b of Hd[letter][b >> 5]), the
        // positions outside [start, len) cleared once, then per site ONE pass over its characters: the masks shifted
        // left by the character's distance from the site's end (v_alignbit across the four words) and ANDed -- what is
        // left are the positions where the site ends.  (Round 2 matched piece by piece with a 16-bit history: seven
        // rounds of LDS read -> masks -> per-character loops -> branches, each waiting for the one before; here the
        // pieces' mask chains are independent of each other and the character loops run once per line.)
        for (uint32_t kw = start >> 4; 16u * kw < len && (rs0 == 0xFFFFFFFFu || rs1 == 0xFFFFFFFFu); kw += 7u) {
            uint32_t Hd[4][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t k = kw + (uint32_t)i;
                if (16u * k < len) {
                    uint4 v = lds_read16(src + 16u * k);
                    v.x &= 0xDFDFDFDFu; v.y &= 0xDFDFDFDFu; v.z &= 0xDFDFDFDFu; v.w &= 0xDFDFDFDFu;      // (a == A for the four letters)
                    const uint32_t m[4] = {eq_mask16_ascii(v, 0x41414141u, 0x7F7F7F7Fu), eq_mask16_ascii(v, 0x43434343u, 0x7F7F7F7Fu),
                                           eq_mask16_ascii(v, 0x54545454u, 0x7F7F7F7Fu), eq_mask16_ascii(v, 0x47474747u, 0x7F7F7F7Fu)};
#pragma unroll
                    for (int c = 0; c < 4; c++) Hd[c][i >> 1] |= m[c] << (16 * (i & 1));
                }
            }
            // text positions that take part, relative to the window: start <= position < len
            const uint32_t rel_lo = start > 16u * kw ? start - 16u * kw : 0u, rel_hi = len - 16u * kw;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                auto below = [&](uint32_t x) -> uint32_t {             // bits of word d below position x
                    return x <= 32u * d ? 0u : x >= 32u * d + 32u ? 0xFFFFFFFFu : (1u << (x - 32u * d)) - 1u;
                };
                const uint32_t R = below(rel_hi) & ~below(rel_lo);
#pragma unroll
                for (int c = 0; c < 4; c++) Hd[c][d] &= R;
            }
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const uint32_t L = which ? p.site1_len : p.site0_len;
                if (L == 0) continue;                                   // (an empty site was settled above)
                uint32_t M[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    uint32_t places = SP[which][c];
                    while (places) {
                        const uint32_t back = (uint32_t)__builtin_ctz(places);
                        places &= places - 1u;
                        if (back == 0) {
#pragma unroll
                            for (int d = 0; d < 4; d++) M[d] &= Hd[c][d];
                        } else {
#pragma unroll
                            for (int d = 3; d >= 0; d--)
                                M[d] &= __builtin_amdgcn_alignbit(Hd[c][d], d ? Hd[c][d - 1] : 0u, 32u - back);
                        }
                    }
                }
                uint32_t pos = 0xFFFFFFFFu;
#pragma unroll
                for (int d = 3; d >= 0; d--) pos = M[d] ? 32u * d + (uint32_t)__builtin_ctz(M[d]) : pos;
                if (pos != 0xFFFFFFFFu) {
                    const uint32_t at = 16u * kw + pos + 1u - L;     // where that site starts
                    if (which) { if (rs1 == 0xFFFFFFFFu) rs1 = at; } else { if (rs0 == 0xFFFFFFFFu) rs0 = at; }
                }
            }
        }
    }
    TD_LSTAMP(9);    // entry loads issued + site search
    if (rs0 != 0xFFFFFFFFu || rs1 != 0xFFFFFFFFu) {
        uint32_t cut;
        if (rs1 == 0xFFFFFFFFu) cut = rs0 + p.site0_len;
        else if (rs0 == 0xFFFFFFFFu) cut = rs1 + p.site1_len;
        else if (rs0 < rs1) cut = rs0 + p.site0_len;
        else cut = rs1 + p.site1_len;
        return make_int2((int)bar, (int)cut);
    }
    // ---- no full site: does the read END with the start of an adapter?  (the reference walks a trie over the reversed
    // read; its stored set is prefix-free, so at most one entry matches.)  The read's last four characters (case
    // folded) against every entry's: an entry whose tail differs is dropped on one compare of registers; one that
    // agrees is compared eight bytes at a time, read against pool, all of its pool words requested together.
    // (no `continue` / `return` inside these nested divergent loops: hipcc 7.2 -O3 miscompiled that, see split_line)
    if (!tail_is_base) return make_int2((int)bar, 999);
    uint32_t t4 = 0;
    {
        const uint32_t n4 = len < 4u ? len : 4u;
        uint32_t w;
        __builtin_memcpy(&w, raw + e - n4, 4);                           // bytes e - n4 .. e - n4 + 3
        t4 = (w << (8u * (4u - n4))) & 0xDFDFDFDFu;                      // the last character in the top byte
    }
    // Of the group's eight entries in registers, a lane takes the FIRST one (the longest) its last four characters agree
    // with and compares the rest of it -- one round of pool loads for the whole wave; only a lane whose candidate fails
    // looks at its next one (two entries of a group share their last four characters only where the adapter repeats
    // itself).  (Round 2 compared entry by entry: up to four pool round trips per pass, one behind the other.)
    int found = 999;
    uint32_t tried = 0;                     // entries already looked at
    const uint32_t ek[8] = {ent[0].x, ent[0].z, ent[1].x, ent[1].z, ent[2].x, ent[2].z, ent[3].x, ent[3].z};
    const uint32_t em[8] = {ent[0].y, ent[0].w, ent[1].y, ent[1].w, ent[2].y, ent[2].w, ent[3].y, ent[3].w};
    for (bool more = true; more;) {
        uint32_t key = 0, meta = 0, sel = 8u;
#pragma unroll
        for (int u = 7; u >= 0; u--) {
            const uint32_t elen = em[u] & 0xFFu;
            const uint32_t kmask = elen >= 4u ? 0xFFFFFFFFu : elen == 0u ? 0u : 0xFFFFFFFFu << (8u * (4u - elen));
            const bool ok = !((tried >> u) & 1u) && elen <= len && elen != 0 && ((t4 ^ ek[u]) & kmask) == 0;
            if (ok) { key = ek[u]; meta = em[u]; sel = (uint32_t)u; }
        }
        (void)key;
        if (sel < 8u) {
            tried |= 1u << sel;
            const uint32_t elen = meta & 0xFFu;
            bool same = true;
            if (elen > 4u) {
                const uint32_t rest = elen - 4u;                          // characters e - elen .. e - 4 against the master's first `rest`
                const uint8_t *a = p.pool2 + (size_t)(meta >> 16) * 128u, *r = raw + e - elen;
                constexpr int NB = 10;                                    // (entries of up to 84 characters in one go; longer: the loop below)
                unsigned long long y[NB];
#pragma unroll
                for (int b = 0; b < NB; b++) { y[b] = 0; if (8u * b < rest) __builtin_memcpy(&y[b], a + 8 * b, 8); }
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (8u * b < rest) {
                        unsigned long long x;
                        __builtin_memcpy(&x, r + 8 * b, 8);
                        const uint32_t n = rest - 8u * b;
                        const unsigned long long m = n >= 8u ? ~0ull : (1ull << (8u * n)) - 1ull;
                        if (((x & 0xDFDFDFDFDFDFDFDFull) ^ y[b]) & m) same = false;
                    }
                }
                for (uint32_t i = 8u * NB; i < rest; i += 8) {
                    unsigned long long x, yy;
                    __builtin_memcpy(&x, r + i, 8);
                    __builtin_memcpy(&yy, a + i, 8);
                    const uint32_t n = rest - i;
                    const unsigned long long m = n >= 8u ? ~0ull : (1ull << (8u * n)) - 1ull;
                    if (((x & 0xDFDFDFDFDFDFDFDFull) ^ yy) & m) same = false;
                }
            }
            if (same) { found = (int)(int8_t)((meta >> 8) & 0xFFu); more = false; }
        } else {
            more = false;
        }
    }
    TD_LSTAMP(10);   // adapter search
    return make_int2((int)bar, found);
}

template <int CPT>
__global__ __launch_bounds__(FBLOCK, 4) void k_split2(const SplitParams p) {
    constexpr int TILE_CH = CPT * FBLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;
    constexpr uint32_t WCH = CPT * 64;
    constexpr uint32_t WBYTES = WCH * 16;
    constexpr uint32_t HALO = SPLIT2_HALO;
    static_assert(FBLOCK == 256 && (CPT % 2) == 0, "256 threads, an even number of chunks per thread");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *L_raw = lds;
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds + TILE + HALO + 64);      // (64 spare bytes: a piece read past the window's end)
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(lds + TILE + HALO + 64 + TILE_CH * 2u);   // 64 dwords
    uint8_t *L_bidx = reinterpret_cast<uint8_t *>(L_misc + 64);
    const unsigned long long *L_bval = reinterpret_cast<const unsigned long long *>(L_bidx);
    const uint32_t *L_bmeta = reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta);
    const uint16_t *L_bdir = reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir);
    // L_misc: [1] a byte >= 0x80 in tile or halo, [2] a wave has more terminators than its list holds, [4..7] wave totals
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += FBLOCK) reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];
    if (tid < 16) L_misc[tid] = 0;
    // the buffer's r-th sequence line has global index seq0 + 4 r
    const uint64_t seq0 = p.first_line + ((1 - (p.first_line & 3)) & 3);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t voff = (uint32_t)wave * WBYTES + (uint32_t)lane * 16u;
    __syncthreads();
#ifdef TD_PHASE_PROF
    unsigned long long prof_acc[PROF_PHASES] = {};
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif

    for (uint32_t t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const uint64_t tbase = (uint64_t)t * TILE;
        TD_STAMP(0);   // loop head
        const uint64_t P = t ? (p.prefix[t - 1] & ~FLAG_INC) : 0ull;            // terminators before this tile (used in C: its latency runs under A and B)
        // ---------------- A: this wave's quarter: raw bytes and terminator masks -> LDS (bytes past the buffer read as zero)
        const uint64_t rem = p.nbytes - tbase;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.buf + tbase), 0,
                                                                           (int)(rem < (uint64_t)(TILE + HALO) ? rem : (uint64_t)(TILE + HALO)), 0x00020000);
        uint4 v[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * 1024, 0);
            v[j] = make_uint4(q.x, q.y, q.z, q.w);
        }
        uint4 vh = make_uint4(0u, 0u, 0u, 0u);
        const bool has_halo = (uint32_t)tid < HALO / 16u;
        if (has_halo) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)tid * 16u, (int)TILE, 0);
            vh = make_uint4(q.x, q.y, q.z, q.w);
        }
        bool wave_crb = false;
        {
            uint32_t hiacc = vh.x | vh.y | vh.z | vh.w;
#pragma unroll
            for (int j = 0; j < CPT; j++) hiacc |= v[j].x | v[j].y | v[j].z | v[j].w;
#pragma unroll
            for (int j = 0; j < CPT; j++) *reinterpret_cast<uint4 *>(L_raw + voff + j * 1024) = v[j];
            if (has_halo) *reinterpret_cast<uint4 *>(L_raw + TILE + (size_t)tid * 16u) = vh;
            uint16_t *Lm = L_mask + wave * WCH + lane;
            const bool general = __any((hiacc & 0x80808080u) != 0);
            uint32_t crs = 0;
            if (__builtin_expect(!general, 1)) {
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t nl = eq_mask16_ascii(v[j], 0x0A0A0A0Au, 0x7F7F7F7Fu), cr = eq_mask16_ascii(v[j], 0x0D0D0D0Du, 0x7F7F7F7Fu);
                    Lm[j * 64] = (uint16_t)(nl | (cr & ~(nl >> 1)));      // (a '\r' in the chunk's last byte: settled in phase B)
                    crs |= cr;
                }
            } else {
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t nl = eq_mask16(v[j], 0x0A0A0A0Au), cr = eq_mask16(v[j], 0x0D0D0D0Du);
                    Lm[j * 64] = (uint16_t)(nl | (cr & ~(nl >> 1)));
                    crs |= cr;
                }
                if (hiacc & 0x80808080u) L_misc[1] = 1;
            }
            wave_crb = __any((crs & 0x8000u) != 0);
        }
        wave_lds_fence();
        TD_STAMP(1);   // A: loads issued, waited for, raw + masks -> LDS

        // ---------------- B: terminators of this thread's CPT consecutive chunks, wave scan, list of line starts
        uint32_t mm[CPT / 2];
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) mm[i] = reinterpret_cast<const uint32_t *>(L_mask)[tid * (CPT / 2) + i];
        const uint32_t span0 = tid * CPT * 16u;
        const uint32_t wend = ((uint32_t)wave + 1u) * WBYTES;
        if (__builtin_expect(wave_crb, 0)) {
            // a chunk whose last byte is '\r': one terminator with the '\n' that opens the next chunk, if there is one
#pragma unroll
            for (int i = 0; i < CPT / 2; i++) {
#pragma unroll
                for (int hbit = 15; hbit < 32; hbit += 16) {
                    if ((mm[i] >> hbit) & 1u) {
                        const uint32_t at = span0 + 32u * i + (uint32_t)hbit;
                        if (L_raw[at] == 0x0Du) {
                            // (the next byte may belong to another wave's quarter, or to the halo: not in LDS yet)
                            const uint64_t g = tbase + at + 1u;
                            const uint32_t nxb = at + 1u < wend ? (uint32_t)L_raw[at + 1u] : g < p.nbytes ? (uint32_t)p.buf[g] : 0u;
                            if (nxb == 0x0Au) mm[i] &= ~(1u << hbit);
                        }
                    }
                }
            }
            vm_settled();
        }
        // (bytes past the buffer's end were read as zeros: no terminators there)
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) cnt += __builtin_popcount(mm[i]);
        const uint32_t incl = wave_incl_scan(cnt, lane);
        {
            uint16_t *Ll = L_mask + wave * WCH;
            const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (__builtin_expect(wtot <= WCH, 1)) {
                uint32_t k = incl - cnt;
#pragma unroll
                for (int i = 0; i < CPT / 2; i++) {
                    uint32_t m = mm[i];
                    while (m) {
                        const uint32_t bit = __builtin_ctz(m);
                        m &= m - 1;
                        Ll[k] = (uint16_t)(span0 + 32u * i + bit + 1u);
                        k++;
                    }
                }
            }
            if (lane == 63) {
                L_misc[4 + wave] = incl;
                if (incl > WCH) L_misc[2] = 1;
            }
        }
        TD_STAMP(2);   // B
        lds_barrier();
        TD_STAMP(3);   // barrier 1

        // ---------------- C: running totals, the tile's first line number
        const uint4 tot4 = *reinterpret_cast<const uint4 *>(L_misc + 4);
        const uint4 flg4 = *reinterpret_cast<const uint4 *>(L_misc);
        const uint32_t wb1 = tot4.x, wb2 = wb1 + tot4.y, wb3 = wb2 + tot4.z, total = wb3 + tot4.w;
        const uint32_t r0 = (4u - (uint32_t)((p.first_line + P) & 3)) & 3u;      // ordinals == r0 (mod 4) precede sequence lines
        const uint32_t nwant = (total + 3u - r0) >> 2;
        const bool regular = t != 0 && (flg4.y | flg4.z) == 0 && tbase + TILE + HALO <= p.nbytes;

        if (__builtin_expect(regular, 1)) {
            // ---------------- D: sequence line j of the tile -> thread j
            const uint32_t j0 = ((uint32_t)tid + 64u * (t & 3u)) & (uint32_t)(FBLOCK - 1);
#pragma nounroll
            for (uint32_t j = j0; j < nwant; j += FBLOCK) {
                const uint32_t o = r0 + 4u * j;
                auto at = [&](uint32_t ord) -> uint32_t {
                    uint32_t sel = 0u;
                    sel = ord >= wb1 ? 1u * WCH - wb1 : sel;
                    sel = ord >= wb2 ? 2u * WCH - wb2 : sel;
                    sel = ord >= wb3 ? 3u * WCH - wb3 : sel;
                    return L_mask[ord + sel];
                };
                const uint32_t srel = at(o);
                uint32_t nx = 0;                                               // start of the next line (0: not inside the window)
                if (o + 1u < total) nx = at(o + 1u);
                else {
                    // the tile's last line: its terminator lies in the halo, if the line is not longer than that -- looked for
                    // sixteen bytes at a time (one lane of the tile walks here while its wave waits: byte by byte this was
                    // an eighth of a pass's time)
                    for (uint32_t q = srel > TILE ? srel : TILE; q < TILE + HALO && nx == 0; q += 16) {
                        const uint4 v = lds_read16(L_raw + q);
                        uint32_t m = eq_mask16(v, 0x0A0A0A0Au) | eq_mask16(v, 0x0D0D0D0Du);
                        if (q + 16u > TILE + HALO) m &= (1u << (TILE + HALO - q)) - 1u;
                        if (m) nx = q + (uint32_t)__builtin_ctz(m) + 1u;
                    }
                }
                const uint64_t line = p.first_line + P + o + 1u;
                int2 r;
#ifdef TD_PHASE_PROF
                if (nx != 0) r = split_line_lds(p, L_bval, L_bmeta, L_bdir, L_raw, srel, nx, tid == 0 ? prof_acc : nullptr, &prof_last);
#else
                if (nx != 0) r = split_line_lds(p, L_bval, L_bmeta, L_bdir, L_raw, srel, nx);
#endif
                else r = split_line(p, L_bval, L_bmeta, L_bdir, tbase + srel);
                p.out[(line - seq0) >> 2] = r;
            }
        } else {
            // ---------------- cold: every thread walks the sequence lines that start in its own span (global memory)
            uint32_t wbase = wave == 0 ? 0u : wave == 1 ? wb1 : wave == 2 ? wb2 : wb3;
            uint64_t ord = P + wbase + incl - cnt;
            if (t == 0 && tid == 0 && p.nbytes > 0 && (p.first_line & 3) == 1)
                p.out[0] = split_line(p, L_bval, L_bmeta, L_bdir, 0);
#pragma unroll
            for (int k = 0; k < CPT / 2; k++) {
                uint32_t m = mm[k];
                while (m) {
                    const uint32_t bit = __builtin_ctz(m);
                    m &= m - 1;
                    const uint64_t line = p.first_line + ord + 1;
                    const uint64_t gpos = tbase + span0 + 32u * k + bit + 1u;
                    ord++;
                    if ((line & 3) == 1 && gpos < p.nbytes)
                        p.out[(line - seq0) >> 2] = split_line(p, L_bval, L_bmeta, L_bdir, gpos);
                }
            }
        }
        TD_STAMP(4);   // C + D (thread 0's share)
        if (tid == 0) { L_misc[1] = 0; L_misc[2] = 0; }
        lds_barrier();                                    // LDS is reused by the next tile
        TD_STAMP(5);   // end barrier
    }
#ifdef TD_PHASE_PROF
    if (tid == 0)
        for (int i = 0; i < PROF_PHASES; i++) atomicAdd(p.stats + 8 + i, prof_acc[i]);
#endif
}

}  // namespace tdk
