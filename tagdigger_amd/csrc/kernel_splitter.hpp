// The barcode splitter's per-read branch (reference tagdigger_fun.py:1251-1283 and the
// loop at :1328-1363): for every sequence line of a FASTQ buffer -- a line whose global index is
// 1 mod 4 -- which barcode+cut site opens it, and where the read must be clipped on its 3' end:
// after the first full restriction site behind the cut site, or where an adapter runs off the end
// of the read.  The kernel only decides; the host writes the clipped records (tagdig.hip).
//
// Line numbers are exact: the caller has run k_count_lines + k_scan_tiles, so `prefix[t]` is the
// number of line terminators before the end of tile t.  One workgroup per tile; each thread owns a
// span of consecutive chunks and handles the sequence lines that START in it, walking their bytes
// in global memory (the tile has just been streamed through L2).  Result r of the buffer belongs to
// its r-th sequence line: out[r] = {barcode index or -1, the reference's slice2 (999 = no clip)}.
#pragma once
#include "kernels.hpp"

namespace tdk {

struct SplitEntry {            // one adapter prefix to look for at the end of a read (build_adapter_tree :1208-1249)
    uint32_t off;              // its characters: pool[off .. off + len)
    uint32_t len;
    int32_t slice;             // the (negative) index the reference slices with when it is found
    uint32_t key;              // its last four characters (fewer: the top bytes), the last one in the top byte
};

struct SplitParams {
    const uint8_t *buf;
    uint64_t nbytes;
    uint64_t first_line;       // global index of the buffer's first line
    const uint64_t *prefix;    // [ntiles] FLAG_INC | terminators up to the end of tile t
    uint32_t ntiles;
    const uint32_t *bblob;     // barcode + cut site index (same layout as the counting path's)
    uint32_t bblob_bytes, off_bmeta, off_bdir;
    uint32_t cutlen;
    unsigned long long site0, site1;   // the two full restriction sites, last character in the low byte
    uint32_t site0_len, site1_len;
    const uint32_t *ent_begin; // [barnum + 1] entries of barcode b: entries[ent_begin[b] .. ent_begin[b + 1])
    const uint32_t *ent_group; // [barnum][4] within those, where the entries whose LAST base has code c begin (sorted by it)
    const SplitEntry *entries;
    const uint8_t *pool;
    // k_split2: group (16 bar + 4 code(last) + code(second last)) of gcap entries each (unused ones: len 0)
    const SplitEntry *entries16;
    uint32_t gcap;
    // k_split2 (round 3): group (64 bar + 16 code(last) + 4 code(second last) + code(third last)) of EIGHT compact entries
    // {key, len | (slice & 0xFF) << 8 | master << 16} in 64 bytes (unused ones: len 0; a short entry sits in every group
    // of its characters); an entry's characters are the first `len` of its master string, pool2[128 master ..]
    const uint2 *entries8;
    const uint8_t *pool2;
    int2 *out;
    unsigned long long *stats; // ST_ERR
    uint32_t dbg;              // timing-only ablations (results wrong when nonzero): 64 no site search, 128 no adapter search
};

__device__ __forceinline__ uint32_t upper_ascii(uint32_t c) { return (c >= 0x61u && c <= 0x7Au) ? c - 0x20u : c; }

// 16 bytes of the buffer at any alignment (zeros past its end)
__device__ __forceinline__ uint4 load16_any(const SplitParams &p, uint64_t g) {
    if (g + 16 <= p.nbytes) {
        const uint8_t *q = p.buf + g;
        if (((uintptr_t)q & 3u) == 0) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(q);
            return make_uint4(w[0], w[1], w[2], w[3]);
        }
        // unaligned: two aligned 16-byte windows would do, but byte assembly keeps it simple and L2-hot
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = q[4 * k] | (q[4 * k + 1] << 8) | (q[4 * k + 2] << 16) | ((uint32_t)q[4 * k + 3] << 24);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16 && g + k < p.nbytes; k++) w[k >> 2] |= (uint32_t)p.buf[g + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One sequence line starting at gpos.
__device__ __forceinline__ int2 split_line(const SplitParams &p, const unsigned long long *L_bval, const uint32_t *L_bmeta,
                                           const uint16_t *L_bdir, uint64_t gpos) {
    const uint8_t *b = p.buf;
    // line.strip(): blanks off both ends; the line ends at the first terminator (or the buffer's end).
    // The scan for the terminator goes 16 bytes at a time.
    uint64_t s = gpos;
    while (s < p.nbytes && is_blank(b[s])) s++;
    uint64_t e = s;
    uint32_t hiacc = 0;
    for (;;) {
        if (e >= p.nbytes) { e = p.nbytes; break; }
        const uint4 v = load16_any(p, e);
        const uint32_t term = eq_mask16(v, 0x0A0A0A0Au) | eq_mask16(v, 0x0D0D0D0Du);
        const uint32_t left = (uint32_t)std::min<uint64_t>(16, p.nbytes - e);
        const uint32_t t2 = term & ((1u << left) - 1u);
        if (t2) {
            const uint32_t k = __builtin_ctz(t2);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};                // bytes >= 0x80 before the terminator only
            for (uint32_t q = 0; q < k; q++) hiacc |= (w[q >> 2] >> (8 * (q & 3))) & 0x80u;
            e += k;
            break;
        }
        hiacc |= (v.x | v.y | v.z | v.w) & 0x80808080u;
        if (left < 16) { e = p.nbytes; break; }
        e += 16;
    }
    while (e > s && is_blank(b[e - 1])) e--;
    if (hiacc) atomicOr(p.stats + ST_ERR, ERR_NONASCII);
    const uint64_t len = e - s;

    // ---- barcode + cut site: the first (up to 32) valid bases, packed like the index
    const uint2 c0 = convert_chunk(load16_any(p, s)), c1 = convert_chunk(load16_any(p, s + 16));
    const unsigned long long Kfull = ((unsigned long long)c0.x << 32) | c1.x;
    const uint32_t invalid = (c0.y & 0xFFFFu) | (c1.y << 16);
    uint32_t nvalid = invalid ? (uint32_t)__builtin_ctz(invalid) : 32u;
    if (nvalid > len) nvalid = (uint32_t)len;
    // (bases past the valid ones must not take part: the index compares whole prefixes)
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
    if (!hit) return make_int2(-1, 999);
    const uint32_t bar = meta >> 16;
    const uint64_t start = ((meta >> 6) & 63u) + p.cutlen;           // searchstart = len(barcode) + len(cutsite)

    // ---- first full restriction site at or after `start` (str.find): a rolling window of the
    // last eight characters against both sites, the characters taken from 16-byte loads
    long long rs0 = -1, rs1 = -1;
    {
        const unsigned long long m0 = p.site0_len >= 8 ? ~0ull : ((1ull << (8 * p.site0_len)) - 1ull);
        const unsigned long long m1 = p.site1_len >= 8 ? ~0ull : ((1ull << (8 * p.site1_len)) - 1ull);
        unsigned long long win = 0;
        for (uint64_t i0 = start; i0 < len && (rs0 < 0 || rs1 < 0); i0 += 16) {
            const uint4 v = load16_any(p, s + i0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const uint32_t n = (uint32_t)std::min<uint64_t>(16, len - i0);
            for (uint32_t q = 0; q < n; q++) {
                const uint64_t i = i0 + q;
                win = (win << 8) | upper_ascii((w[q >> 2] >> (8 * (q & 3))) & 0xFFu);
                const uint64_t have = i - start + 1;
                if (rs0 < 0 && have >= p.site0_len && (win & m0) == p.site0) rs0 = (long long)(i + 1 - p.site0_len);
                if (rs1 < 0 && have >= p.site1_len && (win & m1) == p.site1) rs1 = (long long)(i + 1 - p.site1_len);
            }
        }
        // (an empty site is found at `start` itself whenever start <= len, as str.find does)
        if (p.site0_len == 0 && start <= len) rs0 = (long long)start;
        if (p.site1_len == 0 && start <= len) rs1 = (long long)start;
    }
    if (rs0 >= 0 || rs1 >= 0) {
        long long cut;
        if (rs1 < 0) cut = rs0 + p.site0_len;
        else if (rs0 < 0) cut = rs1 + p.site1_len;
        else if (rs0 < rs1) cut = rs0 + p.site0_len;
        else cut = rs1 + p.site1_len;
        return make_int2((int)bar, (int)cut);
    }
    // ---- no full site: does the read END with the start of an adapter?  (the reference walks a
    // trie over the reversed read; its stored set is prefix-free, so at most one entry matches.)
    // Only the entries whose last base is the read's last base can match: they are stored together.
    if (len == 0) return make_int2((int)bar, 999);
    const uint32_t lastc = upper_ascii(b[e - 1]);
    const uint32_t lcode = (lastc >> 1) & 3u;
    if (lastc != ((0x47544341u >> (8 * lcode)) & 0xFFu)) return make_int2((int)bar, 999);     // not a base: no entry ends with it
    const uint32_t e0 = p.ent_group[4 * bar + lcode];
    const uint32_t e1 = lcode == 3 ? p.ent_begin[bar + 1] : p.ent_group[4 * bar + lcode + 1];
    // (the result is carried out of the loop in `found`: with a `return` from inside these nested,
    // divergent loops hipcc 7.2 -O3 returned the slice of the wrong entry)
    int found = 999;
    for (uint32_t k = e0; k < e1 && found == 999; k++) {
        const uint4 raw = reinterpret_cast<const uint4 *>(p.entries)[k];     // {off, len, slice, -}
        const uint32_t elen = raw.y;
        if (elen > len || elen == 0) continue;
        const uint8_t *a = p.pool + raw.x;
        bool same = true;
        for (uint32_t q = 1; q < elen && same; q++)                   // from the read's second-last character backwards
            same = upper_ascii(b[e - 1 - q]) == a[elen - 1 - q];
        if (same) found = (int)raw.z;
    }
    return make_int2((int)bar, found);
}

template <int CPT>
__global__ __launch_bounds__(BLOCK) void k_split(const SplitParams p) {
    constexpr int TILE_CH = CPT * BLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds);
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(lds + TILE_CH * 2);          // 16 dwords
    uint8_t *L_bidx = lds + TILE_CH * 2 + 64;
    const unsigned long long *L_bval = reinterpret_cast<const unsigned long long *>(L_bidx);
    const uint32_t *L_bmeta = reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta);
    const uint16_t *L_bdir = reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += BLOCK) reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];
    KParams kp{};
    kp.buf = p.buf; kp.nbytes = p.nbytes;
    // the buffer's r-th sequence line has global index seq0 + 4 r
    const uint64_t seq0 = p.first_line + ((1 - (p.first_line & 3)) & 3);

    for (uint32_t t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const uint64_t tbase = (uint64_t)t * TILE;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = j * BLOCK + tid;
            const uint64_t g = tbase + (uint64_t)c * 16u;
            const uint4 v = load_chunk(kp, g);
            uint32_t nl = eq_mask16(v, 0x0A0A0A0Au), cr = eq_mask16(v, 0x0D0D0D0Du);
            uint32_t term = nl | (cr & ~(nl >> 1));
            if (cr & 0x8000u) { const uint64_t nx = g + 16; if (nx < p.nbytes && p.buf[nx] == 0x0A) term &= 0x7FFFu; }
            if (g + 16 > p.nbytes) term &= g < p.nbytes ? ((1u << (uint32_t)(p.nbytes - g)) - 1u) : 0u;
            L_mask[c] = (uint16_t)term;
        }
        __syncthreads();
        uint32_t mm[CPT / 2];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) {
            mm[i] = reinterpret_cast<const uint32_t *>(L_mask)[tid * (CPT / 2) + i];
            cnt += __builtin_popcount(mm[i]);
        }
        const uint32_t incl = wave_incl_scan(cnt, lane);
        if (lane == 63) L_misc[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wave; w++) wbase += L_misc[w];
        // terminators before this tile: the previous tile's inclusive prefix
        const uint64_t P = t ? (p.prefix[t - 1] & ~FLAG_INC) : 0ull;
        uint64_t ord = P + wbase + incl - cnt;          // terminators before this thread's span
        const uint32_t span0 = tid * CPT * 16u;

        // the buffer's own first line belongs to tile 0, thread 0
        if (t == 0 && tid == 0 && p.nbytes > 0 && (p.first_line & 3) == 1)
            p.out[0] = split_line(p, L_bval, L_bmeta, L_bdir, 0);
#pragma unroll
        for (int k = 0; k < CPT / 2; k++) {
            uint32_t m = mm[k];
            while (m) {
                const uint32_t bit = __builtin_ctz(m);
                m &= m - 1;
                const uint64_t line = p.first_line + ord + 1;            // index of the line after this terminator
                const uint64_t gpos = tbase + span0 + 32u * k + bit + 1u;
                ord++;
                if ((line & 3) == 1 && gpos < p.nbytes)
                    p.out[(line - seq0) >> 2] = split_line(p, L_bval, L_bmeta, L_bdir, gpos);
            }
        }
    }
}

}  // namespace tdk
