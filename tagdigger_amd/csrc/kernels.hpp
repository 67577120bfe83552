// Device side of libtagdig: the fused FASTQ -> count-matrix kernel for gfx950.
//
// One launch makes ONE pass over the FASTQ bytes in HBM (the algorithmic
// bytes of the HBM roofline) and replaces the whole record loop of
// tagdigger_fun.find_tags_fastq (reference tagdigger_fun.py:249-274):
//
//   tile claim     a workgroup takes the next tile by ticket (atomic counter),
//                  so every predecessor tile is owned by a running workgroup
//   phase 1a       coalesced 16 B/lane loads -> LDS, and per 16-byte chunk a
//                  16-bit line-terminator mask (\n, \r\n, bare \r: Python's
//                  universal newlines, :241-250) by SWAR + v_dot4
//   phase 1b       each thread owns CPT consecutive chunks: popcount, block
//                  scan, then decoupled look-back over per-tile state words
//                  gives the global line index of every terminator, hence
//                  which lines are sequence lines (lineindex % 4 == 1, :254)
//   phase 2        one lane per sequence line: skip leading blanks (:256),
//                  2-bit pack + validate from LDS (v_perm/v_dot4), barcode
//                  prefix lookup in an LDS directory (:257), tag lookup in a
//                  hash table of packed tags in global memory / L2 (:260),
//                  one no-return atomic add into the count matrix (:267)
//
// The reference's pointer trie (:71-134) is re-laid flat: because the stored
// sequences are prefix-free after the build-time shadowing rules, "walk the
// trie" is equivalent to "find the one stored sequence that is a prefix of
// the read", which a directory / hash probe plus a masked compare answers
// without pointer chasing.  No MFMA: the path is byte/integer work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdk {

constexpr int BLOCK = 256;
constexpr int RLIST_CAP = 1024;            // sequence-line starts kept per tile; overflow is handled inline
constexpr uint64_t FLAG_AGG = 1ull << 62;  // tile state: own terminator count published
constexpr uint64_t FLAG_INC = 2ull << 62;  // tile state: inclusive prefix published
constexpr uint64_t VAL_MASK = (1ull << 62) - 1;
constexpr uint32_t BDIR_BASES = 5;         // barcode directory is keyed on the first 5 bases
constexpr uint32_t BDIR_SIZE = 1u << (2 * BDIR_BASES);
constexpr uint32_t SPIN_LIMIT = 1u << 22;

// device-side error bits (stats[ST_ERR])
constexpr unsigned long long ERR_NONASCII = 1, ERR_SPIN = 2, ERR_TASSEL = 4;
enum { ST_READS = 0, ST_BARCUT = 1, ST_TAG = 2, ST_LINES = 3, ST_ERR = 4 };

struct KParams {
    const uint8_t *buf;      // FASTQ bytes, 16-byte aligned
    uint64_t nbytes;
    uint64_t first_line;     // global index of the first line in buf
    uint64_t limit_line;     // last sequence-line index that may be counted (maxreads)
    uint64_t *state;         // [ntiles] look-back words
    uint32_t *ticket;        // tile dispenser
    uint32_t ntiles;
    uint32_t halo;           // bytes after the tile also staged in LDS (multiple of 16)
    // barcode index blob (copied to LDS): bval u64[nent] | bmeta u32[nent] | bdir u16[1024] | bcand u16[ncand]
    const uint32_t *bblob;
    uint32_t bblob_bytes, off_bmeta, off_bdir, off_bcand;
    // tag hash table
    const uint4 *slots;
    uint32_t slot_mask;
    uint32_t m_bases;        // tags are hashed on their first m bases (1..32)
    const uint4 *shorts;     // tags shorter than m: {u64 bases, u32 len, u32 col}
    uint32_t nshort;
    uint32_t *counts;        // [rows][ncols]
    unsigned long long *counts64; // tassel mode
    uint32_t ncols;
    unsigned long long *stats;
    uint32_t nch;            // 16-byte chunks converted per read (<= 2W+3)
    uint32_t maxwo;          // max (tag offset >> 4)
    uint32_t prefilled;      // state[] already holds inclusive prefixes (two-pass mode)
    // streamed pieces: lines consumed by earlier pieces (added to first_line) and where to
    // leave the running total for the next piece; both may be null
    const unsigned long long *cursor_in;
    unsigned long long *cursor_out;
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}
// 0x01 in every byte of x that equals the byte replicated in c4, else 0x00 (exact)
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t c4) {
    uint32_t y = x ^ c4;
    uint32_t t = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu;
    return (~t) >> 7;
}
// 16-bit mask, bit k = byte k of the chunk equals c
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t c4) {
    uint32_t lo = udot4(eq_bytes(v.x, c4), 0x08040201u, 0u);
    lo = udot4(eq_bytes(v.y, c4), 0x80402010u, lo);
    uint32_t hi = udot4(eq_bytes(v.z, c4), 0x08040201u, 0u);
    hi = udot4(eq_bytes(v.w, c4), 0x80402010u, hi);
    return lo | (hi << 8);
}

__device__ __forceinline__ uint64_t ld_state(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16 bytes at absolute offset g (multiple of 16); bytes at or past nbytes read as NUL
// (not a base, not blank, not a terminator)
__device__ __noinline__ uint4 load_chunk_tail(const uint8_t *buf, uint64_t nbytes, uint64_t g) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (g + k < nbytes) {
            uint32_t b = buf[g + k];
            w[k >> 2] |= b << (8 * (k & 3));
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 load_chunk(const KParams &p, uint64_t g) {
    if (g + 16 <= p.nbytes) return *reinterpret_cast<const uint4 *>(p.buf + g);
    return load_chunk_tail(p.buf, p.nbytes, g);
}

__device__ __forceinline__ bool is_blank(uint32_t b) {  // str.strip() set minus the line terminators
    return b == 0x20u || b == 0x09u || b == 0x0Bu || b == 0x0Cu || (b >= 0x1Cu && b <= 0x1Fu);
}

__device__ __forceinline__ uint32_t hash_key(uint64_t key) {  // must match host hash_key()
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = (lo * 0x9E3779B1u) ^ ((hi + 0x7F4A7C15u) * 0x85EBCA77u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
    return h;
}

// wave-level inclusive scan (64 lanes)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---------------------------------------------------------------- the kernel
// CPT: 16-byte chunks per thread per tile (tile = CPT*4 KiB); W: 64-bit words per packed tag
template <int CPT, int W, bool TASSEL>
__global__ __launch_bounds__(BLOCK) void k_count(const KParams p) {
    constexpr int TILE_CH = CPT * BLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;
    constexpr int NCHMAX = 2 * W + 3;
    constexpr int NS = 2 * W + 4;      // aligned 16-base words kept (zero padded)
    constexpr int SLOT_U4 = (W + 2) / 2;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t win = TILE + p.halo;
    uint8_t *L_data = lds;
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds + win);
    uint16_t *L_rlist = L_mask + TILE_CH;
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(L_rlist + RLIST_CAP);   // 64 dwords
    unsigned long long *L_misc64 = reinterpret_cast<unsigned long long *>(L_misc + 32);
    uint8_t *L_bidx = reinterpret_cast<uint8_t *>(L_misc + 64);
    const unsigned long long *L_bval = reinterpret_cast<const unsigned long long *>(L_bidx);
    const uint32_t *L_bmeta = reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta);
    const uint16_t *L_bdir = reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir);
    const uint16_t *L_bcand = reinterpret_cast<const uint16_t *>(L_bidx + p.off_bcand);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += BLOCK)
        reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];

    uint32_t st_reads = 0, st_bar = 0, st_tag = 0;
    unsigned long long st_lines = 0;
    const unsigned long long carried = p.cursor_in ? *p.cursor_in : 0ull;
    const uint64_t first_line = p.first_line + carried;

    // ------------------------------------------------------------ per-read matcher
    // srel: offset of the line's first byte relative to the tile start (may lie in the halo)
    auto match_read = [&](uint64_t tbase, uint32_t srel, unsigned long long weight) {
        st_reads++;
        auto peek = [&](uint32_t rel) -> uint32_t {
            if (rel < win) return L_data[rel];
            uint64_t g = tbase + rel;
            return g < p.nbytes ? p.buf[g] : 0u;
        };
        while (is_blank(peek(srel))) srel++;   // ends at the terminator at the latest
        const uint32_t a = srel & 15u, c0 = srel >> 4;
        uint32_t codes[NCHMAX + 1];
        uint32_t inv[(NCHMAX + 1) / 2];
#pragma unroll
        for (int i = 0; i < (NCHMAX + 1) / 2; i++) inv[i] = 0;
#pragma unroll
        for (int i = 0; i < NCHMAX; i++) {
            uint32_t cw = 0, iw = 0xFFFFu;
            if (i < (int)p.nch) {
                const uint32_t o = (c0 + i) * 16u;
                uint4 v = (o + 16u <= win) ? *reinterpret_cast<const uint4 *>(L_data + o)
                                           : load_chunk(p, tbase + o);
                const uint32_t x[4] = {v.x, v.y, v.z, v.w};
                iw = 0;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    uint32_t code = (x[d] >> 1) & 0x03030303u;           // A0 C1 T2 G3
                    uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, code);
                    uint32_t diff = (x[d] & 0xDFDFDFDFu) ^ expect;        // 0 where the byte is [ACGTacgt]
                    uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7 & 0x01010101u;
                    cw = (cw << 8) | udot4(code, 0x01041040u, 0u);        // first base in the top bits
                    iw = (iw << 4) | udot4(nz, 0x01020408u, 0u);
                }
            }
            codes[i] = cw;
            if (i & 1) inv[i >> 1] |= iw; else inv[i >> 1] |= iw << 16;
        }
        codes[NCHMAX] = 0;
        if ((NCHMAX & 1)) inv[NCHMAX >> 1] |= 0xFFFFu;   // padding half-word is invalid
        // number of leading valid bases of the read
        inv[0] &= 0xFFFFFFFFu >> a;
        uint32_t nvalid = 0;
        {
            bool found = false;
#pragma unroll
            for (int i = 0; i < (NCHMAX + 1) / 2; i++) {
                if (!found) {
                    if (inv[i]) { nvalid += __builtin_clz(inv[i]); found = true; }
                    else nvalid += 32;
                }
            }
            nvalid -= a;
        }
        // stream aligned to the read start: S[w] holds bases 16w..16w+15
        uint32_t S[NS];
#pragma unroll
        for (int w = 0; w < NS; w++) {
            if (w < NCHMAX) {
                uint64_t pr = ((uint64_t)codes[w] << 32) | codes[w + 1];
                S[w] = (uint32_t)(pr >> (32u - 2u * a));
            } else S[w] = 0;
        }
        // ---- barcode + cut site (reference :257)
        const uint64_t K = ((uint64_t)S[0] << 32) | S[1];
        uint32_t ci = L_bdir[S[0] >> (32 - 2 * BDIR_BASES)];
        uint32_t meta = 0;
        bool bhit = false;
        if (ci != 0xFFFFu) {
            for (;;) {
                uint32_t e = L_bcand[ci];
                uint32_t m = L_bmeta[e & 0x7FFFu];
                uint32_t len = m & 63u;
                if (len <= nvalid && ((K ^ L_bval[e & 0x7FFFu]) >> (64u - 2u * len)) == 0) { meta = m; bhit = true; break; }
                if (e & 0x8000u) break;
                ci++;
            }
        }
        if (!bhit) return;
        st_bar++;
        const uint32_t off = (meta >> 6) & 1023u, row = meta >> 16;
        if (nvalid <= off) return;
        const uint32_t nrem = nvalid - off;
        // ---- tag (reference :260): bases off.. of the read, as 64-bit words
        const uint32_t wo = off >> 4, sh = 2u * (off & 15u);
        for (uint32_t t = 0; t < p.maxwo; t++) {
            if (wo > t) {
#pragma unroll
                for (int w = 0; w < NS - 1; w++) S[w] = S[w + 1];
                S[NS - 1] = 0;
            }
        }
        uint64_t R[W];
#pragma unroll
        for (int w = 0; w < W; w++) {
            uint64_t hi = ((uint64_t)S[2 * w] << 32) | S[2 * w + 1];
            uint64_t lo = ((uint64_t)S[2 * w + 1] << 32) | S[2 * w + 2];
            uint32_t top = (uint32_t)(((hi << sh) >> 32));
            uint32_t bot = (uint32_t)(((lo << sh) >> 32));
            R[w] = ((uint64_t)top << 32) | bot;
        }
        auto prefix_eq = [&](const uint64_t *T, uint32_t len) -> bool {
            bool ok = true;
#pragma unroll
            for (int w = 0; w < W; w++) {
                int nb = (int)len - 32 * w;
                uint64_t mask = nb <= 0 ? 0ull : nb >= 32 ? ~0ull : (~0ull << (64 - 2 * nb));
                ok = ok && (((R[w] ^ T[w]) & mask) == 0);
            }
            return ok;
        };
        bool thit = false;
        uint32_t col = 0;
        if (nrem >= p.m_bases) {
            uint32_t slot = hash_key(R[0] >> (64u - 2u * p.m_bases)) & p.slot_mask;
            for (uint32_t probes = 0; probes <= p.slot_mask; probes++) {
                const uint4 *sp = p.slots + (size_t)slot * SLOT_U4;
                uint32_t raw[SLOT_U4 * 4];
#pragma unroll
                for (int q = 0; q < SLOT_U4; q++) {
                    uint4 v = sp[q];
                    raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
                }
                const uint32_t len = raw[2 * W];
                if (len == 0) break;                       // empty slot: not in the table
                uint64_t T[W];
#pragma unroll
                for (int w = 0; w < W; w++) T[w] = ((uint64_t)raw[2 * w + 1] << 32) | raw[2 * w];
                if (len <= nrem && prefix_eq(T, len)) { thit = true; col = raw[2 * W + 1]; break; }
                slot = (slot + 1) & p.slot_mask;
            }
        }
        if (!thit) {
            for (uint32_t e = 0; e < p.nshort; e++) {
                uint4 v = p.shorts[e];
                uint64_t tv = ((uint64_t)v.y << 32) | v.x;
                uint32_t len = v.z;
                if (len <= nrem && ((R[0] ^ tv) >> (64u - 2u * len)) == 0) { thit = true; col = v.w; break; }
            }
        }
        if (!thit) return;
        st_tag++;
        const size_t cell = (size_t)row * p.ncols + col;
        if (TASSEL) __hip_atomic_fetch_add(p.counts64 + cell, weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p.counts + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // tassel_tagcount (reference :251-253): hrel = start of a header line.  Parses
    // int(line[line.find("count=")+6:].strip()), then counts the following line.
    auto header_then_read = [&](uint64_t tbase, uint32_t hrel) {
        auto gb = [&](uint64_t g) -> uint32_t { return g < p.nbytes ? p.buf[g] : 0x0Au; };
        const uint64_t hs = tbase + hrel;
        uint64_t he = hs;
        while (he < p.nbytes && gb(he) != 0x0Au && gb(he) != 0x0Du) he++;
        uint64_t at = hs + 5;                       // find()==-1 -> slice [5:]
        for (uint64_t q = hs; q + 6 <= he; q++) {
            if (gb(q) == 'c' && gb(q + 1) == 'o' && gb(q + 2) == 'u' && gb(q + 3) == 'n' && gb(q + 4) == 't' && gb(q + 5) == '=') { at = q + 6; break; }
        }
        uint64_t a0 = at, a1 = he;
        auto sp = [&](uint32_t b) { return is_blank(b) || b == 0x0Au || b == 0x0Du; };
        while (a0 < a1 && sp(gb(a0))) a0++;
        while (a1 > a0 && sp(gb(a1 - 1))) a1--;
        bool ok = a0 < a1, neg = false;
        if (ok && (gb(a0) == '+' || gb(a0) == '-')) { neg = gb(a0) == '-'; a0++; ok = a0 < a1; }
        unsigned long long v = 0;
        for (uint64_t q = a0; ok && q < a1; q++) {
            uint32_t b = gb(q);
            if (b < '0' || b > '9') ok = false; else v = v * 10ull + (b - '0');
        }
        if (!ok) { atomicOr(p.stats + ST_ERR, ERR_TASSEL); return; }
        if (neg) v = 0ull - v;
        // the sequence line follows the header's terminator, if the file goes on
        uint64_t ss = he;
        if (ss < p.nbytes) { if (gb(ss) == 0x0Du && gb(ss + 1) == 0x0Au) ss += 2; else ss += 1; }
        if (ss >= p.nbytes) return;
        match_read(tbase, (uint32_t)(ss - tbase), v);
    };

    for (;;) {
        if (tid == 0) { L_misc[0] = atomicAdd(p.ticket, 1u); L_misc[1] = 0; }
        __syncthreads();
        const uint32_t t = L_misc[0];
        if (t >= p.ntiles) break;
        const uint64_t tbase = (uint64_t)t * TILE;

        // ---------------- phase 1a: stream the tile into LDS, terminator masks per chunk
        {
            uint4 v[CPT];
#pragma unroll
            for (int j = 0; j < CPT; j++) v[j] = load_chunk(p, tbase + (uint64_t)(j * BLOCK + tid) * 16u);
            uint32_t hiacc = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const uint32_t c = j * BLOCK + tid;
                const uint64_t g = tbase + (uint64_t)c * 16u;
                uint32_t nl = eq_mask16(v[j], 0x0A0A0A0Au);
                uint32_t cr = eq_mask16(v[j], 0x0D0D0D0Du);
                uint32_t term = nl | (cr & ~(nl >> 1));
                if (cr & 0x8000u) {                    // \r in the chunk's last byte: \r\n across chunks?
                    uint64_t nx = g + 16;
                    if (nx < p.nbytes && p.buf[nx] == 0x0A) term &= 0x7FFFu;
                }
                if (g + 16 > p.nbytes) term &= g < p.nbytes ? ((1u << (uint32_t)(p.nbytes - g)) - 1u) : 0u;
                hiacc |= v[j].x | v[j].y | v[j].z | v[j].w;
                *reinterpret_cast<uint4 *>(L_data + c * 16u) = v[j];
                L_mask[c] = (uint16_t)term;
            }
            for (uint32_t c = TILE_CH + tid; c * 16u < win; c += BLOCK)
                *reinterpret_cast<uint4 *>(L_data + c * 16u) = load_chunk(p, tbase + (uint64_t)c * 16u);
            if (hiacc & 0x80808080u) L_misc[1] = 1;
        }
        __syncthreads();

        const bool tile_has_hi = L_misc[1] != 0;
        // ---------------- phase 1b: terminators of this thread's CPT consecutive chunks
        uint32_t mm[CPT / 2];
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) mm[i] = reinterpret_cast<const uint32_t *>(L_mask)[tid * (CPT / 2) + i];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) cnt += __builtin_popcount(mm[i]);
        uint32_t incl = wave_incl_scan(cnt, lane);
        if (lane == 63) L_misc[4 + wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { uint32_t x = L_misc[4 + w]; if (w < wave) wbase += x; total += x; }
        const uint32_t excl = wbase + incl - cnt;

        // ---------------- decoupled look-back: terminators before this tile
        if (tid == 0 && !p.prefilled) st_state(p.state + t, FLAG_AGG | total);
        uint64_t P = 0;
        {
            int64_t base = t;
            for (;;) {
                const int64_t idx = base - 1 - tid;
                uint64_t s = FLAG_INC;
                if (idx >= 0) {
                    s = ld_state(p.state + idx);
                    uint32_t spins = 0;
                    while ((s >> 62) == 0) {
                        __builtin_amdgcn_s_sleep(2);
                        s = ld_state(p.state + idx);
                        if (++spins > SPIN_LIMIT) { atomicOr(p.stats + ST_ERR, ERR_SPIN); s = FLAG_INC; break; }
                    }
                }
                const bool inc = (s >> 62) == 2;
                const uint64_t b = __ballot(inc);
                const int f = b ? __builtin_ctzll(b) : 64;
                const uint64_t ws = wave_sum64(lane <= f ? (s & VAL_MASK) : 0ull);
                if (lane == 0) { L_misc64[wave] = ws; L_misc[8 + wave] = b ? 1u : 0u; }
                __syncthreads();
                bool done = false;
#pragma unroll
                for (int w = 0; w < BLOCK / 64; w++) {
                    if (!done) { P += L_misc64[w]; done = L_misc[8 + w] != 0; }
                }
                __syncthreads();
                if (done) break;
                base -= BLOCK;
            }
        }
        if (tid == 0 && !p.prefilled) st_state(p.state + t, FLAG_INC | (P + total));
        if (tid == 0) {
            st_lines += total;
            if (p.cursor_out && t == p.ntiles - 1) *p.cursor_out = carried + P + total;
        }

        const uint64_t Lb = first_line + P + excl;   // index of the line this thread's span starts in
        const uint32_t span0 = tid * CPT * 16u;
        const uint32_t want = TASSEL ? 0u : 1u;         // line phase that starts a unit of work
        const uint64_t lim = TASSEL ? p.limit_line - 1 : p.limit_line;

        // ---------------- rare: bytes >= 0x80 in the tile -- are any inside a counted sequence line?
        if (tile_has_hi) {
            uint32_t seen = 0;
#pragma nounroll
            for (uint32_t q = 0; q < CPT * 16u; q++) {
                const uint32_t rel = span0 + q;
                if (L_data[rel] >= 0x80u && tbase + rel < p.nbytes) {
                    const uint64_t line = Lb + seen;
                    if ((line & 3) == 1 && line <= p.limit_line) atomicOr(p.stats + ST_ERR, ERR_NONASCII);
                }
                seen += (L_mask[rel >> 4] >> (rel & 15u)) & 1u;
            }
        }

        // ---------------- lines to process.  In-tile terminator ordinal i (0-based) is followed by
        // line first_line+P+i+1, so the wanted lines follow the ordinals i == r0 (mod 4) and the
        // j-th of them gets list slot j: no scan, no atomics.  Slot 0 of tile 0 is the buffer's
        // first line.  The list is processed in rounds of RLIST_CAP (one round for sane input).
        const uint32_t r0 = (want + 3u - (uint32_t)((first_line + P) & 3)) & 3u;
        const uint32_t extra = t == 0 ? 1u : 0u;
        const uint32_t nslots = extra + (total > r0 ? (total - r0 + 3u) / 4u : 0u);
        for (uint32_t rbase = 0; rbase < nslots; rbase += RLIST_CAP) {
            if (t == 0 && tid == 0 && rbase == 0)
                L_rlist[0] = (p.nbytes > 0 && (first_line & 3) == want && first_line <= lim) ? (uint16_t)0 : (uint16_t)0xFFFF;
            uint32_t i = excl;
#pragma unroll
            for (int k = 0; k < CPT / 2; k++) {
                uint32_t m = mm[k];
                while (m) {
                    const uint32_t bit = __builtin_ctz(m);
                    m &= m - 1;
                    if ((i & 3u) == r0) {
                        const uint32_t slot = extra + ((i - r0) >> 2);
                        if (slot >= rbase && slot < rbase + RLIST_CAP) {
                            const uint64_t line = first_line + P + i + 1;
                            const uint32_t srel = span0 + 32u * k + bit + 1u;
                            const bool ok = line <= lim && tbase + srel < p.nbytes;
                            L_rlist[slot - rbase] = ok ? (uint16_t)srel : (uint16_t)0xFFFF;
                        }
                    }
                    i++;
                }
            }
            __syncthreads();
            // ------------ phase 2: one lane per line
            const uint32_t n = min(nslots - rbase, (uint32_t)RLIST_CAP);
            for (uint32_t e = tid; e < n; e += BLOCK) {
                const uint32_t srel = L_rlist[e];
                if (srel != 0xFFFFu) {
                    if (TASSEL) header_then_read(tbase, srel);
                    else match_read(tbase, srel, 1ull);
                }
            }
            __syncthreads();
        }
    }

    // ---------------- statistics: one atomic per wave
    unsigned long long r = wave_sum64(st_reads), b = wave_sum64(st_bar), g = wave_sum64(st_tag), l = wave_sum64(st_lines);
    if (lane == 0) {
        if (r) atomicAdd(p.stats + ST_READS, r);
        if (b) atomicAdd(p.stats + ST_BARCUT, b);
        if (g) atomicAdd(p.stats + ST_TAG, g);
        if (l) atomicAdd(p.stats + ST_LINES, l);
    }
}

// ---------------------------------------------------------------- terminator count only
// per-tile terminator counts (same masks as k_count phase 1a); tile = CPT*4 KiB
template <int CPT>
__global__ __launch_bounds__(BLOCK) void k_count_lines(const uint8_t *buf, uint64_t nbytes, uint32_t ntiles,
                                                       uint64_t *tile_counts) {
    __shared__ uint32_t part[BLOCK / 64];
    KParams p{};
    p.buf = buf; p.nbytes = nbytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t tbase = (uint64_t)t * (CPT * BLOCK * 16u);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint64_t g = tbase + (uint64_t)(j * BLOCK + tid) * 16u;
            uint4 v = load_chunk(p, g);
            uint32_t nl = eq_mask16(v, 0x0A0A0A0Au), cr = eq_mask16(v, 0x0D0D0D0Du);
            uint32_t term = nl | (cr & ~(nl >> 1));
            if (cr & 0x8000u) { uint64_t nx = g + 16; if (nx < nbytes && buf[nx] == 0x0A) term &= 0x7FFFu; }
            if (g + 16 > nbytes) term &= g < nbytes ? ((1u << (uint32_t)(nbytes - g)) - 1u) : 0u;
            cnt += __builtin_popcount(term);
        }
        uint32_t s = wave_incl_scan(cnt, lane);
        if (lane == 63) part[wave] = s;
        __syncthreads();
        if (tid == 0) { uint32_t tot = 0; for (int w = 0; w < BLOCK / 64; w++) tot += part[w]; tile_counts[t] = tot; }
        __syncthreads();
    }
}

// single block: state[i] = FLAG_INC | inclusive prefix of tile_counts; total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint64_t *tile_counts, uint32_t ntiles, uint64_t *state,
                                                     unsigned long long *total_out) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t i = base + tid;
        unsigned long long v = i < ntiles ? tile_counts[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long off = carry;
        for (int w = 0; w < wave; w++) off += wsum[w];
        if (i < ntiles) state[i] = FLAG_INC | (off + inc);
        __syncthreads();
        if (tid == 1023) carry = off + inc;
        __syncthreads();
    }
    if (tid == 0 && total_out) *total_out = carry;
}

}  // namespace tdk
