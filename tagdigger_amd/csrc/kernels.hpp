// Device side of libtagdig: the fused FASTQ -> count-matrix kernel for gfx950.
//
// One launch makes ONE pass over the FASTQ bytes in HBM (the algorithmic
// bytes of the HBM roofline) and replaces the whole record loop of
// tagdigger_fun.find_tags_fastq (reference tagdigger_fun.py:249-274):
//
//   tile claim     a workgroup takes the next tile by ticket (atomic counter),
//                  so every predecessor tile is owned by a running workgroup
//   phase 1a       coalesced 16 B/lane loads into registers; per 16-byte chunk a
//                  16-bit line-terminator mask (\n, \r\n, bare \r: Python's
//                  universal newlines, :241-250) by SWAR + v_dot4 -> LDS
//   phase 1b       each thread owns CPT consecutive chunks: popcount, block
//                  scan, publish the tile's terminator count, and issue the
//                  loads of the decoupled look-back over per-tile state words
//   phase 1c       while those loads fly, EVERY chunk is upper-cased, validated
//                  and 2-bit packed by all lanes (v_perm/v_dot4) and only the
//                  packed form (8 B per 16 bytes) is staged in LDS
//   look-back      gives the global line index of every terminator, hence
//                  which lines are sequence lines (lineindex % 4 == 1, :254)
//   phase 2        one lane per sequence line: packed bases from LDS, barcode
//                  prefix lookup in an LDS directory (:257), tag lookup in a
//                  bucketed hash table of packed tags in global memory / L2
//                  (:260), one no-return atomic add into the count matrix
//                  (:267).  Lines that start with a non-base byte (blanks to
//                  skip, :256) or run past the staged window take a slow path
//                  that re-reads raw bytes from global memory.
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
constexpr int TLIST_CAP = 2048;            // line starts (one per terminator) listed per round; more -> extra rounds
constexpr int LBQ = 4;                     // look-back: tile states examined per thread per round (window 1024)
constexpr uint64_t FLAG_AGG = 1ull << 62;  // tile state: own terminator count published
constexpr uint64_t FLAG_INC = 2ull << 62;  // tile state: inclusive prefix published
constexpr uint64_t VAL_MASK = (1ull << 62) - 1;
constexpr uint32_t BDIR_BASES = 5;         // barcode directory is keyed on the first 5 bases
constexpr uint32_t BDIR_SIZE = 1u << (2 * BDIR_BASES);
#ifndef TD_BU4
#define TD_BU4 4                    // 16-byte quarters of a tag bucket for tags of up to 96 bases (4: 64-byte buckets; 2: 32-byte ones)
#endif
constexpr uint32_t BMETA_LAST = 1u << 12;  // bmeta: len (6 bits) | tag offset (6 bits) << 6 | last-of-bucket | row << 16
constexpr uint32_t SPIN_LIMIT = 1u << 22;
// per-line result of the matcher (top two bits) | count-matrix cell
constexpr uint64_t R_NONE = 0, R_BAR = 1ull << 62, R_TAG = 2ull << 62, R_DEFER = 3ull << 62, R_CELL = (1ull << 62) - 1;
constexpr int IPL = 4;                     // line starts examined per lane per batch (the select chains below assume 4)

// timing-only ablation switches (td_set_option "debug_ablate")
constexpr uint32_t DBG_NO_ATOMIC = 1, DBG_NO_PROBE = 2, DBG_NO_PHASE2 = 4, DBG_NO_LOOKBACK = 8,
                   DBG_NO_PACK = 16, DBG_STATIC_TILES = 32;

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
    // barcode index blob (copied to LDS): bval u64[nent] | bmeta u32[nent] | bdir u16[1024]; entries are stored
    // bucket by bucket (an entry shorter than the directory key is repeated in every bucket it covers)
    const uint32_t *bblob;
    uint32_t bblob_bytes, off_bmeta, off_bdir, off_bcand;
    // tag hash table: buckets of 64 B (W<=3) or 128 B; dword 0 = which keys went on to the next bucket because this
    // one was full (a 32-bit Bloom filter indexed by the top five bits of the key's hash), then slots of
    // {W x u64 packed bases, u32 meta = col<<10 | len}; empty slot: meta 0
    const uint4 *buckets;
    uint32_t bucket_mask;
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
    uint32_t nt_loads;       // stream the FASTQ with non-temporal loads (keeps the tag table in L2)
    // fast path: tiles >= tail_tile (the last one or two, whose window would cross the buffer's end)
    // are loaded from a zero-padded copy, so that every tile is fetched with plain unconditional loads
    const uint8_t *tail_buf;
    uint32_t tail_tile;
    uint32_t prio;           // fast path: wave priority per phase, see set_prio
    uint32_t stagger, stagger_div;   // start-up stagger of co-resident workgroups (units of 4096 cycles; 0 = off)
    uint32_t dbg;            // timing-only ablations (results wrong when nonzero); see DBG_*
    uint32_t hot_cache;      // k_fast2: count through the per-wave hot-cell cache in LDS
    uint32_t run;            // k_fast2: consecutive tiles per workgroup turn (the line phase is carried inside a run)
    // progress windows (reference :268-271 prints its three counters every 50 000 reads): win[w] holds, for the
    // reads with 0-based ordinal in [w, w + 1) * 50 000, how many had a barcode (low word) and a tag (high word);
    // null = not wanted
    unsigned long long *win;
    uint32_t win_cap;
    uint32_t f4_nprod;       // k_fast4: producer waves of the workgroup's sixteen (the others match); 0: *f4_nprod_dev
    const uint32_t *f4_nprod_dev;   // ... as k_f4_estimate left it in device memory
};
constexpr uint32_t PROG_WINDOW = 50000;

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
    // 24 x 24-bit multiplies only (v_mul_u32_u24 issues at the full rate, v_mul_lo_u32 at a quarter of it);
    // the key's 64 bits enter as 22 + 21 + 21 bit fields
    const uint32_t a = (uint32_t)key & 0x3FFFFFu, b = (uint32_t)(key >> 22) & 0x1FFFFFu, c = (uint32_t)(key >> 43);
    uint32_t h = __umul24(a, 0x9E3779u) ^ (__umul24(b, 0x85EBCBu) + 0x7F4A7C15u);
    h ^= __umul24(c, 0xC2B2AFu) << 3;
    h ^= h >> 15;
    h = __umul24(h & 0xFFFFFFu, 0x2C1B3Du) ^ (h >> 9);
    h ^= h >> 13;
    return h;
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, NOT for its
// global loads (__syncthreads() adds s_waitcnt vmcnt(0), which would stall on the next tile's
// bytes that are deliberately left in flight across the barrier).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Between LDS stores by some lanes of a wave and LDS loads of the same locations by other lanes of
// the SAME wave: the hardware runs one wave's LDS operations in order; this keeps the compiler from
// reordering them (the accesses use different element types) and drains the stores.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// wave-level inclusive scan (64 lanes) in six DPP additions: a scan inside each row of 16 lanes
// (row_shr 1, 2, 4, 8, lanes without a source add 0), then lane 15 of rows 0 and 2 added to rows
// 1 and 3 (row_bcast:15), then lane 31 to rows 2 and 3 (row_bcast:31).  No LDS traffic, no index
// arithmetic (__shfl_up goes through ds_bpermute with four address instructions per step).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
// One read (0-based ordinal ridx; kind 0 nothing, 1 barcode, 2 barcode and tag) into its progress window.
__device__ __forceinline__ void win_add(const KParams &p, uint64_t ridx, uint32_t kind) {
    const uint64_t w = ridx / PROG_WINDOW;
    if (kind && w < p.win_cap) atomicAdd(p.win + w, 1ull | ((unsigned long long)(kind == 2) << 32));
}
// The same for the reads the lanes of a wave hold (every lane of the wave calls; `have` = this lane holds one): the
// lanes' reads are neighbours in the file, so one or two windows take them all -- one atomic each.
__device__ __forceinline__ void win_add_wave(const KParams &p, bool have, uint64_t ridx, uint32_t kind) {
    uint64_t w = have && kind ? ridx / PROG_WINDOW : ~0ull;
    for (int round = 0; round < 2; round++) {
        const uint64_t w0 = wave_min64(w);
        if (w0 == ~0ull) return;
        const uint64_t sum = wave_sum64(w == w0 ? (1ull | ((unsigned long long)(kind == 2) << 32)) : 0ull);
        if ((threadIdx.x & 63) == 0 && w0 < p.win_cap) atomicAdd(p.win + w0, (unsigned long long)sum);
        if (w == w0) w = ~0ull;
    }
    if (w != ~0ull && w < p.win_cap) atomicAdd(p.win + w, 1ull | ((unsigned long long)(kind == 2) << 32));   // (lines of a few bytes)
}

// In-kernel phase stamps: a separate diagnostic build only (-DTD_PHASE_PROF, libtagdig_prof.so).
// Thread 0 of each workgroup adds the shader-clock cycles spent between consecutive stamps to
// stats[8 + phase]; shares are read, the build's run time is never quoted.
#ifdef TD_PHASE_PROF
#define TD_STAMP(i)                                                         \
    do {                                                                    \
        if (tid == 0) {                                                     \
            unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
            prof_acc[i] += now_ - prof_last;                                \
            prof_last = now_;                                               \
        }                                                                   \
    } while (0)
// inside the matcher (wave 0, whatever lanes are active): drains LDS first so that its latency
// lands in the phase that waits for it
#define TD_MSTAMP(cx, i, drain)                                              \
    do {                                                                    \
        if ((cx).pacc && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) { \
            if (drain) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
            unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
            (cx).pacc[i] += now_ - *(cx).plast;                             \
            *(cx).plast = now_;                                             \
        }                                                                   \
    } while (0)
#else
#define TD_STAMP(i) do {} while (0)
#define TD_MSTAMP(cx, i, drain) do {} while (0)
#endif
constexpr int PROF_PHASES = 20;

// ---------------------------------------------------------------- the kernel
// 2-bit codes (first base in the top bits) and per-base invalid flags (bit k = byte k of the chunk)
// of one 16-byte chunk.  A 0, C 1, T 2, G 3 = (byte >> 1) & 3; [ACGTacgt] are the valid bytes.
__device__ __forceinline__ uint2 convert_chunk(const uint4 &v) {
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t cw = 0, iw = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t code = (x[d] >> 1) & 0x03030303u;
        uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, code);
        uint32_t diff = (x[d] & 0xDFDFDFDFu) ^ expect;            // 0 where the byte is a base
        uint32_t nz = ((((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7) & 0x01010101u;
        cw = (cw << 8) | udot4(code, 0x01041040u, 0u);
        iw |= udot4(nz, 0x08040201u, 0u) << (4 * d);
    }
    return make_uint2(cw, iw);
}

// The same two conversions for chunks known to hold only bytes < 0x80 (every FASTQ in practice):
// "byte != 0" is then simply bit 7 of byte + 0x7F, with no carry between bytes.
__device__ __forceinline__ uint2 convert_chunk_ascii(const uint4 &v) {
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t cw = 0, ihi = 0, ilo = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t code = (x[d] >> 1) & 0x03030303u;
        uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, code);
        uint32_t diff = (x[d] & 0xDFDFDFDFu) ^ expect;            // 0 where the byte is a base, < 0x80 otherwise
        uint32_t nz = (diff + 0x7F7F7F7Fu) & 0x80808080u;
        cw = udot4(code, 0x01041040u, cw << 8);                   // (the dot's addend carries the words before)
        // 0x80 * (first byte -> weight 1 ... ), two dwords per accumulator
        if (d == 0) ilo = udot4(nz, 0x08040201u, 0u);
        else if (d == 1) ilo = udot4(nz, 0x80402010u, ilo);
        else if (d == 2) ihi = udot4(nz, 0x08040201u, 0u);
        else ihi = udot4(nz, 0x80402010u, ihi);
    }
    return make_uint2(cw, (ihi << 1) | (ilo >> 7));             // each accumulator is 128 * (8 flag bits)
}
// (a ^ b) + c in one instruction.  v_xad_u32 takes no literal operands, and at most one of its
// sources may be an SGPR: b is kept in an SGPR and c in a VGPR across the loop (written out because
// the compiler, given two literals, emits v_xor + v_add instead).
__device__ __forceinline__ uint32_t xor_add(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// AND-accumulates "no '\r' here" over an all-ASCII chunk: bit 7 of a byte of acc is cleared iff some chunk had '\r' there
__device__ __forceinline__ void cr_absent_ascii(const uint4 &v, uint32_t &acc) {
    acc &= xor_add(v.x, 0x0D0D0D0Du, 0x7F7F7F7Fu) & xor_add(v.y, 0x0D0D0D0Du, 0x7F7F7F7Fu);
    acc &= xor_add(v.z, 0x0D0D0D0Du, 0x7F7F7F7Fu) & xor_add(v.w, 0x0D0D0D0Du, 0x7F7F7F7Fu);
}
// 16-bit mask of bytes equal to '\n' in an all-ASCII chunk
__device__ __forceinline__ uint32_t nl_mask16_ascii(const uint4 &v) {
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        // y = byte ^ c is 0..0x7F; (y ^ 0x7F) + 1 = 0x80 - y has bit 7 set iff y == 0 (no carries between bytes):
        // the mask comes out in the positive sense, with no complement at the end
        const uint32_t u = xor_add(x[d], 0x0A0A0A0Au ^ 0x7F7F7F7Fu, 0x01010101u) & 0x80808080u;       // 0x80 where the byte IS '\n'
        if (d == 0) lo = udot4(u, 0x08040201u, 0u);
        else if (d == 1) lo = udot4(u, 0x80402010u, lo);
        else if (d == 2) hi = udot4(u, 0x08040201u, 0u);
        else hi = udot4(u, 0x80402010u, hi);
    }
    return (lo >> 7) | (hi << 1);             // each dot is 128 * (eight mask bits)
}

// CPT: 16-byte chunks per thread per tile (tile = CPT*4 KiB); W: 64-bit words per packed tag
// s_waitcnt vmcnt(0) the compiler's own wait-count bookkeeping can see.  Placed at the end of rare
// blocks that load from memory (overflow buckets, the short-tag list, the cold matcher): without
// it the compiler assumes those loads may still be pending where the rare block rejoins the hot
// path, and guards later register writes there with waits that would stall on the prefetch.
__device__ __forceinline__ void vm_settled() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// ---------------------------------------------------------------- per-line matcher (shared by both kernels)
// What a workgroup has staged in LDS for its current tile, plus the index.
struct TileCtx {
    const uint2 *L_conv;                 // packed chunks of the tile + halo
    uint32_t win_ch;                     // how many of them
    const unsigned long long *L_bval;    // barcode index (LDS copy)
    const uint32_t *L_bmeta;
    const uint16_t *L_bdir;
#ifdef TD_PHASE_PROF
    unsigned long long *pacc = nullptr, *plast = nullptr;
#endif
};

// gpos: absolute position of the line's first byte; srel: the same relative to the tile (fast mode).
// Fast mode reads packed chunks from LDS; slow mode re-reads raw bytes from global memory (leading
// blanks to strip, :256, or a window beyond what is staged).  Returns R_NONE (no barcode+site),
// R_BAR (barcode+site only), R_TAG | cell, or -- fast mode with DEFER only -- R_DEFER when the raw
// bytes are needed.  With DEFER false the fast mode falls through to the slow mode by itself.
// No side effects: the caller commits.
// MODE: ML_FAST  packed chunks only, returns R_DEFER when the raw bytes are needed (no global-memory code at all);
//       ML_SLOW  raw bytes only;   ML_BOTH  chosen by `slow`, R_DEFER from the fast branch (exact kernel).
enum { ML_FAST = 0, ML_SLOW = 1, ML_BOTH = 2 };
// The matcher comes in two halves so that a caller can put other work between them:
//   match_prepare  everything up to the tag hash and the loads of the tag bucket (left in flight);
//                  returns R_NONE / R_BAR / R_DEFER, or R_PEND with `pd` filled
//   match_finish   the masked compares, overflow buckets, the short list -> R_BAR or R_TAG | cell
template <int W> struct Pending {
    uint64_t R[W];          // the read from the tag offset on, packed
    uint32_t nr;            // valid bases there (bits 0..14), PD_PROBE: a bucket was fetched, row << 16 (count-matrix row)
    uint32_t boff;          // byte offset of that bucket in the table.  Kept (rather than recomputed) on purpose:
                            // the loads take it as their only address VGPR, and a VGPR that stays live is not
                            // overwritten while the loads still wait to read it (which would stall the wave)
    uint4 b[W <= 3 ? TD_BU4 : 8];   // that bucket (loads left in flight by match_prepare)
};
constexpr uint32_t PD_PROBE = 1u << 15;
constexpr uint64_t R_PEND = 1ull << 61;   // (kind bits 0) match_prepare: finish with match_finish

// Stage 1 of the matcher: the read as a stream of 16-base words aligned to its first base
// (S[w] = bases 16w..16w+15, first base in the top bits; zero padded) and the number of leading
// valid bases.  Returns R_DEFER (fast mode: raw bytes needed) or 0.
template <int W, int MODE>
__device__ __forceinline__ uint64_t fetch_stream(const KParams &p, const TileCtx &cx, uint64_t gpos, uint32_t srel, bool slow,
                                                 uint32_t (&S)[2 * W + 4], uint32_t &nvalid) {
    if (MODE == ML_FAST) slow = false;
    if (MODE == ML_SLOW) slow = true;
    constexpr int NCHMAX = 2 * W + 3;
    constexpr int NS = 2 * W + 4;      // aligned 16-base words kept (zero padded)
    const uint2 *L_conv = cx.L_conv;
    const uint32_t win_ch = cx.win_ch;
    if (MODE != ML_SLOW && !slow) {
        const uint32_t c0f = srel >> 4;
        if (c0f + p.nch > win_ch) return R_DEFER;
        // first byte is not a base: a blank to strip (slow path), or simply no match
        if ((L_conv[c0f].y >> (srel & 15u)) & 1u) return R_DEFER;
    }
    if (MODE != ML_FAST && slow) {
        while (gpos < p.nbytes && is_blank(p.buf[gpos])) gpos++;   // ends at the terminator at the latest
    }
    const uint32_t a = (uint32_t)(gpos & 15u);
    const uint64_t g0 = gpos & ~15ull;             // slow path: first chunk (tile bases are 16-aligned)
    const uint32_t c0 = srel >> 4;                 // fast path
    uint32_t codes[NCHMAX + 1];
    uint32_t inv[(NCHMAX + 1) / 2];
#pragma unroll
    for (int i = 0; i < (NCHMAX + 1) / 2; i++) inv[i] = 0;
#pragma unroll
    for (int i = 0; i < NCHMAX; i++) {
        uint2 e = make_uint2(0u, 0xFFFFu);
        if (i < (int)p.nch) {
            if (MODE == ML_FAST || (MODE == ML_BOTH && !slow)) e = L_conv[c0 + i];
            else e = convert_chunk(load_chunk(p, g0 + 16ull * i));
        }
        codes[i] = e.x;
        if (i & 1) inv[i >> 1] |= e.y << 16; else inv[i >> 1] |= e.y;       // 32 bases per word, first base in bit 0
    }
    codes[NCHMAX] = 0;
    if ((NCHMAX & 1)) inv[NCHMAX >> 1] |= 0xFFFF0000u;   // padding half-word is invalid
    // number of leading valid bases of the read
    inv[0] &= 0xFFFFFFFFu << a;
    nvalid = 0;
    {
        bool found = false;
#pragma unroll
        for (int i = 0; i < (NCHMAX + 1) / 2; i++) {
            if (!found) {
                if (inv[i]) { nvalid += __builtin_ctz(inv[i]); found = true; }
                else nvalid += 32;
            }
        }
        nvalid -= a;
    }
    // stream aligned to the read start: S[w] holds bases 16w..16w+15
#pragma unroll
    for (int w = 0; w < NS; w++) {
        if (w < NCHMAX) {
            uint64_t pr = ((uint64_t)codes[w] << 32) | codes[w + 1];
            S[w] = (uint32_t)(pr >> (32u - 2u * a));
        } else S[w] = 0;
    }
    return 0;
}

// Stage 2: barcode+site lookup (reference :257), the read from the tag offset on (:260), tag hash,
// first 16 bytes of the tag bucket put in flight.  Returns R_NONE / R_BAR, or R_PEND with `pd` filled.
template <int W>
__device__ __forceinline__ uint64_t match_stream(const KParams &p, const TileCtx &cx, uint32_t (&S)[2 * W + 4], uint32_t nvalid,
                                                 Pending<W> &pd) {
    constexpr int NS = 2 * W + 4;
    const unsigned long long *L_bval = cx.L_bval;
    const uint32_t *L_bmeta = cx.L_bmeta;
    const uint16_t *L_bdir = cx.L_bdir;
    // ---- barcode + cut site (reference :257)
    const uint64_t K = ((uint64_t)S[0] << 32) | S[1];
    uint32_t ci = L_bdir[S[0] >> (32 - 2 * BDIR_BASES)];     // entries of a bucket are contiguous
    uint32_t meta = 0;
    bool bhit = false;
    if (ci != 0xFFFFu) {
        for (;;) {
            const uint32_t m = L_bmeta[ci];
            const uint32_t len = m & 63u;
            if (len <= nvalid && ((K ^ L_bval[ci]) >> (64u - 2u * len)) == 0) { meta = m; bhit = true; break; }
            if (m & BMETA_LAST) break;
            ci++;
        }
    }
    TD_MSTAMP(cx, 9, 1);    // barcode directory walk
    if (!bhit) return R_NONE;
    const uint32_t off = (meta >> 6) & 63u, row = meta >> 16;
    if (nvalid <= off) return R_BAR;
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
    constexpr int BUCKET_U4_ = W <= 3 ? TD_BU4 : 8;
    TD_MSTAMP(cx, 10, 0);   // tag words
#pragma unroll
    for (int w = 0; w < W; w++) pd.R[w] = R[w];
    pd.nr = min(nrem, 0x7FFFu) | (row << 16);      // (tags are at most 32 W <= 320 bases: the cap loses nothing)
    if (nrem >= p.m_bases && !(p.dbg & DBG_NO_PROBE)) {
        pd.nr |= PD_PROBE;
        const uint32_t hk = hash_key(R[0] >> (64u - 2u * p.m_bases));
        const uint32_t bk = hk & p.bucket_mask;
        pd.boff = bk * (uint32_t)(BUCKET_U4_ * 16);         // (the table is far below 4 GiB: td_set_index checks)
        const uint4 *bp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(p.buckets) + pd.boff);
        pd.boff |= hk >> 27;                                // (low bits are free: the key's bit in the buckets' overflow filters)
#pragma unroll
        for (int q = 0; q < BUCKET_U4_; q++) pd.b[q] = bp[q];   // in flight: first used by match_finish
    } else if (p.nshort == 0) {
        return R_BAR;
    }
    TD_MSTAMP(cx, 14, 0);   // hash + bucket loads issued
    return R_PEND;
}

template <int W, int MODE>
__device__ __forceinline__ uint64_t match_prepare(const KParams &p, const TileCtx &cx, uint64_t gpos, uint32_t srel, bool slow,
                                                  Pending<W> &pd) {
    uint32_t S[2 * W + 4];
    uint32_t nvalid;
    const uint64_t r = fetch_stream<W, MODE>(p, cx, gpos, srel, slow, S, nvalid);
    if (r) return r;
    TD_MSTAMP(cx, 8, 1);    // packed chunks from LDS + alignment
    return match_stream<W>(p, cx, S, nvalid, pd);
}

template <int W>
__device__ __forceinline__ uint64_t match_finish(const KParams &p, const Pending<W> &pd) {
    constexpr int BUCKET_U4 = W <= 3 ? TD_BU4 : 8;          // 64- or 128-byte buckets
    constexpr int SLOT_DW = 2 * W + 1;
    constexpr int SPB = (BUCKET_U4 * 4 - 1) / SLOT_DW;  // slots per bucket
    const uint64_t *R = pd.R;
    const uint32_t nrem = pd.nr & 0x7FFFu;
    // do the first `len` (<= 32 W) bases of the read equal the stored tag's?  Counted rather than masked:
    // the bases before the first differing bit, word by word (no per-slot 64-bit masks to build)
    auto prefix_eq = [&](const uint64_t *T, uint32_t len) -> bool {
        uint32_t same = 0;                      // equal leading bases
        bool open = true;                       // no difference seen yet
#pragma unroll
        for (int w = 0; w < W; w++) {
            const uint64_t d = R[w] ^ T[w];
            const uint32_t z = d ? (uint32_t)__builtin_clzll(d) : 64u;
            if (open) same += z >> 1;
            open = open && z == 64u;
        }
        return len <= same;
    };
    bool thit = false;
    uint32_t col = 0;
    if (pd.nr & PD_PROBE) {
        uint32_t bk = pd.boff / (uint32_t)(BUCKET_U4 * 16);
        uint32_t raw[BUCKET_U4 * 4];
#pragma unroll
        for (int q = 0; q < BUCKET_U4; q++) {
            raw[4 * q] = pd.b[q].x; raw[4 * q + 1] = pd.b[q].y; raw[4 * q + 2] = pd.b[q].z; raw[4 * q + 3] = pd.b[q].w;
        }
        for (uint32_t probes = 0;;) {
#pragma unroll
            for (int sl = 0; sl < SPB; sl++) {
                const uint32_t meta2 = raw[1 + sl * SLOT_DW + 2 * W];
                const uint32_t len = meta2 & 1023u;
                uint64_t T[W];
#pragma unroll
                for (int w = 0; w < W; w++)
                    T[w] = ((uint64_t)raw[1 + sl * SLOT_DW + 2 * w + 1] << 32) | raw[1 + sl * SLOT_DW + 2 * w];
                if (len != 0 && len <= nrem && prefix_eq(T, len)) { thit = true; col = meta2 >> 10; }
            }
            // found, or no key with this one's filter bit ever went on from this bucket
            if (__builtin_expect(thit || !((raw[0] >> (pd.boff & 31u)) & 1u) || ++probes > p.bucket_mask, 1)) break;
            bk = (bk + 1) & p.bucket_mask;
            const uint4 *bp = p.buckets + (size_t)bk * BUCKET_U4;
#pragma unroll
            for (int q = 0; q < BUCKET_U4; q++) {
                const uint4 v = bp[q];
                raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
            }
            vm_settled();
        }
    }
    if (!thit) {
        for (uint32_t e = 0; e < p.nshort; e++) {
            uint4 v = p.shorts[e];
            vm_settled();
            uint64_t tv = ((uint64_t)v.y << 32) | v.x;
            uint32_t len = v.z;
            if (len <= nrem && ((R[0] ^ tv) >> (64u - 2u * len)) == 0) { thit = true; col = v.w; break; }
        }
    }
    if (!thit) return R_BAR;
    return R_TAG | ((uint64_t)(pd.nr >> 16) * p.ncols + col);
}

// both halves back to back
template <int W, int MODE>
__device__ __forceinline__ uint64_t match_line(const KParams &p, const TileCtx &cx, uint64_t gpos, uint32_t srel, bool slow) {
    Pending<W> pd;
    const uint64_t r = match_prepare<W, MODE>(p, cx, gpos, srel, slow, pd);
    if (r != R_PEND) return r;
    const uint64_t r2 = match_finish<W>(p, pd);
    TD_MSTAMP(cx, 5, 1);    // rest of the bucket + compares
    return r2;
}

#ifndef TD_WAVES_PER_SIMD
#define TD_WAVES_PER_SIMD 4   // register budget: 4 workgroups of 256 threads per CU
#endif
template <int CPT, int W, bool TASSEL>
__global__ __launch_bounds__(BLOCK, TD_WAVES_PER_SIMD) void k_count(const KParams p) {
    constexpr int TILE_CH = CPT * BLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;
    // LDS: packed chunks of the tile + halo | terminator masks (later: line-start list) | misc | barcode index
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t halo_ch = p.halo / 16u;
    const uint32_t win_ch = TILE_CH + halo_ch;           // chunks staged in LDS
    uint2 *L_conv = reinterpret_cast<uint2 *>(lds);
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds + (size_t)win_ch * 8u);
    uint16_t *L_tlist = L_mask;                          // aliases the masks once they are in registers
    constexpr uint32_t MASK_BYTES = (TILE_CH * 2 > TLIST_CAP * 2) ? TILE_CH * 2 : TLIST_CAP * 2;
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(lds + (size_t)win_ch * 8u + MASK_BYTES);   // 64 dwords
    unsigned long long *L_misc64 = reinterpret_cast<unsigned long long *>(L_misc + 32);
    uint8_t *L_bidx = reinterpret_cast<uint8_t *>(L_misc + 64);
    const unsigned long long *L_bval = reinterpret_cast<const unsigned long long *>(L_bidx);
    const uint32_t *L_bmeta = reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta);
    const uint16_t *L_bdir = reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += BLOCK)
        reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];

    uint32_t st_reads = 0, st_bar = 0, st_tag = 0;
    unsigned long long st_lines = 0;
#ifdef TD_PHASE_PROF
    unsigned long long prof_acc[PROF_PHASES] = {};
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif
    const unsigned long long carried = p.cursor_in ? *p.cursor_in : 0ull;
    const uint64_t first_line = p.first_line + carried;

    // ------------------------------------------------------------ per-read matcher (match_line above)
    const TileCtx cx{L_conv, win_ch, L_bval, L_bmeta, L_bdir};
    auto match_read = [&](uint64_t gpos, uint32_t srel, bool slow) -> uint64_t {
        return match_line<W, ML_BOTH>(p, cx, gpos, srel, slow);
    };

    // tassel_tagcount (reference :251-253): hrel = start of a header line.  Parses
    // int(line[line.find("count=")+6:].strip()) and returns the position of the following
    // line (or ~0 when there is none / the header is malformed).
    auto parse_header = [&](uint64_t hs, unsigned long long &weight) -> uint64_t {
        auto gb = [&](uint64_t g) -> uint32_t { return g < p.nbytes ? p.buf[g] : 0x0Au; };
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
        if (!ok) { atomicOr(p.stats + ST_ERR, ERR_TASSEL); return ~0ull; }
        weight = neg ? 0ull - v : v;
        uint64_t ss = he;                           // the sequence line follows the header's terminator
        if (ss < p.nbytes) { if (gb(ss) == 0x0Du && gb(ss + 1) == 0x0Au) ss += 2; else ss += 1; }
        return ss < p.nbytes ? ss : ~0ull;
    };

    uint32_t dbg_iter = 0;
    for (;;) {
        TD_STAMP(7);   // (tail of the previous tile)
        if (tid == 0) {
            L_misc[0] = (p.dbg & DBG_STATIC_TILES) ? blockIdx.x + (dbg_iter++) * gridDim.x : atomicAdd(p.ticket, 1u);
            L_misc[1] = 0;
        }
        __syncthreads();
        const uint32_t t = L_misc[0];
        if (t >= p.ntiles) break;
        TD_STAMP(0);   // ticket
        const uint64_t tbase = (uint64_t)t * TILE;

        // ---------------- phase 1a: tile -> registers, terminator mask per chunk -> LDS
        uint4 v[CPT];
        uint4 vh = make_uint4(0u, 0u, 0u, 0u);         // this thread's chunk of the halo (halo <= 4 KiB)
        const bool has_halo = (uint32_t)tid < halo_ch;
#pragma unroll
        for (int j = 0; j < CPT; j++) v[j] = load_chunk(p, tbase + (uint64_t)(j * BLOCK + tid) * 16u);
        if (has_halo) vh = load_chunk(p, tbase + (uint64_t)(TILE_CH + tid) * 16u);
        {
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
                L_mask[c] = (uint16_t)term;
            }
            if (hiacc & 0x80808080u) L_misc[1] = 1;
        }
        TD_STAMP(1);   // tile load + terminator masks
        __syncthreads();
        TD_STAMP(2);   // barrier (other waves' loads)

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
        __syncthreads();                               // (also: every thread now holds its masks; L_tlist may overwrite them)
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { uint32_t x = L_misc[4 + w]; if (w < wave) wbase += x; total += x; }
        const uint32_t excl = wbase + incl - cnt;
        TD_STAMP(3);   // block scan

        // ---------------- decoupled look-back, part 1: publish, then put the state loads in flight
        const bool no_lb = (p.dbg & DBG_NO_LOOKBACK) != 0;
        if (tid == 0 && !p.prefilled && !no_lb) st_state(p.state + t, FLAG_AGG | total);
        uint64_t s[LBQ];
#pragma unroll
        for (int q = 0; q < LBQ; q++) {
            const int64_t idx = (int64_t)t - 1 - (q * BLOCK + tid);
            s[q] = (idx >= 0 && !no_lb) ? ld_state(p.state + idx) : FLAG_INC;
        }

        // ---------------- phase 1c (under the look-back latency): pack every chunk, list the line starts
        if (!(p.dbg & DBG_NO_PACK))
#pragma unroll
        for (int j = 0; j < CPT; j++) L_conv[j * BLOCK + tid] = convert_chunk(v[j]);
        if (has_halo) L_conv[TILE_CH + tid] = convert_chunk(vh);
        auto list_terminators = [&](uint32_t rbase) {
            uint32_t i = excl;
#pragma unroll
            for (int k = 0; k < CPT / 2; k++) {
                uint32_t m = mm[k];
                while (m) {
                    const uint32_t bit = __builtin_ctz(m);
                    m &= m - 1;
                    if (i >= rbase && i < rbase + TLIST_CAP) L_tlist[i - rbase] = (uint16_t)(tid * CPT * 16u + 32u * k + bit + 1u);
                    i++;
                }
            }
        };
        list_terminators(0);
        TD_STAMP(5);   // pack + list

        __syncthreads();                               // packed chunks and the line-start list are visible
        TD_STAMP(2);

        // ---------------- phase 2, speculative: every listed line start is matched NOW, whatever its
        // line phase (headers, '+' and quality lines fail at their first byte), while the look-back
        // loads are in flight and the predecessors publish.  Then the look-back resolves the phase and
        // only the wanted lines are committed.  Lines whose raw bytes are needed (R_DEFER: first byte
        // not a base, or window beyond the staged chunks) are resolved after the look-back, as is
        // everything in tassel mode and in batches beyond the first (pathological line density).
        const uint32_t want = TASSEL ? 0u : 1u;         // line phase that starts a unit of work
        const uint64_t lim = TASSEL ? p.limit_line - 1 : p.limit_line;
        // tile 0 owns the buffer's first line: a virtual terminator with in-tile ordinal -1 (item 0)
        const uint32_t extra = (t == 0 && p.nbytes > 0) ? 1u : 0u;
        const uint32_t nitems = total + extra;
        uint64_t P = 0;
        uint32_t r0 = 0;
        uint32_t rbase = 0;                             // first ordinal held in L_tlist
        for (uint32_t b0 = 0; b0 == 0 || b0 < nitems; b0 += IPL * BLOCK) {
            if (b0) {                                   // next stretch of line starts
                rbase = b0 - extra;
                __syncthreads();
                list_terminators(rbase);
                __syncthreads();
            }
            uint64_t res[IPL];
            uint32_t sr[IPL];
            unsigned long long wgt[IPL];
#pragma unroll
            for (int k = 0; k < IPL; k++) { res[k] = R_DEFER; sr[k] = 0xFFFFFFFFu; wgt[k] = 1ull; }
            uint32_t tried = 0;                         // bit k: item k went through the speculative pass
#pragma nounroll
            for (int pass = (b0 == 0 && !TASSEL) ? 0 : 1; pass < 2; pass++) {
                if (pass == 1 && b0 == 0) {
                    // ------------ look-back, part 2: terminators before this tile
                    int64_t base = t;
                    for (;;) {
#pragma unroll
                        for (int q = 0; q < LBQ; q++) {
                            const int64_t idx = base - 1 - (q * BLOCK + tid);
                            uint32_t spins = 0;
                            while ((s[q] >> 62) == 0) {
                                __builtin_amdgcn_s_sleep(1);
                                s[q] = ld_state(p.state + idx);
                                if (++spins > SPIN_LIMIT) { atomicOr(p.stats + ST_ERR, ERR_SPIN); s[q] = FLAG_INC; break; }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < LBQ; q++) {
                            const uint64_t b = __ballot((s[q] >> 62) == 2);
                            const int f = b ? __builtin_ctzll(b) : 64;
                            const uint64_t ws = wave_sum64(lane <= f ? (s[q] & VAL_MASK) : 0ull);
                            if (lane == 0) { L_misc64[q * 4 + wave] = ws; L_misc[8 + q * 4 + wave] = b ? 1u : 0u; }
                        }
                        __syncthreads();
                        bool done = false;
#pragma unroll
                        for (int w = 0; w < LBQ * (BLOCK / 64); w++) {
                            if (!done) { P += L_misc64[w]; done = L_misc[8 + w] != 0; }
                        }
                        __syncthreads();
                        if (done) break;
                        base -= LBQ * BLOCK;
#pragma unroll
                        for (int q = 0; q < LBQ; q++) {
                            const int64_t idx = base - 1 - (q * BLOCK + tid);
                            s[q] = idx >= 0 ? ld_state(p.state + idx) : FLAG_INC;
                        }
                    }
                    if (tid == 0 && !p.prefilled && !no_lb) st_state(p.state + t, FLAG_INC | (P + total));
                    TD_STAMP(4);   // look-back (what phase 2 did not hide)
                    if (tid == 0) {
                        st_lines += total;
                        if (p.cursor_out && t == p.ntiles - 1) *p.cursor_out = carried + P + total;
                    }
                    // in-tile terminator ordinal i is followed by line first_line+P+i+1: the wanted
                    // lines follow the ordinals i == r0 (mod 4)
                    r0 = (want + 3u - (uint32_t)((first_line + P) & 3)) & 3u;
                }
#pragma nounroll
                for (int k = 0; k < IPL; k++) {
                    // lanes take 4 consecutive items each, so one k = one line phase across the wave:
                    // three of the four rounds end at the first byte for (almost) every lane
                    const uint32_t e = b0 + IPL * tid + k;
                    if (e >= nitems || (p.dbg & DBG_NO_PHASE2)) continue;
                    const bool first_item = e < extra;
                    const uint32_t i = e - extra;                    // in-tile ordinal of the terminator before the line (-1: none)
                    uint32_t srk = k == 0 ? sr[0] : k == 1 ? sr[1] : k == 2 ? sr[2] : sr[3];
                    uint64_t rk = k == 0 ? res[0] : k == 1 ? res[1] : k == 2 ? res[2] : res[3];
                    unsigned long long wk = 1ull;
                    bool run = false, slow = false;
                    uint64_t gpos = 0;
                    if (pass == 0) {
                        srk = first_item ? 0u : L_tlist[i - rbase];
                        gpos = tbase + srk;
                        run = gpos < p.nbytes;
                        tried |= 1u << k;
                    } else {
                        if (srk == 0xFFFFFFFFu) srk = first_item ? 0u : L_tlist[i - rbase];
                        gpos = tbase + srk;
                        const bool wanted = (i & 3u) == r0 && first_line + P + e + 1 - extra <= lim && gpos < p.nbytes;
                        if (!wanted) { srk = 0xFFFFFFFEu; }
                        else if ((rk >> 62) == 3) {                  // not settled by the speculative pass
                            run = slow = true;
                            if (TASSEL) {
                                gpos = parse_header(gpos, wk);
                                if (gpos == ~0ull) { srk = 0xFFFFFFFEu; run = false; }   // no sequence line follows
                            } else if ((tried >> k) & 1u) {
                                // deferred by the fast path: a non-blank non-base first byte inside the
                                // staged window is simply "no barcode"
                                if (!is_blank(p.buf[gpos]) && (srk >> 4) + p.nch <= win_ch) { run = false; rk = R_NONE; }
                            }
                        }
                    }
                    if (run) rk = match_read(gpos, srk, slow);
                    if (k == 0) { sr[0] = srk; res[0] = rk; wgt[0] = wk; }
                    else if (k == 1) { sr[1] = srk; res[1] = rk; wgt[1] = wk; }
                    else if (k == 2) { sr[2] = srk; res[2] = rk; wgt[2] = wk; }
                    else { sr[3] = srk; res[3] = rk; wgt[3] = wk; }
                }
            }
            // ------------ commit the wanted lines of this batch
#pragma unroll
            for (int k = 0; k < IPL; k++) {
                if (p.win) {                                         // (uniform; the whole wave goes through it)
                    const bool have = sr[k] < 0xFFFFFFFEu;
                    // item e = b0 + IPL * tid + k is line first_line + P + e + 1 - extra; a sequence line L is read L >> 2
                    win_add_wave(p, have, (first_line + P + b0 + IPL * tid + k + 1 - extra) >> 2, have ? (uint32_t)(res[k] >> 62) : 0u);
                }
                if (sr[k] >= 0xFFFFFFFEu) continue;                  // no item, or not a wanted line
                st_reads++;
                const uint32_t kind = (uint32_t)(res[k] >> 62);
                if (kind >= 1) st_bar++;
                if (kind == 2) {
                    st_tag++;
                    if (!(p.dbg & DBG_NO_ATOMIC)) {
                        const size_t cell = (size_t)(res[k] & R_CELL);
                        if (TASSEL) __hip_atomic_fetch_add(p.counts64 + cell, wgt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else __hip_atomic_fetch_add(p.counts + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            TD_STAMP(6);   // phase 2 + commit
        }

        // ---------------- rare: bytes >= 0x80 in the tile -- are any inside a counted sequence line?
        if (tile_has_hi) {
            const uint64_t Lb = first_line + P + excl;   // index of the line this thread's span starts in
            const uint32_t span0 = tid * CPT * 16u;
            uint32_t seen = 0;
#pragma unroll
            for (int k = 0; k < CPT / 2; k++) {
                const uint32_t m = mm[k];
#pragma nounroll
                for (uint32_t q = 0; q < 32u; q++) {
                    const uint64_t g = tbase + span0 + 32u * k + q;
                    if (g < p.nbytes && p.buf[g] >= 0x80u) {
                        const uint64_t line = Lb + seen;
                        if ((line & 3) == 1 && line <= p.limit_line) atomicOr(p.stats + ST_ERR, ERR_NONASCII);
                    }
                    seen += (m >> q) & 1u;
                }
            }
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
#ifdef TD_PHASE_PROF
    if (tid == 0)
        for (int i = 0; i < PROF_PHASES; i++) atomicAdd(p.stats + 8 + i, prof_acc[i]);
#endif
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
#ifndef TD_INST_ONLY      // (defined once, in tagdig.hip's translation unit)
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint64_t *tile_counts, uint32_t ntiles, uint64_t *state,
                                                     unsigned long long *total_out) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    // (eight consecutive tiles per thread and round: a round of 8192 tiles costs the same three barriers as one of 1024)
    constexpr uint32_t PER = 8;
    for (uint32_t base = 0; base < ntiles; base += 1024 * PER) {
        const uint32_t i0 = base + (uint32_t)tid * PER;
        unsigned long long v[PER], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) { v[q] = i0 + q < ntiles ? tile_counts[i0 + q] : 0ull; sum += v[q]; }
        unsigned long long inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long off = carry;
        for (int w = 0; w < wave; w++) off += wsum[w];
        unsigned long long run = off + inc - sum;                 // terminators before this thread's first tile
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) { run += v[q]; if (i0 + q < ntiles) state[i0 + q] = FLAG_INC | run; }
        __syncthreads();
        if (tid == 1023) carry = off + inc;
        __syncthreads();
    }
    if (tid == 0 && total_out) *total_out = carry;
}
#endif

}  // namespace tdk
