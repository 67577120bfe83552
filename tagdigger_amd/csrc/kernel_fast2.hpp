// k_fast2: the main pass of the free-running path, second generation ("lazy packing").
//
// k_fast (kernel_fast.hpp) 2-bit-packs and validates EVERY byte of the tile in its phase A -- 59 VALU
// instructions per 16-byte chunk, two thirds of the kernel's instructions (profiles/r01_final) -- although
// only the head of every fourth line is ever matched (reference tagdigger_fun.py:254-261: the sequence line,
// and of it barcode + site + tag).  Here phase A only finds the line terminators; the tile's RAW bytes are
// parked in LDS, and each wanted line is packed by the lane that matches it, from its own first byte
// (unaligned LDS reads: no alignment shifts either):
//
//   A   per chunk: raw bytes -> LDS, '\n' mask (15 VALU) -> LDS; bytes >= 0x80 and '\r' are detected per wave
//       ('\r' takes a second mask and the \r\n rule; bytes >= 0x80 the exact forms, and the tile is left to
//       the fix-up pass); at its end: the line left PENDING by the previous tile is finished, the next
//       tile's loads are issued, the pending count goes through the hot-cell cache
//   B   block scan of the terminator counts; each thread's phase vote from the RAW first 8 bytes of the first
//       line that starts in its span
//   C   the tile's phase (as k_fast); tiles that are not "regular" (first / last of the buffer, bytes >= 0x80,
//       more wanted lines than the list holds) are only COUNTED here (terminators) and flagged TI_SKIP:
//       k_resolve queues them for the fix-up pass (k_fast<.., true>), which handles every irregularity
//   D   wanted lines compacted through an LDS list (as k_fast); per line: 16-byte pieces from the line's first
//       byte -> codes + validity -> barcode directory (LDS) -> tag words -> hash -> bucket loads left in flight
//
// Count updates go through a small per-wave cache of hot cells in LDS (hc_commit): a cell that keeps coming
// back is counted there and written out once per flush, so that skewed libraries do not serialise on one
// address in L2 (measured: Zipf 1.5 over the tags costs k_fast 45 ms instead of 12; same-address atomics
// retire at ~12 ns each whatever the number of CUs issuing them).
#pragma once
#include "kernel_fast.hpp"

namespace tdk {

// timing-only ablations (td_set_option "debug_ablate") are compiled into k_fast2 only with -DTD_ABLATE: in the shipped
// kernel they would be branches on a kernel argument inside the tile loop
#ifdef TD_ABLATE
#define TD_DBG(p) ((p).dbg)
#else
#define TD_DBG(p) 0u
#endif

constexpr int HC_SLOTS = 128;                   // hot-cell cache: slots per wave (direct mapped)
constexpr uint32_t HC_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t HC_AGE_TILES = 64;           // every so many tiles the cache is written out and cleared
constexpr uint32_t HC_REST = 15;                // ageing periods a wave leaves its cache off after one without hits
constexpr uint32_t HC_BYTES_PER_WAVE = HC_SLOTS * 8 + HC_SLOTS;

__device__ __forceinline__ uint32_t hc_hash(uint32_t cell) {
    return (__umul24(cell, 0x9E3779u) >> 13) & (uint32_t)(HC_SLOTS - 1);          // v_mul_u32_u24: full rate
}

// One count per lane with `hit`, through the wave's hot-cell cache (S: HC_SLOTS x {cell, count}, E: one byte
// per slot for electing a writer).  All lanes of the wave execute this together; LDS operations of one wave
// are performed in order, which is all the synchronisation there is.
//   cell cached               -> count it in LDS (no global traffic)
//   not cached, slot is cold  -> ONE of the lanes that want the slot writes the old entry out and takes it
//   otherwise                 -> the plain global atomic
__device__ __forceinline__ void hc_commit(uint32_t *counts, uint2 *S, uint8_t *E, bool hit, uint32_t cell, uint32_t lane, bool use_cache) {
    if (!use_cache) {
        if (hit) __hip_atomic_fetch_add(counts + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint32_t h = hc_hash(cell);
    uint2 e = make_uint2(HC_EMPTY, 0u);
    if (hit) e = S[h];
    const bool same = hit && e.x == cell;
    if (same) __hip_atomic_fetch_add(&S[h].y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool miss = hit && !same;
    const bool repl = miss && e.y <= 1u;
    if (repl) E[h] = (uint8_t)lane;
    wave_lds_fence();
    bool win = false;
    if (repl) win = E[h] == (uint8_t)lane;
    if (win) {
        const uint2 old = S[h];                         // (includes what this step's `same` lanes added)
        S[h] = make_uint2(cell, 1u);
        if (old.y) __hip_atomic_fetch_add(counts + old.x, old.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (miss) {
        __hip_atomic_fetch_add(counts + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_lds_fence();
}
__device__ __forceinline__ void hc_flush(uint32_t *counts, uint2 *S, uint32_t lane) {
#pragma unroll
    for (int q = 0; q < HC_SLOTS / 64; q++) {
        const uint2 e = S[q * 64 + lane];
        if (e.y) __hip_atomic_fetch_add(counts + e.x, e.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S[q * 64 + lane] = make_uint2(HC_EMPTY, 0u);
    }
    wave_lds_fence();
}

// 16-bit mask of bytes equal to the byte replicated in c4, for a chunk of bytes < 0x80
__device__ __forceinline__ uint32_t eq_mask16_ascii(const uint4 &v, uint32_t c4, uint32_t k7f) {
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t u = xor_add(x[d], c4, k7f) & 0x80808080u;       // 0x80 where the byte is NOT c
        if (d == 0) lo = udot4(u, 0x08040201u, 0u);
        else if (d == 1) lo = udot4(u, 0x80402010u, lo);
        else if (d == 2) hi = udot4(u, 0x08040201u, 0u);
        else hi = udot4(u, 0x80402010u, hi);
    }
    return ((lo >> 7) | (hi << 1)) ^ 0xFFFFu;
}

// codes (16 bases, first base in the top bits) and invalid flags (bit k = byte k is not a base) of 16 ASCII bytes.
// Per dword seven instructions: t = x & 0x06060606 is TWICE the base code of every byte (A 0, C 2, T 4, G 6) -- it
// selects the expected letter from an 8-byte table without a shift (v_perm), and its dot product with 64, 16, 4, 1
// is twice the four codes' 8 bits (the shift is folded into the words' placement); flags as before.  The two flag
// dot products of a half are chained through the addend (the five pieces of a line give the scheduler independent
// chains to interleave).
__device__ __forceinline__ uint2 pack16_ascii(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t x[4] = {x0, x1, x2, x3};
    uint32_t c[4], nz[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t t = x[d] & 0x06060606u;
        const uint32_t expect = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, t);          // selector 0 A, 2 C, 4 T, 6 G
        nz[d] = (((x[d] & 0xDFDFDFDFu) ^ expect) + 0x7F7F7F7Fu) & 0x80808080u;               // 0x80 where the byte is not a base
        c[d] = udot4(t, 0x01041040u, 0u);                                                    // 2 x (four codes = 8 bits)
    }
    const uint32_t flo = udot4(nz[1], 0x80402010u, udot4(nz[0], 0x08040201u, 0u));           // 128 x (flags of bytes 0-7)
    const uint32_t fhi = udot4(nz[3], 0x80402010u, udot4(nz[2], 0x08040201u, 0u));           // 128 x (flags of bytes 8-15)
    const uint32_t cw = (((c[0] << 23) | (c[1] << 15)) | (c[2] << 7)) | (c[3] >> 1);
    return make_uint2(cw, (fhi << 1) | (flo >> 7));
}

// 16 bytes of LDS at any byte offset
__device__ __forceinline__ uint4 lds_read16(const uint8_t *p) {
    uint4 q;
    __builtin_memcpy(&q, p, 16);
    return q;
}

// The matcher's first half for a line whose first byte sits at L_raw[srel] (raw ASCII bytes; at least
// 16 * NQ bytes are staged behind it).  Returns 0 no barcode, 2 barcode+site only, 1 pending (pd filled,
// bucket loads in flight), 6 the line opens with a blank (str.strip, reference :256): the caller re-reads it
// from global memory.
// ISSUE false: everything but the bucket loads (pd.R, pd.nr, pd.boff are filled; bucket_issue asks for the bucket later --
// k_fast4's consumers finish the lines of their last tile in between).
// the NQ 16-byte pieces of the line whose first byte sits at L_raw[srel], requested together (line_prepare_q takes them on)
template <int NQ>
__device__ __forceinline__ void line_read(const uint8_t *L_raw, uint32_t srel, uint4 (&q)[NQ]) {
    const uint8_t *src = L_raw + srel;
#pragma unroll
    for (int i = 0; i < NQ; i++) q[i] = lds_read16(src + 16 * i);
}
template <int W, int NQ, bool ISSUE = true>
__device__ __forceinline__ uint32_t line_prepare_q(const KParams &p, const TileCtx &cx, const uint4 (&q)[NQ], Pending<W> &pd);
template <int W, int NQ, bool ISSUE = true>
__device__ __forceinline__ uint32_t line_prepare(const KParams &p, const TileCtx &cx, const uint8_t *L_raw, uint32_t srel,
                                                 Pending<W> &pd) {
    uint4 q[NQ];
    line_read<NQ>(L_raw, srel, q);
    return line_prepare_q<W, NQ, ISSUE>(p, cx, q, pd);
}
template <int W, int NQ, bool ISSUE>
__device__ __forceinline__ uint32_t line_prepare_q(const KParams &p, const TileCtx &cx, const uint4 (&q)[NQ], Pending<W> &pd) {
    // NQ: 16-byte pieces packed from the line's first byte (compile time: all reads are issued before the
    // first piece is packed, and the pieces' dependent chains interleave)
    static_assert(NQ >= 2 && NQ <= 2 * W + 4, "pieces per line");
    uint32_t S[NQ + 4];
    uint32_t inv[(NQ + 1) / 2];
#pragma unroll
    for (int i = 0; i < (NQ + 1) / 2; i++) inv[i] = 0;
    TD_MSTAMP(cx, 9, 1);     // D: line start from the list + the line's pieces from LDS (waited for)
    const uint32_t first = q[0].x & 0xFFu;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const uint2 e = pack16_ascii(q[i].x, q[i].y, q[i].z, q[i].w);
        S[i] = e.x;
        if (i & 1) inv[i >> 1] |= e.y << 16; else inv[i >> 1] |= e.y;
    }
#pragma unroll
    for (int i = NQ; i < NQ + 4; i++) S[i] = 0;
    if constexpr ((NQ & 1) != 0) inv[NQ >> 1] |= 0xFFFF0000u;
    if (inv[0] & 1u) return is_blank(first) ? 6u : 0u;      // the first byte is no base: blanks to strip, or no match
    uint32_t nvalid = 0;
    {
        bool found = false;
#pragma unroll
        for (int i = 0; i < (NQ + 1) / 2; i++) {
            if (!found) {
                if (inv[i]) { nvalid += __builtin_ctz(inv[i]); found = true; }
                else nvalid += 32;
            }
        }
    }
    TD_MSTAMP(cx, 10, 0);    // D: pack + nvalid
    // ---- barcode + cut site (reference :257)
    const uint64_t K = ((uint64_t)S[0] << 32) | S[1];
    uint32_t ci = cx.L_bdir[S[0] >> (32 - 2 * BDIR_BASES)];
    uint32_t meta = 0;
    bool bhit = false;
    if (ci != 0xFFFFu) {
        for (;;) {
            const uint32_t m = cx.L_bmeta[ci];
            const uint32_t len = m & 63u;
            if (len <= nvalid && ((K ^ cx.L_bval[ci]) >> (64u - 2u * len)) == 0) { meta = m; bhit = true; break; }
            if (m & BMETA_LAST) break;
            ci++;
        }
    }
    TD_MSTAMP(cx, 11, 1);    // D: barcode directory walk
    if (!bhit) return 0u;
    const uint32_t off = (meta >> 6) & 63u, row = meta >> 16;
    if (nvalid <= off) return 2u;
    const uint32_t nrem = nvalid - off;
    // ---- the read from the tag offset on (reference :260), as 64-bit words
    const uint32_t wo = off >> 4, sh = 2u * (off & 15u);
    for (uint32_t t = 0; t < p.maxwo; t++) {
        if (wo > t) {
#pragma unroll
            for (int w = 0; w < NQ + 3; w++) S[w] = S[w + 1];
            S[NQ + 3] = 0;
        }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        // (pieces beyond NQ read as zero codes: they lie past every stored tag's end for this index)
        const uint32_t s0 = 2 * w < NQ + 4 ? S[2 * w] : 0u, s1 = 2 * w + 1 < NQ + 4 ? S[2 * w + 1] : 0u, s2 = 2 * w + 2 < NQ + 4 ? S[2 * w + 2] : 0u;
        const uint64_t hi = ((uint64_t)s0 << 32) | s1;
        const uint64_t lo = ((uint64_t)s1 << 32) | s2;
        pd.R[w] = ((uint64_t)(uint32_t)((hi << sh) >> 32) << 32) | (uint32_t)((lo << sh) >> 32);
    }
    constexpr int BUCKET_U4_ = W <= 3 ? TD_BU4 : 8;
    pd.nr = min(nrem, 0x7FFFu) | (row << 16);
    if (nrem >= p.m_bases && !(TD_DBG(p) & DBG_NO_PROBE)) {
        pd.nr |= PD_PROBE;
        const uint32_t hk = hash_key(pd.R[0] >> (64u - 2u * p.m_bases));
        const uint32_t bk = hk & p.bucket_mask;
        pd.boff = bk * (uint32_t)(BUCKET_U4_ * 16);
        const uint4 *bp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(p.buckets) + pd.boff);
        pd.boff |= hk >> 27;                                // (the key's bit in the buckets' overflow filters)
        if (ISSUE) {
#pragma unroll
            for (int q = 0; q < BUCKET_U4_; q++) pd.b[q] = bp[q];   // in flight: first used by match_finish
        }
    } else if (p.nshort == 0) {
        return 2u;
    }
    TD_MSTAMP(cx, 12, 0);    // D: tag words, hash, bucket loads issued
    return 1u;
}
// the loads line_prepare<W, NQ, false> left out
template <int W>
__device__ __forceinline__ void bucket_issue(const KParams &p, Pending<W> &pd) {
    constexpr int BUCKET_U4_ = W <= 3 ? TD_BU4 : 8;
    if (pd.nr & PD_PROBE) {
        const uint4 *bp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(p.buckets) + (pd.boff & ~(uint32_t)(BUCKET_U4_ * 16 - 1)));
#pragma unroll
        for (int q = 0; q < BUCKET_U4_; q++) pd.b[q] = bp[q];
    }
}

// wave priority per phase (s_setprio): the pending line + next tile's loads, phases C-D, phases A-B of the next tile
#ifndef TD_P_PEND
#define TD_P_PEND 3
#endif
#ifndef TD_P_D
#define TD_P_D 2
#endif
#ifndef TD_P_A
#define TD_P_A 0
#endif
#ifndef TD_P_B
#define TD_P_B 1
#endif
#ifndef TD_FAST2_WAVES
#define TD_FAST2_WAVES 0        // 0: derived from the tile size (LDS decides how many workgroups share a CU)
#endif
template <int CPT> struct Fast2Waves { static constexpr int value = TD_FAST2_WAVES ? TD_FAST2_WAVES : (CPT <= 6 ? 4 : 3); };

// Layout of a tile over the workgroup: wave w owns the contiguous quarter [w, w + 1) * TILE / 4 -- it loads it
// (1 KiB per load instruction: coalesced), writes its raw bytes and terminator masks to LDS and reads back
// only its OWN masks, so phases A and B need no workgroup barrier between them; it also lists the starts of
// the lines that follow its terminators, by wave-local ordinal, in the space of its masks.  ONE barrier then
// publishes raw bytes, lists, totals and votes; after it wanted line j of the tile -- the line behind the
// terminator with in-tile ordinal r0 + 4 j -- is found by three compares against the waves' running totals
// and matched by thread j (full lanes).  Two barriers per tile (that one, and the one that frees the LDS).
template <int CPT, int W, int NQ, bool PROG>
__global__ __launch_bounds__(FBLOCK, Fast2Waves<CPT>::value) void k_fast2(const FParams fp) {
    const KParams &p = fp.k;
    constexpr int TILE_CH = CPT * FBLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;
    constexpr uint32_t WCH = CPT * 64;                // chunks per wave
    constexpr uint32_t WBYTES = WCH * 16;
    static_assert(FBLOCK == 256 && (CPT % 2) == 0, "k_fast2 is written for 256 threads and an even number of chunks per thread");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t halo = p.halo;                                       // bytes staged behind the tile (multiple of 64, >= 16 * NQ + 16)
    uint8_t *L_raw = lds;
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds + TILE + halo);  // terminator mask per chunk; then, per wave, its list of line starts
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(lds + TILE + halo + TILE_CH * 2u);   // 64 dwords
    uint8_t *L_hc = reinterpret_cast<uint8_t *>(L_misc + 64);
    uint8_t *L_bidx = L_hc + 4 * HC_BYTES_PER_WAVE;
    TileCtx cx{nullptr, 0u, reinterpret_cast<const unsigned long long *>(L_bidx),
               reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta),
               reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir)};
    // L_misc: [1] the tile holds a byte >= 0x80, [2] a wave has more terminators than its list holds,
    //         [3] the halo holds a byte >= 0x80, [4..7] wave totals, [8..11] wave votes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint2 *hcS = reinterpret_cast<uint2 *>(L_hc + wave * HC_BYTES_PER_WAVE);
    uint8_t *hcE = L_hc + wave * HC_BYTES_PER_WAVE + HC_SLOTS * 8;
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += FBLOCK)
        reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];
    if (tid == 0) { L_misc[0] = 0; L_misc[1] = 0; L_misc[2] = 0; L_misc[3] = 0; }
#pragma unroll
    for (int q = 0; q < HC_SLOTS / 64; q++) hcS[q * 64 + lane] = make_uint2(HC_EMPTY, 0u);
    // The hot-cell cache pays for itself only when cells repeat: each wave watches its own hit rate over an
    // ageing period and leaves the cache off for the next HC_REST periods when hardly anything hit.
    bool hc_on = p.hot_cache != 0;
    uint32_t hc_hits = 0, hc_rest = 0;

    int st_reads = 0, st_bar = 0, st_tag = 0;
    constexpr bool PIPE = W <= 3;
    Pending<W> pd;
    bool pd_valid = false;
    // progress windows: every phase-D pass adds what its lines matched -- barcode hits in the low half, tag hits in the
    // high half -- to the tile's sum in LDS (two slots, by the parity of the workgroup's iteration: the tag hits of a
    // pass's pending lines arrive in the NEXT iteration, before its barrier 1); after that barrier thread 0 writes the
    // previous tile's sum to FParams::tile_sums.  No wave-uniform state is carried (the kernel is short of SGPRs).
    //   L_misc[12 + parity] the sums, L_misc[14] the previous tile's index + 1 (0: it recorded nothing)
    constexpr bool prog = PROG;                             // (its own instantiation)
    uint32_t parity = 0;
    if (prog && tid == 0) { L_misc[12] = 0; L_misc[13] = 0; L_misc[14] = 0; }
    auto sums_add = [&](uint32_t slot, uint32_t add) {
        if (lane == 0 && add) __hip_atomic_fetch_add(L_misc + 12 + slot, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const unsigned long long carried = p.cursor_in ? *p.cursor_in : 0ull;
    const uint64_t first_line = p.first_line + carried;
    const uint32_t nwork = p.ntiles;

    uint4 v[CPT];
    const bool has_halo = (uint32_t)tid < halo / 16u;
    const uint32_t hoff = has_halo ? (uint32_t)tid * 16u : 0x40000000u;      // (beyond every descriptor's range: reads as zero)
    auto tile_base = [&](uint32_t tile) -> const uint8_t * {
        return tile >= p.tail_tile ? p.tail_buf + (uint64_t)(tile - p.tail_tile) * TILE : p.buf + (uint64_t)tile * TILE;
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t misc_addr = (uint32_t)(uintptr_t)L_misc;             // (LDS byte address: the low half of the flat address)
    const uint32_t voff = (uint32_t)wave * WBYTES + (uint32_t)lane * 16u;    // this thread's first chunk; load j is 1 KiB further
    auto tile_rsrc = [&](uint32_t tile) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(tile_base(tile)), 0, (int)(TILE + 4096u), 0x00020000);
    };
    auto fetch_tile = [&](uint32_t tile) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile);
        if (p.nt_loads) {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * 1024, 2 /* nt */);
                v[j] = make_uint4(q.x, q.y, q.z, q.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * 1024, 0);
                v[j] = make_uint4(q.x, q.y, q.z, q.w);
            }
        }
    };
    // one wanted line finished: statistics, and whether / where to count
    bool phit_tag = false;                                  // the line finish_pending finished had a tag
    auto finish_pending = [&](bool &hit, uint32_t &cell) {
        vm_settled();
        const uint64_t res = match_finish<W>(p, pd);
        const uint32_t kind = (uint32_t)(res >> 62);
        st_reads += 1;
        if (kind >= 1) st_bar += 1;
        if (kind == 2) st_tag += 1;
        phit_tag = kind == 2;
        hit = kind == 2 && !(TD_DBG(p) & DBG_NO_ATOMIC);
        cell = (uint32_t)res;
    };

#ifdef TD_PHASE_PROF
    unsigned long long prof_acc[PROF_PHASES] = {};
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
    cx.pacc = prof_acc; cx.plast = &prof_last;
#endif
    // Tiles are dealt to the workgroups in RUNS of p.run consecutive tiles (run r of the buffer goes to workgroup
    // r mod gridDim): inside a run the line phase is CARRIED from tile to tile -- exact given the run's first tile --
    // and only the first tile of a run votes (k_resolve checks every tile's phase all the same).
    const uint32_t RUN = p.run ? p.run : 1u;
    uint32_t run_pos = 0;                                  // position inside the current run
    uint32_t t = blockIdx.x * RUN, aged = 0, carry_r0 = 0;
    bool carry_ok = false;
    if (t < nwork) fetch_tile(t);
    __syncthreads();

    while (t < nwork) {
        const uint64_t tbase = (uint64_t)t * TILE;
        TD_STAMP(0);   // loop head
        // ---------------- A: this wave's quarter: raw bytes and terminator masks -> LDS
        bool wave_crb = false;           // a chunk of this wave ends with '\r' (wave-uniform)
        // the halo (the first bytes of the next tile): requested now by EVERY lane -- the lanes without a halo chunk ask
        // for an offset beyond the descriptor's range, which returns zeros without touching memory -- so that the load
        // is unconditional straight-line code and the waits for this tile's own bytes (older loads) can leave it in
        // flight; it is consumed after phase B and the pending line, just before the next tile's loads are issued
        const u32x4 vhq = __builtin_amdgcn_raw_buffer_load_b128(tile_rsrc(t), hoff, (int)TILE, 0);
        {
            uint32_t hiacc = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++) hiacc |= v[j].x | v[j].y | v[j].z | v[j].w;
#pragma unroll
            for (int j = 0; j < CPT; j++) *reinterpret_cast<uint4 *>(L_raw + voff + j * 1024) = v[j];
            uint16_t *Lm = L_mask + wave * WCH + lane;
            const bool general = __any((hiacc & 0x80808080u) != 0);
            if (__builtin_expect(!general, 1)) {
                uint32_t cr_absent = 0x80808080u;
#pragma unroll
                for (int j = 0; j < CPT; j++) cr_absent_ascii(v[j], cr_absent);
                const bool has_cr = __any((cr_absent & 0x80808080u) != 0x80808080u);
                if (__builtin_expect(!has_cr, 1)) {
#pragma unroll
                    for (int j = 0; j < CPT; j++) Lm[j * 64] = (uint16_t)nl_mask16_ascii(v[j]);
                } else {
                    uint32_t crs = 0;
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t nl = eq_mask16_ascii(v[j], 0x0A0A0A0Au, 0x7F7F7F7Fu), cr = eq_mask16_ascii(v[j], 0x0D0D0D0Du, 0x7F7F7F7Fu);
                        Lm[j * 64] = (uint16_t)(nl | (cr & ~(nl >> 1)));      // (a '\r' in the chunk's last byte: settled in phase B)
                        crs |= cr;
                    }
                    wave_crb = __any((crs & 0x8000u) != 0);
                }
            } else {
                uint32_t crs = 0;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t nl = eq_mask16(v[j], 0x0A0A0A0Au), cr = eq_mask16(v[j], 0x0D0D0D0Du);
                    Lm[j * 64] = (uint16_t)(nl | (cr & ~(nl >> 1)));
                    crs |= cr;
                }
                wave_crb = __any((crs & 0x8000u) != 0);
                if (hiacc & 0x80808080u) L_misc[1] = 1;
            }
        }
        TD_STAMP(1);   // A: wait for the tile's bytes, raw + masks -> LDS (stores issued)
        // (the stores of raw bytes and masks are on their way: the pending line, the next tile's loads and the count go
        // here, between them and the read-back of the masks -- none of it touches those bytes)
        // ---------------- the pending line of the previous tile, the next tile's loads, the pending count
        __builtin_amdgcn_s_setprio(TD_P_PEND);
        const uint32_t nit = run_pos + 1u < RUN ? t + 1u : t + 1u + (gridDim.x - 1u) * RUN;      // next tile of this workgroup
        bool phit = false;
        uint32_t pcell = 0;
        {
            bool tagged = false;
            if (PIPE && pd_valid) { finish_pending(phit, pcell); pd_valid = false; tagged = phit_tag; }
            if (prog) sums_add(parity ^ 1u, (uint32_t)__builtin_popcountll(__ballot(tagged)) << 16);
        }
        TD_STAMP(3);   // pending line: wait for its bucket, compares
        vm_settled();                                       // (the halo: requested at the tile's top, a phase A ago)
        if (has_halo) {
            *reinterpret_cast<uint4 *>(L_raw + TILE + (size_t)tid * 16u) = make_uint4(vhq.x, vhq.y, vhq.z, vhq.w);
            // (lines that begin in this tile are packed from these bytes with the ASCII forms)
            if ((vhq.x | vhq.y | vhq.z | vhq.w) & 0x80808080u) L_misc[3] = 1;
        }
        if (nit < nwork) fetch_tile(nit);
        if (PIPE) {
            if (hc_on) {
                // (cache hits of this step, for the hit-rate watch: lanes whose cell is already cached)
                const uint32_t h = hc_hash(pcell);
                hc_hits += (uint32_t)__builtin_popcountll(__ballot(phit && hcS[h].x == pcell));
            }
            hc_commit(p.counts, hcS, hcE, phit, pcell, (uint32_t)lane, hc_on);
        }
        TD_STAMP(4);   // next tile's loads issued, pending count committed
        __builtin_amdgcn_s_setprio(TD_P_B);
        wave_lds_fence();          // this wave's masks and raw bytes are in LDS (nothing of another wave is read before the barrier)

        // ---------------- B: terminators of this thread's CPT consecutive chunks, wave scan, vote, list of line starts
        uint32_t mm[CPT / 2];
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) mm[i] = reinterpret_cast<const uint32_t *>(L_mask)[tid * (CPT / 2) + i];
        const uint32_t span0 = tid * CPT * 16u;
        const uint32_t wend = ((uint32_t)wave + 1u) * WBYTES;               // end of this wave's quarter
        {
            // a chunk whose last byte is '\r' (bit 15 of its mask is set for it -- or for a '\n' there): one terminator
            // with the '\n' that opens the next chunk, if there is one
            if (__builtin_expect(wave_crb, 0)) {
#pragma unroll
                for (int i = 0; i < CPT / 2; i++) {
#pragma unroll
                    for (int hbit = 15; hbit < 32; hbit += 16) {
                        if ((mm[i] >> hbit) & 1u) {
                            const uint32_t at = span0 + 32u * i + (uint32_t)hbit;
                            if (L_raw[at] == 0x0Du) {
                                // (the next byte may belong to another wave's quarter, or to the halo: not in LDS yet)
                                const uint32_t nx = at + 1u < wend ? (uint32_t)L_raw[at + 1u] : (uint32_t)tile_base(t)[at + 1u];
                                if (nx == 0x0Au) mm[i] &= ~(1u << hbit);
                            }
                        }
                    }
                }
                vm_settled();
            }
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) cnt += __builtin_popcount(mm[i]);
        const uint32_t incl = wave_incl_scan(cnt, lane);
        {
            uint32_t fpos = 0;
            bool found = false;
#pragma unroll
            for (int k = CPT / 2 - 1; k >= 0; k--) {
                if (mm[k]) { fpos = 32u * k + __builtin_ctz(mm[k]); found = true; }
            }
            uint32_t packed = 0;
            if (!carry_ok) {
                // the first line that starts in this span: do its first eight bytes (inside this wave's quarter) read as bases?
                const uint32_t ls = span0 + fpos + 1u;
                uint2 q8 = make_uint2(0u, 0u);
                if (found && ls + 8u <= wend) __builtin_memcpy(&q8, L_raw + ls, 8);
                const uint32_t c0 = (q8.x >> 1) & 0x03030303u, c1 = (q8.y >> 1) & 0x03030303u;
                const uint32_t d0 = (q8.x & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0u, 0x47544341u, c0);
                const uint32_t d1 = (q8.y & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0u, 0x47544341u, c1);
                const bool vote_good = (d0 | d1) == 0;
                const uint32_t lclass = (incl - cnt) & 3u;
                const uint64_t bg = __ballot(vote_good), b0 = __ballot((lclass & 1u) != 0), b1 = __ballot((lclass & 2u) != 0);
                packed = (uint32_t)__builtin_popcountll(bg & ~b0 & ~b1) | ((uint32_t)__builtin_popcountll(bg & b0 & ~b1) << 8) |
                         ((uint32_t)__builtin_popcountll(bg & ~b0 & b1) << 16) | ((uint32_t)__builtin_popcountll(bg & b0 & b1) << 24);
            }
            // the lines behind this wave's terminators, by wave-local ordinal, into the space of its masks (every lane of
            // the wave holds its masks in registers by now)
            uint16_t *Ll = L_mask + wave * WCH;
            uint32_t k = incl - cnt;
            // Two terminators per 32-byte word without a branch (FASTQ: the "+" line's two, two bytes apart) -- a
            // lane without one stores to a spare word; words with more (lines of a few bytes) take the loop, for
            // the whole wave only when some lane has one.  A wave with more terminators than its list holds
            // writes nothing: the tile goes to the fix-up pass (L_misc[2] below).
            const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (__builtin_expect(wtot <= WCH, 1)) {
                uint16_t *spare = reinterpret_cast<uint16_t *>(L_misc + 60);
                uint32_t rest = 0;
#pragma unroll
                for (int i = 0; i < CPT / 2; i++) {
                    const uint32_t m = mm[i], m1 = m & (m - 1u);
                    const uint32_t base = span0 + 32u * i + 1u;
                    uint16_t *d0 = m ? Ll + k : spare;
                    *d0 = (uint16_t)(base + (uint32_t)__builtin_ctz(m | 0x80000000u));
                    k += m ? 1u : 0u;
                    uint16_t *d1 = m1 ? Ll + k : spare;
                    *d1 = (uint16_t)(base + (uint32_t)__builtin_ctz(m1 | 0x80000000u));
                    k += m1 ? 1u : 0u;
                    rest |= m1 & (m1 - 1u);
                }
                if (__builtin_expect(__any(rest != 0), 0)) {
                    k = incl - cnt;
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
            }
            if (lane == 63) {
                L_misc[4 + wave] = incl; L_misc[8 + wave] = packed;
                if (incl > WCH) L_misc[2] = 1;
            }
        }
        TD_STAMP(2);   // B: masks, scan, vote, list
        lds_barrier();
        TD_STAMP(5);   // barrier 1
        __builtin_amdgcn_s_setprio(TD_P_D);

        // ---------------- C: the tile's phase and the waves' running totals
        // (flags and wave totals: two reads issued together and waited for once -- left to the compiler they become
        // four scalarised reads, each waited for before the next is issued)
        u32x4 flgq, totq;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(flgq), "=&v"(totq) : "v"(misc_addr) : "memory");
        const uint32_t t_x = (uint32_t)__builtin_amdgcn_readfirstlane((int)totq.x), t_y = (uint32_t)__builtin_amdgcn_readfirstlane((int)totq.y),
                       t_z = (uint32_t)__builtin_amdgcn_readfirstlane((int)totq.z), t_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)totq.w);
        const uint4 flg4 = make_uint4(0u, (uint32_t)__builtin_amdgcn_readfirstlane((int)flgq.y), (uint32_t)__builtin_amdgcn_readfirstlane((int)flgq.z),
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)flgq.w));
        const uint32_t wb1 = t_x, wb2 = wb1 + t_y, wb3 = wb2 + t_z, total = wb3 + t_w;
        uint32_t r0;
        if (carry_ok) {
            r0 = carry_r0;
        } else if (t != 0) {
            const uint4 pk4 = *reinterpret_cast<const uint4 *>(L_misc + 8);
            auto rot = [](uint32_t pk, uint32_t by) { const uint32_t r = 8u * (by & 3u); return r ? ((pk << r) | (pk >> (32u - r))) : pk; };
            const uint32_t a = pk4.x, b = rot(pk4.y, wb1), c = rot(pk4.z, wb2), d = rot(pk4.w, wb3);
            const uint32_t v02 = (a & 0x00FF00FFu) + (b & 0x00FF00FFu) + (c & 0x00FF00FFu) + (d & 0x00FF00FFu);
            const uint32_t v13 = ((a >> 8) & 0x00FF00FFu) + ((b >> 8) & 0x00FF00FFu) + ((c >> 8) & 0x00FF00FFu) + ((d >> 8) & 0x00FF00FFu);
            const uint32_t votes[4] = {v02 & 0xFFFFu, v13 & 0xFFFFu, v02 >> 16, v13 >> 16};
            uint32_t best = votes[0]; r0 = 0;
            if (votes[1] > best) { best = votes[1]; r0 = 1; }
            if (votes[2] > best) { best = votes[2]; r0 = 2; }
            if (votes[3] > best) { best = votes[3]; r0 = 3; }
        } else {
            r0 = (4u - (uint32_t)(first_line & 3)) & 3u;
        }
        const bool tile_has_hi = flg4.y != 0;
        const uint32_t nwant = (total + 3u - r0) >> 2;
        const bool regular = t != 0 && (flg4.y | flg4.z | flg4.w) == 0 && tbase + TILE + halo <= p.nbytes;
        if (tid == 0) {
            fp.tile_info[t] = total | (r0 << TI_R0_SHIFT) | (tile_has_hi ? TI_HI : 0u) | (regular ? 0u : TI_SKIP);
            if (prog) {
                // (every add to the other slot -- the previous tile's passes, its pending lines -- came before barrier 1)
                const uint32_t prev = L_misc[14];
                if (prev) fp.tile_sums[prev - 1u] = L_misc[12 + (parity ^ 1u)];
                L_misc[12 + (parity ^ 1u)] = 0;
                L_misc[14] = regular ? t + 1u : 0u;
            }
        }
        TD_STAMP(6);   // C: phase

        // ---------------- D: wanted line j follows the terminator with in-tile ordinal r0 + 4 j
        if (regular && !(TD_DBG(p) & DBG_NO_PHASE2)) {
            const uint32_t j0 = ((uint32_t)tid + 64u * (t & 3u)) & (uint32_t)(FBLOCK - 1);
#pragma nounroll
            for (uint32_t j = j0; j < nwant; j += FBLOCK) {
                // (the wave whose list holds ordinal o, and o's place in it: selects, no branches)
                const uint32_t o = r0 + 4u * j;
                uint32_t sel = 0u;
                sel = o >= wb1 ? 1u * WCH - wb1 : sel;
                sel = o >= wb2 ? 2u * WCH - wb2 : sel;
                sel = o >= wb3 ? 3u * WCH - wb3 : sel;
                const uint32_t srel = L_mask[o + sel];
                // (a line that starts in the tile's last bytes is still whole in the staged window: the halo
                // holds 16 NQ bytes and more)
                const uint32_t k = line_prepare<W, NQ, false>(p, cx, L_raw, srel, pd);
                // k: 0 no barcode, 2 barcode only, 1 the tag is to be looked up, 6 leading blank (rare: raw bytes re-read)
                st_reads += k != 6u ? 1 : 0;
                st_bar += k == 2u ? 1 : 0;
                bool barred = k == 1u || k == 2u, tagged = false;
                const bool keep = PIPE && j + FBLOCK >= nwant;                   // (the same for every lane of the wave)
                // (the rare line that opens with a blank, here -- before this pass's buckets are asked for: their sixteen
                // registers are free for the slow matcher, and nothing it waits for is in flight)
                if (__builtin_expect(__any(k == 6u), 0)) {
                    if (k == 6u) {
                        const uint64_t res = match_line<W, ML_SLOW>(p, cx, tbase + srel, srel, true);
                        const uint32_t kind = (uint32_t)(res >> 62);
                        st_reads += 1;
                        if (kind >= 1) { st_bar += 1; barred = true; }
                        if (kind == 2) {
                            st_tag += 1;
                            tagged = true;
                            if (!(TD_DBG(p) & DBG_NO_ATOMIC))
                                __hip_atomic_fetch_add(p.counts + (size_t)(res & R_CELL), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    vm_settled();
                }
                if (k == 1u) {
                    bucket_issue<W>(p, pd);
                    if (keep) { pd_valid = true; st_reads -= 1; }               // (counted when it is finished)
                    else {
                        bool h; uint32_t c;
                        st_reads -= 1;
                        finish_pending(h, c);
                        tagged = phit_tag;
                        hc_commit(p.counts, hcS, hcE, h, c, (uint32_t)lane, false);
                        vm_settled();
                    }
                }
                if (prog) {
                    // (the wave's first lane holds its smallest j, so it is here whenever any lane is; a pending line's
                    // barcode counts now, its tag when it is finished)
                    sums_add(parity, (uint32_t)__builtin_popcountll(__ballot(barred)) | ((uint32_t)__builtin_popcountll(__ballot(tagged)) << 16));
                }
            }
        }
        __builtin_amdgcn_s_setprio(TD_P_A);
        TD_STAMP(7);   // D: lines packed and matched up to the bucket loads (thread 0's share)
        if (tid == 0) { L_misc[1] = 0; L_misc[2] = 0; L_misc[3] = 0; }
        if (p.hot_cache && ++aged == HC_AGE_TILES) {
            aged = 0;
            if (hc_on) {
                hc_flush(p.counts, hcS, (uint32_t)lane);
                // a wave commits ~TILE / 4 / 300 hits a tile: below ~3 % of them cached, the cache is only overhead
                if (p.hot_cache != 2u && hc_hits * 32u < HC_AGE_TILES * (TILE / 1200u)) { hc_on = false; hc_rest = HC_REST; }   // (2: never, for measurements)
                hc_hits = 0;
            } else if (--hc_rest == 0) hc_on = true;
        }
        lds_barrier();                                    // LDS is reused by the next tile
        TD_STAMP(8);   // end barrier (the other waves' matching)
        // the next tile of the run starts (total) terminators further on: its wanted ordinals are r0 - total (mod 4)
        carry_r0 = (r0 - total) & 3u;
        carry_ok = nit == t + 1u && t != 0;
        run_pos = run_pos + 1u < RUN ? run_pos + 1u : 0u;
        t = nit;
        parity ^= 1u;
    }
    {
        bool tagged = false;
        if (PIPE && pd_valid) {
            bool h; uint32_t c;
            finish_pending(h, c);
            tagged = phit_tag;
            hc_commit(p.counts, hcS, hcE, h, c, (uint32_t)lane, false);
        }
        if (prog) sums_add(parity ^ 1u, (uint32_t)__builtin_popcountll(__ballot(tagged)) << 16);
    }
    if (p.hot_cache) hc_flush(p.counts, hcS, (uint32_t)lane);
    if (prog) {
        lds_barrier();                                      // (the last tile's pending lines have added their tag hits)
        if (tid == 0) {
            const uint32_t prev = L_misc[14];
            if (prev) fp.tile_sums[prev - 1u] = L_misc[12 + (parity ^ 1u)];
        }
    }
#ifdef TD_PHASE_PROF
    if (tid == 0)
        for (int i = 0; i < PROF_PHASES; i++) atomicAdd(p.stats + 8 + i, prof_acc[i]);
#endif

    unsigned long long r = wave_sum64((unsigned long long)(long long)st_reads), b = wave_sum64((unsigned long long)(long long)st_bar),
                       g = wave_sum64((unsigned long long)(long long)st_tag);
    if (lane == 0) {
        if (r) atomicAdd(p.stats + ST_READS, r);
        if (b) atomicAdd(p.stats + ST_BARCUT, b);
        if (g) atomicAdd(p.stats + ST_TAG, g);
    }
}

}  // namespace tdk

// (tag width in 64-bit words, 16-byte pieces packed per line) the host picks from: see pick_fast2_c in tagdig.hip
#define TD_FAST2_COMBOS(X) X(1, 3) X(1, 4) X(1, 6) X(2, 5) X(2, 6) X(2, 8) X(3, 7) X(3, 8) X(3, 10)
#ifdef TD_FAST2_EXTERN
// instantiated in inst_fast2.hip's translation units
#define TD_X4F(W, NQ) extern template __global__ void tdk::k_fast2<4, W, NQ, false>(const tdk::FParams);
#define TD_X6F(W, NQ) extern template __global__ void tdk::k_fast2<6, W, NQ, false>(const tdk::FParams);
#define TD_X8F(W, NQ) extern template __global__ void tdk::k_fast2<8, W, NQ, false>(const tdk::FParams);
#define TD_X4T(W, NQ) extern template __global__ void tdk::k_fast2<4, W, NQ, true>(const tdk::FParams);
#define TD_X6T(W, NQ) extern template __global__ void tdk::k_fast2<6, W, NQ, true>(const tdk::FParams);
#define TD_X8T(W, NQ) extern template __global__ void tdk::k_fast2<8, W, NQ, true>(const tdk::FParams);
TD_FAST2_COMBOS(TD_X4F) TD_FAST2_COMBOS(TD_X6F) TD_FAST2_COMBOS(TD_X8F) TD_FAST2_COMBOS(TD_X4T) TD_FAST2_COMBOS(TD_X6T) TD_FAST2_COMBOS(TD_X8T)
#undef TD_X4F
#undef TD_X6F
#undef TD_X8F
#undef TD_X4T
#undef TD_X6T
#undef TD_X8T
#endif
