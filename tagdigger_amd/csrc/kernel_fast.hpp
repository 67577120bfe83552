// The free-running variant of the count kernel: no tickets, no look-back, no waiting.
//
// The only thing that couples one tile of FASTQ to the bytes before it is the line phase
// (lineindex % 4 of reference tagdigger_fun.py:254).  k_count resolves it exactly in flight
// (decoupled look-back); that makes every tile wait for the slowest of ~1000 predecessors.
// Here each tile instead
//   1. PREDICTS its phase from its own content: per line phase, how many lines open with eight
//      valid bases -- sequence lines do, headers ('@'), '+' lines and quality strings do not;
//   2. counts under that prediction, and records its terminator count and the prediction;
// then k_resolve scans the per-tile terminator counts (a few hundred thousand words), derives
// every tile's TRUE phase, and queues a correction for every tile whose prediction was wrong
// (also for tiles that reach past the maxreads limit, and for tiles holding bytes >= 0x80);
// finally k_fast runs again over that queue in fix-up mode, subtracting what was counted under
// the wrong phase and adding the right one.  The result is bit-exact for ANY input; only the
// speed depends on the prediction, and on well-formed FASTQ the queue is empty.
//
// Per tile (tile = CPT*4 KiB, one workgroup of 256 threads; the next tile's bytes are already
// in flight into registers while this one is processed):
//   A  masks (line terminators) + 2-bit packing of every 16-byte chunk, all lanes   -> LDS;
//      at its end the line each thread left PENDING in the previous tile is finished (its tag
//      bucket has arrived meanwhile), the next tile's loads are issued, then the pending count
//   B  block scan of terminator counts: in-tile ordinal of every terminator; each thread's vote
//   C  phase vote (the waves' packed votes rotated by their running totals), or the given phase
//   D  the tile's wanted lines, compacted through an LDS list so that all lanes work, are matched
//      up to the tag-bucket loads (barcode directory in LDS, tag buckets in L2), which stay in flight
#pragma once
#include "kernels.hpp"

namespace tdk {

constexpr uint32_t TI_COUNT_MASK = 0xFFFFFFu;   // tile_info: terminators in the tile
constexpr uint32_t TI_R0_SHIFT = 24;            //            phase (r0) the tile was counted under
constexpr uint32_t TI_HI = 1u << 26;            //            tile holds a byte >= 0x80
constexpr uint32_t TI_SKIP = 1u << 27;          //            the main pass (k_fast2) only counted the tile's terminators
// fix-up queue entry: {tile, code, P lo, P hi}; code = r0 | flags
constexpr uint32_t FX_NEG = 4;                  // subtract instead of add
constexpr uint32_t FX_LIMIT = 8;                // apply the maxreads limit (needs P)
constexpr uint32_t FX_HICHECK = 16;             // only look for bytes >= 0x80 inside counted sequence lines
constexpr uint32_t FX_PREDICT = 32;             // (main pass) predict the phase
constexpr uint32_t FX_WINONLY = 64;             // only add the tile's wanted lines to the progress windows (nothing is counted)

struct FParams {
    KParams k;
    uint32_t *tile_info;     // [ntiles]
    uint4 *fixlist;          // [fix_cap]
    uint32_t *nfix;
    uint32_t fix_cap;
    // progress windows (KParams::win): per tile, how many of its wanted lines had a barcode (low half) and a tag
    // (high half) as the main pass (k_fast2) saw them -- zeroed by the host before the launch, added to by every
    // phase-D pass.  k_resolve adds the sums of the tiles whose phase was right to the window their reads fall
    // into; the one tile in ~450 that straddles a window boundary goes to the fix-up pass (FX_WINONLY), which
    // knows every line's number.
    uint32_t *tile_sums;
};

// Wave priority per phase (KParams::prio: bits 1:0 phase A and the tile's end, 3:2 phases B-C,
// 5:4 phase D, 7:6 the end of phase A: pending line, next tile's loads).  The short, serial,
// latency-bound phases (scan, vote, matching) run above the long arithmetic one: a wave that gets through them sooner has its memory requests out sooner, and
// phase A of the co-resident workgroups fills the issue slots it leaves.
__device__ __forceinline__ void set_prio(uint32_t level) {
    if (level == 0) __builtin_amdgcn_s_setprio(0);
    else if (level == 1) __builtin_amdgcn_s_setprio(1);
    else if (level == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// FIX = false: the main pass over all tiles (phase predicted, nothing to subtract, no limit);
// FIX = true: the fix-up pass over the queue k_resolve left.
// threads per workgroup of k_fast (a power of two; the tile is CPT * FBLOCK chunks)
#ifndef TD_FAST_BLOCK
#define TD_FAST_BLOCK 256
#endif
#ifndef TD_FAST_WAVES_PER_SIMD
#define TD_FAST_WAVES_PER_SIMD TD_WAVES_PER_SIMD
#endif
constexpr int FBLOCK = TD_FAST_BLOCK;
// (FB: threads per workgroup -- 128 x 6 chunks is the 12 KiB tile of k_fast4's fix-up pass)
template <int CPT, int W, bool FIX, int FB = TD_FAST_BLOCK>
__global__ __launch_bounds__(FB, TD_FAST_WAVES_PER_SIMD) void k_fast(const FParams fp) {
    const KParams &p = fp.k;
    constexpr int FBLOCK = FB;
    static_assert(FB == 128 || FB == 256, "k_fast: 128 or 256 threads");
    constexpr int TILE_CH = CPT * FBLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t halo_ch = p.halo / 16u;
    const uint32_t win_ch = TILE_CH + halo_ch;
    uint2 *L_conv = reinterpret_cast<uint2 *>(lds);
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds + (size_t)win_ch * 8u);
    uint16_t *L_inv = L_mask + TILE_CH;                  // per chunk: which bytes are not bases (for the phase vote)
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(lds + (size_t)win_ch * 8u + TILE_CH * 4u);   // 64 dwords
    uint8_t *L_bidx = reinterpret_cast<uint8_t *>(L_misc + 64);
    TileCtx cx{L_conv, win_ch, reinterpret_cast<const unsigned long long *>(L_bidx),
               reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta),
               reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir)};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += FBLOCK)
        reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];
    if (tid == 0) L_misc[1] = 0;

    int st_reads = 0, st_bar = 0, st_tag = 0;       // per thread and launch: far below 2^31 (signed: the fix-up pass subtracts)
    // Main pass, 64-byte buckets: a thread's (single) wanted line is matched up to the tag-bucket loads
    // in its own tile and finished -- compares, count -- in the NEXT tile's phase D, so the probe's
    // latency runs under that tile's phases A-C instead of stalling the wave.
#ifdef TD_NO_PIPE
    constexpr bool PIPE = false;                      // (experiment: trades the pipelined probe for registers)
#else
    constexpr bool PIPE = !FIX && W <= 3;
#endif
    Pending<W> pd;
    bool pd_valid = false;
    // compares of the pending line; returns its count cell, or ~0 (the caller commits: the atomic is
    // issued after the next tile's loads, where the wait for it to leave the wave overlaps the barrier)
    auto finish_pending = [&]() -> uint64_t {
        vm_settled();           // (the bucket has had a whole phase A to arrive; nothing younger is in flight)
        TD_MSTAMP(cx, 16, 0);   // pending: wait for the bucket
        const uint64_t res = match_finish<W>(p, pd);
        TD_MSTAMP(cx, 17, 0);   // pending: compares
        const uint32_t kind = (uint32_t)(res >> 62);
        st_reads += 1;
        if (kind >= 1) st_bar += 1;
        if (kind == 2) st_tag += 1;
        return kind == 2 ? (res & R_CELL) : ~0ull;
    };
    // (base + 32-bit offset -- the host only takes this path with a matrix below 4 GiB -- and the
    // offset register is kept alive, see cell_off's use after phase C: the wave must not stall
    // overwriting it while the atomic still waits to read it)
    uint32_t cell_off = 0;
    auto commit_cell = [&](uint64_t cell) {
        if (cell != ~0ull && !(p.dbg & DBG_NO_ATOMIC)) {
            cell_off = (uint32_t)cell * 4u;
            __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(p.counts) + cell_off), 1u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
#ifdef TD_PHASE_PROF
    unsigned long long prof_acc[PROF_PHASES] = {};
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
    cx.pacc = prof_acc; cx.plast = &prof_last;
#endif
    const unsigned long long carried = p.cursor_in ? *p.cursor_in : 0ull;
    const uint64_t first_line = p.first_line + carried;
    const uint32_t nwork = FIX ? min(*fp.nfix, fp.fix_cap) : p.ntiles;

    // work item -> tile, code, P
    uint32_t it = blockIdx.x;
    uint32_t t = 0, code = FX_PREDICT;
    uint64_t Pg = 0;
    auto fetch_item = [&](uint32_t w, uint32_t &tt, uint32_t &cc, uint64_t &pp) {
        if (w >= nwork) return;
        if (FIX) { const uint4 e = fp.fixlist[w]; tt = e.x; cc = e.y; pp = ((uint64_t)e.w << 32) | e.z; }
        else { tt = w; cc = FX_PREDICT; pp = 0; }
    };
    uint4 v[CPT];
    const bool has_halo = (uint32_t)tid < halo_ch;
    auto tile_base = [&](uint32_t tile) -> const uint8_t * {
        return tile >= p.tail_tile ? p.tail_buf + (uint64_t)(tile - p.tail_tile) * TILE : p.buf + (uint64_t)tile * TILE;
    };
    // Every tile is loaded with plain, independent 16-byte loads: the tiles whose window crosses the
    // buffer's end come from the zero-padded copy the host made of the buffer's tail.  They are
    // buffer loads: a scalar descriptor for the tile's window, a scalar offset per load, and ONE
    // per-thread offset register that never changes.  (A load reads its address registers only
    // when the memory pipeline gets to it; a wave that overwrites them before that -- the next
    // load's address, say -- stalls until it has.)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t voff = (uint32_t)tid * 16u;
    auto tile_rsrc = [&](uint32_t tile) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(tile_base(tile)), 0, (int)(TILE + 4096u), 0x00020000);
    };
    auto fetch_tile = [&](uint32_t tile) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile);
        if (p.nt_loads) {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * (FBLOCK * 16), 2 /* nt */);
                v[j] = make_uint4(q.x, q.y, q.z, q.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * (FBLOCK * 16), 0);
                v[j] = make_uint4(q.x, q.y, q.z, q.w);
            }
        }
    };
    fetch_item(it, t, code, Pg);
    // Stagger the workgroups that share a CU so that their compute-bound (A) and latency-bound (D)
    // phases interleave instead of marching in lockstep: co-resident workgroups are (observed, for
    // speed only) blockIdx apart by the number of CUs.
    if (p.stagger) {
        const uint32_t slot = (blockIdx.x / p.stagger_div) & 3u;
        for (uint32_t q = 0; q < slot * p.stagger; q++) __builtin_amdgcn_s_sleep(64);   // 64*64 cycles each
    }
    if (it < nwork) fetch_tile(t);
    __syncthreads();

    while (it < nwork) {
        const uint64_t tbase = (uint64_t)t * TILE;
        const uint32_t codeq = FIX ? code : (uint32_t)FX_PREDICT;
        const uint32_t tile_rem = (uint32_t)min(p.nbytes - tbase, (uint64_t)(TILE + 4096u));   // bytes from the tile's base to the buffer's end (capped)
        TD_STAMP(0);   // loop head

        // ---------------- A: terminator masks + packing of every chunk, from registers.  A wave whose
        // chunks are all inside the buffer, all ASCII and free of '\r' (the normal case) takes the
        // short forms; any other wave redoes its chunks with the exact general forms.
        {
            // the halo (the first chunks of the next tile) is fetched now and packed at the end of A:
            // it is not carried across the rest of the loop like the tile's own prefetched bytes
            uint4 vh = make_uint4(0u, 0u, 0u, 0u);
            if (has_halo) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(tile_rsrc(t), voff, (int)TILE, 0);
                vh = make_uint4(q.x, q.y, q.z, q.w);
            }
            // (phase A ends by consuming vh unconditionally -- halo_seen -- so that the compiler knows
            // the load has landed and does not guard later writes of these registers with waits)
            auto halo_seen = [&]() { asm volatile("" ::"v"(vh.x), "v"(vh.y), "v"(vh.z), "v"(vh.w)); };
            uint32_t hiacc = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++) hiacc |= v[j].x | v[j].y | v[j].z | v[j].w;
            const bool inside = tbase + TILE + p.halo <= p.nbytes;
            uint32_t cr_absent = 0x80808080u;
            bool general = !inside || __any((hiacc & 0x80808080u) != 0);
            if (!general) {
#pragma unroll
                for (int j = 0; j < CPT; j++) cr_absent_ascii(v[j], cr_absent);
                general = __any((cr_absent & 0x80808080u) != 0x80808080u);
                if (!general) {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t c = j * FBLOCK + tid;
                        const uint2 pk = convert_chunk_ascii(v[j]);
                        L_mask[c] = (uint16_t)nl_mask16_ascii(v[j]);
                        L_inv[c] = (uint16_t)pk.y;
                        L_conv[c] = pk;
                    }
                    if (has_halo) L_conv[TILE_CH + tid] = convert_chunk(vh);
                }
            }
            if (__builtin_expect(general, 0)) {
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t c = j * FBLOCK + tid;
                    const uint64_t g = tbase + (uint64_t)c * 16u;
                    uint32_t nl = eq_mask16(v[j], 0x0A0A0A0Au);
                    uint32_t cr = eq_mask16(v[j], 0x0D0D0D0Du);
                    uint32_t term = nl | (cr & ~(nl >> 1));
                    if (cr & 0x8000u) {                    // \r in the chunk's last byte: \r\n across chunks?
                        uint64_t nx = g + 16;
                        if (nx < p.nbytes && p.buf[nx] == 0x0A) term &= 0x7FFFu;
                    }
                    if (g + 16 > p.nbytes) term &= g < p.nbytes ? ((1u << (uint32_t)(p.nbytes - g)) - 1u) : 0u;
                    const uint2 pk = convert_chunk(v[j]);
                    L_mask[c] = (uint16_t)term;
                    L_inv[c] = (uint16_t)pk.y;
                    L_conv[c] = pk;
                }
                if (has_halo) L_conv[TILE_CH + tid] = convert_chunk(vh);
                if (hiacc & 0x80808080u) L_misc[1] = 1;
            }
            halo_seen();
        }
        // the next work item's bytes: in flight while this tile is finished
        const uint32_t nit = it + gridDim.x;
        uint32_t tn = 0, coden = FX_PREDICT;
        uint64_t Pgn = 0;
        TD_STAMP(1);   // A: wait for this tile's bytes, masks + packing
        // the line this thread left pending in the previous tile: its bucket has had phase A to arrive,
        // and nothing younger is in flight yet
        set_prio((p.prio >> 6) & 3u);
        uint64_t pcell = ~0ull;
        if (PIPE && pd_valid) { pcell = finish_pending(); pd_valid = false; }
        fetch_item(nit, tn, coden, Pgn);
        if (nit < nwork) fetch_tile(tn);
        if (PIPE) commit_cell(pcell);
        TD_STAMP(5);   // pending line committed, next tile's loads issued
        lds_barrier();
        TD_STAMP(2);   // barrier A
        set_prio((p.prio >> 2) & 3u);

        // ---------------- B: terminators of this thread's CPT consecutive chunks, block scan
        const bool tile_has_hi = L_misc[1] != 0;
        uint32_t mm[CPT / 2];
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) mm[i] = reinterpret_cast<const uint32_t *>(L_mask)[tid * (CPT / 2) + i];
        uint32_t ivw[CPT / 2 + 1];                           // invalid-byte bitmap of the span, 32 bytes per word
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) ivw[i] = reinterpret_cast<const uint32_t *>(L_inv)[tid * (CPT / 2) + i];
        ivw[CPT / 2] = 0;                                    // (bytes past the span count as bases: votes only)
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) cnt += __builtin_popcount(mm[i]);
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t span0 = tid * CPT * 16u;

        // ---------------- C: phase.  Each thread votes with the FIRST line that starts in its span
        // (256 samples per tile are plenty): does it open with eight valid bases?  The line follows the
        // terminator with in-tile ordinal wave_base + (incl - cnt); the wave base is only known after
        // the barrier, so each wave publishes its four counts by WAVE-LOCAL class next to its total and
        // every thread rotates them afterwards: one barrier serves both the scan and the vote.
        uint32_t r0 = codeq & 3u;
        const bool predict = (codeq & FX_PREDICT) != 0 && t != 0;
        {
            uint32_t fpos = 0, lo = 0, hi = 0;
            bool found = false;
#pragma unroll
            for (int k = CPT / 2 - 1; k >= 0; k--) {
                if (mm[k]) { fpos = 32u * k + __builtin_ctz(mm[k]); lo = ivw[k]; hi = ivw[k + 1]; found = true; }
            }
            // line start = fpos + 1 (<= 32 past the word's base): a 64-bit funnel
            const uint64_t win = (((uint64_t)hi << 32) | lo) >> ((fpos & 31u) + 1u);
            const bool vote_good = found && (win & 0xFFu) == 0 && span0 + fpos + 9u <= tile_rem;
            const uint32_t lclass = (incl - cnt) & 3u;
            // 4 x 8-bit counts (<= 64 each) of the good votes per class: three ballots, the rest scalar
            const uint64_t bg = __ballot(vote_good), b0 = __ballot((lclass & 1u) != 0), b1 = __ballot((lclass & 2u) != 0);
            const uint32_t packed = (uint32_t)__builtin_popcountll(bg & ~b0 & ~b1) | ((uint32_t)__builtin_popcountll(bg & b0 & ~b1) << 8) |
                                    ((uint32_t)__builtin_popcountll(bg & ~b0 & b1) << 16) | ((uint32_t)__builtin_popcountll(bg & b0 & b1) << 24);
            if (lane == 63) { L_misc[4 + wave] = incl; L_misc[4 + FBLOCK / 64 + wave] = packed; }
        }
        lds_barrier();
        uint32_t wbase = 0, total = 0;
        uint32_t v02 = 0, v13 = 0;                       // 16-bit fields: votes of in-tile classes 0,2 and 1,3
#pragma unroll
        for (int w = 0; w < FBLOCK / 64; w++) {
            const uint32_t x = L_misc[4 + w], pk = L_misc[4 + FBLOCK / 64 + w];
            // wave w's local class c is in-tile class (c + total-so-far) & 3: rotate its four byte fields
            const uint32_t rot = 8u * (total & 3u);
            const uint32_t r = rot ? ((pk << rot) | (pk >> (32u - rot))) : pk;
            v02 += r & 0x00FF00FFu;
            v13 += (r >> 8) & 0x00FF00FFu;
            if (w < wave) wbase += x;
            total += x;
        }
        const uint32_t votes[4] = {v02 & 0xFFFFu, v13 & 0xFFFFu, v02 >> 16, v13 >> 16};
        const uint32_t excl = wbase + incl - cnt;
        TD_STAMP(3);   // B: scan + vote
        if (predict) {
            uint32_t best = votes[0]; r0 = 0;
            if (votes[1] > best) { best = votes[1]; r0 = 1; }
            if (votes[2] > best) { best = votes[2]; r0 = 2; }
            if (votes[3] > best) { best = votes[3]; r0 = 3; }
        } else if (codeq & FX_PREDICT) {
            r0 = (4u - (uint32_t)(first_line & 3)) & 3u;      // tile 0: P = 0, the phase is known
        }
        TD_STAMP(4);   // C: vote
        set_prio((p.prio >> 4) & 3u);
        if (PIPE) asm volatile("" ::"v"(cell_off));       // (keeps the atomic's offset register untouched until here)
        if (tid == 0 && !FIX)
            fp.tile_info[t] = total | (r0 << TI_R0_SHIFT) | (tile_has_hi ? TI_HI : 0u);

        const uint64_t P = FIX ? Pg : 0;                            // valid in fix-up mode only
        const bool use_limit = (codeq & FX_LIMIT) != 0;
        const int sign = (codeq & FX_NEG) ? -1 : (FIX && (codeq & FX_WINONLY)) ? 0 : 1;
        const bool to_windows = FIX && p.win && sign >= 0;     // (FX_NEG undoes a count whose record never reached the windows)

        if (__builtin_expect((codeq & FX_HICHECK) != 0, 0)) {
            // ------------ bytes >= 0x80 inside a counted sequence line?  (fix-up mode, true phase)
            const uint64_t Lb = first_line + P + excl;
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
        } else {
            // ------------ D: the wanted lines that start in this thread's span: in-tile ordinal i is
            // followed by a wanted line iff i == r0 (mod 4), i.e. local class (r0 - excl) & 3.  Tile 0
            // also owns the buffer's first line (ordinal -1 == 3 mod 4).
            // The thread's wanted lines follow its local terminators number lc, lc+4, ... : their count
            // is a closed form and the first one is the lc-th set bit of the span mask (no loop).
            const uint32_t lc = (r0 - excl) & 3u;
            uint32_t nw = cnt > lc ? (cnt - lc + 3u) >> 2 : 0u;
            uint32_t w0 = 0;
            {
                uint32_t r = lc, kbase = 0, m = mm[0];
#pragma unroll
                for (int k = 0; k < CPT / 2 - 1; k++) {
                    const uint32_t c = __builtin_popcount(mm[k]);
                    const bool next = kbase == 32u * k && r >= c;      // still in word k and it holds fewer than r+1 bits
                    if (next) { r -= c; kbase = 32u * (k + 1); m = mm[k + 1]; }
                }
                const uint32_t m1 = m & (m - 1), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);
                const uint32_t sel = r == 0 ? m : r == 1 ? m1 : r == 2 ? m2 : m3;
                w0 = span0 + kbase + (sel ? __builtin_ctz(sel) : 0u) + 1u;
            }
            const bool own_first = t == 0 && tid == 0 && p.nbytes > 0 && r0 == 3u;
            // General enumeration of this thread's q-th wanted line (limit-aware); the common case
            // -- no limit, at most one wanted line in the span -- never calls it.
            auto nth_wanted = [&](uint32_t q, uint32_t &out, uint64_t &line) -> bool {
                uint32_t seen = 0;
                if (own_first && (!use_limit || first_line + P <= p.limit_line)) { if (q == 0) { out = 0; line = first_line + P; return true; } seen = 1; }
                uint32_t i = excl;
#pragma unroll
                for (int k = 0; k < CPT / 2; k++) {
                    uint32_t m = mm[k];
                    while (m) {
                        const uint32_t bit = __builtin_ctz(m);
                        m &= m - 1;
                        const uint32_t sr = span0 + 32u * k + bit + 1u;
                        if ((i & 3u) == r0 && tbase + sr < p.nbytes &&
                            (!use_limit || first_line + P + i + 1 <= p.limit_line)) {
                            if (seen == q) { out = sr; line = first_line + P + i + 1; return true; }
                            seen++;
                        }
                        i++;
                    }
                }
                return false;
            };
            if (p.dbg & DBG_NO_PHASE2) nw = 0;
            auto commit = [&](uint64_t res, uint64_t line) {
                const uint32_t kind = (uint32_t)(res >> 62);
                if (to_windows) win_add(p, line >> 2, kind);       // (sequence line L is read L >> 2)
                st_reads += sign;
                if (kind >= 1) st_bar += sign;
                if (kind == 2) {
                    st_tag += sign;
                    if (sign != 0 && !(p.dbg & DBG_NO_ATOMIC))
                        __hip_atomic_fetch_add(p.counts + (size_t)(res & R_CELL), (uint32_t)sign, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            // A regular tile -- inside the buffer, not its first, no limit to apply, at most one wanted
            // line per thread of the workgroup -- is matched with FULL lanes: wanted line number j of
            // the tile (the one after the j-th terminator whose ordinal is r0 mod 4; a closed form of
            // the scan, no second scan) goes to thread j, through a list in LDS (the masks' space,
            // dead since the scan barrier).  Only about 58 % of the threads find a wanted line in
            // their own 128-byte span; compacted, the matcher's instructions are issued for 2.3
            // waves' worth of lines instead of 4.  Which wave takes which quarter of the list rotates
            // with the tile number, so that no SIMD is favoured.
            const uint32_t nwant = (total + 3u - r0) >> 2;              // wanted ordinals below `total`
            const bool regular = !use_limit && t != 0 && nwant <= (uint32_t)TILE_CH && tbase + TILE + p.halo <= p.nbytes &&
                                 !(p.dbg & DBG_NO_PHASE2);
            bool general = !regular && !(p.dbg & DBG_NO_PHASE2);
            TD_MSTAMP(cx, 12, 0);   // wanted-line selection
            if (regular) {
                uint16_t *L_list = L_mask;
                if (nw) {
                    const uint32_t slot0 = (excl + 3u - r0) >> 2;           // wanted ordinals below this thread's first
                    L_list[slot0] = (uint16_t)w0;
                    if (__builtin_expect(nw > 1, 0)) {                        // (short lines: rare)
                        uint32_t li = 0;
#pragma unroll
                        for (int k = 0; k < CPT / 2; k++) {
                            uint32_t m = mm[k];
                            while (m) {
                                const uint32_t bit = __builtin_ctz(m);
                                m &= m - 1;
                                if (li > lc && ((li - lc) & 3u) == 0) L_list[slot0 + ((li - lc) >> 2)] = (uint16_t)(span0 + 32u * k + bit + 1u);
                                li++;
                            }
                        }
                    }
                }
                lds_barrier();
                // hot: one line per thread and round (one round unless the lines are shorter than
                // ~60 bytes), matched from the packed chunks.  A thread's LAST line is left pending
                // in the pipelined form: no other memory access follows it here, the bucket loads
                // stay in flight.
                const uint32_t j0 = ((uint32_t)tid + 64u * (t & (uint32_t)(FBLOCK / 64 - 1))) & (uint32_t)(FBLOCK - 1);
#pragma nounroll
                for (uint32_t j = j0; j < nwant; j += FBLOCK) {
                    const uint32_t srel = L_list[j];
                    // (a 32-bit status: 1 pending, 0 no barcode, 2 barcode only, 6 raw bytes needed)
                    const uint32_t k = (uint32_t)(match_prepare<W, ML_FAST>(p, cx, tbase + srel, srel, false, pd) >> 61);
                    const uint64_t line = first_line + P + r0 + 4u * j + 1u;     // (P: fix-up pass only)
                    if (k == 1u) {
                        if (PIPE && j + FBLOCK >= nwant) pd_valid = true;
                        else { commit(match_finish<W>(p, pd), line); vm_settled(); }
                    } else if (__builtin_expect(k == 6u, 0)) {
                        // cold: needs its raw bytes (leading blanks to strip, a first byte that is not a
                        // base; a non-blank non-base first byte simply comes back as "no barcode")
                        commit(match_line<W, ML_SLOW>(p, cx, tbase + srel, srel, true), line);
                        vm_settled();
                    } else {
                        if (to_windows && k == 2u) win_add(p, line >> 2, 1u);
                        st_reads += sign;
                        if (k == 2u) st_bar += sign;
                    }
                }
                TD_STAMP(15);           // hot part done
            }
            // cold: the buffer's first and last tiles, the maxreads limit (fix-up pass), tiles with more
            // wanted lines than the list holds: every thread walks the wanted lines of its own span
            if (__builtin_expect(general, 0)) {
#pragma nounroll
                for (uint32_t q = 0;; q++) {
                    uint32_t srel = 0;
                    uint64_t line = 0;
                    if (!nth_wanted(q, srel, line)) break;
                    const uint32_t c0f = srel >> 4;
                    const bool deferred = c0f + p.nch > win_ch || ((L_conv[c0f].y >> (srel & 15u)) & 1u);
                    commit(match_line<W, ML_BOTH>(p, cx, tbase + srel, srel, deferred), line);
                }
                vm_settled();
            }
        }

        set_prio(p.prio & 3u);
        TD_STAMP(6);   // D: match + commit (thread 0's share)
        // ---------------- next work item
        if (tid == 0) L_misc[1] = 0;
        lds_barrier();                                    // LDS is reused by the next tile
        TD_STAMP(7);   // end barrier (other waves' matching)
        it = nit; t = tn; code = coden; Pg = Pgn;
    }
    if (PIPE && pd_valid) commit_cell(finish_pending());   // lines left pending by the last tile
#ifdef TD_PHASE_PROF
    if (tid == 0 && !FIX)
        for (int i = 0; i < PROF_PHASES; i++) atomicAdd(p.stats + 8 + i, prof_acc[i]);
#endif

    // ---------------- statistics: one atomic per wave (two's complement carries the fix-up signs)
    unsigned long long r = wave_sum64((unsigned long long)(long long)st_reads), b = wave_sum64((unsigned long long)(long long)st_bar),
                       g = wave_sum64((unsigned long long)(long long)st_tag);
    if (lane == 0) {
        if (r) atomicAdd(p.stats + ST_READS, r);
        if (b) atomicAdd(p.stats + ST_BARCUT, b);
        if (g) atomicAdd(p.stats + ST_TAG, g);
    }
}

// An exclusive scan of the per-tile terminator counts gives every tile's true line phase; tiles
// counted under another phase, tiles reaching past the maxreads limit and tiles holding bytes
// >= 0x80 are queued for the fix-up pass.  Also leaves the running line total.
// k_resolve runs as ceil(ntiles / 1024) blocks of 1024 threads, one tile per thread.  A block needs
// the number of terminators before its first tile: k_resolve_sums reduces every block's 1024 tile
// counts into super[block] first, and each k_resolve block adds up the super sums before its own.
constexpr uint32_t RESOLVE_SPAN = 1024;

#ifndef TD_INST_ONLY      // (defined once, in tagdig.hip's translation unit)
__global__ __launch_bounds__(256) void k_resolve_sums(const FParams fp, unsigned long long *super) {
    __shared__ unsigned long long wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = blockIdx.x * RESOLVE_SPAN + tid * 4;
    unsigned long long local = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
        if (i0 + k < fp.k.ntiles) local += fp.tile_info[i0 + k] & TI_COUNT_MASK;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d, 64);
    if (lane == 0) wsum[wave] = local;
    __syncthreads();
    if (tid == 0) super[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
#endif

#ifndef TD_INST_ONLY      // (defined once, in tagdig.hip's translation unit)
__global__ __launch_bounds__(1024) void k_resolve(const FParams fp, const unsigned long long *super) {
    const KParams &p = fp.k;
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long wpre[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long carried = p.cursor_in ? *p.cursor_in : 0ull;
    const uint64_t first_line = p.first_line + carried;
    const bool finite = p.limit_line < (~0ull - 16);
    auto push = [&](uint32_t tile, uint32_t code, uint64_t P) {
        const uint32_t slot = atomicAdd(fp.nfix, 1u);
        if (slot < fp.fix_cap) fp.fixlist[slot] = make_uint4(tile, code, (uint32_t)P, (uint32_t)(P >> 32));
        else atomicOr(p.stats + ST_ERR, ERR_SPIN);   // cannot happen: the queue holds 3 entries per tile
    };
    // terminators before this block's first tile
    unsigned long long pre = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += 1024) pre += super[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d, 64);
    const uint32_t i = blockIdx.x * RESOLVE_SPAN + tid;
    const uint32_t info = i < p.ntiles ? fp.tile_info[i] : 0u;
    const unsigned long long v = info & TI_COUNT_MASK;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    if (lane == 0) wpre[wave] = pre;
    __syncthreads();
    unsigned long long off = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { off += wpre[w]; if (w < wave) off += wsum[w]; }
    const uint64_t P = off + inc - v;                                  // terminators before tile i
    if (i < p.ntiles) {
        const uint32_t truth = (4u - (uint32_t)((first_line + P) & 3)) & 3u;   // ordinals == truth (mod 4) precede sequence lines
        const uint32_t pred = (info >> TI_R0_SHIFT) & 3u;
        // lines of this tile: first_line+P (only tile 0's own first line) .. first_line+P+v
        const bool beyond_all = finite && first_line + P + (i == 0 ? 0 : 1) > p.limit_line;
        const bool beyond_some = finite && first_line + P + v > p.limit_line;
        const bool skipped = (info & TI_SKIP) != 0;      // nothing was counted: nothing to take back
        if (beyond_all) {
            if (!skipped) push(i, pred | FX_NEG, P);
        } else if (beyond_some) {
            if (!skipped) push(i, pred | FX_NEG, P);
            push(i, truth | FX_LIMIT, P);
        } else if (skipped) {
            push(i, truth, P);
        } else if (pred != truth) {
            push(i, pred | FX_NEG, P);
            push(i, truth, P);
        }
        if ((info & TI_HI) && !beyond_all) push(i, truth | FX_HICHECK, P);
    }
    if (i == p.ntiles - 1) {
        atomicAdd(p.stats + ST_LINES, P + v);
        if (p.cursor_out) *p.cursor_out = carried + P + v;
    }
    // Progress windows: a tile the main pass counted under the right phase (nothing queued for it above) hands in its
    // sums when all its wanted lines -- reads (first_line + P + truth + 1) / 4 .. + nwant - 1 -- fall into one window;
    // a tile that straddles a window boundary is queued for the fix-up pass, which adds line by line (FX_WINONLY).
    // The 64 tiles of a wave are neighbours: their reads fall into one or two windows -- two atomics per wave.
    if (p.win && fp.tile_sums) {                                    // (uniform)
        const uint32_t truth = (4u - (uint32_t)((first_line + P) & 3)) & 3u;
        const bool mine = i < p.ntiles && !(info & TI_SKIP) && ((info >> TI_R0_SHIFT) & 3u) == truth &&
                          !(finite && first_line + P + v > p.limit_line);
        const uint64_t rid0 = (first_line + P + truth + 1) >> 2;
        const uint32_t nwant = mine ? ((uint32_t)v + 3u - truth) >> 2 : 0u;
        const uint64_t w0 = rid0 / PROG_WINDOW;
        const bool single = nwant != 0 && (rid0 + nwant - 1) / PROG_WINDOW == w0;
        if (nwant != 0 && !single) push(i, truth | FX_WINONLY, P);
        const uint32_t sums = single ? fp.tile_sums[i] : 0u;
        const unsigned long long pk = (unsigned long long)(sums & 0xFFFFu) | ((unsigned long long)(sums >> 16) << 32);
        const uint64_t wb = wave_min64(single ? w0 : ~0ull);
        unsigned long long a = single && w0 == wb ? pk : 0ull, b = single && w0 == wb + 1 ? pk : 0ull;
        if (single && w0 > wb + 1 && pk && w0 < p.win_cap) atomicAdd(p.win + w0, pk);       // (lines of a few bytes)
        a = wave_sum64(a); b = wave_sum64(b);
        if (lane == 0 && wb != ~0ull) {
            if (a && wb < p.win_cap) atomicAdd(p.win + wb, a);
            if (b && wb + 1 < p.win_cap) atomicAdd(p.win + wb + 1, b);
        }
    }
}
#endif

// the main pass's per-tile terminator counts as k_scan_tiles reads them (td_count_and_split_device: the splitter's line
// prefix without a second pass over the bytes)
#ifndef TD_INST_ONLY      // (defined once, in tagdig.hip's translation unit)
__global__ __launch_bounds__(256) void k_info_counts(const uint32_t *tile_info, uint32_t ntiles, uint64_t *tile_counts) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < ntiles) tile_counts[i] = tile_info[i] & TI_COUNT_MASK;
}
#endif

}  // namespace tdk
