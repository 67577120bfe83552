// The fast path cut in two by register footprint.
//
// k_fast does everything for a tile in one workgroup loop; with the matcher inlined it needs 128
// VGPRs, so only 4 waves fit per SIMD, and each wave is a long dependent instruction chain
// (measured: ~50 % SIMD busy, the rest latency).  Here the same work is split:
//
//   k_emit   streams the FASTQ once: terminator masks, 2-bit packing, scan, phase vote -- exactly
//            phases A-C of k_fast -- and then, instead of matching, writes one small RECORD per
//            wanted line: the read's first bases as aligned 16-base words + its count of leading
//            valid bases (24 B at the bench workload).  No matcher code, far fewer registers.
//   k_match  one lane per record, every lane busy, no tile in LDS (only the barcode index):
//            barcode+site lookup, tag hash, bucket compare, atomic.  Lean and fully occupied, so
//            the L2/Infinity-Cache latency of the bucket fetch is hidden by other waves.
//   k_slow   the rare lines that need their raw bytes (leading blanks, a first byte that is not
//            a base, a window past the staged chunks): position-only records, matched by
//            re-reading global memory.
//
// Records are written without any global atomic: every WAVE of k_emit owns a region of the record
// buffer and a cursor; k_match runs with the same geometry and wave w of workgroup b consumes
// region (b, w).  Tiles that do not fit the simple shape (more than one wanted line possible in a
// lane's 128-byte span, tile 0, the buffer's last tile) are flagged TI_DIRECT and counted by the
// fix-up pass of k_fast, as are mispredicted / limit-crossing / high-byte tiles (k_resolve).
#pragma once
#include "kernel_fast.hpp"

#ifndef TD_EMIT_WAVES_PER_SIMD
#define TD_EMIT_WAVES_PER_SIMD 4
#endif

namespace tdk {

constexpr uint32_t MATCH_SPLIT = 4;           // waves of k_match per record region
constexpr uint32_t TI_DIRECT = 1u << 27;      // tile_info: nothing emitted for this tile; count it in the fix-up pass

struct SParams {
    FParams f;
    uint32_t *rec;            // fast records: [region][slot][rec_stride dwords]
    uint32_t *region_count;   // [regions] records written per region
    uint64_t *slow;           // slow records: absolute position of the line's first byte
    uint32_t *nslow;          // number of slow records (global append)
    uint32_t rec_stride;      // dwords per fast record: rec_words stream words + 1 (valid bases), rounded up to even
    uint32_t rec_words;       // aligned 16-base words kept per record
    uint32_t region_cap;      // records per region
    uint32_t slow_cap;
    uint32_t tile_begin, tile_end;   // the slab of tiles this launch covers
};

template <int CPT, int W>
__global__ __launch_bounds__(BLOCK, TD_EMIT_WAVES_PER_SIMD) void k_emit(const SParams sp) {
    const FParams &fp = sp.f;
    const KParams &p = fp.k;
    constexpr int TILE_CH = CPT * BLOCK;
    constexpr uint32_t TILE = TILE_CH * 16;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t halo_ch = p.halo / 16u;
    const uint32_t win_ch = TILE_CH + halo_ch;
    uint2 *L_conv = reinterpret_cast<uint2 *>(lds);
    uint16_t *L_mask = reinterpret_cast<uint16_t *>(lds + (size_t)win_ch * 8u);
    uint16_t *L_inv = L_mask + TILE_CH;
    uint32_t *L_misc = reinterpret_cast<uint32_t *>(lds + (size_t)win_ch * 8u + TILE_CH * 4u);   // 64 dwords
    const TileCtx cx{L_conv, win_ch, nullptr, nullptr, nullptr};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 8) L_misc[16 + tid] = 0;              // vote counters (two banks of four)
    if (tid == 0) { L_misc[1] = 0; L_misc[2] = 0; }

    unsigned long long st_reads = 0;
    const uint32_t region = blockIdx.x * (BLOCK / 64) + wave;
    uint32_t *my_rec = sp.rec + (size_t)region * sp.region_cap * sp.rec_stride;
    uint32_t cursor = 0;                             // wave-uniform

    uint32_t t = sp.tile_begin + blockIdx.x;
    uint4 v[CPT];
    uint4 vh = make_uint4(0u, 0u, 0u, 0u);
    const bool has_halo = (uint32_t)tid < halo_ch;
    auto fetch_tile = [&](uint32_t tile) {
        const uint64_t b = (uint64_t)tile * TILE;
        if (b + TILE + p.halo <= p.nbytes) {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.buf + b);
#pragma unroll
            for (int j = 0; j < CPT; j++) v[j] = src[j * BLOCK + tid];
            if (has_halo) vh = src[TILE_CH + tid];
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) v[j] = load_chunk(p, b + (uint64_t)(j * BLOCK + tid) * 16u);
            if (has_halo) vh = load_chunk(p, b + (uint64_t)(TILE_CH + tid) * 16u);
        }
    };
    if (t < sp.tile_end) fetch_tile(t);
    __syncthreads();

    uint32_t parity = 0;
    while (t < sp.tile_end) {
        const uint64_t tbase = (uint64_t)t * TILE;
        const bool inside = tbase + TILE + p.halo <= p.nbytes;

        // ---------------- A: terminator masks + packing (see k_fast)
        {
            uint32_t hiacc = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++) hiacc |= v[j].x | v[j].y | v[j].z | v[j].w;
            uint32_t cr_absent = 0x80808080u;
            bool general = !inside || __any((hiacc & 0x80808080u) != 0);
            if (!general) {
                uint32_t term[CPT];
#pragma unroll
                for (int j = 0; j < CPT; j++) term[j] = nl_mask16_ascii(v[j], cr_absent);
                general = __any((cr_absent & 0x80808080u) != 0x80808080u);
                if (!general) {
#pragma unroll
                    for (int j = 0; j < CPT; j++) {
                        const uint32_t c = j * BLOCK + tid;
                        const uint2 pk = convert_chunk_ascii(v[j]);
                        L_mask[c] = (uint16_t)term[j];
                        L_inv[c] = (uint16_t)pk.y;
                        L_conv[c] = pk;
                    }
                    if (has_halo) L_conv[TILE_CH + tid] = convert_chunk(vh);
                }
            }
            if (general) {
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const uint32_t c = j * BLOCK + tid;
                    const uint64_t g = tbase + (uint64_t)c * 16u;
                    uint32_t nl = eq_mask16(v[j], 0x0A0A0A0Au);
                    uint32_t cr = eq_mask16(v[j], 0x0D0D0D0Du);
                    uint32_t term = nl | (cr & ~(nl >> 1));
                    if (cr & 0x8000u) {
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
        }
        const uint32_t tn = t + gridDim.x;
        if (tn < sp.tile_end) fetch_tile(tn);
        lds_barrier();
        if (tid < 4) L_misc[16 + 4 * (parity ^ 1u) + tid] = 0;

        // ---------------- B: scan
        const bool tile_has_hi = L_misc[1] != 0;
        uint32_t mm[CPT / 2];
        uint32_t ivw[CPT / 2 + 1];
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) mm[i] = reinterpret_cast<const uint32_t *>(L_mask)[tid * (CPT / 2) + i];
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) ivw[i] = reinterpret_cast<const uint32_t *>(L_inv)[tid * (CPT / 2) + i];
        ivw[CPT / 2] = 0;
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < CPT / 2; i++) cnt += __builtin_popcount(mm[i]);
        const uint32_t incl = wave_incl_scan(cnt, lane);
        if (lane == 63) L_misc[4 + wave] = incl;
        // a lane with more than four terminators may hold two wanted lines: such tiles (very short
        // lines), tile 0 and the buffer's last tile are left to the direct pass
        if (cnt > 4u) L_misc[2] = 1;
        const uint32_t span0 = tid * CPT * 16u;

        // ---------------- C: phase vote with the first line of each span (see k_fast)
        bool vote_good = false;
        {
            uint32_t fpos = 0, lo = 0, hi = 0;
            bool found = false;
#pragma unroll
            for (int k = CPT / 2 - 1; k >= 0; k--) {
                if (mm[k]) { fpos = 32u * k + __builtin_ctz(mm[k]); lo = ivw[k]; hi = ivw[k + 1]; found = true; }
            }
            const uint64_t win = (((uint64_t)hi << 32) | lo) >> ((fpos & 31u) + 1u);
            vote_good = found && (win & 0xFFu) == 0 && tbase + span0 + fpos + 9u <= p.nbytes;
        }
        lds_barrier();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { uint32_t x = L_misc[4 + w]; if (w < wave) wbase += x; total += x; }
        const uint32_t excl = wbase + incl - cnt;
        const bool direct = L_misc[2] != 0 || !inside || t == 0;
        uint32_t r0 = 0;
        {
            uint32_t *bank = L_misc + 16 + 4 * parity;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint64_t bl = __ballot(vote_good && (excl & 3u) == (uint32_t)c);
                if (lane == 0 && bl) atomicAdd(&bank[c], (uint32_t)__builtin_popcountll(bl));
            }
            lds_barrier();
            const uint32_t a0 = bank[0], a1 = bank[1], a2 = bank[2], a3 = bank[3];
            uint32_t best = a0;
            if (a1 > best) { best = a1; r0 = 1; }
            if (a2 > best) { best = a2; r0 = 2; }
            if (a3 > best) { best = a3; r0 = 3; }
        }
        if (tid == 0)
            fp.tile_info[t] = total | (r0 << TI_R0_SHIFT) | (tile_has_hi ? TI_HI : 0u) | (direct ? TI_DIRECT : 0u);

        // ---------------- D: one record per wanted line (at most one per lane here)
        if (!direct) {
            const uint32_t lc = (r0 - excl) & 3u;
            const bool has = cnt > lc;                     // cnt <= 4: at most one wanted line
            uint32_t w0 = 0;
            {
                uint32_t r = lc, kbase = 0, m = mm[0];
#pragma unroll
                for (int k = 0; k < CPT / 2 - 1; k++) {
                    const uint32_t c = __builtin_popcount(mm[k]);
                    const bool next = kbase == 32u * k && r >= c;
                    if (next) { r -= c; kbase = 32u * (k + 1); m = mm[k + 1]; }
                }
                const uint32_t m1 = m & (m - 1), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);
                const uint32_t sel = r == 0 ? m : r == 1 ? m1 : r == 2 ? m2 : m3;
                w0 = span0 + kbase + (sel ? __builtin_ctz(sel) : 0u) + 1u;
            }
            uint32_t S[2 * W + 4];
            uint32_t nvalid = 0;
            bool fast = false, slow = false;
            if (has && !(p.dbg & DBG_NO_PHASE2)) {
                st_reads++;
                const uint64_t r = fetch_stream<W, ML_FAST>(p, cx, tbase + w0, w0, false, S, nvalid);
                fast = r == 0;
                slow = !fast;
            }
            const uint64_t fm = __ballot(fast);
            if (fm) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                if (fast) {
                    uint32_t *dst = my_rec + (size_t)(cursor + rank) * sp.rec_stride;
#pragma unroll
                    for (int w = 0; w < 2 * W + 4; w++)
                        if ((uint32_t)w < sp.rec_words) dst[w] = S[w];
                    dst[sp.rec_words] = nvalid;
                }
                cursor += (uint32_t)__builtin_popcountll(fm);
            }
            if (slow) {
                const uint32_t slot = atomicAdd(sp.nslow, 1u);
                if (slot < sp.slow_cap) sp.slow[slot] = tbase + w0;
                else atomicOr(p.stats + ST_ERR, ERR_SPIN);     // cannot happen: one slot per lane per tile
            }
        }

        if (tid == 0) { L_misc[1] = 0; L_misc[2] = 0; }
        parity ^= 1u;
        lds_barrier();
        t = tn;
    }
    if (lane == 0) sp.region_count[region] = cursor;
    const unsigned long long r = wave_sum64(st_reads);
    if (lane == 0 && r) atomicAdd(p.stats + ST_READS, r);
}

// One lane per fast record.  MATCH_SPLIT waves share each region of k_emit (region = global wave id
// / MATCH_SPLIT), so that the grid is large enough for 8 waves per SIMD.
template <int W>
__global__ __launch_bounds__(BLOCK, 8) void k_match(const SParams sp) {
    const KParams &p = sp.f.k;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += BLOCK)
        reinterpret_cast<uint32_t *>(lds)[i] = p.bblob[i];
    __syncthreads();
    const TileCtx cx{nullptr, 0u, reinterpret_cast<const unsigned long long *>(lds),
                     reinterpret_cast<const uint32_t *>(lds + p.off_bmeta),
                     reinterpret_cast<const uint16_t *>(lds + p.off_bdir)};
    const uint32_t gwave = blockIdx.x * (BLOCK / 64) + wave;
    const uint32_t region = gwave / MATCH_SPLIT, part = gwave % MATCH_SPLIT;
    const uint32_t n = sp.region_count[region];
    const uint32_t *my_rec = sp.rec + (size_t)region * sp.region_cap * sp.rec_stride;
    uint32_t st_bar = 0, st_tag = 0;
    for (uint32_t i = part * 64 + lane; i < n; i += 64 * MATCH_SPLIT) {
        const uint32_t *src = my_rec + (size_t)i * sp.rec_stride;
        uint32_t S[2 * W + 4];
#pragma unroll
        for (int w = 0; w < 2 * W + 4; w++) S[w] = (uint32_t)w < sp.rec_words ? src[w] : 0u;
        const uint32_t nvalid = src[sp.rec_words];
        Pending<W> pd;
        uint64_t res = match_stream<W>(p, cx, S, nvalid, pd);
        if (res == R_PEND) res = match_finish<W>(p, pd);
        const uint32_t kind = (uint32_t)(res >> 62);
        if (kind >= 1) st_bar++;
        if (kind == 2) {
            st_tag++;
            if (!(p.dbg & DBG_NO_ATOMIC))
                __hip_atomic_fetch_add(p.counts + (size_t)(res & R_CELL), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long b = wave_sum64(st_bar), g = wave_sum64(st_tag);
    if (lane == 0) {
        if (b) atomicAdd(p.stats + ST_BARCUT, b);
        if (g) atomicAdd(p.stats + ST_TAG, g);
    }
}

// One lane per slow record (rare): the general matcher on raw bytes.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_slow(const SParams sp) {
    const KParams &p = sp.f.k;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += BLOCK)
        reinterpret_cast<uint32_t *>(lds)[i] = p.bblob[i];
    __syncthreads();
    const TileCtx cx{nullptr, 0u, reinterpret_cast<const unsigned long long *>(lds),
                     reinterpret_cast<const uint32_t *>(lds + p.off_bmeta),
                     reinterpret_cast<const uint16_t *>(lds + p.off_bdir)};
    const uint32_t n = min(*sp.nslow, sp.slow_cap);
    uint32_t st_bar = 0, st_tag = 0;
    for (uint32_t i = blockIdx.x * BLOCK + tid; i < n; i += gridDim.x * BLOCK) {
        const uint64_t res = match_line<W, ML_SLOW>(p, cx, sp.slow[i], 0u, true);
        const uint32_t kind = (uint32_t)(res >> 62);
        if (kind >= 1) st_bar++;
        if (kind == 2) {
            st_tag++;
            if (!(p.dbg & DBG_NO_ATOMIC))
                __hip_atomic_fetch_add(p.counts + (size_t)(res & R_CELL), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long b = wave_sum64(st_bar), g = wave_sum64(st_tag);
    if (lane == 0) {
        if (b) atomicAdd(p.stats + ST_BARCUT, b);
        if (g) atomicAdd(p.stats + ST_TAG, g);
    }
}

}  // namespace tdk
