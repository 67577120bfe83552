// k_fast4: the main pass of the free-running path with the workgroup's waves in TWO ROLES and no workgroup barrier
// in the tile loop.  The DEFAULT main pass (td_set_option("kernel", 4)) since the split of the waves between the roles
// follows the input's line density: 10.1-10.9 ms a step at the bench shape against k_fast2's 10.9-11.2 on the same boxes, and
// ahead of it at every read length from 36 to 250 bp (profiles/r04_final/k_fast4_*.txt).  With a fixed 11 + 5
// it had been 5 % behind.
//
// k_fast2 (kernel_fast2.hpp) takes every tile through phases A-B (all four waves: terminators, lists of line starts),
// a barrier, phases C-D (the wanted lines matched by the two waves whose lanes they fill) and a second barrier: per
// tile the four waves can issue 4 x (A-B + D) instructions' worth of time but only have 4 x A-B + 2 x D to issue --
// 72 % at best -- and every wave stands at two barriers behind the slowest.  Here a workgroup is sixteen waves, one
// workgroup per CU, all of the CU's LDS one ring of five 24 KiB tiles:
//
//   producers (7..13 waves: k_f4_estimate below)   stream the FASTQ.  A job is a quarter of a tile (6 KiB); job 4 k + q goes to producer (4 k + q) mod their number;
//                          a producer's next TWO jobs' bytes are in flight in registers.  Per job: raw bytes -> the tile's
//                          slot in LDS, terminator masks, the list of the quarter's line starts (k_fast2's phases A and B,
//                          nothing shared between the producers).  The producer whose arrival makes four ("closer") waits
//                          for the tile before to be closed, adds up the terminator counts, carries the line phase inside a
//                          run or takes the vote at its start, writes the tile's word for k_resolve, and publishes how many
//                          wanted lines (every fourth line start) the tile holds and how many the workgroup's tiles held
//                          before it.
//   consumers (the others)  match.  The wanted lines of the workgroup's tiles form ONE sequence; a consumer claims the next 64
//                          of it (a compare-and-swap on a cursor in LDS) whatever tiles they lie in -- always full lanes,
//                          whatever the read length -- finds each lane's tile among the ring's slots, reads the lines'
//                          pieces into registers, takes the lines off their tiles' counts (the consumer that takes a tile's
//                          last line gives the slot back: long before its pass is through), packs and looks up (line_prepare_q,
//                          kernel_fast2.hpp), finishes the lines it left pending a pass ago (compares, count) and only then
//                          asks for this pass's tag buckets, which stay in flight until the next pass.
//
// Hand-offs are words in LDS.  A wave's LDS operations execute in order, so a wave that drains its stores (s_waitcnt
// lgkmcnt(0)) before it touches a hand-off word has published them.  Every wait is bounded: a wave that waits too long
// raises ERR_SPIN and sets an abort word that ends every loop of the workgroup (the host then reports TD_E_INTERNAL).
// No wait can last: a tile's production needs its slot's previous tile matched, which needs that tile produced -- an
// earlier one; and a consumer that finds fewer than 64 lines takes what there is as soon as the next tile cannot be
// produced before lines are matched (its slot has not been given back), or the workgroup's tiles are through.
// What was measured on the way (200 M reads x 384 x 100 k, k_fast2 11.1-11.5 ms in the same process): five waves (3 + 2,
// 12 KiB tiles, three workgroups a CU asked for) 16.7 ms -- a five-wave workgroup is admitted only TWO to a CU, ten
// waves; eight waves (4 + 4, then 5 + 3 with the quarter-tile jobs) 13.0-13.9 ms -- both roles wait for each other half
// of their time, a ring of three tiles is too short for a matching pass that lasts two tiles' production; sixteen waves
// with seven 16 KiB slots 12.5 ms at 11 + 5 (10 + 6: 12.7, 12 + 4: 13.5, 13 + 3: 16.1); with five 24 KiB slots (fewer, larger
// jobs) 11.6-11.8 ms against 11.1-11.2; consumers without their priority 13.1, 32 KiB tiles 20.3 (registers), short passes
// taken eagerly by idle consumers 11.6-11.7; the split by read length (36 bp: 8 + 8 ... 250 bp: 12 + 4): ahead of k_fast2
// everywhere -- the consumers never wait, the producers wait for slots: how many consumers a tile's lines need is the input's.
// Tiles that are not "regular" (the buffer's first and last, bytes >= 0x80, '\r' at the end of a chunk next to another
// wave's bytes, more line starts than a list holds) are only counted here (terminators) and flagged TI_SKIP for the
// fix-up pass (k_fast<6, W, true>: the same 24 KiB tile), as in k_fast2.
#pragma once
#include "kernel_fast2.hpp"

namespace tdk {

#ifndef TD_F4_WAVES
#define TD_F4_WAVES 16              // waves of a workgroup (a multiple of four: the waves go to the four SIMDs in turn)
#endif
#ifndef TD_F4_NPROD
#define TD_F4_NPROD 11              // producer waves among them where no estimate is given (they take the quarter-tile jobs in turn); the others match
#endif
#ifndef TD_F4_SLOTS
#define TD_F4_SLOTS 5               // tiles the ring in LDS holds
#endif
constexpr int F4_PROD = 4 /* quarters of a tile */, F4_WAVES = TD_F4_WAVES, F4_BLOCK = 64 * F4_WAVES;
#ifndef TD_F4_CPT
#define TD_F4_CPT 6                 // 16-byte chunks per producer lane and job (4: 16 KiB tiles; 6: 24 KiB)
#endif
constexpr int F4_CPT = TD_F4_CPT;                           // 16-byte chunks per producer lane and tile
constexpr uint32_t F4_WCH = F4_CPT * 64;                    // chunks per producer and tile
constexpr uint32_t F4_WBYTES = F4_WCH * 16;
constexpr uint32_t F4_TILE = F4_PROD * F4_WBYTES;           // 24 KiB
constexpr int F4_SLOTS = TD_F4_SLOTS;
constexpr uint32_t F4_SPIN_LIMIT = 1u << 18;
#ifndef TD_F4_EAGER_MIN
#define TD_F4_EAGER_MIN 65          // a consumer that has waited TD_F4_EAGER_SPINS polls takes fewer than 64 lines if there are this many (65: never)
#endif
#ifndef TD_F4_EAGER_SPINS
#define TD_F4_EAGER_SPINS 4
#endif
#ifndef TD_F4_P_CONS
#define TD_F4_P_CONS 1              // wave priority of the consumers (the producers run at 0)
#endif

// hand-off words of a slot (dwords in LDS; F4_R0..F4_WB3 are read as one 16-byte word, F4_NCUM + F4_NWANT as one 8-byte word)
enum { F4_DONE = 0, F4_FREE = 1, F4_REMAIN = 2, F4_FLAGS = 3, F4_TOT = 4 /* 4 */, F4_VOTE = 8 /* 4 */, F4_R0 = 12, F4_WB1 = 13, F4_WB2 = 14,
       F4_WB3 = 15, F4_TOTAL = 16, F4_SEQ = 17, F4_NCUM = 18, F4_NWANT = 19, F4_TIDX = 20, F4_CTRL_DW = 24 };
constexpr uint32_t F4_FLAG_HI = 1, F4_FLAG_OVER = 2, F4_FLAG_HALO_HI = 4;
// the workgroup's words behind the slots': abort, tiles closed, wanted lines in them, wanted lines claimed
constexpr uint32_t F4_ABORT_DW = F4_SLOTS * F4_CTRL_DW, F4_READY_DW = F4_ABORT_DW + 1, F4_AVAIL_DW = F4_ABORT_DW + 2, F4_CLAIMED_DW = F4_ABORT_DW + 3;
constexpr uint32_t F4_NPROD_MIN = 7, F4_NPROD_MAX = 13;      // producers of the sixteen waves (LDS holds hot-cell caches for 16 - F4_NPROD_MIN consumers)
constexpr uint32_t F4_CTRL_BYTES = 1024;                     // (256 dwords: the slots' words, the workgroup's, a spare word per wave at 240)
static_assert(F4_CLAIMED_DW < 240 && F4_WAVES <= 16, "hand-off words");

__device__ __forceinline__ uint32_t lds_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// waits until *p >= need (a word only ever grows); false: gave up -- the abort word is set and ERR_SPIN raised
__device__ __forceinline__ bool f4_wait(uint32_t *p, uint32_t need, uint32_t *abort_word, unsigned long long *stats) {
    for (uint32_t spins = 0;; spins++) {
        const uint32_t v = lds_ld(p);
        if (v >= need) break;
        if (spins > F4_SPIN_LIMIT || lds_ld(abort_word)) {
            lds_st(abort_word, 1u);
            if ((threadIdx.x & 63) == 0) atomicOr(stats + ST_ERR, ERR_SPIN);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (nothing behind the wait is read before it)
    return true;
}

// The producer count for k_fast4 from the buffer's first bytes: bytes per record = 4 x bytes per line over the first 64 KiB
// (one workgroup, launched in front of the main pass on its stream: no host round trip).
#ifndef TD_INST_ONLY
__global__ __launch_bounds__(256) void k_f4_estimate(const uint8_t *buf, uint64_t nbytes, uint32_t *out) {
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const uint32_t n = (uint32_t)(nbytes < 65536ull ? nbytes : 65536ull) & ~15u;
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x * 16u; i < n; i += 256u * 16u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(buf + i);
        cnt += (uint32_t)__builtin_popcount(eq_mask16(v, 0x0A0A0A0Au) | eq_mask16(v, 0x0D0D0D0Du));    // ("\r\n" counts twice: such records read as shorter, a consumer more)
    }
    atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t rec = total >= 8u ? 4u * n / total : 219u;      // bytes per record
        *out = rec < 150u ? 8u : rec < 195u ? 9u : rec < 270u ? 10u : rec < 420u ? 11u : 12u;
    }
}
#endif

// diagnostic build only (-DTD_PHASE_PROF): lane 0 of every wave adds the shader-clock cycles between stamps to stats[8 + i]
// (producers 0-4: waiting for the slot, A, B + lists, closing a tile, -; consumers 5-8: pending lines, waiting for a tile, matching, rest)
#ifdef TD_PHASE_PROF
#define F4_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); f4acc[i] += now_ - f4last; f4last = now_; } while (0)
#else
#define F4_STAMP(i) do {} while (0)
#endif

// PROG: the progress windows' per-tile sums (kernel_fast.hpp FParams::tile_sums: how many of a tile's wanted lines had a barcode,
// low half, and a tag, high half) -- a consumer's lines lie in up to three tiles, so it adds per tile: the barcodes when the lines
// are packed, the tags a pass later, when the pending lines are settled (each lane remembers its line's tile).
template <int W, int NQ, bool PROG>
__global__ __launch_bounds__(F4_BLOCK, F4_WAVES / 4) void k_fast4(const FParams fp) {
    const KParams &p = fp.k;
#ifdef TD_PHASE_PROF
    unsigned long long f4acc[12] = {};
    unsigned long long f4last = __builtin_amdgcn_s_memtime();
#endif
    constexpr uint32_t TILE = F4_TILE, WCH = F4_WCH, WBYTES = F4_WBYTES;
    constexpr int CPT = F4_CPT;
    static_assert(W <= 3, "k_fast4 has the pipelined probe of the 64-byte buckets only");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t halo = p.halo;                                            // bytes staged behind a tile (multiple of 64, >= 16 * NQ + 16)
    const uint32_t slot_bytes = TILE + halo;
    uint8_t *L_raw0 = lds;
    uint16_t *L_list0 = reinterpret_cast<uint16_t *>(lds + F4_SLOTS * slot_bytes);          // [slot][producer][WCH]: masks, then line starts
    uint32_t *L_ctrl = reinterpret_cast<uint32_t *>(lds + F4_SLOTS * slot_bytes + F4_SLOTS * F4_PROD * WCH * 2u);
    uint32_t *L_abort = L_ctrl + F4_ABORT_DW;
    uint8_t *L_hc = reinterpret_cast<uint8_t *>(L_ctrl) + F4_CTRL_BYTES;
    uint8_t *L_bidx = L_hc + ((uint32_t)F4_WAVES - F4_NPROD_MIN) * HC_BYTES_PER_WAVE;
    TileCtx cx{nullptr, 0u, reinterpret_cast<const unsigned long long *>(L_bidx),
               reinterpret_cast<const uint32_t *>(L_bidx + p.off_bmeta),
               reinterpret_cast<const uint16_t *>(L_bidx + p.off_bdir)};

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < p.bblob_bytes / 4; i += F4_BLOCK)
        reinterpret_cast<uint32_t *>(L_bidx)[i] = p.bblob[i];
    if (tid < (int)(F4_CTRL_BYTES / 4)) L_ctrl[tid] = 0;
    __syncthreads();

    const unsigned long long carried = p.cursor_in ? *p.cursor_in : 0ull;
    const uint64_t first_line = p.first_line + carried;
    const uint32_t nwork = p.ntiles;
    const uint32_t RUN = p.run ? p.run : 1u;
    auto tile_base = [&](uint32_t tile) -> const uint8_t * {
        return tile >= p.tail_tile ? p.tail_buf + (uint64_t)(tile - p.tail_tile) * TILE : p.buf + (uint64_t)tile * TILE;
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto tile_rsrc = [&](uint32_t tile) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(tile_base(tile)), 0, (int)(TILE + 4096u), 0x00020000);
    };

    // ---------------------------------------------------------------- producing: a job is a quarter of a tile; job 4 k + q goes
    // to producer (4 k + q) mod NPROD
    // How many of the sixteen waves produce follows the input: short reads put more lines into a tile and want more
    // consumers, long reads fewer (measured, 96 barcodes x 10 k tags: 36 bp best at 8 producers, 75 bp at 9, 100 bp at 10, 150 bp
    // at 11, 250 bp at 12 -- each better than k_fast2 there, the fixed 11 + 5 only from 100 bp up).  k_f4_estimate, launched in
    // front of this kernel, leaves the number in device memory (KParams::f4_nprod_dev); td_set_option "f4_nprod" fixes it.
    uint32_t np_ = p.f4_nprod ? p.f4_nprod : (p.f4_nprod_dev ? *p.f4_nprod_dev : (uint32_t)TD_F4_NPROD);
    np_ = np_ < F4_NPROD_MIN ? F4_NPROD_MIN : np_ > F4_NPROD_MAX ? F4_NPROD_MAX : np_;
    const uint32_t NPROD = (uint32_t)__builtin_amdgcn_readfirstlane((int)np_);
    {
        // a job's bytes in registers: this lane's CPT chunks, and -- the last quarter's first lanes -- the halo behind the
        // tile (every lane asks: the others for an offset beyond the descriptor's range, which returns zeros without touching
        // memory, so that the loads are unconditional straight-line code the wait counts can leave in flight).  A producer
        // has its next TWO jobs' bytes in flight.
        uint4 va[CPT], vb[CPT];
        u32x4 vha, vhb;
        auto fetch_job = [&](uint4 (&v)[CPT], u32x4 &vh, uint32_t tile, uint32_t q) {
            const uint32_t voff = q * WBYTES + (uint32_t)lane * 16u;               // this lane's first chunk; chunk j is 1 KiB further
            const bool hh = q == (uint32_t)F4_PROD - 1u && (uint32_t)lane < halo / 16u;   // (the halo is the last quarter's)
            const uint32_t hoff = hh ? (uint32_t)lane * 16u : 0x40000000u;         // (beyond every descriptor's range: reads as zero)
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile);
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const u32x4 qq = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * 1024, 2 /* nt */);
                v[j] = make_uint4(qq.x, qq.y, qq.z, qq.w);
            }
            vh = __builtin_amdgcn_raw_buffer_load_b128(rs, hoff, (int)TILE, 0);
        };
        // the workgroup's tiles in the order it takes them: runs of RUN consecutive tiles, run r of the buffer to workgroup r mod grid
        auto tile_of = [&](uint32_t k) -> uint32_t { return (k / RUN) * (gridDim.x * RUN) + blockIdx.x * RUN + k % RUN; };
        uint32_t n_local = 0;
        for (uint32_t first = blockIdx.x * RUN; first < nwork; first += gridDim.x * RUN) n_local += min(RUN, nwork - first);
        // this producer's next job (its bytes in flight)
        uint32_t my_job = (uint32_t)wave;
        if (wave < (int)NPROD) {
            if (my_job < 4u * n_local) fetch_job(va, vha, tile_of(my_job >> 2), my_job & 3u);
            if (my_job + NPROD < 4u * n_local) fetch_job(vb, vhb, tile_of((my_job + NPROD) >> 2), (my_job + NPROD) & 3u);
        }
        uint32_t *g_ready = L_ctrl + F4_READY_DW, *g_avail = L_ctrl + F4_AVAIL_DW, *g_claimed = L_ctrl + F4_CLAIMED_DW;

        // one tile from the registers v (its bytes were requested two tiles ago); afterwards v holds the tile after next
        if (wave < (int)NPROD) {
        // quarter qtr of tile number k of the workgroup (tile t of the buffer) from the registers v -- its bytes were requested
        // a job ago; afterwards v holds the bytes of the producer's next job
        auto produce = [&](uint4 (&v)[CPT], u32x4 &vh, const uint32_t k, const uint32_t t) -> bool {
            const uint32_t qtr = my_job & 3u;
            const uint32_t voff = qtr * WBYTES + (uint32_t)lane * 16u;
            const bool has_halo = qtr == (uint32_t)F4_PROD - 1u && (uint32_t)lane < halo / 16u;
            const uint32_t slot = k % (uint32_t)F4_SLOTS, use = k / (uint32_t)F4_SLOTS;
            uint8_t *L_raw = L_raw0 + slot * slot_bytes;
            uint16_t *Ll = L_list0 + (slot * F4_PROD + qtr) * WCH;
            uint32_t *ctrl = L_ctrl + slot * F4_CTRL_DW;
            const uint64_t tbase = (uint64_t)t * TILE;
            const bool carry_ok = k % RUN != 0u && t != 1u;                       // the tile continues its predecessor's run: the phase is carried
            const uint32_t job2 = my_job + 2u * NPROD;                            // (the job whose bytes take these registers)
            const bool more = job2 < 4u * n_local;
            const uint32_t t2 = more ? tile_of(job2 >> 2) : 0u;
            F4_STAMP(4);
            // the slot must have been given back as often as it has been used
            if (!f4_wait(ctrl + F4_FREE, use, L_abort, p.stats)) return false;
            F4_STAMP(0);
            // ---------------- A: raw bytes and terminator masks -> LDS
            bool wave_crb = false;
            uint32_t myflags = 0;
            {
                uint32_t hiacc = 0;
#pragma unroll
                for (int j = 0; j < CPT; j++) hiacc |= v[j].x | v[j].y | v[j].z | v[j].w;
#pragma unroll
                for (int j = 0; j < CPT; j++) *reinterpret_cast<uint4 *>(L_raw + voff + j * 1024) = v[j];
                uint16_t *Lm = Ll + lane;
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
                            Lm[j * 64] = (uint16_t)(nl | (cr & ~(nl >> 1)));      // (a '\r' in the chunk's last byte: settled below)
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
                    myflags |= F4_FLAG_HI;                          // (wave-uniform: some lane of this wave holds a byte >= 0x80)
                }
            }
            // the halo, then the tile after next: its bytes take this tile's registers
            if (has_halo) *reinterpret_cast<uint4 *>(L_raw + TILE + (uint32_t)lane * 16u) = make_uint4(vh.x, vh.y, vh.z, vh.w);
            // (lines that begin in this tile are packed from these bytes with the ASCII forms)
            if (qtr == (uint32_t)F4_PROD - 1u && __any(has_halo && ((vh.x | vh.y | vh.z | vh.w) & 0x80808080u) != 0)) myflags |= F4_FLAG_HALO_HI;
            const uint32_t my_qtr = qtr;
            if (more) fetch_job(v, vh, t2, job2 & 3u);
            wave_lds_fence();          // this wave's masks and raw bytes are in LDS
            F4_STAMP(1);

            // ---------------- B: terminators of this lane's CPT consecutive chunks, wave scan, vote, list of line starts
            uint32_t mm[CPT / 2];
#pragma unroll
            for (int i = 0; i < CPT / 2; i++) mm[i] = reinterpret_cast<const uint32_t *>(Ll)[lane * (CPT / 2) + i];
            const uint32_t span0 = my_qtr * WBYTES + (uint32_t)lane * (CPT * 16u);
            const uint32_t wend = (my_qtr + 1u) * WBYTES;                          // end of this wave's quarter
            if (__builtin_expect(wave_crb, 0)) {
                // a chunk whose last byte is '\r' (bit 15 of its mask is set for it -- or for a '\n' there): one terminator
                // with the '\n' that opens the next chunk, if there is one
#pragma unroll
                for (int i = 0; i < CPT / 2; i++) {
#pragma unroll
                    for (int hbit = 15; hbit < 32; hbit += 16) {
                        if ((mm[i] >> hbit) & 1u) {
                            const uint32_t at = span0 + 32u * i + (uint32_t)hbit;
                            if (L_raw[at] == 0x0Du) {
                                // (the next byte may belong to another wave's third, or to the halo: not in LDS yet)
                                const uint32_t nx = at + 1u < wend ? (uint32_t)L_raw[at + 1u] : (uint32_t)tile_base(t)[at + 1u];
                                if (nx == 0x0Au) mm[i] &= ~(1u << hbit);
                            }
                        }
                    }
                }
            }
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < CPT / 2; i++) cnt += __builtin_popcount(mm[i]);
            const uint32_t incl = wave_incl_scan(cnt, lane);
            const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t packed = 0;
            if (!carry_ok) {
                // the first line that starts in this lane's span: do its first eight bytes (inside this wave's third) read as bases?
                uint32_t fpos = 0;
                bool found = false;
#pragma unroll
                for (int q = CPT / 2 - 1; q >= 0; q--) {
                    if (mm[q]) { fpos = 32u * q + __builtin_ctz(mm[q]); found = true; }
                }
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
            // the lines behind this wave's terminators, by wave-local ordinal, into the space of its masks (every lane holds its
            // masks in registers by now: the fence below the read-back is the wave's own order of LDS operations)
            if (__builtin_expect(wtot <= WCH, 1)) {
                uint16_t *spare = reinterpret_cast<uint16_t *>(L_ctrl + 240 + (uint32_t)wave);     // (a word nobody reads)
                uint32_t kk = incl - cnt, rest = 0;
#pragma unroll
                for (int i = 0; i < CPT / 2; i++) {
                    const uint32_t m = mm[i], m1 = m & (m - 1u);
                    const uint32_t base = span0 + 32u * i + 1u;
                    uint16_t *d0 = m ? Ll + kk : spare;
                    *d0 = (uint16_t)(base + (uint32_t)__builtin_ctz(m | 0x80000000u));
                    kk += m ? 1u : 0u;
                    uint16_t *d1 = m1 ? Ll + kk : spare;
                    *d1 = (uint16_t)(base + (uint32_t)__builtin_ctz(m1 | 0x80000000u));
                    kk += m1 ? 1u : 0u;
                    rest |= m1 & (m1 - 1u);
                }
                if (__builtin_expect(__any(rest != 0), 0)) {
                    kk = incl - cnt;
#pragma unroll
                    for (int i = 0; i < CPT / 2; i++) {
                        uint32_t m = mm[i];
                        while (m) {
                            const uint32_t bit = __builtin_ctz(m);
                            m &= m - 1;
                            Ll[kk] = (uint16_t)(span0 + 32u * i + bit + 1u);
                            kk++;
                        }
                    }
                }
            } else {
                myflags |= F4_FLAG_OVER;
            }
            if (lane == 0) {
                lds_st(ctrl + F4_TOT + my_qtr, wtot);
                lds_st(ctrl + F4_VOTE + my_qtr, packed);
                if (myflags) __hip_atomic_fetch_or(ctrl + F4_FLAGS, myflags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_lds_fence();          // raw bytes, halo, lists, totals: stored
            F4_STAMP(2);
            // ---------------- the producer that arrives last closes the tile
            uint32_t arrived = 0;
            if (lane == 0) arrived = __hip_atomic_fetch_add(ctrl + F4_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
            arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
            if (arrived == (uint32_t)F4_PROD * (use + 1u)) {
                // (tiles are closed in their order: the one before this is produced by the workgroup's other four waves)
                if (!f4_wait(g_ready, k, L_abort, p.stats)) return false;
                const uint32_t t_x = lds_ld(ctrl + F4_TOT + 0), t_y = lds_ld(ctrl + F4_TOT + 1), t_z = lds_ld(ctrl + F4_TOT + 2), t_w = lds_ld(ctrl + F4_TOT + 3);
                const uint32_t flags = lds_ld(ctrl + F4_FLAGS);
                const uint32_t wb1 = t_x, wb2 = wb1 + t_y, wb3 = wb2 + t_z, total = wb3 + t_w;
                // (the tile before this one is closed, and its slot is not closed again before this tile has been matched)
                const uint32_t *prev = L_ctrl + ((k + (uint32_t)F4_SLOTS - 1u) % (uint32_t)F4_SLOTS) * F4_CTRL_DW;
                uint32_t r0;
                if (carry_ok) {
                    r0 = (lds_ld(prev + F4_R0) - lds_ld(prev + F4_TOTAL)) & 3u;
                } else if (t != 0) {
                    auto rot = [](uint32_t pk, uint32_t by) { const uint32_t r = 8u * (by & 3u); return r ? ((pk << r) | (pk >> (32u - r))) : pk; };
                    const uint32_t a = lds_ld(ctrl + F4_VOTE + 0), b = rot(lds_ld(ctrl + F4_VOTE + 1), wb1), c = rot(lds_ld(ctrl + F4_VOTE + 2), wb2),
                                   d = rot(lds_ld(ctrl + F4_VOTE + 3), wb3);
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
                const bool regular = t != 0 && flags == 0 && tbase + TILE + halo <= p.nbytes && !(TD_DBG(p) & DBG_NO_PHASE2);
                const uint32_t nwant = regular ? (total + 3u - r0) >> 2 : 0u;
                const uint32_t ncum = k ? lds_ld(prev + F4_NCUM) + lds_ld(prev + F4_NWANT) : 0u;      // the wanted lines of the workgroup's tiles before this one
                if (lane == 0) {
                    fp.tile_info[t] = total | (r0 << TI_R0_SHIFT) | ((flags & F4_FLAG_HI) ? TI_HI : 0u) | (regular ? 0u : TI_SKIP);
                    lds_st(ctrl + F4_R0, r0); lds_st(ctrl + F4_WB1, wb1); lds_st(ctrl + F4_WB2, wb2); lds_st(ctrl + F4_WB3, wb3);
                    lds_st(ctrl + F4_TOTAL, total); lds_st(ctrl + F4_SEQ, k); lds_st(ctrl + F4_TIDX, t);
                    lds_st(ctrl + F4_REMAIN, nwant);
                    lds_st(ctrl + F4_FLAGS, 0u);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // (one 8-byte store: a consumer never sees the new tile's line count beside the old tile's first line)
                if (lane == 0)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(ctrl + F4_NCUM), (unsigned long long)ncum | ((unsigned long long)nwant << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) {
                    lds_st(g_avail, ncum + nwant);
                    if (nwant == 0u) lds_st(ctrl + F4_FREE, use + 1u);          // (nothing to match in it: the slot goes straight back)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) lds_st(g_ready, k + 1u);
                F4_STAMP(3);
            }
            return true;
        };
            // ------------------------------------------------------------ a producer: its jobs in turn
            // (two jobs per turn of the loop: the registers a job's bytes come from are named at compile time)
            for (;;) {
                if (!(my_job < 4u * n_local)) break;
                if (!produce(va, vha, my_job >> 2, tile_of(my_job >> 2))) break;
                my_job += NPROD;
                if (!(my_job < 4u * n_local)) break;
                if (!produce(vb, vhb, my_job >> 2, tile_of(my_job >> 2))) break;
                my_job += NPROD;
            }
        } else {
        // ---- the state of the wave's matching (every wave matches whenever it cannot produce): the lines of its last pass,
        // whose tag buckets are in flight
        const int cons = wave < (int)NPROD ? 0 : wave - (int)NPROD;
        uint2 *hcS = reinterpret_cast<uint2 *>(L_hc + cons * HC_BYTES_PER_WAVE);
        uint8_t *hcE = L_hc + cons * HC_BYTES_PER_WAVE + HC_SLOTS * 8;
        if (wave >= (int)NPROD) {
#pragma unroll
            for (int q = 0; q < HC_SLOTS / 64; q++) hcS[q * 64 + lane] = make_uint2(HC_EMPTY, 0u);
            wave_lds_fence();
        }
        bool hc_on = p.hot_cache != 0;
        uint32_t hc_hits = 0, hc_rest = 0, aged = 0;
        int st_reads = 0, st_bar = 0, st_tag = 0;
        Pending<W> pd;                                          // lines whose buckets are in flight
        Pending<W> nx;                                          // a line whose bucket has not been asked for yet (R, nr, boff only)
        bool pd_valid = false;
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
        // the pending lines of the pass before: compares and count
        uint32_t pd_tile = 0;                                   // (PROG) the tile of this lane's pending line
        // one add per tile the flagged lanes' lines lie in (at most the ring's slots)
        auto sums_add = [&](bool flag, uint32_t tile, uint32_t unit) {
            uint64_t left = __ballot(flag);
            while (left) {
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tile, (int)__builtin_ctzll(left));
                const uint64_t m = __ballot(flag && tile == t0);
                if (lane == (int)__builtin_ctzll(left)) atomicAdd(fp.tile_sums + t0, unit * (uint32_t)__builtin_popcountll(m));
                left &= ~m;
            }
        };
        auto settle = [&]() {
            if (__any(pd_valid)) {
                bool phit = false; uint32_t pcell = 0;
                phit_tag = false;
                if (pd_valid) finish_pending(phit, pcell);
                if (PROG) sums_add(pd_valid && phit_tag, pd_tile, 1u << 16);
                pd_valid = false;
                if (hc_on) {
                    const uint32_t h = hc_hash(pcell);
                    hc_hits += (uint32_t)__builtin_popcountll(__ballot(phit && hcS[h].x == pcell));
                }
                hc_commit(p.counts, hcS, hcE, phit, pcell, (uint32_t)lane, hc_on);
            }
        };

        // ---------------------------------------------------------------- matching
        // 64 wanted lines of the workgroup's sequence from number cur on (n of them: the last lanes idle)
        auto match = [&](const uint32_t cur, const uint32_t n) {
        // ---------------- each lane's line: its tile is the slot whose lines [ncum, ncum + nwant) hold the line's number
        const uint32_t g = cur + (uint32_t)lane;
        const bool active = (uint32_t)lane < n;
        uint32_t myslot = F4_SLOTS, jl = 0;
        uint32_t seq_of[F4_SLOTS];
#pragma unroll
        for (int sl = 0; sl < F4_SLOTS; sl++) {
            const unsigned long long cw = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(L_ctrl + sl * F4_CTRL_DW + F4_NCUM),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            seq_of[sl] = lds_ld(L_ctrl + sl * F4_CTRL_DW + F4_SEQ);
            const uint32_t nc = (uint32_t)cw, nw = (uint32_t)(cw >> 32);
            if (active && g - nc < nw) { myslot = (uint32_t)sl; jl = g - nc; }
        }
        uint32_t kres = 7u, srel = 0;                        // (7: no line for this lane)
        uint4 q[NQ];
        const bool mine = active && myslot < (uint32_t)F4_SLOTS;
        uint32_t my_tile = 0;                                // the line's tile (read now: the slot's words are the next tile's once it is given back)
        if (mine) {
            const uint32_t *cs = L_ctrl + myslot * F4_CTRL_DW;
            my_tile = lds_ld(cs + F4_TIDX);
            const uint4 ph = *reinterpret_cast<const uint4 *>(cs + F4_R0);        // r0, wb1, wb2, wb3
            // (the producer whose list holds ordinal o, and o's place in it: selects, no branches)
            const uint32_t o = ph.x + 4u * jl;
            uint32_t sel = 0u;
            sel = o >= ph.y ? 1u * WCH - ph.y : sel;
            sel = o >= ph.z ? 2u * WCH - ph.z : sel;
            sel = o >= ph.w ? 3u * WCH - ph.w : sel;
            srel = (L_list0 + myslot * F4_PROD * WCH)[o + sel];
            // (a line that starts in the tile's last bytes is still whole in the staged window: the halo holds 16 NQ bytes and more)
            line_read<NQ>(L_raw0 + myslot * slot_bytes, srel, q);
        }
        // the lines' bytes are in registers: off their tiles' counts -- who takes a tile's last line gives its slot back, long
        // before the pass is through (packing, the barcode walk and the pending lines take ten times as long as the reads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int sl = 0; sl < F4_SLOTS; sl++) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(myslot == (uint32_t)sl));
            if (cnt && lane == 0) {
                uint32_t *cs = L_ctrl + sl * F4_CTRL_DW;
                const uint32_t before = __hip_atomic_fetch_sub(cs + F4_REMAIN, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (before == cnt) lds_st(cs + F4_FREE, seq_of[sl] / (uint32_t)F4_SLOTS + 1u);
            }
        }
        if (mine) kres = line_prepare_q<W, NQ, false>(p, cx, q, nx);
        // kres: 0 no barcode, 2 barcode only, 1 the tag is to be looked up, 6 leading blank (rare: raw bytes re-read)
        st_reads += kres == 0u || kres == 2u ? 1 : 0;
        st_bar += kres == 2u ? 1 : 0;
        F4_STAMP(7);
        // the lines that have been pending since the pass before: their buckets were asked for a whole line_prepare ago -- then
        // this pass's buckets, which stay in flight
        settle();
        F4_STAMP(5);
        // (behind the wait for the pending lines' buckets, not in front of it: the adds' round trips run under the next pass)
        if (PROG) sums_add(kres == 1u || kres == 2u, my_tile, 1u);
        // (rare: a line that opens with a blank -- str.strip, reference :256 -- is matched from its raw bytes in global memory;
        // here, where no bucket is in flight and the pending registers are free)
        if (__builtin_expect(__any(kres == 6u), 0)) {
            if (kres == 6u) {
                const uint32_t tix = my_tile;
                const uint64_t tb = (uint64_t)tix * TILE;
                const uint64_t res = match_line<W, ML_SLOW>(p, cx, tb + srel, srel, true);
                const uint32_t kind = (uint32_t)(res >> 62);
                st_reads += 1;
                if (kind >= 1) st_bar += 1;
                if (PROG && kind >= 1) atomicAdd(fp.tile_sums + tix, 1u + (kind == 2 ? 1u << 16 : 0u));
                if (kind == 2) {
                    st_tag += 1;
                    if (!(TD_DBG(p) & DBG_NO_ATOMIC))
                        __hip_atomic_fetch_add(p.counts + (size_t)(res & R_CELL), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            vm_settled();
        }
        if (kres == 1u) {
#pragma unroll
            for (int w = 0; w < W; w++) pd.R[w] = nx.R[w];
            pd.nr = nx.nr; pd.boff = nx.boff;
            bucket_issue<W>(p, pd);
            pd_valid = true;
            if (PROG) pd_tile = my_tile;
        }
        if (p.hot_cache && ++aged == HC_AGE_TILES) {
            aged = 0;
            if (hc_on) {
                hc_flush(p.counts, hcS, (uint32_t)lane);
                // a consumer commits ~45 hits a pass: below ~3 % of them cached, the cache is only overhead
                if (p.hot_cache != 2u && hc_hits * 32u < HC_AGE_TILES * 45u) { hc_on = false; hc_rest = HC_REST; }
                hc_hits = 0;
            } else if (--hc_rest == 0) hc_on = true;
        }
        };

            // ------------------------------------------------------------ a consumer: 64 lines at a time
            __builtin_amdgcn_s_setprio(TD_F4_P_CONS);
            for (uint32_t idle = 0;;) {
                F4_STAMP(8);
                // claim the next wanted lines of the workgroup's sequence: 64, or what there is when no more can come
                const uint32_t cur = lds_ld(g_claimed);
                const uint32_t R = lds_ld(g_ready);
                const uint32_t A = lds_ld(g_avail);              // (read last: it counts every line claimed before `cur` was read)
                const uint32_t have = A - cur;
                uint32_t n = 0;
                if (have >= 64u) n = 64u;
                else if (R == n_local) {
                    if (have == 0u) break;                          // the workgroup's tiles are through and matched
                    n = have;
                } else if (have != 0u && lds_ld(L_ctrl + (R % (uint32_t)F4_SLOTS) * F4_CTRL_DW + F4_FREE) < R / (uint32_t)F4_SLOTS) {
                    n = have;                                    // (the next tile's slot has not been given back: these lines first)
                } else if (have >= (uint32_t)TD_F4_EAGER_MIN && idle >= (uint32_t)TD_F4_EAGER_SPINS) {
                    n = have;                                    // (nothing else to do: a short pass now gives slots back sooner)
                }
                if (n) {
                    uint32_t won = 0;
                    if (lane == 0) {
                        uint32_t expected = cur;
                        won = __hip_atomic_compare_exchange_strong(g_claimed, &expected, cur + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                    }
                    if (__builtin_amdgcn_readfirstlane((int)won)) {
                        F4_STAMP(6);
                        match(cur, n);
                        idle = 0;
                    }
                    continue;                                    // (or another consumer took them: look again)
                }
                if (++idle > F4_SPIN_LIMIT || lds_ld(L_abort)) {
                    lds_st(L_abort, 1u);
                    if (lane == 0) atomicOr(p.stats + ST_ERR, ERR_SPIN);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            settle();
            if (p.hot_cache) hc_flush(p.counts, hcS, (uint32_t)lane);
            unsigned long long r = wave_sum64((unsigned long long)(long long)st_reads), b = wave_sum64((unsigned long long)(long long)st_bar),
                               g = wave_sum64((unsigned long long)(long long)st_tag);
            if (lane == 0) {
                if (r) atomicAdd(p.stats + ST_READS, r);
                if (b) atomicAdd(p.stats + ST_BARCUT, b);
                if (g) atomicAdd(p.stats + ST_TAG, g);
            }
        }
#ifdef TD_PHASE_PROF
        if (lane == 0) for (int i = 0; i < 9; i++) atomicAdd(p.stats + 8 + i, f4acc[i]);
#endif
    }
}

}  // namespace tdk

#ifdef TD_FAST4_EXTERN
#define TD_X4(W, NQ) extern template __global__ void tdk::k_fast4<W, NQ, false>(const tdk::FParams); extern template __global__ void tdk::k_fast4<W, NQ, true>(const tdk::FParams);
TD_FAST2_COMBOS(TD_X4)
#undef TD_X4
#endif
