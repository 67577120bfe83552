// Host side of libtagdig: handle, index builder, launches, streaming, C-ABI.
// See include/tagdig.h for the contract and kernels.hpp for the device code.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/tagdig.h"
#include "../../include/td_synth_spec.h"
#include "kernels.hpp"
#include "kernel_fast.hpp"
#define TD_FAST2_EXTERN 1           // the instantiations of k_fast2 live in inst_fast2.hip (six translation units)
#include "kernel_fast2.hpp"
#define TD_FAST4_EXTERN 1           // ... those of k_fast4 in inst_fast4.hip
#include "kernel_fast4.hpp"
#include "kernel_splitter.hpp"
#include "kernel_splitter2.hpp"
#include "gz_source.hpp"
#include "gz_pyrules.hpp"
#include "gpu_inflate.hpp"
#include "gz_resolve.hpp"
#include "gz_gpu.hpp"

namespace {

thread_local std::string g_err;
thread_local uint32_t g_bad = 0;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(TD_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

// base codes must match the device's (byte >> 1) & 3:  A 0, C 1, T 2, G 3
inline int base_code(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'T': return 2; case 'G': return 3; default: return -1; }
}
// first base in the top bits of word 0
void pack_bases(const std::string &s, uint64_t *words, int nwords) {
    for (int w = 0; w < nwords; w++) words[w] = 0;
    for (size_t i = 0; i < s.size(); i++)
        words[i >> 5] |= (uint64_t)base_code(s[i]) << (62 - 2 * (i & 31));
}
inline uint32_t hash_key(uint64_t key) {   // must match tdk::hash_key (24-bit multiplies: low 32 bits of the product of the low 24 bits)
    auto mul24 = [](uint32_t x, uint32_t y) -> uint32_t { return (uint32_t)((uint64_t)(x & 0xFFFFFFu) * (uint64_t)(y & 0xFFFFFFu)); };
    const uint32_t a = (uint32_t)key & 0x3FFFFFu, b = (uint32_t)(key >> 22) & 0x1FFFFFu, c = (uint32_t)(key >> 43);
    uint32_t h = mul24(a, 0x9E3779u) ^ (mul24(b, 0x85EBCBu) + 0x7F4A7C15u);
    h ^= mul24(c, 0xC2B2AFu) << 3;
    h ^= h >> 15;
    h = mul24(h & 0xFFFFFFu, 0x2C1B3Du) ^ (h >> 9);
    h ^= h >> 13;
    return h;
}

// ---------------------------------------------------------------------------
// Shadowing / overlap rules of the reference's trie build (tagdigger_fun.py
// :71-113), applied to a sorted array instead of a tree: at the node reached by
// prefix P the group is every sequence starting with P, in input order.
//   first member == P      -> it is stored, the rest of the group is dropped (:76-77)
//   a later member == P    -> AssertionError with that member's index (:82)
// survivors are prefix-free.  Index = input position mod numseq (:102-108).
// ---------------------------------------------------------------------------
struct Resolver {
    const std::vector<std::string> &seqs;
    uint32_t numseq;
    std::vector<uint32_t> order;                             // sorted by (string, position)
    std::vector<std::pair<std::string, uint32_t>> out;       // survivors: (sequence, index)
    int err = TD_OK;
    uint32_t bad = 0;

    Resolver(const std::vector<std::string> &s, uint32_t n) : seqs(s), numseq(n) {}
    uint32_t idx_of(uint32_t pos) const { return numseq ? pos % numseq : pos; }

    void walk(size_t lo, size_t hi, size_t depth) {
        if (err) return;
        uint32_t first = order[lo];
        for (size_t i = lo + 1; i < hi; i++) first = std::min(first, order[i]);
        if (seqs[first].size() == depth) {
            if (depth == 0) { err = TD_E_ROOTLEAF; return; }
            out.emplace_back(seqs[first], idx_of(first));
            return;
        }
        if (seqs[order[lo]].size() == depth) { err = TD_E_OVERLAP; bad = idx_of(order[lo]); return; }
        size_t i = lo;
        while (i < hi) {
            char c = seqs[order[i]][depth];
            size_t j = i + 1;
            while (j < hi && seqs[order[j]][depth] == c) j++;
            walk(i, j, depth + 1);
            if (err) return;
            i = j;
        }
    }
    int run() {
        if (seqs.empty()) return TD_E_EMPTY;
        for (auto &s : seqs) for (char c : s) if (base_code(c) < 0) return TD_E_ALPHABET;
        if (numseq == 1 && seqs.size() == 1 && seqs[0].empty()) {   // :109-110
            for (const char *b : {"A", "C", "G", "T"}) out.emplace_back(b, 0u);
            return TD_OK;
        }
        order.resize(seqs.size());
        for (uint32_t i = 0; i < seqs.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            int c = seqs[a].compare(seqs[b]);
            return c < 0 || (c == 0 && a < b);
        });
        walk(0, order.size(), 0);
        return err;
    }
};

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    int ensure(size_t want) {
        if (want <= n && p) return TD_OK;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipMalloc(&p, std::max<size_t>(want, 1) * sizeof(T)));
        n = want;
        return TD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

constexpr int W_CHOICES[] = {1, 2, 3, 4, 6, 10};
constexpr uint32_t MAX_SHORT = 16;
constexpr size_t LDS_BUDGET = 64 * 1024;     // per workgroup
constexpr size_t STATS_SLOTS = 32;           // TD_STAT_NSTATS public + diagnostic counters

}  // namespace

struct td_handle {
    int device = 0;
    int num_cu = 256;
    hipStream_t copy_stream = nullptr, work_stream = nullptr;
    // (one copy engine moves 22-30 GB/s out of pinned memory that lies on the far socket, two or three together 50: large
    // uploads are spread over copy_stream and these; TAGDIG_COPY_STREAMS = 1..3, default 3)
    hipStream_t side_copy[2] = {nullptr, nullptr}; hipEvent_t side_done[2] = {nullptr, nullptr};
    // index
    bool have_index = false;
    bool counted = false;                     // something has been counted since the results were last zeroed
    uint32_t barnum = 0, ntags = 0;
    int W = 2;
    uint32_t nch = 0, maxwo = 0, halo = 128, m_bases = 32, nshort = 0, bucket_mask = 0;
    uint32_t nch2 = 0;                        // k_fast2: 16-byte pieces packed from a line's first byte
    uint32_t bblob_bytes = 0, off_bmeta = 0, off_bdir = 0, off_bcand = 0;
    DevBuf<uint32_t> d_bblob;
    DevBuf<uint4> d_slots, d_shorts;
    // results
    DevBuf<uint32_t> d_counts;
    DevBuf<unsigned long long> d_counts64;
    uint32_t *bound_counts = nullptr;
    bool used64 = false;
    DevBuf<unsigned long long> d_stats;       // TD_STAT_NSTATS
    // progress windows (option "progress"): per 50 000 reads, how many had a barcode / a tag (reference :268-271)
    int progress = 0;
    uint32_t zb_members = 1u << 30;           // (tests: BGZF members per GPU batch, below the built-in 49 152)
    unsigned long long *pin_cursor = nullptr; // pinned: the line index after the newest counted BGZF batch
    int split_kernel = 2;                     // 2: k_split2 (tile in LDS), 1: k_split
    bool sp_sites_acgt = false;
    int last_fast_tile_kb = 0;                // tile size of the last free-running count launch (0: it took another path)
    uint32_t last_fast_ntiles = 0;
    uint32_t sp_gcap = 4;                     // entries per (barcode, last two bases) group of entries16
    std::vector<uint64_t> split_win;          // td_split_file's: {with barcode, clipped} per window of 50 000 reads
    DevBuf<unsigned long long> d_win;
    DevBuf<uint32_t> d_f4np;                  // k_fast4's producer count as k_f4_estimate leaves it
    DevBuf<uint32_t> d_tilesums;              // k_fast2's per-tile sums of what its wanted lines matched (progress windows)
    std::vector<uint64_t> host_acc;           // flushed counts
    uint64_t bytes_since_flush = 0;
    uint64_t flush_limit = 0xFFFFFFFFull;     // hits a uint32 cell may have taken before the matrix is flushed (tests lower it)
    // launch state
    DevBuf<uint64_t> d_state, d_tilecounts;
    DevBuf<uint32_t> d_ticket;
    DevBuf<unsigned long long> d_cursor;      // [2] line cursor for streamed pieces
    DevBuf<uint32_t> d_tileinfo, d_nfix;      // fast path: per-tile count+phase, fix-up queue length
    DevBuf<uint8_t> d_tail;                   // fast path: zero-padded copy of the buffer's last tiles
    // barcode splitter (td_set_splitter / td_split_*)
    bool have_splitter = false;
    std::vector<std::string> sp_barcodes;
    // td_split_file's two staging slots, kept between calls (pinned allocations cost tens of milliseconds)
    struct SplitSlot { uint8_t *pin = nullptr, *dev = nullptr; int2 *res_dev = nullptr, *res_pin = nullptr; size_t n = 0;
                       hipEvent_t done = nullptr; bool pending = false; } sp_slot[2];
    uint32_t sp_bblob_bytes = 0, sp_off_bmeta = 0, sp_off_bdir = 0, sp_cutlen = 0;
    unsigned long long sp_site[2] = {0, 0};
    uint32_t sp_site_len[2] = {0, 0};
    DevBuf<uint32_t> d_sp_bblob, d_sp_ent_begin, d_sp_ent_group;
    DevBuf<tdk::SplitEntry> d_sp_entries16;
    DevBuf<uint2> d_sp_e8;                    // k_split2: compact entries, 64 groups of eight per barcode
    DevBuf<uint8_t> d_sp_pool2;               // ... and their master strings at a stride of 128 bytes
    bool sp_compact = false;
    DevBuf<tdk::SplitEntry> d_sp_entries;
    DevBuf<uint8_t> d_sp_pool;
    DevBuf<uint4> d_fixlist;
    DevBuf<uint32_t> d_rowmap;                // td_fold_rows: sample row of every barcode row
    // BGZF members inflated on the GPU (count_bgzf_gpu): two batches in flight
    struct ZSlot { uint8_t *d_in = nullptr, *d_out = nullptr; tdinf::Member *pin_mem = nullptr, *d_mem = nullptr;
                   uint32_t *d_status = nullptr, *pin_status = nullptr; uint8_t *pin_tail = nullptr; hipEvent_t copied = nullptr; } zslot[2];
    struct ZPiece { uint8_t *pin = nullptr; hipEvent_t sent = nullptr; bool busy = false; } zpiece[2];     // pinned staging of the compressed bytes
    ZPiece ldpiece[2];                        // td_load_file_range's pinned staging
    // ordinary gzip: symbols decoded by the host, resolved on the GPU (count_gzip_dev): two batches in flight
    struct GSlot { DevBuf<uint8_t> d_sym, d_out, d_win; DevBuf<tdgz::Block> d_blk; DevBuf<uint32_t> d_crc;
                   tdgz::Block *pin_blk = nullptr; uint32_t *pin_crc = nullptr; uint8_t *pin_tail = nullptr; size_t pin_cap = 0; hipEvent_t done = nullptr;
                   hipEvent_t copied = nullptr; } gslot[2];
    DevBuf<uint32_t> d_gzflag;
    // ordinary gzip decoded on the GPU (count_gzip_gpu, gz_gpu.hpp): the whole file's buffers, kept between files
    struct GzGpu { DevBuf<uint8_t> d_in, d_in2, d_out, d_win, d_carry, d_segwin; DevBuf<uint16_t> d_maps; DevBuf<uint32_t> d_tok, d_crc; DevBuf<uint16_t> d_sym; DevBuf<tdgz2::Chunk> d_chunks;
                   DevBuf<tdgz2::ChunkOut> d_res; DevBuf<uint64_t> d_found, d_symoff; DevBuf<tdgz::Block> d_blk;
                   void release() { d_in.release(); d_in2.release(); d_out.release(); d_win.release(); d_carry.release(); d_tok.release(); d_crc.release(); d_sym.release();
                                    d_chunks.release(); d_res.release(); d_found.release(); d_symoff.release(); d_blk.release(); d_segwin.release(); d_maps.release(); } } gzgpu;
    bool gz_attr_done = false;
    // one gzip file over several devices: this rank's share between td_gz_shard_open / _decode / _resolve
    struct GzShard { bool open = false, decoded = false; std::string path; uint64_t n = 0, base = 0, in_bits = 0, nwords = 0, start_rel = 0;
                     std::vector<uint64_t> found; std::vector<tdgz2::Chunk> chunks; std::vector<tdgz2::ChunkOut> res; std::vector<uint64_t> sym_off;
                     uint64_t total = 0; uint32_t nseg = 0; } gzshard;
    int last_gz_route = 0;                    // how the last .gz file was decoded: 1 Huffman + LZ77 on the GPU, 0 otherwise
    int gpu_huffman = 1;                      // ordinary gzip: Huffman decoding on the GPU too (0: host threads decode, the GPU resolves)
    uint64_t gz_gpu_min = (uint64_t)8 << 20;  // ... for files of this many compressed bytes and more
    uint32_t gz_gpu_terr_kb = 128;            // ... one chunk per this much compressed data
    int gz_gpu_verify = 0;                    // ... the block search decodes a block before it believes its header
    uint32_t gz_gpu_seg_kb = 1u << 20, gz_gpu_margin_kb = 16384;      // ... a segment of compressed data (1 GiB), and how far its last chunk may run past it
    int gz_gpu_false_every = 0;               // (tests) every n-th found block start is moved by some bits: a false start
    int gpu_resolve = 1;                      // ordinary gzip of 8 MiB and more: markers -> bytes and CRC-32 on the GPU (0: all on the host)
    uint8_t *d_zscratch = nullptr; uint32_t *d_crctab = nullptr;
    uint32_t zcap_members = 0; size_t zcap_in = 0;   // what the batch buffers above were allocated for
    int gpu_inflate = 1;                      // BGZF input: inflate on the GPU (0: member-parallel on the host)
    int gpu_inflate_crc = 1;                  // ... and check every member's CRC-32 there
    uint32_t max_need = 0;                    // bases from a read's start that the matcher may look at
    // options
    int tile_kb = 32, blocks_per_cu = 0, prescan = 0, timing = 0, fastpath = 1, nt_loads = 1;
    int kernel_gen = 4;                       // main pass of the free-running path: 4 = k_fast4 (producer / consumer waves), 2 = k_fast2 (lazy packing), 1 = k_fast
    int tile_kb2 = 0;                         // k_fast2's tile (16 | 24 | 32 KiB; 0 = fast2_auto_tile)
    int run = 8;                              // k_fast2: consecutive tiles per workgroup turn (measured: 4-16 alike, 1 and 64+ slower)
    int hot_cache = 1;                        // k_fast2: count through the per-wave hot-cell cache
    int f4_nprod = 0;                         // k_fast4: producer waves of sixteen (0: k_f4_estimate's, from the buffer's line density)
    uint64_t fast_max_matrix = 1ull << 32;    // the free-running kernel addresses cells as base + 32-bit byte offset
    uint32_t debug_ablate = 0;
    double table_load = 0.25;
    int stagger = 0;
    int prio = 0xE4;            // wave priority per phase of the fast path: A 0, B-C 1, D 2, end of A 3 (kernel_fast.hpp set_prio)
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    const void *occ_fn = nullptr; size_t occ_lds = 0; int occ_val = 1;
};

namespace {

// count_bgzf_gpu's batch buffers (sized from the file, kept on the handle between files)
void release_bgzf_buffers(td_handle *h) {
    for (auto &zp : h->zpiece) {
        if (zp.pin) (void)hipHostFree(zp.pin);
        if (zp.sent) (void)hipEventDestroy(zp.sent);
        zp = td_handle::ZPiece();
    }
    for (auto &z : h->zslot) {
        if (z.d_in) (void)hipFree(z.d_in);
        if (z.d_out) (void)hipFree(z.d_out);
        if (z.pin_mem) (void)hipHostFree(z.pin_mem);
        if (z.d_mem) (void)hipFree(z.d_mem);
        if (z.d_status) (void)hipFree(z.d_status);
        if (z.pin_status) (void)hipHostFree(z.pin_status);
        if (z.pin_tail) (void)hipHostFree(z.pin_tail);
        if (z.copied) (void)hipEventDestroy(z.copied);
        z = td_handle::ZSlot();
    }
    if (h->d_zscratch) (void)hipFree(h->d_zscratch);
    h->d_zscratch = nullptr;
    h->zcap_members = 0; h->zcap_in = 0;
}

using KFn = void (*)(const tdk::KParams);
template <int CPT, bool TASSEL> KFn pick_w(int W) {
    switch (W) {
    case 1: return tdk::k_count<CPT, 1, TASSEL>;
    case 2: return tdk::k_count<CPT, 2, TASSEL>;
    case 3: return tdk::k_count<CPT, 3, TASSEL>;
    case 4: return tdk::k_count<CPT, 4, TASSEL>;
    case 6: return tdk::k_count<CPT, 6, TASSEL>;
    default: return tdk::k_count<CPT, 10, TASSEL>;
    }
}
KFn pick_kernel(int tile_kb, int W, bool tassel) {
    if (tassel) return pick_w<4, true>(W);
    return tile_kb == 32 ? pick_w<8, false>(W) : pick_w<4, false>(W);
}

using FFn = void (*)(const tdk::FParams);
template <int CPT, bool FIX> FFn pick_fast_w(int W) {
    switch (W) {
    case 1: return tdk::k_fast<CPT, 1, FIX>;
    case 2: return tdk::k_fast<CPT, 2, FIX>;
    case 3: return tdk::k_fast<CPT, 3, FIX>;
    case 4: return tdk::k_fast<CPT, 4, FIX>;
    case 6: return tdk::k_fast<CPT, 6, FIX>;
    default: return tdk::k_fast<CPT, 10, FIX>;
    }
}
FFn pick_fast(int tile_kb, int W, bool fix) {
    constexpr int C32 = 32 * 1024 / (tdk::FBLOCK * 16), C16 = 16 * 1024 / (tdk::FBLOCK * 16);      // chunks per thread
    if (fix) return tile_kb == 32 ? pick_fast_w<C32, true>(W) : pick_fast_w<C16, true>(W);
    return tile_kb == 32 ? pick_fast_w<C32, false>(W) : pick_fast_w<C16, false>(W);
}

// k_fast2 is instantiated for a few piece counts per tag width (NQ: 16-byte pieces packed per line);
// fast2_pieces rounds an index's need up to the next one
uint32_t fast2_pieces(int W, uint32_t need) {
    static const uint32_t opts[3][3] = {{3, 4, 6}, {5, 6, 8}, {7, 8, 10}};
    for (uint32_t o : opts[W - 1]) if (need <= o) return o;
    return opts[W - 1][2];
}
template <int CPT, bool PROG> FFn pick_fast2_c(int W, uint32_t nq) {
    switch (W) {
    case 1: return nq == 3 ? tdk::k_fast2<CPT, 1, 3, PROG> : nq == 4 ? tdk::k_fast2<CPT, 1, 4, PROG> : tdk::k_fast2<CPT, 1, 6, PROG>;
    case 2: return nq == 5 ? tdk::k_fast2<CPT, 2, 5, PROG> : nq == 6 ? tdk::k_fast2<CPT, 2, 6, PROG> : tdk::k_fast2<CPT, 2, 8, PROG>;
    default: return nq == 7 ? tdk::k_fast2<CPT, 3, 7, PROG> : nq == 8 ? tdk::k_fast2<CPT, 3, 8, PROG> : tdk::k_fast2<CPT, 3, 10, PROG>;
    }
}
// (PROG: the instantiation that also records, per phase-D pass, which wanted lines matched -- progress windows)
FFn pick_fast2(int tile_kb, int W, uint32_t nq, bool prog) {
    if (prog) return tile_kb == 32 ? pick_fast2_c<8, true>(W, nq) : tile_kb == 24 ? pick_fast2_c<6, true>(W, nq) : pick_fast2_c<4, true>(W, nq);
    return tile_kb == 32 ? pick_fast2_c<8, false>(W, nq) : tile_kb == 24 ? pick_fast2_c<6, false>(W, nq) : pick_fast2_c<4, false>(W, nq);
}
FFn pick_fix(int tile_kb, int W) {          // the fix-up pass shares the main pass's tile size
    if (tile_kb == 24) return pick_fast_w<6, true>(W);
    if (tile_kb == 12) return W == 1 ? tdk::k_fast<6, 1, true, 128> : W == 2 ? tdk::k_fast<6, 2, true, 128> : tdk::k_fast<6, 3, true, 128>;   // (k_fast4's tile: 128 threads)
    return pick_fast(tile_kb, W, true);
}
// k_fast4: producer and consumer waves, 16 KiB tiles (kernel_fast4.hpp)
template <bool PROG> FFn pick_fast4_p(int W, uint32_t nq) {
    switch (W) {
    case 1: return nq == 3 ? tdk::k_fast4<1, 3, PROG> : nq == 4 ? tdk::k_fast4<1, 4, PROG> : tdk::k_fast4<1, 6, PROG>;
    case 2: return nq == 5 ? tdk::k_fast4<2, 5, PROG> : nq == 6 ? tdk::k_fast4<2, 6, PROG> : tdk::k_fast4<2, 8, PROG>;
    default: return nq == 7 ? tdk::k_fast4<3, 7, PROG> : nq == 8 ? tdk::k_fast4<3, 8, PROG> : tdk::k_fast4<3, 10, PROG>;
    }
}
FFn pick_fast4(int W, uint32_t nq, bool prog) { return prog ? pick_fast4_p<true>(W, nq) : pick_fast4_p<false>(W, nq); }
size_t lds_bytes_fast4(const td_handle *h) {
    // three slots of raw tile + halo | per slot and producer the masks / line starts | hand-off words | the consumers' hot-cell caches | barcode index
    return (size_t)tdk::F4_SLOTS * (tdk::F4_TILE + h->halo) + (size_t)tdk::F4_SLOTS * tdk::F4_PROD * tdk::F4_WCH * 2 + tdk::F4_CTRL_BYTES +
           (size_t)(tdk::F4_WAVES - tdk::F4_NPROD_MIN) * tdk::HC_BYTES_PER_WAVE + h->bblob_bytes;
}
// k_fast2's tile: four workgroups must share a CU's 160 KiB of LDS (measured: three cost a fifth of the throughput),
// so a large barcode index (many barcodes x several concrete cut sites) takes the smaller tile
int fast2_auto_tile(const td_handle *h);
size_t lds_bytes_fast2(const td_handle *h, int tile_kb) {
    // raw tile + halo | terminator masks (later the list of wanted line starts) | misc | hot-cell cache | barcode index
    const size_t tile = (size_t)tile_kb * 1024;
    return tile + h->halo + tile / 16 * 2 + 256 + 4 * tdk::HC_BYTES_PER_WAVE + h->bblob_bytes;
}

int fast2_auto_tile(const td_handle *h) {
    // measured (profiles/r02_*): 24 KiB tiles with four workgroups per CU first; a barcode index too large for
    // that (config 5: 384 barcodes x 2 concrete cut sites) does better with 32 KiB tiles at three workgroups
    // than with 16 KiB tiles at four (a 16 KiB tile holds 75 reads of 100 bp: a second, nearly empty round of matching)
    if (lds_bytes_fast2(h, 24) <= 160 * 1024 / 4) return 24;
    if (lds_bytes_fast2(h, 32) <= 160 * 1024 / 3) return 32;
    if (lds_bytes_fast2(h, 24) <= LDS_BUDGET) return 24;
    return 16;
}

size_t lds_bytes_fast(const td_handle *h, int tile_kb) {
    size_t tile_ch = (size_t)tile_kb * 1024 / 16, halo_ch = h->halo / 16;
    return (tile_ch + halo_ch) * 8 + tile_ch * 4 + 256 + h->bblob_bytes;
}

size_t lds_bytes(const td_handle *h, int tile_kb) {
    // packed chunks (8 B per 16 bytes of tile + halo) | masks / line-start list | misc | barcode index
    size_t tile_ch = (size_t)tile_kb * 1024 / 16, halo_ch = h->halo / 16;
    return (tile_ch + halo_ch) * 8 + std::max<size_t>(tile_ch * 2, tdk::TLIST_CAP * 2) + 256 + h->bblob_bytes;
}

int zero_results(td_handle *h) {
    // (the handle's own matrix may be left over from a smaller index while a caller's matrix is bound: only what is
    // allocated is cleared; td_bind_counts(NULL) grows and clears it before it is used again)
    if (h->d_counts.p) HIPCHK(hipMemsetAsync(h->d_counts.p, 0, std::min<size_t>((size_t)h->barnum * h->ntags, h->d_counts.n) * 4, h->work_stream));
    if (h->d_counts64.p) HIPCHK(hipMemsetAsync(h->d_counts64.p, 0, (size_t)h->barnum * h->ntags * 8, h->work_stream));
    HIPCHK(hipMemsetAsync(h->d_stats.p, 0, STATS_SLOTS * 8, h->work_stream));
    if (h->d_win.p) HIPCHK(hipMemsetAsync(h->d_win.p, 0, h->d_win.n * 8, h->work_stream));
    HIPCHK(hipStreamSynchronize(h->work_stream));
    std::fill(h->host_acc.begin(), h->host_acc.end(), 0);
    h->bytes_since_flush = 0;
    h->used64 = false;
    h->counted = false;
    return TD_OK;
}

// move device uint32 counts into the host accumulator so cells cannot wrap
int flush_counts(td_handle *h) {
    if (h->bound_counts || !h->d_counts.p) return TD_OK;
    size_t cells = (size_t)h->barnum * h->ntags;
    std::vector<uint32_t> tmp(cells);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(tmp.data(), h->d_counts.p, cells * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d_counts.p, 0, cells * 4));
    if (h->host_acc.size() != cells) h->host_acc.assign(cells, 0);
    for (size_t i = 0; i < cells; i++) h->host_acc[i] += tmp[i];
    h->bytes_since_flush = 0;
    return TD_OK;
}

// Enqueue one pass of the count kernel.  cursor_in/out (device, optional) carry
// the line index between streamed pieces without a host round trip.
int launch_count(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, uint64_t max_reads,
                 int weights, hipStream_t stream, const unsigned long long *cursor_in = nullptr,
                 unsigned long long *cursor_out = nullptr, uint64_t first_line_ub = 0) {
    if (!h->have_index) return fail(TD_E_STATE, "td_set_index has not been called");
    if (((uintptr_t)d_fastq & 15) != 0) return fail(TD_E_ARG, "device FASTQ pointer must be 16-byte aligned");
    if (nbytes == 0) return TD_OK;
    if (max_reads == 0) max_reads = 1;
    h->counted = true;
    const bool tassel = weights != 0;
    // last countable sequence line: ordinal r (1-based) sits on line 4(r-1)+1
    const uint64_t limit_line = max_reads >= (1ull << 60) ? ~0ull - 8 : 4 * (max_reads - 1) + 1;
    // ---- the free-running path (predicted line phase, exact resolve, fix-up pass) is chosen when the maxreads
    // limit cannot bite early (it is still applied exactly, by fix-ups); streamed pieces carry their true first
    // line on the device, first_line_ub bounds it from above.  It addresses count cells as base + 32-bit offset.
    const uint64_t fl_ub = std::max(first_line, first_line_ub);
    const bool limit_far = limit_line >= ~0ull - 16 || limit_line - std::min(limit_line, fl_ub) >= nbytes / 16;
    const bool counts32 = (uint64_t)h->barnum * h->ntags * 4 < h->fast_max_matrix;
    // progress windows are recorded by k_fast4 and k_fast2 (+ k_resolve, fix-up pass) and by the exact kernel; not by k_fast's main pass
    // its main pass: k_fast2 (raw tile in LDS, lines packed by the lane that matches them) where the tag width has
    // the pipelined probe and the tile fits the LDS budget, else k_fast -- which keeps no progress records: with
    // progress on and no k_fast2 (wide tags, or a barcode index too large for its LDS layout) the exact kernel counts
    const int tkb2 = h->tile_kb2 ? h->tile_kb2 : fast2_auto_tile(h);
    // k_fast4 (producer and consumer waves, one workgroup of sixteen waves a CU) where its ring fits the CU's LDS beside the
    // barcode index; else k_fast2
    const bool gen4_fits = h->kernel_gen == 4 && h->W <= 3 && lds_bytes_fast4(h) <= 160 * 1024;     // (one workgroup of sixteen waves a CU)
    const bool gen2_fits = (h->kernel_gen == 2 || (h->kernel_gen == 4 && !gen4_fits)) && h->W <= 3 && lds_bytes_fast2(h, tkb2) <= LDS_BUDGET;
    const bool use_fast = h->fastpath && !tassel && !h->prescan && limit_far && counts32 && !(h->progress && !gen2_fits && !gen4_fits);
    const bool gen4 = use_fast && gen4_fits;
    const bool gen2 = use_fast && gen2_fits && !gen4;
    const int tile_kb = tassel ? 16 : gen4 ? (int)(tdk::F4_TILE / 1024) : gen2 ? tkb2 : h->tile_kb;
    const uint64_t tile = (uint64_t)tile_kb * 1024;
    const uint64_t ntiles64 = (nbytes + tile - 1) / tile;
    if (ntiles64 > 0x7FFFFFFFull) return fail(TD_E_LIMIT, "buffer too large for one launch; split it");
    const uint32_t ntiles = (uint32_t)ntiles64;

    if (!h->bound_counts && !tassel) {       // uint32 cells: a hit needs > 4 bytes of input
        if ((h->bytes_since_flush + nbytes) / 4 >= h->flush_limit) { int rc = flush_counts(h); if (rc) return rc; }
        h->bytes_since_flush += nbytes;
    }
    if (tassel) {
        if (!h->d_counts64.p) {
            int rc = h->d_counts64.ensure((size_t)h->barnum * h->ntags); if (rc) return rc;
            HIPCHK(hipMemsetAsync(h->d_counts64.p, 0, (size_t)h->barnum * h->ntags * 8, stream));
        }
        h->used64 = true;
    }
    { int rc = h->d_state.ensure(ntiles); if (rc) return rc; }

    tdk::KParams p{};
    p.buf = (const uint8_t *)d_fastq; p.nbytes = nbytes; p.first_line = first_line;
    p.limit_line = limit_line;
    p.state = h->d_state.p; p.ticket = h->d_ticket.p; p.ntiles = ntiles; p.halo = h->halo;
    p.bblob = h->d_bblob.p; p.bblob_bytes = h->bblob_bytes; p.off_bmeta = h->off_bmeta;
    p.off_bdir = h->off_bdir; p.off_bcand = h->off_bcand;
    p.buckets = h->d_slots.p; p.bucket_mask = h->bucket_mask; p.m_bases = h->m_bases;
    p.shorts = h->d_shorts.p; p.nshort = h->nshort;
    p.counts = h->bound_counts ? h->bound_counts : h->d_counts.p; p.counts64 = h->d_counts64.p;
    p.ncols = h->ntags; p.stats = h->d_stats.p; p.nch = h->nch; p.maxwo = h->maxwo;
    p.prefilled = h->prescan ? 1u : 0u;
    p.cursor_in = cursor_in; p.cursor_out = cursor_out;
    p.dbg = h->debug_ablate;
    p.stagger = (uint32_t)h->stagger; p.stagger_div = (uint32_t)h->num_cu;
    p.nt_loads = (uint32_t)h->nt_loads;
    p.prio = (uint32_t)h->prio;

    p.hot_cache = (uint32_t)h->hot_cache;
    p.run = (uint32_t)h->run;
    p.f4_nprod = (uint32_t)h->f4_nprod;                 // (0: k_f4_estimate's, below)
    if (h->progress) {
        // windows are indexed by the read's ordinal in the whole stream: a line holds a byte at least, a read four lines
        const uint64_t need = (fl_ub + nbytes) / 4 / tdk::PROG_WINDOW + 2;
        if (need > h->d_win.n) {
            const size_t cap = (size_t)std::max<uint64_t>(need * 2, 1u << 16);
            DevBuf<unsigned long long> bigger;
            int rc = bigger.ensure(cap); if (rc) return rc;
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemset(bigger.p, 0, cap * 8));
            if (h->d_win.p) HIPCHK(hipMemcpy(bigger.p, h->d_win.p, h->d_win.n * 8, hipMemcpyDeviceToDevice));
            h->d_win.release();
            h->d_win = bigger;
        }
        p.win = h->d_win.p; p.win_cap = (uint32_t)std::min<size_t>(h->d_win.n, 0xFFFFFFFFu);
    }
    h->last_fast_tile_kb = use_fast ? tile_kb : 0;        // (td_count_and_split_device: are d_tileinfo's counts the splitter's tiles?)
    h->last_fast_ntiles = use_fast ? ntiles : 0;
    if (use_fast) {
        int rc = h->d_tileinfo.ensure(ntiles); if (rc) return rc;
        const uint32_t fix_cap = 3u * ntiles + 8u;
        rc = h->d_fixlist.ensure(fix_cap); if (rc) return rc;
        rc = h->d_nfix.ensure(4); if (rc) return rc;
        {   // the tiles whose window (tile + halo) crosses the end of the buffer: a zero-padded copy
            const uint64_t TILE = (uint64_t)tile_kb * 1024;
            const uint64_t inside = nbytes >= TILE + h->halo ? (nbytes - h->halo) / TILE : 0;
            const size_t tail_cap = 2 * (size_t)TILE + 2 * (size_t)h->halo + 256;
            rc = h->d_tail.ensure(tail_cap); if (rc) return rc;
            const uint64_t start = inside * TILE;
            if (nbytes - start > tail_cap) return fail(TD_E_INTERNAL, "tail copy larger than its buffer");
            HIPCHK(hipMemsetAsync(h->d_tail.p, 0, tail_cap, stream));
            if (nbytes > start)
                HIPCHK(hipMemcpyAsync(h->d_tail.p, (const uint8_t *)d_fastq + start, nbytes - start, hipMemcpyDeviceToDevice, stream));
            p.tail_buf = h->d_tail.p; p.tail_tile = (uint32_t)inside;
        }
        tdk::FParams fp{};
        fp.k = p; fp.tile_info = h->d_tileinfo.p; fp.fixlist = h->d_fixlist.p; fp.nfix = h->d_nfix.p; fp.fix_cap = fix_cap;
        if (h->progress) {
            rc = h->d_tilesums.ensure(ntiles); if (rc) return rc;
            HIPCHK(hipMemsetAsync(h->d_tilesums.p, 0, (size_t)ntiles * 4, stream));
            fp.tile_sums = h->d_tilesums.p;
        }
        HIPCHK(hipMemsetAsync(h->d_nfix.p, 0, 4, stream));
        FFn ffn = gen4 ? pick_fast4(h->W, h->nch2, h->progress != 0) : gen2 ? pick_fast2(tile_kb, h->W, h->nch2, h->progress != 0) : pick_fast(tile_kb, h->W, false), fixfn = pick_fix(tile_kb, h->W);
        const size_t flds = gen4 ? lds_bytes_fast4(h) : gen2 ? lds_bytes_fast2(h, tile_kb) : lds_bytes_fast(h, tile_kb), fixlds = lds_bytes_fast(h, tile_kb);
        const unsigned main_threads = gen4 ? (unsigned)tdk::F4_BLOCK : (unsigned)tdk::FBLOCK, fix_threads = tile_kb == 12 ? 128u : (unsigned)tdk::FBLOCK;
        if (flds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void *)ffn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
        if (fixlds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void *)fixfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fixlds));
        int bpc = h->blocks_per_cu;
        if (bpc <= 0) {
            if (h->occ_fn != (const void *)ffn || h->occ_lds != flds) {
                int occ = 0;
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)ffn, (int)main_threads, flds));
                h->occ_fn = (const void *)ffn; h->occ_lds = flds; h->occ_val = std::max(1, occ);
            }
            bpc = h->occ_val;
        }
        const uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)h->num_cu * bpc);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->timing) {
            if (h->ev_used == h->ev_pool.size()) {
                hipEvent_t a, b;
                HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
                h->ev_pool.emplace_back(a, b);
            }
            e0 = h->ev_pool[h->ev_used].first; e1 = h->ev_pool[h->ev_used].second; h->ev_used++;
            HIPCHK(hipEventRecord(e0, stream));
        }
        if (gen4 && !h->f4_nprod) {
            // (the tail copy holds the bytes of a buffer shorter than a tile)
            int rc4 = h->d_f4np.ensure(4); if (rc4) return rc4;
            hipLaunchKernelGGL(tdk::k_f4_estimate, dim3(1), dim3(256), 0, stream, (const uint8_t *)d_fastq, nbytes, h->d_f4np.p);
            fp.k.f4_nprod_dev = h->d_f4np.p;
        }
        hipLaunchKernelGGL(ffn, dim3(grid), dim3(main_threads), flds, stream, fp);
        {   // exact line phase of every tile (d_state is free on this path: it holds the block sums)
            const uint32_t rblocks = (ntiles + tdk::RESOLVE_SPAN - 1) / tdk::RESOLVE_SPAN;
            unsigned long long *super = reinterpret_cast<unsigned long long *>(h->d_state.p);
            hipLaunchKernelGGL(tdk::k_resolve_sums, dim3(rblocks), dim3(256), 0, stream, fp, super);
            hipLaunchKernelGGL(tdk::k_resolve, dim3(rblocks), dim3(1024), 0, stream, fp, super);
        }
        hipLaunchKernelGGL(fixfn, dim3(std::min<uint32_t>(grid, (uint32_t)h->num_cu * 2)), dim3(fix_threads), fixlds, stream, fp);
        HIPCHK(hipGetLastError());
        if (h->timing) HIPCHK(hipEventRecord(e1, stream));
        return TD_OK;
    }

    HIPCHK(hipMemsetAsync(h->d_ticket.p, 0, 4, stream));
    if (h->prescan) {
        int rc = h->d_tilecounts.ensure(ntiles); if (rc) return rc;
        uint32_t g = std::min<uint32_t>(ntiles, (uint32_t)h->num_cu * 8);
        if (tile_kb == 32) hipLaunchKernelGGL((tdk::k_count_lines<8>), dim3(g), dim3(tdk::BLOCK), 0, stream, p.buf, nbytes, ntiles, h->d_tilecounts.p);
        else hipLaunchKernelGGL((tdk::k_count_lines<4>), dim3(g), dim3(tdk::BLOCK), 0, stream, p.buf, nbytes, ntiles, h->d_tilecounts.p);
        hipLaunchKernelGGL(tdk::k_scan_tiles, dim3(1), dim3(1024), 0, stream, h->d_tilecounts.p, ntiles, h->d_state.p, (unsigned long long *)nullptr);
    } else {
        HIPCHK(hipMemsetAsync(h->d_state.p, 0, (size_t)ntiles * 8, stream));
    }

    KFn fn = pick_kernel(tile_kb, h->W, tassel);
    const size_t lds = lds_bytes(h, tile_kb);
    if (lds > 48 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = h->blocks_per_cu;
    if (bpc <= 0) {
        if (h->occ_fn != (const void *)fn || h->occ_lds != lds) {
            int occ = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, tdk::BLOCK, lds));
            h->occ_fn = (const void *)fn; h->occ_lds = lds; h->occ_val = std::max(1, occ);
        }
        bpc = h->occ_val;
    }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)h->num_cu * bpc);

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        if (h->ev_used == h->ev_pool.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
            h->ev_pool.emplace_back(a, b);
        }
        e0 = h->ev_pool[h->ev_used].first; e1 = h->ev_pool[h->ev_used].second; h->ev_used++;
        HIPCHK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(tdk::BLOCK), lds, stream, p);
    HIPCHK(hipGetLastError());
    if (h->timing) HIPCHK(hipEventRecord(e1, stream));
    return TD_OK;
}

int check_device_errors(td_handle *h, const unsigned long long *st) {
    unsigned long long e = st[tdk::ST_ERR];
    if (e & tdk::ERR_SPIN) return fail(TD_E_INTERNAL, "look-back wait timed out inside the count kernel");
    if (e & tdk::ERR_NONASCII) return fail(TD_E_NONASCII, "non-ASCII byte in a sequence line");
    if (e & tdk::ERR_TASSEL) return fail(TD_E_TASSEL, "invalid literal for int() with base 10 (count= header)");
    (void)h;
    return TD_OK;
}

}  // namespace

// ============================================================================ C-ABI
namespace {
// barcode (+ cut site) index as the kernels read it from LDS:
// bval u64[ne] | bmeta u32[ne] | bdir u16[1024]; entries stored bucket by bucket.
// `entries`: the resolved (sequence, row) set; tagoff[row] goes into bmeta's offset field.
int build_barcode_blob(const std::vector<std::pair<std::string, uint32_t>> &entries, uint32_t barnum, const uint32_t *tagoff,
                       std::vector<uint8_t> &blob, uint32_t &off_bmeta, uint32_t &off_bdir, uint32_t &max_off) {
    const size_t nb = entries.size();
    if (barnum > 65535) return fail(TD_E_LIMIT, "more than 65535 barcodes");
    max_off = 0;
    std::vector<std::vector<uint32_t>> buckets(tdk::BDIR_SIZE);
    std::vector<uint64_t> eval(nb);
    std::vector<uint32_t> emeta(nb);
    for (size_t e = 0; e < nb; e++) {
        const std::string &s = entries[e].first;
        const uint32_t row = entries[e].second;
        if (s.size() > 32) return fail(TD_E_LIMIT, "barcode+cutsite longer than 32 bases");
        const uint32_t off = tagoff[row];
        if (off > 63) return fail(TD_E_LIMIT, "tag offset beyond 63 bases");
        max_off = std::max(max_off, off);
        pack_bases(s, &eval[e], 1);
        emeta[e] = (uint32_t)s.size() | (off << 6) | (row << 16);
        const uint32_t L = (uint32_t)s.size();
        const uint32_t base = (uint32_t)(eval[e] >> (64 - 2 * tdk::BDIR_BASES));
        const uint32_t span = L >= tdk::BDIR_BASES ? 1u : 1u << (2 * (tdk::BDIR_BASES - L));
        for (uint32_t k = 0; k < span; k++) buckets[base + k].push_back((uint32_t)e);
    }
    std::vector<uint16_t> bdir(tdk::BDIR_SIZE, 0xFFFF);
    std::vector<uint64_t> bval;
    std::vector<uint32_t> bmeta;
    for (uint32_t b = 0; b < tdk::BDIR_SIZE; b++) {
        if (buckets[b].empty()) continue;
        if (bval.size() + buckets[b].size() > 65534) return fail(TD_E_LIMIT, "barcode directory too large");
        bdir[b] = (uint16_t)bval.size();
        for (size_t k = 0; k < buckets[b].size(); k++) {
            bval.push_back(eval[buckets[b][k]]);
            bmeta.push_back(emeta[buckets[b][k]] | (k + 1 == buckets[b].size() ? tdk::BMETA_LAST : 0u));
        }
    }
    const size_t ne = bval.size();
    off_bmeta = (uint32_t)(ne * 8);
    off_bdir = off_bmeta + (uint32_t)((ne * 4 + 7) / 8 * 8);
    blob.assign((off_bdir + tdk::BDIR_SIZE * 2 + 15) / 16 * 16, 0);
    memcpy(blob.data(), bval.data(), ne * 8);
    memcpy(blob.data() + off_bmeta, bmeta.data(), ne * 4);
    memcpy(blob.data() + off_bdir, bdir.data(), tdk::BDIR_SIZE * 2);
    return TD_OK;
}
}  // namespace

extern "C" {

const char *td_last_error(void) { return g_err.c_str(); }
uint32_t td_last_bad_index(void) { return g_bad; }

}  // extern "C"
namespace { void handle_born(); void handle_gone(); }   // (the pinned pool of the gzip decoder follows the handles' lives: below)
extern "C" {
int td_create(td_handle **out, int device_id) {
    if (!out) return fail(TD_E_ARG, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TD_E_HIP, "no usable HIP device (libtagdig has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(TD_E_ARG, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    td_handle *h = new td_handle();
    h->device = device_id;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
        HIPCHK(hipStreamCreateWithFlags(&h->side_copy[k], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->side_done[k], hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&h->work_stream, hipStreamNonBlocking));
    int rc = h->d_stats.ensure(STATS_SLOTS); if (rc) { delete h; return rc; }
    rc = h->d_ticket.ensure(4); if (rc) { delete h; return rc; }
    rc = h->d_cursor.ensure(2); if (rc) { delete h; return rc; }
    HIPCHK(hipMemset(h->d_stats.p, 0, STATS_SLOTS * 8));
    *out = h;
    handle_born();
    return TD_OK;
}

void td_destroy(td_handle *h) {
    if (!h) return;
    handle_gone();
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    h->d_bblob.release(); h->d_slots.release(); h->d_shorts.release(); h->d_counts.release();
    if (h->pin_cursor) (void)hipHostFree(h->pin_cursor);
    h->d_win.release(); h->d_tilesums.release(); h->d_f4np.release(); h->d_sp_entries16.release(); h->d_sp_e8.release(); h->d_sp_pool2.release();
    h->d_counts64.release(); h->d_stats.release(); h->d_state.release(); h->d_tilecounts.release();
    h->d_ticket.release(); h->d_cursor.release(); h->d_tileinfo.release(); h->d_nfix.release(); h->d_tail.release(); h->d_fixlist.release(); h->d_rowmap.release();
    for (auto &ev : h->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto &sl : h->sp_slot) {
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.res_dev) (void)hipFree(sl.res_dev);
        if (sl.res_pin) (void)hipHostFree(sl.res_pin);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    release_bgzf_buffers(h);
    for (auto &lp : h->ldpiece) { if (lp.pin) (void)hipHostFree(lp.pin); if (lp.sent) (void)hipEventDestroy(lp.sent); }
    for (auto &g : h->gslot) {
        g.d_sym.release(); g.d_out.release(); g.d_win.release(); g.d_blk.release(); g.d_crc.release();
        if (g.pin_blk) (void)hipHostFree(g.pin_blk);
        if (g.pin_crc) (void)hipHostFree(g.pin_crc);
        if (g.pin_tail) (void)hipHostFree(g.pin_tail);
        if (g.copied) (void)hipEventDestroy(g.copied);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    h->d_gzflag.release();
    h->gzgpu.release();
    if (h->d_crctab) (void)hipFree(h->d_crctab);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (auto &st : h->side_copy) if (st) (void)hipStreamDestroy(st);
    for (auto &ev : h->side_done) if (ev) (void)hipEventDestroy(ev);
    if (h->work_stream) (void)hipStreamDestroy(h->work_stream);
    delete h;
}

int td_set_index(td_handle *h, const char *const *barcut, uint32_t n_barcut, uint32_t barnum,
                 const uint32_t *tagoff, const char *const *tags, uint32_t ntags) {
    if (!h) return fail(TD_E_ARG, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    h->have_index = false;
    std::vector<std::string> bs(n_barcut), ts(ntags);
    for (uint32_t i = 0; i < n_barcut; i++) bs[i] = barcut[i];
    for (uint32_t i = 0; i < ntags; i++) ts[i] = tags[i];

    Resolver rb(bs, barnum);
    int rc = rb.run();
    if (rc) { g_bad = rb.bad; return fail(rc, rc == TD_E_OVERLAP ? "overlapping barcode+cutsite sequences" : "barcode index build failed"); }
    Resolver rt(ts, ntags);
    rc = rt.run();
    if (rc) { g_bad = rt.bad; return fail(rc, rc == TD_E_OVERLAP ? "overlapping tags" : "tag index build failed"); }

    // ---- barcode blob
    uint32_t max_off = 0;
    std::vector<uint8_t> blob;
    rc = build_barcode_blob(rb.out, barnum, tagoff, blob, h->off_bmeta, h->off_bdir, max_off);
    if (rc) return rc;
    h->off_bcand = 0;
    h->bblob_bytes = (uint32_t)blob.size();

    // ---- tag table
    size_t maxlen = 0;
    std::vector<uint32_t> lens;
    for (auto &t : rt.out) { maxlen = std::max(maxlen, t.first.size()); lens.push_back((uint32_t)t.first.size()); }
    int W = 0;
    for (int w : W_CHOICES) if ((size_t)w * 32 >= maxlen) { W = w; break; }
    if (!W) return fail(TD_E_LIMIT, "tag longer than 320 bases");
    std::sort(lens.begin(), lens.end());
    uint32_t m = 32;
    if (lens.size() > MAX_SHORT) m = std::min<uint32_t>(32, lens[MAX_SHORT]);
    m = std::max<uint32_t>(m, 1);
    // buckets: dword 0 = overflow filter (kernels.hpp KParams::buckets), then SPB slots of {W x u64, u32 meta = col<<10 | len}
    if (ntags >= (1u << 22)) return fail(TD_E_LIMIT, "more than 4M tags");
    const int bucket_dw = W <= 3 ? 4 * TD_BU4 : 32;
    const int slot_dw = 2 * W + 1;
    const int spb = (bucket_dw - 1) / slot_dw;
    size_t nlong = 0;
    for (auto &t : rt.out) if (t.first.size() >= m) nlong++;
    size_t nbuckets = 16;
    // load: a quarter of the slots.  A key that found its bucket full sits in the next one, and a read that
    // matches it costs a second, dependent fetch -- for its whole wave.  At half load 3.5 % of the keys are
    // displaced and four waves in five take that path; at a quarter 0.3 % and one wave in six (measured at
    // 384 x 100 k: 12.4 -> 11.6 ms; at an eighth 11.3 ms but 500 k tags lose what they gain to the larger table)
    while ((double)nbuckets * spb * h->table_load < (double)nlong) nbuckets <<= 1;
    if ((uint64_t)nbuckets * bucket_dw * 4 >= (1ull << 32)) return fail(TD_E_LIMIT, "tag table beyond 4 GiB");   // 32-bit bucket offsets
    std::vector<uint32_t> slots(nbuckets * bucket_dw, 0);
    std::vector<uint32_t> shorts;
    std::vector<uint64_t> words(W);
    for (auto &t : rt.out) {
        const uint32_t L = (uint32_t)t.first.size();
        pack_bases(t.first, words.data(), W);
        if (L < m) {
            shorts.push_back((uint32_t)words[0]); shorts.push_back((uint32_t)(words[0] >> 32));
            shorts.push_back(L); shorts.push_back(t.second);
            continue;
        }
        const uint32_t hk = hash_key(words[0] >> (64 - 2 * m));
        size_t b = hk & (nbuckets - 1);
        for (;;) {
            uint32_t *bp = &slots[b * bucket_dw];
            int free_slot = -1;
            for (int k = 0; k < spb; k++) if (bp[1 + k * slot_dw + 2 * W] == 0) { free_slot = k; break; }
            if (free_slot >= 0) {
                uint32_t *sp = bp + 1 + free_slot * slot_dw;
                for (int w = 0; w < W; w++) { sp[2 * w] = (uint32_t)words[w]; sp[2 * w + 1] = (uint32_t)(words[w] >> 32); }
                sp[2 * W] = (t.second << 10) | L;
                break;
            }
            bp[0] |= 1u << (hk >> 27);        // full: lookups of keys with this filter bit that miss here must go on
            b = (b + 1) & (nbuckets - 1);
        }
    }
    const size_t nslots = nbuckets;           // (device buffer sized in uint4 below)
    const int slot_u4 = bucket_dw / 4;
    h->W = W; h->m_bases = m; h->bucket_mask = (uint32_t)(nbuckets - 1); h->nshort = (uint32_t)(shorts.size() / 4);
    h->maxwo = max_off >> 4;
    const uint32_t need = 15 + max_off + (uint32_t)maxlen;
    h->max_need = max_off + (uint32_t)maxlen;
    h->nch = std::min<uint32_t>(2 * W + 3, std::max<uint32_t>(3, (need + 15) / 16));
    h->nch2 = W <= 3 ? fast2_pieces(W, (h->max_need + 15) / 16) : 0;
    // bytes staged behind a tile: a line that starts in its last byte is matched from there (k_fast: chunk-aligned,
    // nch chunks; k_fast2: nch2 pieces from the line's own first byte, and its vote reads 8 bytes)
    h->halo = (std::max(h->nch * 16, h->nch2 * 16 + 16) + 63) / 64 * 64;
    h->barnum = barnum; h->ntags = ntags;
    if (lds_bytes(h, 16) > LDS_BUDGET) return fail(TD_E_LIMIT, "barcode index does not fit the LDS budget");
    if (lds_bytes(h, h->tile_kb) > LDS_BUDGET) h->tile_kb = 16;
    if (h->tile_kb2 && lds_bytes_fast2(h, h->tile_kb2) > LDS_BUDGET) h->tile_kb2 = 0;     // (back to the automatic choice)

    rc = h->d_bblob.ensure(h->bblob_bytes / 4); if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_bblob.p, blob.data(), h->bblob_bytes, hipMemcpyHostToDevice));
    rc = h->d_slots.ensure(nslots * slot_u4); if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_slots.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
    rc = h->d_shorts.ensure(std::max<size_t>(1, shorts.size() / 4)); if (rc) return rc;
    if (!shorts.empty()) HIPCHK(hipMemcpy(h->d_shorts.p, shorts.data(), shorts.size() * 4, hipMemcpyHostToDevice));
    h->d_counts64.release();
    if (!h->bound_counts) { rc = h->d_counts.ensure((size_t)barnum * ntags); if (rc) return rc; }
    h->host_acc.assign((size_t)barnum * ntags, 0);
    h->have_index = true;
    return zero_results(h);
}

int td_bind_counts(td_handle *h, void *d_counts) {
    if (!h) return fail(TD_E_ARG, "handle is NULL");
    h->bound_counts = (uint32_t *)d_counts;
    if (!d_counts && h->have_index) { int rc = h->d_counts.ensure((size_t)h->barnum * h->ntags); if (rc) return rc; return zero_results(h); }
    return TD_OK;
}

int td_reset(td_handle *h) {
    if (!h) return fail(TD_E_ARG, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    return zero_results(h);
}

int td_count_device(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, uint64_t max_reads,
                    int weights, void *stream) {
    if (!h) return fail(TD_E_ARG, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    return launch_count(h, d_fastq, nbytes, first_line, max_reads, weights, (hipStream_t)stream);
}

int td_count_lines_device(td_handle *h, const void *d_fastq, uint64_t nbytes, void *stream, uint64_t *out) {
    if (!h || !out) return fail(TD_E_ARG, "NULL argument");
    if (((uintptr_t)d_fastq & 15) != 0) return fail(TD_E_ARG, "device FASTQ pointer must be 16-byte aligned");
    HIPCHK(hipSetDevice(h->device));
    *out = 0;
    if (nbytes == 0) return TD_OK;
    const uint64_t tile = 16 * 1024;
    const uint64_t nt = (nbytes + tile - 1) / tile;
    if (nt > 0x7FFFFFFFull) return fail(TD_E_LIMIT, "buffer too large for one launch; split it");
    int rc = h->d_tilecounts.ensure(nt); if (rc) return rc;
    rc = h->d_state.ensure(nt); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    uint32_t g = (uint32_t)std::min<uint64_t>(nt, (uint64_t)h->num_cu * 8);
    hipLaunchKernelGGL((tdk::k_count_lines<4>), dim3(g), dim3(tdk::BLOCK), 0, s, (const uint8_t *)d_fastq, nbytes, (uint32_t)nt, h->d_tilecounts.p);
    hipLaunchKernelGGL(tdk::k_scan_tiles, dim3(1), dim3(1024), 0, s, h->d_tilecounts.p, (uint32_t)nt, h->d_state.p, h->d_cursor.p);
    HIPCHK(hipGetLastError());
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_cursor.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *out = v;
    return TD_OK;
}

}  // extern "C"

// ---- host buffers and files: pieces cut at line ends, staged through pinned memory
namespace {
int copy_lanes() {
    static const int n = getenv("TAGDIG_COPY_STREAMS") ? std::max(1, std::min(3, atoi(getenv("TAGDIG_COPY_STREAMS")))) : 3;
    return n;
}
// host -> device on the copy stream, large ones in up to three parts side by side on the handle's other copy streams; what is
// queued on copy_stream afterwards (an event, say) comes after all of it
int upload(td_handle *h, void *dst, const void *src, size_t n) {
    const int lanes = n >= ((size_t)4 << 20) ? copy_lanes() : 1;
    const size_t part = ((n + lanes - 1) / lanes + 4095) & ~(size_t)4095;
    for (int k = 1; k < lanes; k++) {
        const size_t off = std::min(n, (size_t)k * part), len = std::min(n, off + part) - off;
        if (!len) continue;
        HIPCHK(hipMemcpyAsync((uint8_t *)dst + off, (const uint8_t *)src + off, len, hipMemcpyHostToDevice, h->side_copy[k - 1]));
        HIPCHK(hipEventRecord(h->side_done[k - 1], h->side_copy[k - 1]));
    }
    HIPCHK(hipMemcpyAsync(dst, src, std::min(n, part), hipMemcpyHostToDevice, h->copy_stream));
    for (int k = 1; k < lanes; k++) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->side_done[k - 1], 0));
    return TD_OK;
}

struct Stager {
    td_handle *h = nullptr;
    static constexpr int NB = 3;
    size_t cap = 0;
    uint8_t *pin[NB] = {nullptr, nullptr, nullptr};
    uint8_t *dev[NB] = {nullptr, nullptr, nullptr};
    hipEvent_t copied[NB] = {nullptr, nullptr, nullptr};  // H2D of buffer i landed
    hipEvent_t done[NB] = {nullptr, nullptr, nullptr};    // kernel on buffer i finished
    bool busy[NB] = {false, false, false};
    int cur = 0;
    uint64_t first_line = 0;
    uint64_t bytes_submitted = 0;             // every line holds >= 1 byte: bounds the line index from above
    unsigned pieces = 0;
    unsigned long long *lines_after = nullptr;   // pinned: line terminators counted up to and including buffer i's piece
    uint64_t lines_seen = 0;                     // ... of the newest piece known to have drained (a lower bound of the cursor)
    int init(td_handle *hh, size_t capacity, uint64_t first) {
        h = hh; cap = capacity; first_line = first;
        HIPCHK(hipHostMalloc((void **)&lines_after, NB * sizeof(unsigned long long), hipHostMallocDefault));
        for (int i = 0; i < NB; i++) {
            HIPCHK(hipHostMalloc((void **)&pin[i], cap, hipHostMallocDefault));
            HIPCHK(hipMalloc((void **)&dev[i], cap));
            HIPCHK(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        HIPCHK(hipMemsetAsync(h->d_cursor.p, 0, 16, h->work_stream));
        return TD_OK;
    }
    ~Stager() {
        if (lines_after) (void)hipHostFree(lines_after);
        for (int i = 0; i < NB; i++) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (dev[i]) (void)hipFree(dev[i]);
            if (copied[i]) (void)hipEventDestroy(copied[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
    }
    // buffer to fill next (waits until its previous use has drained)
    int acquire(uint8_t **p) {
        if (busy[cur]) { HIPCHK(hipEventSynchronize(done[cur])); busy[cur] = false; lines_seen = lines_after[cur]; }
        *p = pin[cur];
        return TD_OK;
    }
    // copy on the copy stream, count on the work stream; the line index travels
    // from piece to piece through d_cursor[pieces & 1] on the device
    int submit(size_t n, uint64_t max_reads, int weights) {
        if (n == 0) return TD_OK;
        { const int rc = upload(h, dev[cur], pin[cur], n); if (rc) return rc; }
        HIPCHK(hipEventRecord(copied[cur], h->copy_stream));
        HIPCHK(hipStreamWaitEvent(h->work_stream, copied[cur], 0));
        int rc = launch_count(h, dev[cur], n, first_line, max_reads, weights, h->work_stream,
                              h->d_cursor.p + (pieces & 1), h->d_cursor.p + ((pieces + 1) & 1),
                              first_line + bytes_submitted);
        if (rc) return rc;
        bytes_submitted += n;
        HIPCHK(hipMemcpyAsync(lines_after + cur, h->d_cursor.p + ((pieces + 1) & 1), 8, hipMemcpyDeviceToHost, h->work_stream));
        HIPCHK(hipEventRecord(done[cur], h->work_stream));
        busy[cur] = true;
        pieces++;
        cur = (cur + 1) % NB;
        return TD_OK;
    }
    int finish(uint64_t *lines) {
        unsigned long long v = 0;
        HIPCHK(hipMemcpyAsync(&v, h->d_cursor.p + (pieces & 1), 8, hipMemcpyDeviceToHost, h->work_stream));
        HIPCHK(hipStreamSynchronize(h->work_stream));
        HIPCHK(hipStreamSynchronize(h->copy_stream));
        if (lines) *lines = v;
        return TD_OK;
    }
};

// index just past the last complete line terminator of p[0..n); 0 if none.
// A trailing '\r' is not trusted (its '\n' may be in the next piece).
size_t cut_at_line_end(const uint8_t *p, size_t n) {
    size_t i = n;
    if (i > 0 && p[i - 1] == '\r') i--;
    while (i > 0) {
        if (p[i - 1] == '\n' || p[i - 1] == '\r') return i;
        i--;
    }
    return 0;
}

// Staging a piece into pinned memory on several host threads (one memcpy or pread stream does not
// reach PCIe speed): `part(offset, n)` fills bytes [offset, offset + n) of the piece.
inline int stage_threads() {
    static const int n = []() {
        const char *env = getenv("TAGDIG_STAGE_THREADS");
        // (default: 16 where the host has the cores -- measured 46 vs 45 GB/s from a host buffer, 36 vs 32 GB/s from a
        // plain file against 8 -- else 8)
        const long v = env ? atol(env) : (std::thread::hardware_concurrency() >= 32 ? 16 : 8);
        return (int)std::max<long>(1, std::min<long>(v, 16));
    }();
    return n;
}
template <typename Part>
bool stage_parallel(size_t total, Part &&part) {
    const int nt = (int)std::min<size_t>((size_t)stage_threads(), (total >> 20) + 1);      // (a thread per MiB at most)
    if (nt <= 1) return part(0, total);
    std::atomic<bool> ok{true};
    std::vector<std::thread> pool;
    const size_t chunk = ((total + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (int t = 1; t < nt; t++) {
        const size_t off = std::min(total, (size_t)t * chunk), n = std::min(total, off + chunk) - off;
        if (n) pool.emplace_back([&, off, n]() { if (!part(off, n)) ok = false; });
    }
    if (!part(0, std::min(total, chunk))) ok = false;
    for (auto &th : pool) th.join();
    return ok;
}

// generic pump: `reader(dst, want)` returns bytes produced (0 at end, <0 on error)
template <typename Reader>
int pump(td_handle *h, Reader &&reader, uint64_t size_hint, uint64_t first_line, uint64_t max_reads, int weights,
         uint64_t *lines_out) {
    Stager st;
    size_t cap = (size_t)32 << 20;
    if (size_hint && size_hint < cap) cap = std::max<size_t>(1 << 16, (size_hint + 4095) / 4096 * 4096);
    int rc = st.init(h, cap, first_line); if (rc) return rc;
    std::vector<uint8_t> carry;
    bool eof = false;
    // (reference :272: the loop ends at maxreads -- here the input stops being read once a drained piece's line
    // index shows that the bound has been passed; the kernels ignore reads past it either way)
    const uint64_t stop_line = max_reads >= (1ull << 60) ? ~0ull : 4 * (std::max<uint64_t>(1, max_reads) - 1) + 2;
    while (!eof) {
        uint8_t *buf; rc = st.acquire(&buf); if (rc) return rc;
        if (first_line + st.lines_seen >= stop_line) break;
        size_t have = carry.size();
        if (have) memcpy(buf, carry.data(), have);
        carry.clear();
        while (have < cap) {
            long got = reader(buf + have, cap - have);
            if (got < 0) return fail(TD_E_IO, "read error while streaming FASTQ");
            if (got == 0) { eof = true; break; }
            have += (size_t)got;
        }
        size_t cut = have;
        if (!eof) {
            cut = cut_at_line_end(buf, have);
            if (cut == 0) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
            carry.assign(buf + cut, buf + have);
        }
        rc = st.submit(cut, max_reads, weights); if (rc) return rc;
    }
    return st.finish(lines_out);
}
}  // namespace

// ---- BGZF input inflated on the GPU (second option).  The host only maps the file and walks the member
// headers; batches of compressed members go over PCIe (a quarter of the bytes), tdinf::k_bgzf_inflate inflates one
// member per lane, every member's size and CRC-32 are checked on the device, the batch is cut at its last line end
// and counted where it lies, and what is left of the last line moves to the front of the next batch's buffer.
namespace {
// members per batch: one lane each, and the decoder's speed is all latency (one dependent table look-up and memory
// access after the other) -- 16 384 members are ONE wave per CU (13-19 GB/s measured), 49 152 fill the three waves
// per CU that the LDS tables allow
constexpr uint32_t ZB_MEMBERS = 49152;
constexpr size_t ZB_IN = (size_t)1280 << 20;                 // compressed bytes per batch, at most
constexpr size_t ZB_PIECE = (size_t)64 << 20;                // ... staged through pinned memory in pieces of this size
constexpr size_t ZB_CARRY = (size_t)4 << 20;                 // longest line end-less tail carried to the next batch
constexpr size_t ZB_TAIL = (size_t)1 << 20;                  // bytes of a batch's end looked at for its last line end

// count_bgzf_gpu's / td_bgzf_inflate_range's batch buffers (kept on the handle between files; grown when a larger
// file comes), the CRC tables, the kernel's LDS attribute.  *ok = false: no room on the device.
int ensure_bgzf_buffers(td_handle *h, uint32_t need_members, size_t need_in, bool *ok_out) {
    *ok_out = true;
    if (need_members > h->zcap_members || need_in > h->zcap_in) {
        need_members = std::max(need_members, h->zcap_members);
        need_in = std::max(need_in, h->zcap_in);
        release_bgzf_buffers(h);
        const size_t out_cap = (size_t)need_members * 65536 + ZB_CARRY + 4096;
        bool ok = true;
        auto dev = [&](void **p, size_t n) { if (ok && hipMalloc(p, n) != hipSuccess) { ok = false; (void)hipGetLastError(); } };
        auto pin = [&](void **p, size_t n) { if (ok && hipHostMalloc(p, n, hipHostMallocDefault) != hipSuccess) { ok = false; (void)hipGetLastError(); } };
        for (auto &zp : h->zpiece) {
            pin((void **)&zp.pin, std::min(ZB_PIECE, need_in + 64) + 64);
            if (ok && hipEventCreateWithFlags(&zp.sent, hipEventDisableTiming) != hipSuccess) ok = false;
        }
        for (auto &z : h->zslot) {
            dev((void **)&z.d_in, need_in + 1024);
            dev((void **)&z.d_out, out_cap);
            pin((void **)&z.pin_mem, (size_t)need_members * sizeof(tdinf::Member));
            dev((void **)&z.d_mem, (size_t)need_members * sizeof(tdinf::Member));
            dev((void **)&z.d_status, (size_t)need_members * 4);
            pin((void **)&z.pin_status, (size_t)need_members * 4);
            pin((void **)&z.pin_tail, ZB_TAIL);
            if (ok && hipEventCreateWithFlags(&z.copied, hipEventDisableTiming) != hipSuccess) ok = false;
        }
        dev((void **)&h->d_zscratch, (size_t)need_members * tdinf::SCRATCH_BYTES);
        if (!ok) { release_bgzf_buffers(h); *ok_out = false; return TD_OK; }
        h->zcap_members = need_members; h->zcap_in = need_in;
    }
    if (!h->d_crctab) {
        uint32_t T[1024];                                        // slicing-by-4: T[256 k + b] = CRC of byte b followed by k zero bytes
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; T[i] = c; }
        for (uint32_t k = 1; k < 4; k++)
            for (uint32_t i = 0; i < 256; i++) T[256 * k + i] = T[T[256 * (k - 1) + i] & 0xFFu] ^ (T[256 * (k - 1) + i] >> 8);
        HIPCHK(hipMalloc((void **)&h->d_crctab, sizeof(T)));
        HIPCHK(hipMemcpy(h->d_crctab, T, sizeof(T), hipMemcpyHostToDevice));
    }
    HIPCHK(hipFuncSetAttribute((const void *)tdinf::k_bgzf_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * tdinf::TABLE_U16 * 2 + 4096));
    return TD_OK;
}

int count_bgzf_gpu(td_handle *h, const char *path, uint64_t max_reads, int weights, bool *not_bgzf) {
    *not_bgzf = false;
    tdhost::GzSource src;
    if (!src.map_only(path)) { *not_bgzf = true; return TD_OK; }           // (empty or unreadable: the host path reports it)
    uint32_t bs0 = 0, hs0 = 0;
    if (!tdhost::GzSource::bgzf_header(src.map, src.bsize, &bs0, &hs0)) { *not_bgzf = true; return TD_OK; }
    // Batch buffers sized from the file: a small file is walked first (headers only), which also tells whether EVERY
    // member is BGZF -- gzip.open (reference :240-241) reads a file whose later members are plain gzip, so such a file
    // goes to the host inflater; a file beyond 1 GiB compressed takes the full batch (49 152 members, ~9.5 GB of HBM).
    uint32_t need_members = ZB_MEMBERS;
    size_t need_in = ZB_IN;
    if (src.bsize <= ((size_t)1 << 30)) {
        size_t at = 0;
        uint32_t n = 0;
        while (at < src.bsize) {
            uint32_t bs = 0, hs = 0;
            if (!tdhost::GzSource::bgzf_header(src.map + at, src.bsize - at, &bs, &hs) || bs < hs + 8 || at + bs > src.bsize) {
                *not_bgzf = true;                                  // (a damaged file too: the host path reports it)
                return TD_OK;
            }
            at += bs; n++;
        }
        need_members = std::min<uint32_t>(ZB_MEMBERS, std::max<uint32_t>(64, (n + 63) / 64 * 64));
        need_in = std::min<size_t>(ZB_IN, (src.bsize + 4095) / 4096 * 4096);
    }
    {
        bool ok = true;
        const int rc_b = ensure_bgzf_buffers(h, need_members, need_in, &ok);
        if (rc_b) return rc_b;
        if (!ok) { *not_bgzf = true; return TD_OK; }          // no room on the device: the host inflater takes the file
    }
    const uint32_t batch_members = std::min<uint32_t>(h->zcap_members, h->zb_members);
    const size_t batch_in = h->zcap_in, piece_cap = std::min(ZB_PIECE, h->zcap_in + 64);
    HIPCHK(hipMemsetAsync(h->d_cursor.p, 0, 16, h->work_stream));
    struct Batch { uint32_t n = 0; size_t out_total = 0; bool last = false; };
    size_t pos = 0;
    // members of the next batch: descriptors and compressed bytes into the slot's pinned buffers, then on their way to the device
    auto prepare = [&](int slot, Batch &b) -> int {
        td_handle::ZSlot &z = h->zslot[slot];
        b = Batch();
        const size_t first = pos;
        while (pos < src.bsize && b.n < batch_members) {
            uint32_t bs = 0, hs = 0;
            if (!tdhost::GzSource::bgzf_header(src.map + pos, src.bsize - pos, &bs, &hs) || bs < hs + 8 || pos + bs > src.bsize) {
                if (tdhost::GzSource::only_zeros(src.map + pos, src.bsize - pos)) { pos = src.bsize; break; }     // (padding: gzip.open skips it)
                return fail(TD_E_IO, "damaged BGZF member header (or a member that is not BGZF) in a file that began as BGZF");
            }
            if (pos + bs - first > batch_in) break;
            const uint8_t *tail = src.map + pos + bs - 8;
            const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
            if (isize > 65536) return fail(TD_E_IO, "BGZF member larger than 64 KiB");
            z.pin_mem[b.n] = tdinf::Member{pos + hs - first, b.out_total, bs - hs - 8, isize, crc, 0};
            b.out_total += isize; b.n++;
            pos += bs;
        }
        b.last = pos >= src.bsize;
        const size_t nin = pos - first;
        // the compressed bytes, through the two pinned pieces in turn (+ 64 zero bytes: the decoder reads a little ahead)
        int k = 0;
        for (size_t off = 0; off < nin + 64; off += piece_cap, k ^= 1) {
            td_handle::ZPiece &zp = h->zpiece[k];
            if (zp.busy) { HIPCHK(hipEventSynchronize(zp.sent)); zp.busy = false; }
            const size_t want = std::min(piece_cap, nin + 64 - off), have = off < nin ? std::min(want, nin - off) : 0;
            if (have) stage_parallel(have, [&](size_t o2, size_t len) { memcpy(zp.pin + o2, src.map + first + off + o2, len); return true; });
            if (want > have) memset(zp.pin + have, 0, want - have);
            { const int urc = upload(h, z.d_in + off, zp.pin, want); if (urc) return urc; }
            HIPCHK(hipEventRecord(zp.sent, h->copy_stream));
            zp.busy = true;
        }
        if (b.n) HIPCHK(hipMemcpyAsync(z.d_mem, z.pin_mem, (size_t)b.n * sizeof(tdinf::Member), hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(hipEventRecord(z.copied, h->copy_stream));
        return TD_OK;
    };
    Batch cur, nxt;
    int slot = 0;
    // (reference :272: the loop ends at maxreads -- no further batch is inflated once a counted batch's line index shows
    // that the bound has been passed; the kernels ignore reads past it either way)
    if (!h->pin_cursor) HIPCHK(hipHostMalloc((void **)&h->pin_cursor, 16, hipHostMallocDefault));
    h->pin_cursor[0] = 0;
    const uint64_t stop_line = max_reads >= (1ull << 60) ? ~0ull : 4 * (std::max<uint64_t>(1, max_reads) - 1) + 2;
    int rc = prepare(0, cur); if (rc) return rc;
    size_t carry = 0;                                       // bytes of an unfinished line at the front of this slot's output
    uint64_t bytes_submitted = 0;
    unsigned pieces = 0;
    for (;;) {
        td_handle::ZSlot &z = h->zslot[slot];
        HIPCHK(hipStreamWaitEvent(h->work_stream, z.copied, 0));
        if (cur.n) {
            hipLaunchKernelGGL(tdinf::k_bgzf_inflate, dim3((cur.n + 63) / 64), dim3(64), 64 * tdinf::TABLE_U16 * 2 + 4096, h->work_stream,
                               z.d_in, z.d_out + carry, z.d_mem, cur.n, h->d_zscratch, z.d_status, h->d_crctab, (uint32_t)h->gpu_inflate_crc);
            HIPCHK(hipGetLastError());
        }
        const size_t total = carry + cur.out_total;
        const size_t ntail = std::min(total, ZB_TAIL);
        if (cur.n) HIPCHK(hipMemcpyAsync(z.pin_status, z.d_status, (size_t)cur.n * 4, hipMemcpyDeviceToHost, h->work_stream));
        if (ntail && !cur.last) HIPCHK(hipMemcpyAsync(z.pin_tail, z.d_out + total - ntail, ntail, hipMemcpyDeviceToHost, h->work_stream));
        // the next batch is read and sent while this one inflates
        if (!cur.last) { rc = prepare(slot ^ 1, nxt); if (rc) return rc; }
        HIPCHK(hipStreamSynchronize(h->work_stream));
        if (h->pin_cursor[0] >= stop_line) break;               // (the batches counted so far already hold read number max_reads)
        for (uint32_t i = 0; i < cur.n; i++)
            if (z.pin_status[i]) return fail(TD_E_IO, z.pin_status[i] == 100 ? "BGZF member fails its CRC-32" : "inflate error in a BGZF member");
        size_t cut = total;
        if (!cur.last && total) {
            const size_t c = cut_at_line_end(z.pin_tail, ntail);
            if (c == 0) {                                       // (no line end in sight: everything waits for the next batch)
                if (total > ZB_CARRY) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
                cut = 0;
            } else {
                if (ntail - c > ZB_CARRY) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
                cut = total - ntail + c;
            }
        }
        if (cut) {
            rc = launch_count(h, z.d_out, cut, 0, max_reads, weights, h->work_stream, h->d_cursor.p + (pieces & 1),
                              h->d_cursor.p + ((pieces + 1) & 1), bytes_submitted);
            if (rc) return rc;
            bytes_submitted += cut; pieces++;
            HIPCHK(hipMemcpyAsync(h->pin_cursor, h->d_cursor.p + (pieces & 1), 8, hipMemcpyDeviceToHost, h->work_stream));
        }
        if (cur.last) break;
        carry = total - cut;
        if (carry) HIPCHK(hipMemcpyAsync(h->zslot[slot ^ 1].d_out, z.d_out + cut, carry, hipMemcpyDeviceToDevice, h->work_stream));
        cur = nxt; slot ^= 1;
    }
    HIPCHK(hipStreamSynchronize(h->work_stream));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    return TD_OK;
}
}  // namespace

// ---- ordinary (one-stream) gzip: steps 1-3 of the chunk-parallel decoder on the host threads (block-start search,
// symbolic decode, chain + windows: par_inflate.hpp), step 4 -- markers into bytes, CRC-32 -- on the GPU (gz_resolve.hpp).
// The symbols are decoded straight into pinned memory (the decoder's chunk buffers come from the pool below) and go to
// the device by DMA: no resolve pass, no CRC pass and no copy into a staging buffer on the 16 host threads that bound
// this tier.  *not_applicable: the file is not for this path (BGZF, small, zlib fallback requested): nothing was read.
namespace {
// pinned blocks for the decoder's chunk buffers, kept between files (pinning costs ~0.3 ms per MiB)
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> free_blocks;
    std::vector<void *> unpinned;
    size_t held = 0;
    static constexpr size_t KEEP = (size_t)3 << 30;
    void *alloc(size_t n) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_blocks.size(); i++)
                if (free_blocks[i].second >= n && free_blocks[i].second <= 2 * n) {
                    void *p = free_blocks[i].first;
                    held -= free_blocks[i].second;
                    free_blocks.erase(free_blocks.begin() + (long)i);
                    return p;
                }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, n, hipHostMallocPortable) == hipSuccess) return p;
        // (no more memory can be pinned -- a locked-memory limit, say: ordinary pages then, which the copies stage themselves)
        (void)hipGetLastError();
        p = aligned_alloc(4096, (n + 4095) & ~(size_t)4095);
        if (p) { std::lock_guard<std::mutex> g(mu); unpinned.push_back(p); }
        return p;
    }
    void trim() {                                // (blocks a decoder still holds are not in the list)
        std::vector<std::pair<void *, size_t>> idle;
        { std::lock_guard<std::mutex> g(mu); idle.swap(free_blocks); held = 0; }
        for (auto &b : idle) (void)hipHostFree(b.first);
    }
    void release(void *p, size_t n) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < unpinned.size(); i++)
                if (unpinned[i] == p) { unpinned.erase(unpinned.begin() + (long)i); free(p); return; }
            // (blocks come back with the size they were asked for: the decoder rounds to 2 MiB both times)
            if (held + n <= KEEP) { free_blocks.emplace_back(p, n); held += n; return; }
        }
        (void)hipHostFree(p);
    }
};
PinnedPool g_pinned;
std::atomic<int> g_handles{0};                   // (the last handle to go gives the pool's idle blocks back: td_destroy)
void handle_born() { g_handles++; }
void handle_gone() { if (--g_handles == 0) g_pinned.trim(); }
const tdhost::ParInflate::Allocator g_pinned_alloc = {
    [](size_t n) -> void * { return g_pinned.alloc(n); },
    [](void *p, size_t n) { g_pinned.release(p, n); }};

int count_gzip_dev(td_handle *h, const char *path, uint64_t max_reads, int weights, bool *not_applicable) {
    using PI = tdhost::ParInflate;
    *not_applicable = false;
    tdhost::GzSource src;
    if (!src.open_dev(path, &g_pinned_alloc)) { *not_applicable = true; return TD_OK; }
    int rc = h->d_gzflag.ensure(4); if (rc) return rc;
    HIPCHK(hipMemsetAsync(h->d_gzflag.p, 0, 16, h->work_stream));
    if (!h->d_crctab) { bool ok = true; rc = ensure_bgzf_buffers(h, 0, 0, &ok); if (rc) return rc; }      // (the CRC tables)
    HIPCHK(hipMemsetAsync(h->d_cursor.p, 0, 16, h->work_stream));
    if (!h->pin_cursor) HIPCHK(hipHostMalloc((void **)&h->pin_cursor, 16, hipHostMallocDefault));
    h->pin_cursor[0] = 0;
    // (events the host waits on sleeping: a spinning wait would take a core from the decoder's threads)
    for (auto &g : h->gslot) {
        if (!g.copied) HIPCHK(hipEventCreateWithFlags(&g.copied, hipEventDisableTiming | hipEventBlockingSync));
        if (!g.done) HIPCHK(hipEventCreateWithFlags(&g.done, hipEventDisableTiming | hipEventBlockingSync));
    }
    const int copy_streams = copy_lanes();
    struct Batch {
        bool valid = false, last = false, member_done = false;
        uint32_t want_crc = 0, nblk = 0;
        size_t total = 0;
        std::vector<std::pair<uint32_t, size_t>> crc_len;       // filled from pin_crc after the batch has been resolved
    };
    const uint64_t stop_line = max_reads >= (1ull << 60) ? ~0ull : 4 * (std::max<uint64_t>(1, max_reads) - 1) + 2;
    size_t carry = 0;                                           // bytes of an unfinished line at the front of the current slot's output
    // the next batch from the decoder: block table, symbols and windows on their way to the device; the decoder's buffers
    // are handed back as soon as the copies have been made
    double t_next = 0, t_upload = 0, t_sync = 0;                // (TAGDIG_INFLATE_STATS: waiting for the decoder / the copies / the kernels)
    uint64_t up_bytes = 0, nbatch = 0;
    auto prepare = [&](int slot, Batch &b) -> int {
        td_handle::GSlot &g = h->gslot[slot];
        b = Batch();
        const double tn = PI::now();
        const PI::DevBatch *db = src.pi.dev_next();
        t_next += PI::now() - tn;
        if (!db) {
            if (src.pi.dev_failed()) return fail(TD_E_IO, std::string("gzip: ") + src.pi.error());
            return TD_OK;                                       // (the stream is through)
        }
        if (db->failed) { src.pi.dev_release(); return fail(TD_E_IO, std::string("gzip: ") + src.pi.error()); }
        // blocks of 64 Ki symbols; the windows of the batch's chunks (pieces of one chunk share theirs)
        size_t nblk = 0, sym_bytes = 0;
        for (const PI::DevPiece &pc : db->pieces) { nblk += (pc.len + tdgz::BLOCK_SYMS - 1) / tdgz::BLOCK_SYMS; sym_bytes += (pc.len * (pc.narrow ? 1 : 2) + 15) & ~(size_t)15; }
        std::vector<const uint8_t *> wins;
        if (nblk > g.pin_cap) {
            if (g.pin_blk) (void)hipHostFree(g.pin_blk);
            if (g.pin_crc) (void)hipHostFree(g.pin_crc);
            g.pin_blk = nullptr; g.pin_crc = nullptr;
            const size_t cap = nblk * 2 + 1024;
            HIPCHK(hipHostMalloc((void **)&g.pin_blk, cap * sizeof(tdgz::Block), hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&g.pin_crc, cap * 4, hipHostMallocDefault));
            g.pin_cap = cap;
        }
        if (!g.pin_tail) HIPCHK(hipHostMalloc((void **)&g.pin_tail, ZB_TAIL + 16, hipHostMallocDefault));     // (+ the marker flag)
        int rc2 = g.d_sym.ensure(sym_bytes + 64); if (rc2) return rc2;
        rc2 = g.d_out.ensure(ZB_CARRY + db->total + 4096 + (db->total >> 3)); if (rc2) return rc2;
        rc2 = g.d_blk.ensure(nblk + 1); if (rc2) return rc2;
        rc2 = g.d_crc.ensure(nblk + 1); if (rc2) return rc2;
        size_t at = 0, kb = 0;
        unsigned npiece = 0;
        for (const PI::DevPiece &pc : db->pieces) {
            uint32_t wi = 0;
            while (wi < wins.size() && wins[wi] != pc.window) wi++;
            if (wi == wins.size()) wins.push_back(pc.window);
            const size_t esz = pc.narrow ? 1 : 2;
            const int lane = (int)(npiece++ % (unsigned)copy_streams);           // (the chunks of a batch over the copy engines)
            HIPCHK(hipMemcpyAsync(g.d_sym.p + at, pc.src, pc.len * esz, hipMemcpyHostToDevice, lane ? h->side_copy[lane - 1] : h->copy_stream));
            for (size_t o = 0; o < pc.len; o += tdgz::BLOCK_SYMS, kb++)
                g.pin_blk[kb] = tdgz::Block{at + o * esz, pc.dest_off + o, (uint32_t)std::min<size_t>(tdgz::BLOCK_SYMS, pc.len - o), wi,
                                            pc.min_idx, pc.narrow ? 1u : 0u};
            at += (pc.len * esz + 15) & ~(size_t)15;
        }
        rc2 = g.d_win.ensure(std::max<size_t>(1, wins.size()) * tdgz::WINDOW); if (rc2) return rc2;
        for (size_t w = 0; w < wins.size(); w++)
            HIPCHK(hipMemcpyAsync(g.d_win.p + w * tdgz::WINDOW, wins[w], tdgz::WINDOW, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(hipMemcpyAsync(g.d_blk.p, g.pin_blk, nblk * sizeof(tdgz::Block), hipMemcpyHostToDevice, h->copy_stream));
        for (int k = 0; k + 1 < copy_streams; k++) {
            HIPCHK(hipEventRecord(h->side_done[k], h->side_copy[k]));
            HIPCHK(hipStreamWaitEvent(h->copy_stream, h->side_done[k], 0));
        }
        HIPCHK(hipEventRecord(g.copied, h->copy_stream));
        b.valid = true; b.last = db->last; b.member_done = db->member_done; b.want_crc = db->want_crc;
        b.nblk = (uint32_t)nblk; b.total = db->total;
        b.crc_len.resize(nblk);
        for (size_t k = 0; k < nblk; k++) b.crc_len[k].second = g.pin_blk[k].len;
        // the decoder may have its buffers back once the copies have left them
        const double tu = PI::now();
        HIPCHK(hipEventSynchronize(g.copied));
        t_upload += PI::now() - tu; up_bytes += at;
        src.pi.dev_release();
        return TD_OK;
    };
    Batch cur, nxt;
    int slot = 0;
    rc = prepare(0, cur); if (rc) return rc;
    uint64_t bytes_submitted = 0;
    unsigned pieces = 0;
    while (cur.valid) {
        td_handle::GSlot &g = h->gslot[slot];
        HIPCHK(hipStreamWaitEvent(h->work_stream, g.copied, 0));
        if (cur.nblk) {
            // (the batch's bytes go behind what the batch before left of its last line: d_out + carry)
            hipLaunchKernelGGL(tdgz::k_gz_resolve, dim3(cur.nblk), dim3(256), 0, h->work_stream, g.d_sym.p, g.d_win.p, g.d_out.p + carry, g.d_blk.p, cur.nblk, h->d_gzflag.p);
            hipLaunchKernelGGL(tdgz::k_gz_crc, dim3((cur.nblk + 63) / 64), dim3(64), 0, h->work_stream, g.d_out.p + carry, g.d_blk.p, cur.nblk, h->d_crctab, g.d_crc.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(g.pin_crc, g.d_crc.p, (size_t)cur.nblk * 4, hipMemcpyDeviceToHost, h->work_stream));
        }
        const size_t total = carry + cur.total;
        const size_t ntail = std::min(total, ZB_TAIL);
        if (ntail && !cur.last) HIPCHK(hipMemcpyAsync(g.pin_tail, g.d_out.p + total - ntail, ntail, hipMemcpyDeviceToHost, h->work_stream));
        uint32_t *flag = (uint32_t *)(g.pin_tail + ZB_TAIL);
        HIPCHK(hipMemcpyAsync(flag, h->d_gzflag.p, 4, hipMemcpyDeviceToHost, h->work_stream));
        HIPCHK(hipEventRecord(g.done, h->work_stream));
        // the next batch is taken from the decoder and sent while this one is resolved
        if (!cur.last) { rc = prepare(slot ^ 1, nxt); if (rc) return rc; }
        const double ts = PI::now();
        HIPCHK(hipEventSynchronize(g.done));
        t_sync += PI::now() - ts; nbatch++;
        if (*flag) return fail(TD_E_IO, "gzip: distance reaches before the start of the output");
        for (uint32_t k = 0; k < cur.nblk; k++) cur.crc_len[k].first = g.pin_crc[k];
        if (!src.pi.dev_check(cur.crc_len, cur.member_done, cur.want_crc)) return fail(TD_E_IO, std::string("gzip: ") + src.pi.error());
        if (h->pin_cursor[0] >= stop_line) break;               // (the batches counted so far already hold read number max_reads)
        size_t cut = total;
        if (!cur.last && total) {
            const size_t c = cut_at_line_end(g.pin_tail, ntail);
            if (c == 0) {
                // no line end in sight -- a member that stops inside a line, followed by an empty one or one of a few bytes
                // (a batch closes at every member end): everything waits for the next batch
                if (total > ZB_CARRY) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
                cut = 0;
            } else {
                if (ntail - c > ZB_CARRY) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
                cut = total - ntail + c;
            }
        }
        if (cut) {
            rc = launch_count(h, g.d_out.p, cut, 0, max_reads, weights, h->work_stream, h->d_cursor.p + (pieces & 1),
                              h->d_cursor.p + ((pieces + 1) & 1), bytes_submitted);
            if (rc) return rc;
            bytes_submitted += cut; pieces++;
            HIPCHK(hipMemcpyAsync(h->pin_cursor, h->d_cursor.p + (pieces & 1), 8, hipMemcpyDeviceToHost, h->work_stream));
        }
        if (cur.last) break;
        const size_t carry_next = total - cut;
        if (carry_next) HIPCHK(hipMemcpyAsync(h->gslot[slot ^ 1].d_out.p, g.d_out.p + cut, carry_next, hipMemcpyDeviceToDevice, h->work_stream));
        carry = carry_next;
        cur = nxt; slot ^= 1;
    }
    HIPCHK(hipStreamSynchronize(h->work_stream));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    if (getenv("TAGDIG_INFLATE_STATS"))
        fprintf(stderr, "count_gzip_dev: %lu batches, %.1f MB sent to the GPU; waiting for the decoder %.3f s, for the copies %.3f s, for the kernels %.3f s\n",
                (unsigned long)nbatch, up_bytes / 1e6, t_next, t_upload, t_sync);
    return TD_OK;
}

struct MappedFile {
    const uint8_t *p = nullptr; size_t n = 0; bool ok = false;
    explicit MappedFile(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); return; }
        n = (size_t)sb.st_size;
        ok = true;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ok = false; n = 0; } else { p = (const uint8_t *)m; (void)madvise(m, n, MADV_SEQUENTIAL); }
        }
        ::close(fd);
    }
    ~MappedFile() { if (p) munmap(const_cast<uint8_t *>(p), n); }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

// ---- ordinary gzip decoded on the GPU (gz_gpu.hpp): the compressed file goes to the device as it is; block starts are
// searched, the chunks between them Huffman-decoded into tokens, chained here, turned into symbols and bytes, checked
// against the member's CRC-32 and length, and counted where they lie.  *not_applicable: nothing has been counted and the
// host decoder (count_gzip_dev) should take the file -- too small, no room on the device, a second member, or a stream this
// decoder does not chain.
extern "C" int td_load_file_range(td_handle *h, const char *path, uint64_t offset, uint64_t length, void *d_dst);
// The file goes through the device in SEGMENTS of compressed bytes (1 GiB; what the buffers are sized for): a segment's first
// chunk starts exactly where the segment before ended -- a block boundary, or a member's first block -- its other chunks
// where k_gz_find finds block starts, its last chunk ends at the first block boundary at or past the segment's end (the
// upload reaches 16 MiB further), and the 32 KiB window behind it stays on the device for the next segment (d_carry).  A
// member's end closes a segment: its CRC-32 and length are checked, the next member starts with an empty window.
struct GzGpuStream {
    td_handle *h = nullptr;
    const char *path = nullptr;
    MappedFile mf;
    bool verbose = false;
    uint64_t n = 0;
    uint64_t next_bit = 0;            // where the next segment's first chunk starts (absolute bit of the file)
    uint64_t member_out = 0;          // bytes the current member has inflated to so far
    uint32_t crc_run = 0;
    uint64_t segments = 0;
    bool file_done = false, member_fresh = true;
    const char *why = "";             // (give-ups)
    double t_up = 0, t_find = 0, t_tok = 0, t_rest = 0;
    // the next segment's bytes on their way while this one is decoded (the other of two buffers, a thread of its own: the
    // loader stages through pinned pieces on the calling thread)
    std::thread pf_thread;
    bool pf_active = false, cur_alt = false;
    uint64_t pf_base = 0, pf_nb = 0, pf_cap = 0;
    uint8_t *pf_buf = nullptr;
    int pf_rc = 0;
    ~GzGpuStream() { if (pf_thread.joinable()) pf_thread.join(); }
    uint64_t SEG = (uint64_t)1 << 30, MARGIN = (uint64_t)16 << 20;       // (options gz_gpu_seg_kb, gz_gpu_margin_kb)

    explicit GzGpuStream(td_handle *hh, const char *p) : h(hh), path(p), mf(p) {
        verbose = getenv("TAGDIG_INFLATE_STATS") != nullptr;
        SEG = (uint64_t)hh->gz_gpu_seg_kb << 10; MARGIN = (uint64_t)hh->gz_gpu_margin_kb << 10;
    }

    // false: this file is not for the device decoder (why says why); nothing has been touched
    bool open() {
        using FI = tdhost::FastInflate;
        if (!mf.ok || mf.n < h->gz_gpu_min || mf.n < 64) { why = "small file"; return false; }
        n = mf.n;
        if (n >= ((uint64_t)1 << 44)) { why = "file too large"; return false; }
        { uint32_t bs = 0, hs = 0; if (tdhost::GzSource::bgzf_header(mf.p, mf.n, &bs, &hs)) { why = "a BGZF file (many members)"; return false; } }
        const uint8_t *body = nullptr; const char *herr = nullptr;
        if (FI::parse_member_header(mf.p, mf.p + mf.n, &body, &herr, true) != 1) { why = "no gzip header"; return false; }
        next_bit = (uint64_t)(body - mf.p) * 8;
        // room: a segment's compressed bytes, its tokens (16 bytes per compressed byte) and -- FASTQ deflates to a fifth -- its
        // symbols, text and windows (checked again when they are known)
        const uint64_t seg = std::min<uint64_t>(n, SEG + MARGIN);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); why = "no device"; return false; }
        const td_handle::GzGpu &g = h->gzgpu;
        const uint64_t have = free_b + g.d_in.n + g.d_tok.n * 4 + g.d_sym.n * 2 + g.d_out.n + g.d_win.n;
        if (seg * 34 + ((uint64_t)1 << 30) > have) { why = "no room on the device"; return false; }
        return true;
    }

    // The next segment's text behind `carry` bytes at the front of gzgpu.d_out.  Returns TD_OK and *gave_up = false: *nbytes
    // of text, *last says whether the file is through; *gave_up = true: the decoder leaves the file (why); else an error code.
    int next(size_t carry, uint64_t *nbytes, bool *last, bool *gave_up) {
        using FI = tdhost::FastInflate;
        using PI = tdhost::ParInflate;
        *gave_up = false; *nbytes = 0; *last = false;
        auto giveup = [&](const char *w) { why = w; *gave_up = true; if (verbose) fprintf(stderr, "gz_gpu_inflate: %s (segment %lu)\n", w, (unsigned long)segments); return TD_OK; };
        td_handle::GzGpu &g = h->gzgpu;
        hipStream_t st = h->work_stream;
        // (an allocation that fails -- another process on the card -- is a reason to leave the file to the host decoder, not an error)
        auto no_room = [&]() { (void)hipGetLastError(); return giveup("no room on the device"); };
        const double t0 = PI::now();
        // the segment's bytes
        const uint64_t base = (next_bit >> 3) & ~(uint64_t)4095;
        const uint64_t seg_end = std::min<uint64_t>(n, base + SEG), up_end = std::min<uint64_t>(n, seg_end + MARGIN);
        const uint64_t nb = up_end - base;
        const bool to_file_end = seg_end == n;
        const size_t buf_bytes = std::min<uint64_t>(n, SEG + 2 * MARGIN) + 8192 + 4096;
        int rc = g.d_in.ensure(buf_bytes); if (rc) return no_room();
        HIPCHK(hipStreamSynchronize(st));                                   // (the segment before may still read its bytes)
        const uint8_t *din = nullptr;
        size_t in_cap = 0;
        if (pf_active) {
            pf_thread.join();
            pf_active = false;
            if (pf_rc == 0 && pf_base <= base && base + nb <= pf_base + pf_nb) {
                din = pf_buf + (base - pf_base);
                in_cap = (size_t)(pf_base + pf_cap - base) & ~(size_t)15;
                cur_alt = pf_buf == g.d_in2.p;
            }
        }
        if (!din) {
            in_cap = ((nb + 4096 + 15) & ~(size_t)15);
            HIPCHK(hipMemsetAsync(g.d_in.p + (nb & ~(size_t)15), 0, in_cap - (nb & ~(size_t)15), h->copy_stream));
            HIPCHK(hipStreamSynchronize(h->copy_stream));
            rc = td_load_file_range(h, path, base, nb, g.d_in.p); if (rc) return rc;
            din = g.d_in.p; cur_alt = false;
        }
        const uint64_t nwords = in_cap / 4, in_bits = nb * 8, first_bit = next_bit - base * 8;
        const uint64_t stop_seg = to_file_end ? ~0ull : (seg_end - base) * 8;
        const double t1 = PI::now();
        // 1. block starts
        const uint64_t terr = (uint64_t)h->gz_gpu_terr_kb << 10;
        const uint32_t nterr = (uint32_t)((seg_end - base + terr - 1) / terr);
        rc = g.d_found.ensure(nterr + 4); if (rc) return no_room();
        HIPCHK(hipMemsetAsync(g.d_found.p + nterr, 0, 32, st));
        if (nterr > 1)
            hipLaunchKernelGGL(tdgz2::k_gz_find, dim3((nterr - 1 + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st,
                               din, in_bits, nwords, first_bit, terr * 8, nterr, g.d_found.p, (uint32_t)h->gz_gpu_verify, 1u);
        HIPCHK(hipGetLastError());
        std::vector<uint64_t> found(nterr);
        if (nterr > 1) HIPCHK(hipMemcpyAsync(found.data() + 1, g.d_found.p + 1, (size_t)(nterr - 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        found[0] = tdgz2::NONE;
        if (h->gz_gpu_false_every > 0)
            for (uint32_t t = 1; t < nterr; t++)
                if (found[t] != tdgz2::NONE && t % (uint32_t)h->gz_gpu_false_every == 0 && found[t] + 4099 < in_bits) found[t] += 4099;
        const double t2 = PI::now();
        // 2. the chunks between them, decoded into tokens
        std::vector<tdgz2::Chunk> chunks;
        { tdgz2::Chunk c{}; c.start_bit = first_bit; chunks.push_back(c); }
        for (uint32_t t = 0; t < nterr; t++) if (found[t] != tdgz2::NONE && found[t] > first_bit) { tdgz2::Chunk c{}; c.start_bit = found[t]; chunks.push_back(c); }
        uint32_t nchunks = (uint32_t)chunks.size();
        {
            uint64_t at = 0;
            for (uint32_t i = 0; i < nchunks; i++) {
                chunks[i].stop_bit = i + 1 < nchunks ? chunks[i + 1].start_bit : stop_seg;
                // (the last chunk runs to the first block boundary behind the segment's end: room for a block of the margin's size)
                const uint64_t span_end = i + 1 < nchunks ? chunks[i + 1].start_bit : in_bits;
                const uint64_t span = (span_end - chunks[i].start_bit + 7) / 8;
                const uint64_t cap = std::min<uint64_t>(4 * span + 4096, 0xFFFFFF00u);
                chunks[i].tok_off = at; chunks[i].tok_cap = (uint32_t)cap;
                at += (cap + 63) & ~(uint64_t)63;
            }
            rc = g.d_tok.ensure(at + 64); if (rc) return no_room();
        }
        rc = g.d_chunks.ensure(2 * (size_t)nchunks); if (rc) return no_room();               // (behind the chunks: the ones decoded a second time)
        rc = g.d_res.ensure(2 * (size_t)nchunks); if (rc) return no_room();
        HIPCHK(hipMemcpyAsync(g.d_chunks.p, chunks.data(), (size_t)nchunks * sizeof(tdgz2::Chunk), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(g.d_res.p, 0, (size_t)nchunks * sizeof(tdgz2::ChunkOut), st));
        hipLaunchKernelGGL(tdgz2::k_gz_tokens, dim3((nchunks + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st,
                           din, in_bits, nwords, g.d_chunks.p, nchunks, g.d_tok.p, g.d_res.p);
        HIPCHK(hipGetLastError());
        std::vector<tdgz2::ChunkOut> res(nchunks);
        HIPCHK(hipMemcpyAsync(res.data(), g.d_res.p, (size_t)nchunks * sizeof(tdgz2::ChunkOut), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double t3 = PI::now();
        // 3. the chain: every chunk begins where its predecessor ended.  A chunk that ran past its successor's start met a false
        // one there: the successor is dropped, and what lies between this chunk's end and the next start is decoded in a second
        // launch (into the dropped chunk's token buffer).  A chunk that ends its member ends the segment.
        std::vector<uint8_t> dead(nchunks, 0);
        bool member_done = false;
        for (int round = 0;; round++) {
            std::vector<uint32_t> redo;
            uint32_t i = 0;
            bool assumed = false;                 // behind this round's first gap the chain is a guess: nothing there refuses the file yet
            member_done = false;
            while (true) {
                const tdgz2::ChunkOut &o = res[i];
                if (assumed && (o.status != tdgz2::S_BOUNDARY && o.status != tdgz2::S_FINAL)) break;
                if (o.status == tdgz2::S_TOKCAP) return giveup("a chunk's tokens overflow their buffer");
                if (o.status == tdgz2::S_UNUSUAL) return giveup("a Huffman code this decoder leaves to zlib");
                if (o.status == tdgz2::S_ERR) return giveup("invalid DEFLATE data (or a block longer than the segments' overlap)");
                if (o.status == tdgz2::S_FINAL) {
                    if (assumed) break;
                    for (uint32_t k = i + 1; k < nchunks; k++) dead[k] = 1;      // (what the search found behind the member's end is not this member's)
                    member_done = true;
                    break;
                }
                uint32_t j = i + 1;
                while (j < nchunks && dead[j]) j++;
                if (j == nchunks) {
                    if (assumed) break;
                    if (to_file_end) return giveup("the stream does not end with the file");
                    if (o.end_bit < stop_seg) return giveup("a chunk ends before the segment does");
                    break;
                }
                if (o.end_bit == chunks[j].start_bit) { i = j; continue; }
                // chunks[j] is no block start (the chunk in front of it was decoded up to a boundary at or past it)
                if (o.end_bit < chunks[j].start_bit) { if (assumed) break; return giveup("a chunk ends before its successor's start"); }
                if (assumed) break;               // (a guess must not drop chunks)
                uint32_t k = j + 1;
                while (k < nchunks && (dead[k] || chunks[k].start_bit < o.end_bit)) { dead[k] = 1; k++; }
                if (k < nchunks && chunks[k].start_bit == o.end_bit) { dead[j] = 1; i = k; continue; }
                if (k == nchunks && !to_file_end && o.end_bit >= stop_seg) { dead[j] = 1; break; }      // (it ran past the segment's end: the segment's last chunk)
                // the gap [end, next start): chunk j's place and buffer
                const uint64_t gap_end = k < nchunks ? chunks[k].start_bit : in_bits;
                if ((gap_end - o.end_bit + 7) / 8 * 4 + 4096 > chunks[j].tok_cap) return giveup("a false block start in front of a long stretch without one");
                chunks[j].start_bit = o.end_bit;
                chunks[j].stop_bit = k < nchunks ? chunks[k].start_bit : stop_seg;
                redo.push_back(j);
                // (the walk goes on behind the gap as if its decoding will end on the next start; the next round looks at that)
                if (k == nchunks) break;
                i = k;
                assumed = true;
            }
            if (redo.empty()) break;
            if (round >= 8) return giveup("too many false block starts");
            // the gaps of this round in one launch (a false start is rare: the header checks pass for about one position in 10^10)
            const uint32_t nr = (uint32_t)redo.size();
            std::vector<tdgz2::Chunk> rc_in(nr);
            std::vector<tdgz2::ChunkOut> rc_out(nr);
            for (uint32_t q = 0; q < nr; q++) rc_in[q] = chunks[redo[q]];
            HIPCHK(hipMemcpyAsync(g.d_chunks.p + nchunks, rc_in.data(), (size_t)nr * sizeof(tdgz2::Chunk), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(tdgz2::k_gz_tokens, dim3((nr + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st, din, in_bits, nwords,
                               g.d_chunks.p + nchunks, nr, g.d_tok.p, g.d_res.p + nchunks);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(rc_out.data(), g.d_res.p + nchunks, (size_t)nr * sizeof(tdgz2::ChunkOut), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (uint32_t q = 0; q < nr; q++) res[redo[q]] = rc_out[q];
            if (verbose) fprintf(stderr, "gz_gpu_inflate: %u false block starts; the stretches behind them decoded again\n", nr);
        }
        // the chunks that count, compacted (the kernels behind this walk them by index)
        {
            uint32_t w = 0;
            for (uint32_t i = 0; i < nchunks; i++) if (!dead[i]) { chunks[w] = chunks[i]; res[w] = res[i]; w++; }
            chunks.resize(w); res.resize(w);
            nchunks = w;
            HIPCHK(hipMemcpyAsync(g.d_chunks.p, chunks.data(), (size_t)nchunks * sizeof(tdgz2::Chunk), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(g.d_res.p, res.data(), (size_t)nchunks * sizeof(tdgz2::ChunkOut), hipMemcpyHostToDevice, st));
        }
        std::vector<uint64_t> sym_off(nchunks + 1);
        uint64_t total = 0;
        for (uint32_t i = 0; i < nchunks; i++) { sym_off[i] = total; total += res[i].out_len; }
        sym_off[nchunks] = total;
        const uint64_t end_abs = base * 8 + res[nchunks - 1].end_bit;
        // a member's end: its trailer, and what follows it
        uint32_t want_crc = 0, want_len = 0;
        bool next_member = false;
        uint64_t next_member_bit = 0;
        if (member_done) {
            const uint64_t trailer = (end_abs + 7) / 8;
            if (trailer + 8 > n) return giveup("truncated member");
            memcpy(&want_crc, mf.p + trailer, 4); memcpy(&want_len, mf.p + trailer + 4, 4);
            if (want_len != (uint32_t)(member_out + total)) return giveup("the member fails its length check");
            const uint8_t *body = nullptr; const char *herr = nullptr;
            const int r = FI::parse_member_header(mf.p + trailer + 8, mf.p + n, &body, &herr, false);
            if (r < 0) return giveup("bytes behind the member that are no gzip header");
            if (r == 1) {
                next_member = true; next_member_bit = (uint64_t)(body - mf.p) * 8;
                // (a file of many small members is the host decoder's: a segment each would be mostly launches)
                if (segments == 0 && trailer < ((uint64_t)1 << 20)) return giveup("small members");
            }
        }
        // 4.-6. symbols, windows, bytes, CRC-32
        const uint64_t nblk64 = [&]() { uint64_t k = 0; for (uint32_t i = 0; i < nchunks; i++) k += (res[i].out_len + tdgz::BLOCK_SYMS - 1) / tdgz::BLOCK_SYMS; return k; }();
        if (nblk64 >= 0x7FFFFFFFull) return giveup("segment too large");
        const uint32_t nblk = (uint32_t)nblk64;
        {
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            const uint64_t need = 2 * total + total + ZB_CARRY + (uint64_t)nchunks * tdgz2::WINDOW + (uint64_t)nblk * 40 + ((uint64_t)64 << 20);
            if (need > free_b + g.d_sym.n * 2 + g.d_out.n + g.d_win.n) return giveup("no room on the device for the text");
        }
        rc = g.d_sym.ensure(total + 64); if (rc) return no_room();
        if (g.d_out.n < ZB_CARRY + total + 4096 + (total >> 3)) {
            // (the carried bytes lie at the front of the old buffer)
            DevBuf<uint8_t> bigger;
            rc = bigger.ensure(ZB_CARRY + total + 4096 + (total >> 3) + (total >> 2)); if (rc) return no_room();
            if (carry) HIPCHK(hipMemcpyAsync(bigger.p, g.d_out.p, carry, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            g.d_out.release();
            g.d_out = bigger; bigger.p = nullptr; bigger.n = 0;
        }
        rc = g.d_win.ensure((size_t)nchunks * tdgz2::WINDOW); if (rc) return no_room();
        rc = g.d_carry.ensure(tdgz2::WINDOW); if (rc) return no_room();
        rc = g.d_symoff.ensure(nchunks + 1); if (rc) return no_room();
        rc = g.d_blk.ensure(nblk + 1); if (rc) return no_room();
        rc = g.d_crc.ensure(nblk + 1); if (rc) return no_room();
        rc = h->d_gzflag.ensure(4); if (rc) return no_room();
        if (!h->d_crctab) { bool ok = true; rc = ensure_bgzf_buffers(h, 0, 0, &ok); if (rc) return rc; }      // (the CRC tables)
        std::vector<tdgz::Block> blocks(nblk);
        {
            size_t kb = 0;
            for (uint32_t i = 0; i < nchunks; i++) {
                const uint32_t min_idx = tdgz2::WINDOW - (uint32_t)std::min<uint64_t>(tdgz2::WINDOW, member_out + sym_off[i]);
                for (uint64_t o = 0; o < res[i].out_len; o += tdgz::BLOCK_SYMS)
                    blocks[kb++] = tdgz::Block{(sym_off[i] + o) * 2, sym_off[i] + o, (uint32_t)std::min<uint64_t>(tdgz::BLOCK_SYMS, res[i].out_len - o), i, min_idx, 0u};
            }
        }
        HIPCHK(hipMemcpyAsync(g.d_symoff.p, sym_off.data(), (size_t)(nchunks + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(g.d_blk.p, blocks.data(), (size_t)nblk * sizeof(tdgz::Block), hipMemcpyHostToDevice, st));
        if (member_fresh) HIPCHK(hipMemsetAsync(g.d_carry.p, 0, tdgz2::WINDOW, st));           // (a member begins with nothing behind it)
        HIPCHK(hipMemsetAsync(h->d_gzflag.p, 0, 16, st));
        if (!to_file_end) {
            // the bytes behind this segment on their way while its symbols, windows, bytes and CRC-32 are made and its text is
            // counted: the next segment begins somewhere in the first MARGIN of them.  (Started here, behind this segment's
            // allocations: hipMalloc and hipFree wait for the copies in flight.)
            rc = g.d_in2.ensure(buf_bytes); if (rc) return no_room();
            pf_buf = cur_alt ? g.d_in.p : g.d_in2.p;
            pf_base = seg_end & ~(uint64_t)4095;
            pf_nb = std::min<uint64_t>(n, pf_base + SEG + 2 * MARGIN) - pf_base;
            pf_cap = (pf_nb + 4096 + 15) & ~(uint64_t)15;
            pf_rc = 0;
            pf_active = true;
            pf_thread = std::thread([this]() {
                if (hipSetDevice(h->device) != hipSuccess) { pf_rc = TD_E_INTERNAL; return; }
                if (hipMemsetAsync(pf_buf + (pf_nb & ~(uint64_t)15), 0, pf_cap - (pf_nb & ~(uint64_t)15), h->copy_stream) != hipSuccess ||
                    hipStreamSynchronize(h->copy_stream) != hipSuccess) { pf_rc = TD_E_INTERNAL; return; }
                pf_rc = td_load_file_range(h, path, pf_base, pf_nb, pf_buf);
            });
        }
        hipEvent_t ev[5] = {};
        auto mark = [&](int k) { if (verbose) { if (!ev[k]) (void)hipEventCreate(&ev[k]); (void)hipEventRecord(ev[k], st); } };
        mark(0);
        hipLaunchKernelGGL(tdgz2::k_gz_lz, dim3((nchunks + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st,
                           g.d_tok.p, g.d_chunks.p, g.d_res.p, g.d_symoff.p, nchunks, g.d_sym.p);
        mark(1);
        {   // the windows: segments of chunks (gz_gpu.hpp, 5.)
            const uint32_t seg_len = 32, nseg = (nchunks + seg_len - 1) / seg_len;
            rc = g.d_maps.ensure((size_t)nseg * tdgz2::WINDOW); if (rc) return no_room();
            rc = g.d_segwin.ensure((size_t)nseg * tdgz2::WINDOW); if (rc) return no_room();
            if (!h->gz_attr_done) {                                          // (per device: a handle has one)
                HIPCHK(hipFuncSetAttribute((const void *)tdgz2::k_gz_windows<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (int)tdgz2::WINDOW));
                HIPCHK(hipFuncSetAttribute((const void *)tdgz2::k_gz_windows<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)tdgz2::WINDOW));
                h->gz_attr_done = true;
            }
            hipLaunchKernelGGL(tdgz2::k_gz_windows<uint16_t>, dim3(nseg), dim3(1024), 4 * tdgz2::WINDOW, st, g.d_sym.p, g.d_symoff.p, g.d_res.p, nchunks, seg_len,
                               (const uint16_t *)nullptr, g.d_maps.p, (uint8_t *)nullptr);
            hipLaunchKernelGGL(tdgz2::k_gz_seg_windows, dim3(1), dim3(1024), 0, st, g.d_maps.p, nseg, g.d_segwin.p, g.d_carry.p);
            hipLaunchKernelGGL(tdgz2::k_gz_windows<uint8_t>, dim3(nseg), dim3(1024), 2 * tdgz2::WINDOW, st, g.d_sym.p, g.d_symoff.p, g.d_res.p, nchunks, seg_len,
                               (const uint8_t *)g.d_segwin.p, (uint8_t *)nullptr, g.d_win.p);
        }
        mark(2);
        std::vector<uint32_t> crcs(nblk);
        uint32_t flag = 0;
        if (nblk) {
            hipLaunchKernelGGL(tdgz::k_gz_resolve, dim3(nblk), dim3(256), 0, st, (const uint8_t *)g.d_sym.p, g.d_win.p, g.d_out.p + carry, g.d_blk.p, nblk, h->d_gzflag.p);
            mark(3);
            hipLaunchKernelGGL(tdgz::k_gz_crc, dim3((nblk + 63) / 64), dim3(64), 0, st, g.d_out.p + carry, g.d_blk.p, nblk, h->d_crctab, g.d_crc.p);
            mark(4);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(crcs.data(), g.d_crc.p, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(&flag, h->d_gzflag.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double t4 = PI::now();
        if (verbose && nblk) {
            float a = 0, b = 0, c = 0, d = 0;
            (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]); (void)hipEventElapsedTime(&c, ev[2], ev[3]); (void)hipEventElapsedTime(&d, ev[3], ev[4]);
            uint64_t ntok = 0, nslow = 0; for (const auto &o : res) { ntok += o.ntok; nslow += o.nslow; }
            fprintf(stderr, "gz_gpu_inflate: %.1f %% of the tokens by the scalar code (a code longer than the table's index)\n", 100.0 * (double)nslow / (double)std::max<uint64_t>(1, ntok));
            fprintf(stderr, "gz_gpu_inflate: segment %lu: %.1f MB -> %.1f MB, %u chunks, %.1f M tokens: upload %.1f ms, block search %.1f ms, Huffman decoding %.1f ms, "
                    "k_gz_lz %.1f ms, windows %.1f ms, k_gz_resolve %.1f ms, k_gz_crc %.1f ms\n", (unsigned long)segments, (double)(seg_end - base) / 1e6, total / 1e6,
                    nchunks, ntok / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, a, b, c, d);
        }
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        t_up += t1 - t0; t_find += t2 - t1; t_tok += t3 - t2; t_rest += t4 - t3;
        if (flag) return fail(TD_E_IO, "gzip: distance reaches before the start of the output");
        for (uint32_t k = 0; k < nblk; k++) crc_run = FI::crc32_join(crc_run, crcs[k], blocks[k].len);
        member_out += total;
        member_fresh = false;
        if (member_done) {
            if (crc_run != want_crc) return fail(TD_E_IO, "gzip member fails its CRC-32 check");
            crc_run = 0; member_out = 0; member_fresh = true;
            if (next_member) next_bit = next_member_bit;
            else file_done = true;
        } else {
            next_bit = end_abs;
        }
        segments++;
        *nbytes = total;
        *last = file_done;
        return TD_OK;
    }
};

// *not_applicable: nothing has been counted and the host decoder (count_gzip_dev) should take the file.  What the device
// decoder gives up on LATER in a file (a token buffer that overflows, a Huffman code it does not build) is a refusal like a
// decoding error: the reference's reading rules take the file (gz_pyrules.hpp).
int count_gzip_gpu(td_handle *h, const char *path, uint64_t max_reads, int weights, bool *not_applicable) {
    *not_applicable = true;
    GzGpuStream zs(h, path);
    if (!zs.open()) {
        if (zs.verbose) fprintf(stderr, "gz_gpu_inflate: %s -- the host decoder takes the file\n", zs.why);
        return TD_OK;
    }
    hipStream_t st = h->work_stream;
    HIPCHK(hipMemsetAsync(h->d_cursor.p, 0, 16, st));
    if (!h->pin_cursor) HIPCHK(hipHostMalloc((void **)&h->pin_cursor, 16, hipHostMallocDefault));
    h->pin_cursor[0] = 0;
    uint8_t *pin_tail = nullptr;
    struct Free { uint8_t *&p; ~Free() { if (p) (void)hipHostFree(p); } } free_tail{pin_tail};
    const uint64_t stop_line = max_reads >= (1ull << 60) ? ~0ull : 4 * (std::max<uint64_t>(1, max_reads) - 1) + 2;
    size_t carry = 0;
    uint64_t bytes_submitted = 0;
    unsigned pieces = 0;
    const double t0 = tdhost::ParInflate::now();
    for (;;) {
        uint64_t nb = 0; bool last = false, gave_up = false;
        int rc = zs.next(carry, &nb, &last, &gave_up);
        if (rc) { *not_applicable = false; return rc; }
        if (gave_up) {
            if (pieces == 0 && zs.segments == 0) { if (zs.verbose) fprintf(stderr, "gz_gpu_inflate: the host decoder takes the file\n"); return TD_OK; }
            *not_applicable = false;
            return fail(TD_E_IO, std::string("gzip (device decoder): ") + zs.why);
        }
        *not_applicable = false;                                           // (from here on the file's verdict is this decoder's)
        td_handle::GzGpu &g = h->gzgpu;
        const size_t total = carry + nb;
        size_t cut = total;
        if (!last && total) {
            if (!pin_tail) HIPCHK(hipHostMalloc((void **)&pin_tail, ZB_TAIL, hipHostMallocDefault));
            const size_t ntail = std::min(total, ZB_TAIL);
            HIPCHK(hipMemcpyAsync(pin_tail, g.d_out.p + total - ntail, ntail, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const size_t c = cut_at_line_end(pin_tail, ntail);
            if (c == 0) {
                if (total > ZB_CARRY) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
                cut = 0;
            } else {
                if (ntail - c > ZB_CARRY) return fail(TD_E_LIMIT, "a single line exceeds the staging buffer");
                cut = total - ntail + c;
            }
        }
        if (cut) {
            rc = launch_count(h, g.d_out.p, cut, 0, max_reads, weights, st, h->d_cursor.p + (pieces & 1), h->d_cursor.p + ((pieces + 1) & 1), bytes_submitted);
            if (rc) return rc;
            bytes_submitted += cut; pieces++;
            HIPCHK(hipMemcpyAsync(h->pin_cursor, h->d_cursor.p + (pieces & 1), 8, hipMemcpyDeviceToHost, st));
        }
        const size_t carry_next = total - cut;
        if (!last && carry_next) {
            // (to the front of the same buffer, behind the count: the pieces do not overlap when the text is longer than twice the carry)
            if (cut >= carry_next) HIPCHK(hipMemcpyAsync(g.d_out.p, g.d_out.p + cut, carry_next, hipMemcpyDeviceToDevice, st));
            else {
                DevBuf<uint8_t> tmp; rc = tmp.ensure(carry_next); if (rc) return rc;
                HIPCHK(hipMemcpyAsync(tmp.p, g.d_out.p + cut, carry_next, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(g.d_out.p, tmp.p, carry_next, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                tmp.release();
            }
        }
        HIPCHK(hipStreamSynchronize(st));
        carry = carry_next;
        if (last) break;
        if (h->pin_cursor[0] >= stop_line) break;                          // (the segments counted so far already hold read number max_reads)
    }
    h->last_gz_route = 1;
    if (zs.verbose)
        fprintf(stderr, "count_gzip_gpu: %lu segments in %.1f ms: upload %.1f, block search %.1f, Huffman decoding %.1f, symbols + windows + bytes + CRC %.1f ms\n",
                (unsigned long)zs.segments, (tdhost::ParInflate::now() - t0) * 1e3, zs.t_up * 1e3, zs.t_find * 1e3, zs.t_tok * 1e3, zs.t_rest * 1e3);
    return TD_OK;
}

// ---- a .gz input one of the decoders has refused: what does the reference do with it?  (gz_pyrules.hpp)
int gz_code(int kind) { return kind == tdhost::GZ_EOF ? TD_E_GZ_EOF : kind == tdhost::GZ_BADFILE ? TD_E_GZ_BADFILE : TD_E_GZ_DATA; }
bool gz_refusal(int rc) { return rc == TD_E_IO || rc == TD_E_LIMIT; }

// The fast route's answer `rc` for a .gz file -> the reference's: its exception (class by the code, message in
// td_last_error), or -- where its loop is through before the damage is met, or the file is merely something the fast
// decoders do not take -- the counts, taken again through the reference's own reading rules.  That needs results that
// held nothing before this file (find_tags_fastq's case); into a matrix that was already accumulating the fast route's
// refusal stands.
int count_gz_by_reference_rules(td_handle *h, const char *path, uint64_t max_reads, int weights, int rc, bool was_fresh) {
    const std::string refusal = g_err;
    MappedFile mf(path);
    if (!mf.ok) return fail(rc, refusal);
    const tdhost::GzVerdict v = tdhost::gz_verdict(mf.p, mf.n, max_reads);
    if (v.kind != tdhost::GZ_OK) return fail(gz_code(v.kind), v.message);
    if (!was_fresh)
        return fail(rc, refusal + " (the reference's loop ends before it meets this; the results were accumulating, so the file is not counted again)");
    int rc2 = zero_results(h); if (rc2) return rc2;
    if (h->bound_counts) {
        HIPCHK(hipMemsetAsync(h->bound_counts, 0, (size_t)h->barnum * h->ntags * 4, h->work_stream));
        HIPCHK(hipStreamSynchronize(h->work_stream));
    }
    tdhost::PyGzipReader pr(mf.p, mf.n);
    bool through = false;
    auto reader = [&](uint8_t *dst, size_t want) -> long {
        size_t done = 0;
        while (done < want && !through) {
            const long got = pr.read(dst + done, std::min<size_t>(tdhost::PyGzipReader::CHUNK, want - done));
            if (got <= 0) { through = true; break; }              // (damage behind the bound: the verdict above says the loop never gets there)
            done += (size_t)got;
        }
        return (long)done;
    };
    return pump(h, reader, 0, 0, max_reads, weights, nullptr);
}
}  // namespace

extern "C" {

int td_gunzip_file_gpu(td_handle *h, const char *path, void *dst, uint64_t capacity, uint64_t *n_out, int *on_gpu) {
    if (!h || !path || !n_out || !on_gpu || (!dst && capacity)) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    *n_out = 0; *on_gpu = 0;
    GzGpuStream zs(h, path);
    if (!zs.open()) return TD_OK;
    uint64_t at = 0;
    for (;;) {
        uint64_t nb = 0; bool last = false, gave_up = false;
        const int rc = zs.next(0, &nb, &last, &gave_up);
        if (rc) return rc;
        if (gave_up) return TD_OK;                                         // (whatever was copied so far is not reported)
        if (at + nb > capacity) return fail(TD_E_LIMIT, "destination too small");
        if (nb) HIPCHK(hipMemcpy((uint8_t *)dst + at, h->gzgpu.d_out.p, nb, hipMemcpyDeviceToHost));
        at += nb;
        if (last) break;
    }
    *n_out = at; *on_gpu = 1;
    return TD_OK;
}

int td_last_gz_route(td_handle *h) { return h ? h->last_gz_route : 0; }

// ---- one ordinary gzip file over several devices (tagdigger_amd/multi.py count_file_sharded): a rank's part of the pipeline
// of csrc/gz_gpu.hpp.  Every rank takes a byte range of the compressed file: open (its bytes to the device, the first block
// start in them), decode (up to the next rank's start: symbols, and the MAP of its stretch -- what each place of the window
// behind it holds in terms of the window in front of it), resolve (with the window the maps of the ranks before it give: text).
int td_gz_shard_open(td_handle *h, const char *path, uint64_t byte_lo, uint64_t byte_hi, int first, uint64_t *start_bit, uint64_t *file_bytes) {
    using FI = tdhost::FastInflate;
    if (!h || !path || !start_bit || !file_bytes) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    td_handle::GzShard &sh = h->gzshard;
    sh = td_handle::GzShard();
    MappedFile mf(path);
    if (!mf.ok) return fail(TD_E_IO, std::string("cannot open ") + path);
    const uint64_t n = mf.n;
    *file_bytes = n; *start_bit = ~0ull;
    if (byte_hi > n) byte_hi = n;
    if (byte_lo >= byte_hi) { sh.open = true; sh.path = path; sh.n = n; return TD_OK; }            // (nothing of the file: no start)
    uint64_t first_abs = ~0ull;
    if (first) {
        const uint8_t *body = nullptr; const char *herr = nullptr;
        if (FI::parse_member_header(mf.p, mf.p + n, &body, &herr, true) != 1) return fail(TD_E_IO, "gzip: no member header");
        first_abs = (uint64_t)(body - mf.p) * 8;
        byte_lo = 0;
    }
    const uint64_t MARGIN = (uint64_t)h->gz_gpu_margin_kb << 10, SEG = (uint64_t)h->gz_gpu_seg_kb << 10;
    const uint64_t base = byte_lo & ~(uint64_t)4095, up_end = std::min<uint64_t>(n, byte_hi + MARGIN), nb = up_end - base;
    if (nb > SEG + 2 * MARGIN) return fail(TD_E_LIMIT, "gzip shard larger than a segment");
    td_handle::GzGpu &g = h->gzgpu;
    int rc = g.d_in.ensure(nb + 8192 + 4096); if (rc) return rc;
    const size_t in_cap = ((nb + 4096 + 15) & ~(size_t)15);
    HIPCHK(hipStreamSynchronize(h->work_stream));
    HIPCHK(hipMemsetAsync(g.d_in.p + (nb & ~(size_t)15), 0, in_cap - (nb & ~(size_t)15), h->copy_stream));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    rc = td_load_file_range(h, path, base, nb, g.d_in.p); if (rc) return rc;
    const uint64_t in_bits = nb * 8, nwords = in_cap / 4;
    const uint64_t terr = (uint64_t)h->gz_gpu_terr_kb << 10;
    const uint32_t nterr = (uint32_t)((byte_hi - base + terr - 1) / terr);
    // block starts at or behind byte_lo (the first rank: behind its known start)
    const uint64_t lo_rel = first ? first_abs - base * 8 : std::max<uint64_t>((byte_lo - base) * 8, 1) - 1;
    rc = g.d_found.ensure(nterr + 4); if (rc) return rc;
    hipStream_t st = h->work_stream;
    HIPCHK(hipMemsetAsync(g.d_found.p, 0xFF, (size_t)nterr * 8, st));
    HIPCHK(hipMemsetAsync(g.d_found.p + nterr, 0, 32, st));
    hipLaunchKernelGGL(tdgz2::k_gz_find, dim3((nterr + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st,
                       g.d_in.p, in_bits, nwords, lo_rel, terr * 8, nterr, g.d_found.p, (uint32_t)h->gz_gpu_verify, 0u);
    HIPCHK(hipGetLastError());
    sh.found.assign(nterr, tdgz2::NONE);
    HIPCHK(hipMemcpyAsync(sh.found.data(), g.d_found.p, (size_t)nterr * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // (a start at or past byte_hi is the next rank's)
    const uint64_t hi_rel = (byte_hi - base) * 8;
    for (auto &f : sh.found) if (f != tdgz2::NONE && f >= hi_rel) f = tdgz2::NONE;
    uint64_t start_rel = tdgz2::NONE;
    if (first) start_rel = first_abs - base * 8;
    else for (uint64_t f : sh.found) if (f != tdgz2::NONE) { start_rel = f; break; }
    if (h->gz_gpu_false_every > 0 && !first && start_rel != tdgz2::NONE && start_rel + 4099 < in_bits) start_rel += 4099;      // (tests: a false start at the seam)
    sh.open = true; sh.path = path; sh.n = n; sh.base = base; sh.in_bits = in_bits; sh.nwords = nwords; sh.start_rel = start_rel;
    *start_bit = start_rel == tdgz2::NONE ? ~0ull : base * 8 + start_rel;
    return TD_OK;
}

// stop_bit: where the next rank with a start begins (~0: this rank's stretch runs to the member's end).  *final: it ended the member.
int td_gz_shard_decode(td_handle *h, uint64_t stop_bit, uint64_t *end_bit, uint64_t *out_len, int *final, uint16_t *map_out) {
    if (!h || !end_bit || !out_len || !final || !map_out) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    td_handle::GzShard &sh = h->gzshard;
    td_handle::GzGpu &g = h->gzgpu;
    if (!sh.open || sh.start_rel == tdgz2::NONE) return fail(TD_E_STATE, "td_gz_shard_open found no block start for this rank");
    hipStream_t st = h->work_stream;
    const uint64_t stop_rel = stop_bit == ~0ull ? ~0ull : stop_bit - sh.base * 8;
    if (stop_rel != ~0ull && stop_rel > sh.in_bits) return fail(TD_E_LIMIT, "gzip shard: the next rank's start lies beyond this rank's bytes");
    std::vector<tdgz2::Chunk> chunks;
    { tdgz2::Chunk c{}; c.start_bit = sh.start_rel; chunks.push_back(c); }
    for (uint64_t f : sh.found) if (f != tdgz2::NONE && f > sh.start_rel && f < stop_rel) { tdgz2::Chunk c{}; c.start_bit = f; chunks.push_back(c); }
    uint32_t nchunks = (uint32_t)chunks.size();
    {
        uint64_t at = 0;
        for (uint32_t i = 0; i < nchunks; i++) {
            chunks[i].stop_bit = i + 1 < nchunks ? chunks[i + 1].start_bit : stop_rel;
            const uint64_t span_end = i + 1 < nchunks ? chunks[i + 1].start_bit : (stop_rel == ~0ull ? sh.in_bits : std::min<uint64_t>(sh.in_bits, stop_rel + ((uint64_t)h->gz_gpu_margin_kb << 13)));
            const uint64_t span = (span_end - chunks[i].start_bit + 7) / 8;
            const uint64_t cap = std::min<uint64_t>(4 * span + 4096, 0xFFFFFF00u);
            chunks[i].tok_off = at; chunks[i].tok_cap = (uint32_t)cap;
            at += (cap + 63) & ~(uint64_t)63;
        }
        int rc = g.d_tok.ensure(at + 64); if (rc) return rc;
    }
    int rc = g.d_chunks.ensure(2 * (size_t)nchunks); if (rc) return rc;
    rc = g.d_res.ensure(2 * (size_t)nchunks); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(g.d_chunks.p, chunks.data(), (size_t)nchunks * sizeof(tdgz2::Chunk), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(g.d_res.p, 0, (size_t)nchunks * sizeof(tdgz2::ChunkOut), st));
    hipLaunchKernelGGL(tdgz2::k_gz_tokens, dim3((nchunks + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st,
                       g.d_in.p, sh.in_bits, sh.nwords, g.d_chunks.p, nchunks, g.d_tok.p, g.d_res.p);
    HIPCHK(hipGetLastError());
    std::vector<tdgz2::ChunkOut> res(nchunks);
    HIPCHK(hipMemcpyAsync(res.data(), g.d_res.p, (size_t)nchunks * sizeof(tdgz2::ChunkOut), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // the chain inside the rank (a false start is dropped where the chunk in front of it ends on a later start; anything else
    // sends the file to the one-rank path)
    std::vector<uint8_t> dead(nchunks, 0);
    bool ended = false;
    for (uint32_t i = 0;;) {
        const tdgz2::ChunkOut &o = res[i];
        if (o.status != tdgz2::S_BOUNDARY && o.status != tdgz2::S_FINAL) return fail(TD_E_LIMIT, "gzip shard: a chunk this decoder does not finish");
        if (o.status == tdgz2::S_FINAL) { for (uint32_t k = i + 1; k < nchunks; k++) dead[k] = 1; ended = true; break; }
        uint32_t j = i + 1;
        while (j < nchunks && (dead[j] || chunks[j].start_bit < o.end_bit)) { dead[j] = 1; j++; }
        if (j == nchunks) break;
        if (chunks[j].start_bit != o.end_bit) return fail(TD_E_LIMIT, "gzip shard: a chunk does not end on a block start that was found");
        i = j;
    }
    {
        uint32_t w = 0;
        for (uint32_t i = 0; i < nchunks; i++) if (!dead[i]) { chunks[w] = chunks[i]; res[w] = res[i]; w++; }
        chunks.resize(w); res.resize(w); nchunks = w;
        HIPCHK(hipMemcpyAsync(g.d_chunks.p, chunks.data(), (size_t)nchunks * sizeof(tdgz2::Chunk), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(g.d_res.p, res.data(), (size_t)nchunks * sizeof(tdgz2::ChunkOut), hipMemcpyHostToDevice, st));
    }
    sh.sym_off.assign(nchunks + 1, 0);
    uint64_t total = 0;
    for (uint32_t i = 0; i < nchunks; i++) { sh.sym_off[i] = total; total += res[i].out_len; }
    sh.sym_off[nchunks] = total;
    rc = g.d_sym.ensure(total + 64); if (rc) return rc;
    rc = g.d_symoff.ensure(nchunks + 1); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(g.d_symoff.p, sh.sym_off.data(), (size_t)(nchunks + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(tdgz2::k_gz_lz, dim3((nchunks + tdgz2::WAVES - 1) / tdgz2::WAVES), dim3(64 * tdgz2::WAVES), 0, st,
                       g.d_tok.p, g.d_chunks.p, g.d_res.p, g.d_symoff.p, nchunks, g.d_sym.p);
    const uint32_t seg_len = 32, nseg = (nchunks + seg_len - 1) / seg_len;
    rc = g.d_maps.ensure((size_t)(nseg + 1) * tdgz2::WINDOW); if (rc) return rc;
    if (!h->gz_attr_done) {
        HIPCHK(hipFuncSetAttribute((const void *)tdgz2::k_gz_windows<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (int)tdgz2::WINDOW));
        HIPCHK(hipFuncSetAttribute((const void *)tdgz2::k_gz_windows<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)tdgz2::WINDOW));
        h->gz_attr_done = true;
    }
    HIPCHK(hipFuncSetAttribute((const void *)tdgz2::k_gz_compose_maps, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (int)tdgz2::WINDOW));
    hipLaunchKernelGGL(tdgz2::k_gz_windows<uint16_t>, dim3(nseg), dim3(1024), 4 * tdgz2::WINDOW, st, g.d_sym.p, g.d_symoff.p, g.d_res.p, nchunks, seg_len,
                       (const uint16_t *)nullptr, g.d_maps.p, (uint8_t *)nullptr);
    hipLaunchKernelGGL(tdgz2::k_gz_compose_maps, dim3(1), dim3(1024), 4 * tdgz2::WINDOW, st, g.d_maps.p, nseg, g.d_maps.p + (size_t)nseg * tdgz2::WINDOW);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(map_out, g.d_maps.p + (size_t)nseg * tdgz2::WINDOW, (size_t)tdgz2::WINDOW * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    sh.chunks = chunks; sh.res = res; sh.total = total; sh.nseg = nseg; sh.decoded = true;
    *end_bit = sh.base * 8 + res[nchunks - 1].end_bit;
    *out_len = total;
    *final = ended ? 1 : 0;
    return TD_OK;
}

// window_in: the 32 KiB in front of this rank's stretch (from the maps of the ranks before it); member_out_before: the bytes the
// member has inflated to before it.  *d_text: the stretch's text in device memory (out_len bytes; room behind it for an eighth more),
// *crc32: its CRC-32.
int td_gz_shard_resolve(td_handle *h, const uint8_t *window_in, uint64_t member_out_before, void **d_text, uint32_t *crc32) {
    using FI = tdhost::FastInflate;
    if (!h || !window_in || !d_text || !crc32) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    td_handle::GzShard &sh = h->gzshard;
    td_handle::GzGpu &g = h->gzgpu;
    if (!sh.decoded) return fail(TD_E_STATE, "td_gz_shard_decode has not run");
    hipStream_t st = h->work_stream;
    const uint32_t nchunks = (uint32_t)sh.chunks.size(), seg_len = 32;
    const uint64_t total = sh.total;
    uint64_t nblk64 = 0;
    for (uint32_t i = 0; i < nchunks; i++) nblk64 += (sh.res[i].out_len + tdgz::BLOCK_SYMS - 1) / tdgz::BLOCK_SYMS;
    if (nblk64 >= 0x7FFFFFFFull) return fail(TD_E_LIMIT, "gzip shard too large");
    const uint32_t nblk = (uint32_t)nblk64;
    int rc = g.d_out.ensure(total + 4096 + (total >> 3) + ((size_t)1 << 20)); if (rc) return rc;
    rc = g.d_win.ensure((size_t)nchunks * tdgz2::WINDOW); if (rc) return rc;
    rc = g.d_carry.ensure(tdgz2::WINDOW); if (rc) return rc;
    rc = g.d_segwin.ensure((size_t)sh.nseg * tdgz2::WINDOW); if (rc) return rc;
    rc = g.d_blk.ensure(nblk + 1); if (rc) return rc;
    rc = g.d_crc.ensure(nblk + 1); if (rc) return rc;
    rc = h->d_gzflag.ensure(4); if (rc) return rc;
    if (!h->d_crctab) { bool ok = true; rc = ensure_bgzf_buffers(h, 0, 0, &ok); if (rc) return rc; }
    std::vector<tdgz::Block> blocks(nblk);
    {
        size_t kb = 0;
        for (uint32_t i = 0; i < nchunks; i++) {
            const uint32_t min_idx = tdgz2::WINDOW - (uint32_t)std::min<uint64_t>(tdgz2::WINDOW, member_out_before + sh.sym_off[i]);
            for (uint64_t o = 0; o < sh.res[i].out_len; o += tdgz::BLOCK_SYMS)
                blocks[kb++] = tdgz::Block{(sh.sym_off[i] + o) * 2, sh.sym_off[i] + o, (uint32_t)std::min<uint64_t>(tdgz::BLOCK_SYMS, sh.res[i].out_len - o), i, min_idx, 0u};
        }
    }
    HIPCHK(hipMemcpyAsync(g.d_carry.p, window_in, tdgz2::WINDOW, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(g.d_blk.p, blocks.data(), (size_t)nblk * sizeof(tdgz::Block), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(h->d_gzflag.p, 0, 16, st));
    hipLaunchKernelGGL(tdgz2::k_gz_seg_windows, dim3(1), dim3(1024), 0, st, g.d_maps.p, sh.nseg, g.d_segwin.p, g.d_carry.p);
    hipLaunchKernelGGL(tdgz2::k_gz_windows<uint8_t>, dim3(sh.nseg), dim3(1024), 2 * tdgz2::WINDOW, st, g.d_sym.p, g.d_symoff.p, g.d_res.p, nchunks, seg_len,
                       (const uint8_t *)g.d_segwin.p, (uint8_t *)nullptr, g.d_win.p);
    std::vector<uint32_t> crcs(nblk);
    uint32_t flag = 0;
    if (nblk) {
        hipLaunchKernelGGL(tdgz::k_gz_resolve, dim3(nblk), dim3(256), 0, st, (const uint8_t *)g.d_sym.p, g.d_win.p, g.d_out.p, g.d_blk.p, nblk, h->d_gzflag.p);
        hipLaunchKernelGGL(tdgz::k_gz_crc, dim3((nblk + 63) / 64), dim3(64), 0, st, g.d_out.p, g.d_blk.p, nblk, h->d_crctab, g.d_crc.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(crcs.data(), g.d_crc.p, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipMemcpyAsync(&flag, h->d_gzflag.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (flag) return fail(TD_E_IO, "gzip: distance reaches before the start of the output");
    uint32_t crc = 0;
    for (uint32_t k = 0; k < nblk; k++) crc = FI::crc32_join(crc, crcs[k], blocks[k].len);
    *crc32 = crc;
    *d_text = g.d_out.p;
    return TD_OK;
}

uint32_t td_crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return tdhost::FastInflate::crc32_join(crc_a, crc_b, len_b); }

int td_gzip_check(const char *path, uint64_t max_reads) {
    if (!path) return fail(TD_E_ARG, "NULL argument");
    MappedFile mf(path);
    if (!mf.ok) return fail(TD_E_IO, std::string("cannot open ") + path);
    const tdhost::GzVerdict v = tdhost::gz_verdict(mf.p, mf.n, max_reads);
    if (v.kind != tdhost::GZ_OK) return fail(gz_code(v.kind), v.message);
    return TD_OK;
}

int td_count_host(td_handle *h, const void *fastq, uint64_t nbytes, uint64_t first_line, uint64_t max_reads,
                  int weights, uint64_t *lines_out) {
    if (!h) return fail(TD_E_ARG, "handle is NULL");
    if (!h->have_index) return fail(TD_E_STATE, "td_set_index has not been called");
    HIPCHK(hipSetDevice(h->device));
    const uint8_t *src = (const uint8_t *)fastq;
    uint64_t pos = 0;
    auto reader = [&](uint8_t *dst, size_t want) -> long {
        size_t n = (size_t)std::min<uint64_t>(want, nbytes - pos);
        if (n) stage_parallel(n, [&](size_t off, size_t len) { memcpy(dst + off, src + pos + off, len); return true; });
        pos += n;
        return (long)n;
    };
    return pump(h, reader, nbytes + 1, first_line, max_reads, weights, lines_out);
}

static int gunzip_fast(const char *path, void *dst, uint64_t capacity, uint64_t chunk, uint64_t *n_out);
int td_gunzip_file(const char *path, void *dst, uint64_t capacity, uint64_t chunk, uint64_t *n_out) {
    if (!path || !dst || !n_out) return fail(TD_E_ARG, "NULL argument");
    {
        struct stat sb0;
        if (stat(path, &sb0) == 0 && S_ISREG(sb0.st_mode) && sb0.st_size == 0) { *n_out = 0; return TD_OK; }     // (gzip.open: no data)
    }
    const int rc = gunzip_fast(path, dst, capacity, chunk, n_out);
    if (rc != TD_E_IO) return rc;
    // a decoder has refused the file: the reference's reading rules say how this ends (gz_pyrules.hpp)
    const std::string refusal = g_err;
    MappedFile mf(path);
    if (!mf.ok) return fail(rc, refusal);
    tdhost::PyGzipReader pr(mf.p, mf.n);
    uint64_t n = 0;
    uint8_t extra[tdhost::PyGzipReader::CHUNK];
    for (;;) {
        const bool full = n + tdhost::PyGzipReader::CHUNK > capacity;      // (a request's bytes may not fit any more: aside, then copied)
        const long got = pr.read(full ? extra : (uint8_t *)dst + n, tdhost::PyGzipReader::CHUNK);
        if (got < 0) return fail(gz_code(pr.kind), pr.message);
        if (got == 0) break;
        if (full) {
            if (n + (uint64_t)got > capacity) return fail(TD_E_LIMIT, "destination too small");
            memcpy((uint8_t *)dst + n, extra, (size_t)got);
        }
        n += (uint64_t)got;
    }
    *n_out = n;
    return TD_OK;
}
static int gunzip_fast(const char *path, void *dst, uint64_t capacity, uint64_t chunk, uint64_t *n_out) {
    if (getenv("TAGDIG_GUNZIP_PIPELINE")) {
        // the decoder as count_gzip_dev drives it (dev_next / dev_release / dev_check), markers resolved by the host: the
        // pipeline can be checked where there is no GPU
        static const tdhost::ParInflate::Allocator plain = {[](size_t b) -> void * { return malloc(b); }, [](void *p, size_t) { free(p); }};
        tdhost::GzSource dsrc;
        if (dsrc.open_dev(path, &plain)) {
            uint64_t n = 0;
            for (;;) {
                const tdhost::ParInflate::DevBatch *db = dsrc.pi.dev_next();
                if (!db) {
                    if (dsrc.pi.dev_failed()) return fail(TD_E_IO, std::string("gzip: ") + dsrc.pi.error());
                    break;
                }
                if (db->failed) { const std::string e = dsrc.pi.error(); dsrc.pi.dev_release(); return fail(TD_E_IO, "gzip: " + e); }
                if (n + db->total > capacity) { dsrc.pi.dev_release(); return fail(TD_E_LIMIT, "destination too small"); }
                std::vector<std::pair<uint32_t, size_t>> crc_len;
                bool ok = true;
                for (const auto &pc : db->pieces) {
                    uint32_t c = 0;
                    ok &= tdhost::ParInflate::dev_resolve_on_host(pc, (uint8_t *)dst + n + pc.dest_off, &c);
                    crc_len.emplace_back(c, pc.len);
                }
                n += db->total;
                const bool last = db->last, member_done = db->member_done;
                const uint32_t want = db->want_crc;
                dsrc.pi.dev_release();
                if (!ok) return fail(TD_E_IO, "gzip: distance reaches before the start of the output");
                if (!dsrc.pi.dev_check(crc_len, member_done, want)) return fail(TD_E_IO, std::string("gzip: ") + dsrc.pi.error());
                if (last) break;
            }
            *n_out = n;
            return TD_OK;
        }
    }
    tdhost::GzSource src;
    if (!src.open(path)) return fail(TD_E_IO, std::string("cannot open ") + path);
    uint64_t n = 0;
    if (chunk == 0) chunk = 1 << 20;
    for (;;) {
        if (n == capacity) {                      // full: fine only if the stream ends here
            uint8_t extra;
            const long got = src.read(&extra, 1);
            if (got < 0) return fail(TD_E_IO, "inflate error");
            if (got > 0) return fail(TD_E_LIMIT, "destination too small");
            break;
        }
        const long got = src.read((uint8_t *)dst + n, (size_t)std::min<uint64_t>(chunk, capacity - n));
        if (got < 0) return fail(TD_E_IO, "inflate error");
        if (got == 0) break;
        n += (uint64_t)got;
    }
    *n_out = n;
    return TD_OK;
}

int td_count_file(td_handle *h, const char *path, uint64_t max_reads, int weights) {
    if (!h || !path) return fail(TD_E_ARG, "NULL argument");
    if (!h->have_index) return fail(TD_E_STATE, "td_set_index has not been called");
    HIPCHK(hipSetDevice(h->device));
    const size_t len = strlen(path);
    const bool gz = len >= 2 && (path[len - 2] == 'g' || path[len - 2] == 'G') && (path[len - 1] == 'z' || path[len - 1] == 'Z');
    if (gz) {
        const bool was_fresh = !h->counted;
        h->last_gz_route = 0;
        auto fast = [&]() -> int {
            {   // (an empty file: gzip.open reads it as no data at all)
                struct stat sb0;
                if (stat(path, &sb0) == 0 && S_ISREG(sb0.st_mode) && sb0.st_size == 0) return TD_OK;
            }
            static const bool env_off = getenv("TAGDIG_GPU_INFLATE") && atoi(getenv("TAGDIG_GPU_INFLATE")) == 0;
            if (h->gpu_inflate && !env_off && !getenv("TAGDIG_ZLIB")) {          // BGZF: members inflated on the GPU
                bool not_bgzf = false;
                const int rc = count_bgzf_gpu(h, path, max_reads, weights, &not_bgzf);
                if (rc || !not_bgzf) return rc;
            }
            static const bool resolve_off = getenv("TAGDIG_GPU_RESOLVE") && atoi(getenv("TAGDIG_GPU_RESOLVE")) == 0;
            static const bool huffman_off = getenv("TAGDIG_GPU_HUFFMAN") && atoi(getenv("TAGDIG_GPU_HUFFMAN")) == 0;
            if (h->gpu_huffman && !huffman_off && !getenv("TAGDIG_ZLIB")) {       // ordinary gzip: decoded on the GPU
                bool not_applicable = false;
                const int rc = count_gzip_gpu(h, path, max_reads, weights, &not_applicable);
                if (rc || !not_applicable) return rc;
            }
            if (h->gpu_resolve && !resolve_off) {                                 // ordinary gzip: decoded on the host threads, resolved on the GPU
                bool not_applicable = false;
                const int rc = count_gzip_dev(h, path, max_reads, weights, &not_applicable);
                if (rc || !not_applicable) return rc;
            }
            tdhost::GzSource src;
            if (!src.open(path)) return fail(TD_E_IO, std::string("cannot open ") + path);
            auto reader = [&](uint8_t *dst, size_t want) -> long { return src.read(dst, want); };
            return pump(h, reader, 0, 0, max_reads, weights, nullptr);
        };
        const int rc = fast();
        // a decoder has refused the file: the reference's own reading rules say how this ends (its exception, or -- the loop
        // being through before the damage, or the file merely unusual -- its counts)
        if (gz_refusal(rc)) return count_gz_by_reference_rules(h, path, max_reads, weights, rc, was_fresh);
        return rc;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(TD_E_IO, std::string("cannot open ") + path);
    struct stat sb;
    const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    uint64_t fpos = 0;
    auto reader = [&](uint8_t *dst, size_t want) -> long {
        if (!regular) {                                    // a pipe or device: plain sequential reads
            const ssize_t n = read(fd, dst, want);
            return (long)n;
        }
        const uint64_t size = (uint64_t)sb.st_size;
        const size_t n = (size_t)std::min<uint64_t>(want, fpos < size ? size - fpos : 0);
        if (n == 0) return 0;
        const bool ok = stage_parallel(n, [&](size_t off, size_t len) {
            while (len) {
                const ssize_t got = pread(fd, dst + off, len, (off_t)(fpos + off));
                if (got <= 0) return false;
                off += (size_t)got; len -= (size_t)got;
            }
            return true;
        });
        if (!ok) return -1;
        fpos += n;
        return (long)n;
    };
    int rc = pump(h, reader, 0, 0, max_reads, weights, nullptr);
    close(fd);
    return rc;
}

// ---- a byte range of a file, or a range of BGZF members, brought into DEVICE memory (multi-GPU: one file over several
// ranks, tagdigger_amd/multi.py count_file_sharded): staged through the handle's pinned buffers -- the host never holds
// more than two staging pieces of the shard
int td_load_file_range(td_handle *h, const char *path, uint64_t offset, uint64_t length, void *d_dst) {
    if (!h || !path || (!d_dst && length)) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    if (length == 0) return TD_OK;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(TD_E_IO, std::string("cannot open ") + path);
    struct FdEnd { int fd; ~FdEnd() { close(fd); } } fd_end{fd};
    constexpr size_t PIECE = (size_t)32 << 20;
    for (auto &lp : h->ldpiece) {
        if (lp.pin) continue;
        HIPCHK(hipHostMalloc((void **)&lp.pin, PIECE, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&lp.sent, hipEventDisableTiming));
    }
    int k = 0;
    for (uint64_t pos = 0; pos < length; pos += PIECE, k ^= 1) {
        td_handle::ZPiece &lp = h->ldpiece[k];
        if (lp.busy) { HIPCHK(hipEventSynchronize(lp.sent)); lp.busy = false; }
        const size_t n = (size_t)std::min<uint64_t>(PIECE, length - pos);
        const bool ok = stage_parallel(n, [&](size_t off, size_t len) {
            while (len) {
                const ssize_t got = pread(fd, lp.pin + off, len, (off_t)(offset + pos + off));
                if (got <= 0) return false;
                off += (size_t)got; len -= (size_t)got;
            }
            return true;
        });
        if (!ok) return fail(TD_E_IO, "read error (or the file is shorter than offset + length)");
        { const int urc = upload(h, (uint8_t *)d_dst + pos, lp.pin, n); if (urc) return urc; }
        HIPCHK(hipEventRecord(lp.sent, h->copy_stream));
        lp.busy = true;
    }
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    for (auto &lp : h->ldpiece) lp.busy = false;
    return TD_OK;
}

int td_bgzf_index(const char *path, uint64_t *member_off, uint32_t *member_isize, uint64_t capacity, uint64_t *n_members) {
    if (!path || !n_members || (capacity && (!member_off || !member_isize))) return fail(TD_E_ARG, "NULL argument");
    tdhost::GzSource src;
    if (!src.map_only(path)) return fail(TD_E_IO, std::string("cannot open ") + path);
    uint64_t n = 0;
    for (size_t at = 0; at < src.bsize;) {
        uint32_t bs = 0, hs = 0;
        if (!tdhost::GzSource::bgzf_header(src.map + at, src.bsize - at, &bs, &hs) || bs < hs + 8 || at + bs > src.bsize)
            return fail(TD_E_IO, "not a BGZF file (a member without the BC extra field, or damaged)");
        const uint8_t *tail = src.map + at + bs - 8;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) return fail(TD_E_IO, "BGZF member larger than 64 KiB");
        if (n < capacity) { member_off[n] = at; member_isize[n] = isize; }
        n++;
        at += bs;
    }
    *n_members = n;
    return TD_OK;
}

int td_bgzf_inflate_range(td_handle *h, const char *path, uint64_t off_begin, uint64_t off_end, void *d_dst, uint64_t capacity,
                          uint64_t *nbytes_out) {
    if (!h || !path || !nbytes_out || (!d_dst && capacity)) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    *nbytes_out = 0;
    tdhost::GzSource src;
    if (!src.map_only(path)) return fail(TD_E_IO, std::string("cannot open ") + path);
    if (off_end > src.bsize) off_end = src.bsize;
    if (off_begin >= off_end) return TD_OK;
    uint32_t need_members = ZB_MEMBERS;
    size_t need_in = ZB_IN;
    if (off_end - off_begin <= ((uint64_t)1 << 30)) {
        need_in = std::min<size_t>(ZB_IN, (size_t)((off_end - off_begin + 4095) / 4096 * 4096));
        need_members = (uint32_t)std::min<uint64_t>(ZB_MEMBERS, std::max<uint64_t>(64, ((off_end - off_begin) / 28 + 64) / 64 * 64));   // (a member is 28 bytes at least)
        // (an upper bound only: walk the range when it is small enough for the bound to be wasteful)
        uint32_t n = 0;
        for (size_t at = off_begin; at < off_end; n++) {
            uint32_t bs = 0, hs = 0;
            if (!tdhost::GzSource::bgzf_header(src.map + at, src.bsize - at, &bs, &hs) || bs < hs + 8 || at + bs > src.bsize)
                return fail(TD_E_IO, "damaged BGZF member header");
            at += bs;
        }
        need_members = std::min<uint32_t>(ZB_MEMBERS, std::max<uint32_t>(64, (n + 63) / 64 * 64));
    }
    bool ok = true;
    int rc = ensure_bgzf_buffers(h, need_members, need_in, &ok);
    if (rc) return rc;
    if (!ok) return fail(TD_E_HIP, "no room on the device for the BGZF batch buffers");
    const uint32_t batch_members = std::min<uint32_t>(h->zcap_members, h->zb_members);
    const size_t batch_in = h->zcap_in, piece_cap = std::min(ZB_PIECE, h->zcap_in + 64);
    size_t pos = off_begin;
    uint64_t out_pos = 0;
    struct Batch { uint32_t n = 0; size_t out_total = 0; };
    auto prepare = [&](int slot, Batch &b) -> int {
        td_handle::ZSlot &z = h->zslot[slot];
        b = Batch();
        const size_t first = pos;
        while (pos < off_end && b.n < batch_members) {
            uint32_t bs = 0, hs = 0;
            if (!tdhost::GzSource::bgzf_header(src.map + pos, src.bsize - pos, &bs, &hs) || bs < hs + 8 || pos + bs > src.bsize)
                return fail(TD_E_IO, "damaged BGZF member header");
            if (pos + bs - first > batch_in) break;
            const uint8_t *tail = src.map + pos + bs - 8;
            const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
            if (isize > 65536) return fail(TD_E_IO, "BGZF member larger than 64 KiB");
            z.pin_mem[b.n] = tdinf::Member{pos + hs - first, b.out_total, bs - hs - 8, isize, crc, 0};
            b.out_total += isize; b.n++;
            pos += bs;
        }
        const size_t nin = pos - first;
        int k = 0;
        for (size_t off = 0; off < nin + 64; off += piece_cap, k ^= 1) {
            td_handle::ZPiece &zp = h->zpiece[k];
            if (zp.busy) { HIPCHK(hipEventSynchronize(zp.sent)); zp.busy = false; }
            const size_t want = std::min(piece_cap, nin + 64 - off), have = off < nin ? std::min(want, nin - off) : 0;
            if (have) stage_parallel(have, [&](size_t o2, size_t len) { memcpy(zp.pin + o2, src.map + first + off + o2, len); return true; });
            if (want > have) memset(zp.pin + have, 0, want - have);
            { const int urc = upload(h, z.d_in + off, zp.pin, want); if (urc) return urc; }
            HIPCHK(hipEventRecord(zp.sent, h->copy_stream));
            zp.busy = true;
        }
        if (b.n) HIPCHK(hipMemcpyAsync(z.d_mem, z.pin_mem, (size_t)b.n * sizeof(tdinf::Member), hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(hipEventRecord(z.copied, h->copy_stream));
        return TD_OK;
    };
    Batch cur, nxt;
    int slot = 0;
    rc = prepare(0, cur); if (rc) return rc;
    while (cur.n) {
        td_handle::ZSlot &z = h->zslot[slot];
        if (out_pos + cur.out_total > capacity) return fail(TD_E_LIMIT, "destination too small for the inflated members");
        HIPCHK(hipStreamWaitEvent(h->work_stream, z.copied, 0));
        hipLaunchKernelGGL(tdinf::k_bgzf_inflate, dim3((cur.n + 63) / 64), dim3(64), 64 * tdinf::TABLE_U16 * 2 + 4096, h->work_stream,
                           z.d_in, (uint8_t *)d_dst + out_pos, z.d_mem, cur.n, h->d_zscratch, z.d_status, h->d_crctab, (uint32_t)h->gpu_inflate_crc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(z.pin_status, z.d_status, (size_t)cur.n * 4, hipMemcpyDeviceToHost, h->work_stream));
        // the next batch is read and sent while this one inflates
        rc = prepare(slot ^ 1, nxt); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->work_stream));
        for (uint32_t i = 0; i < cur.n; i++)
            if (z.pin_status[i]) return fail(TD_E_IO, z.pin_status[i] == 100 ? "BGZF member fails its CRC-32" : "inflate error in a BGZF member");
        out_pos += cur.out_total;
        cur = nxt; slot ^= 1;
    }
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    for (auto &zp : h->zpiece) zp.busy = false;
    *nbytes_out = out_pos;
    return TD_OK;
}

int td_get_stats(td_handle *h, uint64_t stats[TD_STAT_NSTATS]) {
    if (!h || !stats) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long st[TD_STAT_NSTATS];
    HIPCHK(hipMemcpy(st, h->d_stats.p, sizeof(st), hipMemcpyDeviceToHost));
    for (int i = 0; i < TD_STAT_NSTATS; i++) stats[i] = st[i];
    return check_device_errors(h, st);
}

int td_get_progress(td_handle *h, uint64_t *out, uint64_t cap, uint64_t *nwindows) {
    if (!h || !nwindows || (cap && !out)) return fail(TD_E_ARG, "NULL argument");
    uint64_t st[TD_STAT_NSTATS];
    int rc = td_get_stats(h, st);                         // (synchronises; raises what a kernel flagged)
    if (rc) return rc;
    if (!h->progress) return fail(TD_E_STATE, "option progress is off");
    const uint64_t n = (st[TD_STAT_READS] + tdk::PROG_WINDOW - 1) / tdk::PROG_WINDOW;      // windows that hold a read
    *nwindows = n;
    // every window the caller has room for, whatever this handle's own read count says: a byte-sharded file's later
    // shards hold reads of high ordinals only (windows the handle never touched read as zero)
    const uint64_t take = std::min<uint64_t>(cap, h->d_win.n);
    std::vector<unsigned long long> w(take);
    if (take) HIPCHK(hipMemcpy(w.data(), h->d_win.p, take * 8, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < cap; i++) {
        const unsigned long long v = i < take ? w[i] : 0ull;
        out[2 * i] = v & 0xFFFFFFFFull; out[2 * i + 1] = v >> 32;
    }
    return TD_OK;
}

int td_split_progress(td_handle *h, uint64_t *out, uint64_t cap, uint64_t *nwindows) {
    if (!h || !nwindows || (cap && !out)) return fail(TD_E_ARG, "NULL argument");
    const uint64_t n = h->split_win.size() / 2;
    *nwindows = n;
    for (uint64_t i = 0; i < std::min(n, cap) * 2; i++) out[i] = h->split_win[i];
    return TD_OK;
}

int td_get_counts(td_handle *h, uint64_t *out) {
    if (!h || !out) return fail(TD_E_ARG, "NULL argument");
    if (!h->have_index) return fail(TD_E_STATE, "td_set_index has not been called");
    uint64_t st[TD_STAT_NSTATS];
    int rc = td_get_stats(h, st);
    if (rc) return rc;
    const size_t cells = (size_t)h->barnum * h->ntags;
    std::vector<uint32_t> tmp(cells);
    const uint32_t *src = h->bound_counts ? h->bound_counts : h->d_counts.p;
    HIPCHK(hipMemcpy(tmp.data(), src, cells * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cells; i++) out[i] = (uint64_t)tmp[i] + (h->host_acc.size() == cells ? h->host_acc[i] : 0);
    if (h->used64 && h->d_counts64.p) {
        std::vector<unsigned long long> t64(cells);
        HIPCHK(hipMemcpy(t64.data(), h->d_counts64.p, cells * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cells; i++) out[i] += t64[i];
    }
    return TD_OK;
}

int td_debug_counters(td_handle *h, uint64_t out[24]) {
    if (!h || !out) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, h->d_stats.p + 8, 24 * 8, hipMemcpyDeviceToHost));
    if (h->d_nfix.p) {          // [12]: length of the fast path's fix-up queue in the last launch
        uint32_t nf = 0;
        HIPCHK(hipMemcpy(&nf, h->d_nfix.p, 4, hipMemcpyDeviceToHost));
        out[11] = nf;
    }
    return TD_OK;
}

int td_set_option(td_handle *h, const char *name, int64_t value) {
    if (!h || !name) return fail(TD_E_ARG, "NULL argument");
    std::string n(name);
    if (n == "tile_kb") {
        if (value != 16 && value != 32) return fail(TD_E_ARG, "tile_kb must be 16 or 32");
        if (h->have_index && lds_bytes(h, (int)value) > LDS_BUDGET) return fail(TD_E_LIMIT, "tile does not fit the LDS budget with this index");
        h->tile_kb = (int)value;
    } else if (n == "blocks_per_cu") h->blocks_per_cu = (int)value;
    else if (n == "prescan") h->prescan = value ? 1 : 0;
    else if (n == "timing") h->timing = value ? 1 : 0;
    else if (n == "fastpath") h->fastpath = value ? 1 : 0;
    else if (n == "kernel") {
        if (value != 1 && value != 2 && value != 4) return fail(TD_E_ARG, "kernel must be 1 (k_fast), 2 (k_fast2) or 4 (k_fast4)");
        h->kernel_gen = (int)value;
    } else if (n == "tile_kb2") {
        if (value != 0 && value != 16 && value != 24 && value != 32) return fail(TD_E_ARG, "tile_kb2 must be 0 (automatic), 16, 24 or 32");
        if (value && h->have_index && lds_bytes_fast2(h, (int)value) > LDS_BUDGET) return fail(TD_E_LIMIT, "tile does not fit the LDS budget with this index");
        h->tile_kb2 = (int)value;
    } else if (n == "f4_nprod") {
        if (value != 0 && (value < tdk::F4_NPROD_MIN || value > tdk::F4_NPROD_MAX)) return fail(TD_E_ARG, "f4_nprod: 7 .. 13 producer waves of k_fast4's sixteen (0: from the input's line density)");
        h->f4_nprod = (int)value;
    } else if (n == "hot_cache") h->hot_cache = value == 2 ? 2 : value ? 1 : 0;
    else if (n == "run") h->run = (int)std::max<int64_t>(1, std::min<int64_t>(value, 4096));
    else if (n == "progress") h->progress = value ? 1 : 0;
    else if (n == "zb_members") h->zb_members = (uint32_t)std::max<int64_t>(64, value);
    else if (n == "split_kernel") h->split_kernel = value == 1 ? 1 : 2;
    else if (n == "gpu_huffman") h->gpu_huffman = value ? 1 : 0;
    else if (n == "gz_gpu_min") h->gz_gpu_min = (uint64_t)std::max<long long>(0, value);
    else if (n == "gz_gpu_terr_kb") { if (value < 16 || value > 4096) return fail(TD_E_ARG, "gz_gpu_terr_kb: 16..4096"); h->gz_gpu_terr_kb = (uint32_t)value; }
    else if (n == "gz_gpu_release") { h->gzgpu.release(); }
    else if (n == "gz_gpu_verify") h->gz_gpu_verify = value ? 1 : 0;
    else if (n == "gz_gpu_seg_kb") { if (value < 64 || value > (4 << 20)) return fail(TD_E_ARG, "gz_gpu_seg_kb: 64..4194304"); h->gz_gpu_seg_kb = (uint32_t)value; }
    else if (n == "gz_gpu_margin_kb") { if (value < 64 || value > (1 << 20)) return fail(TD_E_ARG, "gz_gpu_margin_kb: 64..1048576"); h->gz_gpu_margin_kb = (uint32_t)value; }
    else if (n == "gz_gpu_false_every") h->gz_gpu_false_every = (int)std::max<long long>(0, value);
    else if (n == "gpu_inflate") h->gpu_inflate = value ? 1 : 0;
    else if (n == "gpu_inflate_crc") h->gpu_inflate_crc = value ? 1 : 0;
    else if (n == "gpu_resolve") h->gpu_resolve = value ? 1 : 0;
    else if (n == "stagger") h->stagger = (int)value;
    else if (n == "prio") h->prio = (int)value & 0xFFFF;
    else if (n == "table_load_pct") h->table_load = std::max<int64_t>(10, std::min<int64_t>(value, 95)) / 100.0;
    else if (n == "nt_loads") h->nt_loads = value ? 1 : 0;
    else if (n == "fast_max_matrix_bytes")      // (tests: force the switch to the exact kernel; 0 = the built-in 4 GiB)
        h->fast_max_matrix = value > 0 ? std::min<uint64_t>((uint64_t)value, 1ull << 32) : 1ull << 32;
    else if (n == "flush_limit")                // (tests: flush the uint32 matrix to the host accumulator early; 0 = the built-in 2^32 - 1)
        h->flush_limit = value > 0 ? std::min<uint64_t>((uint64_t)value, 0xFFFFFFFFull) : 0xFFFFFFFFull;
    else if (n == "debug_ablate") h->debug_ablate = (uint32_t)value;   // timing-only ablations, wrong results
    else return fail(TD_E_ARG, "unknown option " + n);
    return TD_OK;
}

int td_kernel_time_ms(td_handle *h, double *ms, uint32_t *launches) {
    if (!h || !ms) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    double tot = 0;
    for (size_t i = 0; i < h->ev_used; i++) {
        HIPCHK(hipEventSynchronize(h->ev_pool[i].second));
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, h->ev_pool[i].first, h->ev_pool[i].second));
        tot += t;
    }
    *ms = h->ev_used ? tot / (double)h->ev_used : 0.0;
    if (launches) *launches = (uint32_t)h->ev_used;
    h->ev_used = 0;
    return TD_OK;
}

int td_kernel_times_ms(td_handle *h, double *out, uint32_t capacity, uint32_t *launches) {
    if (!h || !out || !launches) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    uint32_t n = 0;
    for (size_t i = 0; i < h->ev_used && n < capacity; i++, n++) {
        HIPCHK(hipEventSynchronize(h->ev_pool[i].second));
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, h->ev_pool[i].first, h->ev_pool[i].second));
        out[n] = t;
    }
    *launches = n;
    h->ev_used = 0;
    return TD_OK;
}

int td_dev_alloc(td_handle *h, uint64_t nbytes, void **out) {
    if (!h || !out) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMalloc(out, std::max<uint64_t>(nbytes, 16)));
    return TD_OK;
}
int td_dev_free(td_handle *h, void *p) {
    if (!h) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    if (p) HIPCHK(hipFree(p));
    return TD_OK;
}
int td_memcpy_h2d(td_handle *h, void *dst, const void *src, uint64_t n) {
    if (!h) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    if (n) HIPCHK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
    return TD_OK;
}
int td_memcpy_d2h(td_handle *h, void *dst, const void *src, uint64_t n) {
    if (!h) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    if (n) HIPCHK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return TD_OK;
}
int td_device_sync(td_handle *h) {
    if (!h) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    return TD_OK;
}

}  // extern "C"

// ---- the per-lane DEFLATE decoder of gpu_inflate.hpp, run on the host (tests: against zlib, without a GPU)
extern "C" int td_inflate_raw_host(const void *in, uint32_t in_len, void *out, uint32_t out_len) {
    std::vector<uint8_t> padded((size_t)in_len + 16, 0);
    if (in_len) memcpy(padded.data(), in, in_len);
    std::vector<uint16_t> tab(tdinf::TABLE_U16);
    std::vector<uint8_t> scratch(tdinf::SCRATCH_BYTES);
    tdinf::Stream s{};
    s.in = padded.data(); s.in_len = in_len; s.out = (uint8_t *)out; s.out_len = out_len;
    s.tab = tab.data(); s.tstride = 1; s.scratch = scratch.data();
    return (int)tdinf::run(s);
}

// ---- a row of the count matrix as the text csv.writer gives it (decimal integers, commas): the 38 M cells of a
// 384 x 100 k matrix take the reference's writeCounts (tagdigger_fun.py:1100-1111) half a minute through Python ints
extern "C" int64_t td_format_csv_row(const int64_t *vals, uint64_t n, char *out, uint64_t capacity) {
    char *o = out, *const end = out + capacity;
    for (uint64_t i = 0; i < n; i++) {
        if (end - o < 24) return -1;
        int64_t v = vals[i];
        if (i) *o++ = ',';
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        if (v < 0) *o++ = '-';
        char tmp[20];
        int k = 0;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        while (k) *o++ = tmp[--k];
    }
    return (int64_t)(o - out);
}

// ---- K3: barcode rows of one library -> sample rows of the run (reference combineReadCounts,
// tagdigger_fun.py:1061-1098: equal sample names are summed, on the host there, here on the device so that the
// matrix that is all-reduced over RCCL is the samples x tags one and nothing passes through host lists)
namespace {
__global__ __launch_bounds__(256) void k_fold_rows(const uint32_t *src, uint32_t barnum, uint32_t ncols, const uint32_t *row_of,
                                                   uint32_t *dst) {
    // one workgroup column-strip per source row: dst[row_of[b]][c] += src[b][c]; two barcodes of one library may
    // carry the same sample name, so the additions are atomic (no-return, one per non-zero cell)
    const uint32_t b = blockIdx.y;
    const uint32_t *s = src + (size_t)b * ncols;
    uint32_t *d = dst + (size_t)row_of[b] * ncols;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) {
        const uint32_t v = s[c];
        if (v) atomicAdd(d + c, v);
    }
    (void)barnum;
}
}  // namespace

extern "C" {

int td_fold_rows(td_handle *h, const uint32_t *row_of_barcode, uint32_t n_dst_rows, void *d_dst, void *stream) {
    if (!h || !row_of_barcode || !d_dst) return fail(TD_E_ARG, "NULL argument");
    if (!h->have_index) return fail(TD_E_STATE, "td_set_index has not been called");
    HIPCHK(hipSetDevice(h->device));
    if (h->used64) return fail(TD_E_STATE, "td_fold_rows folds the uint32 matrix; tassel_tagcount weights are 64-bit");
    bool flushed = false;                      // (a library of more than ~17 GB: part of its counts sits in the host accumulator)
    for (uint64_t v : h->host_acc) if (v) { flushed = true; break; }
    for (uint32_t b = 0; b < h->barnum; b++)
        if (row_of_barcode[b] >= n_dst_rows) return fail(TD_E_ARG, "row_of_barcode entry beyond n_dst_rows");
    hipStream_t s = (hipStream_t)stream;
    // every launch enqueued through this handle first (they may be on its own streams), and what they flagged
    HIPCHK(hipStreamSynchronize(h->work_stream));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    int rc = h->d_rowmap.ensure(h->barnum); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->d_rowmap.p, row_of_barcode, (size_t)h->barnum * 4, hipMemcpyHostToDevice, s));
    const uint32_t *src = h->bound_counts ? h->bound_counts : h->d_counts.p;
    const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((h->ntags + 255) / 256, 64));
    hipLaunchKernelGGL(k_fold_rows, dim3(gx, h->barnum), dim3(256), 0, s, src, h->barnum, h->ntags, h->d_rowmap.p, (uint32_t *)d_dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    if (flushed) {
        // the flushed part goes the same way: its low 32 bits (the destination's cells are uint32) through the same kernel
        const size_t cells = (size_t)h->barnum * h->ntags;
        std::vector<uint32_t> low(cells);
        for (size_t i = 0; i < cells; i++) low[i] = (uint32_t)h->host_acc[i];
        DevBuf<uint32_t> d_low;
        rc = d_low.ensure(cells); if (rc) return rc;
        hipError_t e = hipMemcpy(d_low.p, low.data(), cells * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_fold_rows, dim3(gx, h->barnum), dim3(256), 0, s, d_low.p, h->barnum, h->ntags, h->d_rowmap.p, (uint32_t *)d_dst);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
        d_low.release();
        if (e != hipSuccess) return fail(TD_E_HIP, std::string("td_fold_rows (flushed part): ") + hipGetErrorString(e));
    }
    unsigned long long st[TD_STAT_NSTATS];
    HIPCHK(hipMemcpy(st, h->d_stats.p, sizeof(st), hipMemcpyDeviceToHost));
    return check_device_errors(h, st);
}

}  // extern "C"

// ---- synthetic FASTQ generator (bench / tests only)
namespace {
__global__ void k_synth(td_synth_params P, uint64_t first_read, uint64_t nreads, const char *bar_tab,
                        const uint8_t *bar_len, const char *cut_tab, const char *tag_tab,
                        const uint16_t *tag_len, uint8_t *out) {
    const uint64_t rb = td_synth_record_bytes(P.read_len);
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (uint64_t)gridDim.x * blockDim.x)
        td_synth_record(&P, first_read + r, bar_tab, bar_len, cut_tab, tag_tab, tag_len, out + r * rb);
}
// the count matrix the generator's own choices imply (td_synth_hit): no FASTQ is parsed
__global__ void k_synth_expected(td_synth_params P, uint64_t first_read, uint64_t nreads, uint32_t *counts,
                                 unsigned long long *hits) {
    unsigned long long mine = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t j, k;
        if (td_synth_hit(&P, first_read + r, &j, &k)) { atomicAdd(counts + (size_t)j * P.ntags + k, 1u); mine++; }
    }
    if (mine) atomicAdd(hits, mine);
}
}  // namespace

extern "C" {

// the generator's optional draw tables (td_synth_spec.h "skew"): host thresholds -> device copies, pointers swapped in P
static int synth_upload_cdfs(td_synth_params &P, uint64_t *&d_tag_cdf, uint64_t *&d_bar_cdf) {
    d_tag_cdf = d_bar_cdf = nullptr;
    if (P.tag_cdf) {
        HIPCHK(hipMalloc((void **)&d_tag_cdf, (size_t)P.ntags * 8));
        HIPCHK(hipMemcpy(d_tag_cdf, P.tag_cdf, (size_t)P.ntags * 8, hipMemcpyHostToDevice));
        P.tag_cdf = d_tag_cdf;
    }
    if (P.bar_cdf) {
        HIPCHK(hipMalloc((void **)&d_bar_cdf, (size_t)P.nbar * 8));
        HIPCHK(hipMemcpy(d_bar_cdf, P.bar_cdf, (size_t)P.nbar * 8, hipMemcpyHostToDevice));
        P.bar_cdf = d_bar_cdf;
    }
    return TD_OK;
}

int td_synth_fill_device(td_handle *h, const void *params, uint64_t first_read, uint64_t nreads,
                         const char *bar_tab, const uint8_t *bar_len, const char *cut_tab,
                         const char *tag_tab, const uint16_t *tag_len, void *d_out, void *stream) {
    if (!h || !params || !d_out) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    td_synth_params P = *(const td_synth_params *)params;
    if (P.adapter_len > TD_SYNTH_ADAPTER_MAX) return fail(TD_E_ARG, "adapter_len beyond TD_SYNTH_ADAPTER_MAX");
    hipStream_t s = (hipStream_t)stream;
    char *d_bar = nullptr, *d_cut = nullptr, *d_tag = nullptr; uint8_t *d_bl = nullptr; uint16_t *d_tl = nullptr;
    uint64_t *d_tc = nullptr, *d_bc = nullptr;
    { int rc = synth_upload_cdfs(P, d_tc, d_bc); if (rc) return rc; }
    HIPCHK(hipMalloc((void **)&d_bar, (size_t)P.nbar * TD_SYNTH_BAR_STRIDE));
    HIPCHK(hipMalloc((void **)&d_bl, P.nbar));
    HIPCHK(hipMalloc((void **)&d_cut, (size_t)P.ncut * TD_SYNTH_CUT_STRIDE));
    HIPCHK(hipMalloc((void **)&d_tag, (size_t)P.ntags * P.tag_stride));
    HIPCHK(hipMalloc((void **)&d_tl, (size_t)P.ntags * 2));
    HIPCHK(hipMemcpy(d_bar, bar_tab, (size_t)P.nbar * TD_SYNTH_BAR_STRIDE, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_bl, bar_len, P.nbar, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_cut, cut_tab, (size_t)P.ncut * TD_SYNTH_CUT_STRIDE, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tag, tag_tab, (size_t)P.ntags * P.tag_stride, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tl, tag_len, (size_t)P.ntags * 2, hipMemcpyHostToDevice));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((nreads + 255) / 256, (uint64_t)h->num_cu * 32);
    if (nreads) hipLaunchKernelGGL(k_synth, dim3(grid), dim3(256), 0, s, P, first_read, nreads, d_bar, d_bl, d_cut, d_tag, d_tl, (uint8_t *)d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    (void)hipFree(d_bar); (void)hipFree(d_bl); (void)hipFree(d_cut); (void)hipFree(d_tag); (void)hipFree(d_tl);
    if (d_tc) (void)hipFree(d_tc);
    if (d_bc) (void)hipFree(d_bc);
    return TD_OK;
}

int td_synth_expected_device(td_handle *h, const void *params, uint64_t first_read, uint64_t nreads,
                             uint32_t *d_counts, uint64_t *hits_out, void *stream) {
    if (!h || !params || !d_counts) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    td_synth_params P = *(const td_synth_params *)params;
    hipStream_t s = (hipStream_t)stream;
    uint64_t *d_tc = nullptr, *d_bc = nullptr;
    { int rc = synth_upload_cdfs(P, d_tc, d_bc); if (rc) return rc; }
    unsigned long long *d_hits = nullptr;
    HIPCHK(hipMalloc((void **)&d_hits, 8));
    HIPCHK(hipMemsetAsync(d_hits, 0, 8, s));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((nreads + 255) / 256, (uint64_t)h->num_cu * 32);
    if (nreads) hipLaunchKernelGGL(k_synth_expected, dim3(grid), dim3(256), 0, s, P, first_read, nreads, d_counts, d_hits);
    HIPCHK(hipGetLastError());
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, d_hits, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    (void)hipFree(d_hits);
    if (d_tc) (void)hipFree(d_tc);
    if (d_bc) (void)hipFree(d_bc);
    if (hits_out) *hits_out = v;
    return TD_OK;
}

}  // extern "C"

// ================================================================ barcode splitter
// td_set_splitter: the index.  td_split_device: the per-read decisions for a buffer in HBM.
// td_split_file: the reference's barcodeSplitter loop (tagdigger_fun.py:1318-1368) -- the GPU decides
// every read, the host writes the clipped records.
namespace {

// line terminators before the end of every 16 KB tile -> d_state, their total -> d_cursor[0] (async on `s`)
// k_split2 (24 KiB tiles staged in LDS) where the two restriction sites are spelled in ACGT and the index fits beside
// the tile; k_split (16 KiB tiles, lines walked in global memory) otherwise, or on request (option split_kernel = 1)
size_t lds_bytes_split2(const td_handle *h) {
    return (size_t)24 * 1024 + tdk::SPLIT2_HALO + 64 + (size_t)6 * tdk::FBLOCK * 2 + 256 + h->sp_bblob_bytes;
}
bool use_split2(const td_handle *h) {
    return h->split_kernel == 2 && h->sp_sites_acgt && h->sp_compact && lds_bytes_split2(h) <= (size_t)40 * 1024;
}

int launch_split_prefix(td_handle *h, const void *d_fastq, uint64_t nbytes, hipStream_t s, bool counted = false) {
    const uint64_t tile = use_split2(h) ? 24 * 1024 : 16 * 1024;
    const uint64_t nt = (nbytes + tile - 1) / tile;
    if (nt > 0x7FFFFFFFull) return fail(TD_E_LIMIT, "buffer too large for one launch; split it");
    int rc = h->d_tilecounts.ensure(nt); if (rc) return rc;
    rc = h->d_state.ensure(nt); if (rc) return rc;
    const uint32_t g = (uint32_t)std::min<uint64_t>(nt, (uint64_t)h->num_cu * 8);
    // `counted`: the count pass just enqueued on this stream ran over the same bytes in tiles of this size and left
    // every tile's terminator count in d_tileinfo
    if (counted && use_split2(h) && h->last_fast_tile_kb == 24 && h->last_fast_ntiles == (uint32_t)nt)
        hipLaunchKernelGGL(tdk::k_info_counts, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, s, h->d_tileinfo.p, (uint32_t)nt, h->d_tilecounts.p);
    else if (use_split2(h)) hipLaunchKernelGGL((tdk::k_count_lines<6>), dim3(g), dim3(tdk::BLOCK), 0, s, (const uint8_t *)d_fastq, nbytes, (uint32_t)nt, h->d_tilecounts.p);
    else hipLaunchKernelGGL((tdk::k_count_lines<4>), dim3(g), dim3(tdk::BLOCK), 0, s, (const uint8_t *)d_fastq, nbytes, (uint32_t)nt, h->d_tilecounts.p);
    hipLaunchKernelGGL(tdk::k_scan_tiles, dim3(1), dim3(1024), 0, s, h->d_tilecounts.p, (uint32_t)nt, h->d_state.p, h->d_cursor.p);
    HIPCHK(hipGetLastError());
    return TD_OK;
}

// k_split on `s` (after launch_split_prefix); out must hold one int2 per sequence line of the buffer
int launch_split(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, int2 *d_out, hipStream_t s) {
    const bool two = use_split2(h);
    const uint64_t tile = two ? 24 * 1024 : 16 * 1024;
    const uint64_t nt = (nbytes + tile - 1) / tile;
    const uint32_t g = (uint32_t)std::min<uint64_t>(nt, (uint64_t)h->num_cu * (two ? 4 : 8));
    tdk::SplitParams sp{};
    sp.buf = (const uint8_t *)d_fastq; sp.nbytes = nbytes; sp.first_line = first_line;
    sp.prefix = h->d_state.p; sp.ntiles = (uint32_t)nt;
    sp.bblob = h->d_sp_bblob.p; sp.bblob_bytes = h->sp_bblob_bytes; sp.off_bmeta = h->sp_off_bmeta; sp.off_bdir = h->sp_off_bdir;
    sp.cutlen = h->sp_cutlen;
    sp.site0 = h->sp_site[0]; sp.site1 = h->sp_site[1]; sp.site0_len = h->sp_site_len[0]; sp.site1_len = h->sp_site_len[1];
    sp.ent_begin = h->d_sp_ent_begin.p; sp.ent_group = h->d_sp_ent_group.p; sp.entries = h->d_sp_entries.p; sp.pool = h->d_sp_pool.p;
    sp.gcap = h->sp_gcap; sp.entries16 = h->d_sp_entries16.p;
    sp.entries8 = h->d_sp_e8.p; sp.pool2 = h->d_sp_pool2.p;
    sp.out = d_out; sp.stats = h->d_stats.p; sp.dbg = (uint32_t)h->debug_ablate;
    if (two) {
        hipLaunchKernelGGL((tdk::k_split2<6>), dim3(g), dim3(tdk::FBLOCK), lds_bytes_split2(h), s, sp);
        HIPCHK(hipGetLastError());
        return TD_OK;
    }
    const size_t lds = (size_t)4 * tdk::BLOCK * 2 + 64 + h->sp_bblob_bytes;
    hipLaunchKernelGGL((tdk::k_split<4>), dim3(g), dim3(tdk::BLOCK), lds, s, sp);
    HIPCHK(hipGetLastError());
    return TD_OK;
}

// upper bound of the sequence lines among `lines` consecutive lines
inline uint64_t max_seq_lines(uint64_t lines) { return lines / 4 + 2; }

inline bool host_blank(uint8_t b) { return b == 0x20 || b == 0x09 || b == 0x0B || b == 0x0C || (b >= 0x1C && b <= 0x1F); }

// The reference's record loop (:1328-1363) on the host side: lines as text mode yields them,
// str.strip() on each, the clipped record written to its barcode's file.
// The output side shared by the writer threads: barcode b's file and buffer belong to thread
// b % nthreads alone, so nothing is locked.
struct SplitFiles {
    const std::vector<std::string> &barcodes;
    std::vector<FILE *> out;
    std::vector<std::string> pend;            // per output file: bytes not yet handed to stdio
    std::vector<uint8_t> owner;               // per barcode: the writer thread its file belongs to
    explicit SplitFiles(const std::vector<std::string> &b) : barcodes(b) {}
    ~SplitFiles() { for (FILE *f : out) if (f) fclose(f); }
};

// The lines of a piece, found once by all writer threads together (each scans 1/n of the bytes):
// line k is p[starts[k], starts[k + 1] - 1).  Pieces holding a '\r' are walked byte by byte instead.
struct PieceIndex {
    std::vector<std::vector<uint32_t>> part;     // per thread: starts of the lines that begin in its range
    std::vector<uint32_t> starts;
    size_t nlines = 0;
    bool has_cr = false, nonascii = false;
    std::vector<uint8_t> flags;                  // per thread: 1 = saw '\r', 2 = saw a byte >= 0x80
    // whole records dealt out: bucket[s * nthreads + o] = the records of scanner s's share that writer o owns
    std::vector<std::vector<uint32_t>> bucket;
    // a barrier for the threads of one piece
    std::mutex mu; std::condition_variable cv; uint32_t arrived = 0, generation = 0, nthreads = 1;
    void barrier() {
        std::unique_lock<std::mutex> g(mu);
        const uint32_t gen = generation;
        if (++arrived == nthreads) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(g, [&]() { return generation != gen; });
    }
    void scan(uint32_t tid, const uint8_t *p, size_t n) {
        const size_t a = n * tid / nthreads, b = n * (tid + 1) / nthreads;
        std::vector<uint32_t> &v = part[tid];
        v.clear();
        uint8_t any = 0;
        for (size_t i = a; i < b; i++) any |= p[i];                         // (vectorises)
        flags[tid] = (uint8_t)((memchr(p + a, '\r', b - a) ? 1 : 0) | (any & 0x80 ? 2 : 0));
        for (size_t i = a; i < b;) {
            const uint8_t *nl = (const uint8_t *)memchr(p + i, '\n', b - i);
            if (!nl) break;
            i = (size_t)(nl - p) + 1;
            v.push_back((uint32_t)i);
        }
    }
    void merge(size_t n) {                        // (one thread, between two barriers)
        has_cr = false; nonascii = false;
        for (uint8_t f : flags) { has_cr |= (f & 1) != 0; nonascii |= (f & 2) != 0; }
        starts.clear();
        if (n == 0) { nlines = 0; return; }
        starts.push_back(0);
        for (const auto &v : part) starts.insert(starts.end(), v.begin(), v.end());
        if (starts.back() == n) nlines = starts.size() - 1;               // the piece ends with a terminator
        else { nlines = starts.size(); starts.push_back((uint32_t)n + 1); }   // ... or with an unterminated last line
    }
};

// One writer thread.  All threads agree on line numbers and on where maxreads stops (each steps
// through every record of a piece), but a thread copies, clips and writes only the records of its
// own barcodes.
struct SplitWriter {
    SplitFiles &files;
    const std::vector<std::string> &barcodes;
    std::vector<FILE *> &out;
    std::vector<std::string> &pend;
    uint32_t tid = 0, nthreads = 1;
    static constexpr size_t FLUSH_AT = 64 << 10;    // (x several hundred files: keep the buffers cache-resident)
    uint64_t lineindex = 0, reads = 0, barcut = 0, clipped = 0, max_reads = 0;
    std::string comment1, sequence, comment2, quality;
    int cur_bar = -1, cur_slice = 999;
    bool stop = false, nonascii = false, io_error = false;
    // progress windows (reference :1357-1360 prints its counters every 50 000 reads): of the reads this thread looked
    // at for that purpose, per window of 50 000, how many had a barcode and how many were clipped
    std::vector<uint64_t> wbar, wclip;
    void note(uint64_t read0, bool clip) {
        const size_t w = (size_t)(read0 / 50000);
        if (w >= wbar.size()) { wbar.resize(w + 1, 0); wclip.resize(w + 1, 0); }
        wbar[w]++;
        if (clip) wclip[w]++;
    }

    SplitWriter(SplitFiles &f, uint32_t t, uint32_t n) : files(f), barcodes(f.barcodes), out(f.out), pend(f.pend), tid(t), nthreads(n) {}
    bool mine() const { return cur_bar > -1 && files.owner[(size_t)cur_bar] == tid; }

    static void stripped(const uint8_t *p, size_t n, std::string &dst, bool upper) {
        size_t a = 0, b = n;
        while (a < b && host_blank(p[a])) a++;
        while (b > a && host_blank(p[b - 1])) b--;
        dst.assign((const char *)p + a, b - a);
        if (upper) for (char &c : dst) c = (char)(c - (((unsigned)(c - 'a') < 26u) << 5));     // (branch-free: vectorises)
    }
    // s[a:b] with Python's rules for a >= 0 and any b
    static void put_slice(std::string &o, const std::string &s, long a, long b) {
        const long n = (long)s.size();
        if (b < 0) { b += n; if (b < 0) b = 0; }
        if (b > n) b = n;
        if (a < b) o.append(s.data() + a, (size_t)(b - a));
        o.push_back('\n');
    }
    bool flush(size_t k) {
        std::string &o = pend[k];
        static const bool discard = getenv("TAGDIG_SPLIT_DISCARD") != nullptr;      // (timing only: assemble the records, write nothing)
        const bool ok = o.empty() || discard || fwrite(o.data(), 1, o.size(), out[k]) == o.size();
        o.clear();
        return ok;
    }
    // one line (terminator excluded); `res` advances over the piece's sequence-line results
    void line(const uint8_t *p, size_t n, const int2 *&res) {
        switch (lineindex & 3) {
        case 0: stripped(p, n, comment1, false); break;
        case 1: cur_bar = res->x; cur_slice = res->y; res++; if (mine()) stripped(p, n, sequence, true); break;
        case 2: if (mine()) stripped(p, n, comment2, false); break;
        default: {
            reads++;
            if (tid == 0 && cur_bar > -1) note(reads - 1, cur_slice != 999);      // (every thread sees this line: one notes it)
            if (mine()) {
                stripped(p, n, quality, false);
                barcut++;
                const std::string &bc = barcodes[(size_t)cur_bar];
                const long slice1 = (long)bc.size();
                long slice2 = cur_slice;
                if (slice2 == 999) slice2 = (long)sequence.size(); else clipped++;
                std::string &o = pend[(size_t)cur_bar];
                o.append(comment1); o.append(bc); o.push_back('\n');
                put_slice(o, sequence, slice1, slice2);
                if (comment2 == "+") o.append("+\n");
                else { o.append(comment1); o.append(bc); o.push_back('\n'); }
                put_slice(o, quality, slice1, slice2);
                if (o.size() >= FLUSH_AT && !flush((size_t)cur_bar)) io_error = true;
            }
            if (reads >= max_reads) stop = true;
        } }
        lineindex++;
    }
    // one whole record whose four lines are lines k .. k + 3 of the piece
    // (the same from the piece's bytes directly: no intermediate strings)
    struct Span { const uint8_t *p; size_t n; };
    static Span strip(const uint8_t *p, size_t n) {
        size_t a = 0, b = n;
        while (a < b && host_blank(p[a])) a++;
        while (b > a && host_blank(p[b - 1])) b--;
        return {p + a, b - a};
    }
    static void put_slice(std::string &o, Span s, long a, long b, bool upper) {
        const long n = (long)s.n;
        if (b < 0) { b += n; if (b < 0) b = 0; }
        if (b > n) b = n;
        if (a < b) {
            const size_t at = o.size();
            o.append((const char *)s.p + a, (size_t)(b - a));
            if (upper) for (size_t i = at; i < o.size(); i++) { const char c = o[i]; o[i] = (char)(c - (((unsigned)(c - 'a') < 26u) << 5)); }
        }
        o.push_back('\n');
    }
    void own_record(const uint8_t *p, const uint32_t *st, const int2 &d) {
        cur_bar = d.x; cur_slice = d.y;
        {
            const Span c1 = strip(p + st[0], st[1] - 1 - st[0]), sq = strip(p + st[1], st[2] - 1 - st[1]),
                       c2 = strip(p + st[2], st[3] - 1 - st[2]), ql = strip(p + st[3], st[4] - 1 - st[3]);
            barcut++;
            const std::string &bc = barcodes[(size_t)cur_bar];
            const long slice1 = (long)bc.size();
            long slice2 = cur_slice;
            if (slice2 == 999) slice2 = (long)sq.n; else clipped++;
            std::string &o = pend[(size_t)cur_bar];
            o.append((const char *)c1.p, c1.n); o.append(bc); o.push_back('\n');
            put_slice(o, sq, slice1, slice2, true);
            if (c2.n == 1 && c2.p[0] == '+') o.append("+\n");
            else { o.append((const char *)c1.p, c1.n); o.append(bc); o.push_back('\n'); }
            put_slice(o, ql, slice1, slice2, false);
            if (o.size() >= FLUSH_AT && !flush((size_t)cur_bar)) io_error = true;
        }
    }
    // all lines of a piece that ends at a line end (or at the end of the file)
    double t_index = 0, t_records = 0;           // (thread 0's, for TAGDIG_SPLIT_TIMING)
    void piece(const uint8_t *p, size_t n, const int2 *res, PieceIndex &ix) {
        const auto clk = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = clk();
        ix.scan(tid, p, n);
        ix.barrier();
        if (tid == 0) ix.merge(n);
        ix.barrier();
        const double t1 = clk();
        t_index += t1 - t0;
        struct Done { double &acc; double from; decltype(clk) &c; ~Done() { acc += c() - from; } } done{t_records, t1, clk};
        if (tid == 0 && ix.nonascii) nonascii = true;
        if (!ix.has_cr) {                                                // the usual file: '\n' only
            const uint32_t *st = ix.starts.data();
            size_t k = 0;
            // the lines of a record begun in the piece before (every thread walks them: all keep the same state)
            while (k < ix.nlines && (lineindex & 3) != 0 && !stop) { line(p + st[k], st[k + 1] - 1 - st[k], res); k++; }
            // whole records: each thread sorts its share of them by owner, then every writer takes its own
            size_t nrec = stop ? 0 : (ix.nlines - k) / 4;
            if (max_reads - reads < nrec) nrec = (size_t)(max_reads - reads);          // (reads < max_reads while not stopped)
            const size_t lo = nrec * tid / nthreads, hi = nrec * (tid + 1) / nthreads;
            for (uint32_t o = 0; o < nthreads; o++) ix.bucket[(size_t)tid * nthreads + o].clear();
            for (size_t j = lo; j < hi; j++) {
                const int b = res[j].x;
                if (b > -1) { ix.bucket[(size_t)tid * nthreads + files.owner[(size_t)b]].push_back((uint32_t)j); note(reads + j, res[j].y != 999); }
            }
            ix.barrier();
            for (uint32_t sc = 0; sc < nthreads; sc++)
                for (const uint32_t j : ix.bucket[(size_t)sc * nthreads + tid]) own_record(p, st + k + 4 * (size_t)j, res[j]);
            reads += nrec; lineindex += 4 * nrec; res += nrec; k += 4 * nrec;
            if (reads >= max_reads) stop = true;
            ix.barrier();                                                // (the buckets are refilled by the next piece)
            // the lines of a record cut by the piece's end (its barcode is not known yet: every thread keeps them)
            while (k < ix.nlines && !stop) { line(p + st[k], st[k + 1] - 1 - st[k], res); k++; }
            return;
        }
        size_t start = 0, i = 0;
        while (i < n && !stop) {
            const uint8_t c = p[i];
            if (c == '\n') { line(p + start, i - start, res); start = ++i; }
            else if (c == '\r') { line(p + start, i - start, res); i++; if (i < n && p[i] == '\n') i++; start = i; }
            else i++;
        }
        if (!stop && start < n) line(p + start, n - start, res);
    }
};

}  // namespace

extern "C" {

int td_set_splitter(td_handle *h, const char *const *barcodes, uint32_t nbar, const char *cutsite,
                    const char *fullsite0, const char *fullsite1, const uint32_t *ent_begin,
                    const char *const *ent_seq, const int32_t *ent_slice, uint32_t nent) {
    if (!h || !barcodes || !cutsite || !fullsite0 || !fullsite1 || !ent_begin) return fail(TD_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    h->have_splitter = false;
    if (nbar == 0) return fail(TD_E_EMPTY, "empty barcode list");
    const std::string cs(cutsite);
    std::vector<std::string> barcut(nbar);
    std::vector<uint32_t> barlen(nbar);
    h->sp_barcodes.assign(nbar, std::string());
    for (uint32_t i = 0; i < nbar; i++) {
        h->sp_barcodes[i] = barcodes[i];
        barcut[i] = h->sp_barcodes[i] + cs;
        for (char &c : barcut[i]) if (c >= 'a' && c <= 'z') c = (char)(c - 32);     // combine_barcode_and_cutsite upper-cases (:69)
        barlen[i] = (uint32_t)h->sp_barcodes[i].size();
        if (barcut[i].empty()) return fail(TD_E_LIMIT, "empty barcode with an empty cut site is not supported by the splitter");
    }
    Resolver rb(barcut, nbar);
    int rc = rb.run();
    if (rc) { g_bad = rb.bad; return fail(rc, rc == TD_E_OVERLAP ? "overlapping barcode+cutsite sequences" : "barcode index build failed"); }
    std::vector<uint8_t> blob;
    uint32_t max_off = 0;
    rc = build_barcode_blob(rb.out, nbar, barlen.data(), blob, h->sp_off_bmeta, h->sp_off_bdir, max_off);
    if (rc) return rc;
    h->sp_bblob_bytes = (uint32_t)blob.size();
    if ((size_t)4 * tdk::BLOCK * 2 + 64 + blob.size() > LDS_BUDGET) return fail(TD_E_LIMIT, "barcode index does not fit the LDS budget");
    h->sp_cutlen = (uint32_t)cs.size();
    const char *sites[2] = {fullsite0, fullsite1};
    h->sp_sites_acgt = true;
    for (int k = 0; k < 2; k++) {
        const size_t L = strlen(sites[k]);
        if (L > 8) return fail(TD_E_LIMIT, "restriction site longer than 8 bases");
        unsigned long long v = 0;
        for (size_t q = 0; q < L; q++) v = (v << 8) | (uint8_t)sites[k][q];
        h->sp_site[k] = v; h->sp_site_len[k] = (uint32_t)L;
        for (size_t q = 0; q < L; q++) if (!strchr("ACGT", sites[k][q])) h->sp_sites_acgt = false;
    }
    if (ent_begin[nbar] != nent) return fail(TD_E_ARG, "ent_begin[nbar] must equal nent");
    // per barcode the entries are stored by the code of their LAST base (A C T G = (byte >> 1) & 3): only
    // the group of the read's last base has to be looked at
    std::vector<tdk::SplitEntry> ents;
    std::vector<uint32_t> groups((size_t)nbar * 4, 0);
    std::vector<uint8_t> pool;
    ents.reserve(nent);
    // The entries' characters are interned: an entry that is a prefix of a string already in the pool points into it.
    // (What a read may end with are the beginnings of ONE adapter per cutter -- build_adapter_tree :1208-1249 -- so a
    // barcode's ~100 entries are prefixes of two strings, one of them the same for every barcode: the pool shrinks
    // from ~1 MB to ~25 KB at 384 barcodes, and the compare's loads find their lines in the caches.)
    std::unordered_map<std::string, uint32_t> pool_prefix;          // every prefix of every placed string -> its offset
    {
        std::vector<std::string> distinct;
        for (uint32_t e = 0; e < nent; e++) distinct.emplace_back(ent_seq[e]);
        std::sort(distinct.begin(), distinct.end(), [](const std::string &a, const std::string &c) { return a.size() != c.size() ? a.size() > c.size() : a < c; });
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        for (const std::string &str : distinct) {                    // longest first
            if (str.empty() || pool_prefix.count(str)) continue;
            const uint32_t off = (uint32_t)pool.size();
            pool.insert(pool.end(), str.begin(), str.end());
            for (size_t L = 1; L <= str.size(); L++) pool_prefix.emplace(str.substr(0, L), off);
        }
    }
    for (uint32_t b = 0; b < nbar; b++) {
        if (ent_begin[b] > ent_begin[b + 1] || ent_begin[b + 1] > nent) return fail(TD_E_ARG, "ent_begin must not decrease");
        for (uint32_t code = 0; code < 4; code++) {
            groups[(size_t)b * 4 + code] = (uint32_t)ents.size();
            for (uint32_t e = ent_begin[b]; e < ent_begin[b + 1]; e++) {
                const size_t L = strlen(ent_seq[e]);
                const uint32_t c = L ? (((uint8_t)ent_seq[e][L - 1] >> 1) & 3u) : 0u;
                if (c != code) continue;
                for (size_t q = 0; q < L; q++)
                    if (!strchr("ACGT", ent_seq[e][q])) return fail(TD_E_ALPHABET, "adapter entries must be upper-case ACGT");
                tdk::SplitEntry en;
                en.off = L ? pool_prefix.at(std::string(ent_seq[e], L)) : 0u; en.len = (uint32_t)L; en.slice = ent_slice[e];
                // its last (up to) four characters as the word read from a read's last four bytes holds them: the last
                // character in the top byte (k_split2 looks at the rest only when these agree)
                en.key = 0;
                for (size_t q = 0; q < std::min<size_t>(L, 4); q++) en.key |= (uint32_t)(uint8_t)ent_seq[e][L - 1 - q] << (8 * (3 - q));
                ents.push_back(en);
            }
        }
    }
    rc = h->d_sp_bblob.ensure(blob.size() / 4); if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_sp_bblob.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    rc = h->d_sp_ent_begin.ensure(nbar + 1); if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_sp_ent_begin.p, ent_begin, (size_t)(nbar + 1) * 4, hipMemcpyHostToDevice));
    {   // k_split2's view of the same entries: per barcode 16 groups by the codes of the LAST TWO characters (a one-character
        // entry sits in all four groups of its character), every group padded to the same capacity (empty entries: len 0),
        // so a read finds its group by address alone -- the loads can be issued before the group is needed -- and looks
        // at a sixteenth of its barcode's entries
        std::vector<std::vector<tdk::SplitEntry>> grp((size_t)nbar * 16);
        size_t cap = 4;
        for (uint32_t b = 0; b < nbar; b++) {
            const uint32_t lo = groups[(size_t)b * 4], hi = b + 1 < nbar ? groups[(size_t)(b + 1) * 4] : (uint32_t)ents.size();
            for (uint32_t g = 0; g < 16; g++) {
                auto &v = grp[(size_t)b * 16 + g];
                for (uint32_t k = lo; k < hi; k++) {
                    const tdk::SplitEntry &en = ents[k];
                    if (en.len == 0) continue;
                    const uint32_t last = ((en.key >> 24) >> 1) & 3u, prev = ((en.key >> 16) >> 1) & 3u;
                    if (last == (g >> 2) && (en.len == 1 || prev == (g & 3u))) v.push_back(en);
                }
                // longest first: a read that ran into the adapter usually carries a long piece of it
                std::stable_sort(v.begin(), v.end(), [](const tdk::SplitEntry &a, const tdk::SplitEntry &c) { return a.len > c.len; });
                cap = std::max(cap, (v.size() + 3) / 4 * 4);
            }
        }
        std::vector<tdk::SplitEntry> e16((size_t)nbar * 16 * cap, tdk::SplitEntry{0u, 0u, 999, 0u});
        for (size_t g = 0; g < grp.size(); g++) std::copy(grp[g].begin(), grp[g].end(), e16.begin() + g * cap);
        h->sp_gcap = (uint32_t)cap;
        rc = h->d_sp_entries16.ensure(std::max<size_t>(1, e16.size())); if (rc) return rc;
        HIPCHK(hipMemcpy(h->d_sp_entries16.p, e16.data(), e16.size() * sizeof(tdk::SplitEntry), hipMemcpyHostToDevice));
    }
    {   // k_split2's view (round 3): per barcode 64 groups by the codes of the LAST THREE characters (an entry shorter than
        // that sits in every group of its characters), eight compact entries of 8 bytes per group -- ONE 64-byte fetch per
        // read, issued as soon as its barcode is known.  An entry's characters are the first `len` of its master string
        // (the interned pool string it points into), kept at a fixed stride so that the compact entry needs no offset.
        // Entry sets that do not fit (a group of more than eight, strings beyond 128 characters, slices beyond a byte)
        // leave sp_compact false: the splitter then runs k_split.
        h->sp_compact = true;
        std::unordered_map<uint32_t, uint32_t> master_of;            // pool offset -> master index
        std::vector<uint8_t> pool2;
        auto master = [&](uint32_t off, uint32_t len) -> uint32_t {
            auto it = master_of.find(off);
            if (it != master_of.end()) {
                // (the longest entry of a master comes first -- entries are interned longest first --, but a later, longer
                // use of the same offset would need more characters than were copied: copy up to 128 from the pool anyway)
                return it->second;
            }
            const uint32_t idx = (uint32_t)master_of.size();
            master_of.emplace(off, idx);
            pool2.resize((size_t)(idx + 1) * 128 + 8, 0);
            for (uint32_t q = 0; q < 128 && off + q < pool.size(); q++) pool2[(size_t)idx * 128 + q] = pool[off + q];
            (void)len;
            return idx;
        };
        std::vector<uint2> e8((size_t)nbar * 64 * 8, make_uint2(0u, 0u));
        for (uint32_t b = 0; b < nbar && h->sp_compact; b++) {
            const uint32_t lo = groups[(size_t)b * 4], hi = b + 1 < nbar ? groups[(size_t)(b + 1) * 4] : (uint32_t)ents.size();
            for (uint32_t g = 0; g < 64 && h->sp_compact; g++) {
                std::vector<tdk::SplitEntry> v;
                for (uint32_t k = lo; k < hi; k++) {
                    const tdk::SplitEntry &en = ents[k];
                    if (en.len == 0) continue;
                    const uint32_t c1 = ((en.key >> 24) >> 1) & 3u, c2 = ((en.key >> 16) >> 1) & 3u, c3 = ((en.key >> 8) >> 1) & 3u;
                    if (c1 == (g >> 4) && (en.len < 2 || c2 == ((g >> 2) & 3u)) && (en.len < 3 || c3 == (g & 3u))) v.push_back(en);
                }
                std::stable_sort(v.begin(), v.end(), [](const tdk::SplitEntry &a, const tdk::SplitEntry &c) { return a.len > c.len; });
                if (v.size() > 8) { h->sp_compact = false; break; }
                for (size_t k = 0; k < v.size(); k++) {
                    if (v[k].len > 128 || v[k].slice < -128 || v[k].slice > 127 || master_of.size() >= 65535) { h->sp_compact = false; break; }
                    const uint32_t m = master(v[k].off, v[k].len);
                    e8[((size_t)b * 64 + g) * 8 + k] = make_uint2(v[k].key, v[k].len | (((uint32_t)v[k].slice & 0xFFu) << 8) | (m << 16));
                }
            }
        }
        if (h->sp_compact) {
            pool2.resize(pool2.size() + 128, 0);                     // (the compare reads eight bytes at a time)
            rc = h->d_sp_e8.ensure(e8.size()); if (rc) return rc;
            HIPCHK(hipMemcpy(h->d_sp_e8.p, e8.data(), e8.size() * sizeof(uint2), hipMemcpyHostToDevice));
            rc = h->d_sp_pool2.ensure(pool2.size()); if (rc) return rc;
            HIPCHK(hipMemcpy(h->d_sp_pool2.p, pool2.data(), pool2.size(), hipMemcpyHostToDevice));
        }
    }
    rc = h->d_sp_ent_group.ensure((size_t)nbar * 4); if (rc) return rc;
    HIPCHK(hipMemcpy(h->d_sp_ent_group.p, groups.data(), groups.size() * 4, hipMemcpyHostToDevice));
    rc = h->d_sp_entries.ensure(std::max<size_t>(1, nent)); if (rc) return rc;
    if (nent) HIPCHK(hipMemcpy(h->d_sp_entries.p, ents.data(), (size_t)nent * sizeof(tdk::SplitEntry), hipMemcpyHostToDevice));
    pool.insert(pool.end(), 8, 0);                        // (k_split2 compares eight bytes at a time: may read past the last string)
    rc = h->d_sp_pool.ensure(std::max<size_t>(1, pool.size())); if (rc) return rc;
    if (!pool.empty()) HIPCHK(hipMemcpy(h->d_sp_pool.p, pool.data(), pool.size(), hipMemcpyHostToDevice));
    rc = h->d_cursor.ensure(2); if (rc) return rc;
    h->have_splitter = true;
    return TD_OK;
}

static int split_device_impl(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, int32_t *d_out,
                             uint64_t out_capacity, void *stream, uint64_t *n_terminators, bool counted);

int td_split_device(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, int32_t *d_out,
                    uint64_t out_capacity, void *stream, uint64_t *n_terminators) {
    return split_device_impl(h, d_fastq, nbytes, first_line, d_out, out_capacity, stream, n_terminators, false);
}

// (counted: a count pass over the same bytes was just enqueued on `stream` -- its per-tile terminator counts serve)
static int split_device_impl(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, int32_t *d_out,
                             uint64_t out_capacity, void *stream, uint64_t *n_terminators, bool counted) {
    if (!h || !d_out) return fail(TD_E_ARG, "NULL argument");
    if (!h->have_splitter) return fail(TD_E_STATE, "td_set_splitter has not been called");
    if (((uintptr_t)d_fastq & 15) != 0) return fail(TD_E_ARG, "device FASTQ pointer must be 16-byte aligned");
    HIPCHK(hipSetDevice(h->device));
    if (n_terminators) *n_terminators = 0;
    if (nbytes == 0) return TD_OK;
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_split_prefix(h, d_fastq, nbytes, s, counted); if (rc) return rc;
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_cursor.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (n_terminators) *n_terminators = v;
    if (out_capacity < max_seq_lines(v + 1)) return fail(TD_E_ARG, "output must hold (terminators + 1) / 4 + 2 results");
    rc = launch_split(h, d_fastq, nbytes, first_line, (int2 *)d_out, s); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s));
    unsigned long long st[TD_STAT_NSTATS];
    HIPCHK(hipMemcpy(st, h->d_stats.p, sizeof(st), hipMemcpyDeviceToHost));
    return check_device_errors(h, st);
}

int td_count_and_split_device(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line, uint64_t max_reads,
                              int32_t *d_out, uint64_t out_capacity, void *stream, uint64_t *n_terminators) {
    // BASELINE config 5: the counting path and the adapter-trim branch over ONE buffer resident in HBM -- the count
    // pass is enqueued, the splitter's line prefix and decisions follow it on the same stream (the bytes are still
    // in the Infinity Cache for buffers that fit it); one synchronisation at the end
    if (!h || !d_out) return fail(TD_E_ARG, "NULL argument");
    if (!h->have_splitter) return fail(TD_E_STATE, "td_set_splitter has not been called");
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_count(h, d_fastq, nbytes, first_line, max_reads, 0, (hipStream_t)stream);
    if (rc) return rc;
    return split_device_impl(h, d_fastq, nbytes, first_line, d_out, out_capacity, stream, n_terminators, true);
}

static int split_file_impl(td_handle *h, const char *in_path, const char *const *out_paths, uint64_t max_reads, uint64_t stats[3],
                           bool by_reference_rules, bool *gz_refused);
int td_split_file(td_handle *h, const char *in_path, const char *const *out_paths, uint64_t max_reads, uint64_t stats[3]) {
    if (!h || !in_path || !out_paths) return fail(TD_E_ARG, "NULL argument");
    if (!h->have_splitter) return fail(TD_E_STATE, "td_set_splitter has not been called");
    bool gz_refused = false;
    const int rc = split_file_impl(h, in_path, out_paths, max_reads, stats, false, &gz_refused);
    if (!gz_refused) return rc;
    // a decoder has refused the .gz input: barcodeSplitter reads it through gzip.open too (reference :1318-1319) and leaves
    // its loop behind the quality line of read number maxreads (:1361-1362) -- its exception, or, where the loop is through
    // before the damage, the split taken again through the reference's own reading rules (gz_pyrules.hpp)
    const std::string refusal = g_err;
    MappedFile mf(in_path);
    if (!mf.ok) return fail(rc, refusal);
    const uint64_t r = std::max<uint64_t>(1, max_reads);
    const tdhost::GzVerdict v = tdhost::gz_verdict_lines(mf.p, mf.n, r >= (1ull << 60) ? ~0ull : 4 * r);
    if (v.kind != tdhost::GZ_OK) return fail(gz_code(v.kind), v.message);
    return split_file_impl(h, in_path, out_paths, max_reads, stats, true, &gz_refused);
}
static int split_file_impl(td_handle *h, const char *in_path, const char *const *out_paths, uint64_t max_reads, uint64_t stats[3],
                           bool by_reference_rules, bool *gz_refused) {
    HIPCHK(hipSetDevice(h->device));
    const size_t len = strlen(in_path);
    const bool gz = len >= 2 && (in_path[len - 2] == 'g' || in_path[len - 2] == 'G') && (in_path[len - 1] == 'z' || in_path[len - 1] == 'Z');
    tdhost::GzSource zsrc; FILE *pf = nullptr;
    std::unique_ptr<MappedFile> rules_map;
    std::unique_ptr<tdhost::PyGzipReader> rules;
    bool rules_through = false;
    bool opened = false;
    if (gz && by_reference_rules) {
        rules_map.reset(new MappedFile(in_path));
        opened = rules_map->ok;
        if (opened) rules.reset(new tdhost::PyGzipReader(rules_map->p, rules_map->n));
    } else if (gz) {
        struct stat sb0;
        if (stat(in_path, &sb0) == 0 && S_ISREG(sb0.st_mode) && sb0.st_size == 0) { pf = fopen(in_path, "rb"); opened = pf != nullptr; }   // (gzip.open: no data)
        else opened = zsrc.open(in_path);
    } else { pf = fopen(in_path, "rb"); opened = pf != nullptr; }
    if (!opened) return fail(TD_E_IO, std::string("cannot open ") + in_path);
    auto reader = [&](uint8_t *dst, size_t want) -> long {
        if (rules) {
            size_t done = 0;
            while (done < want && !rules_through) {
                const long got = rules->read(dst + done, std::min<size_t>(tdhost::PyGzipReader::CHUNK, want - done));
                if (got <= 0) { rules_through = true; break; }      // (damage behind the bound: the verdict says the loop never gets there)
                done += (size_t)got;
            }
            return (long)done;
        }
        if (gz && !pf) { const long got = zsrc.read(dst, want); if (got < 0) *gz_refused = true; return got; }
        size_t n = fread(dst, 1, want, pf);
        if (n == 0 && ferror(pf)) return -1;
        return (long)n;
    };
    SplitFiles files(h->sp_barcodes);
    const char *tenv = getenv("TAGDIG_SPLIT_THREADS");
    const uint32_t nthr = (uint32_t)std::max<long>(1, std::min<long>({tenv ? atol(tenv) : 16L, 256L, (long)h->sp_barcodes.size(),
                                                                      (long)std::max<unsigned>(1, std::thread::hardware_concurrency())}));
    std::vector<SplitWriter> writers;
    for (uint32_t t = 0; t < nthr; t++) { writers.emplace_back(files, t, nthr); writers.back().max_reads = std::max<uint64_t>(1, max_reads); }
    SplitWriter &w = writers[0];                 // (every writer sees the same lines: thread 0 speaks for the line numbers)
    PieceIndex pindex;
    pindex.nthreads = nthr; pindex.part.resize(nthr); pindex.flags.assign(nthr, 0); pindex.bucket.resize((size_t)nthr * nthr);
    // the writer threads live as long as this call: a piece is announced to them, the caller works as
    // writer 0, and all meet again at the end of the piece
    struct Job { const uint8_t *p = nullptr; size_t n = 0; const int2 *res = nullptr; uint64_t seq = 0; bool quit = false;
                 std::mutex mu; std::condition_variable cv; uint32_t finished = 0; } job;
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < nthr; t++)
        pool.emplace_back([&, t]() {
            uint64_t seen = 0;
            for (;;) {
                const uint8_t *p; size_t n; const int2 *res;
                {
                    std::unique_lock<std::mutex> g(job.mu);
                    job.cv.wait(g, [&]() { return job.quit || job.seq != seen; });
                    if (job.quit) return;
                    seen = job.seq; p = job.p; n = job.n; res = job.res;
                }
                writers[t].piece(p, n, res, pindex);
                { std::lock_guard<std::mutex> g(job.mu); job.finished++; }
                job.cv.notify_all();
            }
        });
    struct PoolEnd { Job &j; std::vector<std::thread> &pool; ~PoolEnd() {
        { std::lock_guard<std::mutex> g(j.mu); j.quit = true; }
        j.cv.notify_all();
        for (auto &th : pool) th.join();
    } } pool_end{job, pool};
    auto write_piece = [&](const uint8_t *p, size_t n, const int2 *res) {
        { std::lock_guard<std::mutex> g(job.mu); job.p = p; job.n = n; job.res = res; job.finished = 0; job.seq++; }
        job.cv.notify_all();
        writers[0].piece(p, n, res, pindex);
        std::unique_lock<std::mutex> g(job.mu);
        job.cv.wait(g, [&]() { return job.finished == nthr - 1; });
    };
    int rc = TD_OK;
    const size_t cap = (size_t)32 << 20;
    // two slots: the GPU decides piece k + 1 while the host writes piece k
    using Slot = td_handle::SplitSlot;
    Slot (&slot)[2] = h->sp_slot;                      // (allocated on first use, freed by td_destroy)
    auto cleanup = [&]() {
        zsrc.close();
        if (pf) fclose(pf);
        pf = nullptr;
    };
#define SPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(TD_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    for (auto &sl : slot) {
        sl.pending = false; sl.n = 0;
        if (!sl.pin) SPCHK(hipHostMalloc((void **)&sl.pin, cap, hipHostMallocDefault));
        if (!sl.dev) SPCHK(hipMalloc((void **)&sl.dev, cap));
        if (!sl.res_dev) SPCHK(hipMalloc((void **)&sl.res_dev, max_seq_lines(cap) * sizeof(int2)));          // (a line holds at least its terminator)
        if (!sl.res_pin) SPCHK(hipHostMalloc((void **)&sl.res_pin, max_seq_lines(cap) * sizeof(int2), hipHostMallocDefault));
        if (!sl.done) SPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    for (uint32_t b = 0; b < h->sp_barcodes.size(); b++) {
        FILE *f = fopen(out_paths[b], "wb");
        if (!f) { cleanup(); return fail(TD_E_IO, std::string("cannot open ") + out_paths[b] + " for writing"); }
        setvbuf(f, nullptr, _IONBF, 0);          // (SplitWriter keeps its own buffer per file)
        files.out.push_back(f);
    }
    files.pend.assign(files.out.size(), std::string());
    files.owner.resize(files.out.size());
    for (size_t b = 0; b < files.owner.size(); b++) files.owner[b] = (uint8_t)(b % nthr);
    std::vector<uint8_t> carry;
    bool eof = false;
    // TAGDIG_SPLIT_TIMING=1: where the wall time of this call went, on stderr
    const bool timing = getenv("TAGDIG_SPLIT_TIMING") != nullptr;
    double t_read = 0, t_wait = 0, t_write = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto hipfail = [&](const char *what, hipError_t e) { return fail(TD_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
    unsigned long long *terms_pin = nullptr;          // the piece's terminator count, read back before k_split
    {
        hipError_t e = hipHostMalloc((void **)&terms_pin, 16, hipHostMallocDefault);
        if (e != hipSuccess) { cleanup(); return hipfail("hipHostMalloc", e); }
    }
    // wait for the slot's decisions, then write its records (the host's share of the loop)
    auto write_out = [&](Slot &sl) -> int {
        if (!sl.pending) return TD_OK;
        const double t0 = now();
        hipError_t e = hipEventSynchronize(sl.done);
        if (e != hipSuccess) return hipfail("hipEventSynchronize", e);
        sl.pending = false;
        const double t1 = now();
        if (!w.stop) write_piece(sl.pin, sl.n, sl.res_pin);
        t_wait += t1 - t0; t_write += now() - t1;
        return TD_OK;
    };
    // Per piece: read it and start its copy and its line prefix on the GPU while the writer threads
    // write the PREVIOUS piece's records (that also tells the exact line index this piece starts at),
    // then have the GPU decide this piece's reads (a millisecond) and fetch the decisions.
    int cur = 0;
    Slot *prev = nullptr;
    while (!eof && !w.stop && !rc) {
        Slot &sl = slot[cur];
        int wrc = TD_OK;
        std::thread wt;
        if (prev) {
            Slot *pv = prev;
            wt = std::thread([&, pv]() { (void)hipSetDevice(h->device); wrc = write_out(*pv); });
        }
        prev = nullptr;
        const double tr0 = now();
        size_t have = carry.size();
        if (have) memcpy(sl.pin, carry.data(), have);
        carry.clear();
        while (have < cap) {
            long got = reader(sl.pin + have, cap - have);
            if (got < 0) { rc = TD_E_IO; break; }
            if (got == 0) { eof = true; break; }
            have += (size_t)got;
        }
        size_t cut = have;
        if (!rc && !eof) {
            cut = cut_at_line_end(sl.pin, have);
            if (cut == 0) rc = TD_E_LIMIT;
            else carry.assign(sl.pin + cut, sl.pin + have);
        }
        sl.n = cut;
        t_read += now() - tr0;
        hipError_t e = hipSuccess;
        const char *what = "";
        if (!rc && cut) {
            e = hipMemcpyAsync(sl.dev, sl.pin, cut, hipMemcpyHostToDevice, h->work_stream); what = "hipMemcpyAsync";
            if (e == hipSuccess) {
                const int prc = launch_split_prefix(h, sl.dev, cut, h->work_stream);
                if (prc) rc = prc;
                else e = hipMemcpyAsync(terms_pin, h->d_cursor.p, 8, hipMemcpyDeviceToHost, h->work_stream);
            }
        }
        if (wt.joinable()) wt.join();
        if (rc == TD_E_IO) { rc = fail(TD_E_IO, "read error while streaming FASTQ"); break; }
        if (rc == TD_E_LIMIT) { rc = fail(TD_E_LIMIT, "a single line exceeds the staging buffer"); break; }
        if (rc) break;
        if (e != hipSuccess) { rc = hipfail(what, e); break; }
        if (wrc) { rc = wrc; break; }
        if (cut && !w.stop) {
            const double t0 = now();
            e = hipStreamSynchronize(h->work_stream);                     // (copy + prefix ran under the writing)
            if (e != hipSuccess) { rc = hipfail("hipStreamSynchronize", e); break; }
            t_wait += now() - t0;
            const uint64_t lines = *terms_pin + 1;                        // (+1: an unterminated last line, at most)
            rc = launch_split(h, sl.dev, cut, w.lineindex, sl.res_dev, h->work_stream); if (rc) break;
            e = hipMemcpyAsync(sl.res_pin, sl.res_dev, max_seq_lines(lines) * sizeof(int2), hipMemcpyDeviceToHost, h->work_stream);
            if (e != hipSuccess) { rc = hipfail("hipMemcpyAsync", e); break; }
            (void)hipEventRecord(sl.done, h->work_stream);
            sl.pending = true;
            prev = &sl;
        }
        cur ^= 1;
    }
    if (!rc && prev) rc = write_out(*prev);
    for (auto &sl : slot) if (sl.pending) { (void)hipEventSynchronize(sl.done); sl.pending = false; }
    if (terms_pin) (void)hipHostFree(terms_pin);
    (void)hipStreamSynchronize(h->work_stream);
    unsigned long long st[TD_STAT_NSTATS];
    const bool have_st = hipMemcpy(st, h->d_stats.p, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess;
    cleanup();
#undef SPCHK
    bool io_error = false;
    uint64_t n_barcut = 0, n_clipped = 0;
    for (size_t k = 0; k < files.out.size(); k++) if (!w.flush(k)) io_error = true;
    for (FILE *&f : files.out) { if (f && fclose(f) != 0) io_error = true; f = nullptr; }
    for (auto &wr : writers) { io_error |= wr.io_error; n_barcut += wr.barcut; n_clipped += wr.clipped; }
    h->split_win.assign(2 * (size_t)((w.reads + 49999) / 50000), 0);
    for (auto &wr : writers)
        for (size_t k = 0; k < wr.wbar.size() && 2 * k + 1 < h->split_win.size(); k++) { h->split_win[2 * k] += wr.wbar[k]; h->split_win[2 * k + 1] += wr.wclip[k]; }
    if (io_error && !rc) rc = fail(TD_E_IO, "error writing an output file");
    if (stats) { stats[0] = w.reads; stats[1] = n_barcut; stats[2] = n_clipped; }
    if (timing) fprintf(stderr, "td_split_file: read %.3f s, waiting for the GPU %.3f s, writing %.3f s (thread 0: line index %.3f s, records %.3f s)\n",
                        t_read, t_wait, t_write, w.t_index, w.t_records);
    if (rc) return rc;
    if (w.nonascii) return fail(TD_E_NONASCII, "the splitter accepts ASCII FASTQ only (a byte >= 0x80 was found)");
    if (have_st) return check_device_errors(h, st);
    return TD_OK;
}

}  // extern "C"
