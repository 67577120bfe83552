// Chunk-parallel decoder for ordinary (single-stream) gzip on the host side of libtagdig (no GPU code).
//
// A DEFLATE stream has no index, and a block may copy from the 32 KiB before it, so a stream is
// normally decoded by one thread.  This decoder works through the compressed file in batches of two
// chunks per thread (1 MiB of compressed data each by default; the first batches are shorter), the
// chunks of a step taken up by the threads as they become free:
//   1. for every chunk but the first a block start is searched inside it: bit positions are tried
//      in turn for a non-final dynamic-Huffman header that only a compressor would write (all three
//      codes complete), that block is decoded, and a second valid header must follow it;
//   2. the chunk is then decoded from there into 16-bit symbols: literals as themselves, and a copy that reaches
//      into the unknown 32 KiB before the chunk as a marker 0x8000 | position-in-that-window (markers
//      are copied around like literals).  The first chunk starts at the known position with the
//      known window and is decoded as bytes.  Every chunk stops at the first block boundary at or past
//      the start its successor found (or past 12x its compressed size of output: that bounds the
//      buffers on highly compressible input, and the next batch goes on from there);
//   3. the chunks are chained on the calling thread: a chunk counts only if its predecessor stopped
//      exactly on its start (so a false start from step 1 is dropped with everything behind it, and
//      the next batch begins where the chain ended -- the result never depends on the search), and
//      each chunk's window follows from its predecessor's window and last 32 KiB;
//   4. all threads turn markers into bytes and take the CRC-32 of their part; the parts' CRCs are
//      combined, and CRC and length of every member are checked.
// Batches are decoded ahead of the reader by a producer thread (two output buffers), so read() is a
// copy that overlaps the next batch; and step 4 of one batch shares the threads with step 2 of the
// next (two sets of chunk buffers): its pieces are taken up as threads run out of chunks to decode,
// which fills the idle tail that chunks of unequal duration leave.
// Files without findable block starts (stored or fixed-Huffman blocks only) are decoded by the first
// thread alone.
// Device mode (dev_open / dev_next / dev_release / dev_check, what td_count_file drives: tagdig.hip count_gzip_dev) stops
// after step 3 -- the symbols go to the GPU as they are, step 4 runs there -- and has no batches on the decoding side:
// the territories lie on a fixed grid, the threads decode ahead into a ring of chunk buffers, and one thread chains behind
// them (see produce_device).  In both modes a chunk goes over from symbols to plain bytes at the first block boundary
// where its last 32 KiB hold no marker (streams with full flushes: pigz -i, bgzip-like writers).
// After: Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random
// access to DNA sequences" (2019), without its text heuristics -- step 3 makes the search exact.
#pragma once
#include <emmintrin.h>
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <memory>
#include <mutex>
#include <thread>

#include "fast_inflate.hpp"

namespace tdhost {

class ParInflate {
  public:
    // Device mode (dev_open): the chunks' buffers come from the caller's allocator -- pinned host memory, so that the
    // symbols can go to the GPU by DMA where the decoder left them -- instead of anonymous mappings.
    struct Allocator { void *(*alloc)(size_t bytes); void (*release)(void *p, size_t bytes); };
    static constexpr size_t PAD = FastInflate::PAD;          // readable bytes the caller guarantees behind the input
    // batches run, chunks on the chains, chunks decoded for nothing, block-start guesses rejected; seconds of
    // the producer in step 1 / steps 2 + 4 / step 3 and waiting for a free output buffer, of all threads in steps 1 / 2 / 4
    struct Stats {
        uint64_t batches = 0, chunks = 0, dropped = 0, rejected = 0, out_bytes = 0, as_bytes = 0;
        double t_search = 0, t_decode = 0, t_chain = 0, t_resolve = 0, t_wait = 0, t_find = 0, t_busy = 0;
    } stats;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

    ~ParInflate() { close(); }
    void open(const uint8_t *data, size_t n, int threads, size_t chunk_bytes) {
        close();
        data_ = data; n_ = n;
        device_ = false; al_ = nullptr;                          // (dev_open sets them again)
        threads_ = std::max(1, std::min(threads, MAX_CHUNKS / 4));
        const char *ov = getenv("TAGDIG_INFLATE_OVERSUB");          // (more chunks than threads: they take unequal time)
        max_chunks_ = std::min(MAX_CHUNKS, std::max(1, ov ? atoi(ov) : 2) * threads_);
        chunk_ = std::max<size_t>(chunk_bytes, 1024);
        // a chunk stops at the first block boundary past this much output (bounds the buffers on highly
        // compressible input: the chain then ends there and the next batch tries fewer chunks)
        cap_ = std::max<size_t>(chunk_ * 12, (size_t)1 << 22);
        cur_ = std::max(1, threads_ / 2);                        // a short first batch (the reader waits for two), doubling from there
        for (auto &set : sets_) set.reset(new Chunk[max_chunks_]);
        chunks_ = sets_[0].get();
        pend_.valid = false; crc_run_ = 0;
        failed_ = false; end_ = false; err_ = "";
        member_out_ = 0;
        stats = Stats();
        for (Slot &o : slot_) { o.len = o.pos = 0; o.ready = false; o.last = false; o.failed = false; }
        rd_ = 0; stop_ = false; drained_ = false; read_failed_ = false;
        begin_member(0);
    }
    // stops the producer (the input must stay mapped until then)
    void close() {
        if (producer_.joinable()) {
            { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
            cv_.notify_all();
            producer_.join();
            if (getenv("TAGDIG_INFLATE_STATS"))
                fprintf(stderr, "par_inflate: %d threads, %zu KiB chunks: %lu batches, %lu chunks (+%lu dropped), %lu guesses rejected, %.1f MB out (%.1f MB decoded as bytes); "
                        "producer: search %.3f s, decode + markers (device mode: waiting for chunks) %.3f s, chain %.3f s, waiting for the reader %.3f s; "
                        "threads: search %.3f s, decode %.3f s, markers + CRC %.3f s\n",
                        threads_, chunk_ >> 10, (unsigned long)stats.batches, (unsigned long)stats.chunks, (unsigned long)stats.dropped,
                        (unsigned long)stats.rejected, stats.out_bytes / 1e6, stats.as_bytes / 1e6, stats.t_search, stats.t_decode, stats.t_chain, stats.t_wait,
                        stats.t_find, stats.t_busy, stats.t_resolve);
        }
    }
    const char *error() const { return err_; }

    // up to `want` decompressed bytes into dst; 0 at the end of the stream, < 0 on error
    // (one thread copies 8-10 GB/s: the reader's copy out of the decoder's buffer would cap the whole pipeline there)
    static void par_memcpy(uint8_t *dst, const uint8_t *src, size_t n) {
        const size_t per = (size_t)4 << 20;
        const int nt = (int)std::min<size_t>(8, n / per);
        if (nt <= 1) { memcpy(dst, src, n); return; }
        std::vector<std::thread> th;
        const size_t chunk = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
        for (int t = 1; t < nt; t++) {
            const size_t off = std::min(n, (size_t)t * chunk), len = std::min(n, off + chunk) - off;
            if (len) th.emplace_back([=]() { memcpy(dst + off, src + off, len); });
        }
        memcpy(dst, src, std::min(n, chunk));
        for (auto &t : th) t.join();
    }

    long read(uint8_t *dst, size_t want) {
        if (!producer_.joinable() && !drained_) producer_ = std::thread([this]() { produce(); });
        size_t done = 0;
        while (done < want) {
            if (read_failed_) return done ? (long)done : -1;
            if (drained_) break;
            Slot &o = slot_[rd_];
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&]() { return o.ready; });
            }
            const size_t n = std::min(want - done, o.len - o.pos);
            par_memcpy(dst + done, o.b.p + o.pos, n);
            o.pos += n; done += n;
            if (o.pos == o.len) {
                if (o.failed) read_failed_ = true;
                else if (o.last) drained_ = true;
                else {
                    { std::lock_guard<std::mutex> g(mu_); o.ready = false; }
                    cv_.notify_all();
                    rd_ ^= 1;
                }
            }
        }
        return (long)done;
    }

    // ---------------------------------------------------------------- device mode
    // Steps 1-3 as above; step 4 -- markers into bytes, the CRC-32 -- is the CALLER's (on the GPU: tagdig.hip
    // count_gzip_dev).  What has been decoded and chained is handed over in batches: pieces of symbols (or bytes: chunks
    // decoded with a known window, and the part of a chunk behind its last marker) with the 32 KiB window each chunk's
    // markers point into; the caller uploads them, calls dev_release() as soon as the buffers may be decoded into
    // again, and dev_check() with the pieces' CRC-32s.
    struct DevPiece {
        const void *src;          // len symbols (uint16_t), or len bytes when `narrow`
        size_t len;
        bool narrow;
        const uint8_t *window;    // 32 KiB: marker 0x8000 | i is window[i]
        uint32_t min_idx;         // a marker below it points before the member's start: the stream is invalid
        size_t dest_off;          // where the piece's bytes go in the batch's output
    };
    struct DevBatch {
        std::vector<DevPiece> pieces;
        size_t total = 0;
        bool member_done = false; uint32_t want_crc = 0;
        bool failed = false, last = false;
        int state = 0;            // 0 free (the producer may fill it), 1 ready for the caller
        std::vector<int> slots;   // (the decoder's: the buffers the pieces lie in, given back by dev_release())
        std::vector<void *> spares;
    };
    void dev_open(const uint8_t *data, size_t n, int threads, size_t chunk_bytes, const Allocator *al) {
        open(data, n, threads, chunk_bytes);
        device_ = true; al_ = al;
        for (auto &set : sets_)
            for (int k = 0; k < max_chunks_; k++) { set[k].wide.al = al; set[k].narrow.al = al; }
        for (DevBatch &b : dev_) { b.state = 0; b.pieces.clear(); }
        dev_rd_ = 0; dev_done_ = false;
    }
    // the next batch (blocks until the producer has one); nullptr when the stream is through or has failed (error())
    const DevBatch *dev_next() {
        if (dev_done_) return nullptr;
        if (!producer_.joinable()) producer_ = std::thread([this]() { produce_device(); });
        DevBatch &b = dev_[dev_rd_];
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&]() { return b.state == 1; });
        }
        if (b.failed && b.pieces.empty()) { dev_done_ = true; return nullptr; }
        return &b;
    }
    // the batch dev_next() returned has been read out of the decoder's buffers
    void dev_release() {
        DevBatch &b = dev_[dev_rd_];
        cur_last_ = b.last || b.failed;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (int s : b.slots) free_ring_slot(s);
            for (void *c : b.spares) spare_free_.push_back((Chunk *)c);
            b.slots.clear(); b.spares.clear();
            b.state = 0;
        }
        cv_.notify_all();
        dev_rd_ ^= 1;
        if (cur_last_) dev_done_ = true;
    }
    // the CRC-32s of the batch's pieces, in order (a copy of what dev_next() returned must be kept by the caller:
    // the batch itself may be reused after dev_release()); false: a member fails its check
    bool dev_check(const std::vector<std::pair<uint32_t, size_t>> &crc_len, bool member_done, uint32_t want_crc) {
        for (const auto &cl : crc_len) crc_run_ = FI::crc32_join(crc_run_, cl.first, cl.second);
        if (member_done) {
            const bool ok = crc_run_ == want_crc;
            crc_run_ = 0;
            if (!ok) { err_ = "gzip member fails its CRC-32 check"; return false; }
        }
        return true;
    }
    bool dev_failed() const { return failed_; }
    // a piece's bytes and their CRC-32 by the host (what the GPU does in count_gzip_dev; for checks without one); false: a marker
    // points before the member's start
    static bool dev_resolve_on_host(const DevPiece &pc, uint8_t *dst, uint32_t *crc) {
        bool ok = true;
        if (pc.narrow) memcpy(dst, pc.src, pc.len);
        else ok = resolve((const uint16_t *)pc.src, pc.len, pc.window, dst, pc.min_idx);
        *crc = FastInflate::crc32_update(0, dst, pc.len);
        return ok;
    }

  private:
    using FI = FastInflate;
    static constexpr uint32_t WIN = 32768;
    static constexpr uint64_t NONE = ~0ull;

    template <typename T>
    struct Buf {                                              // uninitialised storage that can grow: its own mapping,
        T *p = nullptr; size_t cap = 0;                       // on huge pages where the system hands them out on request
        const Allocator *al = nullptr;                        // (or the caller's allocator: growth = new block, copy, release)
        ~Buf() { drop(); }                                    // (a page fault per 4 KiB of these buffers is most of a short run)
        void drop() {
            if (p) { if (al) al->release(p, bytes(cap)); else munmap(p, bytes(cap)); }
            p = nullptr; cap = 0;
        }
        static size_t bytes(size_t n) { return (n * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); }
        void reserve(size_t n) {
            if (n <= cap) return;
            const size_t want = bytes(n);
            if (al) {
                void *q = al->alloc(want);
                if (!q) abort();
                if (p) { memcpy(q, p, cap * sizeof(T)); al->release(p, bytes(cap)); }
                p = (T *)q; cap = want / sizeof(T);
                return;
            }
            void *q = p ? mremap(p, bytes(cap), want, MREMAP_MAYMOVE)
                        : mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) abort();
            (void)madvise(q, want, MADV_HUGEPAGE);
            p = (T *)q; cap = want / sizeof(T);
        }
    };
    struct Tables { uint32_t lit[FI::LIT_ENTRIES]; uint32_t dist[FI::DIST_ENTRIES]; };

    // the stream's next bits in the low end of a 64-bit word (as in FastInflate)
    struct Bits {
        const uint8_t *in = nullptr;
        uint64_t buf = 0; uint32_t cnt = 0;
        static uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
        void seek(const uint8_t *base, uint64_t bitpos) {
            in = base + (bitpos >> 3); buf = 0; cnt = 0;
            refill();
            buf >>= (bitpos & 7); cnt -= (uint32_t)(bitpos & 7);
        }
        uint64_t bitpos(const uint8_t *base) const { return (uint64_t)(in - base) * 8 - cnt; }
        void refill() { buf |= load64(in) << cnt; in += (63 - cnt) >> 3; cnt |= 56; }
        uint32_t bits(uint32_t n) { const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1)); buf >>= n; cnt -= n; return v; }
        void drop(uint32_t n) { buf >>= n; cnt -= n; }
    };

    struct Chunk {
        uint64_t search_from = 0, search_to = 0;     // its territory in bits (first chunk: search_from is its exact start)
        uint64_t start = NONE;                       // its first block (bit position), or NONE
        uint64_t resume = 0, target = 0;             // where step 2 goes on (behind the block step 1 decoded) and up to where
        uint64_t stop = 0;                           // the block boundary it stopped on
        bool member_done = false, failed = false;
        const char *err = "";
        Buf<uint16_t> wide; Buf<uint8_t> narrow;     // [WIN elements before the chunk | its output]
        size_t out_len = 0;
        size_t valid_back = 0;                       // how far before the chunk a copy may reach (elements)
        std::unique_ptr<Tables> tables;
        // after chaining
        uint8_t window[WIN];                         // the 32 KiB of output before it
        uint64_t member_before = 0;                  // bytes of the current member before it
        size_t dest_off = 0;                         // where its bytes go in the batch's output
        size_t wide_len = 0;                         // the first wide_len symbols of the output lie in `wide`, the rest as bytes in `narrow`
        double t_begin = 0, t_end = 0;               // (step 2, for TAGDIG_INFLATE_STATS=2)
    };

    const uint8_t *data_ = nullptr; size_t n_ = 0;
    static constexpr int MAX_CHUNKS = 128;
    int threads_ = 1, max_chunks_ = 2, cur_ = 1, nch_ = 0;
    size_t chunk_ = 0, cap_ = 0;
    std::unique_ptr<Chunk[]> sets_[2];               // the chunks of even and of odd batches
    Chunk *chunks_ = nullptr;                        // ... of the batch being decoded
    uint64_t pos_ = 0;                               // bit position of the next block of the current member
    uint64_t batch_end_ = 0;
    uint64_t member_out_ = 0;                        // bytes of the current member up to the batch being decoded
    uint32_t crc_run_ = 0;                           // CRC-32 of the current member up to the batch last resolved
    uint8_t window_[WIN];
    bool failed_ = false, end_ = false;              // (producer side)
    const char *err_ = "";
    // the two output buffers between the producer and read()
    struct Slot { Buf<uint8_t> b; size_t len = 0, pos = 0; bool ready = false, last = false, failed = false; } slot_[2];
    std::thread producer_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    int rd_ = 0;
    bool drained_ = false, read_failed_ = false;     // (reader side)

    // a batch that has been decoded and chained and waits for step 4
    struct Piece { Chunk *c; size_t off, len; uint32_t crc; bool bad; };
    struct Pending {
        bool valid = false;
        std::vector<Piece> pieces;
        Slot *slot = nullptr;
        size_t total = 0;
        bool member_done = false; uint32_t want_crc = 0;   // the batch ends its member: the CRC-32 its trailer states
        bool failed = false, last = false;                 // the stream is bad / ends behind this batch
    } pend_;

    bool device_ = false, dev_done_ = false, cur_last_ = false;
    const Allocator *al_ = nullptr;
    DevBatch dev_[2];
    int dev_rd_ = 0;

    // ---- device mode: no batches on the decoding side.  The file is cut into territories of chunk_ bytes on a fixed
    // grid; the worker threads take them in order, each searching its territory for a block start (step 1) and decoding
    // from there to the first block boundary behind the territory (step 2), into a ring of 2 * max_chunks_ chunk
    // buffers.  This thread follows with step 3: a chunk counts if it began exactly where the stream stands; where none
    // did (a false start, a territory without a findable start, the first block of a member, a chunk that stopped
    // early) it decodes up to the next territory itself, as bytes, with the window it knows.  What has been chained is
    // handed to the caller in batches of up to a quarter of the ring -- fewer when the next chunk is not ready yet --
    // whose buffers return to the ring with dev_release().
    enum : uint8_t { S_FREE = 0, S_BUSY, S_DONE, S_HELD };
    std::vector<uint8_t> st_;                        // state of the ring's buffers        } all under mu_
    std::vector<uint64_t> turn_;                     // ... and the territory each is for  }
    std::vector<std::unique_ptr<Chunk>> spare_all_;  // chunks this thread decodes into (taken and returned under mu_)
    std::vector<Chunk *> spare_free_;
    std::atomic<uint64_t> next_g_{0};
    bool quit_ = false;                              // (the stream is through: workers go home)
    int ring_ = 0;
    Chunk &ring_chunk(int s) { return s < max_chunks_ ? sets_[0][s] : sets_[1][s - max_chunks_]; }
    uint64_t grid_lo(uint64_t g) const { return g * (uint64_t)chunk_ * 8; }
    uint64_t grid_hi(uint64_t g) const { return (uint64_t)std::min<uint64_t>((g + 1) * (uint64_t)chunk_, n_) * 8; }
    void reset_chunk(Chunk &c, uint64_t from, uint64_t to) {
        c.search_from = from; c.search_to = to; c.target = to;
        c.start = NONE; c.resume = from;
        c.stop = 0; c.member_done = false; c.failed = false; c.err = ""; c.out_len = 0; c.wide_len = 0;
    }
    void free_ring_slot(int s) {                     // (mu_ held)
        st_[(size_t)s] = S_FREE;
        turn_[(size_t)s] += (uint64_t)ring_;
    }

    void device_worker() {
        for (;;) {
            const uint64_t g = next_g_.fetch_add(1);
            if (g * (uint64_t)chunk_ >= n_) return;
            const int s = (int)(g % (uint64_t)ring_);
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&]() { return (st_[(size_t)s] == S_FREE && turn_[(size_t)s] == g) || stop_ || quit_; });
                if (stop_ || quit_) return;
                st_[(size_t)s] = S_BUSY;
            }
            Chunk &c = ring_chunk(s);
            reset_chunk(c, grid_lo(g), grid_hi(g));
            find_start(c);
            decode_chunk(c, false);
            { std::lock_guard<std::mutex> lk(mu_); st_[(size_t)s] = S_DONE; }
            cv_.notify_all();
        }
    }

    // this thread's own decoding: from where the stream stands (pos_, window_ known) to the first block boundary at or past `target`
    Chunk *decode_here(uint64_t target) {
        Chunk *c = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!spare_free_.empty()) { c = spare_free_.back(); spare_free_.pop_back(); }
        }
        if (!c) {
            spare_all_.emplace_back(new Chunk);
            c = spare_all_.back().get();
            c->wide.al = al_; c->narrow.al = al_;
        }
        reset_chunk(*c, pos_, target);
        c->start = pos_;
        decode_chunk(*c, true);
        return c;
    }

    void produce_device() {
        ring_ = 2 * max_chunks_;
        st_.assign((size_t)ring_, S_FREE);
        turn_.resize((size_t)ring_);
        uint64_t gi = (pos_ >> 3) / chunk_;                        // the territory step 3 looks at next
        for (int s = 0; s < ring_; s++) {
            uint64_t g = gi - gi % (uint64_t)ring_ + (uint64_t)s;
            if (g < gi) g += (uint64_t)ring_;
            turn_[(size_t)s] = g;
        }
        next_g_ = gi; quit_ = false;
        std::vector<std::thread> workers;
        if (!failed_ && !end_)
            for (int i = 0; i < threads_; i++) {
                try { workers.emplace_back([this]() { device_worker(); }); } catch (...) { break; }     // (fewer workers, or this thread alone)
            }
        const int most = std::max(2, ring_ / 4), least = std::max(1, std::min(most, threads_ / 2));
        auto leave = [&]() {
            { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
            cv_.notify_all();
            for (auto &t : workers) t.join();
        };
        auto wait_done = [&](uint64_t g) -> bool {                 // false: asked to stop
            const size_t s = (size_t)(g % (uint64_t)ring_);
            const double tw = now();
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&]() { return (st_[s] == S_DONE && turn_[s] == g) || stop_; });
            stats.t_decode += now() - tw;
            return !stop_;
        };
        auto held = [&](uint64_t g) -> bool {
            const size_t s = (size_t)(g % (uint64_t)ring_);
            std::lock_guard<std::mutex> lk(mu_);
            return st_[s] == S_HELD;
        };
        const bool have_workers = !workers.empty();
        for (int b = 0;; b++) {
            DevBatch &d = dev_[b & 1];
            {
                const double tw = now();
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&]() { return d.state == 0 || stop_; });
                if (stop_) { lk.unlock(); leave(); return; }
                stats.t_wait += now() - tw;
            }
            d.pieces.clear(); d.slots.clear(); d.spares.clear();
            d.total = 0; d.member_done = false; d.want_crc = 0; d.failed = false; d.last = false;
            int count = 0;
            const double t0 = now();
            while (!failed_ && !end_ && count < most) {
                // territories the stream has already left behind (a long block, or this thread's own decoding, went through them)
                bool own = false;
                while (have_workers && gi * (uint64_t)chunk_ < n_ && pos_ >= grid_hi(gi)) {
                    if (count && held(gi)) { own = true; break; }
                    if (!wait_done(gi)) { leave(); return; }
                    { std::lock_guard<std::mutex> lk(mu_); free_ring_slot((int)(gi % (uint64_t)ring_)); }
                    cv_.notify_all();
                    gi++;
                }
                const bool grid = have_workers && gi * (uint64_t)chunk_ < n_;
                // (the buffer this territory is decoded into may still belong to the batch being put together -- long blocks
                // make a batch's chunks lie far apart: hand the batch over first)
                if (own || (grid && count && held(gi))) break;
                Chunk *c = nullptr;
                if (grid && pos_ >= grid_lo(gi)) {
                    const size_t s = (size_t)(gi % (uint64_t)ring_);
                    if (count >= least) {                          // (rather hand over what there is than wait)
                        std::lock_guard<std::mutex> lk(mu_);
                        if (!(st_[s] == S_DONE && turn_[s] == gi)) break;
                    }
                    if (!wait_done(gi)) { leave(); return; }
                    Chunk &r = ring_chunk((int)s);
                    if (r.start == pos_) {
                        c = &r;
                        { std::lock_guard<std::mutex> lk(mu_); st_[s] = S_HELD; }
                        d.slots.push_back((int)s);
                        gi++;
                    } else {                                       // a false start, or none: the territory is decoded here
                        { std::lock_guard<std::mutex> lk(mu_); free_ring_slot((int)s); }
                        cv_.notify_all();
                        stats.dropped++;
                        c = decode_here(grid_hi(gi));
                        d.spares.push_back((void *)c);
                        gi++;
                    }
                } else {
                    // the stream stands before the next territory (a member has begun, a chunk stopped early) or there are none left
                    c = decode_here(grid ? grid_lo(gi) : ~0ull);
                    d.spares.push_back((void *)c);
                }
                if (c->failed) { failed_ = true; err_ = c->err; break; }
                c->member_before = member_out_;
                c->dest_off = d.total;
                if (c->wide_len == c->out_len && c->wide_len) {
                    memcpy(c->window, window_, WIN);
                    for (uint32_t i = 0; i < WIN; i++) { const uint32_t v = c->wide.p[c->out_len + i]; window_[i] = v & 0x8000u ? c->window[v & 0x7FFFu] : (uint8_t)v; }
                } else if (c->out_len) {
                    if (c->wide_len) memcpy(c->window, window_, WIN);
                    memcpy(window_, c->narrow.p + (c->out_len - c->wide_len), WIN);
                }
                const uint32_t min_idx = WIN - (uint32_t)std::min<uint64_t>(WIN, member_out_);   // (a marker is a place in the 32 KiB before the chunk)
                if (c->wide_len) d.pieces.push_back(DevPiece{c->wide.p + WIN, c->wide_len, false, c->window, min_idx, d.total});
                if (c->out_len > c->wide_len) d.pieces.push_back(DevPiece{c->narrow.p + WIN, c->out_len - c->wide_len, true, c->window, min_idx, d.total + c->wide_len});
                stats.chunks++; stats.as_bytes += c->out_len - c->wide_len;
                d.total += c->out_len; member_out_ += c->out_len; pos_ = c->stop;
                count++;
                if (c->member_done) {
                    const size_t at = (size_t)((pos_ + 7) >> 3);
                    uint32_t want_len = 0;
                    if (at + 8 > n_) { failed_ = true; err_ = "truncated gzip member (no CRC / length)"; break; }
                    memcpy(&d.want_crc, data_ + at, 4); memcpy(&want_len, data_ + at + 4, 4);
                    if (want_len != (uint32_t)member_out_) { failed_ = true; err_ = "gzip member fails its length check"; break; }
                    d.member_done = true;
                    begin_member(at + 8);
                    break;                                         // (a batch carries one CRC-32 to check)
                }
            }
            stats.t_chain += now() - t0;
            stats.batches++;
            stats.out_bytes += d.total;
            d.failed = failed_; d.last = failed_ || end_;
            const bool fin = d.last;
            { std::lock_guard<std::mutex> lk(mu_); d.state = 1; }
            cv_.notify_all();
            if (fin) { leave(); return; }
        }
    }

    void produce() {
        for (int b = 0;; b++) {
            const bool more = !failed_ && !end_;
            double t0 = now(), tf = t0;
            if (more) {
                chunks_ = sets_[b & 1].get();
                territories();
                parallel(nch_ - 1, [this](int k) { find_start(chunks_[k + 1]); });
                for (int k = nch_ - 1; k >= 0; k--)               // every chunk heads for the next start found
                    chunks_[k].target = k == nch_ - 1 ? batch_end_ : (chunks_[k + 1].start != NONE ? chunks_[k + 1].start : chunks_[k + 1].target);
                tf = now();
                stats.t_search += tf - t0;
            } else {
                nch_ = 0;
            }
            if (pend_.valid) {                                    // its output buffer must have been read out
                Slot &o = *pend_.slot;
                const double tw = now();
                {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&]() { return !o.ready || stop_; });
                    if (stop_) return;
                }
                stats.t_wait += now() - tw;
                o.b.reserve(pend_.total + 1);
                o.len = pend_.total; o.pos = 0;
            } else if (!more) {                                   // (nothing at all: an empty or bad stream)
                publish(slot_[b & 1], 0);
                return;
            }
            // ---- step 2 of this batch, then step 4 of the one before as threads become free
            const double td = now();
            const int ndec = nch_, npieces = pend_.valid ? (int)pend_.pieces.size() : 0;
            parallel(ndec + npieces, [this, ndec](int i) {
                if (i >= ndec) resolve_piece(pend_.pieces[(size_t)(i - ndec)]);
                else decode_chunk(chunks_[i], i == 0);
            });
            const double t1 = now();
            stats.t_decode += t1 - td;
            if (more) {
                stats.batches++;
                if (const char *e = getenv("TAGDIG_INFLATE_STATS")) if (atoi(e) >= 2) {
                    fprintf(stderr, "par_inflate batch: search %.1f ms; decode per chunk [begin-end ms, KiB out]:", (tf - t0) * 1e3);
                    for (int k = 0; k < nch_; k++) fprintf(stderr, " %.1f-%.1f/%zu", (chunks_[k].t_begin - td) * 1e3, (chunks_[k].t_end - td) * 1e3, chunks_[k].out_len >> 10);
                    fprintf(stderr, " | with %d pieces of the batch before: %.1f ms\n", npieces, (t1 - td) * 1e3);
                }
            }
            if (pend_.valid) {
                const bool fin = finish_pending();
                if (fin) return;
            }
            if (more) chain(slot_[b & 1]);
            stats.t_chain += now() - t1;
        }
    }
    void publish(Slot &o, size_t len) {
        o.len = len; o.pos = 0;
        o.failed = failed_; o.last = failed_ || end_;
        { std::lock_guard<std::mutex> g(mu_); o.ready = true; }
        cv_.notify_all();
    }
    // CRCs of the resolved batch; true when it was the last one (end of the stream, or an error)
    bool finish_pending() {
        for (const Piece &pc : pend_.pieces) {
            if (pc.bad && !pend_.failed) { pend_.failed = true; err_ = "distance reaches before the start of the output"; }
            crc_run_ = FI::crc32_join(crc_run_, pc.crc, pc.len);
        }
        if (pend_.member_done) {
            if (!pend_.failed && crc_run_ != pend_.want_crc) { pend_.failed = true; err_ = "gzip member fails its CRC-32 check"; }
            crc_run_ = 0;
        }
        stats.out_bytes += pend_.total;
        Slot &o = *pend_.slot;
        const bool fin = pend_.failed || pend_.last;
        if (pend_.failed) failed_ = true;
        o.failed = pend_.failed; o.last = fin;
        { std::lock_guard<std::mutex> g(mu_); o.ready = true; }
        cv_.notify_all();
        pend_.valid = false;
        return fin;
    }
    void resolve_piece(Piece &pc) {
        const double t0 = now();
        Chunk &c = *pc.c;
        uint8_t *dst = pend_.slot->b.p + c.dest_off + pc.off;
        if (pc.off >= c.wide_len) memcpy(dst, c.narrow.p + WIN + pc.off - c.wide_len, pc.len);
        else {
            const uint64_t before = c.member_before;          // (a marker is a place in the 32 KiB before the CHUNK)
            pc.bad = !resolve(c.wide.p + WIN + pc.off, pc.len, c.window, dst, WIN - (uint32_t)std::min<uint64_t>(WIN, before));
        }
        pc.crc = FI::crc32_update(0, dst, pc.len);
        const double dt = now() - t0;
        std::lock_guard<std::mutex> g(mu_);
        stats.t_resolve += dt;
    }

    void begin_member(size_t at_byte) {
        const uint8_t *body = nullptr;
        const int r = FI::parse_member_header(data_ + at_byte, data_ + n_, &body, &err_, at_byte == 0);
        if (r < 0) { failed_ = true; return; }
        if (r == 0) { end_ = true; return; }
        pos_ = (uint64_t)(body - data_) * 8;
        member_out_ = 0;
    }

    // ---- block headers.  strict: accept only what a compressor writes (complete codes)
    enum { H_BAD = -1, H_STORED = 0, H_FIXED = 1, H_DYNAMIC = 2 };
    int block_header(Bits &b, Tables &t, bool strict, bool &final, uint32_t &stored_len, const char *&err) const {
        if (b.in > data_ + n_ + 16) { err = "truncated deflate stream"; return H_BAD; }
        b.refill();
        final = b.bits(1) != 0;
        const uint32_t type = b.bits(2);
        if (type == 0) {
            b.drop(b.cnt & 7);
            if (b.cnt < 32) b.refill();
            const uint32_t len = b.bits(16), nlen = b.bits(16);
            if ((len ^ 0xFFFFu) != nlen) { err = "stored block length check failed"; return H_BAD; }
            stored_len = len;
            return H_STORED;
        }
        if (type == 1) {
            uint8_t ll[288], dl[30];
            for (int i = 0; i < 144; i++) ll[i] = 8;
            for (int i = 144; i < 256; i++) ll[i] = 9;
            for (int i = 256; i < 280; i++) ll[i] = 7;
            for (int i = 280; i < 288; i++) ll[i] = 8;
            for (int i = 0; i < 30; i++) dl[i] = 5;
            FI::build_block_tables(t.lit, t.dist, ll, 288, dl, 30, true);
            return H_FIXED;
        }
        if (type == 3) { err = "reserved deflate block type"; return H_BAD; }
        const uint32_t nlit = b.bits(5) + 257, ndist = b.bits(5) + 1, nclen = b.bits(4) + 4;
        if (nlit > 286 || ndist > 30) { err = "too many length or distance codes"; return H_BAD; }
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        b.refill();
        for (uint32_t i = 0; i < nclen; i++) { if (b.cnt < 8) b.refill(); cl[order[i]] = (uint8_t)b.bits(3); }
        uint32_t cltab[128 + 19];
        bool complete = false;
        if (!FI::build(cltab, 7, cl, 19, FI::F_SUB, [&](int s, int l) -> uint32_t { return ((uint32_t)s << 16) | (uint32_t)l; }, &complete)) {
            err = "over-subscribed code-length code"; return H_BAD;
        }
        if (strict && !complete) return H_BAD;
        uint8_t lens[286 + 30];
        uint32_t i = 0;
        while (i < nlit + ndist) {
            b.refill();
            if (b.in > data_ + n_ + 16) { err = "truncated deflate stream"; return H_BAD; }
            const uint32_t e = cltab[b.buf & 127];
            if (!(e & 0xFF)) { err = "invalid code-length code"; return H_BAD; }
            b.drop(e & 0xFF);
            const uint32_t s = e >> 16;
            if (s < 16) { lens[i++] = (uint8_t)s; continue; }
            uint32_t rep, val = 0;
            if (s == 16) { if (i == 0) { err = "repeat with no previous length"; return H_BAD; } val = lens[i - 1]; rep = 3 + b.bits(2); }
            else if (s == 17) rep = 3 + b.bits(3);
            else rep = 11 + b.bits(7);
            if (i + rep > nlit + ndist) { err = "code lengths run past the end"; return H_BAD; }
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) { err = "no end-of-block code"; return H_BAD; }
        if (strict) {                                            // (before the tables are built: most guesses end here)
            uint32_t lsum = 0, dsum = 0, used = 0;
            for (uint32_t k = 0; k < nlit; k++) lsum += lens[k] ? 32768u >> lens[k] : 0;
            for (uint32_t k = 0; k < ndist; k++) { dsum += lens[nlit + k] ? 32768u >> lens[nlit + k] : 0; used += lens[nlit + k] != 0; }
            if (lsum != 32768u || !(dsum == 32768u || used <= 1)) return H_BAD;
        }
        if (!FI::build_block_tables(t.lit, t.dist, lens, (int)nlit, lens + nlit, (int)ndist, true)) {
            err = "over-subscribed Huffman code"; return H_BAD;
        }
        return H_DYNAMIC;
    }

    // the first bit position in [p, to) where a non-final dynamic block with a complete code-length
    // code could begin (its first 17 + 3 * nclen bits say so); `to` if there is none
    uint64_t next_plausible(uint64_t p, uint64_t to) const {
        static const KraftTable kraft;
        for (uint64_t q = p >> 3; q * 8 < to; q++) {
            const uint64_t v = Bits::load64(data_ + q);
            // BFINAL = 0, BTYPE = 2: bits 0, 0, 1 -- for all eight starts in this byte at once
            uint32_t m = (uint32_t)(~v & ~(v >> 1) & (v >> 2)) & 0xFFu;
            if (q * 8 < p) m &= 0xFFu << (p - q * 8);
            while (m) {
                const uint32_t sh = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                const uint64_t x = v >> sh;
                if (((x >> 3) & 31) > 29 || ((x >> 8) & 31) > 29) continue;
                const uint32_t nclen = (uint32_t)((x >> 13) & 15) + 4;
                const uint64_t at = q * 8 + sh + 17;
                uint64_t w = (Bits::load64(data_ + (at >> 3)) >> (at & 7)) & ((1ull << (3 * nclen)) - 1);
                uint32_t sum = 0;
                for (int j = 0; j < 7; j++, w >>= 9) sum += kraft.t[w & 511];
                if (sum == 128 && q * 8 + sh < to) return q * 8 + sh;
            }
        }
        return to;
    }
    struct KraftTable {                                        // three 3-bit code lengths -> their sum of 2^(7 - length)
        uint8_t t[512];
        KraftTable() {
            for (uint32_t i = 0; i < 512; i++) {
                uint32_t s = 0;
                for (uint32_t k = 0; k < 3; k++) { const uint32_t l = (i >> (3 * k)) & 7; s += l ? 128u >> l : 0; }
                t[i] = (uint8_t)std::min(s, 255u);
            }
        }
    };

    // ---- one Huffman block into out (growing the buffer as needed).  0: ended on its end-of-block code
    template <typename T>
    int huff_block(Bits &b, const Tables &t, Buf<T> &buf, T *&out, size_t valid_back, size_t limit, const char *&err) const {
        const uint64_t lmask = (1u << FI::LTB) - 1, dmask = (1u << FI::DTB) - 1;
        const uint8_t *const in_stop = data_ + n_ + 16;
        constexpr size_t MARGIN = 320;
        T *out_stop = buf.p + buf.cap - MARGIN;
        const T *begin = buf.p + WIN - valid_back;
        for (;;) {
            if (out >= out_stop) {
                const size_t off = (size_t)(out - buf.p);
                if (off > limit) { err = "block too long"; return -1; }
                buf.reserve(buf.cap + buf.cap / 2);
                out = buf.p + off; out_stop = buf.p + buf.cap - MARGIN; begin = buf.p + WIN - valid_back;
            }
            if (b.in > in_stop) { err = "truncated deflate stream"; return -1; }
            b.refill();
            uint32_t e = t.lit[b.buf & lmask];
            if (e & FI::F_SUB) e = t.lit[((e >> 16) & 0x1FFFu) + (uint32_t)((b.buf >> FI::LTB) & ((1u << ((e >> 8) & 0xFF)) - 1))];
            if (e & FI::F_LIT) {
                b.drop(e & 0xFF);
                out[0] = (T)(uint8_t)(e >> 16); out[1] = (T)(uint8_t)(e >> 8);
                out += 1 + ((e >> 28) & 1u);
                uint32_t e2 = t.lit[b.buf & lmask];
                if (e2 & FI::F_LIT) {
                    b.drop(e2 & 0xFF);
                    out[0] = (T)(uint8_t)(e2 >> 16); out[1] = (T)(uint8_t)(e2 >> 8);
                    out += 1 + ((e2 >> 28) & 1u);
                    e2 = t.lit[b.buf & lmask];
                    if (e2 & FI::F_LIT) {
                        b.drop(e2 & 0xFF);
                        out[0] = (T)(uint8_t)(e2 >> 16); out[1] = (T)(uint8_t)(e2 >> 8);
                        out += 1 + ((e2 >> 28) & 1u);
                    }
                }
                continue;
            }
            const uint32_t clen = e & 0xFF;
            if (clen == 0) { err = "invalid literal/length code"; return -1; }
            b.drop(clen);
            if (e & FI::F_EOB) return 0;
            const uint32_t lext = (e >> 8) & 0xFF;
            const uint32_t length = ((e >> 16) & 0x1FF) + (uint32_t)(b.buf & ((1u << lext) - 1));
            b.drop(lext);
            if (b.cnt < 32) b.refill();
            uint32_t d = t.dist[b.buf & dmask];
            if (d & FI::D_SUB) d = t.dist[((d >> 16) & 0x1FFFu) + (uint32_t)((b.buf >> FI::DTB) & ((1u << ((d >> 8) & 0xFF)) - 1))];
            const uint32_t dlen = d & 0xFF;
            if (dlen == 0) { err = "invalid distance code"; return -1; }
            b.drop(dlen);
            const uint32_t dext = (d >> 8) & 0xFF;
            const uint32_t distance = ((d >> 16) & 0x7FFF) + (uint32_t)(b.buf & ((1u << dext) - 1));
            b.drop(dext);
            if (distance > (size_t)(out - begin)) { err = "distance reaches before the start of the output"; return -1; }
            const T *src = out - distance;
            T *const end = out + length;
            constexpr uint32_t W = 16 / sizeof(T);               // elements per 16-byte copy
            if (distance >= 16) {
                do { memcpy(out, src, 16 * sizeof(T)); out += 16; src += 16; } while (out < end);
            } else if (distance >= W) {
                do { memcpy(out, src, 16); out += W; src += W; } while (out < end);
            } else if (distance == 1) {
                const T v = *src;
                T pat[W];
                for (uint32_t k = 0; k < W; k++) pat[k] = v;
                do { memcpy(out, pat, 16); out += W; } while (out < end);
            } else {
                // a short period: lay the pattern down by elements until a multiple of the period that is at
                // least W lies behind, then copy from that far back sixteen bytes at a time
                T *q = out;
                const uint32_t head = distance * ((W - 1 + distance) / distance);
                for (uint32_t k = 0; k < head && q < end; k++) *q++ = *src++;
                if (q < end) {
                    const T *s2 = q - head;
                    do { memcpy(q, s2, 16); q += W; s2 += W; } while (q < end);
                }
            }
            out = end;
        }
    }

    // ---- step 1 for chunk k >= 1: the first position in its territory that passes as a block start.  The
    // block decoded for the check stays in the chunk's buffer; step 2 goes on behind it
    void find_start(Chunk &c) {
        if (!c.tables) c.tables.reset(new Tables);
        Tables &t = *c.tables;
        Buf<uint16_t> &buf = c.wide;
        buf.reserve(WIN + chunk_ * 8 + 4096);
        for (uint32_t i = 0; i < WIN; i++) buf.p[i] = (uint16_t)(0x8000u | i);
        c.valid_back = WIN;
        uint64_t rejected = 0;
        const double t0 = now();
        const char *err = "";
        for (uint64_t p = c.search_from; (p = next_plausible(p, c.search_to)) < c.search_to; p++, rejected++) {
            Bits b;
            b.seek(data_, p);
            bool final = false; uint32_t slen = 0;
            if (block_header(b, t, true, final, slen, err) != H_DYNAMIC || final) continue;
            uint16_t *out = buf.p + WIN;
            if (huff_block<uint16_t>(b, t, buf, out, WIN, WIN + ((size_t)1 << 22), err) != 0) continue;
            Bits b2 = b;                                           // a second header must follow the block
            if (block_header(b2, t, true, final, slen, err) == H_BAD) continue;
            c.start = p; c.resume = b.bitpos(data_); c.out_len = (size_t)(out - buf.p) - WIN;
            break;
        }
        std::lock_guard<std::mutex> g(mu_);
        stats.rejected += rejected; stats.t_find += now() - t0;
    }

    // ---- step 2: chunk k from c.resume up to the first block boundary at or past c.target.  A chunk decoded in symbols
    // goes over to plain bytes at the first block boundary where its last 32 KiB hold no marker (nothing behind can
    // reach one any more): the byte decoder is faster and its output needs no step 4 beyond a copy
    template <typename T>
    int blocks(Chunk &c, Bits &b, Tables &t, Buf<T> &buf, T *&out, size_t valid_back, bool first, bool may_switch) {   // 1: go over to bytes
        const char *err = "";
        size_t scanned = 0, clean_from = 0;                          // (symbols of the output looked at for markers; none in [clean_from, scanned))
        for (;; first = false) {
            const uint64_t here = b.bitpos(data_);
            const size_t n = (size_t)(out - buf.p) - WIN;
            if (!first && c.wide_len + n > cap_) { c.stop = here; return 0; }
            if (!first && here >= c.target) {
                // on the grid of device mode a chunk's successor did not tell where it begins: stop where step 1 can have
                // found it -- in front of a non-final dynamic block (not in front of the empty stored block of a flush)
                if (!device_) { c.stop = here; return 0; }
                if (b.cnt < 3) b.refill();
                if ((b.buf & 7) == 4) { c.stop = here; return 0; }
            }
            if (may_switch && sizeof(T) == 2 && n >= WIN) {
                const uint16_t *sy = (const uint16_t *)buf.p + WIN;
                size_t i = scanned;
                for (; i + 4 <= n; i += 4) { uint64_t v; memcpy(&v, sy + i, 8); if (v & 0x8000800080008000ull) clean_from = i + 4; }
                for (; i < n; i++) if (sy[i] & 0x8000u) clean_from = i + 1;
                scanned = n;
                if (n - clean_from >= WIN) return 1;
            }
            bool final = false; uint32_t slen = 0;
            const int h = block_header(b, t, false, final, slen, err);
            if (h == H_BAD) { c.failed = true; c.err = err; c.stop = here; return 0; }
            if (h == H_STORED) {
                const uint8_t *s = b.in - (b.cnt >> 3);
                if (s + slen > data_ + n_) { c.failed = true; c.err = "truncated stored block"; c.stop = here; return 0; }
                const size_t off = (size_t)(out - buf.p);
                buf.reserve(off + slen + 1024);
                out = buf.p + off;
                for (uint32_t i = 0; i < slen; i++) out[i] = (T)s[i];
                out += slen;
                b.in = s + slen; b.buf = 0; b.cnt = 0;
            } else if (huff_block<T>(b, t, buf, out, valid_back, ~(size_t)0, err) != 0) {
                c.failed = true; c.err = err; c.stop = here; return 0;
            }
            if (final) { c.member_done = true; c.stop = b.bitpos(data_); return 0; }
        }
    }
    void decode_chunk(Chunk &c, bool exact) {
        if (c.start == NONE) return;
        const double t0 = now();
        if (!c.tables) c.tables.reset(new Tables);
        Tables &t = *c.tables;
        Bits b;
        b.seek(data_, c.resume);
        c.wide_len = 0;
        if (exact) {                                                 // (the window is known: bytes from the start)
            c.narrow.reserve(WIN + chunk_ * 8 + 4096);
            memcpy(c.narrow.p, window_, WIN);
            c.valid_back = (size_t)std::min<uint64_t>(WIN, member_out_);
            uint8_t *out = c.narrow.p + WIN;
            blocks<uint8_t>(c, b, t, c.narrow, out, c.valid_back, true, false);
            c.out_len = (size_t)(out - c.narrow.p) - WIN;
        } else {
            uint16_t *out = c.wide.p + WIN + c.out_len;
            static const bool no_switch = getenv("TAGDIG_INFLATE_NO_BYTES") != nullptr;
            const int r = blocks<uint16_t>(c, b, t, c.wide, out, c.valid_back, false, !no_switch);
            c.wide_len = c.out_len = (size_t)(out - c.wide.p) - WIN;
            if (r == 1) {
                c.narrow.reserve(WIN + chunk_ * 8 + 4096);
                const uint16_t *tail = c.wide.p + c.wide_len;          // (= the last WIN symbols: WIN in front of the output)
                for (uint32_t i = 0; i < WIN; i++) c.narrow.p[i] = (uint8_t)tail[i];
                uint8_t *o8 = c.narrow.p + WIN;
                blocks<uint8_t>(c, b, t, c.narrow, o8, WIN, false, false);
                c.out_len = c.wide_len + ((size_t)(o8 - c.narrow.p) - WIN);
            }
        }
        c.t_begin = t0; c.t_end = now();
        std::lock_guard<std::mutex> g(mu_);
        stats.t_busy += c.t_end - t0;
    }

    // fn(0) .. fn(n - 1) on up to threads_ threads, in order, each taking the next when it is free
    template <typename F>
    void parallel(int n, F &&fn) {
        if (n <= 0) return;
        std::atomic<int> next{0};
        auto work = [&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); };
        std::vector<std::thread> pool;
        for (int i = 1; i < std::min(n, threads_); i++) {
            try { pool.emplace_back(work); } catch (...) { break; }      // (no more threads to be had: fewer workers, same work)
        }
        work();
        for (auto &t : pool) t.join();
    }

    // markers -> bytes: sixteen marker-free symbols at a time by a pack, sixteen consecutive markers by a
    // copy, the rest through a 64 Ki-entry table (a literal is its own entry, marker 0x8000 | i is the
    // window's byte i).  min_idx: window positions below
    // it lie before the member's start, and a marker pointing there makes the stream invalid
    static bool resolve(const uint16_t *src, size_t n, const uint8_t *win, uint8_t *dst, uint32_t min_idx) {
        std::unique_ptr<uint8_t[]> lut(new uint8_t[65536]);
        for (uint32_t v = 0; v < 256; v++) lut[v] = (uint8_t)v;
        memcpy(lut.get() + 0x8000, win, WIN);
        bool ok = true;
        if (min_idx) {                                           // (only in the first 32 KiB of a member)
            for (size_t i = 0; i < n; i++) ok &= !(src[i] & 0x8000u) || (src[i] & 0x7FFFu) >= min_idx;
        }
        const uint8_t *t = lut.get();
        size_t i = 0;
        for (; i + 16 <= n; i += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 8));
            if ((_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) == 0) {
                _mm_storeu_si128((__m128i *)(dst + i), _mm_packus_epi16(a, b));
                continue;
            }
            // sixteen markers in a row pointing at sixteen window bytes in a row (a copied string: most
            // marker blocks of FASTQ are): one 16-byte copy from the window
            const uint32_t first = src[i];
            if (first >= 0x8000u && first <= 0xFFF0u) {
                const __m128i base = _mm_set1_epi16((short)first);
                const __m128i ea = _mm_add_epi16(base, _mm_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7));
                const __m128i eb = _mm_add_epi16(base, _mm_setr_epi16(8, 9, 10, 11, 12, 13, 14, 15));
                if (_mm_movemask_epi8(_mm_and_si128(_mm_cmpeq_epi16(a, ea), _mm_cmpeq_epi16(b, eb))) == 0xFFFF) {
                    _mm_storeu_si128((__m128i *)(dst + i), _mm_loadu_si128((const __m128i *)(win + (first & 0x7FFFu))));
                    continue;
                }
            }
            for (size_t k = i; k < i + 16; k++) dst[k] = t[src[k]];
        }
        for (; i < n; i++) dst[i] = t[src[i]];
        return ok;
    }

    // ---- the territories of the next batch
    void territories() {
        const size_t first = (size_t)(pos_ >> 3);
        nch_ = 0;
        for (int k = 0; k < cur_; k++) {
            const size_t lo = first + (size_t)k * chunk_;
            if (k > 0 && lo >= n_) break;
            Chunk &c = chunks_[k];
            c.search_from = k == 0 ? pos_ : (uint64_t)lo * 8;
            c.search_to = (uint64_t)std::min(lo + chunk_, n_) * 8;
            c.start = k == 0 ? pos_ : NONE; c.resume = pos_;
            c.stop = 0; c.member_done = false; c.failed = false; c.err = ""; c.out_len = 0;
            c.wide_len = 0;
            nch_++;
        }
        batch_end_ = chunks_[nch_ - 1].search_to;
    }

    // ---- step 3: the chain, windows, destinations in `o`, the pieces of step 4; moves on to where the
    // chain ended (over the member's trailer and the next header if it ended the member)
    void chain(Slot &o) {
        Chunk *chain[MAX_CHUNKS];
        int nchain = 0, last = 0;
        size_t total = 0;
        pend_.failed = false;
        for (int k = 0;;) {
            Chunk &c = chunks_[k];
            if (c.failed) {                                         // (a chunk on the chain decoded true blocks: the stream is bad)
                pend_.failed = true; failed_ = true; err_ = c.err;
                break;
            }
            chain[nchain++] = &c;
            last = k;
            c.member_before = member_out_ + total;
            if (k > 0) memcpy(c.window, window_, WIN);
            if (c.wide_len == c.out_len && k > 0) {
                for (uint32_t i = 0; i < WIN; i++) { const uint32_t v = c.wide.p[c.out_len + i]; window_[i] = v & 0x8000u ? c.window[v & 0x7FFFu] : (uint8_t)v; }
            } else {
                // (a chunk that went over to bytes took its last 32 KiB along, in front of them)
                memcpy(window_, c.narrow.p + (c.out_len - c.wide_len), WIN);
            }
            total += c.out_len;
            pos_ = c.stop;
            if (c.member_done) break;
            int j = k + 1;
            while (j < nch_ && chunks_[j].start == NONE) j++;
            if (j >= nch_ || chunks_[j].start != c.stop) break;
            k = j;
        }
        stats.chunks += (uint64_t)nchain;
        for (int i = 0; i < nchain; i++) stats.as_bytes += chain[i]->out_len - chain[i]->wide_len;
        // chunks behind the end of the chain were decoded for nothing (a false start, or a member ended
        // inside the batch): try fewer next time, more again when all were used
        const bool whole = nchain > 0 && (pos_ >= batch_end_ || last == nch_ - 1);
        if (!whole) stats.dropped += (uint64_t)(nch_ - 1 - last);
        cur_ = whole ? std::min(max_chunks_, cur_ * 2) : std::max(1, last + 1);
        member_out_ += total;
        pend_.valid = true;
        pend_.slot = &o;
        pend_.total = total;
        pend_.pieces.clear();
        size_t off = 0;
        for (int i = 0; i < nchain; i++) {
            Chunk &c = *chain[i];
            c.dest_off = off;
            // (tasks of step 4 on the host; for the device, whose copies want to be large, a chunk's symbols and its bytes whole)
            const size_t piece = device_ ? ~(size_t)0 >> 1 : PIECE;
            for (size_t at = 0; at < c.wide_len; at += std::min(piece, c.wide_len - at)) pend_.pieces.push_back({&c, at, std::min(piece, c.wide_len - at), 0, false});
            for (size_t at = c.wide_len; at < c.out_len; at += std::min(piece, c.out_len - at)) pend_.pieces.push_back({&c, at, std::min(piece, c.out_len - at), 0, false});
            off += c.out_len;
        }
        pend_.member_done = !pend_.failed && nchain && chunks_[last].member_done;
        if (pend_.member_done) {
            const size_t at = (size_t)((pos_ + 7) >> 3);
            uint32_t want_len = 0;
            if (at + 8 > n_) { pend_.failed = true; err_ = "truncated gzip member (no CRC / length)"; }
            else {
                memcpy(&pend_.want_crc, data_ + at, 4); memcpy(&want_len, data_ + at + 4, 4);
                if (want_len != (uint32_t)member_out_) { pend_.failed = true; err_ = "gzip member fails its length check"; }
                else begin_member(at + 8);
            }
            if (pend_.failed) failed_ = true;
        }
        pend_.failed |= failed_;
        pend_.last = end_;
    }
    static constexpr size_t PIECE = (size_t)1 << 19;          // symbols per task of step 4
};


}  // namespace tdhost
