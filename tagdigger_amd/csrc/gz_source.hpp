// Host-side gzip reader of libtagdig (no GPU code): what td_count_file, td_split_file and
// td_gunzip_file read .gz inputs through.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "fast_inflate.hpp"
#include "par_inflate.hpp"

namespace tdhost {

// Gzip input.  Ordinary .gz streams are memory-mapped and decoded by the library's own DEFLATE
// decoder: small files by FastInflate on the calling thread, files from 8 MiB by ParInflate on
// TAGDIG_INFLATE_THREADS threads (chunks of TAGDIG_INFLATE_CHUNK compressed bytes, default 1 MiB;
// TAGDIG_PAR_INFLATE=0/1 forces the choice; TAGDIG_ZLIB=1: zlib's gzread instead).  BGZF files (bgzip: a series of <= 64 KiB gzip members, each announcing its
// compressed size in a 'BC' extra field and ending with its uncompressed size) are inflated
// member-parallel: the members of one request are located first, then worker threads inflate
// them straight into the destination at their prefix offsets, each checking size and CRC-32.
struct GzSource {
    gzFile zf = nullptr;        // plain gzip through zlib
    FastInflate fi;             // plain gzip through the decoder of fast_inflate.hpp, over a mapping of the file
    ParInflate pi;              // ... or through the chunk-parallel decoder of par_inflate.hpp
    bool use_pi = false;
    uint8_t *map = nullptr;
    size_t map_len = 0;
    bool use_fi = false;
    bool bgzf = false;          // BGZF: members located in the mapping of the file, inflated by a pool that lives as long as the source
    size_t bpos = 0, bsize = 0; //       next member / size of the file
    int threads = 1;
    std::vector<uint8_t> spill; // a member that did not fit the caller's buffer, handed out in parts

    // worker threads for the BGZF members of one request after another (created once: a request is 32 MiB, a
    // millisecond of work for thirty threads -- creating them per request costs as much again)
    struct Pool {
        std::vector<std::thread> th;
        std::mutex mu;
        std::condition_variable cv;
        uint64_t gen = 0;
        uint32_t running = 0;
        bool quit = false;
        std::function<void()> job;
        void start(int n) {
            for (int t = 0; t < n; t++)
                th.emplace_back([this]() {
                    uint64_t seen = 0;
                    for (;;) {
                        std::function<void()> j;
                        {
                            std::unique_lock<std::mutex> g(mu);
                            cv.wait(g, [&]() { return quit || gen != seen; });
                            if (quit) return;
                            seen = gen; j = job;
                        }
                        j();
                        { std::lock_guard<std::mutex> g(mu); running--; }
                        cv.notify_all();
                    }
                });
        }
        void run(const std::function<void()> &j) {        // on every worker and on the caller; returns when all are done
            { std::lock_guard<std::mutex> g(mu); job = j; running = (uint32_t)th.size(); gen++; }
            cv.notify_all();
            j();
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&]() { return running == 0; });
        }
        void stop() {
            { std::lock_guard<std::mutex> g(mu); quit = true; }
            cv.notify_all();
            for (auto &t : th) t.join();
            th.clear(); quit = false;
        }
    } pool;
    size_t spill_pos = 0;
    bool bad = false;

    // decoder threads when TAGDIG_INFLATE_THREADS does not say: the host's cores shared out over the ranks of this node
    // (LOCAL_WORLD_SIZE, set by torch.distributed.run: eight ranks on one host each starting sixteen threads would be 128
    // threads on its cores), at most 16
    static int default_threads() {
        unsigned hw = std::max<unsigned>(1, std::thread::hardware_concurrency());
        const char *lw = getenv("LOCAL_WORLD_SIZE");
        const int ranks = lw ? atoi(lw) : 1;
        if (ranks > 1) hw = std::max<unsigned>(2, hw / (unsigned)ranks);
        return (int)std::min<unsigned>(16, hw);
    }
    static bool only_zeros(const uint8_t *p, size_t n) {
        for (size_t i = 0; i < n; i++) if (p[i]) return false;
        return true;
    }
    static bool bgzf_header(const uint8_t *p, size_t n, uint32_t *block_size, uint32_t *header_size) {
        if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
        const uint32_t xlen = p[10] | (p[11] << 8);
        if (12 + (size_t)xlen > n) return false;
        for (uint32_t o = 0; o + 4 <= xlen;) {
            const uint8_t *e = p + 12 + o;
            const uint32_t slen = e[2] | (e[3] << 8);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && o + 6 <= xlen) {
                *block_size = (uint32_t)(e[4] | (e[5] << 8)) + 1u;
                *header_size = 12 + xlen;
                return (p[3] & ~4) == 0;              // no name / comment / header CRC in BGZF blocks
            }
            o += 4 + slen;
        }
        return false;
    }
    bool open(const char *path) {
        FILE *f = fopen(path, "rb");
        if (!f) return false;
        uint8_t head[64];
        const size_t n = fread(head, 1, sizeof(head), f);
        uint32_t bs = 0, hs = 0;
        const char *env = getenv("TAGDIG_INFLATE_THREADS");
        int want = env ? atoi(env) : default_threads();
        fclose(f);
        if (want > 1 && bgzf_header(head, n, &bs, &hs) && map_only(path)) {
            bgzf = true; bpos = 0; threads = want;
            pool.start(want - 1);
            return true;
        }
        if (!getenv("TAGDIG_ZLIB") && map_file(path, want)) return true;
        zf = gzopen(path, "rb");
        if (zf) gzbuffer(zf, 1 << 20);
        return zf != nullptr;
    }
    // the file followed by at least FastInflate::PAD readable zero bytes: an anonymous mapping one
    // page longer than the file, the file mapped over its beginning
    bool map_only(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size == 0) { ::close(fd); return false; }
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t n = (size_t)sb.st_size;
        const size_t total = (n + page - 1) / page * page + page;
        void *base = mmap(nullptr, total, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == MAP_FAILED) { ::close(fd); return false; }
        void *over = mmap(base, n, PROT_READ, MAP_PRIVATE | MAP_FIXED, fd, 0);
        ::close(fd);
        if (over == MAP_FAILED) { munmap(base, total); return false; }
        (void)madvise(base, n, MADV_SEQUENTIAL);
        map = (uint8_t *)base; map_len = total; bsize = n;
        return true;
    }
    // Device mode of the chunk-parallel decoder (par_inflate.hpp): only for what it takes -- one ordinary gzip stream of
    // 8 MiB and more, several threads; false: open() the file the usual way.
    bool open_dev(const char *path, const ParInflate::Allocator *al) {
        const char *env = getenv("TAGDIG_INFLATE_THREADS");
        const int want = env ? atoi(env) : default_threads();
        const char *par = getenv("TAGDIG_PAR_INFLATE");
        if (want <= 1 || getenv("TAGDIG_ZLIB") || (par && atoi(par) <= 0) || !map_only(path)) return false;
        uint32_t bs = 0, hs = 0;
        if (bgzf_header(map, bsize, &bs, &hs) || (!par && bsize < ((size_t)8 << 20))) { close(); return false; }
        const char *cb = getenv("TAGDIG_INFLATE_CHUNK");
        pi.dev_open(map, bsize, want, cb ? (size_t)atol(cb) : (size_t)1 << 20, al);
        use_pi = true;
        return true;
    }
    bool map_file(const char *path, int want_threads) {
        if (!map_only(path)) return false;
        const size_t n = bsize;
        // chunk-parallel above 8 MiB of compressed data (TAGDIG_PAR_INFLATE=1: always, =0: never)
        const char *par = getenv("TAGDIG_PAR_INFLATE");
        if (want_threads > 1 && (par ? atoi(par) > 0 : n >= ((size_t)8 << 20))) {
            const char *cb = getenv("TAGDIG_INFLATE_CHUNK");
            pi.open(map, n, want_threads, cb ? (size_t)atol(cb) : (size_t)1 << 20);
            use_pi = true;
        } else {
            fi.open(map, n);
            use_fi = true;
        }
        return true;
    }
    void close() {
        pi.close();
        if (bgzf) pool.stop();
        if (zf) gzclose(zf);
        if (map) munmap(map, map_len);
        zf = nullptr; bgzf = false; map = nullptr; use_fi = false; use_pi = false;
    }
    ~GzSource() { close(); }

    // up to `want` uncompressed bytes into dst; 0 at the end, < 0 on error
    long read(uint8_t *dst, size_t want) {
        if (use_pi) return pi.read(dst, want);
        if (use_fi) return fi.read(dst, want);
        if (zf) {
            // (TAGDIG_ZLIB=1, a comparator for tests: zlib's gzread conventions -- a stream that stops early is an error
            // here too, bytes behind the last member that are no gzip header are silently left unread)
            const int got = gzread(zf, dst, (unsigned)std::min<size_t>(want, 1u << 30));
            if (got <= 0) { int e = Z_OK; (void)gzerror(zf, &e); if (e != Z_OK && e != Z_STREAM_END) return -1; }
            return got;
        }
        if (bad || !bgzf) return -1;
        if (spill_pos < spill.size()) {
            const size_t n = std::min(want, spill.size() - spill_pos);
            memcpy(dst, spill.data() + spill_pos, n);
            spill_pos += n;
            return (long)n;
        }
        struct Member { size_t at, size, hs, out_off; uint32_t out_len, crc; };   // at/size: the whole member in the mapping
        std::vector<Member> mem;
        size_t out_total = 0;
        while (bpos < bsize) {
            uint32_t bs = 0, hs = 0;
            if (!bgzf_header(map + bpos, bsize - bpos, &bs, &hs) || bs < hs + 8 || bpos + bs > bsize) {
                if (only_zeros(map + bpos, bsize - bpos)) { bpos = bsize; break; }     // (padding: gzip.open skips it)
                bad = true; return -1;
            }
            const uint8_t *tail = map + bpos + bs - 8;
            const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
            if (out_total + isize > want) {                            // does not fit any more
                if (!mem.empty()) break;                                   // next request
                // not even one member fits: inflate it aside and hand it out in parts
                spill.assign(isize, 0); spill_pos = 0;
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return -1; }
                zs.next_in = map + bpos + hs; zs.avail_in = (uInt)(bs - hs - 8);
                zs.next_out = spill.data(); zs.avail_out = isize;
                const int r = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (r != Z_STREAM_END || zs.avail_out != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), spill.data(), isize) != crc) { bad = true; return -1; }
                bpos += bs;
                return read(dst, want);
            }
            mem.push_back({bpos, (size_t)bs, (size_t)hs, out_total, isize, crc});
            out_total += isize;
            bpos += bs;
        }
        if (mem.empty()) return 0;
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        const bool own_decoder = !getenv("TAGDIG_ZLIB");
        // members are claimed in runs of eight (neighbours in the file and in the destination)
        auto work = [&]() {
            FastInflate dec;
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (!own_decoder && inflateInit2(&zs, -15) != Z_OK) { failed = true; return; }
            for (;;) {
                const size_t k0 = next.fetch_add(8);
                if (k0 >= mem.size() || failed) break;
                for (size_t k = k0; k < std::min(mem.size(), k0 + 8); k++) {
                    const Member &m = mem[k];
                    if (own_decoder) {                                     // every member is a complete gzip member
                        dec.open(map + m.at, m.size);
                        uint8_t extra;
                        if (dec.read(dst + m.out_off, m.out_len) != (long)m.out_len || dec.read(&extra, 1) != 0) { failed = true; break; }
                    } else {
                        inflateReset(&zs);
                        zs.next_in = map + m.at + m.hs; zs.avail_in = (uInt)(m.size - m.hs - 8);
                        zs.next_out = dst + m.out_off; zs.avail_out = m.out_len;
                        const int r = m.out_len || zs.avail_in ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
                        if (r != Z_STREAM_END || zs.avail_out != 0 ||
                            (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + m.out_off, m.out_len) != m.crc) { failed = true; break; }
                    }
                }
            }
            if (!own_decoder) inflateEnd(&zs);
        };
        pool.run(work);
        if (failed) { bad = true; return -1; }
        return (long)out_total;
    }
};

}  // namespace tdhost
