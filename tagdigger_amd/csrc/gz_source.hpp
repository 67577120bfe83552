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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    FILE *bf = nullptr;         // BGZF
    int threads = 1;
    std::vector<uint8_t> comp;  // compressed bytes of the request being served
    std::vector<uint8_t> spill; // a member that did not fit the caller's buffer, handed out in parts
    size_t spill_pos = 0;
    bool bad = false;

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
        int want = env ? atoi(env) : (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
        if (want > 1 && bgzf_header(head, n, &bs, &hs)) {
            rewind(f);
            bf = f; threads = want;
            return true;
        }
        fclose(f);
        if (!getenv("TAGDIG_ZLIB") && map_file(path, want)) return true;
        zf = gzopen(path, "rb");
        if (zf) gzbuffer(zf, 1 << 20);
        return zf != nullptr;
    }
    // the file followed by at least FastInflate::PAD readable zero bytes: an anonymous mapping one
    // page longer than the file, the file mapped over its beginning
    bool map_file(const char *path, int want_threads) {
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
        map = (uint8_t *)base; map_len = total;
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
        if (zf) gzclose(zf);
        if (bf) fclose(bf);
        if (map) munmap(map, map_len);
        zf = nullptr; bf = nullptr; map = nullptr; use_fi = false; use_pi = false;
    }
    ~GzSource() { close(); }

    // up to `want` uncompressed bytes into dst; 0 at the end, < 0 on error
    long read(uint8_t *dst, size_t want) {
        if (use_pi) return pi.read(dst, want);
        if (use_fi) return fi.read(dst, want);
        if (zf) return gzread(zf, dst, (unsigned)std::min<size_t>(want, 1u << 30));
        if (bad) return -1;
        if (spill_pos < spill.size()) {
            const size_t n = std::min(want, spill.size() - spill_pos);
            memcpy(dst, spill.data() + spill_pos, n);
            spill_pos += n;
            return (long)n;
        }
        struct Member { size_t in_off, in_len, out_off; uint32_t out_len, crc; size_t at, size; };   // at/size: the whole member
        std::vector<Member> mem;
        comp.clear();
        size_t out_total = 0;
        for (;;) {
            uint8_t head[18];
            const long at = ftell(bf);
            const size_t n = fread(head, 1, sizeof(head), bf);
            if (n == 0) break;                                         // end of file
            uint32_t bs = 0, hs = 0;
            if (!bgzf_header(head, n, &bs, &hs) || hs > 18 || bs < hs + 8) { bad = true; return -1; }
            const size_t base = comp.size();
            comp.resize(base + bs);
            memcpy(comp.data() + base, head, n);
            if (fread(comp.data() + base + n, 1, bs - n, bf) != bs - n) { bad = true; return -1; }
            const uint8_t *tail = comp.data() + base + bs - 8;
            const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
            if (out_total + isize > want) {                            // does not fit any more
                if (!mem.empty()) { comp.resize(base); fseek(bf, at, SEEK_SET); break; }     // next request
                // not even one member fits: inflate it aside and hand it out in parts
                spill.assign(isize, 0); spill_pos = 0;
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return -1; }
                zs.next_in = comp.data() + base + hs; zs.avail_in = (uInt)(bs - hs - 8);
                zs.next_out = spill.data(); zs.avail_out = isize;
                const int r = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (r != Z_STREAM_END || zs.avail_out != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), spill.data(), isize) != crc) { bad = true; return -1; }
                comp.clear();
                return read(dst, want);
            }
            mem.push_back({base + hs, (size_t)bs - hs - 8, out_total, isize, crc, base, (size_t)bs});
            out_total += isize;
        }
        if (mem.empty()) return 0;
        comp.resize(comp.size() + FastInflate::PAD, 0);              // (the decoder may read that far past a member)
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        const bool own_decoder = !getenv("TAGDIG_ZLIB");
        auto work_fast = [&]() {                                      // every member is a complete gzip member
            FastInflate dec;
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= mem.size() || failed) break;
                const Member &m = mem[k];
                dec.open(comp.data() + m.at, m.size);
                uint8_t extra;
                if (dec.read(dst + m.out_off, m.out_len) != (long)m.out_len || dec.read(&extra, 1) != 0) { failed = true; break; }
            }
        };
        auto work = [&]() {
            if (own_decoder) { work_fast(); return; }
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { failed = true; return; }
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= mem.size() || failed) break;
                const Member &m = mem[k];
                inflateReset(&zs);
                zs.next_in = comp.data() + m.in_off; zs.avail_in = (uInt)m.in_len;
                zs.next_out = dst + m.out_off; zs.avail_out = m.out_len;
                const int r = m.out_len || m.in_len ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
                if (r != Z_STREAM_END || zs.avail_out != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + m.out_off, m.out_len) != m.crc) { failed = true; break; }
            }
            inflateEnd(&zs);
        };
        const int nt = (int)std::min<size_t>((size_t)threads, mem.size());
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (failed) { bad = true; return -1; }
        return (long)out_total;
    }
};

}  // namespace tdhost
