// Host-only harness over the library's gzip readers, for the CPU sanitizer builds (make -C tagdigger_amd/csrc san:
// g++ -fsanitize=thread and -fsanitize=address,undefined; never the GPU build).  No HIP header is included: the readers
// -- fast_inflate.hpp (one thread), par_inflate.hpp (chunk-parallel: its batch form and the device-mode pipeline that
// td_count_file drives for the GPU, with the markers resolved here), gz_source.hpp (BGZF member pool) and gz_pyrules.hpp
// (the reference's reading rules over zlib) -- are plain C++.
//
//   gz_san FILE.gz
// decodes FILE.gz through every reader and compares: on a stream gzip.open reads to its end all readers must give the
// same bytes; on one it refuses, no reader may give a clean end.  Exit code 0: consistent.  Decoder choices follow the
// environment as in the library (TAGDIG_INFLATE_THREADS, TAGDIG_INFLATE_CHUNK).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../gz_source.hpp"
#include "../gz_pyrules.hpp"

using tdhost::GzSource;
using tdhost::ParInflate;

static bool read_all(GzSource &src, std::vector<uint8_t> &out) {
    std::vector<uint8_t> buf(1 << 20);
    for (;;) {
        const long got = src.read(buf.data(), buf.size());
        if (got < 0) return false;
        if (got == 0) return true;
        out.insert(out.end(), buf.begin(), buf.begin() + got);
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: gz_san FILE.gz\n"); return 2; }
    const char *path = argv[1];
    // ---- the reference's rules: what gzip.open does with the file
    std::vector<uint8_t> want;
    bool want_ok = true;
    {
        GzSource m;
        if (!m.map_only(path)) { fprintf(stderr, "cannot map %s\n", path); return 2; }
        tdhost::PyGzipReader pr(m.map, m.bsize);
        uint8_t chunk[tdhost::PyGzipReader::CHUNK];
        for (;;) {
            const long got = pr.read(chunk, sizeof(chunk));
            if (got < 0) { want_ok = false; break; }
            if (got == 0) break;
            want.insert(want.end(), chunk, chunk + got);
        }
    }
    int bad = 0;
    auto verdict = [&](const char *who, bool ok, const std::vector<uint8_t> &got) {
        // (a reader that refuses a stream the reference reads sends the library to the reference's rules: slower, not wrong)
        if (ok && !want_ok) { fprintf(stderr, "%s: read a stream to a clean end that gzip.open refuses\n", who); bad++; }
        if (ok && want_ok && got != want) { fprintf(stderr, "%s: %zu bytes differ from gzip.open's %zu\n", who, got.size(), want.size()); bad++; }
        if (!ok && want_ok) fprintf(stderr, "%s: refused a stream gzip.open reads (note, not an error)\n", who);
    };
    // ---- one thread
    {
        setenv("TAGDIG_PAR_INFLATE", "0", 1);
        GzSource src; std::vector<uint8_t> got;
        const bool ok = src.open(path) && read_all(src, got);
        verdict("fast_inflate", ok, got);
    }
    // ---- chunk-parallel, batch form
    {
        setenv("TAGDIG_PAR_INFLATE", "1", 1);
        GzSource src; std::vector<uint8_t> got;
        const bool ok = src.open(path) && read_all(src, got);
        verdict("par_inflate (batches)", ok, got);
    }
    // ---- chunk-parallel, the pipeline of device mode (markers resolved here)
    {
        static const ParInflate::Allocator plain = {[](size_t b) -> void * { return malloc(b); }, [](void *p, size_t) { free(p); }};
        GzSource src; std::vector<uint8_t> got;
        bool ok = true;
        if (src.open_dev(path, &plain)) {
            for (;;) {
                const ParInflate::DevBatch *db = src.pi.dev_next();
                if (!db) { ok = !src.pi.dev_failed(); break; }
                if (db->failed) { src.pi.dev_release(); ok = false; break; }
                const size_t base = got.size();
                got.resize(base + db->total);
                std::vector<std::pair<uint32_t, size_t>> crc_len;
                bool res_ok = true;
                for (const auto &pc : db->pieces) {
                    uint32_t c = 0;
                    res_ok &= ParInflate::dev_resolve_on_host(pc, got.data() + base + pc.dest_off, &c);
                    crc_len.emplace_back(c, pc.len);
                }
                const bool last = db->last, member_done = db->member_done;
                const uint32_t wcrc = db->want_crc;
                src.pi.dev_release();
                if (!res_ok || !src.pi.dev_check(crc_len, member_done, wcrc)) { ok = false; break; }
                if (last) break;
            }
            verdict("par_inflate (pipeline)", ok, got);
        }
    }
    if (bad) return 1;
    printf("ok: %zu bytes, gzip.open %s\n", want.size(), want_ok ? "reads it" : "refuses it");
    return 0;
}
