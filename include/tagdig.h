/* libtagdig -- C-ABI of the MI355X tag-counting engine.
 *
 * Drop-in boundary for ONE path of lvclark/tagdigger: the per-read
 * barcode-demux + known-tag count loop, tagdigger_fun.find_tags_fastq
 * (reference tagdigger_fun.py:192-277) and the index primitives under it
 * (:60-190).  The reference has no FFI of its own (it is pure Python); these
 * are the entry points a ctypes binding of that function needs, and
 * tagdigger_amd/_binding.py is that binding (INTEGRATION.md shows the stub a
 * reference maintainer would add).
 *
 * Conventions: plain pointers and sizes only; every buffer is caller-owned
 * unless said otherwise; functions return 0 on success or a negative TD_E_*
 * code with a message available from td_last_error(); nothing here calls
 * exit()/abort().  One handle drives one GPU; a handle is not thread-safe.
 * There is NO CPU fallback: without a usable HIP device td_create fails.
 */
#ifndef TAGDIG_H
#define TAGDIG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct td_handle td_handle;

enum {
    TD_OK = 0,
    TD_E_HIP = -1,        /* a HIP runtime call failed                                   */
    TD_E_ARG = -2,        /* bad argument (NULL, misaligned device pointer, ...)          */
    TD_E_OVERLAP = -3,    /* index build: reference's AssertionError "Problematic
                             sequence: {idx}" (tagdigger_fun.py:82); idx via td_last_bad_index */
    TD_E_EMPTY = -4,      /* index build: empty barcode or tag list (IndexError, :76)     */
    TD_E_ROOTLEAF = -5,   /* index build: first sequence empty, not the :109 special case;
                             the reference dies at its first lookup       */
    TD_E_ALPHABET = -6,   /* non-ACGT character in an index sequence (:198-204 asserts)   */
    TD_E_LIMIT = -7,      /* beyond an implementation limit (barcode+site > 32 bases,
                             tag > 320 bases, > 32767 barcode entries, ...)               */
    TD_E_NONASCII = -8,   /* a counted sequence line holds a byte >= 0x80                 */
    TD_E_STATE = -9,      /* call order (no index set, ...)                               */
    TD_E_INTERNAL = -10,  /* look-back timeout or other should-not-happen condition       */
    TD_E_IO = -11,        /* file could not be opened / read / inflated                   */
    TD_E_TASSEL = -12,    /* tassel_tagcount: header without a parsable count= value      */
    /* a .gz input ends the way gzip.open(fqfile, 'rt') ends it in the reference's loop
     * (tagdigger_fun.py:240-243, :250; csrc/gz_pyrules.hpp): the binding raises the same class
     * with the same message (td_last_error)                                               */
    TD_E_GZ_EOF = -13,    /* the compressed stream stops before its end-of-stream marker: EOFError */
    TD_E_GZ_BADFILE = -14,/* a member fails its CRC-32 / ISIZE check, or what follows a member is no
                             gzip header: gzip.BadGzipFile (an OSError)                    */
    TD_E_GZ_DATA = -15    /* invalid DEFLATE data: zlib.error                              */
};

/* stats[] slots filled by td_get_stats (all cumulative since td_reset) */
enum {
    TD_STAT_READS = 0,    /* readscount   of tagdigger_fun.py:246,255 */
    TD_STAT_BARCUT = 1,   /* barcutcount  of :247,259                 */
    TD_STAT_TAG = 2,      /* tagcount     of :248,263                 */
    TD_STAT_LINES = 3,    /* line terminators seen                    */
    TD_STAT_NSTATS = 8
};

const char *td_last_error(void);
uint32_t td_last_bad_index(void);

/* ---- lifetime ----------------------------------------------------------- */
int td_create(td_handle **out, int device_id);
void td_destroy(td_handle *h);

/* ---- index: replaces build_sequence_tree x2 inside find_tags_fastq --------
 * (tagdigger_fun.py:207-233).  The caller passes exactly the two string
 * lists the reference hands to build_sequence_tree:
 *   barcut[n_barcut]  upper-case barcode+cutsite strings, all cut-site
 *                     variants concatenated as at :215-217; entry k belongs to
 *                     barcode row k % barnum (:102-108)
 *   tagoff[barnum]    barcutlen of :209/:231 -- where the tag search starts
 *   tags[ntags]       upper-case tags after the strip decision of :222-231;
 *                     tag k is count-matrix column k
 * Duplicate / extension shadowing and the overlap assertion of :76-82 are
 * reproduced (TD_E_OVERLAP).  The count matrix becomes barnum x ntags, zeroed. */
int td_set_index(td_handle *h,
                 const char *const *barcut, uint32_t n_barcut, uint32_t barnum,
                 const uint32_t *tagoff,
                 const char *const *tags, uint32_t ntags);

/* Use caller-provided device memory (barnum*ntags uint32, zeroed by the
 * caller) for the count matrix, e.g. a torch tensor that is later all-reduced
 * over RCCL.  NULL returns to the internal buffer.
 * A bound matrix is the caller's: the library never moves it into its 64-bit host
 * accumulator (the internal matrix is flushed there before a cell could wrap), so a
 * cell wraps silently past 2^32 - 1 hits -- of ONE (barcode, tag) pair, summed over every
 * file counted into the matrix and, after an all-reduce, over every rank.  The reference's
 * default maxreads is 5e9 reads per file; BASELINE's largest job is 1.6e9 reads over
 * 384 x 500 000 cells.  A caller whose single cell may pass 4.29e9 must flush the
 * matrix into wider cells itself (or use the internal matrix and td_get_counts). */
int td_bind_counts(td_handle *h, void *d_counts);

/* Zero the count matrix, the statistics and the host-side accumulators. */
int td_reset(td_handle *h);

/* ---- the hot path: replaces the record loop of find_tags_fastq ------------
 * (tagdigger_fun.py:249-274) on a buffer already resident in HBM.
 *   d_fastq      device pointer, 16-byte aligned, nbytes bytes of FASTQ text
 *                made of whole lines (the last line may lack a terminator)
 *   first_line   global index of the buffer's first line (lineindex of :249)
 *   max_reads    reads (sequence lines) with ordinal > max_reads are ignored;
 *                pass max(1, ceil(maxreads)) for the semantics of :272-273
 *   weights      0 for the plain +1 count (:267); 1 for tassel_tagcount
 *                (:251-253,:264-265): header lines carry count=N
 *   stream       hipStream_t to launch on (NULL = default stream)
 * Asynchronous: returns once the work is enqueued.  ONE stream in flight per handle: the launch's
 * scratch (per-tile words, fix-up queue, tail copy, block sums) belongs to the handle, so work
 * enqueued through the same handle on a second stream must be ordered behind the first (an event,
 * or a synchronise) -- two unordered launches of one handle would race on it.  Use one handle per
 * concurrent stream. */
int td_count_device(td_handle *h, const void *d_fastq, uint64_t nbytes,
                    uint64_t first_line, uint64_t max_reads, int weights, void *stream);

/* Same for a host buffer: staged through pinned memory in pieces cut at line
 * ends, copies overlapped with counting.  Synchronous.  *lines_out (optional)
 * receives the number of lines consumed; like the reference's loop (:272) the
 * input stops being consumed soon after read number max_reads. */
int td_count_host(td_handle *h, const void *fastq, uint64_t nbytes,
                  uint64_t first_line, uint64_t max_reads, int weights, uint64_t *lines_out);

/* Whole file, plain or gzip (chosen by name as at :240: last two characters
 * 'gz' in any case), streamed.  BGZF is inflated on the GPU; any other gzip stream is
 * decoded by the host's threads and -- from 8 MiB of compressed data -- resolved and
 * CRC-checked on the GPU (options "gpu_inflate", "gpu_resolve").  Synchronous.
 * Environment: TAGDIG_INFLATE_THREADS (default: the host's cores, at most 16),
 * TAGDIG_INFLATE_CHUNK (bytes of compressed data per chunk, default 1 MiB),
 * TAGDIG_COPY_STREAMS (1..3 copy streams side by side for large uploads, default 3),
 * TAGDIG_INFLATE_STATS=1 (a line of timings on stderr). */
int td_count_file(td_handle *h, const char *path, uint64_t max_reads, int weights);

/* The gzip reader td_count_file / td_split_file use, on its own (host only, no GPU; for tests):
 * inflates `path` into dst[0..capacity), asking the reader for `chunk` bytes at a time (0 = 1 MiB).
 * BGZF files (bgzip) are inflated member-parallel on TAGDIG_INFLATE_THREADS threads (default: the
 * host's cores, at most 16); any other gzip stream (what gzip.open reads at tagdigger_fun.py:241,
 * multi-member included) by the library's own DEFLATE decoder: on the calling thread below 8 MiB
 * of compressed data, chunk-parallel on the same number of threads from there (csrc/par_inflate.hpp).
 * Every member's CRC-32 and length are checked.
 * TAGDIG_GUNZIP_PIPELINE=1: the chunk-parallel decoder is driven the way td_count_file drives it for the GPU (its
 * device mode: dev_next / dev_release / dev_check), the markers resolved by the host -- so that the pipeline
 * can be tested where there is no GPU. */
int td_gunzip_file(const char *path, void *dst, uint64_t capacity, uint64_t chunk, uint64_t *n_out);

/* An ordinary (single-member) .gz file inflated ON THE DEVICE (csrc/gz_gpu.hpp: block search, Huffman decoding into
 * tokens, LZ77 and the 32 KiB windows between chunks, CRC-32 and length check -- what td_count_file does with such a
 * file before it counts), its text copied to host memory `dst`.  *on_gpu = 0 and *n_out = 0: the device decoder leaves
 * this file to the host decoders (too small -- option "gz_gpu_min", default 8 MiB of compressed data -- several members,
 * no room on the device, or a stream it does not chain); td_gunzip_file reads such a file.  TD_E_IO: the member fails its
 * CRC-32 check.  Replaces gzip.open(fqfile, 'rt') of tagdigger_fun.py:240-241 for the test of that decoder alone. */
int td_gunzip_file_gpu(td_handle *h, const char *path, void *dst, uint64_t capacity, uint64_t *n_out, int *on_gpu);
/* 1: the .gz file td_count_file counted last was inflated by the device decoder; 0: by one of the others. */
int td_last_gz_route(td_handle *h);

/* ONE ordinary gzip file over several devices (tagdigger_amd/multi.py count_file_sharded; what gzip.open of
 * tagdigger_fun.py:240-241 reads, decoded by N ranks): a rank's part of the pipeline of csrc/gz_gpu.hpp.
 * td_gz_shard_open: the rank's byte range [byte_lo, byte_hi) of the compressed file goes to the device (and the bytes a
 *   margin further); *start_bit = the first block start in it (`first` != 0 -- rank 0: the member's first block), ~0: none.
 * td_gz_shard_decode: the stretch from there to stop_bit -- the next rank's start; ~0: to the member's end -- is decoded
 *   into symbols; *end_bit where it ended (must equal the next rank's start: the caller checks the seams), *out_len its
 *   bytes, *final whether it ended the member, map_out[32768]: what each place of the 32 KiB window behind the stretch
 *   holds -- a byte, or 0x8000 | a place of the window in front of it.
 * td_gz_shard_resolve: with window_in[32768] (the caller applies the maps of the ranks before this one to an empty
 *   window) and the bytes the member inflated to before the stretch: the stretch's text in device memory (*d_text) and its
 *   CRC-32 (td_crc32_join combines the ranks' in order; the caller checks the member's trailer).
 * TD_E_LIMIT: the file is not one this scheme takes (a chunk that does not chain, a stretch larger than a segment): the
 * caller lets one rank count it through td_count_file. */
int td_gz_shard_open(td_handle *h, const char *path, uint64_t byte_lo, uint64_t byte_hi, int first, uint64_t *start_bit, uint64_t *file_bytes);
int td_gz_shard_decode(td_handle *h, uint64_t stop_bit, uint64_t *end_bit, uint64_t *out_len, int *final, uint16_t *map_out);
int td_gz_shard_resolve(td_handle *h, const uint8_t *window_in, uint64_t member_out_before, void **d_text, uint32_t *crc32);
uint32_t td_crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

/* What the reference's loop over gzip.open(path, 'rt'), left at read number max_reads (:272-273), meets in this
 * file (host only, no GPU): TD_OK -- it ends without an exception -- or TD_E_GZ_EOF / TD_E_GZ_BADFILE /
 * TD_E_GZ_DATA with the exception's message in td_last_error.  td_count_file, td_gunzip_file and td_split_file
 * ask this whenever one of their decoders has refused a file (csrc/gz_pyrules.hpp: Lib/gzip.py restated call for
 * call over the same zlib); a file the reference reads to the bound although it is damaged further on is then
 * counted through that reader.  TD_E_IO: the file cannot be opened. */
int td_gzip_check(const char *path, uint64_t max_reads);

/* Line terminators (\n, \r\n, bare \r) in a device buffer -- what a shard of a
 * byte-split file must know about the shards before it.  Synchronous. */
int td_count_lines_device(td_handle *h, const void *d_fastq, uint64_t nbytes,
                          void *stream, uint64_t *terminators_out);

/* ---- one file over several GPUs (tagdigger_amd/multi.py count_file_sharded; the reference reads a file with one
 * text-mode loop, tagdigger_fun.py:240-250: a rank's byte range, or its range of BGZF members, replaces that loop's
 * input for the rank).  Both bring the bytes into DEVICE memory through the handle's pinned staging pieces -- the host
 * never holds more than two of them -- and return when they have landed. */
/* bytes [offset, offset + length) of `path` -> d_dst[0 .. length) */
int td_load_file_range(td_handle *h, const char *path, uint64_t offset, uint64_t length, void *d_dst);
/* the members of a BGZF file: file offset and inflated size of each (the end-of-file member included); *n_members is
 * the count whatever `capacity` holds.  TD_E_IO when some member is not BGZF. */
int td_bgzf_index(const char *path, uint64_t *member_off, uint32_t *member_isize, uint64_t capacity, uint64_t *n_members);
/* the members that START in [off_begin, off_end) of the file (off_begin must be a member's offset) inflated on the GPU,
 * one after the other, into d_dst[0 .. *nbytes); every member's size and CRC-32 are checked on the device. */
int td_bgzf_inflate_range(td_handle *h, const char *path, uint64_t off_begin, uint64_t off_end, void *d_dst, uint64_t capacity,
                          uint64_t *nbytes_out);

/* ---- barcode splitter (the adapter-trim branch) -----------------------------
 * Replaces the record loop of barcodeSplitter (tagdigger_fun.py:1318-1368) with its per-read
 * decisions -- sequence_index_lookup on barcode+cutsite (:1340) and findAdapterSeq (:1251-1283)
 * -- on the GPU; the host writes the clipped records.
 *
 * td_set_splitter: barcodes[nbar] and the single ACGT cut site (:1292-1293); fullsite0/1 = the two
 * full restriction sites (adapter[k][0] without '^', :1311-1312, at most 8 bases); and, per
 * barcode b, the adapter beginnings to look for at the END of a read, entries
 * ent_begin[b] .. ent_begin[b+1]-1: ent_seq[e] (forward orientation) and ent_slice[e] = the index
 * build_adapter_tree (:1208-1249) pairs with it.  The caller resolves that list with the
 * reference's rules (tagdigger_amd/tagdigger_fun.py does). */
int td_set_splitter(td_handle *h, const char *const *barcodes, uint32_t nbar, const char *cutsite,
                    const char *fullsite0, const char *fullsite1, const uint32_t *ent_begin,
                    const char *const *ent_seq, const int32_t *ent_slice, uint32_t nent);

/* Decisions for a buffer in device memory: d_out[2r], d_out[2r+1] = barcode index (-1: none) and
 * findAdapterSeq's return value (999: nothing to clip) for the buffer's r-th sequence line (lines
 * whose global index first_line + k is 1 mod 4).  out_capacity (in results) must be at least
 * (terminators + 1) / 4 + 2, terminators as td_count_lines_device reports them.
 * Synchronous; *n_terminators (optional) receives the buffer's line terminators. */
int td_split_device(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line,
                    int32_t *d_out, uint64_t out_capacity, void *stream, uint64_t *n_terminators);

/* Both per-read branches over one resident buffer (BASELINE config 5: counting + adapter trim): td_count_device's
 * pass, then td_split_device's, enqueued on the same stream; arguments as theirs.  Synchronous. */
int td_count_and_split_device(td_handle *h, const void *d_fastq, uint64_t nbytes, uint64_t first_line,
                              uint64_t max_reads, int32_t *d_out, uint64_t out_capacity, void *stream,
                              uint64_t *n_terminators);

/* The whole loop on a file (plain or gzip by name, :1318-1321): out_paths[nbar] are created
 * (truncated) and receive the clipped records of their barcode; stops after max_reads records
 * (:1361-1362).  stats = reads, reads with barcode+cut site, reads clipped on the 3' end (:1359). */
int td_split_file(td_handle *h, const char *in_path, const char *const *out_paths,
                  uint64_t max_reads, uint64_t stats[3]);
/* What the splitter's loop prints every 50 000 reads (:1357-1360), for the last td_split_file of this handle:
 * out[2 w] = reads with barcode + cut site, out[2 w + 1] = reads clipped on the 3' end, among the reads of
 * window w (50 000 consecutive reads); *nwindows = ceil(reads / 50 000).  Running sums are the printed numbers. */
int td_split_progress(td_handle *h, uint64_t *out, uint64_t cap, uint64_t *nwindows);

/* ---- results ---------------------------------------------------------------
 * Both synchronise with all work enqueued through this handle first and
 * return TD_E_NONASCII / TD_E_INTERNAL if a kernel flagged a problem. */
int td_get_counts(td_handle *h, uint64_t *out_rows_by_cols);   /* barnum*ntags, row-major */
int td_get_stats(td_handle *h, uint64_t stats[TD_STAT_NSTATS]);

/* The reference prints its three counters every 50 000 reads (tagdigger_fun.py:268-271).  With
 * td_set_option(h, "progress", 1) set before counting, the device keeps, per window of 50 000 consecutive reads
 * (read ordinals of the whole stream, across streamed pieces), how many reads had a barcode + cut site and how
 * many a tag; this returns them: out[2 w] = barcutcount and out[2 w + 1] = tagcount of the reads in window w,
 * for w < cap (windows without reads are zero), and *nwindows = ceil(reads / 50 000) for the reads THIS handle
 * counted -- a handle that counted from read 0 on needs no more than that many; the shard of a byte-sharded file
 * (first_line > 0) asks for the windows of the whole file.  Running sums over w give the numbers the reference
 * prints after read 50 000 (w + 1).  Cumulative since td_reset; synchronises like td_get_stats. */
int td_get_progress(td_handle *h, uint64_t *out, uint64_t cap, uint64_t *nwindows);

/* K3 of SURVEY 8e: add this library's barcode rows into the run's sample rows on the device --
 * d_dst[row_of_barcode[b]][c] += counts[b][c] for the handle's barnum x ntags uint32 matrix (bound or internal);
 * d_dst is n_dst_rows x ntags uint32 in device memory (e.g. the torch tensor that is all-reduced over RCCL
 * afterwards).  This is what the reference's combineReadCounts (tagdigger_fun.py:1061-1098) does with Python lists
 * after every file: rows whose sample name was seen before are summed.  row_of_barcode is a HOST array of barnum
 * entries.  Synchronous; waits for the handle's own work first and reports what a kernel flagged. */
int td_fold_rows(td_handle *h, const uint32_t *row_of_barcode, uint32_t n_dst_rows, void *d_dst, void *stream);

/* The raw-DEFLATE decoder the GPU runs one BGZF member per lane with (csrc/gpu_inflate.hpp), on the host: inflates
 * in[0..in_len) into out[0..out_len), out_len being the exact inflated size; 0 or a decoder error code (tests). */
int td_inflate_raw_host(const void *in, uint32_t in_len, void *out, uint32_t out_len);

/* Host helper of the CSV writers: vals[0..n) as decimal integers separated by commas (what csv.writer writes for a
 * row of ints, reference writeCounts tagdigger_fun.py:1100-1111) into out[0..capacity); returns the bytes written,
 * -1 when they do not fit (24 bytes per value always do). */
int64_t td_format_csv_row(const int64_t *vals, uint64_t n, char *out, uint64_t capacity);

/* ---- environment ------------------------------------------------------------
 * TAGDIG_STAGE_THREADS    host threads that copy / pread a piece into pinned memory (default 16 on hosts with 32 cores or more, else 8; 1..16)
 * TAGDIG_INFLATE_THREADS  host threads for BGZF member-parallel and gzip chunk-parallel inflate (default: cores, at most 16)
 * TAGDIG_PAR_INFLATE      0: ordinary gzip always on one thread; 1: always chunk-parallel (default: from 8 MiB compressed)
 * TAGDIG_INFLATE_CHUNK    compressed bytes per chunk of the chunk-parallel decoder (default 1 MiB; two chunks per thread and batch)
 * TAGDIG_INFLATE_STATS    set: the chunk-parallel decoder reports batches, chunks and where its time went, on stderr
 * TAGDIG_ZLIB             set: ordinary gzip through zlib's gzread, BGZF members through zlib's inflate
 * TAGDIG_SPLIT_THREADS    writer threads of td_split_file (default 16, at most the host's cores and the number of barcodes)
 * TAGDIG_SPLIT_TIMING     set: td_split_file reports where its wall time went, on stderr
 * TAGDIG_SPLIT_DISCARD    set: td_split_file assembles the records but writes nothing (timing only) */

/* ---- tuning / introspection ------------------------------------------------ */
/* Defaults are the measured best; every setting gives the same counts.  name:
 *   "tile_kb"        16 | 32 (default)           bytes of FASTQ per workgroup step
 *   "blocks_per_cu"  0 = what the occupancy query allows (default)
 *   "fastpath"       1 (default): predicted line phase + resolve + fix-up (kernel_fast.hpp);
 *                    0: the exact in-flight kernel with decoupled look-back (kernels.hpp)
 *   "prescan"        1: the exact kernel with a separate line-count pass instead of look-back
 *   "nt_loads"       1 (default): stream the FASTQ with non-temporal loads
 *   "prio"           wave priority per phase of the fast path, two bits each: phase A | B-C << 2 |
 *                    D << 4 | end of A << 6 (default 0xE4)
 *   "table_load_pct" fill of the tag hash table, 10..95 (default 25); applies to the next td_set_index
 *   "kernel"         main pass of the free-running path: 2 (default) k_fast2 -- raw tile in LDS, lines packed by the
 *                    lane that matches them, hot-cell cache (kernel_fast2.hpp); 1 k_fast (kernel_fast.hpp)
 *   "tile_kb2"       k_fast2's tile: 0 (default: chosen from the barcode index's LDS footprint) | 16 | 24 | 32
 *   "hot_cache"      1 (default): k_fast2 counts through its per-wave cache of hot cells in LDS (a wave rests its cache
 *                    while hardly anything hits); 0: plain atomics; 2: the cache never rests (measurements)
 *   "run"            consecutive tiles a workgroup of k_fast2 takes per turn (default 8): the line phase is carried
 *                    inside a run, only its first tile votes
 *   "stagger"        start-up stagger of co-resident workgroups, in 4096-cycle units (default 0)
 *   "timing"         1: record HIP events around every launch for td_kernel_time_ms
 *   "fast_max_matrix_bytes"  count matrices of this many bytes and more go to the exact kernel (the
 *                    free-running one addresses cells as base + 32-bit offset); 0 = the built-in 4 GiB
 *   "progress"       1: keep the reference's progress counters per window of 50 000 reads (td_get_progress); the main pass
 *                    is then k_fast2's recording instantiation (+6 % kernel time) or the exact kernel
 *   "split_kernel"   the splitter's per-read branch: 2 (default) k_split2 -- tile in LDS, one lane per read
 *                    (kernel_splitter2.hpp); 1 k_split (kernel_splitter.hpp)
 *   "gpu_resolve"    1 (default): ordinary gzip of 8 MiB and more -- DEFLATE decoded into 16-bit symbols on the host's
 *                    threads (copies that reach before a chunk stay markers), markers -> bytes and every member's CRC-32
 *                    on the GPU (csrc/gz_resolve.hpp), counted where it lands; 0: all of it on the host, the bytes then
 *                    staged like a plain file's.  (Environment TAGDIG_GPU_RESOLVE=0: the same, for every handle.)
 *   "gpu_inflate"    1 (default): BGZF members are inflated on the GPU; 0: on host threads
 *   "gpu_inflate_crc" 1 (default): every member's CRC-32 is checked on the device
 *   "zb_members"     BGZF members per GPU batch (tests; the built-in 49 152 is also the maximum)
 *   "debug_ablate"   timing-only ablation bits -- the counts are WRONG when non-zero
 * Returns TD_E_ARG for unknown names. */
int td_set_option(td_handle *h, const char *name, int64_t value);
/* Average device time (ms) of the count kernel over the launches since the
 * last call (HIP events on the launch stream); launches_out optional. */
int td_kernel_time_ms(td_handle *h, double *ms_per_launch, uint32_t *launches_out);
/* The same per launch: out[0 .. min(launches, capacity)) in launch order; *launches_out = how many were written. */
int td_kernel_times_ms(td_handle *h, double *out, uint32_t capacity, uint32_t *launches_out);

/* Diagnostic counters (24 x uint64).  All zero in the shipped build; the phase-stamp
 * build (make prof -> libtagdig_prof.so) fills [0..7] with shader-clock cycles per phase. */
int td_debug_counters(td_handle *h, uint64_t out[24]);

/* ---- device memory helpers (so a binding needs no other GPU runtime) ------- */
int td_dev_alloc(td_handle *h, uint64_t nbytes, void **d_out);
int td_dev_free(td_handle *h, void *d_ptr);
int td_memcpy_h2d(td_handle *h, void *d_dst, const void *src, uint64_t nbytes);
int td_memcpy_d2h(td_handle *h, void *dst, const void *d_src, uint64_t nbytes);
int td_device_sync(td_handle *h);

/* ---- bench/test utility: canonical synthetic FASTQ written straight into
 * HBM (include/td_synth_spec.h).  Not on the counting path. */
struct td_synth_params_s;
int td_synth_fill_device(td_handle *h, const void *params /* td_synth_params* */,
                         uint64_t first_read, uint64_t nreads,
                         const char *bar_tab, const uint8_t *bar_len, const char *cut_tab,
                         const char *tag_tab, const uint16_t *tag_len,
                         void *d_out, void *stream);

/* The count matrix that stream implies by the generator's own choices (td_synth_hit: read i is a
 * counted hit of barcode j and tag k), ADDED to d_counts (uint32 [nbar][ntags]); *hits_out = hits.
 * What the bench checks the counting kernels against at full size, without parsing any FASTQ. */
int td_synth_expected_device(td_handle *h, const void *params /* td_synth_params* */,
                             uint64_t first_read, uint64_t nreads, uint32_t *d_counts,
                             uint64_t *hits_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
