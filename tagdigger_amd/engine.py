"""Host-side driver of the MI355X tag counter: one Engine = one GPU handle.

Mirrors the set-up half of the reference's find_tags_fastq
(tagdigger_fun.py:197-233) in Python -- asserts, cut-site enumeration,
barcode+cutsite list, strip-or-shift decision -- and hands the two string
lists to libtagdig (td_set_index), which builds the flat device index.  The
record loop (:239-277) runs on the GPU.
"""
import ctypes as C
import os
import math

from . import _binding as B

# IUPAC codes in the order the reference expands them (tagdigger_fun.py:140-189)
_IUPAC = (("R", "AG"), ("Y", "CT"), ("K", "GT"), ("M", "AC"), ("S", "CG"), ("W", "AT"),
          ("B", "CGT"), ("D", "AGT"), ("H", "ACT"), ("V", "ACG"), ("N", "ACGT"))


def enumerate_cut_sites(cutsite):
    """All concrete cut sites of an IUPAC cut site, in the reference's order
    (tagdigger_fun.py:136-190): codes are expanded one kind at a time, leftmost
    occurrence first, the new lists concatenated per replacement base."""
    out = [cutsite]
    for code, bases in _IUPAC:
        while out[0].find(code) > -1:
            nxt = []
            for b in bases:
                nxt.extend(x.replace(code, b, 1) for x in out)
            out = nxt
    return out


def combine_barcode_and_cutsite(barcodes, cutsite):
    """(barcode + cutsite).upper() per barcode (tagdigger_fun.py:60-69)."""
    assert all([set(barcode.upper()) <= set('ACGT') for barcode in barcodes]), "Non-ACGT barcode."
    assert set(cutsite.upper()) <= set('ACGT'), "Invalid cut site."
    return [(barcode + cutsite).upper() for barcode in barcodes]


def effective_maxreads(maxreads):
    """The reference tests `readscount >= maxreads` after each read
    (tagdigger_fun.py:272-273): one read is always processed and a fractional
    bound rounds up."""
    if maxreads >= 2 ** 62:
        return 2 ** 62
    return max(1, int(math.ceil(maxreads)))


def _c_strings(strings):
    arr = (C.c_char_p * max(1, len(strings)))()
    for i, s in enumerate(strings):
        arr[i] = s.encode("ascii")
    return arr


class Engine:
    """Owns a td_handle on one GPU."""

    def __init__(self, device=0):
        self._L = B.load()
        h = C.c_void_p()
        B.check(self._L.td_create(C.byref(h), int(device)))
        self._h = h
        self.device = device
        self.barnum = 0
        self.ntags = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.td_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ index
    def set_index(self, barcodes, tags, cutsite="TGCAG"):
        """Set-up of find_tags_fastq, tagdigger_fun.py:197-233.  The index on the device is kept when the
        same barcodes, tags and cut site come again (the libraries of one run usually share a barcode set):
        only the counts are zeroed then, as a fresh index would have them."""
        key = (tuple(barcodes), tuple(tags), cutsite)
        if key == getattr(self, "_index_key", None):
            self.reset()
            return
        self._index_key = None
        assert all([set(barcode.upper()) <= set('ACGT') for barcode in barcodes]), "Non-ACGT barcode."
        cutsite = cutsite.upper()
        assert set(cutsite) <= set('ACGTNRYKMSWBDHV'), "Invalid cut site."
        tags = [tag.upper() for tag in tags]
        assert all([set(tag) <= set('ACGT') for tag in tags]), "Non-ACGT tag."
        cutlen = len(cutsite)
        barcutlen = [len(x) + cutlen for x in barcodes]
        barnum = len(barcodes)
        cutsites = enumerate_cut_sites(cutsite)
        barcut = []
        for cut in cutsites:
            barcut += combine_barcode_and_cutsite(barcodes, cut)
        if set(x[:cutlen] for x in tags).issubset(set(cutsites)):
            if len(cutsites) == 1:
                tags = [x[cutlen:] for x in tags]          # site already checked with the barcode
            else:
                barcutlen = [x - cutlen for x in barcutlen]  # tags keep the (variable) site
        off = (C.c_uint32 * max(1, barnum))(*barcutlen)
        B.check(self._L.td_set_index(self._h, _c_strings(barcut), len(barcut), barnum, off,
                                     _c_strings(tags), len(tags)))
        self.barnum, self.ntags = barnum, len(tags)
        self._index_key = key

    # ------------------------------------------------------------------ counting
    def reset(self):
        B.check(self._L.td_reset(self._h))

    def set_option(self, name, value):
        B.check(self._L.td_set_option(self._h, name.encode(), int(value)))

    def bind_counts(self, device_ptr):
        B.check(self._L.td_bind_counts(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def count_device(self, d_ptr, nbytes, first_line=0, maxreads=5e9, tassel_tagcount=False, stream=0):
        """Enqueue one pass over a FASTQ buffer already in HBM (asynchronous)."""
        B.check(self._L.td_count_device(self._h, C.c_void_p(d_ptr), nbytes, first_line,
                                        effective_maxreads(maxreads), 1 if tassel_tagcount else 0,
                                        C.c_void_p(stream) if stream else None))

    def count_bytes(self, data, first_line=0, maxreads=5e9, tassel_tagcount=False):
        """Count a host buffer of whole lines; returns the number of line terminators consumed."""
        n = len(data)
        lines = C.c_uint64(0)
        if n:
            # no copy: bytes objects and ctypes arrays are passed by address, numpy arrays by .ctypes.data
            if isinstance(data, (bytes, C.Array)):
                ptr = data
            elif hasattr(data, "ctypes"):
                ptr = C.c_void_p(data.ctypes.data)
            else:
                ptr = (C.c_char * n).from_buffer_copy(data)
            B.check(self._L.td_count_host(self._h, ptr, n, first_line, effective_maxreads(maxreads),
                                          1 if tassel_tagcount else 0, C.byref(lines)))
        return lines.value

    def count_file(self, path, maxreads=5e9, tassel_tagcount=False):
        """Record loop of find_tags_fastq (tagdigger_fun.py:239-277) over a file."""
        # same failure modes as the reference's open() / gzip.open(): the OSError of a file that cannot be opened here;
        # what a .gz that is damaged, padded or no gzip at all ends in comes from the library (td_count_file, csrc/gz_pyrules.hpp:
        # EOFError / gzip.BadGzipFile / zlib.error with gzip.open's messages, raised by _binding.check)
        open(path, 'rb').close()
        B.check(self._L.td_count_file(self._h, path.encode(), effective_maxreads(maxreads),
                                      1 if tassel_tagcount else 0))

    def gunzip_file_gpu(self, path, capacity):
        """An ordinary .gz file inflated by the device decoder (csrc/gz_gpu.hpp): its text as bytes, or None where that
        decoder leaves the file to the host decoders (see td_gunzip_file_gpu in include/tagdig.h)."""
        buf = (C.c_uint8 * max(1, capacity))()
        n, on_gpu = C.c_uint64(0), C.c_int(0)
        B.check(self._L.td_gunzip_file_gpu(self._h, path.encode(), buf, capacity, C.byref(n), C.byref(on_gpu)))
        return bytes(memoryview(buf)[:n.value]) if on_gpu.value else None

    # one ordinary gzip file over several ranks (multi.count_file_sharded): include/tagdig.h td_gz_shard_*
    def gz_shard_open(self, path, byte_lo, byte_hi, first):
        """-> (first block start in the range as a bit position of the file, or None; the file's size)"""
        start, size = C.c_uint64(0), C.c_uint64(0)
        B.check(self._L.td_gz_shard_open(self._h, path.encode(), int(byte_lo), int(byte_hi), 1 if first else 0, C.byref(start), C.byref(size)))
        return (None if start.value == 2 ** 64 - 1 else start.value), size.value

    def gz_shard_decode(self, stop_bit):
        """-> (end bit, bytes of text, ended the member, map: numpy uint16[32768])"""
        import numpy as np
        end, n, fin = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        m = np.zeros(32768, dtype=np.uint16)
        B.check(self._L.td_gz_shard_decode(self._h, 2 ** 64 - 1 if stop_bit is None else int(stop_bit), C.byref(end), C.byref(n), C.byref(fin),
                                           m.ctypes.data_as(C.c_void_p)))
        return end.value, n.value, bool(fin.value), m

    def gz_shard_resolve(self, window_in, member_out_before):
        """-> (device pointer of the stretch's text, its CRC-32)"""
        import numpy as np
        w = np.ascontiguousarray(window_in, dtype=np.uint8)
        assert w.size == 32768
        ptr, crc = C.c_void_p(0), C.c_uint32(0)
        B.check(self._L.td_gz_shard_resolve(self._h, w.ctypes.data_as(C.c_void_p), int(member_out_before), C.byref(ptr), C.byref(crc)))
        return int(ptr.value or 0), crc.value

    def crc32_join(self, crc_a, crc_b, len_b):
        return int(self._L.td_crc32_join(int(crc_a), int(crc_b), int(len_b)))

    def last_gz_route(self):
        """1: the .gz file counted last was inflated on the device; 0: by a host decoder."""
        return int(self._L.td_last_gz_route(self._h))

    # ------------------------------------------------------------------ barcode splitter
    def set_splitter(self, barcodes, cutsite, fullsite0, fullsite1, entries):
        """entries[b] = [(adapter beginning to look for at the end of a read, slice index), ...] for
        barcode b (what build_adapter_tree, tagdigger_fun.py:1208-1249, resolves to)."""
        begin = [0]
        seqs, slices = [], []
        for per_barcode in entries:
            for seq, sl in per_barcode:
                seqs.append(seq)
                slices.append(sl)
            begin.append(len(seqs))
        B.check(self._L.td_set_splitter(self._h, _c_strings(barcodes), len(barcodes), cutsite.encode("ascii"),
                                        fullsite0.encode("ascii"), fullsite1.encode("ascii"),
                                        (C.c_uint32 * len(begin))(*begin), _c_strings(seqs),
                                        (C.c_int32 * max(1, len(slices)))(*slices), len(seqs)))

    def split_device(self, d_ptr, nbytes, first_line=0, stream=0):
        """[(barcode index or -1, findAdapterSeq value), ...] for the sequence lines of a device buffer."""
        import numpy as np
        cap = (self.count_lines_device(d_ptr, nbytes, stream) + 1) // 4 + 2
        d_out = self.dev_alloc(cap * 8)
        try:
            terms = C.c_uint64(0)
            B.check(self._L.td_split_device(self._h, C.c_void_p(d_ptr), nbytes, first_line, C.c_void_p(d_out), cap,
                                            C.c_void_p(stream) if stream else None, C.byref(terms)))
            raw = self.d2h(d_out, cap * 8)
        finally:
            self.dev_free(d_out)
        # lines of the buffer: one per terminator, plus an unterminated last line (the caller knows)
        return np.frombuffer(raw, dtype=np.int32).reshape(-1, 2), terms.value

    def count_and_split_device(self, d_ptr, nbytes, first_line=0, maxreads=5e9, stream=0):
        """Counting and the splitter's per-read decisions over ONE buffer in HBM (BASELINE config 5); the counts land
        in the engine's matrix, the decisions come back as split_device's."""
        import numpy as np
        cap = (self.count_lines_device(d_ptr, nbytes, stream) + 1) // 4 + 2
        d_out = self.dev_alloc(cap * 8)
        try:
            terms = C.c_uint64(0)
            B.check(self._L.td_count_and_split_device(self._h, C.c_void_p(d_ptr), nbytes, first_line, effective_maxreads(maxreads),
                                                      C.c_void_p(d_out), cap, C.c_void_p(stream) if stream else None, C.byref(terms)))
            raw = self.d2h(d_out, cap * 8)
        finally:
            self.dev_free(d_out)
        return np.frombuffer(raw, dtype=np.int32).reshape(-1, 2), terms.value

    def split_file(self, in_path, out_paths, maxreads=500000000):
        """The record loop of barcodeSplitter (tagdigger_fun.py:1318-1368); returns (reads, with
        barcode and cut site, clipped on the 3' end)."""
        st = (C.c_uint64 * 3)()
        B.check(self._L.td_split_file(self._h, in_path.encode(), _c_strings(out_paths),
                                      effective_maxreads(maxreads), st))
        return st[0], st[1], st[2]

    def split_progress_lines(self, in_path, reads):
        """The lines barcodeSplitter's loop prints while it reads (tagdigger_fun.py:1357-1360), for the last split_file."""
        n = C.c_uint64(0)
        B.check(self._L.td_split_progress(self._h, None, 0, C.byref(n)))
        out = (C.c_uint64 * max(1, 2 * n.value))()
        B.check(self._L.td_split_progress(self._h, out, n.value, C.byref(n)))
        lines, bar, clip = [], 0, 0
        for k in range(n.value):
            done = 50000 * (k + 1)
            if done > reads:
                break
            bar, clip = bar + out[2 * k], clip + out[2 * k + 1]
            if done % 1000000 == 0:
                lines.append(in_path)
            lines.append("Reads: {0} With barcode and cut site: {1} Clipped on 3' end: {2}".format(done, bar, clip))
        return lines

    def count_lines_device(self, d_ptr, nbytes, stream=0):
        out = C.c_uint64(0)
        B.check(self._L.td_count_lines_device(self._h, C.c_void_p(d_ptr), nbytes,
                                              C.c_void_p(stream) if stream else None, C.byref(out)))
        return out.value

    def load_file_range(self, path, offset, length, d_dst):
        """bytes [offset, offset + length) of a file -> device memory at d_dst, through the library's pinned staging pieces"""
        B.check(self._L.td_load_file_range(self._h, os.fsencode(path), int(offset), int(length), C.c_void_p(d_dst)))

    def bgzf_inflate_range(self, path, off_begin, off_end, d_dst, capacity):
        """the BGZF members starting in [off_begin, off_end) of a file, inflated on the GPU into d_dst; returns the bytes written"""
        n = C.c_uint64(0)
        B.check(self._L.td_bgzf_inflate_range(self._h, os.fsencode(path), int(off_begin), int(off_end), C.c_void_p(d_dst), int(capacity), C.byref(n)))
        return n.value

    def fold_rows(self, rows, d_dst, n_dst_rows, stream=0):
        """K3: add this library's barcode rows into sample rows on the device (d_dst: n_dst_rows x ntags uint32 in
        device memory; rows[b] = the sample row of barcode b) -- combineReadCounts (tagdigger_fun.py:1061-1098)
        without host lists."""
        arr = (C.c_uint32 * max(1, self.barnum))(*rows)
        B.check(self._L.td_fold_rows(self._h, arr, n_dst_rows, C.c_void_p(d_dst), C.c_void_p(stream) if stream else None))

    # ------------------------------------------------------------------ results
    def stats(self):
        st = (C.c_uint64 * B.TD_STAT_NSTATS)()
        B.check(self._L.td_get_stats(self._h, st))
        return {"reads": st[0], "barcut": st[1], "tag": st[2], "lines": st[3]}

    def progress_windows(self, nwindows=None):
        """[(reads with barcode + cut site, reads with tag)] per window of 50 000 reads, in read order (option
        "progress" must have been on while counting); the last window may be partly filled.  nwindows: that many
        windows from read 0 on (a shard of a byte-sharded file holds reads of high ordinals only); default: the
        windows of the reads this engine counted."""
        n = C.c_uint64(0)
        B.check(self._L.td_get_progress(self._h, None, 0, C.byref(n)))
        want = n.value if nwindows is None else int(nwindows)
        out = (C.c_uint64 * max(1, 2 * want))()
        B.check(self._L.td_get_progress(self._h, out, want, C.byref(n)))
        return [(out[2 * i], out[2 * i + 1]) for i in range(want)]

    def progress_lines(self, fqfile):
        """The lines the reference's loop prints while it reads (tagdigger_fun.py:268-271): the file name after
        every 1 000 000 reads, the three counters after every 50 000."""
        reads = self.stats()["reads"]
        lines, bar, tag = [], 0, 0
        for k, (b, t) in enumerate(self.progress_windows()):
            done = 50000 * (k + 1)
            if done > reads:
                break
            bar, tag = bar + b, tag + t
            if done % 1000000 == 0:
                lines.append(fqfile)
            lines.append("Reads: {0} With barcode and cut site: {1} With tag: {2}".format(done, bar, tag))
        return lines

    def counts_flat(self):
        out = (C.c_uint64 * max(1, self.barnum * self.ntags))()
        B.check(self._L.td_get_counts(self._h, out))
        return out

    def counts(self, signed=False):
        """list[list[int]] shaped [barcodes][tags], like the reference's mycounts (:237)."""
        # (numpy's tolist() builds the Python ints in C: 38 M cells in a second instead of half a minute)
        return self.counts_numpy(signed=signed).tolist()

    def counts_numpy(self, signed=False):
        """The same matrix as a numpy array (uint64; int64 when `signed`: tassel weights may be negative)."""
        import numpy as np
        flat = self.counts_flat()
        a = np.frombuffer(flat, dtype=np.int64 if signed else np.uint64, count=self.barnum * self.ntags)
        return a.reshape(self.barnum, self.ntags).copy()

    def debug_counters(self):
        out = (C.c_uint64 * 24)()
        B.check(self._L.td_debug_counters(self._h, out))
        return list(out)

    def kernel_time_ms(self):
        ms = C.c_double(0)
        n = C.c_uint32(0)
        B.check(self._L.td_kernel_time_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_times_ms(self, capacity=4096):
        """Device time of every launch since the last call, in launch order (needs option "timing")."""
        out = (C.c_double * capacity)()
        n = C.c_uint32(0)
        B.check(self._L.td_kernel_times_ms(self._h, out, capacity, C.byref(n)))
        return list(out[:n.value])

    # ------------------------------------------------------------------ device memory helpers
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        B.check(self._L.td_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        B.check(self._L.td_dev_free(self._h, C.c_void_p(ptr)))

    def h2d(self, d_ptr, data):
        n = len(data)
        if n:
            buf = (C.c_char * n).from_buffer_copy(data)
            B.check(self._L.td_memcpy_h2d(self._h, C.c_void_p(d_ptr), buf, n))

    def d2h(self, d_ptr, nbytes):
        buf = (C.c_char * max(1, nbytes))()
        if nbytes:
            B.check(self._L.td_memcpy_d2h(self._h, buf, C.c_void_p(d_ptr), nbytes))
        return bytes(buf[:nbytes])

    def sync(self):
        B.check(self._L.td_device_sync(self._h))


_default = {}


def default_engine(device=0):
    eng = _default.get(device)
    if eng is None:
        eng = _default[device] = Engine(device)
    return eng
