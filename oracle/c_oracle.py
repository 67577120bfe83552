"""ctypes front-end of the C oracle (oracle.c, synth_ref.c) -- TEST
INFRASTRUCTURE ONLY.  Same results as tagdigger_oracle.count_bytes, fast
enough for full-size parity runs and for bench.py's cpu_baseline leg."""
import ctypes as C
import math
import os
import subprocess

from . import tagdigger_oracle as pyorc

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "synth_ref.c", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "td_synth_spec.h"))
    stale = not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("nbar", C.c_uint32), ("ntags", C.c_uint32),
                ("ncut", C.c_uint32), ("read_len", C.c_uint32), ("cut_len", C.c_uint32),
                ("tag_stride", C.c_uint32), ("adapter_pct", C.c_uint32), ("adapter_len", C.c_uint32),
                ("tag_cdf", C.c_void_p), ("bar_cdf", C.c_void_p), ("adapter", C.c_char * 64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_build.restype = C.c_int
        L.orc_build.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32,
                                C.POINTER(C.c_char_p), C.c_uint32,
                                C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.orc_free.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_count.restype = C.c_int
        L.orc_count.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                C.c_int, C.POINTER(C.c_uint32), C.c_uint32,
                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.synth_fill_host.restype = None
        L.synth_fill_host.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_uint64,
                                      C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p,
                                      C.c_void_p, C.c_void_p]
        L.synth_expected.restype = C.c_uint64
        L.synth_expected.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def effective_maxreads(maxreads):
    """`readscount >= maxreads` is tested after each read (:272-273), so at
    least one read is always processed and a float bound rounds up."""
    if maxreads >= 2 ** 63:
        return 2 ** 63
    return max(1, int(math.ceil(maxreads)))


def _raise(rc, bad):
    if rc == -1:
        raise AssertionError("Problematic sequence: {}.  Likely due to overlapping tags.".format(bad))
    if rc == -2:
        raise IndexError("list index out of range")
    if rc == -3:
        raise IndexError("string index out of range")
    if rc == -4:
        raise TypeError("'int' object is not subscriptable")
    if rc == -5:
        raise IndexError("list index out of range")
    if rc == -6:
        raise pyorc.NonAsciiSequence("non-ASCII byte in a sequence line")
    if rc == -8:
        raise ValueError("invalid literal for int() with base 10")
    raise RuntimeError("C oracle error %d" % rc)


class COracle:
    """Index built once, then count_bytes() on as many buffers as wanted."""

    def __init__(self, barcodes, tags, cutsite="TGCAG"):
        barcut, barnum, tags2, barcutlen = pyorc.prepare_lists(barcodes, tags, cutsite)
        self.barnum, self.ntags = barnum, len(tags2)
        self._bl = (C.c_uint32 * max(1, barnum))(*barcutlen)
        L = lib()
        bc = (C.c_char_p * max(1, len(barcut)))(*[s.encode() for s in barcut])
        tg = (C.c_char_p * max(1, len(tags2)))(*[s.encode() for s in tags2])
        h = C.c_void_p()
        bad = C.c_uint32(0)
        rc = L.orc_build(bc, len(barcut), barnum, tg, len(tags2), C.byref(h), C.byref(bad))
        if rc != 0:
            _raise(rc, bad.value)
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    def count_bytes(self, data, maxreads=5e9, tassel_tagcount=False, first_line=0, stats=None,
                    counts=None):
        import numpy as np
        if counts is None:
            counts = np.zeros((self.barnum, self.ntags), dtype=np.uint64)
        st = (C.c_uint64 * 4)()
        if isinstance(data, np.ndarray):
            ptr, n = data.ctypes.data, data.nbytes
        else:
            buf = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else None
            ptr, n = (C.addressof(buf) if buf is not None else None), len(data)
        rc = lib().orc_count(self._h, ptr, n, first_line, effective_maxreads(maxreads),
                             1 if tassel_tagcount else 0, self._bl, self.ntags,
                             counts.ctypes.data_as(C.POINTER(C.c_uint64)), st)
        if rc != 0:
            _raise(rc, 0)
        if stats is not None:
            stats["reads"], stats["barcut"], stats["tag"], stats["lines"] = st[0], st[1], st[2], st[3]
        return counts


def count_bytes(data, barcodes, tags, cutsite="TGCAG", maxreads=5e9, tassel_tagcount=False):
    o = COracle(barcodes, tags, cutsite)
    c = o.count_bytes(data, maxreads, tassel_tagcount)
    if tassel_tagcount:
        return [[int(v) for v in row] for row in c.astype("int64")]
    return [[int(v) for v in row] for row in c]


def find_tags_fastq(fqfile, barcodes, tags, cutsite="TGCAG", maxreads=5e9, tassel_tagcount=False):
    o = COracle(barcodes, tags, cutsite)
    c = o.count_bytes(pyorc.read_fastq_bytes(fqfile), maxreads, tassel_tagcount)
    return [[int(v) for v in row] for row in (c.astype("int64") if tassel_tagcount else c)]
