"""ctypes binding of libtagdig.so (include/tagdig.h).

There is no fallback: if the library or a HIP runtime cannot be loaded, or no
GPU is present, importing/using this module raises.  The CPU checker kept elsewhere in the
repository is test infrastructure and is never imported from here.

One HIP runtime per process: libtagdig.so carries no NEEDED entry for
libamdhip64, so this module first loads exactly one runtime with RTLD_GLOBAL --
PyTorch's bundled copy when torch is installed (so device pointers, streams
and RCCL collectives can be shared with torch), otherwise /opt/rocm's.
Override with TAGDIG_HIP_RUNTIME=/path/to/libamdhip64.so.
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAGDIG_LIB") or os.path.join(_HERE, "libtagdig.so")

TD_STAT_NSTATS = 8
TD_E = {
    -1: "TD_E_HIP", -2: "TD_E_ARG", -3: "TD_E_OVERLAP", -4: "TD_E_EMPTY", -5: "TD_E_ROOTLEAF",
    -6: "TD_E_ALPHABET", -7: "TD_E_LIMIT", -8: "TD_E_NONASCII", -9: "TD_E_STATE",
    -10: "TD_E_INTERNAL", -11: "TD_E_IO", -12: "TD_E_TASSEL",
    -13: "TD_E_GZ_EOF", -14: "TD_E_GZ_BADFILE", -15: "TD_E_GZ_DATA",
}


class TagdigError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (TD_E.get(code, code), message))
        self.code = code
        self.detail = message


class NonAsciiSequence(ValueError):
    """A counted sequence line holds a byte >= 0x80 ('bytes >= 0x80')."""


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("nbar", C.c_uint32), ("ntags", C.c_uint32),
                ("ncut", C.c_uint32), ("read_len", C.c_uint32), ("cut_len", C.c_uint32),
                ("tag_stride", C.c_uint32), ("adapter_pct", C.c_uint32), ("adapter_len", C.c_uint32),
                ("tag_cdf", C.c_void_p), ("bar_cdf", C.c_void_p), ("adapter", C.c_char * 64)]


def _hip_runtime_candidates():
    env = os.environ.get("TAGDIG_HIP_RUNTIME")
    if env:
        yield env
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            yield p
    for p in ("/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
        yield p


_lib = None
_runtime_path = None


def load():
    """Load (once) and return the ctypes library object."""
    global _lib, _runtime_path
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "tagdigger_amd: %s is missing -- build it with `make -C tagdigger_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
    errs = []
    for cand in _hip_runtime_candidates():
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            _runtime_path = cand
            break
        except OSError as e:
            errs.append("%s: %s" % (cand, e))
    else:
        raise ImportError("tagdigger_amd: no HIP runtime could be loaded:\n  " + "\n  ".join(errs))
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    pp = C.POINTER(C.c_char_p)
    sig("td_last_error", C.c_char_p)
    sig("td_last_bad_index", u32)
    sig("td_create", i32, C.POINTER(vp), i32)
    sig("td_destroy", None, vp)
    sig("td_set_index", i32, vp, pp, u32, u32, C.POINTER(u32), pp, u32)
    sig("td_bind_counts", i32, vp, vp)
    sig("td_reset", i32, vp)
    sig("td_count_device", i32, vp, vp, u64, u64, u64, i32, vp)
    sig("td_count_host", i32, vp, vp, u64, u64, u64, i32, C.POINTER(u64))
    sig("td_count_file", i32, vp, C.c_char_p, u64, i32)
    sig("td_count_lines_device", i32, vp, vp, u64, vp, C.POINTER(u64))
    sig("td_load_file_range", i32, vp, C.c_char_p, u64, u64, vp)
    sig("td_bgzf_index", i32, C.c_char_p, vp, vp, u64, C.POINTER(u64))
    sig("td_bgzf_inflate_range", i32, vp, C.c_char_p, u64, u64, vp, u64, C.POINTER(u64))
    sig("td_gunzip_file", i32, C.c_char_p, vp, u64, u64, C.POINTER(u64))
    sig("td_gzip_check", i32, C.c_char_p, u64)
    sig("td_gunzip_file_gpu", i32, vp, C.c_char_p, vp, u64, C.POINTER(u64), C.POINTER(C.c_int))
    sig("td_last_gz_route", i32, vp)
    sig("td_gz_shard_open", i32, vp, C.c_char_p, u64, u64, C.c_int, C.POINTER(u64), C.POINTER(u64))
    sig("td_gz_shard_decode", i32, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int), vp)
    sig("td_gz_shard_resolve", i32, vp, vp, u64, C.POINTER(vp), C.POINTER(u32))
    sig("td_crc32_join", u32, u32, u32, u64)
    sig("td_set_splitter", i32, vp, C.POINTER(C.c_char_p), u32, C.c_char_p, C.c_char_p, C.c_char_p,
        C.POINTER(u32), C.POINTER(C.c_char_p), C.POINTER(C.c_int32), u32)
    sig("td_split_device", i32, vp, vp, u64, u64, vp, u64, vp, C.POINTER(u64))
    sig("td_count_and_split_device", i32, vp, vp, u64, u64, u64, vp, u64, vp, C.POINTER(u64))
    sig("td_split_file", i32, vp, C.c_char_p, C.POINTER(C.c_char_p), u64, C.POINTER(u64))
    sig("td_fold_rows", i32, vp, C.POINTER(u32), u32, vp, vp)
    sig("td_inflate_raw_host", i32, vp, u32, vp, u32)
    sig("td_format_csv_row", C.c_int64, vp, u64, vp, u64)
    sig("td_get_counts", i32, vp, vp)
    sig("td_get_stats", i32, vp, C.POINTER(u64))
    sig("td_get_progress", i32, vp, C.POINTER(u64), u64, C.POINTER(u64))
    sig("td_split_progress", i32, vp, C.POINTER(u64), u64, C.POINTER(u64))
    sig("td_set_option", i32, vp, C.c_char_p, C.c_int64)
    sig("td_kernel_time_ms", i32, vp, C.POINTER(C.c_double), C.POINTER(u32))
    sig("td_kernel_times_ms", i32, vp, C.POINTER(C.c_double), u32, C.POINTER(u32))
    sig("td_debug_counters", i32, vp, C.POINTER(u64))
    sig("td_dev_alloc", i32, vp, u64, C.POINTER(vp))
    sig("td_dev_free", i32, vp, vp)
    sig("td_memcpy_h2d", i32, vp, vp, vp, u64)
    sig("td_memcpy_d2h", i32, vp, vp, vp, u64)
    sig("td_device_sync", i32, vp)
    sig("td_synth_fill_device", i32, vp, vp, u64, u64, C.c_char_p, vp, C.c_char_p, C.c_char_p, vp, vp, vp)
    sig("td_synth_expected_device", i32, vp, vp, u64, u64, vp, C.POINTER(u64), vp)
    _lib = L
    return L


def runtime_path():
    return _runtime_path


EXPORTS = [
    "td_last_error", "td_last_bad_index", "td_create", "td_destroy", "td_set_index",
    "td_bind_counts", "td_reset", "td_count_device", "td_count_host", "td_count_file",
    "td_count_lines_device", "td_load_file_range", "td_bgzf_index", "td_bgzf_inflate_range", "td_gunzip_file", "td_gzip_check", "td_gunzip_file_gpu", "td_last_gz_route", "td_gz_shard_open", "td_gz_shard_decode", "td_gz_shard_resolve", "td_crc32_join", "td_set_splitter", "td_split_device", "td_count_and_split_device", "td_split_file", "td_fold_rows", "td_inflate_raw_host", "td_format_csv_row", "td_get_counts", "td_get_stats", "td_get_progress", "td_split_progress", "td_set_option",
    "td_kernel_time_ms", "td_kernel_times_ms", "td_debug_counters", "td_dev_alloc", "td_dev_free", "td_memcpy_h2d", "td_memcpy_d2h",
    "td_device_sync", "td_synth_fill_device", "td_synth_expected_device",
]


def check(rc):
    if rc == 0:
        return
    L = load()
    msg = (L.td_last_error() or b"").decode("utf-8", "replace")
    if rc == -3:      # reference tagdigger_fun.py:82
        raise AssertionError("Problematic sequence: {}.  Likely due to overlapping tags.".format(
            L.td_last_bad_index()))
    if rc == -4:      # reference :76 on an empty list
        raise IndexError("list index out of range")
    if rc == -5:      # reference dies at its first lookup; see DESIGN.md
        raise IndexError("string index out of range")
    if rc == -8:
        raise NonAsciiSequence(msg)
    if rc == -12:     # int() of a malformed count= header, reference :253
        raise ValueError(msg)
    # a .gz input that ends the way gzip.open ends it in the reference (:240-243, :250): same class, same message
    if rc == -13:
        raise EOFError(msg)
    if rc == -14:
        import gzip
        raise gzip.BadGzipFile(msg)
    if rc == -15:
        import zlib
        raise zlib.error(msg)
    raise TagdigError(rc, msg)
