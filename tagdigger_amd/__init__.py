"""tagdigger_amd -- MI355X-native tag counting behind TagDigger's own call boundary.

The hot path (reference tagdigger_fun.find_tags_fastq, tagdigger_fun.py:192-277)
runs as hand-written HIP kernels for gfx950 inside libtagdig.so; this package is
the thin Python host side: `tagdigger_fun` mirrors the reference module's
functions for that path (same names, arguments, results and exceptions) and
`Engine` drives one GPU.  There is no CPU fallback.
"""
from .engine import Engine, default_engine  # noqa: F401
from ._binding import TagdigError, NonAsciiSequence  # noqa: F401

__all__ = ["Engine", "default_engine", "TagdigError", "NonAsciiSequence"]
