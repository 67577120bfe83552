"""a3 / a4 pinned DIRECTLY on the device index: the lookup tables and build-error cases the real reference produced
(tests/golden/hotpath_primitives.json["lookup"], hotpath_errors.json; build_sequence_tree / sequence_index_lookup,
reference tagdigger_fun.py:71-134) replayed through the C-ABI -- td_set_index with the recorded `numseq`, then one
read per query through td_count_host -- not only through the CPU checker (tests/test_oracle_golden.py).

How a recorded result becomes observable on the device (no checker involved, string operations on the fixture only):
  * barcode side (LDS directory; numseq may differ from len(sequences): rows wrap):  the table's sequences are the
    barcode+site list, row = index mod numseq, tag offset of a row = the length of its sequences, tags = A, C, G, T.
      result r >= 0: the read is the query cut at its first non-base character + "AAA".  The stored set is prefix-free
                     after the shadow rules, so the one stored prefix of the query is the one stored prefix of that
                     read; what follows it is all bases: exactly one count, in row r.
      result -1:     the read is the query + "N" (N ends every match, :119-123): no barcode, no count.
  * tag side (hash table in global memory; numseq == len(sequences)): ONE 1-base barcode+site "T", tags = the table's
    sequences, read = "T" + query + "N": one count in column r, or none.
Queries that the read path would change before the lookup (str.strip().upper(), :256) are left to hotpath_cases.json.
A table whose first sequence is empty (not the :109 special case) fails at build time here (TD_E_ROOTLEAF, a
documented divergence: the reference fails lazily, at its first lookup -- the recorded IndexError / TypeError)."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

PRIM = load_golden("hotpath_primitives.json")
ERRORS = load_golden("hotpath_errors.json")
BASES = set("ACGT")

pytestmark = pytest.mark.gpu


def _strings(xs):
    arr = (C.c_char_p * max(1, len(xs)))()
    for i, x in enumerate(xs):
        arr[i] = x.encode()
    return arr


class RawIndex:
    """td_create / td_set_index / td_count_host / td_get_counts by hand (no Engine set-up in between)."""

    def __init__(self):
        from tagdigger_amd import _binding as B
        self.B, self.L = B, B.load()
        self.h = C.c_void_p()
        assert self.L.td_create(C.byref(self.h), 0) == 0

    def close(self):
        self.L.td_destroy(self.h)

    def set_index(self, barcut, barnum, tagoff, tags):
        off = (C.c_uint32 * max(1, barnum))(*tagoff)
        rc = self.L.td_set_index(self.h, _strings(barcut), len(barcut), barnum, off, _strings(tags), len(tags))
        self.barnum, self.ntags = barnum, len(tags)
        return rc

    def lookup(self, read):
        """counts matrix and (reads, barcut, tag) after ONE record holding `read`"""
        assert self.L.td_reset(self.h) == 0
        rec = b"@q\n" + read.encode() + b"\n+\n" + b"I" * len(read) + b"\n"
        lines = C.c_uint64(0)
        assert self.L.td_count_host(self.h, rec, len(rec), 0, 1 << 62, 0, C.byref(lines)) == 0
        out = np.zeros(self.barnum * self.ntags, dtype=np.uint64)
        assert self.L.td_get_counts(self.h, C.c_void_p(out.ctypes.data)) == 0
        st = (C.c_uint64 * self.B.TD_STAT_NSTATS)()
        assert self.L.td_get_stats(self.h, st) == 0
        return out.reshape(self.barnum, self.ntags), (st[0], st[1], st[2])


def _usable(q):
    return q == q.strip().upper() and all(ord(c) < 128 for c in q)


def _row_offsets(seqs, numseq):
    off = [None] * numseq
    for i, s in enumerate(seqs):
        r = i % numseq
        if off[r] is None:
            off[r] = len(s)
        assert off[r] == len(s), "fixture: one row, two lengths"
    return [o or 0 for o in off]


def _first_empty(tab):
    return len(tab["sequences"]) > 1 and tab["sequences"][0] == ""


@pytest.mark.parametrize("tab", PRIM["lookup"], ids=lambda t: "|".join(t["sequences"])[:40] or "lone-empty")
def test_lookup_table_on_the_barcode_directory(tab):
    seqs, numseq = tab["sequences"], tab["numseq"]
    ix = RawIndex()
    try:
        rc = ix.set_index(seqs, numseq, _row_offsets(seqs, numseq), ["A", "C", "G", "T"])
        if _first_empty(tab):
            assert rc == -5                                       # TD_E_ROOTLEAF (the reference: lazy IndexError / TypeError)
            assert any(isinstance(r, dict) for r in tab["result"])
            return
        assert rc == 0, ix.L.td_last_error()
        checked = 0
        for q, want in zip(tab["queries"], tab["result"]):
            if not _usable(q):
                continue
            if want >= 0:
                cut = next((i for i, c in enumerate(q) if c not in BASES), len(q))
                m, st = ix.lookup(q[:cut] + "AAA")
                assert st == (1, 1, 1), (q, st)
                assert int(m.sum()) == 1 and int(m[want].sum()) == 1, (q, want, np.argwhere(m))
            else:
                m, st = ix.lookup(q + "N")
                assert st == (1, 0, 0) and int(m.sum()) == 0, (q, st)
            checked += 1
        assert checked >= 3
    finally:
        ix.close()


@pytest.mark.parametrize("tab", [t for t in PRIM["lookup"] if t["numseq"] == len(t["sequences"])],
                         ids=lambda t: "|".join(t["sequences"])[:40] or "lone-empty")
def test_lookup_table_on_the_tag_hash_table(tab):
    seqs = tab["sequences"]
    ix = RawIndex()
    try:
        rc = ix.set_index(["T"], 1, [1], seqs)
        if _first_empty(tab):
            assert rc == -5
            return
        assert rc == 0, ix.L.td_last_error()
        checked = 0
        for q, want in zip(tab["queries"], tab["result"]):
            if not _usable(q):
                continue
            m, st = ix.lookup("T" + q + "N")
            if want >= 0:
                assert st == (1, 1, 1) and int(m.sum()) == 1 and int(m[0, want]) == 1, (q, want, st, np.argwhere(m))
            else:
                assert st == (1, 1, 0) and int(m.sum()) == 0, (q, st)
            checked += 1
        assert checked >= 3
    finally:
        ix.close()


@pytest.mark.parametrize("side", ["barcodes", "tags"])
@pytest.mark.parametrize("e", ERRORS, ids=lambda e: ",".join(e["sequences"]) or "empty")
def test_build_errors_through_td_set_index(e, side):
    """hotpath_errors.json: what build_sequence_tree raised (or not) for a sequence list, on either index of
    td_set_index: AssertionError 'Problematic sequence: k' -> TD_E_OVERLAP with td_last_bad_index() == k;
    the empty list -> TD_E_EMPTY; a first-empty list (built lazily-broken by the reference) -> TD_E_ROOTLEAF."""
    seqs, numseq = e["sequences"], e["numseq"]
    if side == "tags" and numseq != len(seqs):
        pytest.skip("the tag index has numseq == len(tags) (tagdigger_fun.py:233)")
    ix = RawIndex()
    try:
        if side == "barcodes":
            off = [0] * max(1, numseq)
            for i, s in enumerate(seqs):
                off[i % numseq] = len(s)
            rc = ix.set_index(seqs, numseq, off, ["A", "C", "G", "T"])
        else:
            rc = ix.set_index(["T"], 1, [1], seqs)
        if e.get("ok"):
            if len(seqs) > 1 and seqs[0] == "":
                assert rc == -5
            else:
                assert rc == 0, ix.L.td_last_error()
        elif e["raises"] == "AssertionError":
            assert rc == -3
            k = ix.L.td_last_bad_index()
            assert e["message"] == "Problematic sequence: {}.  Likely due to overlapping tags.".format(k)
            with pytest.raises(AssertionError) as ei:             # ... and what the binding makes of it
                ix.B.check(rc)
            assert str(ei.value) == e["message"]
        else:
            assert e["raises"] == "IndexError" and rc == -4
            with pytest.raises(IndexError) as ei:
                ix.B.check(rc)
            assert str(ei.value) == e["message"]
    finally:
        ix.close()
