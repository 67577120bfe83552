"""The CPU oracle (oracle/tagdigger_oracle.py) against every fixture captured
from the real reference (tests/golden/make_golden.py).  CPU only."""
import pytest

from conftest import load_golden, write_case_file
from oracle import tagdigger_oracle as orc

PRIM = load_golden("hotpath_primitives.json")
CASES = load_golden("hotpath_cases.json") + load_golden("hotpath_random.json")
ERRORS = load_golden("hotpath_errors.json")

EXC = {"AssertionError": AssertionError, "IndexError": IndexError, "TypeError": TypeError,
       "ValueError": ValueError, "FileNotFoundError": FileNotFoundError}


@pytest.mark.parametrize("cutsite", sorted(PRIM["enumerate_cut_sites"]))
def test_enumerate_cut_sites(cutsite):
    assert orc.enumerate_cut_sites(cutsite) == PRIM["enumerate_cut_sites"][cutsite]


@pytest.mark.parametrize("tab", PRIM["lookup"], ids=lambda t: "|".join(t["sequences"])[:40])
def test_lookup_tables(tab):
    tree = orc.build_sequence_tree(tab["sequences"], tab["numseq"])
    for q, want in zip(tab["queries"], tab["result"]):
        if isinstance(want, dict):
            with pytest.raises(EXC[want["raises"]]):
                orc.sequence_index_lookup(q, tree)
        else:
            assert orc.sequence_index_lookup(q, tree) == want, q


@pytest.mark.parametrize("e", ERRORS, ids=lambda e: ",".join(e["sequences"]) or "empty")
def test_build_errors(e):
    if e.get("ok"):
        orc.build_sequence_tree(e["sequences"], e["numseq"])
    else:
        with pytest.raises(EXC[e["raises"]]) as ei:
            orc.build_sequence_tree(e["sequences"], e["numseq"])
        assert str(ei.value) == e["message"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_find_tags_fastq(case, tmp_path):
    path = write_case_file(case, tmp_path)
    if case.get("filename_override"):
        path = str(tmp_path / "nope" / case["filename_override"])
    if "raises" in case:
        with pytest.raises(EXC[case["raises"]]) as ei:
            orc.find_tags_fastq(path, case["barcodes"], case["tags"], **case["kwargs"])
        if case["message"]:
            assert str(ei.value) == case["message"]
    else:
        got = orc.find_tags_fastq(path, case["barcodes"], case["tags"], **case["kwargs"])
        assert got == case["counts"]


# ---- the C restatement (oracle/oracle.c) against the same fixtures ----------
from oracle import c_oracle  # noqa: E402


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_c_oracle_find_tags_fastq(case, tmp_path):
    path = write_case_file(case, tmp_path)
    if case.get("filename_override"):
        path = str(tmp_path / "nope" / case["filename_override"])
    if "raises" in case:
        with pytest.raises(EXC[case["raises"]]) as ei:
            c_oracle.find_tags_fastq(path, case["barcodes"], case["tags"], **case["kwargs"])
        if case["message"] and case["raises"] != "ValueError":
            assert str(ei.value) == case["message"]
    else:
        got = c_oracle.find_tags_fastq(path, case["barcodes"], case["tags"], **case["kwargs"])
        assert got == case["counts"]


@pytest.mark.parametrize("e", ERRORS, ids=lambda e: ",".join(e["sequences"]) or "empty")
def test_c_oracle_build_errors(e):
    import ctypes as C
    L = c_oracle.lib()
    seqs = e["sequences"]
    arr = (C.c_char_p * max(1, len(seqs)))(*[s.encode() for s in seqs])
    one = (C.c_char_p * 1)(b"A")
    h = C.c_void_p()
    bad = C.c_uint32()
    # build the barcode side from `seqs`, a trivial tag side
    rc = L.orc_build(arr, len(seqs), e["numseq"], one, 1, C.byref(h), C.byref(bad))
    if e.get("ok"):
        assert rc == 0
        L.orc_free(h)
    elif e["raises"] == "AssertionError":
        assert rc == -1 and "Problematic sequence: %d." % bad.value in e["message"]
    else:
        assert rc == -2


def test_iter_lines_matches_splitlines():
    import random
    rnd = random.Random(5)
    for _ in range(300):
        data = bytes(rnd.choice(b"AC\n\r\r\n\x0b\x0c\x1c\x85 ") for _ in range(rnd.randint(0, 40)))
        assert list(orc.iter_lines(data)) == data.splitlines()
