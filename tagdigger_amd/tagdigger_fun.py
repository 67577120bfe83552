"""Host-side mirror of the reference's tagdigger_fun module for the counting path.

`find_tags_fastq` has the reference's signature, defaults, return type and
exceptions (tagdigger_fun.py:192-277) but its record loop runs on an MI355X.
The index primitives it is built from are re-exported under their reference
names so that code written against the reference module keeps working.
"""
from .engine import (Engine, default_engine, enumerate_cut_sites,  # noqa: F401
                     combine_barcode_and_cutsite, effective_maxreads)

# restriction enzyme cut sites as they appear after the barcode (reference tagdigger_fun.py:19-20)
enzymes = {'ApeKI': 'CWGC', 'EcoT22I': 'TGCAT', 'NcoI': 'CATGG',
           'NsiI': 'TGCAT', 'PstI': 'TGCAG', 'SbfI': 'TGCAGG', 'None': ''}


def find_tags_fastq(fqfile, barcodes, tags, cutsite="TGCAG", maxreads=5e9, tassel_tagcount=False,
                    device=0):
    """Count barcode x tag combinations in one FASTQ file (plain or .gz by name).

    Returns list[list[int]] shaped [len(barcodes)][len(tags)], rows and columns
    in the order given -- exactly what the reference returns (tagdigger_fun.py
    :237,:277).  Differences, all documented in DESIGN.md: no progress prints;
    a sequence line holding a byte >= 0x80 raises NonAsciiSequence; an index
    whose first sequence is empty raises IndexError at build time (the
    reference raises at its first lookup).
    """
    eng = default_engine(device)
    eng.set_index(barcodes, tags, cutsite)          # asserts + index, before the file is opened (:198-233)
    eng.count_file(fqfile, maxreads, tassel_tagcount)
    return eng.counts(signed=bool(tassel_tagcount))
