"""Host-side mirror of the reference's tagdigger_fun module for the counting path.

`find_tags_fastq` has the reference's signature, defaults, return type and
exceptions (tagdigger_fun.py:192-277) but its record loop runs on an MI355X.
Of the primitives under it, `enumerate_cut_sites` and `combine_barcode_and_cutsite`
(reference :136-190, :60-69) are exported under their reference names.  The
reference's nested-list trees (`build_sequence_tree` / `sequence_index_lookup`,
:71-134) have no counterpart here: their rules live in the flat device index that
`Engine.set_index` builds (csrc/tagdig.hip, td_set_index) and in the matching
kernels; a restatement of the two functions exists only as test infrastructure
(the CPU checker beside the tests).
"""
from .engine import (Engine, default_engine, enumerate_cut_sites,  # noqa: F401
                     combine_barcode_and_cutsite, effective_maxreads)

# restriction enzyme cut sites as they appear after the barcode (reference tagdigger_fun.py:19-20)
enzymes = {'ApeKI': 'CWGC', 'EcoT22I': 'TGCAT', 'NcoI': 'CATGG',
           'NsiI': 'TGCAT', 'PstI': 'TGCAG', 'SbfI': 'TGCAGG', 'None': ''}


def find_tags_fastq(fqfile, barcodes, tags, cutsite="TGCAG", maxreads=5e9, tassel_tagcount=False,
                    device=0, as_array=False, progress=True):
    """Count barcode x tag combinations in one FASTQ file (plain or .gz by name).

    Returns list[list[int]] shaped [len(barcodes)][len(tags)], rows and columns
    in the order given -- exactly what the reference returns (tagdigger_fun.py
    :237,:277) -- and prints the reference's progress lines (:268-271: the three
    counters after every 50 000 reads, the file name after every 1 000 000; the
    numbers are kept on the device per window of reads and printed once the file
    is through; `progress=False` leaves them out).  Differences, all documented
    in DESIGN.md: a sequence line holding a byte >= 0x80 raises NonAsciiSequence; an index
    whose first sequence is empty raises IndexError at build time (the
    reference raises at its first lookup).
    """
    eng = default_engine(device)
    eng.set_index(barcodes, tags, cutsite)          # asserts + index, before the file is opened (:198-233)
    eng.set_option("progress", 1 if progress else 0)
    eng.count_file(fqfile, maxreads, tassel_tagcount)
    if progress:
        for line in eng.progress_lines(fqfile):
            print(line)
    if as_array:                                    # (this build only: the matrix as a numpy array, no Python lists)
        return eng.counts_numpy(signed=bool(tassel_tagcount))
    return eng.counts(signed=bool(tassel_tagcount))


def find_tags_fastq_many(files, barcodes, tags, cutsite="TGCAG", maxreads=5e9, tassel_tagcount=False,
                         devices=(0,)):
    """`find_tags_fastq` over several files, the files dealt out to the GPUs in `devices` and counted
    concurrently (one host thread and one engine per GPU; the library calls release the GIL).

    `barcodes` is either one list used for every file or a list of lists, one per file -- the way
    the reference's callers loop `find_tags_fastq(f, bckeys[f][0], tags[1], cutsite=...)` over the
    files of a key file (tagdigger_script.py:124-126).  Returns the matrices in the order of `files`;
    each equals the single-file call's.  An exception in any file is raised after the others finish.
    (Several processes instead of threads, with an all-reduce into sample rows: tagdigger_amd.multi.)
    """
    import threading
    files = list(files)
    per_file = bool(barcodes) and isinstance(barcodes[0], (list, tuple))
    if per_file and len(barcodes) != len(files):
        raise ValueError("one barcode list per file expected")
    devs = list(dict.fromkeys(int(d) for d in devices))
    if not devs:
        raise ValueError("no device given")
    results, errors = [None] * len(files), [None] * len(files)
    todo = list(range(len(files)))
    lock = threading.Lock()

    def work(dev):
        while True:
            with lock:
                if not todo:
                    return
                k = todo.pop(0)
            try:
                results[k] = find_tags_fastq(files[k], barcodes[k] if per_file else barcodes, tags, cutsite=cutsite,
                                             maxreads=maxreads, tassel_tagcount=tassel_tagcount, device=dev,
                                             progress=False)        # (several files at once: no interleaved prints)
            except BaseException as exc:      # noqa: BLE001 -- reported to the caller below
                errors[k] = exc
    threads = [threading.Thread(target=work, args=(d,)) for d in devs[1:]]
    for t in threads:
        t.start()
    work(devs[0])
    for t in threads:
        t.join()
    for exc in errors:
        if exc is not None:
            raise exc
    return results


# =============================================================================
# Periphery of the counting path: plain host
# Python, no acceleration -- kept so that the reference's command line stays a
# drop-in.  Same return values, same printed messages, same files written.
# =============================================================================
import bisect as _bisect
import csv as _csv
import gzip as _gzip
import re as _re


def isFastq(filename):
    """1 for a plain FASTQ file, 2 for a gzipped one (chosen by name), 0 otherwise or if it
    cannot be opened -- judged from the first three lines only (reference tagdigger_fun.py:279-307)."""
    gz = filename[-2:].lower() == 'gz'
    try:
        con = _gzip.open(filename, 'rt') if gz else open(filename, 'r')
    except IOError:
        return 0
    verdict = 2 if gz else 1
    try:
        header = con.readline()
        if header[0] != '@':
            verdict = 0
        if not set(con.readline().strip()) <= set('ACGTNacgtn'):
            verdict = 0
        if con.readline()[0] != '+':
            verdict = 0
    finally:
        con.close()
    return verdict


def readBarcodeKeyfile(filename, forSplitter=False):
    """Key file (CSV with File/Barcode/Sample columns, any order) -> {file: [[barcodes], [samples]]},
    or None after printing what is wrong (reference tagdigger_fun.py:309-374)."""
    cols = ("Input File", "Barcode", "Output File") if forSplitter else ("File", "Barcode", "Sample")
    try:
        result = {}
        with open(filename, 'r', newline='') as con:
            where = None
            rownum = 1                                         # the reference's row numbers skip blank lines
            for row in _csv.reader(con):
                if where is None:
                    where = [row.index(c) for c in cols]      # ValueError if a column is missing
                    continue
                f, b, s = row[where[0]].strip(), row[where[1]].strip().upper(), row[where[2]].strip()
                if f == "" and b == "" and s == "":
                    continue
                rownum += 1
                if f == "":
                    raise Exception("Blank cell found where file name should be in row {}.".format(rownum))
                if s == "":
                    raise Exception("Blank cell found where sample name should be in row {}.".format(rownum))
                if not set(b) <= set('ACGT'):
                    raise Exception("{0} in row {1} is not a valid barcode.".format(b, rownum))
                entry = result.setdefault(f, [[], []])
                if b in entry[0]:
                    raise Exception("Each barcode can only be present once for each file.")
                entry[0].append(b)
                entry[1].append(s)
        if forSplitter:
            outs = [s for v in result.values() for s in v[1]]
            if len(set(outs)) < len(outs):
                raise Exception("All output files must have unique names for barcode splitter.")
    except IOError:
        print("Could not read file {}.".format(filename))
        return None
    except ValueError:
        print("File header needed containing '{}', '{}', and '{}'.".format(*cols))
        return None
    except Exception as err:
        print(err.args[0])
        return None
    return result


def readMarkerNames(filename):
    """List of marker names to keep: commas and surrounding whitespace dropped, empty lines
    skipped (reference tagdigger_fun.py:921-934)."""
    try:
        with open(filename, mode='r') as con:
            lines = con.readlines()
    except IOError:
        print("File {} not readable.".format(filename))
        return None
    cleaned = [ln.replace(",", "").strip() for ln in lines]
    return [x for x in cleaned if x != ""]


def compareTags(taglist, trim=True):
    """Variable sites among tags of one locus: [(position, [base per tag]), ...]
    (reference tagdigger_fun.py:376-393)."""
    assert type(taglist) is list, "taglist must be list."
    assert all([set(t) <= set('ATCG') for t in taglist]), "taglist must be a list of ACGT strings."
    lengths = set(len(t) for t in taglist)
    if len(lengths) > 1:
        if trim:
            taglist = [t[:min(lengths)] for t in taglist]
        else:
            taglist = [t.ljust(max(lengths), 'N') for t in taglist]
    out = []
    for i in range(len(taglist[0])):
        col = [t[i] for t in taglist]
        if len(set(c for c in col if c != 'N')) > 1:
            out.append((i, col))
    return out


class _SeqList(list):
    """The list of tag sequences a reader builds, with a set beside it: the reference's uniqueness
    checks are `x in seqlist` on a plain list (quadratic; hours at 500 k tags).  Same answers."""
    def __init__(self):
        super().__init__()
        self._seen = set()

    def __contains__(self, x):
        return x in self._seen

    def append(self, x):
        self._seen.add(x)
        super().append(x)

    def extend(self, xs):
        xs = list(xs)
        self._seen.update(xs)
        super().extend(xs)


def _keep_set(toKeep):
    """Marker names to keep as a set (the reference tests `name in toKeep` on the list)."""
    return None if toKeep is None else set(toKeep)


def _read_tag_table(filename, needed, header_msg, handle_row):
    """Shared skeleton of the CSV tag readers: header check, per-row callback, error printing."""
    names, seqs = [], _SeqList()
    try:
        with open(filename, mode='r') as con:
            where = None
            for rownum, row in enumerate(_csv.reader(con), start=1):
                if where is None:
                    if not set(needed) <= set(row):
                        raise Exception(header_msg)
                    where = [row.index(c) for c in needed]
                else:
                    handle_row(row, where, rownum, names, seqs)
    except IOError:
        print("File {} not readable.".format(filename))
        return None
    except Exception as err:
        print(err.args[0])
        return None
    return [names, list(seqs)]


def readTags_Merged(filename, toKeep=None, allowDuplicates=False):
    """Merged format: marker name + tag with the variable region as [A/C] (reference
    tagdigger_fun.py:563-618).  Names come out as marker_allele_index."""
    toKeep = _keep_set(toKeep)

    def row_fn(row, where, rownum, names, seqs):
        cell = row[where[1]]
        if not set('[/]') < set(cell):
            raise Exception("Characters '[/]' not found in row {}.".format(rownum))
        marker = row[where[0]].strip()
        if '_' in marker:
            raise Exception("Marker {}: marker names cannot contain underscores.".format(row[where[0]]))
        if toKeep != None and marker not in toKeep:
            return
        left, right = cell.find('['), cell.find(']')
        alleles = [a.strip().upper() for a in cell[left + 1:right].split('/')]
        tags = [(cell[:left] + a + cell[right + 1:]).upper().strip().replace('-', '') for a in alleles]
        if not allowDuplicates and any([t in seqs for t in tags]):
            print("Non-unique sequence found: line {0}.".format(rownum))
            print("Marker {} skipped.".format(marker))
            return
        seqs.extend(tags)
        if not all([set(t) <= set('ACGT') for t in tags]):
            raise Exception("Tag sequence not formatted correctly in row {}.".format(rownum))
        names.extend(["{}_{}_{}".format(marker, alleles[i], i) for i in range(len(tags))])
    return _read_tag_table(filename, ("Marker name", "Tag sequence"),
                           "Need 'Marker name' and 'Tag sequence' in header row.", row_fn)


def readTags_Rows(filename, toKeep=None):
    """One row per allele: marker name, allele name, tag (reference tagdigger_fun.py:475-514)."""
    toKeep = _keep_set(toKeep)

    def row_fn(row, where, rownum, names, seqs):
        marker = row[where[0]].strip()
        if '_' in marker:
            raise Exception("Marker {}: marker names cannot contain underscores.".format(marker))
        if toKeep != None and marker not in toKeep:
            return
        allele = row[where[1]].strip()
        tag = row[where[2]].upper().strip()
        if not set(tag) <= set('ACGT'):
            raise Exception("Tag sequence not formatted as ACGT in row {}.".format(rownum))
        if tag in seqs:
            raise Exception("Non-unique sequence found: line {0}.".format(rownum))
        names.append(marker + '_' + allele)
        seqs.append(tag)
    return _read_tag_table(filename, ("Marker name", "Allele name", "Tag sequence"),
                           "Need 'Marker name', 'Allele name', and 'Tag sequence' in header row.", row_fn)


def readTags_Columns(filename, toKeep=None):
    """One row per marker with two tags (reference tagdigger_fun.py:516-561)."""
    toKeep = _keep_set(toKeep)

    def row_fn(row, where, rownum, names, seqs):
        marker = row[where[0]].strip()
        if '_' in marker:
            raise Exception("Marker {}: marker names cannot contain underscores.".format(marker))
        if toKeep != None and marker not in toKeep:
            return
        tag0, tag1 = row[where[1]].upper().strip(), row[where[2]].upper().strip()
        if not set(tag0 + tag1) <= set('ACGT'):
            raise Exception("Tag sequence not formatted as ACGT in row {}.".format(rownum))
        if tag0 in seqs or tag1 in seqs:
            raise Exception("Non-unique sequence found: line {0}.".format(rownum))
        seqs.extend([tag0, tag1])
        diff = compareTags([tag0, tag1])
        names.append(marker + '_' + ''.join(d[1][0] for d in diff) + '_0')
        names.append(marker + '_' + ''.join(d[1][1] for d in diff) + '_1')
    return _read_tag_table(filename, ("Marker name", "Tag sequence 0", "Tag sequence 1"),
                           "Need 'Marker name', 'Tag sequence 0', and 'Tag sequence 1' in header row.", row_fn)


def readTags_UNEAK_FASTA(filename, toKeep=None):
    """Tag pairs from a TASSEL-UNEAK FASTA: four lines per pair, '>TPn_query_len' / sequence /
    '>TPn_hit_len' / sequence (reference tagdigger_fun.py:395-473)."""
    names, seqs = [], _SeqList()
    toKeep = _keep_set(toKeep)
    try:
        with open(filename, mode='r') as con:
            name1 = name2 = seq1 = seq2 = None
            len1 = len2 = 0
            for n, line in enumerate(con):
                part = n % 4
                if part in (0, 2):
                    if line[:3] != ">TP":
                        raise Exception("Line {0} of {1} does not start with '>TP'.".format(n + 1, filename))
                    cut = line.rfind("_")
                    if part == 0:
                        name1 = line[1:cut]
                        len1 = int(line[cut + 1:].strip())     # real tag length (some are padded with A's)
                    else:
                        name2 = line[1:cut]
                        if name1[:name1.find("_")] != name2[:name2.find("_")]:
                            raise Exception("Tag name in line {0} does not match tag name in line {1}.".format(n + 1, n - 1))
                        len2 = int(line[cut + 1:].strip())
                    continue
                seq = line.strip().upper()
                seq = seq[:len1] if part == 1 else seq[:len2]
                if not set(seq) <= set('ACGT'):
                    raise Exception("Line {0} is not ACGT sequence.".format(n + 1))
                if seq in seqs:
                    raise Exception("Non-unique sequence found: line {0}.".format(n + 1))
                if part == 1:
                    seq1 = seq
                    continue
                seq2 = seq
                marker = name1[:name1.find("_")]
                if toKeep != None and marker not in toKeep:
                    continue
                shortest = min(len1, len2)
                if len1 != len2 and seq1[:shortest] == seq2[:shortest]:
                    print("{} skipped because tags cannot be distinguished.".format(marker))
                    continue
                diff = compareTags([seq1, seq2])
                base1, base2 = diff[0][1][0], diff[0][1][1]
                first_is_0 = base1 < base2                       # alphabetical, to match hapMap2numeric
                names.extend([name1 + "_" + base1 + ("_0" if first_is_0 else "_1"),
                              name2 + "_" + base2 + ("_1" if first_is_0 else "_0")])
                seqs.extend([seq1[:shortest], seq2[:shortest]])
    except IOError:
        print("File {} not readable.".format(filename))
        return None
    except Exception as err:
        print(err.args[0])
        return None
    return [names, list(seqs)]


def reverseComplement(sequence):
    """Reverse complement; characters other than A, C, G, T pass through unchanged (reference
    tagdigger_fun.py:1203-1206)."""
    return sequence.translate({65: 'T', 67: 'G', 71: 'C', 84: 'A'})[::-1]


def _open_maybe_gz(path):
    """Text-mode handle; gzip when the name ends in '.gz' (as the Stacks reader decides, :630)."""
    return _gzip.open(path, mode='rt') if path.endswith('.gz') else open(path, mode='r')


def _stacks_rows(path):
    """Rows of a Stacks catalog table (tab separated), comment rows ('#...') left out."""
    with _open_maybe_gz(path) as con:
        for row in _csv.reader(con, delimiter='\t'):
            if not row[0].startswith("#"):
                yield row


def readTags_Stacks(tagsfile, snpsfile, allelesfile, toKeep=None, binaryOnly=False, version=1):
    """Tags from a Stacks catalog: consensus sequences (tags.tsv), SNP columns (snps.tsv) and
    haplotypes (alleles.tsv) -> one tag per haplotype, named locus_haplotype (reference
    tagdigger_fun.py:620-719).  The column layout depends on the Stacks version."""
    if version == 1:
        col_locus, col_seq, col_hap, col_pos = 2, 9, 3, 3
    else:
        col_locus, col_seq, col_hap, col_pos = 1, 5, 2, 2
    wanted = lambda locus: toKeep == None or locus in toKeep
    try:
        consensus = {}
        for row in _stacks_rows(tagsfile):
            if wanted(row[col_locus]):
                consensus[row[col_locus]] = row[col_seq]
        haplotypes = []
        for row in _stacks_rows(allelesfile):
            if wanted(row[col_locus]):
                haplotypes.append((row[col_locus], row[col_hap]))
        snp_columns = {}
        for row in _stacks_rows(snpsfile):
            if wanted(row[col_locus]):
                snp_columns.setdefault(row[col_locus], []).append(int(row[col_pos]))

        names, seqs = [], []
        for locus, hap in haplotypes:
            seq = consensus[locus]
            if len(hap) > 0:
                # the haplotype's k-th character replaces the consensus base at the k-th SNP column
                cols = snp_columns[locus]
                pieces = [seq[:cols[0]]]
                for k, base in enumerate(hap):
                    pieces.append(base)
                    pieces.append(seq[cols[k] + 1:] if k + 1 == len(hap) else seq[cols[k] + 1:cols[k + 1]])
                seq = "".join(pieces)
            seq = seq.upper()
            if set(seq) <= set('ACGT'):
                names.append(locus + '_' + hap)
                seqs.append(seq)
            else:
                print("{}_{} skipped for having non-ACGT nucleotides.".format(locus, hap))
        if binaryOnly:
            # loci with exactly two haplotypes; 0 / 1 by alphabetical order of the haplotypes
            kept_names, kept_seqs = [], []
            for alleles, where in extractMarkers(names)[1]:
                if len(alleles) != 2:
                    continue
                first_is_0 = alleles[0] < alleles[1]
                kept_names.append(names[where[0]] + ('_0' if first_is_0 else '_1'))
                kept_names.append(names[where[1]] + ('_1' if first_is_0 else '_0'))
                kept_seqs.extend([seqs[where[0]], seqs[where[1]]])
            names, seqs = kept_names, kept_seqs
        return [names, seqs]
    except IOError:
        print("Files not readable.")
    except (IndexError, ValueError):
        print("Files in wrong format.")
    except KeyError:
        print("Locus names not matching properly.")
    except Exception as err:
        print(err.args[0])
    return None


_SAM_UNALIGNED = {4 + f for f in (0, 1, 2, 8, 16, 32, 64, 128)}     # the forms of flag 4 the reference tests for (:750)
_SAM_REVERSE = {16 + f for f in (0, 1, 2, 8, 32, 64, 128)}          # and of flag 16 (:761)


def readTags_TASSELSAM(filename, toKeep=None, binaryOnly=False, noMonomorphic=False,
                       writeMarkerKey=False, keyfilename=None):
    """Tags from a SAM file of aligned TASSEL-GBSv2 tags; alignments that start at the same
    position and strand are the tags of one marker, named chromosome-position-strand (reference
    tagdigger_fun.py:721-854).  toKeep holds TASSEL SNP names (e.g. S03_350622)."""
    assert (not writeMarkerKey) or keyfilename != None, "keyfilename needed."
    names, seqs, snp_key = [], [], []
    by_marker = {}
    width = 0            # digits of the longest chromosome length seen so far (@SQ LN:)
    try:
        with open(filename, mode='r') as con:
            for line in con:
                if line[0:3] == '@SQ':
                    width = max(width, len(line.split()[2][3:]))
                    continue
                if line[0] == '@':
                    continue
                f = line.split()
                flags = int(f[1])
                if flags in _SAM_UNALIGNED:
                    continue
                chrom = f[2].replace('_', '*')            # '_' separates marker from allele in tag names
                pos = int(f[3])
                seq = f[9]
                strand = "top"
                if flags in _SAM_REVERSE:
                    strand = "bot"
                    seq = reverseComplement(seq)
                    # the tag starts at the cut site, i.e. at the alignment's last reference base
                    cigar = f[5]
                    deleted = sum(int(x[:-1]) for x in _re.findall(r'\d+D', cigar))
                    inserted = sum(int(x[:-1]) for x in _re.findall(r'\d+I', cigar))
                    pos = pos + len(seq) - inserted + deleted - 1
                marker = "{}-{:0>{width}}-{}".format(chrom, pos, strand, width=width)
                if marker not in by_marker:
                    by_marker[marker] = [seq]
                    continue
                # of two tags one of which is a prefix of the other, the shorter stays (:776-785)
                kept = [t for t in by_marker[marker] if not t.startswith(seq)]
                by_marker[marker] = kept
                if not any(seq.startswith(t) for t in kept):
                    kept.append(seq)

        for marker in sorted(by_marker):
            tags = by_marker[marker]
            if (binaryOnly and len(tags) != 2) or (noMonomorphic and len(tags) == 1):
                continue
            diff = compareTags(tags, trim=False)
            if toKeep != None or writeMarkerKey:
                chrom, postext, strand = marker.split('-')[:3]
                chrom = chrom.upper()
                if chrom.startswith("CHROMOSOME"):
                    chrom = chrom[10:]
                if chrom.startswith("CHR"):
                    chrom = chrom[3:]
                step = 1 if strand == 'top' else -1
                tassel_names = ['S{}_{}'.format(chrom, int(postext) + step * d[0]) for d in diff]
                if toKeep != None and all(n not in toKeep for n in tassel_names):
                    continue
                if writeMarkerKey:
                    snp_key.extend((n, marker) for n in tassel_names)
            alleles = [''.join(d[1][i] for d in diff) for i in range(len(tags))]
            tagnames = [marker + '_' + a for a in alleles]
            if binaryOnly and alleles[0] != alleles[1]:
                low = 0 if alleles[0] < alleles[1] else 1
                tagnames[low] += '_0'
                tagnames[1 - low] += '_1'
            names.extend(tagnames)
            seqs.extend(tags)
        if len(names) == 0:
            raise Exception("No markers output; is list of markers to keep in right format (e.g. S03_350622)?")
    except IOError:
        print("Could not read file {}.".format(filename))
        return None
    except Exception as err:
        print(err.args[0])
        return None
    if writeMarkerKey:
        try:
            with open(keyfilename, mode='w', newline='') as out:
                w = _csv.writer(out)
                w.writerow(["TASSEL-GBSv2 marker name", "TagDigger marker name"])
                for pair in snp_key:
                    w.writerow(pair)
        except IOError:
            print("Could not write file {}.".format(keyfilename))
            return None
    return [names, seqs]


def _pyrad_locus(seqset, marker, binaryOnly):
    """Names and sequences for the alleles of one pyRAD locus (reference tagdigger_fun.py:865-889):
    trim to the shortest, drop trailing gap columns, drop alleles with N, sort, name by the
    variable columns."""
    n = min(len(x) for x in seqset)
    aligned = [x[:n] for x in seqset]
    while any(x[-1] == '-' for x in aligned):
        aligned = [x[:-1] for x in aligned]
        n -= 1
    aligned = sorted(set(x for x in aligned if 'N' not in x))
    if not ((len(aligned) != 0 and not binaryOnly) or len(aligned) == 2):
        return [], []
    variable = [i for i in range(n) if len(set(x[i] for x in aligned)) > 1]
    names = ['{}_{}_{}'.format(marker, ''.join(x[i] for i in variable), k) for k, x in enumerate(aligned)]
    return names, [x.replace('-', '') for x in aligned]


def readTags_pyRAD(filename, toKeep=None, binaryOnly=False):
    """Tags from a pyRAD .alleles file: '>sample  sequence' lines, each locus closed by a '//' line
    that carries its number (reference tagdigger_fun.py:856-919)."""
    names, seqs = [], []
    current = set()
    linenum = 0
    try:
        with open(filename, mode='r') as con:
            for line in con:
                if line[0] == '>':
                    seq = line.split()[1]
                    if not set(seq) <= set('ACGT-N'):
                        raise Exception("Character other than ACGTN- detected in sequence.")
                    current.add(seq)
                elif line[0] == '/':
                    marker = line.split()[-1][1:-1]
                    for ch in "|*-":
                        marker = marker.replace(ch, "")
                    if toKeep == None or marker in toKeep:
                        n, q = _pyrad_locus(current, marker, binaryOnly)
                        names.extend(n)
                        seqs.extend(q)
                    current = set()
                else:
                    raise Exception("File not in pyRAD format.")
                linenum += 1
    except IOError:
        print("File {} not readable.".format(filename))
        return None
    except Exception as err:
        print("Line {}:".format(linenum))
        print(err.args[0])
        return None
    return [names, seqs]


# adapter sets for the barcode splitter: (restriction site with ^ where genomic sequence ends,
# top-strand adapter after the overhang; [barcode] = reverse complement of the barcode)
# (reference tagdigger_fun.py:27-47)
_P5_BARCODED = '[barcode]AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGTAGATCTCGGTGGTCGCCGTATCATT'
_HALL_COMMON = 'CTCAGGCATCACTCGATTCCTCCGTCGTATGCCGTCTTCTGCTTG'
_CLARK_COMMON = 'CTCAGGCATCACTCGATTCCTATCTCGTATGCCGTCTTCTGCTTG'
adapters = {'PstI-MspI-Hall': [('CCG^G', _HALL_COMMON), ('CTGCA^G', _P5_BARCODED)],
            'NsiI-MspI-Hall': [('CCG^G', _HALL_COMMON), ('ATGCA^T', _P5_BARCODED)],
            'PstI-MspI-Clark': [('CCG^G', _CLARK_COMMON), ('CTGCA^G', _P5_BARCODED)],
            'NsiI-MspI-Clark': [('CCG^G', _CLARK_COMMON), ('ATGCA^T', _P5_BARCODED)],
            'PstI-MspI-Poland': [('CCG^G', 'AGATCGGAAGAGCGGTTCAGCAGGAATGCCGAGACCGATCTCGTATGCCGTCTTCTGCTTG'),
                                 ('CTGCA^G', _P5_BARCODED)]}


def _trie_survivors(sequences):
    """Which of `sequences` (ACGT strings) a prefix trie built in this order would hold, as
    (sequence, position) pairs -- the rules of tree_one_level (reference tagdigger_fun.py:71-86):
    among sequences sharing a path, if the FIRST one ends there it is kept and the others
    (duplicates, extensions) are dropped silently; if a LATER one ends inside the first, that is an
    AssertionError naming its position.  Children are visited in A, C, G, T order."""
    kept = []

    def walk(group, depth):
        first = group[0]
        if len(first[0]) == depth:
            kept.append(first)
            return
        branches = ([], [], [], [])
        for item in group:
            assert len(item[0]) > depth, \
                "Problematic sequence: {}.  Likely due to overlapping tags.".format(item[1])
            branches["ACGT".find(item[0][depth])].append(item)
        for branch in branches:
            if branch:
                walk(branch, depth + 1)

    if sequences:
        walk([(s, k) for k, s in enumerate(sequences)], 0)
    return kept


def _adapter_ends(adapter, barcodes):
    """For every barcode, the adapter beginnings the splitter looks for at the END of a read, each
    with the (negative) index the read is then sliced with: what build_adapter_tree (reference
    tagdigger_fun.py:1208-1249) builds as a reversed-sequence trie, resolved to a flat list.
    Every beginning keeps at least one base beyond the remains of the restriction site."""
    def beginnings(site, tail):
        remains = site.find('^')
        full = site[:remains] + tail
        # longest first, down to one base past the site's remains; reversed, as the trie stores them
        rev = full[::-1]
        cut = [rev[i:] for i in range(len(rev) - remains)]
        return remains, cut, [remains - len(c) for c in cut]

    remains0, common, common_idx = beginnings(adapter[0][0], adapter[0][1])
    out = []
    for bc in barcodes:
        remains1, rare, rare_idx = beginnings(adapter[1][0], adapter[1][1].replace('[barcode]', reverseComplement(bc)))
        everything, indices = common + rare, common_idx + rare_idx
        try:
            kept = _trie_survivors(everything)
        except AssertionError:
            # some beginning is the end of another: keep the shorter of each such (sorted-adjacent) pair
            print("Some overlap of adapter sequence for barcode {}.".format(bc))
            everything = sorted(everything)
            drop = set()
            for k in range(len(everything) - 1):
                if everything[k + 1].startswith(everything[k]):
                    drop.add(k + 1)
                    print("Won't search for {0} at end of sequence since {1} is already being searched for.".format(
                        everything[k + 1][::-1], everything[k][::-1]))
            everything = [x for k, x in enumerate(everything) if k not in drop]
            indices = [remains1 - len(x) for x in everything]          # (the rare cutter's offset for all, as :1246)
            kept = _trie_survivors(everything)
        out.append([(seq[::-1], indices[pos]) for seq, pos in kept])
    return out


def barcodeSplitter(inputFile, barcodes, outputFiles, cutsite='TGCAG', adapter=adapters["PstI-MspI-Hall"],
                    maxreads=500000000, device=0):
    """Split one FASTQ file into one file per barcode, removing the barcode and, on the 3' end,
    anything from the first full restriction site or from an adapter that runs off the read
    (reference tagdigger_fun.py:1286-1368).  The per-read decisions are made on the GPU.
    The progress lines of the reference's loop (:1357-1360) are printed once the file is through."""
    assert set(cutsite) <= set('ACGT'), "Only ACGT cut sites allowed."
    assert all([set(bc) <= set('ACGT') for bc in barcodes]), "Found non-ACGT barcodes."
    assert len(adapter) == 2
    assert all([set(a[0]) <= set('ACGT^') for a in adapter])
    assert set(adapter[0][1]) <= set('ACGT')
    assert set(adapter[1][1]) <= set('[barcode]ACGT')

    print("Building indices for rapid searching...")
    entries = _adapter_ends(adapter, barcodes)
    eng = default_engine(device)
    eng.set_splitter(barcodes, cutsite, adapter[0][0].replace('^', ''), adapter[1][0].replace('^', ''), entries)
    print("Done with indexing setup.")
    print(inputFile)

    # the same failures as the reference's open() calls, in its order (:1318-1327)
    if inputFile[-2:].lower() == 'gz':
        with open(inputFile, 'rb') as fh:
            head = fh.read(2)
        if head and head != b'\x1f\x8b':
            raise _gzip.BadGzipFile("Not a gzipped file (%r)" % head)
    else:
        open(inputFile, 'r').close()
    for name in outputFiles:
        open(name, mode='w').close()
    reads, _, _ = eng.split_file(inputFile, outputFiles, maxreads)
    for line in eng.split_progress_lines(inputFile, reads):
        print(line)
    return None


def sanitizeTags(taglist):
    """Drop every marker one of whose tags is a prefix of (or equal to) another tag, so that the
    tag set handed to find_tags_fastq is prefix-free (reference tagdigger_fun.py:1030-1058).
    Mutates and returns taglist.  Marker membership is a NAME-PREFIX test, as in the reference:
    removing 'TP27' also removes 'TP276...'."""
    assert len(taglist) == 2, "'taglist' should have two elements."
    assert len(taglist[0]) == len(taglist[1]), \
        "List of tag names should be the same as list of tag sequences."
    print("\nSanitizing tags...")
    names, seqs = taglist
    ordered = sorted(seqs)
    for k in range(len(ordered) - 1):
        short = ordered[k]
        if not ordered[k + 1].startswith(short) or short not in seqs:
            continue
        owner = names[seqs.index(short)]
        marker = owner[:owner.find("_")]
        print("Removing " + marker + " for overlap with another marker.")
        for j in sorted((j for j in range(len(seqs)) if names[j].startswith(marker)), reverse=True):
            print(names.pop(j))
            print(seqs.pop(j))
    return taglist


def _is_array(x):
    return type(x).__module__ == "numpy"


def sample_rows(bckeys):
    """Global sample order exactly as combineReadCounts builds it, and per file the row of each barcode."""
    order, slot, rows = [], {}, {}
    for f in sorted(bckeys.keys()):
        rows[f] = []
        for sample in bckeys[f][1]:
            if sample not in slot:
                slot[sample] = len(order)
                order.append(sample)
            rows[f].append(slot[sample])
    return order, rows


def combineReadCounts(countsdict, bckeys):
    """Per-barcode rows of every library -> per-sample rows: files in sorted order, samples in
    order of first appearance, equal names summed (reference tagdigger_fun.py:1061-1098).
    This is also what a multi-GPU run must equal after its all-reduce.
    Matrices given as numpy arrays (Engine.counts_numpy, what the command line uses) are summed with
    numpy and the totals come back as one int64 array; lists give lists, as in the reference."""
    files = sorted(bckeys.keys())
    if files and all(_is_array(countsdict[f]) for f in files):
        import numpy as np
        order, rows = sample_rows(bckeys)
        totals = np.zeros((len(order), countsdict[files[0]].shape[1]), dtype=np.int64)
        for f in files:
            np.add.at(totals, rows[f], countsdict[f].astype(np.int64, copy=False))
        return [order, totals]
    order, totals = [], []
    slot = {}
    for f in files:
        for row, sample in enumerate(bckeys[f][1]):
            if sample in slot:
                k = slot[sample]
                totals[k] = [a + b for a, b in zip(countsdict[f][row], totals[k])]
            else:
                slot[sample] = len(order)
                order.append(sample)
                totals.append(countsdict[f][row])
    return [order, totals]


def _csv_cell(text):
    """One field exactly as csv.writer (default dialect, minimal quoting) writes it as the FIRST of several fields
    (a row of one empty field is written as '""', an empty first field of a longer row as nothing)."""
    import io
    buf = io.StringIO()
    _csv.writer(buf).writerow([text, "x"])
    return buf.getvalue()[:-4]                      # (without ',x' and the row's \r\n)


def writeCounts(filename, counts, samnames, tagnames):
    """Samples x tags CSV, csv.writer defaults (CRLF rows) (reference tagdigger_fun.py:1100-1111).
    A numpy matrix is written row by row with ndarray.tofile (decimal integers, the same bytes)."""
    assert len(samnames) == len(counts), "Length of samnames should be the same as length of counts."
    assert len(tagnames) == len(counts[0]), "Length of tagnames should be length of second dimension of counts."
    if _is_array(counts):
        import io
        import locale
        import numpy as np
        enc = locale.getpreferredencoding(False)            # (what open(..., 'w') would encode with)
        head = io.StringIO()
        _csv.writer(head).writerow([""] + tagnames)
        rows = np.ascontiguousarray(counts, dtype=np.int64)
        import ctypes
        from . import _binding
        fmt = _binding.load().td_format_csv_row                # (integer rows formatted by the library: 200 M cells/s)
        buf = ctypes.create_string_buffer(24 * max(1, rows.shape[1]))
        with open(filename, mode='wb') as fb:
            fb.write(head.getvalue().encode(enc))
            for name, row in zip(samnames, rows):
                n = fmt(row.ctypes.data, rows.shape[1], buf, len(buf))
                if n < 0:
                    raise RuntimeError("td_format_csv_row: buffer too small")
                fb.write((_csv_cell(name) + ",").encode(enc))
                fb.write(memoryview(buf)[:n])
                fb.write(b"\r\n")
        return
    with open(filename, mode='w', newline='') as fh:
        out = _csv.writer(fh)
        out.writerow([""] + tagnames)
        for name, row in zip(samnames, counts):
            out.writerow([name] + row)


def extractMarkers(tagnames):
    """[marker names in first-seen order, per marker [[allele names], [tag indices]]]
    (reference tagdigger_fun.py:1113-1142)."""
    if len(tagnames) != len(set(tagnames)):
        raise Exception("Non-unique tag names found.")
    markers, alleles, where = [], [], {}
    for k, t in enumerate(tagnames):
        m = t[:t.find('_')]
        if m not in where:
            where[m] = len(markers)
            markers.append(m)
            alleles.append([[], []])
        alleles[where[m]][0].append(t[t.rfind('_') + 1:])
        alleles[where[m]][1].append(k)
    return [markers, alleles]


def writeDiploidGeno(filename, counts, samnames, tagnames):
    """0 / 1 / 2 / blank genotype calls from the allele-0 and allele-1 counts of every marker
    (reference tagdigger_fun.py:1144-1180)."""
    assert len(samnames) == len(counts), "Length of samnames should be the same as length of counts."
    assert len(tagnames) == len(counts[0]), "Length of tagnames should be length of second dimension of counts."
    markers, alleles = extractMarkers(tagnames)
    try:
        if not all(set(a[0]) <= {'0', '1'} for a in alleles):
            raise Exception("All allele names must be '0' or '1'.")
        if _is_array(counts):
            import numpy as np
            i0 = np.array([a[1][a[0].index('0')] for a in alleles], dtype=np.int64)
            i1 = np.array([a[1][a[0].index('1')] for a in alleles], dtype=np.int64)
            has0, has1 = counts[:, i0] > 0, counts[:, i1] > 0
            table = np.array(['', '0', '2', '1'])                  # neither, allele 0 only, allele 1 only, both
            rows = table[has0 + 2 * has1].tolist()
        else:
            rows = []
            for s in range(len(samnames)):
                calls = []
                for a in alleles:
                    c0 = counts[s][a[1][a[0].index('0')]]
                    c1 = counts[s][a[1][a[0].index('1')]]
                    calls.append('1' if c0 > 0 and c1 > 0 else '0' if c0 > 0 else '2' if c1 > 0 else '')
                rows.append(calls)
        with open(filename, mode='w', newline='') as fh:
            out = _csv.writer(fh)
            out.writerow([""] + markers)
            for name, calls in zip(samnames, rows):
                out.writerow([name] + calls)
    except IOError:
        print("Could not write file {}.".format(filename))
    except Exception as err:
        print(err.args[0])
    return None
