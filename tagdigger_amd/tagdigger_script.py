#!/usr/bin/env python3
"""Counting from the command line, with the flags of the reference's tagdigger_script.py (:10-35) so
that existing invocations keep working; its rules (:38-120) are checked by small functions here, the
tag readers are looked up in a table, and the per-file counting (:123-126) runs on the GPU.

    python -m tagdigger_amd.tagdigger_script -e PstI --MergedTags tags.csv -b key.csv -o counts.csv [-g geno.csv]

Flags that exist only in this build carry a --td- prefix and never change the results.
"""
import argparse
import os
import sys

from . import tagdigger_fun as tf

# tag formats: name -> (the options that must all be given, how the table is read)
TAG_FORMATS = {
    "UNEAK": (("UNEAKtags",), lambda a, keep, binary: tf.readTags_UNEAK_FASTA(a.UNEAKtags, toKeep=keep)),
    "merged": (("MergedTags",), lambda a, keep, binary: tf.readTags_Merged(a.MergedTags, toKeep=keep)),
    "columns": (("ColumnTags",), lambda a, keep, binary: tf.readTags_Columns(a.ColumnTags, toKeep=keep)),
    "rows": (("RowTags",), lambda a, keep, binary: tf.readTags_Rows(a.RowTags, toKeep=keep)),
    "Stacks": (("StacksTags", "StacksSnps", "StacksAlleles"),
               lambda a, keep, binary: tf.readTags_Stacks(a.StacksTags, a.StacksSnps, a.StacksAlleles, toKeep=keep,
                                                          binaryOnly=binary)),
    "TASSEL": (("TASSELSAM",),
               lambda a, keep, binary: tf.readTags_TASSELSAM(a.TASSELSAM, toKeep=keep, binaryOnly=binary,
                                                             writeMarkerKey=a.TASSELkeyFile is not None,
                                                             keyfilename=a.TASSELkeyFile)),
    "pyRAD": (("pyRADalleles",), lambda a, keep, binary: tf.readTags_pyRAD(a.pyRADalleles, toKeep=keep, binaryOnly=binary)),
}
FILE_OPTIONS = {                       # option -> what the file holds
    "UNEAKtags": "UNEAK FASTA", "MergedTags": "merged tag table", "ColumnTags": "two-column tag table",
    "RowTags": "one-tag-per-row table", "StacksTags": "Stacks tags.tsv", "StacksSnps": "Stacks snps.tsv",
    "StacksAlleles": "Stacks alleles.tsv", "TASSELSAM": "TASSEL SAM file", "pyRADalleles": "pyRAD .alleles file",
}
IUPAC = frozenset("ACGTRYSWKMBDHVN")


def build_parser():
    ap = argparse.ArgumentParser(description="Count known tags per barcode in FASTQ files on an MI355X "
                                             "(same options as TagDigger 1.1's tagdigger_script.py)")
    site = ap.add_argument_group("restriction site (one of the two)")
    site.add_argument("-e", "--enzyme", choices=sorted(tf.enzymes), help="enzyme whose remnant follows the barcode")
    site.add_argument("-c", "--cutsite", help="that remnant spelled out (IUPAC codes allowed)")
    ap.add_argument("-w", "--directory", help="change to this directory first")
    fmt = ap.add_argument_group("tags (exactly one format)")
    for opt, what in FILE_OPTIONS.items():
        fmt.add_argument("--" + opt, metavar="FILE", help=what)
    ap.add_argument("-k", "--tokeep", metavar="FILE", help="marker names to keep, one per line")
    ap.add_argument("--binaryOnly", choices=["T", "F"], default="F", help="T: drop markers with more than two alleles")
    ap.add_argument("--TASSELkeyFile", metavar="FILE", help="write the key to the TASSEL marker names here")
    ap.add_argument("-b", "--barcodefile", required=True, metavar="FILE", help="key file: FASTQ file, barcode, sample")
    ap.add_argument("-o", "--outputcounts", required=True, metavar="FILE", help="samples x tags CSV to write")
    ap.add_argument("-g", "--outputgen", metavar="FILE", help="0/1/2 genotype CSV to write (binary markers only)")
    ap.add_argument("--td-device", type=int, default=0, help="GPU to count on (this build only)")
    ap.add_argument("--td-devices", metavar="LIST", help="several GPUs, e.g. 0,1,2,3: one process per GPU, the key file's "
                    "libraries dealt over them, barcode rows folded into sample rows on the device and ONE all-reduce of the "
                    "samples x tags matrix over RCCL (this build only; same output files)")
    ap.add_argument("--td-timing", action="store_true", help="print where the wall time went, per stage (this build only)")
    return ap


def cut_site(args):
    """What follows the barcode in a read (reference :38-49)."""
    named = tf.enzymes[args.enzyme] if args.enzyme is not None else None
    spelled = args.cutsite.upper() if args.cutsite is not None else None
    if named is None and spelled is None:
        raise Exception("Need either restriction enzyme name or cutsite sequence.  Use '-e None' if no restriction site "
                        "is present in reads.")
    if named is not None and spelled is not None and named != spelled:
        raise Exception("Restriction enzyme name and cutsite do not match.  Note that only one of these two arguments "
                        "is required.")
    if named is None and not set(spelled) <= IUPAC:
        raise Exception("Cut site contains unexpected characters.")
    return named if named is not None else spelled


def tag_format(args):
    """The one format whose options were given (reference :58-69)."""
    chosen = []
    for name, (opts, _) in TAG_FORMATS.items():
        have = [getattr(args, o) is not None for o in opts]
        if any(have) and not all(have):
            raise Exception("Need all three files for Stacks format.")
        if all(have):
            chosen.append(name)
    if len(chosen) != 1:
        raise Exception("Exactly one tag format required.")
    return chosen[0]


def checked(value, what):
    """The readers print their complaint and return None (reference :362-374); the script then stops."""
    if value is None:
        raise Exception("Problem reading {}.".format(what))
    return value


def main(argv=None):
    args = build_parser().parse_args(argv)
    clock = _Clock(args.td_timing)
    site = cut_site(args)
    if args.directory is not None:
        if not os.path.isdir(args.directory):
            raise Exception("Directory {} not found".format(args.directory))
        os.chdir(args.directory)
    fmt = tag_format(args)
    keep = checked(tf.readMarkerNames(args.tokeep), "marker names to keep") if args.tokeep is not None else None
    tags = checked(TAG_FORMATS[fmt][1](args, keep, args.binaryOnly == "T"), "tags")
    clock.lap("tag reader")
    names, sequences = tf.sanitizeTags(tags)
    clock.lap("sanitizeTags")
    keys = checked(tf.readBarcodeKeyfile(args.barcodefile), "barcode file")
    libraries = sorted(keys)
    unreadable = [f for f in libraries if not tf.isFastq(f)]
    if unreadable:
        print("Cannot read the following as FASTQ files:")
        print(unreadable)
        raise Exception("Cannot read all FASTQ files.")
    if args.outputgen is not None and {n[-1] for n in names} != {"0", "1"}:
        raise Exception("Cannot output numeric genotypes for non-binary markers.")
    clock.lap("checks")
    devices = [int(d) for d in args.td_devices.split(",")] if args.td_devices else [args.td_device]
    if len(devices) > 1:
        # one process per GPU (reference loop :123-126 dealt over the ranks); rank 0 writes the files
        import socket
        import torch.multiprocessing as mp
        if "MASTER_PORT" not in os.environ:             # (a free port for the ranks' rendezvous on this host)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        mp.spawn(_rank_main, args=(devices, keys, sequences, site, names, args.outputcounts, args.outputgen),
                 nprocs=len(devices), join=True)
        clock.lap("count on %d GPUs + combine + write" % len(devices))
        clock.report()
        return 0
    # the hot path; matrices stay numpy arrays from the device to the CSV writer (no Python lists of ints)
    per_file = {}
    for f in libraries:
        per_file[f] = tf.find_tags_fastq(f, keys[f][0], sequences, cutsite=site, device=devices[0], as_array=True)
    clock.lap("find_tags_fastq (index build + count + counts to host)")
    samples, counts = tf.combineReadCounts(per_file, keys)
    clock.lap("combineReadCounts")
    tf.writeCounts(args.outputcounts, counts, samples, names)
    clock.lap("writeCounts")
    if args.outputgen is not None:
        tf.writeDiploidGeno(args.outputgen, counts, samples, names)
        clock.lap("writeDiploidGeno")
    clock.report()
    return 0


class _Clock:
    """--td-timing: wall time per stage of the command (reference stages: readers, sanitizeTags, key file, the
    per-file find_tags_fastq loop, combineReadCounts, the writers)."""
    def __init__(self, on):
        import time
        self.on, self.now, self.t, self.laps = on, time.perf_counter, time.perf_counter(), []

    def lap(self, what):
        t = self.now()
        self.laps.append((what, t - self.t))
        self.t = t

    def report(self):
        if self.on:
            for what, dt in self.laps:
                print("[td-timing] %-58s %8.3f s" % (what, dt), file=sys.stderr)
            print("[td-timing] %-58s %8.3f s" % ("total", sum(dt for _, dt in self.laps)), file=sys.stderr)


def _rank_main(rank, devices, keys, sequences, site, names, outputcounts, outputgen):
    """One rank of a --td-devices run: RCCL ("nccl") over distinct GPUs; the same GPU listed more than once is a
    rehearsal of the code path over gloo."""
    import torch
    import torch.distributed as dist
    from . import multi
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", devices[rank])
    torch.cuda.set_device(dev)
    distinct = len(set(devices)) == len(devices)
    if distinct:
        dist.init_process_group("nccl", rank=rank, world_size=len(devices), device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=len(devices))
    try:
        samples, counts = multi.count_libraries(keys, sequences, site, device=dev, as_array=True, progress=True)
        if rank == 0:
            tf.writeCounts(outputcounts, counts, samples, names)
            if outputgen is not None:
                tf.writeDiploidGeno(outputgen, counts, samples, names)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
