#!/usr/bin/env python3
"""Command-line mirror of the reference's barcode_splitter_script.py: split every FASTQ file named
in a key file ('Input File', 'Barcode', 'Output File' columns) into one clipped FASTQ per barcode.
Same arguments; the per-read decisions are made on the GPU (tagdigger_fun.barcodeSplitter)."""
import argparse
import sys

from . import tagdigger_fun


def build_parser():
    ap = argparse.ArgumentParser(description="TagDigger v. 1.1 barcode splitter command line script by Lindsay V. Clark "
                                             "(MI355X engine)")
    ap.add_argument('-b', '--barcodefile', help='Name of barcode key file', required=True)
    ap.add_argument('-a', '--adapter', help='Name of the adapter set', required=True,
                    choices=sorted(tagdigger_fun.adapters.keys()))
    ap.add_argument('--td-device', type=int, default=0, help='GPU to use (this build only)')
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    bckeys = tagdigger_fun.readBarcodeKeyfile(args.barcodefile, forSplitter=True)
    if bckeys == None:
        raise Exception("Problem reading barcode file.")
    adapter = tagdigger_fun.adapters[args.adapter]
    cutsite = tagdigger_fun.enzymes[args.adapter[:args.adapter.find("-")]]     # the set's name starts with the enzyme
    fqfiles = sorted(bckeys.keys())
    fqok = [tagdigger_fun.isFastq(f) for f in fqfiles]
    if not all(fqok):
        print("Cannot read the following as FASTQ files:")
        print([fqfiles[i] for i in range(len(fqfiles)) if not fqok[i]])
        raise Exception("Cannot read all FASTQ files.")
    for f in fqfiles:
        tagdigger_fun.barcodeSplitter(f, bckeys[f][0], bckeys[f][1], cutsite=cutsite, adapter=adapter,
                                      device=args.td_device)
    return 0


if __name__ == "__main__":
    sys.exit(main())
