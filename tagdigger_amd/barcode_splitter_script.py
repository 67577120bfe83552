#!/usr/bin/env python3
"""Splitting from the command line: every FASTQ file of a key file ('Input File', 'Barcode',
'Output File' columns) into one clipped FASTQ per barcode, the per-read decisions made on the GPU
(tagdigger_fun.barcodeSplitter).  Takes the two options of the reference's barcode_splitter_script.py.

    python -m tagdigger_amd.barcode_splitter_script -b splitkey.csv -a PstI-MspI-Hall
"""
import argparse
import sys

from . import tagdigger_fun as tf


def build_parser():
    ap = argparse.ArgumentParser(description="Split FASTQ files by barcode and trim adapter read-through on an MI355X")
    ap.add_argument("-b", "--barcodefile", required=True, metavar="FILE", help="key file: input file, barcode, output file")
    ap.add_argument("-a", "--adapter", required=True, choices=sorted(tf.adapters),
                    help="enzyme pair and adapter design (its first word names the enzyme at the barcode end)")
    ap.add_argument("--td-device", type=int, default=0, help="GPU to use (this build only)")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    keys = tf.readBarcodeKeyfile(args.barcodefile, forSplitter=True)
    if keys is None:
        raise Exception("Problem reading barcode file.")
    enzyme = args.adapter.split("-", 1)[0]
    inputs = sorted(keys)
    unreadable = [f for f in inputs if not tf.isFastq(f)]
    if unreadable:
        print("Cannot read the following as FASTQ files:")
        print(unreadable)
        raise Exception("Cannot read all FASTQ files.")
    for f in inputs:
        barcodes, outputs = keys[f][0], keys[f][1]
        tf.barcodeSplitter(f, barcodes, outputs, cutsite=tf.enzymes[enzyme], adapter=tf.adapters[args.adapter],
                           device=args.td_device)
    return 0


if __name__ == "__main__":
    sys.exit(main())
