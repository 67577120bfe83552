#!/usr/bin/env python3
"""Command-line front end with the reference's flags (tagdigger_script.py:10-35), rules
(:38-120) and outputs (:128-133); the per-file counting (:123-126) runs on the GPU.

    python -m tagdigger_amd.tagdigger_script -e PstI --MergedTags tags.csv -b key.csv -o counts.csv [-g geno.csv]

Extra flags of this build carry a --td- prefix and never change the results.
"""
import argparse
import os
import sys

from . import tagdigger_fun


def build_parser():
    ap = argparse.ArgumentParser(description="TagDigger v. 1.1 command line script by Lindsay V. Clark "
                                             "(MI355X counting engine)")
    ap.add_argument('-e', '--enzyme', help='Restriction enzyme name', choices=sorted(tagdigger_fun.enzymes.keys()))
    ap.add_argument('-c', '--cutsite', help='Restriction cut site sequence expected in sequencing reads')
    ap.add_argument('-w', '--directory', help='Working directory')
    ap.add_argument('--UNEAKtags', help='File name for tags in UNEAK format')
    ap.add_argument('--MergedTags', help='File name for tags in merged format')
    ap.add_argument('--ColumnTags', help='File name for tags in column format')
    ap.add_argument('--RowTags', help='File name for tags in row format.')
    ap.add_argument('--StacksTags', help='File name for Stacks tags.tsv file.')
    ap.add_argument('--StacksSnps', help='File name for Stacks snps.tsv file.')
    ap.add_argument('--StacksAlleles', help='File name for Stacks alleles.tsv file.')
    ap.add_argument('--TASSELSAM', help='File name for TASSEL SAM file')
    ap.add_argument('--pyRADalleles', help='File name for pyRAD .alleles file.')
    ap.add_argument('-k', '--tokeep', help='File name listing tags to keep')
    ap.add_argument('--binaryOnly', help="'T' to retain only binary markers; 'F' to retain all markers.",
                    default='F', choices=['T', 'F'])
    ap.add_argument('--TASSELkeyFile', help='File name to output for key to TASSEL SNP names')
    ap.add_argument('-b', '--barcodefile', help='Name of barcode key file', required=True)
    ap.add_argument('-o', '--outputcounts', help='Output file name for read counts', required=True)
    ap.add_argument('-g', '--outputgen', help='Output file name for numeric genotypes')
    ap.add_argument('--td-device', type=int, default=0, help='GPU to count on (this build only)')
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)

    # restriction cut site (reference tagdigger_script.py:38-49)
    if args.enzyme == None and args.cutsite == None:
        raise Exception("Need either restriction enzyme name or cutsite sequence.  Use '-e None' if no restriction site is present in reads.")
    if args.enzyme != None and args.cutsite != None:
        cutsite = args.cutsite.upper()
        if cutsite != tagdigger_fun.enzymes[args.enzyme]:
            raise Exception("Restriction enzyme name and cutsite do not match.  Note that only one of these two arguments is required.")
    elif args.enzyme != None:
        cutsite = tagdigger_fun.enzymes[args.enzyme]
    else:
        cutsite = args.cutsite.upper()
        if not set(cutsite) <= set('ACGTRYSWKMBDHVN'):
            raise Exception("Cut site contains unexpected characters.")

    if args.directory != None:
        if not os.path.isdir(args.directory):
            raise Exception("Directory {} not found".format(args.directory))
        os.chdir(args.directory)

    # exactly one tag format (:58-69)
    given = [args.UNEAKtags != None, args.MergedTags != None, args.ColumnTags != None, args.RowTags != None,
             args.StacksTags != None, args.StacksSnps != None, args.StacksAlleles != None,
             args.TASSELSAM != None, args.pyRADalleles != None]
    if given[4] or given[5] or given[6]:
        if not (given[4] and given[5] and given[6]):
            raise Exception("Need all three files for Stacks format.")
    del given[5:7]
    if sum(given) != 1:
        raise Exception('Exactly one tag format required.')
    toKeep = None
    if args.tokeep != None:
        toKeep = tagdigger_fun.readMarkerNames(args.tokeep)
        if toKeep == None:
            raise Exception("Problem reading marker names to keep.")

    binaryOnly = args.binaryOnly == 'T'
    if given[0]:
        tags = tagdigger_fun.readTags_UNEAK_FASTA(args.UNEAKtags, toKeep=toKeep)
    elif given[1]:
        tags = tagdigger_fun.readTags_Merged(args.MergedTags, toKeep=toKeep)
    elif given[2]:
        tags = tagdigger_fun.readTags_Columns(args.ColumnTags, toKeep=toKeep)
    elif given[3]:
        tags = tagdigger_fun.readTags_Rows(args.RowTags, toKeep=toKeep)
    elif given[4]:
        tags = tagdigger_fun.readTags_Stacks(args.StacksTags, args.StacksSnps, args.StacksAlleles,
                                             toKeep=toKeep, binaryOnly=binaryOnly)
    elif given[5]:
        tags = tagdigger_fun.readTags_TASSELSAM(args.TASSELSAM, toKeep=toKeep, binaryOnly=binaryOnly,
                                                writeMarkerKey=args.TASSELkeyFile != None,
                                                keyfilename=args.TASSELkeyFile)
    else:
        tags = tagdigger_fun.readTags_pyRAD(args.pyRADalleles, toKeep=toKeep, binaryOnly=binaryOnly)
    if tags == None:
        raise Exception("Problem reading tags.")
    tags = tagdigger_fun.sanitizeTags(tags)

    bckeys = tagdigger_fun.readBarcodeKeyfile(args.barcodefile)
    if bckeys == None:
        raise Exception("Problem reading barcode file.")
    fqfiles = sorted(bckeys.keys())
    fqok = [tagdigger_fun.isFastq(f) for f in fqfiles]
    if not all(fqok):
        print("Cannot read the following as FASTQ files:")
        print([fqfiles[i] for i in range(len(fqfiles)) if not fqok[i]])
        raise Exception("Cannot read all FASTQ files.")

    if args.outputgen != None:
        if set([t[-1] for t in tags[0]]) != {'0', '1'}:
            raise Exception("Cannot output numeric genotypes for non-binary markers.")

    countsdict = dict()
    for f in fqfiles:                                   # the hot path (:123-126)
        countsdict[f] = tagdigger_fun.find_tags_fastq(f, bckeys[f][0], tags[1], cutsite=cutsite,
                                                      device=args.td_device)
    combres = tagdigger_fun.combineReadCounts(countsdict, bckeys)

    tagdigger_fun.writeCounts(args.outputcounts, combres[1], combres[0], tags[0])
    if args.outputgen != None:
        tagdigger_fun.writeDiploidGeno(args.outputgen, combres[1], combres[0], tags[0])
    return 0


if __name__ == "__main__":
    sys.exit(main())
