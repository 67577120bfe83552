#!/usr/bin/env python3
"""Condense a tools/profile_round.sh run into profiles/<tag>/ (bench.py measures roofline.traffic itself since round 3:
its own rocprofv3 --pmc child runs; this script is for the per-kernel table of a manual profile).

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch. On gfx950 FETCH_SIZE tallies the 128-B
requests of a wide coalesced stream at 64 B (MI355X_MICROARCH.md, "HBM"), so the read side is
doubled; WRITE_SIZE is taken as is.

  usage: tools/pmc_summary.py <tag>
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def per_kernel(path, counter):
    """-> {kernel name: [values per dispatch]}"""
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    stats = one(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
    fetch = one(os.path.join(src, "fetch", "**", "*counter_collection.csv"))
    write = one(os.path.join(src, "write", "**", "*counter_collection.csv"))
    if stats:
        shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
    log = os.path.join(src, "bench_stats.log")
    if os.path.exists(log):
        shutil.copy(log, os.path.join(dst, "bench_line.log"))
    summary = {"tag": tag, "kernels": {}}
    fk = per_kernel(fetch, "FETCH_SIZE") if fetch else {}
    wk = per_kernel(write, "WRITE_SIZE") if write else {}
    rows = []
    for name in sorted(set(fk) | set(wk)):
        if "tdk::" not in name:
            continue
        f = fk.get(name, [])
        w = wk.get(name, [])
        # k_fast is launched twice per pass (the second launch drains the fix-up queue, normally
        # empty): keep the full-size dispatches only
        f = [v for v in f if v >= 0.1 * max(f)] if f else f
        w = [v for v in w if v >= 0.1 * max(w)] if w else w
        favg = sum(f) / len(f) * 1024 if f else 0.0
        wavg = sum(w) / len(w) * 1024 if w else 0.0
        summary["kernels"][name] = {"dispatches": max(len(f), len(w)), "FETCH_SIZE_bytes_raw": favg,
                                    "fetch_bytes_corrected_x2": 2 * favg, "WRITE_SIZE_bytes": wavg}
        rows.append((name, len(f), favg, wavg))
    # the dominant kernel = the one with the most fetched bytes
    if rows:
        dom = max(rows, key=lambda r: r[2])
        hbm = 2 * dom[2] + dom[3]
        summary["dominant_kernel"] = dom[0]
        summary["hbm_bytes_per_launch"] = hbm
        summary["fabric_bytes_per_launch"] = hbm
        summary["note"] = ("fabric_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), mean over the full-size "
                           "dispatches of the dominant kernel: requests at the L2's memory side, Infinity-Cache hits included "
                           "(an upper bound of the HBM bytes); the calibration of FETCH_SIZE on this access pattern is "
                           "profiles/r01_final/pmc_calibration.txt")
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    with open(os.path.join(dst, "pmc_per_kernel.csv"), "w") as f:
        f.write("kernel,dispatches,FETCH_SIZE_bytes_raw,fetch_bytes_x2,WRITE_SIZE_bytes\n")
        for name, n, fa, wa in rows:
            f.write('"%s",%d,%.0f,%.0f,%.0f\n' % (name, n, fa, 2 * fa, wa))
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
