#!/usr/bin/env python3
"""Per-(kernel, grid) HBM-side traffic table from two rocprofv3 counter passes: python scripts/pmc_kernel_table.py <FETCH dir> <WRITE dir> [steps]
(FETCH_SIZE x 1024 x 2 and WRITE_SIZE x 1024 bytes per launch, as scripts/pmc_traffic.py; `steps` = optimiser steps in the run -> launches per step)"""
import collections, csv, glob, sys


def load(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:44] or "(anonymous: dw3_* / dec_* kernels)"
        a = acc[(n, r.get("Grid_Size", ""))]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for k in fe:
    n = fe[k][0]; fb = fe[k][1] * 2048 / n; wb = wr[k][1] * 1024 / max(1, wr[k][0]) if k in wr else 0.0
    rows.append((n * (fb + wb), k, n, fb, wb))
rows.sort(reverse=True)
print(f"{'kernel':46s} {'grid thr':>9s} {'launch/step':>11s} {'fetch MB':>9s} {'write MB':>9s} {'GB/step':>8s}")
for tot, k, n, fb, wb in rows[:60]:
    print(f"{k[0]:46s} {k[1]:>9s} {n / steps:11.1f} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {tot / steps / 1e9:8.2f}")
print(f"all kernels: {sum(r[0] for r in rows) / steps / 1e9:.1f} GB per step")
