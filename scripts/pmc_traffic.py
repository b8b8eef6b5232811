#!/usr/bin/env python3
"""HBM-side bytes per launch per kernel family from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE collected separately):
   python scripts/pmc_traffic.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run> > profiles/r01_pmc_hbm_traffic.json
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; FETCH_SIZE counts the
128-byte requests of 16-byte-per-lane loads at 64 bytes, hence x2."""
import collections, csv, glob, json, sys


def load(d, counter):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"))[-1]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        fam = ("gemm_nt" if "gemm_nt" in n else "gemm_tn" if "gemm_tn" in n else "attention" if n.startswith("attn_") else
               "conv_projections" if "dw3_" in n else "layernorm" if "layernorm" in n else "other")
        a = acc[fam]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in fe:
    n = fe[fam][0]
    fb, wb = fe[fam][1] * 1024 * 2 / n, wr[fam][1] * 1024 / max(1, wr[fam][0])
    out[fam] = {"launches": n, "fetch_bytes_per_launch_corrected_x2": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
out["_provenance"] = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a second run, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-scst",
    "unit": "bytes per launch; FETCH_SIZE (KiB) x 1024 x 2 (gfx950: the counter tallies 128-B requests at 64 B for 16-B-per-lane loads, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE (KiB) x 1024; Infinity-Cache hits are included (memory-side fabric requests)",
    "round": 1, "mode": "train", "kernels": "end of round 1 (fused projection passes, fused decoder q/k/v and cross-K/V GEMMs, bf16 logits)"}
print(json.dumps(out, indent=1))
