#!/usr/bin/env python3
"""HBM-side bytes per launch per kernel family from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE collected separately):
   python scripts/pmc_traffic.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run> [units] [note] > profiles/rNN_pmc_....json
`units` (e.g. the number of decoded token-steps of the profiled command) adds whole-run bytes per unit.
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; FETCH_SIZE counts the
128-byte requests of 16-byte-per-lane loads at 64 bytes, hence x2."""
import collections, csv, glob, json, sys


def family(n):
    for key, fam in (("gemm_nt", "gemm_nt"), ("gemm_strip", "gemm_nt"), ("gemm_ws", "gemm_nt"), ("gemm_tn", "gemm_tn"), ("dec_gemm", "dec_gemm"), ("attn_decode", "attn_decode"), ("attn_cross_mfma", "attn_decode"), ("pack_cross_kv", "cross_kv_packing"), ("gemm_skinny", "gemm_skinny"),
                     ("gemm_fp8", "gemm_fp8"), ("beam_", "beam_step"), ("gather_multi", "cache_reorder"), ("sample_", "token_selection"),
                     ("select_token", "token_selection"), ("decode_step", "step_inputs_embedding"), ("dw3_", "conv_projections"), ("layernorm", "layernorm")):
        if key in n:
            return fam
    return "attention" if n.startswith("attn_") or "_attn_" in n or "attn_fwd" in n or "attn_bwd" in n else "other"


def load(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[family(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
units = float(sys.argv[3]) if len(sys.argv) > 3 else None
out = {}
tot = 0.0
for fam in fe:
    n = fe[fam][0]
    fb, wb = fe[fam][1] * 1024 * 2 / n, wr[fam][1] * 1024 / max(1, wr[fam][0])
    out[fam] = {"launches": n, "fetch_bytes_per_launch_corrected_x2": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    tot += (fb + wb) * n
    if units:
        out[fam]["launches_per_unit"] = n / units
        out[fam]["hbm_bytes_per_unit"] = (fb + wb) * n / units
if units:
    out["_all_kernels"] = {"units": units, "hbm_bytes_per_unit": tot / units}
out["_provenance"] = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a second run, --pmc WRITE_SIZE) --output-format csv -- " + (sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py ..."),
    "unit": "bytes per launch; FETCH_SIZE (KiB) x 1024 x 2 (gfx950: the counter tallies 128-B requests at 64 B for 16-B-per-lane loads, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE (KiB) x 1024; Infinity-Cache hits are included (memory-side fabric requests)"}
print(json.dumps(out, indent=1))
