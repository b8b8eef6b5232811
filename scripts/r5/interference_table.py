#!/usr/bin/env python3
"""Where the weight-gradient stream costs the main stream its time: per-kernel average duration of the TF step WITH the weight-gradient GEMMs on the
side stream against the same step WITHOUT them (CXR_WGRAD_SKIP=1, timing experiment), from two rocprofv3 kernel-trace summaries of the same command
on the same box.   usage: interference_table.py with.csv skip.csv steps"""
import csv, sys

def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}

w, s, steps = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
SIDE = ("gemm_tn", "FillFunctor")          # weight-gradient kernels that also run without linear_bwd_weight (patch-embedding gradient) / allocator fills
for name, (n, t) in w.items():
    if name in s and s[name][0] > 0 and not any(k in name for k in SIDE):
        n2, t2 = s[name]
        rows.append(((t / n - t2 / n2) * n / steps / 1e6, name, n / steps, t / n / 1e3, t2 / n2 / 1e3))
only = [(t / steps / 1e6, name, n / steps) for name, (n, t) in w.items() if name not in s]
rows.sort(reverse=True)
short = lambda k: k.replace(".kd", "")[:64]
print(f"{'kernel (both runs)':66s} {'calls/step':>10s} {'us with':>9s} {'us without':>10s} {'slowdown':>9s} {'ms/step lost':>12s}")
tot = 0.0
for d, name, n, a, b in rows:
    tot += d
    if abs(d) >= 0.02:
        print(f"{short(name):66s} {n:10.1f} {a:9.1f} {b:10.1f} {a / b:9.2f} {d:12.3f}")
print(f"sum over the kernels both runs launch: {tot:.2f} ms per step of main-stream kernel time lost beside the weight-gradient stream")
print("kernels only the run WITH weight gradients launches (side stream):")
for t, name, n in sorted(only, reverse=True)[:8]:
    print(f"  {short(name):64s} {n:10.1f} calls/step {t:8.2f} ms/step")
