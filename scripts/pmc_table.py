"""Per-kernel table of a rocprofv3 --pmc counter_collection.csv: python scripts/pmc_table.py <dir> [kernel substring]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("(")[0][:44], r.get("Grid_Size", ""))
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
names = sorted({n for k in acc for n in acc[k]})
print(f"{'kernel':46s} {'grid':>9s} " + " ".join(f"{n[3:][:16]:>16s}" for n in names))
for k in acc:
    print(f"{k[0]:46s} {k[1]:>9s} " + " ".join(f"{acc[k][n] / max(cnt[k][n], 1):16.0f}" for n in names))
    if "SQ_WAVE_CYCLES" in acc[k]:
        wc = acc[k]["SQ_WAVE_CYCLES"] / cnt[k]["SQ_WAVE_CYCLES"]
        print(f"{'   fraction of WAVE_CYCLES':46s} {'':>9s} " + " ".join(f"{acc[k][n] / max(cnt[k][n], 1) / wc:16.3f}" for n in names))
