import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
acc = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
for r in rows:
    if r["Counter_Name"] != sys.argv[2] or "gemm_nt" not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"][:40], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(k, n, "KiB/launch", round(v / n, 1), " x2 MB:", round(v / n * 2 * 1024 / 1e6, 2))
