"""Per-step kernel table from a rocprofv3 --stats CSV: python scripts/kstats.py gpurun_out/profX [steps_incl_warmup]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 11
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:64]:64s} calls/step {int(r['Calls'])/n:7.1f}  ms/step {int(r['TotalDurationNs'])/n/1e6:7.3f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
print("total ms/step", sum(int(r['TotalDurationNs']) for r in rows) / n / 1e6)
