import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for th in (8, 16, 32, 64, 128):
    r = bench.cpu_baseline(256, 30000, budget_s=12.0, threads=th)
    print(th, round(r["value"], 1), r["sample"][:12], flush=True)
