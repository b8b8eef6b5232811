#!/usr/bin/env python3
"""2-process DP step on one GPU over gloo, with tracebacks (debug aid for tests/test_dp_gpu.py)."""
import os, sys, tempfile, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

def run(rank, world, path):
    try:
        import test_dp_gpu
        import torch.multiprocessing as mp
        class Q:
            def put(self, x): print("rank", x[0], "loss", x[1], flush=True)
        test_dp_gpu._worker(rank, world, path, Q())
    except Exception:
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)

if __name__ == "__main__":
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(run, args=(2, os.path.join(d, "rdzv")), nprocs=2, join=True)
