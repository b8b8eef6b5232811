"""The configs[4]-shaped SCST measurement alone (bench.py's `scst_c5` key): python scripts/scst_c5_bench.py [steps]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    args = argparse.Namespace(eval_mode=False, warmup=2, new_tokens=255)
    torch.cuda.set_device(0)
    print(json.dumps(bench.scst_bench(args, 0, 1, torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 3, c5=True)))
