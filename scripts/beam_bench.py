"""Beam-generation measurement alone (bench.py's `beam_generation` key): python scripts/beam_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.beam_bench(None, torch.device("cuda:0"), host_loop_too="--no-host" not in sys.argv)))
