#!/usr/bin/env python3
"""The reference callers' call sequences on the drop-in classes (bench.py tf_dropin / scst_dropin) for rocprofv3: python3 scripts/dropin_profile.py tf|scst"""
import os, sys, types, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = types.SimpleNamespace(batch=32, seq_len=256, eval_mode=False, new_tokens=255)
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "tf"
r = bench.tf_dropin(args, dev, 2, steps=4) if which == "tf" else bench.scst_dropin(args, dev, steps=2)
print(json.dumps({k: v for k, v in r.items() if k != "what"}))
