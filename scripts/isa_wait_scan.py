#!/usr/bin/env python3
"""Per kernel of one .hip source: global loads, s_waitcnt vmcnt(0) and exec-mask branches INSIDE loops (the signature of guarded loads that hipcc
serialises): python scripts/isa_wait_scan.py cxrmate_amd/csrc/gemm.hip"""
import re, sys, subprocess, os
src = sys.argv[1]
out = "/tmp/scan_" + os.path.basename(src) + ".s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-I/root/repo/include", "-I/root/repo/cxrmate_amd/csrc", "-S", "-o", out, src, "--cuda-device-only"], stderr=subprocess.DEVNULL)
kernel = None; in_loop = False; stats = {}
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m: kernel = m.group(1); stats[kernel] = [0, 0, 0]; in_loop = False; continue
    if kernel is None: continue
    if "s_endpgm" in line: kernel = None; continue
    if "in Loop" in line or "Loop Header" in line: in_loop = True
    elif re.match(r"^\.LBB\d+_\d+:\s*$", line): in_loop = False
    if in_loop:
        if "global_load" in line or "buffer_load" in line: stats[kernel][0] += 1
        if re.search(r"s_waitcnt vmcnt\(0\)", line): stats[kernel][1] += 1
        if "s_cbranch_exec" in line: stats[kernel][2] += 1
for k, (l, w, b) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if w: print(f"{k[:90]:90s} loads-in-loops {l:4d}  vmcnt(0)-in-loops {w:3d}  exec-branches {b:3d}")
