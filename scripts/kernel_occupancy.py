#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel as compiled for gfx950 (no GPU needed): python scripts/kernel_occupancy.py"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "cxrmate_amd", "csrc")
for f in sorted(os.listdir(csrc)):
    if not f.endswith(".hip"):
        continue
    out = os.path.join(tempfile.gettempdir(), "occ_" + f + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + csrc, os.path.join(csrc, f), "-o", out],
                   capture_output=True)
    txt = open(out).read()
    names = re.findall(r"^\s+\.globl\s+(\S+)", txt, re.M)
    stats = re.findall(r"; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n; ScratchSize: (\d+)\n(?:;.*\n)*?; LDSByteSize: (\d+).*\n(?:;.*\n)*?; Occupancy: (\d+)", txt)
    for n, st in zip(names, stats):
        dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0][:64]
        print(f"{f:18s} {dn:64s} vgpr {st[0]:>4s} agpr {st[1]:>3s} total {st[2]:>4s} scratch {st[3]:>3s} lds {int(st[4]) // 1024:>3d}K occ(regs) {st[5]}")
