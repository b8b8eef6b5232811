#!/usr/bin/env python3
"""Which compute units does bit i of a hipExtStreamCreateWithCUMask mask select? Launches a probe on streams with different masks and prints the XCC ids /
HW_ID fields of the workgroups (gfx950, 256 CUs in 8 XCDs)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops

def words(bits):
    w = [0] * 8
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return w

def show(name, bits, wgs=512):
    st = ops.masked_stream(words(bits))
    out = ops.probe_placement(wgs, stream=st)
    st.synchronize()
    o = out.cpu().numpy().astype("uint32")
    xcc = collections.Counter(int(v) for v in o[:, 0])
    cus = {(int(x), int((h >> 13) & 7), int((h >> 12) & 1), int((h >> 8) & 15)) for x, h in zip(o[:, 0], o[:, 1])}
    print(f"{name:34s} bits={len(bits):3d}  workgroups per XCC {dict(sorted(xcc.items()))}  distinct (xcc, se, sh, cu): {len(cus)}")
    return cus

full = ops.probe_placement(2048)
torch.cuda.synchronize()
o = full.cpu().numpy().astype("uint32")
print("unmasked launch of 2048 workgroups: per XCC", dict(sorted(collections.Counter(int(v) for v in o[:, 0]).items())),
      "distinct (xcc, se, sh, cu):", len({(int(x), int((h >> 13) & 7), int((h >> 12) & 1), int((h >> 8) & 15)) for x, h in zip(o[:, 0], o[:, 1])}))
for b in (0, 1, 2, 7, 8, 9, 31, 32, 33, 64, 255):
    print(b, sorted(show(f"single bit {b}", [b], 64)))
show("bits 0..31", range(32))
show("bits 0..63", range(64))
show("bits 0..95", range(96))
show("every 8th bit from 0 (32 bits)", range(0, 256, 8))
show("every 8th bit from 0 and 1 (64 bits)", [b for b in range(256) if b % 8 in (0, 1)])
show("every 4th bit (64 bits)", range(0, 256, 4))
show("bits 192..255", range(192, 256))
