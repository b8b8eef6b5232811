#!/bin/bash
# GPU call 23 of round 4: radix-select passes of the top-k threshold kernel with 16-byte loads, two in flight per thread
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "topk or top_k or select or sample or thr or loss or ce or reinforce" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_reward_scst_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "scst or sample or reinforce or scores or wrapped" 2>&1 | tail -3
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from cxrmate_amd import ops
g = torch.Generator().manual_seed(0)
x = (torch.randn(4080, 30000, generator=g) * 3).cuda()
for name, t in (("4080 x 30000 fp32", x), ("odd row stride (scalar path)", x[:, :29999])):
    for _ in range(3): ops.topk_threshold(t, 50)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): thr = ops.topk_threshold(t, 50)
    e1.record(); torch.cuda.synchronize()
    ref = t.topk(50, dim=1).values[:, -1]
    print(name, "us", e0.elapsed_time(e1) * 100, "exact", bool(torch.equal(thr.view(-1), ref)))
PY
