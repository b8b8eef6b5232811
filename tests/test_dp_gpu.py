"""GPU: the data-parallel training step with world_size 2 (both ranks share the one visible GPU; gloo backend so that no second
device is needed). Checks the control flow the driver's multi-GPU bench exercises: side-stream gradient reduction between the
backward segments, 1/world scaling inside the fused optimiser, identical replicas after the step, and equality with a
single-process step on the concatenated batch."""
import os
import tempfile

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _worker(rank, world, path, q):
    import torch.distributed as dist
    from cxrmate_amd import modelling
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
    m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=21, perturb=0.05)
    opt = FusedAdamW(m, lr=1e-3)
    g = torch.Generator().manual_seed(5)
    px = torch.randn(4, 3, 96, 96, generator=g)
    ids = torch.randint(12, 1000, (4, 17), generator=g)
    ids[:, 0] = 1
    sl = slice(rank * 2, rank * 2 + 2)                     # study-level shard
    inp, lab = ids[sl, :-1].cuda(), ids[sl, 1:].cuda()
    tt = m.token_ids_to_token_type_ids(inp, [3])
    loss = tf_train_step(m, opt, px[sl].cuda(), inp, torch.ones_like(inp), tt, lab, pad_token_id=4)
    torch.cuda.synchronize()
    w = m.f32("decoder.bert.encoder.layer.1.output.dense.weight").cpu().clone()
    e = m.f32("encoder.cvt.encoder.stages.2.layers.0.intermediate.dense.weight").cpu().clone()
    q.put((rank, float(loss.item()), w.numpy(), e.numpy()))         # by value: the producer may exit before the consumer reads
    dist.destroy_process_group()


def test_dp2_train_step_matches_single_process():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from cxrmate_amd import modelling
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=_worker, args=(r, 2, os.path.join(d, "rdzv"), q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    (_, l0, w0, e0), (_, l1, w1, e1) = [(r, l, torch.from_numpy(w), torch.from_numpy(e)) for r, l, w, e in res]
    assert torch.equal(w0, w1) and torch.equal(e0, e1)                 # replicas stay identical
    # single process, whole batch: mean over 4 studies with equal token counts == mean of the two rank means
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
    m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=21, perturb=0.05)
    opt = FusedAdamW(m, lr=1e-3)
    g = torch.Generator().manual_seed(5)
    px = torch.randn(4, 3, 96, 96, generator=g)
    ids = torch.randint(12, 1000, (4, 17), generator=g)
    ids[:, 0] = 1
    inp, lab = ids[:, :-1].cuda(), ids[:, 1:].cuda()
    tt = m.token_ids_to_token_type_ids(inp, [3])
    loss = tf_train_step(m, opt, px.cuda(), inp, torch.ones_like(inp), tt, lab, pad_token_id=4)
    assert abs(loss.item() - 0.5 * (l0 + l1)) < 2e-3
    ws = m.f32("decoder.bert.encoder.layer.1.output.dense.weight").cpu()
    es = m.f32("encoder.cvt.encoder.stages.2.layers.0.intermediate.dense.weight").cpu()
    # AdamW's first step moves every weight by ~lr * sign(grad): compare the UPDATE directions / magnitudes
    ref = modelling.SingleCXREncoderDecoderModel(cfg, device="cpu", seed=21, perturb=0.05)
    for name, a, b in (("dec", ws, w0), ("enc", es, e0)):
        key = "decoder.bert.encoder.layer.1.output.dense.weight" if name == "dec" else "encoder.cvt.encoder.stages.2.layers.0.intermediate.dense.weight"
        base = ref.f32(key)
        da, db = (a - base).flatten(), (b - base).flatten()
        cos = float(da @ db / (da.norm() * db.norm() + 1e-30))
        assert cos > 0.98, (name, cos)
