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


def _scst_worker(rank, world, path, q):
    import torch.distributed as dist
    from cxrmate_amd import modelling
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96, lora_r=8)
    m = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device="cuda:0", seed=33, perturb=0.05)
    m.train()                                                   # the reference never leaves train mode inside training_step
    for p in m.decoder.parameters():
        p.requires_grad_(True)                                   # scst/gt_prompt.py:38-40
    opt = FusedAdamW(m, lr=1e-3)
    g = torch.Generator().manual_seed(7)
    images = torch.randn(4, 2, 3, 96, 96, generator=g)
    images[1, 1] = 0                                             # a study with one image
    sl = slice(rank * 2, rank * 2 + 2)                           # study-level shard
    prompt = torch.tensor([[8, 10, 9, 11, 1]] * 2, device="cuda")
    torch.manual_seed(100 + rank)                                # per-rank sampling / dropout streams

    def reward_fn(ids):                                          # deterministic stand-in reward: share of even token ids
        return (ids % 2 == 0).float().mean(1)

    out = scst_step(m, opt, reward_fn, images[sl].cuda(), prompt, None, dict(bos=1, eos=2, sep=3, pad=4, pmt_sep=9), decoder_max_len=12)
    torch.cuda.synchronize()
    w = m.f32("decoder.base_model.model.bert.encoder.layer.1.output.dense.weight").cpu().clone()
    la = m.f32("decoder.base_model.model.bert.encoder.layer.0.attention.self.query.lora_A.default.weight").cpu().clone()
    gl = out["global"]
    q.put((rank, float(out["loss"].item()), float(out["reward"]), w.numpy(), la.numpy(), gl["sampled"].cpu().numpy(), gl["reward"].cpu().numpy(),
           out["sampled"].cpu().numpy()))
    dist.destroy_process_group()


def test_dp2_scst_step_train_mode_replicas_and_global_statistics():
    """Two ranks (gloo, one device) run scst_step under model.train() on their own study shards: the decoder replicas (LoRA included) stay
    identical after the all-reduced AdamW step, and every rank sees the same all-gathered sampled sequences / rewards, its own shard at its
    rank's position."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=_scst_worker, args=(r, 2, os.path.join(d, "rdzv"), q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    (_, l0, r0, w0, a0, gs0, gr0, s0), (_, l1, r1, w1, a1, gs1, gr1, s1) = res
    assert np.array_equal(w0, w1) and np.array_equal(a0, a1)            # replicas identical after the step
    assert np.isfinite(l0) and np.isfinite(l1)
    assert np.array_equal(gs0, gs1) and np.array_equal(gr0, gr1) and r0 == r1      # the same global view on both ranks
    assert gs0.shape[0] == 4 and np.array_equal(gs0[:2, :s0.shape[1]], s0) and np.array_equal(gs0[2:, :s1.shape[1]], s1)
    assert abs(r0 - float(gr0.mean())) < 1e-6
    assert not np.array_equal(s0, s1)                                   # different studies / random streams per rank
