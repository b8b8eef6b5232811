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


def test_gradient_accumulation_equals_one_step_on_the_concatenated_batch():
    """tf_train_step(accumulate=(j, k)): k micro-steps (reference config/train/single_tf.yaml:16-17: mbatch_size 8 of accumulated_mbatch_size 32;
    DDP `no_sync` on the first k - 1) == ONE step on the concatenated batch. Eval-mode BatchNorm / no dropout so that the two computations are the
    same function of the data (batch statistics are per micro-batch in the reference too)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cxrmate_amd import modelling
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
    g = torch.Generator().manual_seed(9)
    px = torch.randn(4, 3, 96, 96, generator=g).cuda()
    ids = torch.randint(12, 1000, (4, 17), generator=g)
    ids[:, 0] = 1
    inp, lab = ids[:, :-1].cuda(), ids[:, 1:].cuda()
    names = ["decoder.bert.encoder.layer.1.output.dense.weight", "encoder.cvt.encoder.stages.2.layers.0.intermediate.dense.weight",
             "decoder.bert.embeddings.word_embeddings.weight"]

    def run(k):
        m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=21, perturb=0.05)
        opt = FusedAdamW(m, lr=1e-3)
        tt = m.token_ids_to_token_type_ids(inp, [3])
        per = 4 // k
        losses = []
        for j in range(k):
            sl = slice(j * per, (j + 1) * per)
            before = m.f32(names[0]).clone()
            losses.append(float(tf_train_step(m, opt, px[sl], inp[sl], torch.ones_like(inp[sl]), tt[sl], lab[sl], pad_token_id=4, accumulate=(j, k)).item()))
            if j + 1 < k:
                assert torch.equal(before, m.f32(names[0]))          # no optimiser step inside the accumulation window
        torch.cuda.synchronize()
        return np.mean(losses), {n: m.f32(n).cpu().clone() for n in names}, int(opt.t)

    l1, w1, t1 = run(1)
    l2, w2, t2 = run(2)
    l4, w4, t4 = run(4)
    assert t1 == t2 == t4 == 1
    assert abs(l1 - l2) < 2e-3 and abs(l1 - l4) < 2e-3
    for n in names:
        # AdamW's first step moves every weight by lr * sign-like(g): compare the UPDATES (equal up to the summation order of the gradients)
        torch.testing.assert_close(w2[n], w1[n], atol=2e-4, rtol=0)
        torch.testing.assert_close(w4[n], w1[n], atol=2e-4, rtol=0)


def _order_worker(rank, world, path, q):
    import torch.distributed as dist
    from cxrmate_amd import modelling
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
    m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=21, perturb=0.05)
    opt = FusedAdamW(m, lr=1e-3)
    log = opt.reducer.log = []
    g = torch.Generator().manual_seed(5)
    px = torch.randn(8, 3, 96, 96, generator=g)
    ids = torch.randint(12, 1000, (8, 17), generator=g)
    ids[:, 0] = 1
    marks = []
    for step in range(2):                                   # two optimiser steps of two micro-batches each
        for j in range(2):
            sl = slice(rank * 4 + j * 2, rank * 4 + j * 2 + 2)
            inp, lab = ids[sl, :-1].cuda(), ids[sl, 1:].cuda()
            tf_train_step(m, opt, px[sl].cuda(), inp, torch.ones_like(inp), m.token_ids_to_token_type_ids(inp, [3]), lab, pad_token_id=4, accumulate=(j, 2))
            marks.append(len(log))
    torch.cuda.synchronize()
    ranges = dict(split=opt.split, total=m._param_total, stages={s_: opt.stage_range(s_) for s_ in opt.stage_start})
    q.put((rank, list(log), marks, ranges, m.f32("encoder.cvt.encoder.stages.1.layers.0.intermediate.dense.weight").cpu().numpy()))
    dist.destroy_process_group()


def test_dp2_collective_order_is_identical_on_both_ranks_with_per_stage_buckets_under_accumulation():
    """Two ranks (gloo, one device), accumulate=(j, 2): no collective in the first micro-step (DDP no_sync), and in the last one every rank issues
    the SAME sequence of all-reduces -- decoder range first (under the encoder backward), then the encoder stage by stage as each stage's backward
    completes: stage 2 + projection head, stage 1, stage 0 (the reference's DDP buckets, config/train/single_tf.yaml:8). A rank that issued them
    in another order would deadlock or mix buckets on RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=_order_worker, args=(r, 2, os.path.join(d, "rdzv"), q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    (_, log0, marks0, rg, w0), (_, log1, marks1, _, w1) = res
    assert log0 == log1 and marks0 == marks1
    assert marks0[0] == 0 and marks0[2] == marks0[1]                   # micro-step 0 of each optimiser step: gradients stay local
    per_step = log0[: marks0[1]]
    assert log0[marks0[1]:] == per_step                                # the second optimiser step repeats the sequence
    st = rg["stages"]
    want = [(rg["split"], rg["total"]), st[2], st[1], st[0]]           # decoder | stage 2 + head | stage 1 | stage 0
    assert st[2][1] == rg["split"] and st[0][0] == 0 and st[0][1] == st[1][0] and st[1][1] == st[2][0]
    assert per_step == want, (per_step, want)
    assert np.array_equal(w0, w1)                                      # replicas identical after two accumulated steps


def _nccl_worker(path, q):
    import torch.distributed as dist
    from cxrmate_amd import dp, modelling
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    os.environ["CXR_DP_FORCE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{path}", rank=0, world_size=1)
    try:
        assert dp.active() and dp.world_size() == 1
        cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
        g = torch.Generator().manual_seed(5)
        px = torch.randn(2, 3, 96, 96, generator=g).cuda()
        ids = torch.randint(12, 1000, (2, 17), generator=g)
        ids[:, 0] = 1
        inp, lab = ids[:, :-1].cuda(), ids[:, 1:].cuda()
        out = {}
        for wire in (None, torch.bfloat16):
            m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=21, perturb=0.05)
            opt = FusedAdamW(m, lr=1e-3)
            opt.reducer.comm_dtype = wire
            calls = []
            real = dist.all_reduce

            def spy(t, *a, **k):
                calls.append((t.dtype, t.numel()))
                return real(t, *a, **k)

            dist.all_reduce = spy
            try:
                tt = m.token_ids_to_token_type_ids(inp, [3])
                loss = tf_train_step(m, opt, px, inp, torch.ones_like(inp), tt, lab, pad_token_id=4)
                torch.cuda.synchronize()
            finally:
                dist.all_reduce = real
            out[str(wire)] = (float(loss.item()), m.f32("decoder.bert.encoder.layer.1.output.dense.weight").cpu().numpy(), calls)
        # SCST statistics gather (one all_gather_into_tensor over RCCL)
        st = dp.gather_scst_statistics(torch.full((2, 4), 10, dtype=torch.int64, device="cuda"), torch.full((2, 6), 20, dtype=torch.int64, device="cuda"),
                                       torch.tensor([0.1, 0.2], device="cuda"), torch.tensor([0.3, 0.4], device="cuda"), 4, max_sampled=5, max_greedy=6)
        torch.cuda.synchronize()
        q.put(("ok", out, {k: v.cpu().numpy() for k, v in st.items()}))
    except Exception as e:                                       # pragma: no cover
        import traceback
        q.put(("error", traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_executes_the_reducer_and_the_gather_at_world_1():
    """backend='nccl' (= RCCL on ROCm) with ONE rank and CXR_DP_FORCE=1: the training step's gradient all-reduces (async work handles on the
    reducer's private stream, fp32 and bf16 wire format) and the SCST gather run through RCCL on the hardware. Sums over one rank are the identity,
    so the step must equal the collective-free step. (No multi-GPU box is available to this build: this is the RCCL code path, not a scaling number.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from cxrmate_amd import modelling
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        p = ctx.Process(target=_nccl_worker, args=(os.path.join(d, "rdzv"), q))
        p.start()
        status, out, st = q.get(timeout=600)
        p.join(120)
    assert status == "ok", out
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
    g = torch.Generator().manual_seed(5)
    px = torch.randn(2, 3, 96, 96, generator=g).cuda()
    ids = torch.randint(12, 1000, (2, 17), generator=g)
    ids[:, 0] = 1
    inp, lab = ids[:, :-1].cuda(), ids[:, 1:].cuda()
    m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=21, perturb=0.05)
    opt = FusedAdamW(m, lr=1e-3)
    loss = tf_train_step(m, opt, px, inp, torch.ones_like(inp), m.token_ids_to_token_type_ids(inp, [3]), lab, pad_token_id=4)
    ref = m.f32("decoder.bert.encoder.layer.1.output.dense.weight").cpu().numpy()
    l32, w32, c32 = out["None"]
    l16, w16, c16 = out["torch.bfloat16"]
    assert c32 and all(dt == torch.float32 for dt, _ in c32) and sum(n for _, n in c32) == sum(b - a for a, b in opt.reducer.buckets)
    assert c16 and all(dt == torch.bfloat16 for dt, _ in c16)
    assert abs(l32 - float(loss.item())) < 1e-6 and np.array_equal(w32, ref)        # fp32 wire at one rank: bit-identical step
    assert abs(l16 - float(loss.item())) < 1e-6 and np.abs(w16 - ref).max() < 2.5e-3   # bf16 wire: AdamW's first step is lr-sized (1e-3) whatever the gradient's last bits
    assert st["sampled"].shape == (2, 5) and st["greedy"].shape == (2, 6) and np.allclose(st["reward"], [0.1, 0.2]) and np.allclose(st["baseline"], [0.3, 0.4])


def _rehearsal_worker(rank, world, path, q):
    """One of EIGHT ranks sharing the one visible GPU (gloo on device tensors): two accumulated TF optimiser steps with the collective log, then
    one train-mode SCST step on the same replica family (a longitudinal model of its own)."""
    import torch.distributed as dist
    from cxrmate_amd import modelling
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    import time
    t_start = time.time()

    def phase(name):
        if rank == 0 and os.environ.get("CXR_TEST_PHASES"):
            print(f"[dp8 rank 0] {name}: {time.time() - t_start:.1f} s", flush=True)

    try:
        torch.set_num_threads(1)                                      # eight ranks share this host's cores
        dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        phase("process group up")
        cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
        # (the seeded initial weights are drawn ONCE by the parent and read from its scratch directory: eight concurrent CPU initialisations of
        # 28 M parameters took 55 s each)
        m = modelling.SingleCXREncoderDecoderModel(cfg, device="cuda:0", seed=None)
        m.load_state_dict(torch.load(os.path.join(os.path.dirname(path), "tf.pt")))
        opt = FusedAdamW(m, lr=1e-3)
        phase("TF model built")
        log = opt.reducer.log = []
        g = torch.Generator().manual_seed(5)
        px = torch.randn(2 * world, 3, 96, 96, generator=g)
        ids = torch.randint(12, 1000, (2 * world, 17), generator=g)
        ids[:, 0] = 1
        marks = []
        for step in range(2):
            for j in range(2):                                        # accumulate=(j, 2): one study per micro-step and rank
                sl = slice(rank * 2 + j, rank * 2 + j + 1)
                inp, lab = ids[sl, :-1].cuda(), ids[sl, 1:].cuda()
                tf_train_step(m, opt, px[sl].cuda(), inp, torch.ones_like(inp), m.token_ids_to_token_type_ids(inp, [3]), lab, pad_token_id=4, accumulate=(j, 2))
                marks.append(len(log))
        torch.cuda.synchronize()
        phase("TF steps done")
        tf_w = m.flat32[: m._param_total].detach().cpu().numpy().copy()
        # SCST: per-rank study shard, per-rank random streams
        cfl = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96, lora_r=8)
        ml = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfl, device="cuda:0", seed=None)
        ml.load_state_dict(torch.load(os.path.join(os.path.dirname(path), "scst.pt")))
        ml.train()
        ml.graph_decode = False                                       # (eight captures of the same step graphs would only cost test time)
        for p in ml.decoder.parameters():
            p.requires_grad_(True)
        optl = FusedAdamW(ml, lr=1e-3)
        phase("SCST model built")
        gi = torch.Generator().manual_seed(7)
        images = torch.randn(2 * world, 2, 3, 96, 96, generator=gi)
        prompt = torch.tensor([[8, 10, 9, 11, 1]] * 2, device="cuda")
        torch.manual_seed(100 + rank)
        out = scst_step(ml, optl, lambda t: (t % 2 == 0).float().mean(1), images[rank * 2: rank * 2 + 2].cuda(), prompt, None,
                        dict(bos=1, eos=2, sep=3, pad=4, pmt_sep=9), decoder_max_len=10)
        torch.cuda.synchronize()
        gl = out["global"]
        # (parameters only: the BatchNorm running statistics behind them are per-rank by design -- plain BN, no SyncBN, as in the reference)
        q.put((rank, "ok", list(log), marks, tf_w, ml.flat32[: ml._param_total].detach().cpu().numpy().copy(), gl["sampled"].cpu().numpy(), gl["reward"].cpu().numpy(),
               out["sampled"].cpu().numpy(), float(out["reward"])))
    except Exception:                                                 # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_dp8_rehearsal_on_one_device():
    """What can be proven about 8 ranks without an 8-GPU node (this build has never had one: NO scaling curve exists): EIGHT processes share the
    one GPU (gloo on device tensors) and run the data-parallel TF step under accumulate=(j, 2) and the train-mode SCST step. Every rank issues the
    same collective sequence (a different order deadlocks or mixes buckets on RCCL), the replicas are bit-identical afterwards, and
    gather_scst_statistics returns 8 x B records in rank order. Reference: config/train/single_tf.yaml:8 (`strategy: ddp`), data/prompt.py:142-213."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from cxrmate_amd import weights
    with tempfile.TemporaryDirectory() as d:
        torch.save(weights.init_encoder_decoder(gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96), seed=21, perturb=0.05), os.path.join(d, "tf.pt"))
        torch.save(weights.init_encoder_decoder(gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96, lora_r=8), seed=33, perturb=0.05),
                   os.path.join(d, "scst.pt"))
        procs = [ctx.Process(target=_rehearsal_worker, args=(r, world, os.path.join(d, "rdzv"), q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(120)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"][:1]
    assert all(p.exitcode == 0 for p in procs)
    _, _, log0, marks0, tfw0, sw0, gs0, gr0, _, rw0 = res[0]
    assert marks0[0] == 0 and marks0[2] == marks0[1] and len(log0) == 2 * marks0[1] and marks0[1] == 4      # decoder | stage 2 + head | stage 1 | stage 0
    for r in res[1:]:
        assert r[2] == log0 and r[3] == marks0                          # identical collective order on all 8 ranks
        assert np.array_equal(r[4], tfw0)                               # TF replicas bit-identical after two accumulated steps
        assert np.array_equal(r[5], sw0)                                # SCST replicas (LoRA included) bit-identical
        assert np.array_equal(r[6], gs0) and np.array_equal(r[7], gr0) and r[9] == rw0          # the same global view everywhere
    assert gs0.shape[0] == 2 * world and gr0.shape[0] == 2 * world
    for r in res:                                                       # rank order: rank r's own sampled rows sit at [2r, 2r + 2)
        own = r[8]
        assert np.array_equal(gs0[2 * r[0]: 2 * r[0] + 2, : own.shape[1]], own)
    assert len({r[8].tobytes() for r in res}) > 1                        # per-rank random streams / studies


def test_bench_gpus_8_rehearsal_and_refusal():
    """`bench.py --gpus 8` end to end with 8 ranks on the one GPU (CXR_BENCH_REHEARSAL=1: tiny configuration; gloo; every rank on device 0): spawn_ranks
    finds a port, hands each rank its slice of the host cores, the ranks rendezvous, run the TF and SCST keys with their all-reduces / gather, and
    rank 0 prints ONE JSON line for n_gpus = 8 that says it is a rehearsal. A process group whose size is not N is refused."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CXR_BENCH_REHEARSAL="1", CXR_SINGLE_DEVICE="1", CXR_DIST_BACKEND="gloo", CXR_BENCH_PREWARM="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0", "--batch", "2", "--seq-len", "16",
                        "--scst-steps", "1", "--new-tokens", "9", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["rccl_ranks"] == 8 and out["config"]["global_batch"] == 16 and "rehearsal" in out
    assert out["scaling"] == "weak" and out["value"] > 0 and np.isfinite(out["config"]["loss"])
    assert out["scst"]["value"] and np.isfinite(out["scst"]["loss"])
    # refusal: a one-rank environment asked for an 8-GPU number
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env1, capture_output=True,
                       text=True, timeout=600, cwd=root)
    assert r.returncode != 0 and "refusing to report" in (r.stderr + r.stdout)
