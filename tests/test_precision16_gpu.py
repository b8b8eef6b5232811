"""GPU: the reference's `precision: 16` caller mode on the drop-in classes.

`config/train/single_tf.yaml:21` (`precision: 16`, with `strategy: 'ddp'` at `:8`) makes Lightning wrap the reference's `training_step`
(modules/lightning_modules/single.py:449-475; SCST: longitudinal/scst/gt_prompt.py:62-142) in `torch.autocast('cuda', dtype=torch.float16)` and drive
the optimiser (`:426-431`) through `torch.amp.GradScaler`: `scale(loss).backward()`, `unscale_(optimizer)`, `step(optimizer)`, `update()`.
The engines compute in bf16 whatever the autocast state is (their kernels are not torch ops: autocast has nothing to cast), so what has to hold is:

  * `.logits` / `scores` are fp32 and bit-identical to the run without autocast (eval mode), and the callers' loss calls are still served by the
    BoundaryTensor handlers (autocast's own cross_entropy / log_softmax / nll_loss policies are fp32 and must not get in the way);
  * the scaler's 65536 x loss goes through the autograd bridges: gradients after `unscale_` equal the unscaled run's (a power-of-two scale commutes
    with every bf16 / fp32 rounding in the backward chain, so the tolerance is the run-to-run noise of the atomically summed gradients);
  * an inf in any gradient makes `scaler.step` skip the optimiser: master weights, bf16 shadow and AdamW moments untouched, scale halved;
  * both `torch.optim.AdamW` and `cxrmate_amd.optim.AdamW`, and the one-rank DistributedDataParallel wrap Lightning applies under `strategy: ddp`.
"""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cxrmate_amd import modelling
    return modelling


def _make_opt(kind, params, lr):
    if kind == "torch":
        return torch.optim.AdamW(params, lr=lr)
    from cxrmate_amd.optim import AdamW
    return AdamW(params, lr=lr)


def _close(a, b, rtol=1e-3):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) <= rtol * max(float(b.abs().max()), 1e-12) + 1e-9


@pytest.mark.parametrize("opt_kind", ["torch", "fused"])
def test_tf_caller_sequence_under_fp16_autocast_and_gradscaler(M, opt_kind):
    """reference single.py:449-475 (training_step) + :426-431 (AdamW) as Lightning runs them with `precision: 16`."""
    from cxrmate_amd.modelling import BoundaryTensor
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()

    def fresh():
        m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        kw = dict(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(),
                  decoder_token_type_ids=m.token_ids_to_token_type_ids(inp, [gu.SEP]), return_dict=True)
        return m, kw

    def step(m, kw, opt, scaler, poison=None):
        with torch.autocast("cuda", dtype=torch.float16, enabled=scaler is not None):
            logits = m(**kw).logits
            assert isinstance(logits, BoundaryTensor) and logits.dtype == torch.float32
            loss = torch.nn.functional.cross_entropy(logits.permute([0, 2, 1]), lab.cuda(), ignore_index=gu.PAD)
            assert loss.dtype == torch.float32 and type(loss) is torch.Tensor
        opt.zero_grad()
        if scaler is None:
            loss.backward()
        else:
            scaler.scale(loss).backward()
            if poison is not None:
                m.param(poison).grad.view(-1)[3] = float("inf")
            scaler.unscale_(opt)
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}
        if scaler is None:
            opt.step()
        else:
            scaler.step(opt)
            scaler.update()
        return logits.detach().as_subclass(torch.Tensor).clone(), float(loss), grads

    m0, kw0 = fresh()
    lg0, l0, g0 = step(m0, kw0, _make_opt(opt_kind, m0.parameters(), 1e-3), None)
    m1, kw1 = fresh()
    opt1 = _make_opt(opt_kind, m1.parameters(), 1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    lg1, l1, g1 = step(m1, kw1, opt1, scaler)
    assert torch.equal(lg0, lg1) and l0 == l1                          # autocast changes nothing in the forward
    bad = [n for n in g0 if not _close(g1[n], g0[n])]
    assert not bad, bad[:5]
    assert scaler.get_scale() == 65536.0
    # the update happened, and is the un-scaled run's update
    for (n, a), b in zip(m0.named_parameters(), m1.parameters()):
        d = (a - b).abs()
        assert float(d.mean()) < 2e-5 and float((d > 2e-3).float().mean()) < 1e-3, n
    assert float((m1.f32("decoder.cls.predictions.bias") - sd["decoder.cls.predictions.bias"].cuda()).abs().max()) > 0
    # second step with an inf planted in one gradient: the scaler skips the optimiser and backs the scale off; nothing of the store moves
    with torch.no_grad():
        m1(**kw1)                                                      # (after torch.optim's step the bf16 shadow follows at the next forward)
    w32, w16 = m1.flat32.clone(), m1.flat16.clone()
    step(m1, kw1, opt1, scaler, poison="decoder.bert.encoder.layer.0.output.dense.weight")
    torch.cuda.synchronize()
    assert torch.equal(m1.flat32, w32) and torch.equal(m1.flat16, w16)
    assert scaler.get_scale() == 32768.0
    # ... and the next clean step trains again, on the halved scale
    lg3, l3, g3 = step(m1, kw1, opt1, scaler)
    assert not torch.equal(m1.flat32, w32) and np.isfinite(l3) and l3 < l1
    key = "decoder.bert.encoder.layer.0.output.dense.weight"
    with torch.no_grad():
        m1(**kw1)
    assert torch.equal(m1.w16(key), m1.f32(key).to(torch.bfloat16))   # shadow == master for the forward that follows the (scaler-driven) step


def test_scst_caller_sequence_under_fp16_autocast_and_gradscaler(M):
    """reference scst/gt_prompt.py:144-246 (sample -> stack -> log_softmax(dim=1) -> nll_loss -> advantage-weighted mean) under autocast + scaler:
    same sampled ids, same per-token nll, same LoRA gradients after unscale_ as the plain run; the step is applied."""
    from cxrmate_amd.modelling import BoundaryTensor
    from cxrmate_amd.optim import AdamW
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    adv = torch.tensor([0.37, -0.21], device="cuda")

    def run(amp, opt_kind):
        m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        params = [p for p in m.parameters() if p.requires_grad]
        opt = _make_opt(opt_kind, params, 1e-3)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0) if amp else None
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            with torch.no_grad():
                eo = m.encoder(x.cuda())
            assert eo.last_hidden_state.dtype in (torch.bfloat16, torch.float32)
            torch.manual_seed(0)
            smp = m.generate.__wrapped__(m, input_ids=prompt.cuda(), special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS,
                                         eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True,
                                         num_beams=1, use_cache=True, output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=10)
            base = m.generate(encoder_outputs=eo, decoder_input_ids=prompt.cuda(), special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP],
                              max_length=11 + prompt.shape[1], bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD,
                              num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"]
            seqs = smp["sequences"][:, 1:] if torch.all(smp["sequences"][:, 0] == 1) else smp["sequences"]
            assert all(isinstance(s_, BoundaryTensor) and s_.dtype == torch.float32 for s_ in smp["scores"])
            logits = torch.stack(smp["scores"], dim=-1)
            sampled = seqs[:, prompt.shape[1]:]
            nll = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), sampled, ignore_index=gu.PAD, reduction="none")
            assert nll.dtype == torch.float32
            loss = (nll.sum(-1) * adv).mean()
        opt.zero_grad()
        if amp:
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
        else:
            loss.backward()
        torch.cuda.synchronize()
        grads = [p.grad.detach().float().clone() for p in params]
        before = [p.detach().clone() for p in params]
        if amp:
            scaler.step(opt); scaler.update()
        else:
            opt.step()
        torch.cuda.synchronize()
        moved = max(float((p.detach() - b).abs().max()) for p, b in zip(params, before))
        return seqs.cpu(), base.cpu(), nll.detach().cpu(), float(loss), grads, moved

    ref = run(False, "torch")
    for kind in ("torch", "fused"):
        got = run(True, kind)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])          # same sampled ids, same greedy baseline
        assert torch.equal(got[2], ref[2]) and got[3] == ref[3]
        for a, b in zip(got[4], ref[4]):
            assert _close(a, b, rtol=2e-3)
        assert got[5] > 0.0


def test_one_rank_ddp_wrap_trains_under_fp16_autocast_and_gradscaler(M, tmp_path):
    """`strategy: ddp` + `precision: 16` (single_tf.yaml:8,21) on ONE rank: DistributedDataParallel's reducer hooks receive the scaled gradients through
    the autograd route (binding is off under a process group), the scaler unscales and steps, the loss goes down."""
    import torch.distributed as dist
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    kw = dict(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(),
              decoder_token_type_ids=m.token_ids_to_token_type_ids(inp, [gu.SEP]), return_dict=True)
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(m)
        from cxrmate_amd.optim import AdamW
        opt = AdamW(m.parameters(), lr=1e-3)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        losses = []
        for _ in range(4):
            with torch.autocast("cuda", dtype=torch.float16):
                loss = torch.nn.functional.cross_entropy(ddp(**kw).logits.permute([0, 2, 1]), lab.cuda(), ignore_index=gu.PAD)
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())
            scaler.step(opt)
            scaler.update()
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert scaler.get_scale() == 65536.0 and losses[-1] < losses[0] - 1e-2, losses
    finally:
        dist.destroy_process_group()
