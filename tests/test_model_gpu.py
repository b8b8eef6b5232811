"""GPU: the MI355X model path (HIP kernels through the C ABI) against (a) the golden fixtures produced by the REFERENCE and
(b) the fp32 CPU oracle on the same seeded inputs.

Tolerances (the reference's own bf16-vs-fp32 deviation is rel-rms 1.1-1.2 %, cosine 0.9999):
  activations / logits : rel-rms <= 2.5e-2 and cosine >= 0.9995 vs fp32
  gradients            : rel-rms <= 6e-2 (bf16 activations in the backward chain)
  greedy / argmax      : identical wherever the fp32 top-1/top-2 margin exceeds MARGIN (measured logit error bound), teacher-forced
  token indexing       : bit-exact
"""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

ACT_RMS, ACT_COS, GRAD_RMS, MARGIN = 2.5e-2, 0.9995, 6e-2, 0.05


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cxrmate_amd import modelling
    return modelling


def check_act(a, ref, what, rms=ACT_RMS, cos=ACT_COS):
    r, c = gu.rel_rms(a, ref), gu.cosine(a, ref)
    assert np.isfinite(a).all(), what
    assert r <= rms and c >= cos, f"{what}: rel_rms {r:.4f} cosine {c:.6f}"


def test_state_dict_keys_match_reference_layout(M):
    from cxrmate_amd import weights
    cfg = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96)
    m = M.MultiCXREncoderDecoderModel(cfg, seed=3)
    ref_keys = list(weights.encoder_decoder_param_shapes(cfg).keys())
    assert sorted(m.state_dict().keys()) == sorted(ref_keys)
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(s)) for k, s in weights.encoder_decoder_param_shapes(cfg).items()
                                                         if not weights.is_buffer(k) and k not in weights.tied_aliases(cfg.decoder))
    sd = weights.init_encoder_decoder(cfg, seed=9, perturb=0.05)
    m.load_state_dict(sd)
    back = m.state_dict()
    for k in ("encoder.cvt.encoder.stages.2.cls_token", "decoder.cls.predictions.decoder.weight", "decoder.bert.embeddings.word_embeddings.weight"):
        assert torch.equal(back[k].cpu(), sd[k])
    cfgl = gu.tiny_config(vocab_size=1000, decoder_layers=2, image_size=96, lora_r=8)
    ml = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfgl, seed=3)
    t = sum(p.numel() for p in ml.decoder.parameters() if p.requires_grad)
    assert t == 2 * 2 * 2 * 8 * 768                         # layers x {query,key} x {A,B} x r x d  (147456 for 6 layers, cxrmate.ipynb:89)
    assert all(not p.requires_grad for p in ml.encoder.parameters())


def _stage_activations(m, x):
    """CvtModel hidden_states (NCHW, class token split off) from the engine's token-major stage outputs"""
    taps = []
    with torch.no_grad():
        m._enc.forward(m._pixels(x).view(-1, *x.shape[-3:]), stage_outputs=taps)
    return [t.float().view(t.shape[0], H, W, t.shape[2]).permute(0, 3, 1, 2).contiguous().cpu() for t, H, W in taps]


def test_encoder_matches_reference_fixture(M):
    g, cfg, sd, x = gu.encoder_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for i, h in enumerate(_stage_activations(m, x.cuda())):                          # per-stage activations of the reference's CvtModel
        assert list(h.shape) == g[f"stage{i}_shape"].tolist()
        check_act(gu.sample(h), g[f"stage{i}_sample"], f"stage {i} hidden state")
    with torch.no_grad():
        out = m.encoder(x.cuda())
    h = out.last_hidden_state.float().cpu()
    assert list(h.shape) == g["last_hidden_state_shape"].tolist()
    assert np.array_equal(out.attention_mask.cpu().numpy(), g["attention_mask"])
    check_act(gu.sample(h, 16384), g["last_hidden_state_sample"], "encoder last_hidden_state")
    np.testing.assert_allclose(gu.stats(h)[1], g["last_hidden_state_stats"][1], rtol=2e-2)


def _grads_by_name(m, names):
    return {n: m.param(n).grad.detach().float().cpu() for n in names}


def test_tf_single_logits_loss_grads(M):
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    tt_dev = m.token_ids_to_token_type_ids(inp, [gu.SEP])
    assert np.array_equal(tt_dev.cpu().numpy(), g["token_type_ids"])
    out = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt_dev, return_dict=True)
    logits = out.logits
    assert logits.dtype == torch.float32 and list(logits.shape) == [3, 24, 1000]
    check_act(gu.sample(logits, 16384), g["logits_sample"], "tf logits")
    loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)      # the caller's own loss (single.py:467)
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    safe = g["logits_margin"] > MARGIN
    assert np.array_equal(logits.argmax(-1).cpu().numpy()[safe], g["logits_argmax"][safe])
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    grads = _grads_by_name(m, names)
    for i, n in enumerate(names):
        r = gu.rel_rms(gu.sample(grads[n], 2048), g[f"grad{i}_sample"])
        assert r < GRAD_RMS, f"{n}: rel_rms {r:.4f}"
    # fused loss kernel (bench path) == caller's loss on the same logits
    from cxrmate_amd import ops
    w = ops.ce_weights(lab.cuda().reshape(-1), gu.PAD)
    l2, _, _ = ops.softmax_ce(logits.detach().reshape(-1, 1000), lab.cuda().reshape(-1), gu.PAD, w, need_grad=False)
    assert abs(l2.item() - loss.item()) < 1e-4


def test_decoder_inputs_embeds_equals_input_ids(M):
    """forward(decoder_inputs_embeds=E[ids]) == forward(decoder_input_ids=ids) (BertEmbeddings with inputs_embeds, TF5:bert:84-108); the gradient
    wrt the given vectors is what the word-embedding rows would have received through the lookup."""
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    ttd = m.token_ids_to_token_type_ids(inp, [gu.SEP])
    kw = dict(pixel_values=x.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=ttd, return_dict=True)
    wkey = "decoder.bert.embeddings.word_embeddings.weight"
    a = m(decoder_input_ids=inp.cuda(), **kw).logits
    a.float().square().mean().backward()
    gw_ids = m.param(wkey).grad.clone()
    for p_ in m.parameters():
        p_.grad = None
    E = m.f32(wkey)[inp.cuda()].to(torch.bfloat16).float().requires_grad_(True)          # the bf16 table rows the lookup reads
    b = m(decoder_inputs_embeds=E, **kw).logits
    assert torch.equal(a, b)
    b.float().square().mean().backward()
    gw_emb = m.param(wkey).grad                                                           # only the tied LM-head part is left on the table
    scat = torch.zeros_like(gw_ids).index_add_(0, inp.cuda().reshape(-1), E.grad.reshape(-1, E.shape[-1]))
    err = float((gw_emb + scat - gw_ids).norm() / gw_ids.norm())
    assert err < 2e-3, err
    with pytest.raises(ValueError):
        m(decoder_input_ids=inp.cuda(), decoder_inputs_embeds=E, **kw)


def test_torch_optimizer_updates_reach_the_kernels(M):
    """INTEGRATION.md path A: the caller's own torch.optim.AdamW over model.parameters() (reference single.py:426-431). The optimiser edits the
    fp32 master through the Parameter views; the bf16 shadow every kernel reads must follow without any explicit call."""
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    ttd = m.token_ids_to_token_type_ids(inp, [gu.SEP])
    kw = dict(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=ttd, return_dict=True)
    losses = []
    for _ in range(3):
        logits = m(**kw).logits
        loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[1] < losses[0] - 1e-3 and losses[2] < losses[1] - 1e-3, losses          # the steps act on the network that is evaluated
    with torch.no_grad():
        after = m(**kw).logits
    assert float((after - logits).abs().max()) > 1e-3
    key = "decoder.bert.encoder.layer.0.output.dense.weight"
    assert torch.equal(m.w16(key), m.f32(key).to(torch.bfloat16))                      # shadow == master after the refresh


def test_autograd_bridges_bind_gradients_with_torch_semantics(M, monkeypatch):
    """The bridges hand parameter gradients over as views of the flat gradient buffer when nothing foreign has to be accumulated into
    (modelling._grads_bindable). Against the plain path (CXR_BIND_GRADS=0: gradients returned through autograd) on the same eval-mode model:
    one backward after zero_grad(set_to_none=True); two backward calls without zeroing (accumulation: 2x); zero_grad(set_to_none=False) in between;
    and foreign .grad tensors (the aliasing ones are detached first, autograd accumulates)."""
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    ttd = None

    def grads(bind, script):
        monkeypatch.setenv("CXR_BIND_GRADS", "1" if bind else "0")
        m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        kw = dict(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(),
                  decoder_token_type_ids=m.token_ids_to_token_type_ids(inp, [gu.SEP]), return_dict=True)
        ps = list(m.parameters())

        def backward():
            torch.nn.functional.cross_entropy(m(**kw).logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD).backward()

        script(backward, ps)
        torch.cuda.synchronize()
        return [p.grad.detach().float().clone() for p in ps]

    def once(backward, ps):
        backward()

    def twice(backward, ps):
        backward(); backward()

    def zero_in_place(backward, ps):
        backward()
        for p in ps:
            p.grad.zero_()                                       # optimizer.zero_grad(set_to_none=False)
        backward()

    def foreign(backward, ps):
        backward()
        for i, p in enumerate(ps):
            if i % 2:
                p.grad = torch.ones_like(p)                      # somebody else's tensors in half of the slots
        backward()

    ref1 = grads(False, once)
    for name, script, scale in (("fresh", once, 1.0), ("accumulate", twice, 2.0), ("zeroed in place", zero_in_place, 1.0)):
        got = grads(True, script)
        for a, b in zip(got, ref1):
            assert torch.allclose(a, scale * b, rtol=2e-3, atol=2e-5 * max(1.0, float(b.abs().max()))), name
    got, ref = grads(True, foreign), grads(False, foreign)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-5 * max(1.0, float(b.abs().max()))), ("foreign", i)


def test_gradient_binding_is_off_wherever_autograd_hooks_could_observe_it(M, tmp_path):
    """Binding p.grad to the flat gradient buffer bypasses autograd's AccumulateGrad nodes: Tensor.register_hook, post-accumulate-grad hooks and the
    DistributedDataParallel reducer (which hangs its hooks on those nodes) would never fire. With a hook on any parameter of a bridge's half of the
    model, or under a torch.distributed process group, the gradients take the autograd route: hooks see them, and a one-rank DDP wrap of the model -- what Lightning does to the
    reference's module under `strategy: ddp` -- trains for several iterations (the reducer raises in the second one if a gradient never arrives)."""
    import torch.distributed as dist
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    kw = dict(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(),
              decoder_token_type_ids=m.token_ids_to_token_type_ids(inp, [gu.SEP]), return_dict=True)

    def loss_of(model):
        return torch.nn.functional.cross_entropy(model(**kw).logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)

    loss_of(m).backward()
    torch.cuda.synchronize()
    ref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    lo, hi = m.gflat.data_ptr(), m.gflat.data_ptr() + 4 * m.gflat.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in m.parameters())                # no hooks: bound
    seen = {}
    name = "decoder.bert.encoder.layer.0.output.dense.weight"
    p0 = dict(m.named_parameters())[name]
    h1 = p0.register_hook(lambda gr: seen.__setitem__("tensor_hook", gr.detach().clone()) or gr * 2.0)
    h2 = p0.register_post_accumulate_grad_hook(lambda p_: seen.__setitem__("post_hook", p_.grad.detach().clone()))
    m.zero_grad(set_to_none=True)
    loss_of(m).backward()
    torch.cuda.synchronize()
    assert torch.allclose(seen["tensor_hook"], ref[name], rtol=1e-3, atol=1e-6) and torch.allclose(seen["post_hook"], 2.0 * ref[name], rtol=1e-3, atol=1e-6)
    assert torch.allclose(p0.grad, 2.0 * ref[name], rtol=1e-3, atol=1e-6)           # the hook's return value is what accumulates, as in torch
    # the hooked parameter's half of the model (one autograd bridge per half) went through autograd; the other half has nothing that could observe
    # the binding and keeps it
    assert not any(lo <= p.grad.data_ptr() < hi for n, p in m.named_parameters() if n.startswith("decoder."))
    assert all(lo <= p.grad.data_ptr() < hi for n, p in m.named_parameters() if n.startswith("encoder."))
    h1.remove(); h2.remove()
    # one-rank DistributedDataParallel (gloo on device tensors): three iterations with torch.optim.AdamW, gradients equal to the unwrapped model's
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(m)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            loss_of(ddp).backward()
            if it == 0:
                torch.cuda.synchronize()
                for n, p in m.named_parameters():
                    assert torch.allclose(p.grad, ref[n], rtol=2e-3, atol=2e-5 * max(1.0, float(ref[n].abs().max()))), n
            opt.step()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def test_fused_adamw_behind_the_torch_optimizer_interface(M):
    """cxrmate_amd.optim.AdamW(model.parameters(), lr) -- what a reference caller's configure_optimizers (modules/lightning_modules/single.py:426-431)
    returns instead of torch.optim.AdamW: same weights as torch's optimiser after several caller-style steps (zero_grad(set_to_none=True) ->
    backward -> step), parameter groups / lr changes honoured, state_dict round trip into a fresh optimiser, and the bf16 shadow follows."""
    from cxrmate_amd.optim import AdamW
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()

    def run(make_opt, steps=4, reload_at=None):
        m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        kw = dict(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(),
                  decoder_token_type_ids=m.token_ids_to_token_type_ids(inp, [gu.SEP]), return_dict=True)
        opt = make_opt(m)
        losses = []
        for it in range(steps):
            if reload_at == it:
                state = opt.state_dict()
                opt = make_opt(m)
                opt.load_state_dict(state)
            if it == 2:
                for grp in opt.param_groups:
                    grp["lr"] = 5e-4                                   # what an lr scheduler does
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(m(**kw).logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        return m, losses, opt

    groups = lambda m: [{"params": [p for n, p in m.named_parameters() if n.startswith("encoder.")], "weight_decay": 0.0},
                        {"params": [p for n, p in m.named_parameters() if not n.startswith("encoder.")]}]
    mt, lt, _ = run(lambda m: torch.optim.AdamW(groups(m), lr=1e-3))
    mf, lf, of = run(lambda m: AdamW(groups(m), lr=1e-3))
    mr, lr_, _ = run(lambda m: AdamW(groups(m), lr=1e-3), reload_at=2)
    assert len(of._plans) == 2                                          # the steady-state path (one launch per contiguous run) was taken
    assert np.allclose(lt, lf, atol=5e-3) and np.allclose(lf, lr_, atol=1e-3), (lt, lf, lr_)
    for (n, a), b, c in zip(mt.named_parameters(), mf.parameters(), mr.parameters()):
        # Adam's first steps move every weight by ~lr * sign(g): the two implementations may differ where a gradient is numerically zero
        d = (a - b).abs()
        if not n.endswith("key.bias"):                                  # (a key bias shifts every score of a softmax row alike: its gradient is rounding noise)
            assert float(d.mean()) < 4e-5 and float((d > 2e-3).float().mean()) < 1e-3, (n, float(d.mean()), float(d.max()))
        # state_dict round trip mid-training changes nothing (two RUNS are compared: the embedding-table / LayerNorm parameter gradients are summed
        # with float atomics, so their last bits differ from run to run and Adam turns a flipped sign of a ~0 gradient into 2 * lr)
        d2 = (b - c).abs()
        assert float(d2.mean()) < 1e-5 and float((d2 > 2e-3).float().mean()) < 1e-3, (n, float(d2.mean()), float(d2.max()))
    key = "decoder.bert.encoder.layer.0.output.dense.weight"
    assert torch.equal(mf.w16(key), mf.f32(key).to(torch.bfloat16))     # shadow written with the master
    st = of.state_dict()["state"]
    assert len(st) == len(list(mf.parameters())) and all(int(v["step"]) == 4 for v in st.values())


def test_graphed_tf_step_leaves_no_stale_weight_copies(M):
    """After K replays of GraphedTFStep every weight-derived buffer the engines cache per weight version (BN folds, tap re-layouts, LoRA merges,
    transposed copies, packed decode weights, decode sessions) must describe the CURRENT weights: forward and generate of the stepped model ==
    those of a fresh model loaded with its state_dict. (Graphed and eager steps themselves agree only up to the fp32-atomics noise Adam
    amplifies on near-zero gradients, so they are compared through the loss.)"""
    from cxrmate_amd import training
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    res = {}
    for graphed in (False, True):
        m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        opt = training.FusedAdamW(m, lr=1e-3)
        ttd = m.token_ids_to_token_type_ids(inp, [gu.SEP])
        args = (x.cuda(), inp.cuda(), am.cuda(), ttd, lab.cuda())
        kw = dict(pixel_values=args[0], decoder_input_ids=args[1], decoder_attention_mask=args[2], decoder_token_type_ids=ttd)
        gen = dict(pixel_values=args[0], special_token_ids=[gu.SEP], max_length=8, bos_token_id=gu.BOS, eos_token_id=None, pad_token_id=gu.PAD)
        with torch.no_grad():
            m(**kw); m.generate(**gen)                                     # derive the version-keyed buffers (and a decode session) BEFORE stepping
            if graphed:
                step = training.GraphedTFStep(m, opt, *args, gu.PAD, warmup=2)         # two eager steps inside, then three replays
                losses = [float(step(*args)) for _ in range(3)]
            else:
                losses = [float(training.tf_train_step(m, opt, *args, gu.PAD)) for _ in range(5)][2:]
            lg, seq = m(**kw).logits, m.generate(**gen)
            fresh = M.SingleCXREncoderDecoderModel(cfg, seed=None)
            fresh.load_state_dict(m.state_dict())
            lg2, seq2 = fresh(**kw).logits, fresh.generate(**gen)
        assert torch.equal(lg, lg2), float((lg - lg2).abs().max())
        assert torch.equal(seq, seq2)
        res[graphed] = losses
    assert res[True][0] > res[True][-1]                                   # the replayed steps train
    np.testing.assert_allclose(res[True], res[False], rtol=2e-2)


def test_training_step_gradients_do_not_depend_on_stream_overlap(M, monkeypatch):
    """The eager training step puts every weight-gradient kernel (and the embedding-table / LayerNorm parameter sums) on a side stream and joins
    it once per step: the gradients it leaves in the flat buffer must equal those of the same step issued on ONE stream (fp32 atomics reorder,
    nothing else), and the bf16-logits loss must match the fp32-logits loss."""
    from cxrmate_amd import ops, training
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    res = {}
    # overlap = the number of weight-gradient launches issued behind one fork event (ops._side_defer; 1000: only the joins flush), 0 = one stream
    for overlap in (4, 1, 1000, 0):
        monkeypatch.setattr(ops, "_WGRAD_BATCH", max(overlap, 1))
        m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        opt = training.FusedAdamW(m, lr=0.0)
        ttd = m.token_ids_to_token_type_ids(inp, [gu.SEP])
        args = (m, opt, x.cuda(), inp.cuda(), am.cuda(), ttd, lab.cuda(), gu.PAD, None, 0)
        for rep in range(3):                                      # repeated: a missing cross-stream dependency shows up as run-to-run noise
            if overlap:
                with training.wgrad_overlap():
                    loss, esaved, denc = training._phase_fwd_loss_decbwd(*args, join=False)
                    training._phase_encbwd(m, esaved, denc)
            else:
                assert ops.WGRAD_STREAM is None
                loss, esaved, denc = training._phase_fwd_loss_decbwd(*args)
                training._phase_encbwd(m, esaved, denc)
            torch.cuda.synchronize()
            res.setdefault(overlap, []).append((float(loss.item()), m.gflat.clone()))
            assert not ops._SIDE_DEFERRED and not ops._SIDE_PENDING                    # every step ends flushed and joined
    (l0, g0) = res[0][0]
    for l1, g1 in res[4] + res[1] + res[1000] + res[0][1:]:
        assert abs(l1 - l0) < 1e-5
        err = float((g1 - g0).norm() / g0.norm())
        assert err < 2e-4, err


def test_early_decoder_adamw_equals_the_single_update(M, monkeypatch):
    """On one rank tf_train_step updates the decoder's parameters on the weight-gradient stream while the encoder backward runs and the encoder's at
    the end (training._EARLY_DEC_ADAMW); the step counter advances once. Same parameters, moments and losses as one AdamW launch at the end of the
    step, bit for bit where the gradient kernels are deterministic (the loss trajectory otherwise)."""
    from cxrmate_amd import training
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    out = {}
    for early in (True, "decoder only", False):                     # True: the decoder's range AND the encoder's last stage + head are updated early
        monkeypatch.setattr(training, "_EARLY_DEC_ADAMW", bool(early))
        monkeypatch.setattr(training, "_EARLY_ENC_ADAMW", early is True)
        m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
        m.load_state_dict(sd)
        m.eval()                                                    # no dropout seeds: the two runs see the same network
        opt = training.FusedAdamW(m, lr=1e-3)
        ttd = m.token_ids_to_token_type_ids(inp, [gu.SEP])
        losses = [float(training.tf_train_step(m, opt, x.cuda(), inp.cuda(), am.cuda(), ttd, lab.cuda(), gu.PAD)) for _ in range(3)]
        torch.cuda.synchronize()
        out[early] = (losses, m.flat32.clone(), opt.m.clone(), opt.v.clone(), opt.t)
    assert out[True][4] == out["decoder only"][4] == out[False][4] == 3
    assert out[True][0][-1] < out[True][0][0]
    for mode in (True, "decoder only"):
        np.testing.assert_allclose(out[mode][0], out[False][0], rtol=1e-4)
        for a, b in zip(out[mode][1:4], out[False][1:4]):
            err = float((a - b).norm() / b.norm())
            assert err < 1e-4, (mode, err)
    lo, hi = training.FusedAdamW.stage_range(opt, opt.enc_last_stage)          # the early-updated encoder range is not empty in this configuration
    assert 0 < lo < hi <= opt.split


def _hash_masks(m, cfg, B, T, S, enc_seed, dec_seed, Bn):
    """The dropout / DropPath factors the kernels regenerate from their counter-based hash, materialised for the CPU oracle."""
    from cxrmate_amd import ops
    from cxrmate_amd.decoder import SITE_EMBED, _site
    d, H, D = cfg.decoder, cfg.decoder.num_attention_heads, cfg.decoder.hidden_size
    ph, pa = d.hidden_dropout_prob, d.attention_probs_dropout_prob
    hid = lambda site: ops.dropout_mask(B * T, D, ph, dec_seed, site, T, factor=True).view(B, T, D).cpu()
    prob = lambda site, Tk: ops.dropout_mask(B * H * T, Tk, pa, dec_seed, site, T, factor=True).view(B, H, T, Tk).cpu()
    dropout = {"embed": hid(SITE_EMBED)}
    for l in range(d.num_hidden_layers):
        if d.lora_r:
            lora = lambda site: ops.dropout_mask(B * T, D, d.lora_dropout, dec_seed, site, T, factor=True).view(B, T, D).cpu()
            dropout[(l, "lora_q")], dropout[(l, "lora_k")] = lora(_site(l, 5)), lora(_site(l, 6))
        dropout[(l, "self_probs")], dropout[(l, "self_out")] = prob(_site(l, 0), T), hid(_site(l, 1))
        dropout[(l, "cross_probs")], dropout[(l, "cross_out")] = prob(_site(l, 2), S), hid(_site(l, 3))
        dropout[(l, "ffn_out")] = hid(_site(l, 4))
    paths = {}
    e = cfg.encoder
    for s_ in range(len(e.depth)):
        rate = m._enc._drop_path_rate(s_)
        for l in range(e.depth[s_]):
            if rate > 0:
                gl = sum(e.depth[:s_]) + l
                paths[(s_, l)] = tuple(ops.dropout_mask(Bn, 1, rate, enc_seed, 1000 + 2 * gl + j, 1, factor=True).view(Bn).cpu() for j in (0, 1))
    return dropout, paths


def test_tf_train_mode_matches_oracle_with_same_masks(M):
    """model.train(): batch-statistics BatchNorm with running-stat updates, dropout (hidden + attention probabilities) and DropPath.
    The oracle -- pinned against the reference's own train-mode pass in tests/test_oracle_golden.py -- gets the masks the kernels hash."""
    from oracle import bert as obert, cvt as ocvt, generate as ogen
    g, cfg, sd, x, inp, lab, am, tt, _, _ = gu.tf_single_train_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    assert not m.training                                                  # constructed like from_pretrained(): eval
    m.train()
    torch.manual_seed(11)
    tt_dev = m.token_ids_to_token_type_ids(inp, [gu.SEP])
    out = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt_dev, return_dict=True)
    logits = out.logits
    loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
    loss.backward()
    torch.manual_seed(11)                                                  # the two seeds the pass drew from torch's CPU generator
    enc_seed = torch.full((1,), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), dtype=torch.int32, device="cuda")
    dec_seed = torch.full((1,), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), dtype=torch.int32, device="cuda")
    B, T = inp.shape
    S = cfg.encoder.tokens_per_image
    dropout, paths = _hash_masks(m, cfg, B, T, S, enc_seed, dec_seed, x.shape[0])
    assert len(paths) == cfg.encoder.depth[2] and abs(m._enc._drop_path_rate(2) - 0.1) < 1e-6       # linspace(0, .1, 3)[2] (quirk Q2)
    names = [str(n) for n in g["grad_names"]]
    sd2 = {k: v.clone() for k, v in sd.items()}
    leaves = {n: sd2[n].requires_grad_(True) for n in names}
    h, _ = ocvt.encoder_forward(x, sd2, cfg.encoder, bn_train=True, bn_momentum=cfg.encoder.bn_momentum, drop_path=paths)
    ref = obert.decoder_forward(inp, sd2, cfg.decoder, h, None, am, tt, None, dropout=dropout)
    rloss = ogen.tf_cross_entropy(ref, lab, gu.PAD)
    rloss.backward()
    check_act(logits.detach().float().cpu().numpy(), ref.detach().numpy(), "train-mode logits")
    assert abs(loss.item() - rloss.item()) < 2e-2
    grads = _grads_by_name(m, names)
    for n in names:
        rn = leaves[n].grad.norm().item()
        if rn < 1e-7:
            # structurally zero: BatchNorm over batch statistics after a DEPTHWISE conv cancels any per-channel rescaling of the conv input,
            # so the weight of the LayerNorm feeding q/k/v gets no gradient in train mode (fp32 reference: 6e-9)
            assert grads[n].norm().item() < 1e-6, n
            continue
        r = gu.rel_rms(grads[n].numpy(), leaves[n].grad.numpy())
        assert r < GRAD_RMS, f"{n}: rel_rms {r:.4f}"
    got = m.state_dict()
    for i in range(3):
        key = str(g[f"bn{i}_key"])
        np.testing.assert_allclose(got[key + "running_mean"].cpu().numpy(), sd2[key + "running_mean"].numpy(), rtol=2e-2, atol=2e-3)
        np.testing.assert_allclose(got[key + "running_var"].cpu().numpy(), sd2[key + "running_var"].numpy(), rtol=2e-2, atol=2e-3)
        assert int(got[key + "num_batches_tracked"]) == 1
    # eval() afterwards: deterministic, uses the MOVED running statistics
    m.eval()
    with torch.no_grad():
        e1 = m.encoder(x.cuda()).last_hidden_state.float().cpu()
        e2 = m.encoder(x.cuda()).last_hidden_state.float().cpu()
        href, _ = ocvt.encoder_forward(x, {k: v.detach() for k, v in sd2.items()}, cfg.encoder)
    assert torch.equal(e1, e2)
    check_act(e1.numpy(), href.numpy(), "eval after train-mode step")


def test_tf_longitudinal_train_mode_lora_dropout(M):
    """Longitudinal model under model.train(): frozen encoder with batch-statistics BatchNorm + DropPath, decoder dropouts, and
    lora_dropout on the input of the rank-8 branch (which therefore cannot be merged into the weight) -- against the oracle fed with the
    masks the kernels hash; the oracle itself is pinned on the reference's train-mode pass (tests/test_oracle_golden.py)."""
    from oracle import bert as obert, cvt as ocvt, generate as ogen
    g, cfg, sd, x, prompt, inp, lab, am, tt, pos, _, _ = gu.tf_longitudinal_train_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    m.train()
    torch.manual_seed(23)
    tt_dev = m.token_ids_to_token_type_ids(inp, [gu.PMT_SEP, gu.BOS, gu.SEP], [0, 1, 0, 1])
    out = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt_dev,
            decoder_position_ids=pos.cuda(), return_dict=True)
    P = prompt.shape[1]
    loss = torch.nn.functional.cross_entropy(out.logits[:, P:].permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
    loss.backward()
    torch.manual_seed(23)
    enc_seed = torch.full((1,), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), dtype=torch.int32, device="cuda")
    dec_seed = torch.full((1,), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), dtype=torch.int32, device="cuda")
    B, T = inp.shape
    S = x.shape[1] * cfg.encoder.tokens_per_image
    dropout, paths = _hash_masks(m, cfg, B, T, S, enc_seed, dec_seed, x.shape[0] * x.shape[1])
    assert (0, "lora_q") in dropout
    names = [str(n) for n in g["grad_names"]]
    sd2 = {k: v.clone() for k, v in sd.items()}
    leaves = {n: sd2[n].requires_grad_(True) for n in names}
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, sd2, cfg.encoder, bn_train=True, bn_momentum=cfg.encoder.bn_momentum, drop_path=paths)
    ref = obert.decoder_forward(inp, sd2, cfg.decoder, h, emask, am, tt, pos, dropout=dropout)
    rloss = ogen.tf_cross_entropy(ref[:, P:], lab, gu.PAD)
    rloss.backward()
    check_act(out.logits.detach().float().cpu().numpy(), ref.detach().numpy(), "longitudinal train-mode logits")
    assert abs(loss.item() - rloss.item()) < 2e-2
    grads = _grads_by_name(m, names)
    for n in names:
        r = gu.rel_rms(grads[n].numpy(), leaves[n].grad.numpy())
        assert r < GRAD_RMS, f"{n}: rel_rms {r:.4f}"
    # without lora_dropout the branch would be merged: the logits must differ from the merged (eval-LoRA) computation with the same masks
    dropout_nolora = {k: v for k, v in dropout.items() if not (isinstance(k, tuple) and str(k[1]).startswith("lora"))}
    with torch.no_grad():
        merged = obert.decoder_forward(inp, sd2, cfg.decoder, h, emask, am, tt, pos, dropout=dropout_nolora)
    assert gu.rel_rms(merged.numpy(), ref.detach().numpy()) > 1e-4


def test_tf_longitudinal_lora_prompt(M):
    g, cfg, sd, x, prompt, inp, lab, am, tt, pos = gu.tf_longitudinal_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)                                   # SCST unfreezes the decoder (scst/gt_prompt.py:38-40)
    # teacher forcing: the CALLER derives position ids from its shifted mask (longitudinal/gt_prompt.py:198-208)
    pos_dev = torch.nn.functional.relu(torch.cumsum(am, dim=1) - 1).cuda()
    assert np.array_equal(pos_dev.cpu().numpy(), g["position_ids"])
    tt_dev = m.token_ids_to_token_type_ids(inp, [gu.PMT_SEP, gu.BOS, gu.SEP], [0, 1, 0, 1])
    assert np.array_equal(tt_dev.cpu().numpy(), g["token_type_ids"])
    eo = m.encoder(x.cuda())
    assert np.array_equal(eo.attention_mask.cpu().numpy(), g["enc_mask"])
    out = m(encoder_outputs=eo, decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt_dev,
            decoder_position_ids=pos_dev, return_dict=True)
    check_act(gu.sample(out.logits, 16384), g["logits_sample"], "longitudinal logits")
    loss = torch.nn.functional.cross_entropy(out.logits[:, prompt.shape[1]:].permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    grads = _grads_by_name(m, names)
    for i, n in enumerate(names):
        r = gu.rel_rms(gu.sample(grads[n], 2048), g[f"grad{i}_sample"])
        assert r < GRAD_RMS, f"{n}: rel_rms {r:.4f}"


def test_greedy_and_beam_multi(M):
    g, cfg, sd, x = gu.generate_multi_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    ref = torch.from_numpy(g["greedy"])
    L = ref.shape[1]
    # teacher-forced stepping along the reference's greedy sequence: cached argmax == reference argmax wherever the margin is safe
    out = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                     pad_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True, forced_tokens=ref[:, 1:])
    assert torch.equal(out["sequences"].cpu(), ref)
    safe = g["greedy_margin"] > MARGIN
    assert safe.mean() > 0.5
    assert np.array_equal(out["greedy_tokens"].cpu().numpy()[safe], g["greedy_argmax"][safe])
    np.testing.assert_allclose(out["greedy_margins"].cpu().numpy()[safe], g["greedy_margin"][safe], atol=0.05)
    # free-running greedy: identical up to the first unsafe position of each row
    free = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                      pad_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"].cpu()
    for b in range(ref.shape[0]):
        unsafe = np.nonzero(~safe[b])[0]
        upto = (unsafe[0] if len(unsafe) else L - 1) + 1
        assert torch.equal(free[b, :upto], ref[b, :upto]), (b, free[b], ref[b])
    # cache consistency: cached generate (decode kernels + hipGraph replay) == no-cache argmax loop through forward() of the SAME engine
    # (teacher-forced kernels), compared up to the first position whose top-1/top-2 margin is below the bf16 error bound
    eo = m.encoder(x.cuda())
    ids = torch.full((3, 1), gu.BOS, dtype=torch.int64, device="cuda")
    mg = []
    with torch.no_grad():
        for _ in range(L - 1):
            tt = m.token_ids_to_token_type_ids(ids, [gu.SEP])
            lg = m(encoder_outputs=eo, decoder_input_ids=ids, decoder_token_type_ids=tt).logits[:, -1]
            t2 = torch.topk(lg, 2, dim=-1)[0]
            mg.append((t2[:, 0] - t2[:, 1]).cpu())
            ids = torch.cat([ids, lg.argmax(-1, keepdim=True)], 1)
    mg = torch.stack(mg, 1).numpy()
    runs = {}
    for graph in (True, False):
        m.graph_decode = graph
        free_noeos = runs[graph] = m.generate(encoder_outputs=eo, special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=None,
                                pad_token_id=gu.PAD, num_beams=1, use_cache=True).cpu()
        for b in range(3):
            unsafe = np.nonzero(mg[b] < MARGIN)[0]
            upto = (unsafe[0] if len(unsafe) else L - 1) + 1
            assert torch.equal(free_noeos[b, :upto], ids[b, :upto].cpu()), (graph, b, free_noeos[b], ids[b])
    m.graph_decode = True
    again = m.generate(encoder_outputs=eo, special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=None,
                       pad_token_id=gu.PAD, num_beams=1, use_cache=True).cpu()
    assert torch.equal(again, runs[True])                         # a second replay of the captured steps reproduces the first bit for bit
    assert torch.equal(runs[True], runs[False])                   # eager launches and graph replay run the same kernels
    # beam-4 returns a well-formed result; equal to the reference when every decision on its path is safe
    beam = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                      pad_token_id=gu.PAD, num_beams=4, return_dict_in_generate=True, use_cache=True, output_scores=True)
    bs = beam["sequences"].cpu()
    assert bs.shape[0] == 3 and bool((bs[:, 0] == gu.BOS).all()) and bs.shape[1] <= L
    np.testing.assert_allclose(beam["sequences_scores"].cpu().numpy(), g["beam4_scores"], atol=0.05)
    # ... and the reference's beam-4 SEQUENCES. With random-init weights the four final hypotheses lie within 0.005-0.013 of each other
    # (generate_multi_beams.npz: all four beams and scores), far inside the bf16 score error, so the winner may be any of them: the returned
    # sequence must BE one of the reference's final beams, with that beam's score.
    gb = gu.load("generate_multi_beams.npz")
    assert float(gb["beam4_margin"].max()) < MARGIN
    # A row that is none of them must still be a hypothesis the fp32 search ranks inside that cluster: its sequence is re-scored by the fp32 oracle
    # (teacher-forced log-probabilities / length, the beam score's definition) and may not fall below the reference's 4th-best final beam by more
    # than the cluster's own width; the score the device reports for it must be that fp32 score to the bf16 tolerance. EVERY row is checked.
    from oracle import bert as obert, cvt as ocvt, token_ops as otok
    with torch.no_grad():
        h_or, mask_or = ocvt.encoder_forward(x, sd, cfg.encoder)
    hits = 0
    for b in range(3):
        for j in range(4):
            rb = torch.from_numpy(gb["beam4_all"][b, j])
            Lb = min(bs.shape[1], rb.shape[0])
            if torch.equal(bs[b, :Lb], rb[:Lb]):
                assert abs(float(beam["sequences_scores"][b]) - float(gb["beam4_all_scores"][b, j])) < 0.05
                hits += 1
                break
        else:
            seq = bs[b:b + 1]
            is_eos = (seq[0, 1:] == gu.EOS).nonzero()
            n_gen = int(is_eos[0]) + 1 if len(is_eos) else seq.shape[1] - 1
            inp = seq[:, :n_gen]
            with torch.no_grad():
                tt = torch.from_numpy(otok.token_ids_to_token_type_ids(inp.numpy(), [gu.SEP]))
                lp = torch.log_softmax(obert.decoder_forward(inp, sd, cfg.decoder, h_or[b:b + 1], mask_or[b:b + 1], None, tt, None).float(), -1)
            fp32_score = float(lp[0, torch.arange(n_gen), seq[0, 1:n_gen + 1]].sum()) / n_gen
            ref_sc = gb["beam4_all_scores"][b]
            assert fp32_score >= float(ref_sc.min()) - float(ref_sc.max() - ref_sc.min()) - 1e-3, (b, fp32_score, ref_sc)
            assert abs(float(beam["sequences_scores"][b]) - fp32_score) < 0.05, (b, float(beam["sequences_scores"][b]), fp32_score)
    assert hits >= 2, (hits, bs, gb["beam4_all"])
    # EOS handling (EOS -> PAD fill, stop/trim when every row has finished): bias the EOS logit well past the fixture's threshold
    with torch.no_grad():
        m.param("decoder.cls.predictions.bias")[gu.EOS] += float(g["eos_bias"]) + 1.0      # in-place edit: the bf16 shadow follows by itself
    eos_seq = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                         pad_token_id=gu.PAD, num_beams=1, use_cache=True).cpu()
    is_eos = eos_seq == gu.EOS
    assert bool(is_eos.any(1).all()) and eos_seq.shape[1] < L
    first = is_eos.int().argmax(1)
    assert eos_seq.shape[1] == int(first.max()) + 1               # HF stops on the step the last row finishes
    for b in range(3):
        assert bool((eos_seq[b, first[b] + 1:] == gu.PAD).all())
        assert torch.equal(eos_seq[b, :4], torch.from_numpy(g["greedy_eos"])[0, :4]) or first[b] < 3
    beam_e = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                        pad_token_id=gu.PAD, num_beams=4, use_cache=True).cpu()
    assert bool(((beam_e == gu.EOS).sum(1) == 1).all())


def test_prompted_generate_and_scst_scores(M):
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    ref = torch.from_numpy(g["greedy"])
    P = prompt.shape[1]
    new = ref.shape[1] - P
    eo = m.encoder(x.cuda())
    out = m.generate(encoder_outputs=eo, decoder_input_ids=prompt.cuda(), special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP],
                     max_length=new + 1 + P, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD,
                     num_beams=1, return_dict_in_generate=True, use_cache=True, forced_tokens=ref[:, P:])
    seq = out["sequences"]
    assert bool(torch.all(seq[:, 0] == gu.BOS))                  # prepended BOS, callers strip it (scst/gt_prompt.py:117-118)
    assert torch.equal(seq[:, 1:].cpu(), ref)
    safe = g["greedy_margin"] > MARGIN
    assert np.array_equal(out["greedy_tokens"].cpu().numpy()[safe], g["greedy_argmax"][safe])
    # SCST sampling path: scores carry autograd, exactly top_k finite entries, REINFORCE loss matches the oracle on the same sampled ids
    torch.manual_seed(0)
    smp = m.generate.__wrapped__(m, input_ids=prompt.cuda(), special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS,
                                 eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True,
                                 num_beams=1, use_cache=True, output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=new)
    if torch.all(smp["sequences"][:, 0] == 1):
        smp["sequences"] = smp["sequences"][:, 1:]               # item assignment, as the reference caller does
    scores = torch.stack(smp["scores"], dim=-1)                   # [B, V, T]
    sampled = smp["sequences"][:, P:]
    assert scores.requires_grad and scores.shape[0] == 2 and scores.shape[2] == sampled.shape[1]
    finite = torch.isfinite(scores).sum(1)
    assert bool((finite >= 50).all()) and bool((finite <= 52).all())
    assert bool(torch.isfinite(torch.gather(scores, 1, sampled[:, None, :])).all())     # every sampled id lies inside its step's top-k set
    # nucleus sampling on top of top-k (scst_sample_top_p < 1): processed scores keep between 1 and 50 entries, sampled ids inside them
    torch.manual_seed(1)
    nuc = m.generate.__wrapped__(m, input_ids=prompt.cuda(), special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS,
                                 eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True,
                                 num_beams=1, use_cache=True, output_scores=True, top_p=0.6, top_k=50, temperature=1.0, max_new_tokens=new)
    nsc = torch.stack(nuc["scores"], dim=-1)
    nseq = nuc["sequences"][:, 1:] if torch.all(nuc["sequences"][:, 0] == 1) else nuc["sequences"]
    nfin = torch.isfinite(nsc).sum(1)
    assert bool((nfin >= 1).all()) and bool((nfin <= 52).all())
    assert bool(torch.isfinite(torch.gather(nsc, 1, nseq[:, P:][:, None, :]))[nseq[:, P:][:, None, :] != gu.PAD].all())
    reward = torch.tensor([0.37, -0.21], device="cuda")
    nll = torch.nn.functional.nll_loss(torch.log_softmax(scores, dim=1), sampled, ignore_index=gu.PAD, reduction="none")
    loss = (nll.sum(-1) * reward).mean()
    loss.backward()
    gq = m.param("decoder.base_model.model.bert.encoder.layer.0.attention.self.query.lora_A.default.weight").grad
    assert gq is not None and float(gq.abs().sum()) > 0
    # oracle on the same sampled sequence
    from oracle import bert as obert, cvt as ocvt, generate as ogen
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, sd, cfg.encoder)
        seqs = smp["sequences"].cpu()
        fed, am, tt, pos = ogen.step_inputs("longitudinal", seqs, [gu.BOS, gu.SEP], gu.PAD, gu.BOS)
        lg = obert.decoder_forward(fed, sd, cfg.decoder, h, emask, am, tt, pos)
        olg = lg[:, P - 1:-1].float()
        # a token sampled at the edge of the bf16 top-50 can fall just outside the fp32 top-50: the oracle's filter keeps its own 50 AND the sampled
        # token, so EVERY position is compared (the two filtered distributions then differ by at most that boundary entry)
        osc = ogen.top_k_filter(olg, 50, keep=seqs[:, P:]).permute(0, 2, 1)
        onll = torch.nn.functional.nll_loss(torch.log_softmax(osc, dim=1), seqs[:, P:], ignore_index=gu.PAD, reduction="none")
        # ... and such a token really is a boundary case: its fp32 logit lies within the bf16 logit error of the fp32 k-th largest
        kth = torch.topk(olg, 50)[0][..., -1]
        at = torch.gather(olg, 2, seqs[:, P:, None])[..., 0]
    real = seqs[:, P:] != gu.PAD
    assert bool(torch.isfinite(onll).all())
    assert bool((at[real] >= kth[real] - MARGIN).all()), (at - kth)[real].min()
    np.testing.assert_allclose(nll.detach().cpu()[real].numpy(), onll[real].numpy(), atol=0.08)
    # the processed scores themselves: the finite set of every step is the fp32 top-50 up to boundary entries
    ofin = torch.isfinite(ogen.top_k_filter(olg, 50)).permute(0, 2, 1)
    diff = (torch.isfinite(scores.detach().cpu()) != ofin).sum(1)
    assert int(diff.max()) <= 4, diff


def test_reference_caller_loss_calls_are_served_by_the_boundary_tensors(M):
    """What the reference's Lightning modules do with the model's outputs (single.py:467-469; scst/gt_prompt.py:189,230-235) gives the same numbers
    whether torch computes it on plain tensors or modelling.BoundaryTensor recognises the call -- loss values, and the gradients that reach the
    parameters."""
    from cxrmate_amd.modelling import BoundaryTensor
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    names = ["decoder.bert.encoder.layer.0.output.dense.weight", "decoder.cls.predictions.bias", "encoder.projection_head.projection.weight"]

    def tf_loss(plain):
        for p in m.parameters():
            p.grad = None
        logits = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt.cuda(), return_dict=True).logits
        assert isinstance(logits, BoundaryTensor) and logits.dtype == torch.float32
        if plain:
            logits = logits.as_subclass(torch.Tensor)
        loss = torch.nn.functional.cross_entropy(logits.permute([0, 2, 1]), lab.cuda(), ignore_index=gu.PAD)
        assert type(loss) is torch.Tensor
        loss.backward()
        return loss.item(), {n: m.param(n).grad.float().cpu().clone() for n in names}

    l0, g0 = tf_loss(True)
    l1, g1 = tf_loss(False)
    assert abs(l0 - l1) < 2e-3 * max(1.0, abs(l0)), (l0, l1)
    for n in names:
        assert gu.rel_rms(g1[n].numpy(), g0[n].numpy()) < 2e-2, n
    # everything else a caller might do with .logits behaves like a tensor and returns plain tensors
    with torch.no_grad():
        lg = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_token_type_ids=tt.cuda()).logits
    assert type(lg[:, -1]) is torch.Tensor and type(lg.argmax(-1)) is torch.Tensor and type(lg.permute(2, 0, 1)) is torch.Tensor
    assert type(torch.nn.functional.cross_entropy(lg.permute(0, 2, 1), lab.cuda(), reduction="sum")) is torch.Tensor      # not the recognised call: torch's own
    # SCST: stack of all score steps -> log_softmax(dim=1) -> nll_loss
    g2, cfg2, sd2, x2, prompt = gu.generate_longitudinal_case()
    m2 = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg2, seed=None)
    m2.load_state_dict(sd2)
    for p in m2.decoder.parameters():
        p.requires_grad_(True)
    eo = m2.encoder(x2.cuda())
    lname = "decoder.base_model.model.bert.encoder.layer.0.attention.self.query.lora_A.default.weight"

    def scst_loss(plain):
        for p in m2.parameters():
            p.grad = None
        torch.manual_seed(0)
        smp = m2.generate.__wrapped__(m2, input_ids=prompt.cuda(), special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS,
                                      eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True,
                                      num_beams=1, use_cache=True, output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=10)
        seqs = smp["sequences"][:, 1:] if torch.all(smp["sequences"][:, 0] == 1) else smp["sequences"]
        scores = smp["scores"]
        assert all(isinstance(s_, BoundaryTensor) for s_ in scores)
        if plain:
            scores = tuple(s_.as_subclass(torch.Tensor) for s_ in scores)
        logits = torch.stack(scores, dim=-1)
        assert logits.shape == (2, cfg2.decoder.vocab_size, len(scores))
        sampled = seqs[:, prompt.shape[1]:]
        nll = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(logits, dim=1), sampled, ignore_index=gu.PAD, reduction="none")
        loss = (nll.sum(-1) * torch.tensor([0.37, -0.21], device="cuda")).mean()
        loss.backward()
        return loss.item(), nll.detach().cpu(), m2.param(lname).grad.float().cpu().clone(), seqs.cpu()

    a = scst_loss(True)
    b = scst_loss(False)
    assert torch.equal(a[3], b[3])                                 # same seed -> same sampled ids
    assert bool(torch.isfinite(a[1]).all()) and bool(torch.isfinite(b[1]).all())
    torch.testing.assert_close(a[1], b[1], atol=1e-4, rtol=1e-4)
    assert abs(a[0] - b[0]) < 1e-4 * max(1.0, abs(a[0]))
    assert gu.rel_rms(b[2].numpy(), a[2].numpy()) < 5e-3           # (the served pair emits d(scores) in bf16, like the fused SCST step)
    # the log_softmax of the recognised stack is PENDING until nll_loss(reduction='none') consumes it; any other use computes it and is torch's result
    torch.manual_seed(0)
    smp = m2.generate.__wrapped__(m2, input_ids=prompt.cuda(), special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                                  pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True, num_beams=1, use_cache=True,
                                  output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=10)
    stack = torch.stack(smp["scores"], dim=-1)
    lsm = torch.nn.functional.log_softmax(stack, dim=1)
    want = torch.log_softmax(stack.as_subclass(torch.Tensor), dim=1)
    assert lsm.shape == want.shape and lsm.dtype == want.dtype and lsm.device == want.device and lsm.requires_grad
    assert type(lsm.exp()) is torch.Tensor and torch.equal(lsm.exp(), want.exp()) and torch.equal(lsm[:, 3], want[:, 3]) and torch.equal(lsm + 0.0, want)
    tgt = smp["sequences"][:, -len(smp["scores"]):]
    assert torch.allclose(torch.nn.functional.nll_loss(lsm, tgt, ignore_index=gu.PAD, reduction="mean"),
                          torch.nn.functional.nll_loss(want, tgt, ignore_index=gu.PAD, reduction="mean"))      # not the recognised reduction: torch's own
    # a partial / reordered stack is not the recognised call: torch's own stack
    part = torch.stack(m2.generate.__wrapped__(m2, input_ids=prompt.cuda(), special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS,
                                               eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True,
                                               output_scores=True, top_k=50, max_new_tokens=6)["scores"][1:], dim=-1)
    assert type(part) is torch.Tensor and part.is_contiguous()


def test_speculative_greedy_baseline_equals_the_separate_greedy_call(M):
    """The SCST caller's greedy `generate` right after its sampling `generate.__wrapped__` (reference scst/gt_prompt.py:84-118) is served from the rows
    the sampling call decoded along (the arguments are learned from the previous step's greedy call): the sequences are those of a separate greedy
    call, bit for bit (eval mode), and any change of arguments / prompt / encoder outputs / weights falls back to a real decode."""
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    P = prompt.shape[1]
    pr = prompt.cuda()

    def caller_step(eo, spec):
        m.speculative_baseline = spec
        torch.manual_seed(4)
        smp = m.generate.__wrapped__(m, input_ids=pr, special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                                     pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True, num_beams=1, use_cache=True,
                                     output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=11)
        had = getattr(m, "_spec_result", None) is not None
        base = m.generate(encoder_outputs=eo, decoder_input_ids=pr, special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP], max_length=12 + P, bos_token_id=gu.BOS,
                          eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"]
        return smp["sequences"].cpu(), base.cpu(), had

    with torch.no_grad():
        eo = m.encoder(x.cuda())
    s0, b0, had0 = caller_step(eo, False)
    assert not had0
    s1, b1, had1 = caller_step(eo, True)                          # first speculative-mode step: the pattern is known from the greedy call above
    s2, b2, had2 = caller_step(eo, True)
    assert had1 and had2
    assert torch.equal(b0, b1) and torch.equal(b0, b2)
    assert torch.equal(s1, s2) and s1.shape == s0.shape           # (the 2B-row decode draws its uniforms for 2B rows: the samples differ from the B-row call's, not their law)
    # a greedy call with other arguments is a real decode of ITS arguments
    other = m.generate(encoder_outputs=eo, decoder_input_ids=pr, special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP], max_length=8 + P, bos_token_id=gu.BOS,
                       eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"].cpu()
    assert other.shape[1] <= 8 + P + 1 and torch.equal(other[:, : other.shape[1]], b0[:, : other.shape[1]])
    # new encoder outputs between the two calls: the stale rows are not handed out
    m.speculative_baseline = True
    torch.manual_seed(4)
    m.generate.__wrapped__(m, input_ids=pr, special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD,
                           mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True, output_scores=True, top_k=50, max_new_tokens=11)
    x2 = x.clone(); x2[1] = x[0]
    with torch.no_grad():
        eo2 = m.encoder(x2.cuda())
    base2 = m.generate(encoder_outputs=eo2, decoder_input_ids=pr, special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP], max_length=12 + P, bos_token_id=gu.BOS,
                       eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"].cpu()
    m.speculative_baseline = False
    ref2 = m.generate(encoder_outputs=eo2, decoder_input_ids=pr, special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP], max_length=12 + P, bos_token_id=gu.BOS,
                      eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"].cpu()
    assert torch.equal(base2, ref2)


def test_train_mode_scst_step_matches_the_oracle_with_the_hashed_masks(M):
    """scst_step under model.train() -- the mode the reference's training_step runs in: frozen encoder with batch-statistics
    BatchNorm and DropPath, decoder dropout and LoRA dropout active in the sampling decode, in the greedy baseline and in the re-scoring pass.
    The CPU oracle gets the masks the kernels hash (cxr_dropout_mask with the step's seeds) and must reproduce the REINFORCE loss of the ids the
    step sampled (reference scst/gt_prompt.py:211-246)."""
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW
    from oracle import bert as obert, cvt as ocvt, generate as ogen
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    m.train()
    opt = FusedAdamW(m, lr=1e-3)

    def reward_fn(ids):
        return ((ids % 7).float().mean(1) / 7.0).to(torch.float32)

    special = dict(bos=gu.BOS, eos=gu.EOS, sep=gu.SEP, pad=gu.PAD, pmt_sep=gu.PMT_SEP)
    torch.manual_seed(5)
    out = scst_step(m, opt, reward_fn, x.cuda(), prompt.cuda(), None, special, decoder_max_len=10)
    torch.cuda.synchronize()
    assert out["dropout_seed"] is not None and out["encoder_seed"] is not None
    sampled = out["sampled"].cpu()
    P = prompt.shape[1]
    adv = (reward_fn(out["sampled"]) - reward_fn(out["baseline_ids"][:, P:])).cpu()
    seqs = torch.cat([prompt, sampled], 1)
    B, T = seqs.shape
    S = x.shape[1] * cfg.encoder.tokens_per_image
    dropout, paths = _hash_masks(m, cfg, B, T, S, out["encoder_seed"], out["dropout_seed"], x.shape[0] * x.shape[1])
    assert (0, "lora_q") in dropout and paths
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, {k: v.clone() for k, v in sd.items()}, cfg.encoder, bn_train=True, bn_momentum=cfg.encoder.bn_momentum, drop_path=paths)
        fed, am, tt, pos = ogen.step_inputs("longitudinal", seqs, [gu.BOS, gu.SEP], gu.PAD, gu.BOS)
        lg = obert.decoder_forward(fed, sd, cfg.decoder, h, emask, am, tt, pos, dropout=dropout)
        sc = ogen.top_k_filter(lg[:, P - 1:-1].float(), 50, keep=sampled).permute(0, 2, 1)
        nll = torch.nn.functional.nll_loss(torch.log_softmax(sc, 1), sampled, ignore_index=gu.PAD, reduction="none")
        # the same ids WITHOUT the masks: dropout 0.1 moves the loss by more than the tolerance below, i.e. the comparison does see the masks
        lg0 = obert.decoder_forward(fed, sd, cfg.decoder, h, emask, am, tt, pos)
        sc0 = ogen.top_k_filter(lg0[:, P - 1:-1].float(), 50, keep=sampled).permute(0, 2, 1)
        nll0 = torch.nn.functional.nll_loss(torch.log_softmax(sc0, 1), sampled, ignore_index=gu.PAD, reduction="none")
    assert bool(torch.isfinite(nll).all())
    oloss = (nll.sum(-1) * adv).mean()
    tol = 0.05 * max(1.0, abs(oloss.item()))
    assert abs(out["loss"].item() - oloss.item()) < tol, (out["loss"].item(), oloss.item())
    assert float((nll - nll0).abs().max()) > 0.05                 # (per-token: the masked and unmasked networks are different networks)


def test_train_mode_cached_decode_equals_teacher_forcing_with_same_seed(M):
    """Under model.train() (how the reference runs its SCST decodes) every cached step applies dropout keyed by
    (seed, site, sequence, ABSOLUTE position): a teacher-forced pass with the same seed reproduces the step logits (that is what makes
    sample-then-rescore equal to keeping the autograd graph of the sampling loop); another seed does not."""
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    m.train()
    dev = m.device
    ref = torch.from_numpy(g["greedy"]).cuda()                                  # any fixed token sequence [B, P + new]
    B, L = ref.shape
    P = prompt.shape[1]
    special = [gu.PMT_SEP, gu.BOS, gu.SEP]
    with torch.no_grad():
        eo = m.encoder(x.cuda())
        enc16, emask8 = eo.last_hidden_state.contiguous(), eo.attention_mask.to(torch.uint8).contiguous()
        seed = torch.full((1,), 4242, dtype=torch.int32, device=dev)
        cache = m._dec.new_cache(B, L, dev)
        step_logits = []
        for cur in range(P, L):                                                 # prefill on the prompt, then one token at a time
            fed = ref[:, :cur]
            new, mask, tt, pos = m._step_inputs(fed, special, gu.PAD, prefill=cache.len == 0)
            step_logits.append(m._dec.decode(cache, new.contiguous(), enc16, emask8, mask, tt.contiguous(), pos.contiguous(), seed=seed).float().cpu())
        tf_in = ref[:, :L - 1].contiguous()
        mask, pos = __import__("cxrmate_amd.ops", fromlist=["x"]).mask_position_ids(tf_in, gu.PAD)
        tt = m.token_ids_to_token_type_ids(tf_in, special, [0, 1, 0, 1])
        same, _ = m._dec.forward(tf_in, enc16, emask8, mask, tt, pos, seed=seed)
        other, _ = m._dec.forward(tf_in, enc16, emask8, mask, tt, pos, seed=seed + 1)
        m.eval()
        plain, _ = m._dec.forward(tf_in, enc16, emask8, mask, tt, pos)
    steps = torch.stack(step_logits, 1).numpy()                                 # [B, L-P, V]: logits after feeding positions P-1 .. L-2
    same, other, plain = (t[:, P - 1:].float().cpu().numpy() for t in (same, other, plain))
    r_same, r_other, r_plain = gu.rel_rms(steps, same), gu.rel_rms(steps, other), gu.rel_rms(steps, plain)
    assert r_same < ACT_RMS, r_same
    assert r_other > 4 * r_same and r_plain > 4 * r_same, (r_same, r_other, r_plain)


def test_reference_sampled_sequence_scores_fixture(M):
    """Processed-score statistics for the REFERENCE's own sampled ids (fixture) reproduced through the fused loss kernel."""
    from cxrmate_amd import ops
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    seqs = torch.from_numpy(g["sampled_sequences"]).cuda()
    P = prompt.shape[1]
    with torch.no_grad():
        eo = m.encoder(x.cuda())
        mask, pos = m.position_ids_from_mask_token(seqs, gu.PAD)
        tt = m.token_ids_to_token_type_ids(seqs, [gu.BOS, gu.SEP], [0, 1, 0, 1])
        lg = m(encoder_outputs=eo, decoder_input_ids=seqs, decoder_attention_mask=mask, decoder_token_type_ids=tt, decoder_position_ids=pos).logits
        sc = lg[:, P - 1:-1].contiguous()
        B, T, V = sc.shape
        thr = ops.topk_threshold(sc.view(-1, V), 50)
        sampled = seqs[:, P:].contiguous().view(-1)
        w = ops.ce_weights(sampled, gu.PAD, mode=1, reward=torch.from_numpy(g["reward"]).cuda(), T=T)
        loss, row_loss, _ = ops.softmax_ce(sc.view(-1, V), sampled, gu.PAD, w, thr=thr, need_grad=False)
    assert abs(loss.item() - float(g["reinforce_loss"])) < 0.05 * max(1.0, abs(float(g["reinforce_loss"])))
    np.testing.assert_allclose(row_loss.view(B, T).cpu().numpy(), g["nll"], atol=0.08)


# ------------------------------------------------------------------------------------------------ full-size fixtures (CvT-21, BERT-6, vocab 30000)
def test_full_depth_encoder_matches_reference_fixture(M):
    """CvT-21 at depth (1, 4, 16): the bf16 error of 21 layers measured against the reference's fp32 activations, stage by stage."""
    g, cfg, sd, x = gu.encoder_full_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    errs = []
    for i, h in enumerate(_stage_activations(m, x.cuda())):
        assert list(h.shape) == g[f"stage{i}_shape"].tolist()
        errs.append(gu.rel_rms(gu.sample(h, 16384), g[f"stage{i}_sample"]))
        check_act(gu.sample(h, 16384), g[f"stage{i}_sample"], f"full-depth stage {i}")
    with torch.no_grad():
        out = m.encoder(x.cuda())
    h = out.last_hidden_state.float().cpu()
    assert list(h.shape) == [2, 1152, 768] and np.array_equal(out.attention_mask.cpu().numpy(), g["attention_mask"])
    check_act(gu.sample(h, 32768), g["last_hidden_state_sample"], "full-depth encoder output")
    print("bf16 rel-rms per stage / output:", [round(e, 5) for e in errs], round(gu.rel_rms(gu.sample(h, 32768), g["last_hidden_state_sample"]), 5))


def test_full_size_tf_logits_loss_argmax(M):
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_full_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    tt_dev = m.token_ids_to_token_type_ids(inp, [gu.SEP])
    assert np.array_equal(tt_dev.cpu().numpy(), g["token_type_ids"])
    with torch.no_grad():
        logits = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt_dev).logits
        loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
    assert list(logits.shape) == [2, 256, 30000]
    check_act(gu.sample(logits, 65536), g["logits_sample"], "full-size logits")
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    safe = g["logits_margin"] > MARGIN
    assert safe.mean() > 0.3
    assert np.array_equal(logits.argmax(-1).cpu().numpy()[safe], g["logits_argmax"][safe])
    rows = torch.from_numpy(g["logits_rows"])
    np.testing.assert_allclose(logits[:, rows, :512].float().cpu().numpy(), g["logits_row_slices"].astype(np.float32), atol=0.08)


def _grad_report(names, grads, g, prefix, bound, bounds=None):
    """rel-rms of every gradient slice against the fixture; returns the table (printed with -s) and asserts the bound on each (bounds: per-name
    overrides)."""
    rows = []
    for i, n in enumerate(names):
        want = g[f"{prefix}{i}_sample"]
        got = gu.sample(grads[n], 4096)
        assert np.isfinite(got).all(), n
        rows.append((n, gu.rel_rms(got, want), gu.cosine(got, want), float(np.linalg.norm(got) / max(np.linalg.norm(want), 1e-30))))
    for n, r, c, ratio in rows:
        print(f"  {r:.4f} cos {c:.5f} norm-ratio {ratio:.4f}  {n}")
    bad = [(n, round(r, 4)) for n, r, _, _ in rows if not r < (bounds or {}).get(n, bound)]
    assert not bad, bad
    return rows


def test_full_size_tf_gradients(M):
    """The benchmark's training step (single.py:449-475: forward -> F.cross_entropy -> loss.backward()) at the size it is measured on -- CvT-21 @384,
    2 x 2 images, BERT-6, V = 30000, T = 256, eval-mode dropout -- against the REFERENCE's own gradients (tf_full.npz): 34 parameter slices spanning the
    patch-embedding convolutions, stage-1 depthwise taps + BatchNorm gamma, stage-1 / 2 / 3 attention linears (the 9216 x 2304 and 2304 x 576 attention
    backward shapes), MLPs, projection head, decoder self / cross attention, FFN, LayerNorm, embeddings (tied LM head); total norm within 2 %."""
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_full_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    tt_dev = m.token_ids_to_token_type_ids(inp, [gu.SEP])
    logits = m(pixel_values=x.cuda(), decoder_input_ids=inp.cuda(), decoder_attention_mask=am.cuda(), decoder_token_type_ids=tt_dev, return_dict=True).logits
    loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), lab.cuda(), ignore_index=gu.PAD)
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    assert len(names) >= 30
    _grad_report(names, _grads_by_name(m, names), g, "grad", GRAD_RMS)

    def norm(prefix):
        return float(torch.sqrt(sum((p.grad.double() ** 2).sum() for n, p in m.named_parameters() if p.grad is not None and n.startswith(prefix))))
    for key, prefix in (("grad_total_norm", ""), ("grad_encoder_norm", "encoder."), ("grad_decoder_norm", "decoder.")):
        assert abs(norm(prefix) / float(g[key]) - 1.0) < 0.02, (key, norm(prefix), float(g[key]))


def test_longitudinal_c5_scst_reinforce_gradients(M):
    """configs[4] shape (3 images per study, 128-token prompt, LoRA, every decoder parameter trainable: scst/gt_prompt.py:38-40): the REFERENCE's
    sampled ids are pushed through the grad-enabled generate body (generate.__wrapped__, :162-180) and the caller's own reinforce_loss (:211-246) +
    backward; kept sets, per-token nll, loss and 16 decoder gradient slices (LoRA adapters, base weights, embeddings) against longitudinal_c5.npz."""
    g, cfg, sd, x = gu.longitudinal_c5_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for p in m.encoder.parameters():
        p.requires_grad_(False)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    prompt = torch.from_numpy(g["prompt_ids"]).cuda()
    P = prompt.shape[1]
    ref_seq = torch.from_numpy(g["scst_sampled_sequences"])
    new = ref_seq.shape[1] - P
    with torch.no_grad():
        eo = m.encoder(x.cuda())
    smp = m.generate.__wrapped__(m, input_ids=prompt, special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                                 pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True, num_beams=1, use_cache=True,
                                 output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=new, forced_tokens=ref_seq[:, P:])
    if torch.all(smp["sequences"][:, 0] == 1):
        smp["sequences"] = smp["sequences"][:, 1:]
    assert torch.equal(smp["sequences"].cpu(), ref_seq)
    scores = torch.stack(smp["scores"], dim=-1)                   # [B, V, T]
    sampled = smp["sequences"][:, P:]
    assert scores.requires_grad and list(scores.shape) == [2, 30000, new]
    fin = torch.isfinite(scores).sum(1).cpu().numpy()
    assert (np.abs(fin - g["scst_finite_count"]) <= 2).all(), fin
    at = torch.gather(scores, 1, sampled[:, None, :])[:, 0]
    np.testing.assert_allclose(at.detach().float().cpu().numpy(), g["scst_scores_at_sampled"], atol=0.08)
    adv = torch.from_numpy(g["scst_advantage"]).cuda()
    nll = torch.nn.functional.nll_loss(torch.log_softmax(scores, dim=1), sampled, ignore_index=gu.PAD, reduction="none")
    np.testing.assert_allclose(nll.detach().cpu().numpy(), g["scst_nll"], atol=0.08)
    loss = (nll.sum(-1) * adv).mean()
    assert abs(loss.item() - float(g["scst_reinforce_loss"])) < 0.05 * max(1.0, abs(float(g["scst_reinforce_loss"])))
    loss.backward()
    names = [str(n) for n in g["scst_grad_names"]]
    # The tied word-embedding / LM-head matrix: its STRIDED sample is dominated by the ~2400 rows of kept-but-not-sampled top-50 entries, and which
    # entry holds the 50th place differs between bf16 and fp32 logits at ~2 entries per position (measured: those rows 17 % apart, the whole matrix
    # 2.6 %, scripts/r6/c5_wordemb_diag.py). The sample therefore gets a wider bound, and the rows that carry the norm (99 % of it: the tokens that
    # were fed or sampled, whole rows recorded in the fixture) are held to the usual one.
    W = "decoder.base_model.model.bert.embeddings.word_embeddings.weight"
    grads = _grads_by_name(m, names)
    _grad_report(names, grads, g, "scst_grad", GRAD_RMS, bounds={W: 0.12})
    wrows = grads[W][torch.from_numpy(g["scst_wordemb_rows"])].numpy()
    r = gu.rel_rms(wrows, g["scst_wordemb_row_grads"])
    assert r < GRAD_RMS, f"word-embedding rows of the fed / sampled tokens: rel_rms {r:.4f}"
    tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.decoder.parameters() if p.grad is not None)))
    assert abs(tot / float(g["scst_grad_total_norm"]) - 1.0) < 0.02, (tot, float(g["scst_grad_total_norm"]))
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in m.encoder.parameters())


def test_longitudinal_c5_three_images_128_token_prompt(M):
    """BASELINE.json configs[4] shape: 3 images per study (one zero-padded), 128-token previous-report prompt with interior PADs: teacher-forced
    logits / loss of the report and KV-cached greedy steps against the reference."""
    g, cfg, sd, x = gu.longitudinal_c5_case()
    m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    inp, am = torch.from_numpy(g["input_ids"]).cuda(), torch.from_numpy(g["attention_mask"]).cuda()
    prompt = torch.from_numpy(g["prompt_ids"])
    P = prompt.shape[1]
    special = [gu.PMT_SEP, gu.BOS, gu.SEP]
    tt = m.token_ids_to_token_type_ids(inp, special, [0, 1, 0, 1])
    assert np.array_equal(tt.cpu().numpy(), g["token_type_ids"])
    _, pos = m.position_ids_from_mask_token(inp, gu.PAD)
    assert np.array_equal(pos.cpu().numpy(), g["position_ids"])
    with torch.no_grad():
        eo = m.encoder(x.cuda())
        assert np.array_equal(eo.attention_mask.cpu().numpy(), g["enc_mask"]) and eo.last_hidden_state.shape[1] == 3 * 576
        check_act(gu.sample(eo.last_hidden_state.float(), 16384), g["enc_sample"], "C5 encoder output")
        logits = m(encoder_outputs=eo, decoder_input_ids=inp, decoder_attention_mask=am, decoder_token_type_ids=tt, decoder_position_ids=pos).logits
        lg = logits[:, P - 1:]
        lab = torch.from_numpy(g["full_ids"])[:, 1:].cuda()
        loss = torch.nn.functional.cross_entropy(lg.permute(0, 2, 1), lab, ignore_index=gu.PAD)
    check_act(gu.sample(lg, 65536), g["logits_sample"], "C5 logits")
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    safe = g["logits_margin"] > MARGIN
    assert np.array_equal(lg.argmax(-1).cpu().numpy()[safe], g["logits_argmax"][safe])
    # KV-cached greedy decode behind the 128-token prompt, teacher-forced along the reference's sequence
    ref = torch.from_numpy(g["greedy"])
    new = ref.shape[1] - P
    out = m.generate(encoder_outputs=eo, decoder_input_ids=prompt.cuda(), special_token_ids=special, max_length=new + 1 + P, bos_token_id=gu.BOS,
                     eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD, num_beams=1, return_dict_in_generate=True, use_cache=True,
                     forced_tokens=ref[:, P:])
    gsafe = g["greedy_margin"] > MARGIN
    assert np.array_equal(out["greedy_tokens"].cpu().numpy()[gsafe], g["greedy_argmax"][gsafe])
    free = m.generate(encoder_outputs=eo, decoder_input_ids=prompt.cuda(), special_token_ids=special, max_length=new + 1 + P, bos_token_id=gu.BOS,
                      eos_token_id=None, pad_token_id=gu.PAD, mask_token_id=gu.PAD, num_beams=1, use_cache=True)[:, 1:].cpu()
    for b in range(2):
        unsafe = np.nonzero(~gsafe[b])[0]
        upto = P + (unsafe[0] if len(unsafe) else new)
        assert torch.equal(free[b, :upto], ref[b, :upto]), (b, free[b, P:], ref[b, P:])


def _same_beams(a, b, pad):
    """Equal sequences after right-padding to one width."""
    L = max(a.shape[1], b.shape[1])
    pa = torch.full((a.shape[0], L), pad, dtype=a.dtype); pa[:, :a.shape[1]] = a
    pb = torch.full((b.shape[0], L), pad, dtype=b.dtype); pb[:, :b.shape[1]] = b
    return torch.equal(pa, pb)


@pytest.mark.parametrize("beams", [4, 2])
def test_device_beam_search_equals_host_loop(M, beams):
    """generate(num_beams>1): the device-side search (beam-major rows sharing the cross K/V, bookkeeping kernel, graph-replayed steps, polled stop)
    returns what the host-loop restatement of the library's `_beam_search` returns -- multi-image model and prompted longitudinal model, free
    running and with an EOS that ends hypotheses early, twice (the second run replays the captured graphs)."""
    g, cfg, sd, x = gu.generate_multi_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    L = 24
    eo = m.encoder(x.cuda())

    def run(model, device_side, **kw):
        model.device_beam_search = device_side
        o = model.generate(encoder_outputs=eo_, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD, num_beams=beams,
                           return_dict_in_generate=True, use_cache=True, output_scores=True, **kw)
        return o["sequences"].cpu(), o["sequences_scores"].cpu()

    eo_ = eo
    for bias in (0.0, float(g["eos_bias"]) - 1.0):
        with torch.no_grad():
            m.param("decoder.cls.predictions.bias")[gu.EOS] += bias
        kw = dict(special_token_ids=[gu.SEP], max_length=L)
        hs, hsc = run(m, False, **kw)
        for _ in range(2):
            ds, dsc = run(m, True, **kw)
            torch.testing.assert_close(dsc, hsc, atol=2e-2, rtol=0)
            assert _same_beams(ds, hs, gu.PAD), (bias, ds, hs)
    # prompted longitudinal model: BOS stripped per step, position ids from the mask, prompt of several tokens
    g2, cfg2, sd2, x2, prompt = gu.generate_longitudinal_case()
    m2 = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg2, seed=None)
    m2.load_state_dict(sd2)
    eo_ = m2.encoder(x2.cuda())
    kw = dict(decoder_input_ids=prompt.cuda(), special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP], max_length=prompt.shape[1] + 14, mask_token_id=gu.PAD)
    hs, hsc = run(m2, False, **kw)
    for _ in range(2):
        ds, dsc = run(m2, True, **kw)
        torch.testing.assert_close(dsc, hsc, atol=2e-2, rtol=0)
        assert _same_beams(ds, hs, gu.PAD), (ds, hs)


def test_device_beam_search_full_size_fused_steps(M):
    """BERT-base widths: the cached beam steps run on the fused decode-step kernels (decode activation layout, 4 beams per cross K/V stream in
    the attention kernel). Random-init hypotheses lie within the bf16 error of each other, so the comparison with the host loop (whose steps run
    the generic kernels) is on the scores, plus well-formedness and replay determinism."""
    from cxrmate_amd.config import EncoderDecoderConfig
    cfg = EncoderDecoderConfig()
    m = M.MultiCXREncoderDecoderModel(cfg, device="cuda", seed=3)
    m.eval()
    x = torch.randn(2, 2, 3, 384, 384, generator=torch.Generator().manual_seed(11)).cuda()
    eo = m.encoder(x)
    kw = dict(encoder_outputs=eo, special_token_ids=[gu.SEP], max_length=20, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD, num_beams=4,
              return_dict_in_generate=True, use_cache=True, output_scores=True)
    m.device_beam_search = False
    h = m.generate(**kw)
    m.device_beam_search = True
    d1 = m.generate(**kw)
    d2 = m.generate(**kw)
    assert torch.equal(d1["sequences"], d2["sequences"]) and torch.equal(d1["sequences_scores"], d2["sequences_scores"])
    torch.testing.assert_close(d1["sequences_scores"], h["sequences_scores"], atol=0.05, rtol=0)
    s = d1["sequences"]
    assert s.shape[0] == 2 and bool((s[:, 0] == gu.BOS).all()) and s.shape[1] <= 20 and int(s.max()) < cfg.decoder.vocab_size
    ses = [v for k, v in m._decode_sessions.items() if k[-1] == 4]
    assert ses and any(k[0] == "beam" for s_ in ses for (_, _, _, k) in s_.graphs)          # the steps were replayed from captured graphs


def test_single_image_model_generate_matches_the_reference(M):
    """BASELINE.json configs[0] (C1): `SingleCXREncoderDecoderModel.generate` (reference modelling_single.py:217-249: the cached step runs cross-attention
    WITHOUT an encoder mask; caller single.py:483-493,552-562). Fixture generate_single.npz: greedy with cache == no-cache in the reference, and a
    beam-4 search that survives bf16-sized logit noise in the generator -- so sequences AND scores must be reproduced on every row."""
    g, cfg, sd, x = gu.generate_single_case()
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    ref = torch.from_numpy(g["greedy"])
    L = ref.shape[1]
    kw = dict(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=L, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD,
              return_dict_in_generate=True, use_cache=True)
    # greedy: every step's top-1 / top-2 margin is far above the bf16 logit error (1 % of the logit spread): the whole sequence is pinned
    err = 0.012 * float(g["logit_std"])
    assert float(g["greedy_margin"].min()) > 5 * err
    out = m.generate(num_beams=1, forced_tokens=ref[:, 1:], **kw)
    assert np.array_equal(out["greedy_tokens"].cpu().numpy(), g["greedy_argmax"])
    np.testing.assert_allclose(out["greedy_margins"].cpu().numpy(), g["greedy_margin"], atol=4 * err)
    for graph in (True, False):
        m.graph_decode = graph
        free = m.generate(num_beams=1, **kw)["sequences"].cpu()
        assert torch.equal(free, ref), (graph, free, ref)
    m.graph_decode = True
    # no-cache argmax loop through forward() of the same engine (teacher-forcing kernels) == the cached decode == the reference
    eo = m.encoder(x.cuda())
    assert "attention_mask" not in eo or eo.get("attention_mask") is None      # single-image encoder outputs carry no mask (modelling_single.py:53-78)
    ids = torch.full((3, 1), gu.BOS, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        for _ in range(L - 1):
            lg = m(encoder_outputs=eo, decoder_input_ids=ids, decoder_token_type_ids=m.token_ids_to_token_type_ids(ids, [gu.SEP])).logits[:, -1]
            ids = torch.cat([ids, lg.argmax(-1, keepdim=True)], 1)
    assert torch.equal(ids.cpu(), ref)
    # the same loop with the engine's KV cache handed back and forth through forward(use_cache=True, past_key_values=...): a caller that owns its decoding
    # loop (what transformers' generate does through prepare_inputs_for_generation) -- new tokens only, last-position logits [B, 1, V]
    ids, past = torch.full((3, 1), gu.BOS, dtype=torch.int64, device="cuda"), None
    with torch.no_grad():
        for _ in range(L - 1):
            tt = m.token_ids_to_token_type_ids(ids, [gu.SEP])
            fed, tt_new = (ids, tt) if past is None else (ids[:, -1:], tt[:, -1:])
            out = m(encoder_outputs=eo, decoder_input_ids=fed, decoder_token_type_ids=tt_new, past_key_values=past, use_cache=True)
            assert out.logits.shape[:2] == (3, fed.shape[1])    # the library's contract: logits for every FED position
            past = out.past_key_values
            ids = torch.cat([ids, out.logits[:, -1].argmax(-1, keepdim=True)], 1)
    assert past.len == L - 1 and torch.equal(ids.cpu(), ref)
    # use_cache=True without a cache on a multi-token input (a teacher-forced evaluation that merely leaves the flag on, or a prompted prefill): logits
    # for ALL positions, equal to the plain call's, plus a cache holding them; one more token on top of it continues the sequence
    with torch.no_grad():
        tt = m.token_ids_to_token_type_ids(ids, [gu.SEP])
        plain = m(encoder_outputs=eo, decoder_input_ids=ids[:, :-1], decoder_token_type_ids=tt[:, :-1]).logits
        pre = m(encoder_outputs=eo, decoder_input_ids=ids[:, :-1], decoder_token_type_ids=tt[:, :-1], use_cache=True)
        assert pre.logits.shape == plain.shape and torch.equal(pre.logits, plain) and pre.past_key_values.len == ids.shape[1] - 1
        nxt = m(encoder_outputs=eo, decoder_input_ids=ids[:, -1:], decoder_token_type_ids=tt[:, -1:], past_key_values=pre.past_key_values, use_cache=True)
        full = m(encoder_outputs=eo, decoder_input_ids=ids, decoder_token_type_ids=tt).logits[:, -1]
        check_act(nxt.logits[:, 0].float().cpu().numpy(), full.float().cpu().numpy(), "cached step on a prefilled cache vs the teacher-forced pass")
        with pytest.raises(NotImplementedError):                 # several new tokens on top of a non-empty cache
            m(encoder_outputs=eo, decoder_input_ids=ids[:, -2:], decoder_token_type_ids=tt[:, -2:], past_key_values=nxt.past_key_values, use_cache=True)
    for p_ in m.decoder.parameters():
        p_.requires_grad_(True)
    tt = m.token_ids_to_token_type_ids(ids, [gu.SEP])
    tr = m(encoder_outputs=eo, decoder_input_ids=ids, decoder_token_type_ids=tt, use_cache=True)      # under autograd: a teacher-forced pass, no cache
    assert tr.past_key_values is None and tr.logits.requires_grad and tr.logits.shape[:2] == tuple(ids.shape)
    with pytest.raises(RuntimeError):                            # the cached kernels have no backward
        m(encoder_outputs=eo, decoder_input_ids=ids[:, -1:], decoder_token_type_ids=tt[:, -1:], past_key_values=past, use_cache=True)
    for p_ in m.decoder.parameters():
        p_.requires_grad_(False)
    # beam-4: device-side search and the host loop, both equal to the reference's best hypothesis and score on every row
    rb, rs = torch.from_numpy(g["beam4_all"][:, 0]), g["beam4_all_scores"]
    gap = rs[:, 0] - rs[:, 1]
    for device_side in (True, False):
        m.device_beam_search = device_side
        b = m.generate(num_beams=4, output_scores=True, **kw)
        bs = b["sequences"].cpu()
        assert torch.equal(bs, rb[:, :bs.shape[1]]) and bool((rb[:, bs.shape[1]:] == gu.PAD).all()), (device_side, bs, rb)
        assert bool((np.abs(b["sequences_scores"].cpu().numpy() - rs[:, 0]) < 0.4 * gap).all()), (b["sequences_scores"], rs[:, 0], gap)
    m.device_beam_search = True


def test_graph_capture_survives_a_garbage_collection_inside_it(M):
    """A cyclic-GC run that STARTS inside a hipGraph capture used to destroy the previous session's graphs there (HIP calls that are illegal while
    a stream captures in global mode: the process aborted, depending on how many Python objects earlier tests had allocated). ops.graph_capture holds
    the collector off; here every allocation would trigger a collection (threshold 1) while garbage with CUDA graphs is waiting."""
    import gc
    g, cfg, sd, x = gu.generate_single_case()
    ref = torch.from_numpy(g["greedy"])
    kw = dict(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=ref.shape[1], bos_token_id=gu.BOS, eos_token_id=gu.EOS,
              pad_token_id=gu.PAD, return_dict_in_generate=True, use_cache=True, num_beams=1)
    old = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    old.load_state_dict(sd)
    assert torch.equal(old.generate(**kw)["sequences"].cpu(), ref)          # its session holds captured step graphs
    cycle = [old]
    cycle.append(cycle)                                                         # reachable only through a reference cycle once the names are dropped
    del old, cycle
    m = M.SingleCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    thr = gc.get_threshold()
    gc.set_threshold(1)
    try:
        out = m.generate(**kw)["sequences"].cpu()
    finally:
        gc.set_threshold(*thr)
    assert torch.equal(out, ref)
    assert gc.isenabled()


@pytest.mark.parametrize("case", ["plain", "lp2", "lp05", "eos"])
def test_device_beam_search_equals_the_reference_on_every_row(M, case):
    """generate_beam_safe.npz: beam-4 decodes of the multi-image model whose every decision survives bf16-sized logit noise (checked by noise
    injection in the generator) and whose three studies each have their OWN best hypothesis: plain, an EOS that ends the best hypothesis of some
    rows early, and length_penalty 2.0 / 0.5 ranking hypotheses of different length. The device-side search and the host loop must both return the
    reference's best hypothesis for EVERY study -- the three sequences are pairwise different, so a cross-study mix-up in the beam bookkeeping or the
    cache reorder cannot reproduce them -- with its score to a tolerance below the gap to the row's runner-up."""
    c = gu.beam_safe_case(case)
    assert c is not None, case
    cfg, sd, x, eos_bias, lp, ref_all, ref_scores, steps, tol = c
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    if eos_bias:
        with torch.no_grad():
            m.param("decoder.cls.predictions.bias")[gu.EOS] += eos_bias
    rb = torch.from_numpy(ref_all[:, 0])
    assert len({tuple(r.tolist()) for r in rb}) == rb.shape[0]
    for device_side in (True, False, True):                      # (the second device-side run replays the captured step graphs)
        m.device_beam_search = device_side
        o = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=steps + 1, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                       pad_token_id=gu.PAD, num_beams=4, length_penalty=lp, return_dict_in_generate=True, use_cache=True, output_scores=True)
        bs = o["sequences"].cpu()
        assert torch.equal(bs, rb[:, :bs.shape[1]]) and bool((rb[:, bs.shape[1]:] == gu.PAD).all()), (case, device_side, bs, rb)
        assert bool((np.abs(o["sequences_scores"].cpu().numpy() - ref_scores[:, 0]) < tol).all()), (case, o["sequences_scores"], ref_scores[:, 0], tol)
    if case == "eos":
        assert bool((rb == gu.EOS).any(1).any()) and not bool((rb == gu.EOS).any(1).all())


def test_device_beam_search_c3_size_is_replay_deterministic_and_agrees_with_the_host_loop(M):
    """BASELINE.json configs[2] at full size: 8 studies x 2 images x beam 4 x 256 tokens (EOS disabled: all 255 steps run). Size-independent
    properties: the device-side search is bit-reproducible across graph replays, every returned row is a well-formed sequence, and its scores agree
    with the host loop's (random-init hypotheses lie within the bf16 score error of each other, so the winner itself is not pinned at this size)."""
    from cxrmate_amd.config import EncoderDecoderConfig
    cfg = EncoderDecoderConfig()
    m = M.MultiCXREncoderDecoderModel(cfg, device="cuda", seed=5)
    m.eval()
    x = torch.randn(8, 2, 3, 384, 384, generator=torch.Generator().manual_seed(12)).cuda()
    x[3, 1] = 0.0
    eo = m.encoder(x)
    kw = dict(encoder_outputs=eo, special_token_ids=[gu.SEP], max_length=256, bos_token_id=gu.BOS, eos_token_id=None, pad_token_id=gu.PAD, num_beams=4,
              return_dict_in_generate=True, use_cache=True, output_scores=True)
    d1 = m.generate(**kw)
    d2 = m.generate(**kw)
    assert torch.equal(d1["sequences"], d2["sequences"]) and torch.equal(d1["sequences_scores"], d2["sequences_scores"])
    s = d1["sequences"]
    assert s.shape == (8, 256) and bool((s[:, 0] == gu.BOS).all()) and int(s.max()) < cfg.decoder.vocab_size and int(s.min()) >= 0
    m.device_beam_search = False
    h = m.generate(**kw)
    m.device_beam_search = True
    torch.testing.assert_close(d1["sequences_scores"], h["sequences_scores"], atol=0.05, rtol=0)
    # the score of a returned hypothesis is the length-normalised sum of its own token log-probabilities under teacher forcing
    tt = m.token_ids_to_token_type_ids(s[:, :-1], [gu.SEP])
    with torch.no_grad():
        lp = torch.log_softmax(m(encoder_outputs=eo, decoder_input_ids=s[:, :-1], decoder_token_type_ids=tt).logits.float(), -1)
    tok = lp.gather(2, s[:, 1:, None])[..., 0].sum(1) / 256.0
    torch.testing.assert_close(d1["sequences_scores"].float(), tok, atol=0.05, rtol=0)


FP8_RMS = 0.05          # stated tolerance of the e4m3 encoder (per-tensor scales, 126 quantised GEMMs in sequence) against the reference's fp32 output


def test_fp8_encoder_against_the_fp32_reference_fixture(M):
    """BASELINE.json configs[4] encoder: CvT-21 with every Linear of the 21 blocks as an e4m3 (OCP) MFMA GEMM, per-tensor weight scales and static
    per-tensor activation scales calibrated on OTHER images, against the reference's fp32 activations (tests/golden/encoder_full.npz)."""
    g, cfg, sd, x = gu.encoder_full_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        h16 = m.encoder(x.cuda()).last_hidden_state.float().cpu()
        calib = torch.randn(2, 2, 3, 384, 384, generator=torch.Generator().manual_seed(777))       # not the test images
        m.enable_fp8_encoder(calib)
        assert m._enc.fp8 is not None and len(m._enc.fp8["w"]) == 21 * 6
        out = m.encoder(x.cuda())
        h8 = out.last_hidden_state.float().cpu()
    assert np.array_equal(out.attention_mask.cpu().numpy(), g["attention_mask"])
    ref = g["last_hidden_state_sample"]
    r8, r16 = gu.rel_rms(gu.sample(h8, 32768), ref), gu.rel_rms(gu.sample(h16, 32768), ref)
    c8 = gu.cosine(gu.sample(h8, 32768), ref)
    print(f"encoder output rel-rms vs fp32 reference: e4m3 {r8:.4f} (cosine {c8:.5f}), bf16 {r16:.4f}")
    assert np.isfinite(h8.numpy()).all() and r8 < FP8_RMS and c8 > 0.995
    assert not torch.equal(h8, h16)                                     # the e4m3 GEMMs really ran
    # the decoder on top of it: teacher-forced logits stay close to the bf16-encoder logits (what the frozen-encoder SCST consumes)
    m.disable_fp8_encoder()
    with torch.no_grad():
        again = m.encoder(x.cuda()).last_hidden_state.float().cpu()
    assert torch.equal(again, h16)
    # trainable encoder weights that moved are re-quantised by the next forward; load_state_dict drops the fp8 state (back to bf16)
    m.enable_fp8_encoder(calib)
    wname = "encoder.cvt.encoder.stages.2.layers.0.output.dense.weight"
    m.param(wname).requires_grad_(True)
    before = m._enc.fp8["w"][wname][0].clone()
    with torch.no_grad():
        m.param(wname).mul_(1.5)
        m.encoder(x.cuda())
    assert not torch.equal(before.view(torch.uint8), m._enc.fp8["w"][wname][0].view(torch.uint8)) or abs(m._enc.fp8["w"][wname][1]) > 0
    m.load_state_dict(sd)
    assert m._enc.fp8 is None


def test_forward_with_labels_matches_torch_cross_entropy(M):
    """forward(labels=...) (reference modelling_single.py:205-210: CrossEntropyLoss() over the flattened logits, ignore_index -100): the loss and the
    gradient that reaches the parameters equal torch's cross entropy applied to the same logits."""
    g, cfg, sd, x = gu.generate_multi_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    for p in m.parameters():
        p.requires_grad_(True)
    torch.manual_seed(0)
    ids = torch.randint(5, cfg.decoder.vocab_size, (3, 9), device="cuda")
    labels = ids.roll(-1, 1).clone()
    labels[:, -1] = -100
    labels[1, 3:] = -100
    tt = m.token_ids_to_token_type_ids(ids, [gu.SEP])
    out = m(pixel_values=x.cuda(), decoder_input_ids=ids, decoder_token_type_ids=tt, labels=labels)
    ref = torch.nn.functional.cross_entropy(out.logits.detach().float().reshape(-1, cfg.decoder.vocab_size), labels.reshape(-1))
    assert abs(float(out.loss.detach()) - float(ref)) < 2e-3
    name = "decoder.cls.predictions.bias"
    m.zero_grad()
    out.loss.backward()
    g1 = m.param(name).grad.detach().clone()
    m.zero_grad()
    out2 = m(pixel_values=x.cuda(), decoder_input_ids=ids, decoder_token_type_ids=tt)
    torch.nn.functional.cross_entropy(out2.logits.float().reshape(-1, cfg.decoder.vocab_size), labels.reshape(-1)).backward()
    g2 = m.param(name).grad.detach()
    assert gu.rel_rms(g1.cpu().numpy(), g2.cpu().numpy()) < 2e-2


def test_save_pretrained_from_pretrained_round_trip(M, tmp_path):
    """HF directory layout (config.json + model.safetensors with the reference's key names): a saved model comes back through from_pretrained with
    identical logits -- also when the tied LM-projection keys are missing from the file (transformers writes tied tensors once) -- and a model id
    that is not a local directory is refused (no Hub access)."""
    g, cfg, sd, x = gu.generate_multi_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    ids = torch.randint(5, cfg.decoder.vocab_size, (3, 7), device="cuda")
    tt = m.token_ids_to_token_type_ids(ids, [gu.SEP])
    with torch.no_grad():
        ref = m(pixel_values=x.cuda(), decoder_input_ids=ids, decoder_token_type_ids=tt).logits
    d = str(tmp_path / "ckpt")
    m.save_pretrained(d)
    from safetensors.torch import load_file, save_file
    import os
    f = os.path.join(d, "model.safetensors")
    st = {k: v for k, v in load_file(f).items() if "cls.predictions.decoder" not in k}
    save_file(st, f)
    m2 = M.MultiCXREncoderDecoderModel.from_pretrained(d)
    assert not m2.training and m2.config.encoder.depth == cfg.encoder.depth and m2.config.decoder.vocab_size == cfg.decoder.vocab_size
    with torch.no_grad():
        out = m2(pixel_values=x.cuda(), decoder_input_ids=ids, decoder_token_type_ids=tt).logits
    assert torch.equal(out, ref)
    with pytest.raises(OSError):
        M.MultiCXREncoderDecoderModel.from_pretrained("aehrc/cxrmate-multi-tf")


def test_beam_counts_without_a_shared_kv_kernel_fall_back_to_the_host_loop(M):
    """The device-side search shares a study's cross K/V between its beams through the 2- and 4-row attention kernels; other beam counts run the
    host-loop restatement (same decode kernels, replicated K/V) and return a well-formed result."""
    g, cfg, sd, x = gu.generate_multi_case()
    m = M.MultiCXREncoderDecoderModel(cfg, seed=None)
    m.load_state_dict(sd)
    out = m.generate(pixel_values=x.cuda(), special_token_ids=[gu.SEP], max_length=14, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD,
                     num_beams=3, return_dict_in_generate=True, use_cache=True, output_scores=True)
    s = out["sequences"].cpu()
    assert s.shape[0] == 3 and bool((s[:, 0] == gu.BOS).all()) and s.shape[1] <= 14 and bool(torch.isfinite(out["sequences_scores"]).all())


# ------------------------------------------------------------------------------------------------ the A/B switches that ship
# Every environment switch of the package / the library selects a path that is NOT the default one. They are read once per process, so each group runs
# the reference-fixture parity tests in a CHILD process with the switches set (round-5 review: nothing that ships may go unexecuted by the suite).
_SWITCH_GROUPS = {
    "decoder fallbacks": (dict(CXR_CROSS_KV_FUSED="0", CXR_LORA_IN_KERNEL="0", CXR_CROSS_KV_SHARED="0", CXR_LORA_MULTI="0", CXR_SELF_QKV_FUSED="0", CXR_CROSS_Q_FUSED="0"),
                          "tf_single_logits_loss_grads or tf_longitudinal_lora_prompt or greedy_and_beam_multi or prompted_generate_and_scst_scores or tf_longitudinal_train_mode"),
    "VALU cross-attention": (dict(CXR_CROSS_MFMA="0", CXR_CROSS_WG_KEYS="288"), "greedy_and_beam_multi or prompted_generate_and_scst_scores"),
    "encoder fallbacks": (dict(CXR_EARLY_PATCH_COL="0", CXR_DWPROJ="0", CXR_DW3_STATS_FROM_Y="0", CXR_IMPLICIT_EMBED="0", CXR_PATCH_EMBED_FUSED="0", CXR_FP8_FUSED="0"),
                          "encoder_matches_reference_fixture or tf_single_logits_loss_grads or tf_train_mode_matches_oracle or fp8_encoder_against"),
    "library kernels": (dict(CXR_TN2="0", CXR_TN_STAGES="2", CXR_TN5="0", CXR_STRIP_GROUP="0", CXR_LN_BWD_PF="0", CXR_GEMM_WS="0", CXR_IM2COL_ROWS="0", CXR_CE_BF16ROW="0", CXR_GEMM_LDS_EPILOGUE="0"),
                        "encoder_matches_reference_fixture or tf_single_logits_loss_grads or tf_train_mode_matches_oracle or forward_with_labels"),
    "library kernels 2": (dict(CXR_TN2_MIN="1", CXR_TN_WGS="64", CXR_TN2_WGS="48", CXR_LN_BWD_GRID="128", CXR_GEMM_BK="32", CXR_GEMM_STAGES="3", CXR_DW3_BAND="4", CXR_TAIL_ON_MAIN="0"),
                          "tf_single_logits_loss_grads or tf_train_mode_matches_oracle"),
    "training-step schedule": (dict(CXR_ZERO_ON_SIDE="0", CXR_EARLY_DEC_ADAMW="0", CXR_EARLY_ENC_ADAMW="0", CXR_BF16_LOGITS="0", CXR_WGRAD_OVERLAP="0", CXR_BIND_GRADS="0"),
                               "tf_single_logits_loss_grads or torch_optimizer_updates or graphed_tf_step or training_step_gradients_do_not_depend"),
    # (test_fused_adamw_behind_the_torch_optimizer_interface asserts the DEFAULT path's launch plan, which CXR_BIND_GRADS=0 replaces by design)
}


@pytest.mark.parametrize("group", list(_SWITCH_GROUPS))
def test_parity_tests_under_the_shipped_ab_switches(M, group):
    import os
    import subprocess
    import sys
    env_add, expr = _SWITCH_GROUPS[group]
    if os.environ.get("CXR_SWITCH_CHILD") == "1":
        pytest.skip("child of this test")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, CXR_SWITCH_CHILD="1", **env_add)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_model_gpu.py"), "-x", "-q", "-m", "gpu", "-k", expr, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(here))
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, f"{group}: {env_add}\n{tail}"
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
