"""CPU: the oracle (oracle/) must reproduce what the REFERENCE produced for the same seeded inputs
(fixtures written by tests/golden/make_golden.py, which imports /root/reference in the build container)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import bert as obert
from oracle import cvt as ocvt
from oracle import generate as ogen
from oracle import token_ops

FP32_TOL = 2e-4   # oracle and reference are both fp32 on CPU; only summation order differs


def test_token_type_ids_random_and_documented():
    g = gu.load("token_ops.json")
    for case in g["random"] + g["documented"]:
        tt = token_ops.token_ids_to_token_type_ids(case["ids"], case["special"], case["sections"])
        ttp = token_ops.token_ids_to_token_type_ids_past(case["ids"], case["special"], case["sections"])
        assert tt.tolist() == case["token_type_ids"], case
        assert ttp.tolist() == case["token_type_ids_past"], case


def test_documented_quirks():
    # SURVEY.md Q4 / Q5 / A.5
    assert token_ops.token_ids_to_token_type_ids([[1, 50, 51, 3, 60, 61, 2]], [3]).tolist() == [[0, 0, 0, 0, 1, 1, 1]]
    assert token_ops.token_ids_to_token_type_ids([[1, 50, 51, 60, 61, 2, 3]], [3]).tolist() == [[0] * 7]
    ids = [[8, 50, 9, 60, 1, 70, 71, 3, 80, 2, 4]]
    assert token_ops.token_ids_to_token_type_ids(ids, [9, 1, 3], [0, 1, 0, 1]).tolist() == [[0, 0, 0, 1, 1, 0, 0, 0, 1, 1, 1]]
    assert token_ops.token_ids_to_token_type_ids(ids, [1, 3], [0, 1, 0, 1]).tolist() == [[0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0]]
    assert token_ops.token_ids_to_token_type_ids_past([[1, 50, 3, 60]], [3]).tolist() == [[1]]
    assert token_ops.token_ids_to_token_type_ids_past([[1, 50, 60, 3]], [3]).tolist() == [[0]]
    assert token_ops.position_ids_from_mask([[1, 1, 0, 0, 1, 1]]).tolist() == [[0, 1, 1, 1, 2, 3]]
    assert token_ops.position_ids_from_mask([[0, 0, 1]]).tolist() == [[0, 0, 0]]


def test_encoder_matches_reference():
    g, cfg, sd, x = gu.encoder_case()
    with torch.no_grad():
        h, mask, stages = ocvt.encoder_forward(x, sd, cfg.encoder, return_stages=True)
    assert list(h.shape) == g["last_hidden_state_shape"].tolist()
    assert np.array_equal(mask.numpy(), g["attention_mask"])
    for i, s in enumerate(stages):
        assert list(s.shape) == g[f"stage{i}_shape"].tolist()
        assert gu.rel_rms(gu.sample(s), g[f"stage{i}_sample"]) < FP32_TOL, i
    assert gu.rel_rms(gu.sample(h, 16384), g["last_hidden_state_sample"]) < FP32_TOL
    np.testing.assert_allclose(gu.stats(h), g["last_hidden_state_stats"], rtol=1e-4)


def _tf_single_logits(sd, cfg, x, inp, am, tt):
    h, _ = ocvt.encoder_forward(x, sd, cfg.encoder)
    return obert.decoder_forward(inp, sd, cfg.decoder, h, None, am, tt, None)


def test_tf_single_logits_loss_grads():
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_single_case()
    assert np.array_equal(token_ops.token_ids_to_token_type_ids(inp.numpy(), [gu.SEP]), g["token_type_ids"])
    names = [str(n) for n in g["grad_names"]]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)
    logits = _tf_single_logits(sd2, cfg, x, inp, am, tt)
    loss = ogen.tf_cross_entropy(logits, lab, gu.PAD)
    assert gu.rel_rms(gu.sample(logits, 16384), g["logits_sample"]) < FP32_TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    # argmax agrees wherever the reference margin is not numerically degenerate
    safe = g["logits_margin"] > 1e-3
    assert np.array_equal(logits.argmax(-1).numpy()[safe], g["logits_argmax"][safe])
    loss.backward()
    for i, n in enumerate(names):
        gr = leaves[n].grad
        if n.endswith("word_embeddings.weight"):
            pass  # tied with the LM projection: both contributions flow into the same leaf here as in the reference
        assert gu.rel_rms(gu.sample(gr, 2048), g[f"grad{i}_sample"]) < 2e-3, n
        np.testing.assert_allclose(gu.stats(gr)[3], g[f"grad{i}_stats"][3], rtol=2e-3, err_msg=n)


def test_tf_single_train_mode_with_recorded_masks():
    """model.train(): batch-statistics BatchNorm (+ running statistics), dropout on hidden states / attention probabilities, DropPath.
    The reference's drawn masks are inputs of the oracle, so the whole stochastic pass is pinned: logits, loss, gradients, running stats."""
    g, cfg, sd, x, inp, lab, am, tt, dropout, paths = gu.tf_single_train_case()
    assert any(bool((f == 0).any()) for pair in paths.values() for f in pair)          # a residual branch really was dropped
    names = [str(n) for n in g["grad_names"]]
    sd2 = {k: v.clone() for k, v in sd.items()}
    leaves = {n: sd2[n].requires_grad_(True) for n in names}
    h, _ = ocvt.encoder_forward(x, sd2, cfg.encoder, bn_train=True, bn_momentum=cfg.encoder.bn_momentum, drop_path=paths)
    logits = obert.decoder_forward(inp, sd2, cfg.decoder, h, None, am, tt, None, dropout=dropout)
    loss = ogen.tf_cross_entropy(logits, lab, gu.PAD)
    assert gu.rel_rms(gu.sample(logits, 16384), g["logits_sample"]) < FP32_TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for i, n in enumerate(names):
        assert gu.rel_rms(gu.sample(leaves[n].grad, 2048), g[f"grad{i}_sample"]) < 2e-3, n
    for i in range(3):
        key = str(g[f"bn{i}_key"])
        np.testing.assert_allclose(sd2[key + "running_mean"].numpy(), g[f"bn{i}_running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd2[key + "running_var"].numpy(), g[f"bn{i}_running_var"], rtol=1e-4, atol=1e-6)
        assert int(sd2[key + "num_batches_tracked"]) == int(g[f"bn{i}_num_batches_tracked"]) == 1
    # the DropPath rate every stage-3 layer uses is linspace(0, 0.1, depth)[stage] (quirk Q2), recorded by the reference's modules
    assert abs(float(g["path0_p"]) - float(torch.linspace(0, 0.1, cfg.encoder.depth[2])[2])) < 1e-7


def test_tf_longitudinal_lora_prompt():
    g, cfg, sd, x, prompt, inp, lab, am, tt, pos = gu.tf_longitudinal_case()
    assert np.array_equal(token_ops.token_ids_to_token_type_ids(inp.numpy(), [gu.PMT_SEP, gu.BOS, gu.SEP], [0, 1, 0, 1]), g["token_type_ids"])
    assert np.array_equal(token_ops.position_ids_from_mask(am.numpy()), g["position_ids"])
    names = [str(n) for n in g["grad_names"]]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)
    h, emask = ocvt.encoder_forward(x, sd2, cfg.encoder)
    assert np.array_equal(emask.numpy(), g["enc_mask"])
    logits = obert.decoder_forward(inp, sd2, cfg.decoder, h, emask, am, tt, pos)
    assert gu.rel_rms(gu.sample(logits, 16384), g["logits_sample"]) < FP32_TOL
    loss = ogen.tf_cross_entropy(logits[:, prompt.shape[1]:], lab, gu.PAD)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for i, n in enumerate(names):
        assert gu.rel_rms(gu.sample(leaves[n].grad, 2048), g[f"grad{i}_sample"]) < 2e-3, n


def test_tf_longitudinal_train_mode_lora_dropout():
    """Longitudinal model under model.train(): lora_dropout on the rank-8 branch input (peft Linear), decoder dropouts, train-mode
    BatchNorm / DropPath in the frozen encoder -- with the reference's recorded masks."""
    g, cfg, sd, x, prompt, inp, lab, am, tt, pos, dropout, paths = gu.tf_longitudinal_train_case()
    assert (0, "lora_q") in dropout and (1, "lora_k") in dropout and dropout[(0, "lora_q")].shape == (2, inp.shape[1], 768)
    names = [str(n) for n in g["grad_names"]]
    sd2 = {k: v.clone() for k, v in sd.items()}
    leaves = {n: sd2[n].requires_grad_(True) for n in names}
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, sd2, cfg.encoder, bn_train=True, bn_momentum=cfg.encoder.bn_momentum, drop_path=paths)
    logits = obert.decoder_forward(inp, sd2, cfg.decoder, h, emask, am, tt, pos, dropout=dropout)
    assert gu.rel_rms(gu.sample(logits, 16384), g["logits_sample"]) < FP32_TOL
    loss = ogen.tf_cross_entropy(logits[:, prompt.shape[1]:], lab, gu.PAD)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for i, n in enumerate(names):
        assert gu.rel_rms(gu.sample(leaves[n].grad, 2048), g[f"grad{i}_sample"]) < 2e-3, n


def test_greedy_and_beam_multi():
    g, cfg, sd, x = gu.generate_multi_case()
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, sd, cfg.encoder)

        def fn(ids, am, tt, pos, h=h, emask=emask, sd=sd):
            hh = h if ids.shape[0] == h.shape[0] else h.repeat_interleave(ids.shape[0] // h.shape[0], 0)
            mm = emask if ids.shape[0] == h.shape[0] else emask.repeat_interleave(ids.shape[0] // h.shape[0], 0)
            return obert.decoder_forward(ids, sd, cfg.decoder, hh, mm, None, tt, pos)

        max_len = g["greedy"].shape[1]
        seq, argm, margin = ogen.greedy(fn, "multi", 3, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, max_len, return_margins=True)
        assert np.array_equal(seq.numpy(), g["greedy"])
        np.testing.assert_allclose(margin, g["greedy_margin"], atol=2e-4)
        beam, score = ogen.beam_search(fn, "multi", 3, 4, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, max_len)
        assert np.array_equal(beam.numpy(), g["beam4"])
        np.testing.assert_allclose(score.numpy(), g["beam4_scores"], atol=1e-4)
        # EOS-biased variant: EOS -> PAD fill and beam finalisation
        sd_e = dict(sd)
        sd_e["decoder.cls.predictions.bias"] = sd["decoder.cls.predictions.bias"].clone()
        sd_e["decoder.cls.predictions.bias"][gu.EOS] += float(g["eos_bias"])
        fn_e = lambda *a: fn(*a, sd=sd_e)
        seq = ogen.greedy(fn_e, "multi", 3, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, max_len)
        assert np.array_equal(seq.numpy(), g["greedy_eos"])
        beam, score = ogen.beam_search(fn_e, "multi", 3, 4, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, max_len)
        assert np.array_equal(beam.numpy(), g["beam4_eos"]), (beam, g["beam4_eos"])
        np.testing.assert_allclose(score.numpy(), g["beam4_eos_scores"], atol=1e-4)


def test_single_image_generate_and_noise_robust_beam_fixtures():
    """generate_single.npz (reference SingleCXREncoderDecoderModel: greedy with cache == no-cache, beam-4) and generate_beam_safe.npz (multi-image
    beam-4: plain, an early EOS in some rows, length penalties that decide rankings between hypotheses of different length; every case survived
    bf16-sized logit noise in the generator and every study has its own hypothesis): the oracle's greedy / beam search reproduce them."""
    g, cfg, sd, x = gu.generate_single_case()
    with torch.no_grad():
        h, _ = ocvt.encoder_forward(x, sd, cfg.encoder)

        def fn(ids, am, tt, pos):
            hh = h if ids.shape[0] == h.shape[0] else h.repeat_interleave(ids.shape[0] // h.shape[0], 0)
            return obert.decoder_forward(ids, sd, cfg.decoder, hh, None, None, tt, pos)       # single-image model: no encoder mask (modelling_single.py:176)

        L = g["greedy"].shape[1]
        seq, argm, margin = ogen.greedy(fn, "single", 3, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, L, return_margins=True)
        assert np.array_equal(seq.numpy(), g["greedy"])
        np.testing.assert_allclose(margin, g["greedy_margin"], atol=2e-3)
        beam, score = ogen.beam_search(fn, "single", 3, 4, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, L)
        assert np.array_equal(beam.numpy(), g["beam4_all"][:, 0, : beam.shape[1]])
        np.testing.assert_allclose(score.numpy(), g["beam4_all_scores"][:, 0], atol=2e-4)
    for case in ("plain", "lp2", "lp05", "eos"):
        c = gu.beam_safe_case(case)
        assert c is not None, case
        cfg, sd, x, eos_bias, lp, ref_all, ref_scores, steps, tol = c
        # the fixture's own guarantees: every study has its OWN best hypothesis (a cross-study mix-up cannot reproduce the sequences), and the score
        # tolerance separates the best hypothesis of a row from its runner-up
        assert len({tuple(r.tolist()) for r in ref_all[:, 0]}) == 3, case
        assert 0 < tol < (ref_scores[:, 0] - ref_scores[:, 1]).min(), case
        e = ref_all == gu.EOS
        lens = np.where(e.any(-1), e.argmax(-1) + 1, ref_all.shape[-1])
        if case == "eos":
            assert (lens[:, 0] < ref_all.shape[-1]).any() and not (lens[:, 0] < ref_all.shape[-1]).all()
        if case in ("lp2", "lp05"):
            assert any(len(set(lens[b].tolist())) > 1 for b in range(3)), case            # hypotheses of different length were ranked
        sd = dict(sd)
        if eos_bias:
            sd["decoder.cls.predictions.bias"] = sd["decoder.cls.predictions.bias"].clone()
            sd["decoder.cls.predictions.bias"][gu.EOS] += eos_bias
        with torch.no_grad():
            h, emask = ocvt.encoder_forward(x, sd, cfg.encoder)

            def fn2(ids, am, tt, pos):
                r = ids.shape[0] // h.shape[0]
                return obert.decoder_forward(ids, sd, cfg.decoder, h.repeat_interleave(r, 0), emask.repeat_interleave(r, 0), None, tt, pos)

            beam, score = ogen.beam_search(fn2, "multi", 3, 4, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, steps + 1, length_penalty=lp, return_all=True)
        assert np.array_equal(beam.numpy(), ref_all[:, :, : beam.shape[-1]]) and beam.shape[-1] == ref_all.shape[-1], case
        np.testing.assert_allclose(score.numpy(), ref_scores, atol=2e-4, err_msg=case)


def test_beam_search_as_index_work_on_the_references_recorded_logits():
    """beam_index.npz: the reference's beam-4 generate with its fp32 logits recorded at every step (early EOS, hypotheses of different length,
    length_penalty 0.5 / 1 / 2, searches that stop before max_length). Fed those logits, the oracle's search must walk the same beams at every step
    (the token each running beam is fed) and end with the same four hypotheses per study, bit for bit, scores to fp32 rounding."""
    steps, cases = gu.beam_index_cases()
    seen_lengths, stopped_early = set(), 0
    for name, lp, logits, fed, ref_all, ref_scores in cases:
        fn = gu.replay_logits_fn(logits, fed)
        trace = []
        seqs, scores = ogen.beam_search(fn, "multi", 3, 4, [gu.SEP], gu.BOS, gu.EOS, gu.PAD, steps + 1, length_penalty=lp, return_all=True, trace=trace)
        assert fn.state["t"] == logits.shape[0], (name, fn.state["t"])                    # the search stopped on the step the reference stopped on
        assert torch.equal(seqs, ref_all[:, :, : seqs.shape[-1]]) and seqs.shape == ref_all.shape, name
        np.testing.assert_allclose(scores.numpy(), ref_scores.numpy(), rtol=0, atol=2e-6, err_msg=name)
        assert min(t["min_gap"] for t in trace) > 1e-5, name                             # no selection of the search is a near-tie: exactness is well defined
        e = ref_all == gu.EOS
        lens = torch.where(e.any(-1), e.int().argmax(-1) + 1, torch.full(e.shape[:-1], ref_all.shape[-1]))
        seen_lengths |= set(lens.reshape(-1).tolist())
        stopped_early += int(logits.shape[0] < steps)
    assert len(seen_lengths) >= 6 and stopped_early >= 3


def test_prompted_greedy_scores_and_reinforce():
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, sd, cfg.encoder)
        fn = lambda ids, am, tt, pos: obert.decoder_forward(ids, sd, cfg.decoder, h, emask, am, tt, pos)
        new = g["greedy"].shape[1] - prompt.shape[1]
        seq = ogen.greedy(fn, "longitudinal", 2, [gu.PMT_SEP, gu.BOS, gu.SEP], gu.BOS, gu.EOS, gu.PAD, None,
                          prompt_ids=prompt, mask_token_id=gu.PAD, max_new_tokens=new)
        assert bool(torch.all(seq[:, 0] == gu.BOS))           # HF prepends BOS; callers strip it
        assert np.array_equal(token_ops.strip_prepended_bos(seq.numpy()), g["greedy"])
        # processed scores of the sampling path for the reference's sampled ids (teacher-forced, quirk Q5 special ids)
        seqs = torch.from_numpy(g["sampled_sequences"])
        fed, am, tt, pos = ogen.step_inputs("longitudinal", seqs, [gu.BOS, gu.SEP], gu.PAD, gu.BOS)
        logits = fn(fed, am, tt, pos)
        p = prompt.shape[1]
        scores = ogen.top_k_filter(logits[:, p - 1:-1].float(), 50).permute(0, 2, 1)      # [B,V,T]
        assert list(scores.shape) == g["scores_shape"].tolist()
        finite = torch.isfinite(scores)
        assert np.array_equal(finite.sum(1).numpy(), g["scores_finite_count"])
        assert np.array_equal(np.packbits(finite.numpy(), axis=1), g["scores_finite_mask"])
        sampled = seqs[:, p:]
        np.testing.assert_allclose(torch.gather(scores, 1, sampled[:, None, :])[:, 0].numpy(), g["scores_at_sampled"], atol=2e-4)
        loss = ogen.reinforce_loss(scores, sampled, torch.from_numpy(g["reward"]), gu.PAD)
        assert abs(loss.item() - float(g["reinforce_loss"])) < 2e-4


def test_reward_trunk_matches_transformers_bert():
    g, cfg, sd, ids, am = gu.reward_trunk_case()
    with torch.no_grad():
        h = obert.embeddings(ids, None, None, sd, "bert.embeddings.", cfg.layer_norm_eps)
        mask = torch.zeros(ids.shape[0], 1, 1, ids.shape[1]).masked_fill(~am.bool().view(ids.shape[0], 1, 1, -1), obert.NEG)
        h = obert.bert_layers(h, sd, "bert.", cfg, mask)
        np.testing.assert_allclose(h[:, 0].numpy(), g["cls_state"], atol=2e-4)
        cos = obert.reward_cosine(ids, am, ids.flip(0), am.flip(0), sd, cfg)
        assert cos.shape == (3,) and bool((cos.abs() <= 1.0 + 1e-6).all())
        same = obert.reward_cosine(ids, am, ids, am, sd, cfg)
        np.testing.assert_allclose(same.numpy(), 1.0, atol=1e-5)


def test_top_k_top_p_filters_match_transformers_warpers():
    """The oracle's logits warpers against the classes the reference's generate() runs (TF5 generation/logits_process.py)."""
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(6, 500, generator=g) * 3
    ids = torch.zeros(6, 1, dtype=torch.long)
    for k, p in ((50, 1.0), (50, 0.9), (20, 0.5), (50, 0.05)):
        ref = TopKLogitsWarper(top_k=k)(ids, scores.clone())
        got = ogen.top_k_filter(scores, k)
        if p < 1.0:
            ref = TopPLogitsWarper(top_p=p)(ids, ref)
            got = ogen.top_p_filter(got, p)
        assert torch.equal(torch.isfinite(ref), torch.isfinite(got)) and torch.equal(ref[torch.isfinite(ref)], got[torch.isfinite(got)])
        assert bool((torch.isfinite(got).sum(-1) >= 1).all())


# ------------------------------------------------------------------------------------------------ full-size fixtures (CvT-21, BERT-6, vocab 30000)
def test_full_depth_encoder_matches_reference():
    g, cfg, sd, x = gu.encoder_full_case()
    with torch.no_grad():
        h, mask, stages = ocvt.encoder_forward(x, sd, cfg.encoder, return_stages=True)
    assert list(h.shape) == g["last_hidden_state_shape"].tolist() == [2, 1152, 768]
    assert np.array_equal(mask.numpy(), g["attention_mask"])
    for i, s in enumerate(stages):
        assert list(s.shape) == g[f"stage{i}_shape"].tolist()
        assert gu.rel_rms(gu.sample(s, 16384), g[f"stage{i}_sample"]) < FP32_TOL, i
    assert gu.rel_rms(gu.sample(h, 32768), g["last_hidden_state_sample"]) < FP32_TOL


def test_full_size_tf_logits_and_loss_match_reference():
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_full_case()
    assert np.array_equal(token_ops.token_ids_to_token_type_ids(inp.numpy(), [gu.SEP]), g["token_type_ids"])
    with torch.no_grad():
        h, mask = ocvt.encoder_forward(x, sd, cfg.encoder)
        logits = obert.decoder_forward(inp, sd, cfg.decoder, h, mask, am, tt, None)
        loss = ogen.tf_cross_entropy(logits, lab, gu.PAD)
    assert list(logits.shape) == [2, 256, 30000]
    assert gu.rel_rms(gu.sample(logits, 65536), g["logits_sample"]) < FP32_TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    safe = g["logits_margin"] > 1e-3
    assert np.array_equal(logits.argmax(-1).numpy()[safe], g["logits_argmax"][safe])


def test_full_size_tf_gradients_match_reference():
    """The training step's backward at the size the headline is measured on (CvT-21 @384, BERT-6, V = 30000, T = 256; single.py:449-475): the
    oracle's autograd gradients of 34 parameters spanning every kernel family equal the reference's, and so does the total norm over them."""
    g, cfg, sd, x, inp, lab, am, tt = gu.tf_full_case()
    names = [str(n) for n in g["grad_names"]]
    assert len(names) >= 30
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)                                           # (the oracle's LM head reads word_embeddings.weight itself: tied as in the reference)
    h, mask = ocvt.encoder_forward(x, sd2, cfg.encoder)
    logits = obert.decoder_forward(inp, sd2, cfg.decoder, h, mask, am, tt, None)
    loss = ogen.tf_cross_entropy(logits, lab, gu.PAD)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    for i, n in enumerate(names):
        gr = leaves[n].grad
        assert gu.rel_rms(gu.sample(gr, 4096), g[f"grad{i}_sample"]) < 2e-3, (n, gu.rel_rms(gu.sample(gr, 4096), g[f"grad{i}_sample"]))
        np.testing.assert_allclose(gu.stats(gr)[3], g[f"grad{i}_stats"][3], rtol=2e-3, err_msg=n)


def test_longitudinal_c5_scst_reinforce_gradients_match_reference():
    """configs[4] shape (3 images, 128-token prompt, LoRA): the reference's sampling call (generate.__wrapped__, top-k 50) + reinforce_loss +
    backward with every decoder parameter trainable (scst/gt_prompt.py:38-40,162-180,211-246), restated as ONE teacher-forced pass over the
    reference's sampled ids: kept sets, per-token nll, loss and 16 gradient slices."""
    g, cfg, sd, x = gu.longitudinal_c5_case()
    seqs = torch.from_numpy(g["scst_sampled_sequences"])
    P = g["prompt_ids"].shape[1]
    sampled = seqs[:, P:]
    names = [str(n) for n in g["scst_grad_names"]]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)                                           # (the oracle's LM head reads word_embeddings.weight itself: tied as in the reference)
    with torch.no_grad():
        h, mask = ocvt.encoder_forward(x, sd, cfg.encoder)
    fed, am, tt, pos = ogen.step_inputs("longitudinal", seqs, [gu.BOS, gu.SEP], gu.PAD, gu.BOS)
    lg = obert.decoder_forward(fed, sd2, cfg.decoder, h, mask, am, tt, pos)[:, P - 1:-1]
    sc = ogen.top_k_filter(lg, 50).permute(0, 2, 1)
    assert np.array_equal(torch.isfinite(sc).sum(1).numpy(), g["scst_finite_count"])
    np.testing.assert_allclose(torch.gather(sc, 1, sampled[:, None, :])[:, 0].detach().numpy(), g["scst_scores_at_sampled"], atol=2e-4)
    nll = torch.nn.functional.nll_loss(torch.log_softmax(sc, dim=1), sampled, ignore_index=gu.PAD, reduction="none")
    np.testing.assert_allclose(nll.detach().numpy(), g["scst_nll"], atol=2e-4)
    loss = ogen.reinforce_loss(sc, sampled, torch.from_numpy(g["scst_advantage"]), gu.PAD)
    assert abs(loss.item() - float(g["scst_reinforce_loss"])) < 2e-4
    loss.backward()
    for i, n in enumerate(names):
        gr = leaves[n].grad
        assert gu.rel_rms(gu.sample(gr, 4096), g[f"scst_grad{i}_sample"]) < 2e-3, (n, gu.rel_rms(gu.sample(gr, 4096), g[f"scst_grad{i}_sample"]))
        np.testing.assert_allclose(gu.stats(gr)[3], g[f"scst_grad{i}_stats"][3], rtol=2e-3, err_msg=n)
    wg = leaves["decoder.base_model.model.bert.embeddings.word_embeddings.weight"].grad[torch.from_numpy(g["scst_wordemb_rows"])]
    assert gu.rel_rms(wg.numpy(), g["scst_wordemb_row_grads"]) < 2e-3


def test_longitudinal_c5_shape_matches_reference():
    g, cfg, sd, x = gu.longitudinal_c5_case()
    inp, am, pos = (torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "position_ids"))
    P = g["prompt_ids"].shape[1]
    assert np.array_equal(token_ops.token_ids_to_token_type_ids(inp.numpy(), [gu.PMT_SEP, gu.BOS, gu.SEP], [0, 1, 0, 1]), g["token_type_ids"])
    assert np.array_equal(token_ops.position_ids_from_mask(am.numpy()), g["position_ids"])
    with torch.no_grad():
        h, mask = ocvt.encoder_forward(x, sd, cfg.encoder)
        assert np.array_equal(mask.numpy(), g["enc_mask"]) and h.shape[1] == 3 * 576
        assert gu.rel_rms(gu.sample(h, 16384), g["enc_sample"]) < FP32_TOL
        logits = obert.decoder_forward(inp, sd, cfg.decoder, h, mask, am, torch.from_numpy(g["token_type_ids"]), pos)[:, P - 1:]
        lab = torch.from_numpy(g["full_ids"])[:, 1:]
        loss = ogen.tf_cross_entropy(logits, lab, gu.PAD)
    assert gu.rel_rms(gu.sample(logits, 65536), g["logits_sample"]) < FP32_TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-4


def test_beam_safe_rows_are_told_apart_by_their_sequences():
    """generate_beam_safe.npz: the recorded score tolerance (0.4 x the best / runner-up gap of a row) identifies the hypothesis inside a row. The rows
    themselves are identified by pairwise-different best sequences (asserted by beam_safe_case); a score alone identifies the row only where the
    tolerance is below the smallest row-to-row score distance: true for plain / lp2 / lp05, NOT for eos (0.0229 > 0.0118) -- stated, not hidden."""
    g = gu.load("generate_beam_safe.npz")
    for name in ("plain", "lp2", "lp05", "eos"):
        c = gu.beam_safe_case(name)
        assert c is not None
        sc = c[6]
        assert abs(float(g[f"{name}_score_tol"]) - 0.4 * float((sc[:, 0] - sc[:, 1]).min())) < 1e-7        # the tolerance IS the stated derivation
        best = sc[:, 0]
        dist = min(abs(float(best[i] - best[j])) for i in range(len(best)) for j in range(i))
        assert abs(dist - float(g[f"{name}_row_dist"])) < 1e-7
    assert [gu.beam_safe_rows_differ_by_score(n) for n in ("plain", "lp2", "lp05", "eos")] == [True, True, True, False]
