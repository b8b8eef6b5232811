"""GPU: CXR-BERT stand-in reward and the SCST step against the fixtures / CPU oracle."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def test_reward_trunk_and_cosine(cuda):
    from cxrmate_amd.reward import CXRBERTReward
    from oracle import bert as obert
    g, cfg, sd, ids, am = gu.reward_trunk_case()
    r = CXRBERTReward(cuda, config=cfg, state_dict=sd)
    hidden, _ = r.engine.forward(ids.cuda(), attn_mask=am.to(torch.uint8).cuda(), causal=False, lm_head=False)
    # bidirectional trunk vs transformers.BertModel (fixture): fp tolerance of bf16 kernels
    assert gu.rel_rms(hidden[:, 0].float().cpu().numpy(), g["cls_state"]) < 2.5e-2
    with torch.no_grad():
        ref = obert.reward_cosine(ids, am, ids.flip(0), am.flip(0), sd, cfg)
    got = r.reward_from_ids(ids, am, ids.flip(0), am.flip(0)).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-2)           # cosine of 128-d projections, bf16 trunk
    same = r.reward_from_ids(ids, am, ids, am).cpu()
    np.testing.assert_allclose(same.numpy(), 1.0, atol=1e-5)
    with pytest.raises(AssertionError):
        r.reward("not a list", [["x"]])
    with pytest.raises(AssertionError):
        r.reward(["a"], ["not a list of lists"])


def test_reward_with_tokenizer_surface(cuda):
    import os
    import transformers
    from cxrmate_amd.reward import CXRBERTReward
    tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(gu.GOLDEN, "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]",
                                               cls_token="[BOS]", sep_token="[SEP]", mask_token="[MASK]")
    cfg = gu.BertConfig(vocab_size=600, num_hidden_layers=2, is_decoder=False, add_cross_attention=False, cls_projection_size=128)
    r = CXRBERTReward(cuda, tokenizer=tok, config=cfg)
    preds = ["The lungs are clear.", "Mild cardiomegaly is stable."]
    labels = [["The lungs are clear."], ["No acute cardiopulmonary process."]]
    out = r(preds, labels)
    assert out.shape == (2,) and abs(out[0].item() - 1.0) < 1e-4 and out[1].item() < 0.9999
    assert len(r._label_cache) == 1
    out2 = r(["Normal.", "Normal."], labels)                    # label embeddings are reused (reference embeds them twice per step)
    assert len(r._label_cache) == 1 and out2.shape == (2,)


def test_cxr_bert_metric_matches_reference_flow(cuda, tmp_path):
    """Evaluation-time twin of the reward (reference tools/metrics/cxr_bert.py): same update/compute surface, mini-batches, per-study
    mean over DICOMs, CSV output; similarities equal the oracle's cosine of the stand-in embeddings."""
    import glob
    import os
    import pandas as pd
    import transformers
    from cxrmate_amd.metrics import CXRBERT
    from cxrmate_amd.reward import CXRBERTReward
    from oracle import bert as obert
    tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(gu.GOLDEN, "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]",
                                               cls_token="[BOS]", sep_token="[SEP]", mask_token="[MASK]")
    cfg = gu.BertConfig(vocab_size=600, num_hidden_layers=2, is_decoder=False, add_cross_attention=False, cls_projection_size=128)
    r = CXRBERTReward(cuda, tokenizer=tok, config=cfg)
    metric = CXRBERT("test", None, 2, str(tmp_path), accumulate_over_dicoms=True, reward=r)
    preds = ["The lungs are clear.", "Mild cardiomegaly is stable.", "No pleural effusion.", "The lungs are clear.", "Normal."]
    labels = [["The lungs are clear."], ["No acute cardiopulmonary process."], ["Small left pleural effusion."], ["Normal."], ["Normal."]]
    metric.update(preds[:3], labels[:3], [10, 10, 11], ["a", "b", "c"])
    metric.update(preds[3:], labels[3:], [12, 12], ["d", "d"])                     # duplicated dicom (DDP padding) is dropped
    with pytest.raises(AssertionError):
        metric.update("not a list", labels, [1], ["x"])
    score = metric.compute(epoch=3)
    files = glob.glob(os.path.join(str(tmp_path), "cxr_bert", "test_epoch-3_scores_*.csv"))
    assert len(files) == 1
    df = pd.read_csv(files[0])
    assert list(df.columns) == ["dicom_id", "study_id", "similarity"] and len(df) == 4
    sd = {k: v.detach().float().cpu() for k, v in r.model.state_dict().items()}
    kw = dict(add_special_tokens=True, padding="longest", return_tensors="pt", truncation=True, max_length=cfg.max_position_embeddings)
    want = []
    for p_, l_ in zip(preds[:4], labels[:4]):
        a, b = tok([p_], **kw), tok([l_[0]], **kw)
        with torch.no_grad():
            want.append(float(obert.reward_cosine(a.input_ids, a.attention_mask, b.input_ids, b.attention_mask, sd, cfg)))
    np.testing.assert_allclose(df.similarity.to_numpy(), np.array(want), atol=2e-2)
    per_study = [np.mean(want[:2]), want[2], want[3]]
    assert abs(score - float(np.mean(per_study))) < 2e-2
    metric.reset()
    assert metric.reports == []


def test_chexbert_labeller_heads(cuda):
    """CheXbert (reference tools/chexbert.py): trunk + 14 heads as one GEMM + segmented argmax == fp32 oracle trunk + per-head argmax."""
    import os
    import transformers
    from cxrmate_amd.chexbert import CheXbert, HEAD_CLASSES
    from oracle import bert as obert
    tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(gu.GOLDEN, "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]",
                                               cls_token="[BOS]", sep_token="[SEP]", mask_token="[MASK]")
    cfg = gu.BertConfig(vocab_size=600, num_hidden_layers=2, is_decoder=False, add_cross_attention=False, cls_projection_size=0)
    cb = CheXbert(cuda, tokenizer=tok, config=cfg)
    reports = ["  The lungs are clear.\nNo pleural effusion. ", "Mild cardiomegaly is stable.", "Support devices in place.", "Normal."]
    labels = cb(list(reports))
    assert labels.shape == (4, 14) and labels.dtype == torch.int64
    assert int(labels[:, :13].max()) <= 3 and int(labels[:, 13].max()) <= 1
    clean = [r.strip().replace("\n", " ").strip() for r in reports]
    t = tok(clean, padding="longest", return_tensors="pt", truncation=True, max_length=cfg.max_position_embeddings)
    got, logits = cb.label_ids(t["input_ids"], t["attention_mask"], t.get("token_type_ids"))
    assert torch.equal(got, labels)
    sd = {k: v.detach().float().cpu() for k, v in cb.model.state_dict().items()}
    with torch.no_grad():
        b, tlen = t["input_ids"].shape
        h = obert.embeddings(t["input_ids"], t.get("token_type_ids"), None, sd, "bert.embeddings.", cfg.layer_norm_eps)
        mask = torch.zeros(b, 1, 1, tlen).masked_fill(~t["attention_mask"].bool().view(b, 1, 1, tlen), obert.NEG)
        cls = obert.bert_layers(h, sd, "bert.", cfg, mask, None, None, cfg.layer_norm_eps)[:, 0]
        ref = torch.cat([torch.nn.functional.linear(cls, sd[f"linear_heads.{i}.weight"], sd[f"linear_heads.{i}.bias"]) for i in range(14)], 1)
    np.testing.assert_allclose(logits.cpu().numpy(), ref.numpy(), atol=3e-2)
    o = 0
    for i, n in enumerate(HEAD_CLASSES):                              # argmax agrees wherever the fp32 margin exceeds the bf16 error
        seg = ref[:, o:o + n]
        top2 = seg.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 6e-2
        assert torch.equal(labels[:, i].cpu()[safe], seg.argmax(1)[safe]), i
        assert torch.equal(labels[:, i], logits[:, o:o + n].argmax(1))      # segmented argmax kernel == torch.argmax on the same logits
        o += n


@pytest.mark.parametrize("temperature", [1.0, 0.7])
def test_scst_step_matches_oracle_reinforce(cuda, temperature):
    from cxrmate_amd import modelling
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW
    from oracle import bert as obert, cvt as ocvt, generate as ogen
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device=cuda, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)                                   # scst/gt_prompt.py:38-40
    opt = FusedAdamW(m, lr=1e-3)
    before = m.f32("decoder.base_model.model.bert.encoder.layer.0.output.dense.weight").clone()
    enc_before = m.f32("encoder.projection_head.projection.weight").clone()
    calls = []

    def reward_fn(ids):                                          # deterministic synthetic reward of the generated ids
        calls.append(ids)
        return ((ids % 7).float().mean(1) / 7.0).to(torch.float32)

    special = dict(bos=gu.BOS, eos=gu.EOS, sep=gu.SEP, pad=gu.PAD, pmt_sep=gu.PMT_SEP)
    torch.manual_seed(3)
    out = scst_step(m, opt, reward_fn, x.cuda(), prompt.cuda(), None, special, decoder_max_len=10, temperature=temperature)      # scst_sample_temperature
    torch.cuda.synchronize()
    assert len(calls) == 2 and np.isfinite(out["loss"].item())
    sampled = out["sampled"].cpu()
    P = prompt.shape[1]
    assert sampled.shape[0] == 2 and 1 <= sampled.shape[1] <= 9
    assert out["baseline_ids"].shape[1] <= P + 10
    # oracle REINFORCE for the same sampled ids / advantages (weights BEFORE the update)
    adv = (reward_fn(out["sampled"]) - reward_fn(out["baseline_ids"][:, P:])).cpu()
    with torch.no_grad():
        h, emask = ocvt.encoder_forward(x, sd, cfg.encoder)
        seqs = torch.cat([prompt, sampled], 1)
        fed, am, tt, pos = ogen.step_inputs("longitudinal", seqs, [gu.BOS, gu.SEP], gu.PAD, gu.BOS)
        lg = obert.decoder_forward(fed, sd, cfg.decoder, h, emask, am, tt, pos)
        # the oracle's filter keeps its fp32 top-50 AND the sampled token (which can sit at the edge of the bf16 top-50, see test_model_gpu):
        # the loss is compared unconditionally
        sc = ogen.top_k_filter(lg[:, P - 1:-1].float() / temperature, 50, keep=sampled).permute(0, 2, 1)        # TemperatureLogitsWarper, then top-k
        nll = torch.nn.functional.nll_loss(torch.log_softmax(sc, 1), sampled, ignore_index=gu.PAD, reduction="none")
    assert bool(torch.isfinite(nll).all())
    oloss = (nll.sum(-1) * adv).mean()
    assert abs(out["loss"].item() - oloss.item()) < 0.05 * max(1.0, abs(oloss.item())), (out["loss"].item(), oloss.item())
    after = m.f32("decoder.base_model.model.bert.encoder.layer.0.output.dense.weight")
    assert not torch.equal(before, after)                        # decoder updated
    assert torch.equal(enc_before, m.f32("encoder.projection_head.projection.weight"))      # encoder frozen (scst/gt_prompt.py:35-36)


def test_fused_sample_and_greedy_decode(cuda):
    """One 2B-row decode (rows [0,B) sampled, rows [B,2B) greedy, different separator sets) == the two separate generate() calls of
    scst/gt_prompt.py:94-118,162-180: greedy half bit-identical, sampled half inside the top-k of its own recorded step inputs."""
    from cxrmate_amd import modelling
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device=cuda, seed=None)
    m.load_state_dict(sd)
    P = prompt.shape[1]
    L = 12
    with torch.no_grad():
        eo = m.encoder(x.cuda())
        base = m.generate(encoder_outputs=eo, decoder_input_ids=prompt.cuda(), special_token_ids=[gu.PMT_SEP, gu.BOS, gu.SEP],
                          max_length=L + P, bos_token_id=gu.BOS, eos_token_id=gu.EOS, pad_token_id=gu.PAD, mask_token_id=gu.PAD,
                          num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"]
        torch.manual_seed(5)
        smp, grd, rec = m.sample_and_greedy(eo, prompt.cuda(), [gu.BOS, gu.SEP], [gu.PMT_SEP, gu.BOS, gu.SEP], gu.PAD, L + P, gu.BOS, gu.EOS,
                                            gu.PAD, top_k=5)
    assert torch.equal(grd, base)
    assert smp.shape[0] == 2 and smp.shape[1] <= L + P and len(rec["tt"]) >= smp.shape[1] - (P + 1)
    # every sampled token is one of the 5 best under the teacher-forced re-scoring with the recorded per-step inputs
    n_new = smp.shape[1] - (P + 1)
    fed = smp[:, 1:]
    tf_in = fed[:, : P + n_new - 1].contiguous()
    tt = torch.cat(rec["tt"][:n_new], 1).contiguous()
    pos = torch.cat(rec["pos"][:n_new], 1).contiguous()
    with torch.no_grad():
        lg, _ = m._dec.forward(tf_in, eo.last_hidden_state.contiguous(), eo.attention_mask.to(torch.uint8).contiguous(),
                               (tf_in != gu.PAD).to(torch.uint8), tt, pos)
    # the session projected the cross-attention K / V of all layers in one GEMM at prefill; the re-scoring pass reads that buffer instead of projecting
    # the encoder output again: same logits bit for bit, same parameter gradients, and a stale buffer (other weights / another batch shape) is refused
    enc16, em8 = eo.last_hidden_state.contiguous(), eo.attention_mask.to(torch.uint8).contiguous()
    ckv = m._session_cross_kv(rec, enc16)
    assert ckv is not None and ckv.tensor.shape == (enc16.shape[0], enc16.shape[1], 2 * cfg.decoder.num_hidden_layers * cfg.decoder.hidden_size)
    with torch.no_grad():
        lg2, _ = m._dec.forward(tf_in, enc16, em8, (tf_in != gu.PAD).to(torch.uint8), tt, pos, cross_kv=ckv)
    assert torch.equal(lg, lg2)
    grads = []
    for use in (None, ckv):
        m.zero_grads_prefix("decoder.")
        lgs, saved = m._dec.forward(tf_in, enc16, em8, (tf_in != gu.PAD).to(torch.uint8), tt, pos, save=True, cross_kv=use)
        dl = torch.zeros_like(lgs, dtype=torch.bfloat16)
        dl[:, :, 7] = 1.0
        m._dec.backward(saved, dlogits=dl.view(-1, dl.shape[-1]), need_denc=False)
        ops_ = __import__("cxrmate_amd.ops", fromlist=["x"])
        ops_.wgrad_join()
        torch.cuda.synchronize()
        grads.append(m.gflat.clone())
    assert torch.allclose(grads[0], grads[1], rtol=1e-5, atol=1e-7)
    # the LM head on positions t0 .. T-1 only (the re-scoring pass scores the sampled positions, not the prompt's): the logits ARE the slice of the full
    # ones, and a gradient that is zero in front of t0 gives the same parameter gradients either way
    t0 = P - 1
    with torch.no_grad():
        lg3, _ = m._dec.forward(tf_in, enc16, em8, (tf_in != gu.PAD).to(torch.uint8), tt, pos, logit_from=t0)
    assert lg3.is_contiguous() and torch.equal(lg3, lg[:, t0:, :])
    gsl = []
    for frm in (0, t0):
        m.zero_grads_prefix("decoder.")
        lgs, saved = m._dec.forward(tf_in, enc16, em8, (tf_in != gu.PAD).to(torch.uint8), tt, pos, save=True, logit_from=frm)
        dl = torch.zeros_like(lgs, dtype=torch.bfloat16)
        dl[:, (t0 - frm):, 7] = 1.0
        dl[:, (t0 - frm):, 11] = -0.5
        m._dec.backward(saved, dlogits=dl.view(-1, dl.shape[-1]), need_denc=False)
        ops_.wgrad_join()
        torch.cuda.synchronize()
        gsl.append(m.gflat.clone())
    assert torch.allclose(gsl[0], gsl[1], rtol=1e-4, atol=1e-6), float((gsl[0] - gsl[1]).abs().max())
    assert m._session_cross_kv(rec, enc16[:1]) is None
    # The lent buffer is the decode session's own storage: a decode of ANOTHER batch of the same geometry between the re-scoring forward and its
    # backward overwrites the K / V that attention_bwd would read (round-4 advisor finding). The loan carries the session's prefill count: the
    # backward refuses, a new loan is refused, and the caller's fallback (project again) gives the gradients of the un-shared pass.
    m.zero_grads_prefix("decoder.")
    lgs, saved = m._dec.forward(tf_in, enc16, em8, (tf_in != gu.PAD).to(torch.uint8), tt, pos, save=True, cross_kv=m._session_cross_kv(rec, enc16))
    with torch.no_grad():
        x2 = x.flip(0).contiguous()
        eo2 = m.encoder(x2.cuda())
        m.sample_and_greedy(eo2, prompt.cuda(), [gu.BOS, gu.SEP], [gu.PMT_SEP, gu.BOS, gu.SEP], gu.PAD, L + P, gu.BOS, gu.EOS, gu.PAD, top_k=5)
    dl = torch.zeros_like(lgs, dtype=torch.bfloat16)
    dl[:, :, 7] = 1.0
    with pytest.raises(RuntimeError, match="another prefill"):
        m._dec.backward(saved, dlogits=dl.view(-1, dl.shape[-1]), need_denc=False)
    ops_.wgrad_join()
    assert m._session_cross_kv(rec, enc16) is None                             # the same record can no longer borrow: the caller projects again
    m.zero_grads_prefix("decoder.")
    lgs, saved = m._dec.forward(tf_in, enc16, em8, (tf_in != gu.PAD).to(torch.uint8), tt, pos, save=True, cross_kv=m._session_cross_kv(rec, enc16))
    m._dec.backward(saved, dlogits=dl.view(-1, dl.shape[-1]), need_denc=False)
    ops_.wgrad_join()
    torch.cuda.synchronize()
    assert torch.allclose(m.gflat, grads[0], rtol=1e-5, atol=1e-7)
    m.flat32.add_(0.0)                                                         # any in-place torch edit of the weights moves the stamp
    assert m._session_cross_kv(rec, enc16) is None
    sc = lg[:, P - 1:, :].float()
    new = fed[:, P:]
    kth = sc.topk(6, dim=-1).values[..., -1]                                   # one rank of slack for bf16 cached-vs-TF differences
    tok_sc = sc.gather(-1, new.clamp_min(0)[..., None])[..., 0]
    live = new != gu.PAD
    assert bool(((tok_sc >= kth) | ~live).all())
    # the step built on it still trains (fused path is the default of scst_step)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    opt = FusedAdamW(m, lr=1e-3)
    out = scst_step(m, opt, lambda ids: ((ids % 5).float().mean(1) / 5.0), x.cuda(), prompt.cuda(), None,
                    dict(bos=gu.BOS, eos=gu.EOS, sep=gu.SEP, pad=gu.PAD, pmt_sep=gu.PMT_SEP), decoder_max_len=10, fused_decode=False)
    assert np.isfinite(out["loss"].item())


def test_scst_step_through_real_strings(cuda):
    """The reference's reward path inside scst_step (gt_prompt.py:90-91,120-128,192-197): ids -> findings / impression strings -> CXR-BERT
    tokenizer -> reward, on pinned host copies taken while the re-scoring forward already runs. The rewards it reports equal ReportReward
    applied to the sequences it returns; the label embeddings are computed once."""
    import os
    import transformers
    from cxrmate_amd import modelling
    from cxrmate_amd.reward import CXRBERTReward, ReportReward
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device=cuda, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    opt = FusedAdamW(m, lr=1e-3)
    tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(gu.GOLDEN, "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]",
                                               cls_token="[BOS]", sep_token="[SEP]", mask_token="[MASK]", bos_token="[BOS]", eos_token="[EOS]")
    rcfg = gu.BertConfig(vocab_size=600, num_hidden_layers=2, is_decoder=False, add_cross_attention=False, cls_projection_size=128)
    reward = CXRBERTReward(cuda, tokenizer=tok, config=rcfg)
    labels = [["The lungs are clear. No acute cardiopulmonary process."], ["Mild cardiomegaly is stable. Small left pleural effusion."]]
    rfn = ReportReward(m, tok, reward, labels, gu.BOS, gu.SEP, gu.EOS)
    n_label_embeds = []
    orig = reward.embed_ids
    reward.embed_ids = lambda ids, mask: (n_label_embeds.append(int(ids.shape[0])), orig(ids, mask))[1]
    special = dict(bos=gu.BOS, eos=gu.EOS, sep=gu.SEP, pad=gu.PAD, pmt_sep=gu.PMT_SEP)
    torch.manual_seed(3)
    out = scst_step(m, opt, rfn, x.cuda(), prompt.cuda(), None, special, decoder_max_len=10, reward_on_host=True)
    torch.cuda.synchronize()
    assert np.isfinite(out["loss"].item())
    # ONE reward forward per step: sampled + greedy predictions (2 x 2 studies) and, in the same launch chain, the step's 2 label rows (embedded once)
    assert n_label_embeds == [6], n_label_embeds
    P = prompt.shape[1]
    full_s = torch.cat([prompt, out["sampled"].cpu()], 1)
    want_r = rfn(full_s).float().cpu().numpy()
    want_b = rfn(out["baseline_ids"].cpu()).float().cpu().numpy()
    np.testing.assert_allclose(out["global"]["reward"].cpu().numpy(), want_r, atol=2e-3)       # (one 4-row pass vs two 2-row passes: other padding)
    np.testing.assert_allclose(out["global"]["baseline"].cpu().numpy(), want_b, atol=2e-3)


def test_scst_generated_prompt_steps_chain_through_the_written_back_report(cuda):
    """SCSTGeneratedPrompt.scst_step (reference scst/gen_prompt.py:174-259, configs[4]'s caller) at its mbatch_size of 1: three consecutive
    studies of one patient -- the prompt of each step is tokenised from the PREVIOUS step's greedy report strings (written back :243-246), the
    first from the no-previous-study tokens. Checks the prompt ids the step fed, that the returned strings are the decoded greedy rows, and that
    the model trains."""
    import os
    import transformers
    from cxrmate_amd import modelling
    from cxrmate_amd.reward import CXRBERTReward
    from cxrmate_amd.scst import scst_generated_prompt_step
    from cxrmate_amd.training import FusedAdamW
    g, cfg, sd, x, _ = gu.generate_longitudinal_case()
    m = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device=cuda, seed=None)
    m.load_state_dict(sd)
    m.train()
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    opt = FusedAdamW(m, lr=1e-3)
    tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(gu.GOLDEN, "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]",
                                               cls_token="[BOS]", sep_token="[SEP]", mask_token="[MASK]", bos_token="[BOS]", eos_token="[EOS]")
    rcfg = gu.BertConfig(vocab_size=600, num_hidden_layers=2, is_decoder=False, add_cross_attention=False, cls_projection_size=128)
    reward = CXRBERTReward(cuda, tokenizer=tok, config=rcfg)
    prev_f, prev_i = [None], [None]                               # first study of the patient: [NPF] / [NPI]
    w0 = m.param("decoder.base_model.model.bert.encoder.layer.0.output.dense.weight").detach().clone()
    torch.manual_seed(5)
    for study in range(3):
        images = x[study % 2:study % 2 + 1].cuda()                # one study (2 images) per step
        out = scst_generated_prompt_step(m, opt, reward, tok, images, prev_f, prev_i, ["The lungs are clear."], ["No acute process."],
                                         decoder_max_len=12)
        torch.cuda.synchronize()
        assert np.isfinite(out["loss"].item())
        want_prompt = m.tokenize_prompt(prev_f, prev_i, tok, 12, add_bos_token_id=True)["input_ids"]
        assert torch.equal(out["prompt_ids"], want_prompt)
        if study == 0:
            assert out["prompt_ids"].cpu().tolist() == [[gu.PMT, gu.NPF, gu.PMT_SEP, gu.NPI, gu.BOS]]
        base = out["baseline_ids"].cpu()
        _, f, i = m.split_and_decode_sections(base, [gu.BOS, gu.SEP, gu.EOS], tok)
        assert out["baseline_findings"] == f and out["baseline_impression"] == i and len(f) == 1
        assert base.shape[1] <= out["prompt_ids"].shape[1] + 12
        prev_f, prev_i = out["baseline_findings"], out["baseline_impression"]          # written back: the next study's prompt
    assert not torch.equal(w0, m.param("decoder.base_model.model.bert.encoder.layer.0.output.dense.weight").detach())


@pytest.mark.parametrize("children", [2, 4])
def test_scst_step_with_the_string_worker_equals_the_in_process_path(cuda, children, monkeypatch):
    """(children = CXR_STRING_WORKERS: one | two child processes per half of the rows.) reward.ReportReward(worker=True): the CPU part of the reward (ids -> strings -> reward-tokenizer ids) runs in a child process between
    pair_start() and pair_finish() while scst_step queues its re-scoring pass. Same seeds, lr = 0: the step with the worker reports the rewards,
    the advantage-weighted loss and the greedy sections of the in-process step bit for bit, the child did serve it, and a worker that dies between
    two steps leaves a step that still works (in-process fallback)."""
    import os
    import transformers
    from cxrmate_amd import modelling
    from cxrmate_amd.reward import CXRBERTReward, ReportReward
    from cxrmate_amd.scst import scst_step
    from cxrmate_amd.training import FusedAdamW
    g, cfg, sd, x, prompt = gu.generate_longitudinal_case()
    m = modelling.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device=cuda, seed=None)
    m.load_state_dict(sd)
    for p in m.decoder.parameters():
        p.requires_grad_(True)
    opt = FusedAdamW(m, lr=0.0, weight_decay=0.0)
    tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(gu.GOLDEN, "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]",
                                               cls_token="[BOS]", sep_token="[SEP]", mask_token="[MASK]", bos_token="[BOS]", eos_token="[EOS]")
    rcfg = gu.BertConfig(vocab_size=600, num_hidden_layers=2, is_decoder=False, add_cross_attention=False, cls_projection_size=128)
    reward = CXRBERTReward(cuda, tokenizer=tok, config=rcfg, max_length=48)
    labels = [["The lungs are clear. No acute cardiopulmonary process."], ["Mild cardiomegaly is stable. Small left pleural effusion."]]
    special = dict(bos=gu.BOS, eos=gu.EOS, sep=gu.SEP, pad=gu.PAD, pmt_sep=gu.PMT_SEP)

    def run(rfn):
        torch.manual_seed(3)
        out = scst_step(m, opt, rfn, x.cuda(), prompt.cuda(), None, special, decoder_max_len=10, reward_on_host=True)
        torch.cuda.synchronize()
        return (out["loss"].item(), out["global"]["reward"].cpu().numpy(), out["global"]["baseline"].cpu().numpy(), out["sampled"].cpu().numpy(),
                tuple(map(tuple, rfn.last_sections)))

    plain = ReportReward(m, tok, reward, labels, gu.BOS, gu.SEP, gu.EOS)
    ref = run(plain)
    monkeypatch.setenv("CXR_STRING_WORKERS", str(children))
    wrk = ReportReward(m, tok, reward, labels, gu.BOS, gu.SEP, gu.EOS, worker=True)
    try:
        assert wrk.worker is not None and wrk.worker.alive and len(wrk.workers) == children
        got = run(wrk)
        assert wrk.worker_used == 1
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]) and got[4] == ref[4]
        wrk.worker.proc.kill(); wrk.worker.proc.wait()
        again = run(wrk)                                          # the child is gone: the step falls back to the in-process path, same numbers
        assert wrk.worker_used == 1 and again[0] == ref[0] and np.array_equal(again[1], ref[1]) and again[4] == ref[4]
    finally:
        wrk.close()
