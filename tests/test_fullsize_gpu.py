"""GPU: the BASELINE.json model (CvT-21 @ 384 + BERT-6, vocab 30000) at full width and depth, checked through size-independent properties --
the CPU oracle cannot run these shapes in seconds:
  * KV-cached greedy decode == teacher-forced argmax of the sequence it produced (cache vs no-cache), wherever the top-2 margin is safe;
  * batch invariance of generate (a study decoded alone == the same study inside a ragged batch);
  * ragged studies: a zero-padded image is masked out of cross-attention -- replacing it by a different zero-masked "image" cannot change logits;
  * a train-mode optimisation loop on a fixed batch drives the loss down, gradients and running statistics stay finite;
  * the longest sequence the position table allows (512) runs and matches the same model on the 256-token prefix (causality).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MARGIN = 0.05


@pytest.fixture(scope="module")
def model():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cxrmate_amd.config import EncoderDecoderConfig
    from cxrmate_amd.modelling import MultiCXREncoderDecoderModel
    m = MultiCXREncoderDecoderModel(EncoderDecoderConfig(), device="cuda", seed=0, perturb=0.05)
    assert sum(p.numel() for p in m.parameters()) == 112_301_680                     # SURVEY.md appendix A.1 (probe of the reference)
    return m


def _images(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, N, 3, 384, 384, generator=g)


def test_cached_greedy_equals_teacher_forcing_and_is_batch_invariant(model):
    m = model.eval()
    x = _images(3, 2, 1)
    x[1, 1] = 0.0                                                                     # ragged study
    with torch.no_grad():
        eo = m.encoder(x.cuda())
        assert eo.last_hidden_state.shape == (3, 1152, 768)
        assert eo.attention_mask.cpu()[:, ::576].tolist() == [[True, True], [True, False], [True, True]]
        kw = dict(special_token_ids=[3], max_length=33, bos_token_id=1, eos_token_id=None, pad_token_id=4, num_beams=1, use_cache=True,
                  return_dict_in_generate=True)
        seq = m.generate(encoder_outputs=eo, **kw)["sequences"]                        # [3, 33], starts with BOS
        tt = m.token_ids_to_token_type_ids(seq[:, :-1], [3])
        logits = m(encoder_outputs=eo, decoder_input_ids=seq[:, :-1], decoder_attention_mask=torch.ones_like(seq[:, :-1]),
                   decoder_token_type_ids=tt).logits.float()
        top2 = logits.topk(2, dim=-1)
        safe = (top2.values[..., 0] - top2.values[..., 1]) > MARGIN
        assert safe.float().mean() > 0.5
        assert torch.equal(top2.indices[..., 0][safe], seq[:, 1:][safe])               # cache == no cache at full size
        # batch invariance: study 1 (the ragged one) decoded alone
        from cxrmate_amd.modelling import ModelOutput
        eo1 = ModelOutput(last_hidden_state=eo.last_hidden_state[1:2].contiguous(), attention_mask=eo.attention_mask[1:2].contiguous())
        alone = m.generate(encoder_outputs=eo1, **kw)["sequences"]
        first_unsafe = int((~safe[1]).float().argmax()) if bool((~safe[1]).any()) else seq.shape[1] - 1
        assert torch.equal(alone[0, :first_unsafe + 1], seq[1, :first_unsafe + 1])
        # the masked image's content is irrelevant: swap its encoder tokens for noise, keep the mask
        noisy = eo.last_hidden_state.clone()
        noisy[1, 576:] = torch.randn_like(noisy[1, 576:])
        eo2 = ModelOutput(last_hidden_state=noisy, attention_mask=eo.attention_mask)
        logits2 = m(encoder_outputs=eo2, decoder_input_ids=seq[:, :-1], decoder_attention_mask=torch.ones_like(seq[:, :-1]),
                    decoder_token_type_ids=tt).logits.float()
        assert torch.equal(logits2[1], logits[1])


def test_max_length_sequence_is_causal(model):
    m = model.eval()
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(12, 30000, (2, 512), generator=g).cuda()
    ids[:, 0] = 1
    x = _images(2, 1, 2)
    with torch.no_grad():
        eo = m.encoder(x.cuda())
        tt = m.token_ids_to_token_type_ids(ids, [3])
        full = m(encoder_outputs=eo, decoder_input_ids=ids, decoder_attention_mask=torch.ones_like(ids), decoder_token_type_ids=tt).logits
        half = m(encoder_outputs=eo, decoder_input_ids=ids[:, :256], decoder_attention_mask=torch.ones_like(ids[:, :256]),
                 decoder_token_type_ids=tt[:, :256]).logits
    assert full.shape == (2, 512, 30000) and bool(torch.isfinite(full).all())
    a, b = full[:, :256].float().cpu().numpy(), half.float().cpu().numpy()
    rel = float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
    assert rel < 5e-3, rel                                                           # same prefix, different tiling of the key range only


def test_train_mode_loop_reduces_loss(model):
    from cxrmate_amd.training import FusedAdamW, tf_train_step
    m = model.train()
    opt = FusedAdamW(m, lr=2e-4)
    g = torch.Generator().manual_seed(3)
    B, T = 4, 64
    x = _images(B, 2, 4).cuda()
    full = torch.randint(12, 30000, (B, T + 1), generator=g)
    full[:, 0] = 1
    inp, lab = full[:, :-1].cuda(), full[:, 1:].contiguous().cuda()
    am = torch.ones_like(inp)
    tt = m.token_ids_to_token_type_ids(inp, [3])
    nbt0 = int(m.num_batches_tracked[0])
    torch.manual_seed(0)
    losses = [float(tf_train_step(m, opt, x, inp, am, tt, lab, 4).item()) for _ in range(8)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0] - 0.5, losses                                       # memorises the fixed batch
    assert int(m.num_batches_tracked[0]) == nbt0 + 8                                  # BatchNorm counted every train-mode forward
    sd = m.state_dict()
    for k in ("encoder.cvt.encoder.stages.2.layers.15.attention.attention.convolution_projection_key.convolution_projection.normalization.running_var",
              "encoder.cvt.encoder.stages.0.layers.0.attention.attention.convolution_projection_query.convolution_projection.normalization.running_mean"):
        assert bool(torch.isfinite(sd[k]).all())
    m.eval()
