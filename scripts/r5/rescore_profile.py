#!/usr/bin/env python3
"""(round 5: the backward runs inside training.wgrad_overlap() as in scst.scst_step -- the round-4 script ran it on one stream, which is why its
kernel table showed 49 un-batched gemm_tn_reduce_kernel launches that the step itself does not issue.)
The re-scoring phase of the SCST step alone (TF forward over the sampled rows, REINFORCE loss, decoder backward), 10 times: run under
rocprofv3 --kernel-trace --stats for its kernel table."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
from cxrmate_amd.reward import CXRBERTReward
from cxrmate_amd.training import FusedAdamW
from cxrmate_amd import ops

dev = torch.device("cuda")
B, N = 16, 2
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).train()
for p in m.decoder.parameters():
    p.requires_grad_(True)
opt = FusedAdamW(m, lr=5e-6)
reward = CXRBERTReward(dev, seed=1)
g = torch.Generator().manual_seed(0)
images = torch.randn(B, N, 3, 384, 384, generator=g).to(dev)
prompt = torch.tensor([[8, 10, 9, 11, 1]] * B, device=dev)
ones = torch.ones(B, 128, dtype=torch.int64, device=dev)
lab = torch.randint(1000, 30000, (B, 128), generator=g).to(dev)

def T(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()                  # host enqueue time alone (no wait): equal to the wall time = the phase is launch-bound
    for _ in range(n):
        out = fn()
    host = (time.perf_counter() - t) / n * 1e3
    torch.cuda.synchronize()
    print(f"{name:50s} {wall:8.2f} ms   (host enqueue {host:6.2f} ms)")
    return out

with torch.no_grad():
    eo = m.encoder(images)
    seqs, base, rec = m.sample_and_greedy(eo, prompt, [1, 3], [9, 1, 3], 4, 256 + 5, 1, None, 4)
    s2 = seqs[:, 1:]
    P = prompt.shape[1]
    n_new = s2.shape[1] - P
    tf_in = s2[:, :P + n_new - 1].contiguous()
    tt = torch.cat(rec["tt"][:n_new], 1).contiguous(); pos = torch.cat(rec["pos"][:n_new], 1).contiguous()
    mask = (tf_in != 4).to(torch.uint8)
    enc, em = eo.last_hidden_state.contiguous(), eo.attention_mask.to(torch.uint8).contiguous()

    def rescore():                                            # as scst.scst_step: K / V of the session's prefill, LM head on the sampled positions only
        opt.zero_grad()
        logits, saved = m._dec.forward(tf_in, enc, em, mask, tt, pos, save=True, seed=rec["seed"], cross_kv=m._session_cross_kv(rec, enc), logit_from=P - 1)
        Bq, _, V = logits.shape
        flat = logits.view(-1, V)
        thr = ops.topk_threshold(flat, 50)
        labels = s2[:, P:].reshape(-1)
        w = ops.ce_weights(labels, 4, mode=1, reward=torch.ones(Bq, device=dev), T=n_new)
        loss, _, dl = ops.softmax_ce(flat, labels, 4, w, thr=thr)
        from cxrmate_amd.training import wgrad_overlap
        with wgrad_overlap():                                 # as scst.scst_step: weight-gradient GEMMs on the side stream, their split sums added by batched launches
            m._dec.backward(saved, dlogits=dl, need_denc=False)
            ops.wgrad_join()
    torch.cuda.synchronize()
    T("re-score: TF fwd + REINFORCE loss + decoder bwd", rescore, 10)
