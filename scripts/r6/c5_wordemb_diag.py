"""Where does the word-embedding gradient of the C5-size SCST parity test differ from the fp32 oracle? (rows of the tied matrix: LM-head part vs
input-embedding part, per-row relative error of the rows that carry the norm)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import golden_util as gu
from cxrmate_amd import modelling as M
from oracle import bert as obert, cvt as ocvt, generate as ogen

g, cfg, sd, x = gu.longitudinal_c5_case()
seqs = torch.from_numpy(g["scst_sampled_sequences"]); P = g["prompt_ids"].shape[1]; sampled = seqs[:, P:]
adv = torch.from_numpy(g["scst_advantage"])
W = "decoder.base_model.model.bert.embeddings.word_embeddings.weight"


class Split(dict):
    """the first read of the tied matrix (embedding lookup) and the second (LM head) get different leaves"""
    def __init__(self, d, a, b):
        super().__init__(d); self.a, self.b, self.n = a, b, 0
    def __getitem__(self, k):
        if k == W:
            self.n += 1
            return self.a if self.n == 1 else self.b
        return super().__getitem__(k)


a = sd[W].clone().requires_grad_(True); b = sd[W].clone().requires_grad_(True)
sd2 = Split(sd, a, b)
with torch.no_grad():
    h, mask = ocvt.encoder_forward(x, sd, cfg.encoder)
fed, am, tt, pos = ogen.step_inputs("longitudinal", seqs, [gu.BOS, gu.SEP], gu.PAD, gu.BOS)
lg = obert.decoder_forward(fed, sd2, cfg.decoder, h, mask, am, tt, pos)[:, P - 1:-1]
assert sd2.n == 2
sc = ogen.top_k_filter(lg, 50).permute(0, 2, 1)
loss = ogen.reinforce_loss(sc, sampled, adv, gu.PAD)
loss.backward()
ge, gh = a.grad, b.grad
print("oracle: embedding part norm", ge.norm().item(), "head part norm", gh.norm().item(), "sum", (ge + gh).norm().item(), "fixture", g["scst_grad13_stats"][3])

m = M.LongitudinalPromptMultiCXREncoderDecoderModel(cfg, seed=None); m.load_state_dict(sd)
for p in m.encoder.parameters(): p.requires_grad_(False)
for p in m.decoder.parameters(): p.requires_grad_(True)
prompt = torch.from_numpy(g["prompt_ids"]).cuda()
with torch.no_grad():
    eo = m.encoder(x.cuda())
smp = m.generate.__wrapped__(m, input_ids=prompt, special_token_ids=[gu.BOS, gu.SEP], encoder_outputs=eo, bos_token_id=gu.BOS, eos_token_id=gu.EOS,
                             pad_token_id=gu.PAD, mask_token_id=gu.PAD, return_dict_in_generate=True, do_sample=True, num_beams=1, use_cache=True,
                             output_scores=True, top_p=1.0, top_k=50, temperature=1.0, max_new_tokens=sampled.shape[1], forced_tokens=sampled)
s2 = smp["sequences"][:, 1:] if torch.all(smp["sequences"][:, 0] == 1) else smp["sequences"]
scores = torch.stack(smp["scores"], dim=-1)
nll = torch.nn.functional.nll_loss(torch.log_softmax(scores, dim=1), s2[:, P:], ignore_index=gu.PAD, reduction="none")
l2 = (nll.sum(-1) * adv.cuda()).mean(); l2.backward()
ours = m.param(W).grad.float().cpu()
ref = ge + gh
print("ours norm", ours.norm().item(), "rel-rms full matrix", ((ours - ref).norm() / ref.norm()).item())
fed_rows = torch.unique(fed)
in_fed = torch.zeros(ref.shape[0], dtype=torch.bool); in_fed[fed_rows] = True
for name, sel in (("rows fed as inputs", in_fed), ("other rows (head only)", ~in_fed)):
    print(name, int(sel.sum()), "ref norm", ref[sel].norm().item(), "ours", ours[sel].norm().item(), "rel err", ((ours[sel] - ref[sel]).norm() / ref[sel].norm()).item())
print("vs oracle head part on non-fed rows:", ((ours[~in_fed] - gh[~in_fed]).norm() / gh[~in_fed].norm()).item())
rn = ref.norm(dim=1); top = torch.argsort(rn, descending=True)[:12]
for r in top.tolist():
    print(f"row {r:6d} fed={bool(in_fed[r])} count_in_fed={(fed == r).sum().item():3d} ref {rn[r]:.4f} ours {ours[r].norm():.4f} emb-part {ge[r].norm():.4f} head-part {gh[r].norm():.4f} "
          f"err {(ours[r] - ref[r]).norm() / rn[r]:.4f}")
# kept sets
fin = torch.isfinite(scores).sum(1).cpu()
print("kept-set sizes ours min/max", int(fin.min()), int(fin.max()), "oracle", int(torch.isfinite(sc).sum(1).min()), int(torch.isfinite(sc).sum(1).max()))
ok = torch.isfinite(scores).cpu() == torch.isfinite(sc)
print("kept-set disagreements per position:", (~ok).sum(1).tolist())
