#!/usr/bin/env python3
"""Host / GPU timeline of one SCST step with the string round trip (configs[3] shape): where the host is when the GPU reaches each phase.
usage: scst_timeline.py [worker]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import transformers
from cxrmate_amd import scst
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
from cxrmate_amd.reward import CXRBERTReward, ReportReward
from cxrmate_amd.strings import FoldedVocabTokenizer
from cxrmate_amd.training import FusedAdamW

dev = torch.device("cuda")
B, N = 16, 2
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).train()
for p in m.decoder.parameters():
    p.requires_grad_(True)
opt = FusedAdamW(m, lr=5e-6)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tok = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(root, "tests", "golden", "tokenizer.json"), unk_token="[UNK]", pad_token="[PAD]", cls_token="[BOS]",
                                           sep_token="[SEP]", mask_token="[MASK]", bos_token="[BOS]", eos_token="[EOS]")
reward = CXRBERTReward(dev, tokenizer=tok, seed=1, max_length=128)
labels = [["The lungs are clear without focal consolidation. No pleural effusion or pneumothorax. No acute cardiopulmonary process."]] * B
rfn = ReportReward(m, FoldedVocabTokenizer(tok), reward, labels, 1, 3, 2, worker="worker" in sys.argv[1:])
g = torch.Generator().manual_seed(0)
images = torch.randn(B, N, 3, 384, 384, generator=g).to(dev)
prompt = torch.tensor([[8, 10, 9, 11, 1]] * B, device=dev)
special = dict(bos=1, eos=None, sep=3, pad=4, pmt_sep=9)
step = lambda: scst.scst_step(m, opt, rfn, images, prompt, None, special, decoder_max_len=256, reward_on_host=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(3):
    scst.TRACE = []
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tr, scst.TRACE = scst.TRACE, None
    e0 = tr[0][2]
    print(f"--- step {rep}: {1e3 * (t1 - t0):.2f} ms wall ({'child process' if rfn.worker is not None else 'in-process'} strings)")
    for label, th, ev in tr:
        print(f"  {label:48s} host reaches it at {1e3 * (th - tr[0][1]):8.2f} ms   GPU reaches it at {e0.elapsed_time(ev):8.2f} ms")
rfn.close()
