"""Frozen e4m3 encoder forward (BASELINE.json configs[4] per-GPU shape: 16 studies x 3 images = 48 images), fused vs separate quantisation:
CXR_FP8_FUSED=0/1 python scripts/fp8_encoder_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
dev = torch.device("cuda")
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).eval()
images = torch.randn(16, 3, 3, 384, 384, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    ref = m.encoder(images)["last_hidden_state"].float()
    bf = t(lambda: m.encoder(images))
    m.enable_fp8_encoder(images) if hasattr(m, "enable_fp8_encoder") else m._enc.enable_fp8(images.flatten(0, 1))
    out = m.encoder(images)["last_hidden_state"].float()
    f8 = t(lambda: m.encoder(images))
print(f"fused={os.environ.get('CXR_FP8_FUSED', '1')}: bf16 {bf:.3f} ms, e4m3 {f8:.3f} ms; rel-rms e4m3 vs bf16 {((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.4f}")
