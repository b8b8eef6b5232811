"""GPU tail of the reference's input pipeline: `ToTensor` + `Normalize(mean, std)` of the image transforms
(modules/lightning_modules/single.py:248-262) and the `pad_sequence(..., padding_value=0.0)` of `collate_fn` (multi.py:155-164) in one
kernel, from the decoded / resized / cropped uint8 images. JPEG decode, resize, crop and rotation stay on the host (dataloader workers);
what crosses PCIe is 1 byte per sample instead of 4.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import LIB

IMAGENET_MEAN = (0.485, 0.456, 0.406)          # image_processor.image_mean / image_std of microsoft/cvt-21-384-22k
IMAGENET_STD = (0.229, 0.224, 0.225)


def collate_images(studies, device, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """studies: list (one entry per study) of uint8 tensors [n_i, H, W, 3] (HWC, as PIL / decode hands them over).
    -> `images` fp32 [B, max n_i, 3, H, W] on `device`, normalised, zero-padded like the reference batch['images']."""
    counts = [int(s.shape[0]) for s in studies]
    H, W = int(studies[0].shape[1]), int(studies[0].shape[2])
    packed = torch.cat([s.reshape(-1, H, W, 3) for s in studies], dim=0).contiguous()
    assert packed.dtype == torch.uint8
    first = torch.zeros(len(studies) + 1, dtype=torch.int64)
    first[1:] = torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0)
    packed, first = packed.to(device, non_blocking=True), first.to(device, non_blocking=True)
    B, Nmax = len(studies), max(counts)
    out = torch.empty((B, Nmax, 3, H, W), dtype=torch.float32, device=device)
    LIB.call("cxr_pixels_u8_to_f32", ops._p(packed), ops._p(first), ops._p(out), B, Nmax, H, W, *[float(m) for m in mean], *[float(s) for s in std],
             ops._s())
    return out
