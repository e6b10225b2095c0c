#!/usr/bin/env python3
"""Where does the bf16 mode's logit error come from?  Full-depth MiT-B5 + DAFormerHead (train-mode BatchNorm, stochastic layers
off, seeded weights), one 512x512 batch: features and logits of the bf16 mode against the fp32 mode, the bf16 decoder alone on the
fp32 features, with and without the fp32 residual stream."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cmda_amd  # noqa: E402,F401
import cmda_amd.runtime as rt  # noqa: E402
from cmda_amd.registry import build_segmentor  # noqa: E402
from weights import seeded_fill, seeded_randn  # noqa: E402
from test_fullsize import DECODER, HEAD  # noqa: E402

dev = torch.device('cuda:0')
cfg = dict(type='EncoderDecoder', backbone=dict(type='mit_b5', style='pytorch', drop_path_rate=0.0),
           decode_head=dict(type='DAFormerHead', dropout_ratio=0.0, decoder_params=dict(DECODER), **HEAD))
B = 2
img = seeded_randn((B, 3, 512, 512), 5, 'img').to(dev)


def run(dtype, res32, feats_in=None):
    rt.set_compute_dtype(dtype)
    rt.set_residual_fp32(res32)
    model = build_segmentor(cfg)
    seeded_fill(model, 7)
    model.to(dev).train()
    with torch.no_grad():
        if feats_in is None:
            feats, _ = model.backbone.fwd(img, save=False)
        else:
            feats = [(f.to(dtype).contiguous(), h, w) for f, h, w in feats_in]
        logits, _ = model.decode_head.fwd(feats, B)
    torch.cuda.synchronize()
    return [(f.float(), h, w) for f, h, w in feats], logits.float()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


f32, l32 = run(torch.float32, False)
for res32 in (False, True):
    f16, l16 = run(torch.bfloat16, res32)
    fe = [rel(a[0], b[0]) for a, b in zip(f16, f32)]
    agree = (l16.argmax(-1) == l32.argmax(-1)).float().mean().item()
    print(f'bf16 (residual fp32 = {res32}): feature rel err per level {[f"{e:.2e}" for e in fe]}; logits rel err {rel(l16, l32):.2e}; argmax agreement {agree:.4f}')
_, lc = run(torch.bfloat16, True, feats_in=f32)
agree = (lc.argmax(-1) == l32.argmax(-1)).float().mean().item()
print(f'bf16 decoder alone on the fp32 features: logits rel err {rel(lc, l32):.2e}; argmax agreement {agree:.4f}')
srt = l32.sort(-1).values
margin = (srt[..., -1] - srt[..., -2]) / l32.abs().max()
for t in (1e-3, 5e-3, 1e-2, 2e-2):
    print(f'pixels whose top-2 margin is below {t:.0e} of the logit range: {(margin < t).float().mean().item():.4f}')
rt.set_compute_dtype(torch.float32)
