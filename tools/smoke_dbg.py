import sys; sys.path.insert(0,'/root/repo')
import torch
import cmda_amd.runtime as rt
from cmda_amd import segmentors, backbones, decode_heads
from cmda_amd.registry import build_segmentor
from oracle import head as ohd, mit as omit, segmentor as oseg
torch.manual_seed(0)
dev = torch.device('cuda:0')
depths = [1, 1, 1, 1]
cfg = dict(type='EncoderDecoder',
           backbone=dict(type='MixVisionTransformer', embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8],
                         qkv_bias=True, depths=depths, sr_ratios=[8, 4, 2, 1], drop_path_rate=0.0),
           decode_head=dict(type='DAFormerHead', in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256,
                            dropout_ratio=0.0, num_classes=19, norm_cfg=dict(type='BN'), align_corners=False,
                            decoder_params=dict(embed_dims=256, embed_cfg=dict(type='mlp'), embed_neck_cfg=dict(type='mlp'),
                                                fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False))))
model = build_segmentor(cfg)
ref = oseg.EncoderDecoder(omit.MixVisionTransformer(depths=depths, drop_path_rate=0.0, eps=1e-5), ohd.DAFormerHead(dropout_ratio=0.0))
ref.load_state_dict(model.state_dict())
model.to(dev).train(); ref.train()
rt.set_compute_dtype(torch.float32)
img = torch.randn(1, 3, 64, 64); gt = torch.randint(0, 19, (1, 1, 64, 64))
losses, logits = model.forward_train(img.to(dev), None, gt.to(dev))
losses['decode.loss_seg'].backward()
rl, rlog = ref.forward_train(img, gt); rl['decode.loss_seg'].backward()
torch.cuda.synchronize()
errs=[]
for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
    d=(p.grad.cpu()-q.grad).abs()
    errs.append((d.max().item()/(q.grad.abs().max().item()+1e-12), n, (d>1e-3*q.grad.abs().max()).float().mean().item()))
errs.sort(reverse=True)
for e in errs[:10]: print(f'{e[0]:.3e} frac_bad={e[2]:.4f} {e[1]}')
