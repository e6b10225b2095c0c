"""On-device loader pipeline: the per-sample preprocessing the reference runs on CPU dataloader
workers with PIL / numpy / h5py, as batched device kernels behind the C ABI (csrc/pipeline.hip).

  target  mmseg/datasets/dsec.py:189-339  crop 400x400 -> flip -> PIL BILINEAR resize 512x512 -> ToTensor/Normalize; real-time ISR of the
                                          resized frame (get_image_change_from_pil, :256-263); :341-366 events: rectify gather ->
                                          voxel grid -> events_norm; :314-322 crop / flip / bilinear / x3
  source  mmseg/datasets/cityscapes_ic.py:147-210  PIL BILINEAR resize 2048x1024 -> 1024x512 -> crop 512 -> flip (image, time residual)
          create_cityscapes_image_change.py:16-35   the time residual of two consecutive frames (offline pre-step)

Pillow's resize (third-party, Pillow 8.3.1 pinned by requirements.txt; src/libImaging/Resample.c) is restated: `pil_coeffs`
builds the coefficient tables in double precision exactly as precompute_coeffs / normalize_coeffs_8bpc do; the kernels apply
them with Pillow's fixed-point arithmetic (22 fraction bits, the horizontal pass rounded to uint8 before the vertical one), so
uint8 results are bit-identical to PIL's (tests/test_pipeline.py checks against the installed Pillow).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib as L
from . import ops
from ._lib import c_f32, c_i32, c_i64, call, check_dev, ptr, stream_of

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # dsec.py:163 / cityscapes_ic.py: torchvision Normalize on [0,1] tensors
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2


def _declare():
    lib = L.lib()
    lib.cmda_pil_resize_u8.restype = ctypes.c_int
    return lib


def pil_coeffs(in_size, out_size):
    """Pillow precompute_coeffs(inSize, in0=0, in1=inSize, outSize, BILINEAR) + normalize_coeffs_8bpc.
    Returns (bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize], ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale            # BILINEAR support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            pre = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + pre * (1 << PRECISION_BITS)) if pre < 0 else int(0.5 + pre * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_COEFFS = {}


def _coeffs_dev(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _COEFFS.get(key)
    if t is None:
        b, k, ks = pil_coeffs(in_size, out_size)
        t = _COEFFS[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return t


def make_samp(B, device, src_x0=0, src_y0=0, flip_src=0, out_x0=0, out_y0=0, flip_out=0):
    """int32 [B,8] per-sample window / flip record (scalars broadcast, sequences per sample) -- see include/cmda_hip.h"""
    cols = [src_x0, src_y0, flip_src, out_x0, out_y0, flip_out, 0, 0]
    rows = [[int(c[b]) if hasattr(c, '__len__') else int(c) for c in cols] for b in range(B)]
    return torch.tensor(rows, dtype=torch.int32).to(device)


def pil_resize_u8(src, samp, in_size, res_size, out_size=None, want_u8=False, norm=None, want_gray=False, rep3=False):
    """PIL BILINEAR resize of uint8 HWC frames src [B,IH,IW,C].  in_size = (w, h) of the window the resize sees (at samp.src_*),
    res_size = (w, h) of the resized image, out_size = (w, h) of the window of it that is produced (default: all of it).
    norm = (mean3, std3) -> fp32 NCHW output (u8/255 - mean)/std.  Returns dict(u8=, f=, gray=)."""
    check_dev(src, samp)
    _declare()
    B, IH, IW, C = src.shape
    in_w, in_h = in_size
    OW, OH = out_size if out_size is not None else res_size
    hb, hk, hks = _coeffs_dev(in_w, res_size[0], src.device)
    vb, vk, vks = _coeffs_dev(in_h, res_size[1], src.device)
    tmp = torch.empty(B * in_h * OW * C, dtype=torch.uint8, device=src.device)
    out_u8 = torch.empty(B, OH, OW, C, dtype=torch.uint8, device=src.device) if want_u8 else None
    Cout = 3 if (C == 1 and rep3) else C
    out_f = torch.empty(B, Cout, OH, OW, dtype=torch.float32, device=src.device) if norm is not None else None
    gray = torch.empty(B, OH, OW, dtype=torch.uint8, device=src.device) if want_gray else None
    mean3 = (ctypes.c_float * 3)(*(norm[0] if norm is not None else (0, 0, 0)))
    std3 = (ctypes.c_float * 3)(*(norm[1] if norm is not None else (1, 1, 1)))
    call('cmda_pil_resize_u8', ptr(src), c_i32(B), c_i32(IH), c_i32(IW), c_i32(C), ptr(samp), c_i32(in_w), c_i32(in_h), ptr(hb),
         ptr(hk), c_i32(hks), ptr(vb), ptr(vk), c_i32(vks), c_i32(OW), c_i32(OH), ptr(tmp), ptr(out_u8), ptr(out_f), ptr(gray),
         std3, mean3, c_i32(int(rep3)), stream_of(src))
    return dict(u8=out_u8, f=out_f, gray=gray)


def luma_u8(rgb):
    """PIL Image.convert('L') of uint8 frames [..., 3] (interleaved RGB) -> uint8 [...]"""
    check_dev(rgb)
    out = torch.empty(rgb.shape[:-1], dtype=torch.uint8, device=rgb.device)
    call('cmda_luma_u8', ptr(rgb), ptr(out), c_i64(out.numel()), stream_of(rgb))
    return out


_TR_LUT = {}


def time_residual_u8(now, front, log_add=50.0, threshold=0.1, clip_range=0.8):
    """create_cityscapes_image_change.py:16-35 get_image_change(image_now, image_front) on uint8 'L' frames [B,H,W] -> uint8"""
    check_dev(now, front)
    B, H, W = now.shape
    key = (float(log_add), str(now.device))
    lut = _TR_LUT.get(key)
    if lut is None:
        lut = _TR_LUT[key] = torch.from_numpy(np.log(np.arange(256, dtype=np.float32) + np.float32(log_add))).to(now.device)
    mm = torch.empty(B * 4, dtype=torch.int32, device=now.device)
    out = torch.empty_like(now)
    call('cmda_time_residual_u8', ptr(now), ptr(front), ptr(lut), ptr(mm), ptr(out), c_i32(B), c_i32(H), c_i32(W),
         c_f32(threshold), c_f32(clip_range), stream_of(now))
    return out


def event_prep(t, x, y, p, rect_map=None, H=480, W=640):
    """dsec.py:341-353: t int64, x / y int32, p uint8 (device) -> (t_norm, x_rect, y_rect, pol) fp32"""
    check_dev(t, x, y, p, rect_map)
    N = t.numel()
    outs = [torch.empty(N, dtype=torch.float32, device=t.device) for _ in range(4)]
    call('cmda_event_prep', ptr(t), ptr(x), ptr(y), ptr(p), ptr(rect_map), c_i32(H), c_i32(W), *[ptr(o) for o in outs], c_i64(N),
         stream_of(t))
    return outs


def crop_flip_resize_f32(x, samp, crop, out_size, rep=1):
    """x fp32 [B,C,IH,IW]; crop = (w, h) window at samp.src_*; bilinear (align_corners=False) to out_size = (w, h)"""
    check_dev(x, samp)
    B, C, IH, IW = x.shape
    out = torch.empty(B, C * rep, out_size[1], out_size[0], dtype=torch.float32, device=x.device)
    call('cmda_crop_flip_resize_f32', ptr(x), ptr(out), ptr(samp), c_i32(B), c_i32(C), c_i32(IH), c_i32(IW), c_i32(crop[0]),
         c_i32(crop[1]), c_i32(out_size[1]), c_i32(out_size[0]), c_i32(rep), stream_of(x))
    return out


# ------------------------------------------------------------------------------------------------------------ composed pipelines
def dsec_target_sample(warp_u8, events, rect_map, x0, y0, flip, isr_parms, shift_direction='rightdown', crop=(400, 400),
                       out_size=(512, 512), events_bins=1, events_clip_range=None):
    """One batch of DSEC training samples (dsec.py:189-339 with outputs {'warp_image', 'events_vg', 'warp_img_self_res'},
    isr_type 'real_time', enforce_3_channels): warp_u8 uint8 [B,480,640,3]; events = list of B tuples (t i64, x i32, y i32, p u8);
    x0 / y0 / flip: per-sample crop origin and flip flag (the loader's random draws).  Returns the reference's target dict."""
    B = warp_u8.shape[0]
    dev = warp_u8.device
    samp = make_samp(B, dev, src_x0=x0, src_y0=y0, flip_src=flip)
    r = pil_resize_u8(warp_u8, samp, crop, out_size, norm=(IMAGENET_MEAN, IMAGENET_STD), want_gray=True)
    isr = ops.isr_from_gray(r['gray'], isr_parms['val_range'], isr_parms['_threshold'], isr_parms['_clip_range'],
                            isr_parms['shift_pixel'], shift_direction)
    H, W = warp_u8.shape[1], warp_u8.shape[2]
    grids = []
    for (t, x, y, p) in events:
        tn, xr, yr, pol = event_prep(t, x, y, p, rect_map, H, W)
        g = ops.events_to_voxel_grid(tn, xr, yr, pol, events_bins, H, W)
        n = t.numel()
        clip = events_clip_range if events_clip_range is not None else (n - 1) / 500000 * 1.5
        grids.append(ops.events_norm(g, clip))
    vg = torch.stack(grids)                                           # [B,bins,480,640]
    ev = crop_flip_resize_f32(vg, samp, crop, out_size, rep=3 if events_bins == 1 else 1)
    return dict(warp_image=r['f'], events_vg=ev, warp_img_self_res=isr)


def cityscapes_source_sample(frame_u8, prev_u8, x0, y0, flip, isr_parms, shift_direction='rightdown', resize=(1024, 512),
                             crop=(512, 512)):
    """One batch of source samples (cityscapes_ic.py:147-260 with outputs {'image', 'img_time_res', 'img_self_res'}) from raw
    frames: frame_u8 / prev_u8 uint8 [B,1024,2048,3] = the labelled frame and its predecessor in the sequence.  The time residual
    (create_cityscapes_image_change.py: 'L' frames -> log-difference -> uint8 PNG) is computed on the fly instead of being read
    from the pre-computed leftImg8bit_IC1 PNGs."""
    B, IH, IW, _ = frame_u8.shape
    dev = frame_u8.device
    samp = make_samp(B, dev, out_x0=x0, out_y0=y0, flip_out=flip)
    r = pil_resize_u8(frame_u8, samp, (IW, IH), resize, crop, norm=(IMAGENET_MEAN, IMAGENET_STD), want_gray=True)
    isr = ops.isr_from_gray(r['gray'], isr_parms['val_range'], isr_parms['_threshold'], isr_parms['_clip_range'],
                            isr_parms['shift_pixel'], shift_direction)
    tr = time_residual_u8(luma_u8(frame_u8), luma_u8(prev_u8))                          # uint8 'L' at full resolution
    t = pil_resize_u8(tr.view(B, IH, IW, 1), samp, (IW, IH), resize, crop, norm=((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)), rep3=True)
    return dict(image=r['f'], img_time_res=t['f'], img_self_res=isr)
