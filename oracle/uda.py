"""ORACLE (test infrastructure): the self-training pieces of DACS and the two "extractors".

Follows
  uda/dacs.py            _init_ema_weights/_update_ema :250-272; teacher pseudo-labels :674-682,701-711;
                         mixing loop :716-771 (ISR of the mixed image :729-744)
  models/utils/dacs_transforms.py  get_class_masks :101-112, generate_class_mask :115-119, one_mix :122-131,
                         denorm :52-53, get_mean_std :38-49
  datasets/utils.py      tensor_normalize_to_range :10-14, get_ic :87-105, get_image_change_from_pil :108-152
  datasets/dsec.py       events_to_voxel_grid :26-70, events_norm :80-121
PIL's Image.convert('L') (third-party, Pillow 8.3.1 pinned by requirements.txt) is restated as its published
integer formula L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.  Pinned by tests/golden.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

IMG_MEAN = (123.675, 116.28, 103.53)
IMG_STD = (58.395, 57.12, 57.375)


# ---------------------------------------------------------------- EMA teacher
def ema_alpha(it, alpha):
    return min(1 - 1 / (it + 1), alpha)


def update_ema(ema_params, params, it, alpha):
    """it == 0: copy; it > 0: ema = a*ema + (1-a)*p with a = min(1 - 1/(it+1), alpha).  Parameters only."""
    with torch.no_grad():
        if it == 0:
            for e, p in zip(ema_params, params):
                e.copy_(p)
            return
        a = ema_alpha(it, alpha)
        for e, p in zip(ema_params, params):
            e.copy_(a * e + (1 - a) * p)


# ---------------------------------------------------------------- pseudo labels
def upsample_exact(logits, size):
    """Bilinear align_corners=False written as individually rounded fp32 ops, in the order the HIP kernels use:
    ((v00*wx0 + v01*wx1)*wy0) + ((v10*wx0 + v11*wx1)*wy1).  Equals F.interpolate up to fp contraction."""
    B, C, h, w = logits.shape
    H, W = size

    def taps(n_in, n_out):
        if n_in == n_out:
            i = torch.arange(n_out)
            return i, i, torch.ones(n_out), torch.zeros(n_out)
        scale = np.float32(n_in) / np.float32(n_out)
        real = torch.arange(n_out, dtype=torch.float32).add(0.5).mul(float(scale)).sub(0.5).clamp_min(0.0)
        i0 = real.floor().long().clamp_max(n_in - 1)
        lam = (real - i0.float()).clamp(0.0, 1.0)
        i1 = i0 + (i0 < n_in - 1).long()
        return i0, i1, 1.0 - lam, lam

    y0, y1, wy0, wy1 = taps(h, H)
    x0, x1, wx0, wx1 = taps(w, W)
    r0, r1 = logits[:, :, y0], logits[:, :, y1]
    top = r0[..., x0] * wx0 + r0[..., x1] * wx1
    bot = r1[..., x0] * wx0 + r1[..., x1] * wx1
    return top * wy0[:, None] + bot * wy1[:, None]


def pseudo_labels(fusion_logits_lowres, size, threshold, ignore_top=0, ignore_bottom=0, exact=False):
    """softmax -> max over the up-sampled teacher logits; pseudo_weight = mean(prob >= thr) as one scalar per batch."""
    up = upsample_exact(fusion_logits_lowres, size) if exact else F.interpolate(
        fusion_logits_lowres, size=size, mode='bilinear', align_corners=False)
    prob, label = torch.softmax(up, dim=1).max(dim=1)
    count = int((prob >= threshold).sum())
    w = count / label.numel()
    weight = w * torch.ones(prob.shape)
    if ignore_top > 0:
        weight[:, :ignore_top, :] = 0
    if ignore_bottom > 0:
        weight[:, -ignore_bottom:, :] = 0
    return label, prob, weight, count


def pseudo_labels_fullres(fusion_logits, threshold, ignore_top=0, ignore_bottom=0):
    """dacs.py:680-682,701-711 on the already up-sampled teacher logits [B,nc,H,W] (what encode_decode returns)."""
    prob, label = torch.softmax(fusion_logits, dim=1).max(dim=1)
    count = int((prob >= threshold).sum())
    weight = (count / label.numel()) * torch.ones(prob.shape)
    if ignore_top > 0:
        weight[:, :ignore_top, :] = 0
    if ignore_bottom > 0:
        weight[:, -ignore_bottom:, :] = 0
    return label, prob, weight, count


# ---------------------------------------------------------------- ClassMix
def choose_classes(labels, rng):
    """get_class_masks' class draw: unique over the WHOLE batch tensor, ceil(n/2) classes per sample."""
    out = []
    classes = torch.unique(labels)
    n = classes.shape[0]
    for _ in range(labels.shape[0]):
        pick = rng.choice(n, int((n + n % 2) / 2), replace=False)
        out.append(classes[torch.as_tensor(pick).long()])
    return out


def class_mask(label, classes):
    """label [1,H,W] (one sample of the [B,1,H,W] batch), classes [K] -> mask [1,H,W] of 0/1 (int64)."""
    return (label == classes[:, None, None]).sum(0, keepdim=True)


def one_mix(mask, a, b):
    return mask * a + (1 - mask) * b


def denorm(img, mean=IMG_MEAN, std=IMG_STD):
    m = torch.tensor(mean).view(1, 3, 1, 1)
    s = torch.tensor(std).view(1, 3, 1, 1)
    return img.mul(s).add(m) / 255.0


# ---------------------------------------------------------------- Image Content-Extractor (ISR)
def pil_luma(rgb_u8):
    """PIL 'RGB' -> 'L' (ITU-R 601-2, 16-bit fixed point with rounding)."""
    r, g, b = (rgb_u8[..., i].astype(np.uint32) for i in range(3))
    return ((19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16).astype(np.uint8)


def _normalize_to_range(t, lo, hi):
    tmin, tmax = t.min(), t.max()
    return (t - tmin) / (tmax - tmin + 1e-8) * (hi - lo) + lo


def get_ic(front, now, val_range, threshold, clip_range):
    front = np.log(np.asarray(front, dtype=np.float32) / 255 * (val_range[1] - val_range[0]) + val_range[0])
    now = np.log(np.asarray(now, dtype=np.float32) / 255 * (val_range[1] - val_range[0]) + val_range[0])
    d = torch.from_numpy(now - front)[None]
    span = np.log(val_range[1]) - np.log(val_range[0])
    thr, clip = span * threshold, span * clip_range
    d = torch.where(d.abs() <= thr, torch.zeros_like(d), d)
    pos = _normalize_to_range(d.clamp(0, clip), 0, 1)
    neg = _normalize_to_range(d.clamp(-clip, 0), -1, 0)
    return pos + neg


def image_change(gray_u8, shift_pixel, val_range, threshold, clip_range, shift_direction='rightdown'):
    """get_image_change_from_pil after the 'L' conversion; gray_u8 [H,W] uint8 -> [1,H,W] float in [-1,1]."""
    g = gray_u8
    H, W = g.shape
    s = shift_pixel

    def row(direction):
        if direction == 'left':
            return np.concatenate((g[:, s:], g[:, W - s:]), axis=1)
        return np.concatenate((g[:, :s], g[:, :W - s]), axis=1)

    def col(direction):
        if direction == 'up':
            return np.concatenate((g[s:, :], g[H - s:, :]), axis=0)
        return np.concatenate((g[:s, :], g[:H - s, :]), axis=0)

    kw = dict(val_range=val_range, threshold=threshold, clip_range=clip_range)
    if shift_direction == 'all':
        parts = [get_ic(g, col('up'), **kw), get_ic(g, row('left'), **kw), get_ic(g, col('down'), **kw), get_ic(g, row('right'), **kw)]
        return parts[0] / 4 + parts[1] / 4 + parts[2] / 4 + parts[3] / 4
    r = row('left' if 'left' in shift_direction else 'right')
    c = col('up' if 'up' in shift_direction else 'down')
    return get_ic(g, r, **kw) / 2 + get_ic(g, c, **kw) / 2


def mixed_image_to_isr(mixed_img, shift_pixel, val_range, threshold, clip_range, shift_direction):
    """dacs.py:729-744 for one sample: normalised image [1,3,H,W] -> ISR [1,3,H,W]."""
    u8 = np.uint8(np.transpose((torch.clamp(denorm(mixed_img), 0, 1) * 255).numpy()[0], (1, 2, 0)))
    isr = image_change(pil_luma(u8), shift_pixel, val_range, threshold, clip_range, shift_direction)
    return isr.repeat(3, 1, 1)[None]


def random_shift_direction(color_jitter_u):
    direct = [['leftdown', 'leftup'], ['rightdown', 'rightup']]
    return direct[int(color_jitter_u * 10) % 2][int(color_jitter_u * 100) % 2]


# ---------------------------------------------------------------- event voxel grid
def events_to_voxel_grid(time, x, y, pol, width, height, num_bins):
    grid = torch.zeros(num_bins * height * width, dtype=torch.float32)
    C, H, W = num_bins, height, width
    t_norm = (C - 1) * (time - time[0]) / (time[-1] - time[0])
    x0, y0, t0 = x.int(), y.int(), t_norm.int()
    value = 2 * pol - 1
    for xl in (x0, x0 + 1):
        for yl in (y0, y0 + 1):
            for tl in (t0, t0 + 1):
                ok = (xl < W) & (xl >= 0) & (yl < H) & (yl >= 0) & (tl >= 0) & (tl < C)
                wgt = value * (1 - (xl - x).abs()) * (1 - (yl - y).abs()) * (1 - (tl - t_norm).abs())
                idx = H * W * tl.long() + W * yl.long() + xl.long()
                grid.put_(idx[ok], wgt[ok], accumulate=True)
    return grid.view(C, H, W)


def events_norm(events, clip_range=1.0, final_range=1.0):
    """events_norm(..., enforce_no_events_zero=True) as the DSEC loader calls it."""
    nz = events != 0
    n = nz.sum()
    if n > 0:
        mean = events.sum() / n
        std = torch.sqrt((events ** 2).sum() / n - mean ** 2)
        events = nz.float() * (events - mean) / (std + 1e-8)
    pos = _normalize_to_range(events.clamp(0, clip_range), 0, final_range)
    neg = _normalize_to_range(events.clamp(-clip_range, 0), -final_range, 0)
    return pos + neg


# ---------------------------------------------------------------- optimiser / schedule (mmcv 1.3.7 semantics)
def poly_warm_lr(base_lr, it, max_iters=40000, power=1.0, min_lr=0.0, warmup_iters=1500, warmup_ratio=1e-6):
    lr = (base_lr - min_lr) * (1 - it / max_iters) ** power + min_lr
    if it < warmup_iters:
        k = (1 - it / warmup_iters) * (1 - warmup_ratio)
        lr = lr * (1 - k)
    return lr


def param_group_options(name, base_lr, base_wd, custom_keys):
    """DefaultOptimizerConstructor: first matching key in sorted(sorted(keys), key=len, reverse=True)."""
    lr, wd = base_lr, base_wd
    for key in sorted(sorted(custom_keys.keys()), key=len, reverse=True):
        if key in name:
            lr = base_lr * custom_keys[key].get('lr_mult', 1.0)
            wd = base_wd * custom_keys[key].get('decay_mult', 1.0)
            break
    return lr, wd


# ---------------------------------------------------------------- strong augmentation (kornia 0.5.8, restated; unpinned)
def _rgb_to_hsv(img):
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    mx, mn = img.max(1)[0], img.min(1)[0]
    d = mx - mn
    s = d / (mx + 1e-6)
    dd = torch.where(d == 0, torch.ones_like(d), d)
    h = torch.where(mx == r, (g - b) / dd, torch.where(mx == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) / 6.0
    h = h - torch.floor(h)
    h = torch.where(d == 0, torch.zeros_like(h), h * (2 * math.pi))
    return h, s, mx


def _hsv_to_rgb(h, s, v):
    hp = h / (2 * math.pi) * 6.0
    hi = torch.floor(hp)
    f = hp - hi
    i = hi.long() % 6
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    sel = lambda *c: torch.stack(c, 0).gather(0, i[None])[0]
    return torch.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)], 1)


def color_jitter(img, order, fb, fc, fs, fh, mean=IMG_MEAN, std=IMG_STD):
    """img: normalised [B,3,H,W]; ops 0..3 = brightness, contrast, saturation, hue applied in `order`."""
    m = torch.tensor(mean).view(1, 3, 1, 1)
    sd = torch.tensor(std).view(1, 3, 1, 1)
    x = (img * sd + m) / 255.0
    for op in order:
        if op == 0:
            x = (x + (fb - 1.0)).clamp(0, 1)
        elif op == 1:
            x = (x * fc).clamp(0, 1)
        else:
            h, s, v = _rgb_to_hsv(x)
            if op == 2:
                s = (s * fs).clamp(0, 1)
            else:
                h = torch.fmod(h + fh * 2 * math.pi, 2 * math.pi)
                h = torch.where(h < 0, h + 2 * math.pi, h)
            x = _hsv_to_rgb(h, s, v)
    return (x * 255.0 - m) / sd


def gaussian_blur(img, k, sigma):
    x = torch.arange(k, dtype=torch.float32) - k // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    g = g / g.sum()
    C = img.shape[1]
    pad = k // 2
    y = F.pad(img, (pad, pad, 0, 0), mode='reflect')
    y = F.conv2d(y, g.view(1, 1, 1, k).repeat(C, 1, 1, 1), groups=C)
    y = F.pad(y, (0, 0, pad, pad), mode='reflect')
    return F.conv2d(y, g.view(1, 1, k, 1).repeat(C, 1, 1, 1), groups=C)


def gaussian_blur_hw(img, ky, kx, sigma):
    """GaussianBlur2d(kernel_size=(ky, kx), sigma=(sigma, sigma)), border 'reflect' (dacs_transforms.py:82-98: ky from the
    image height, kx from its width)."""
    def taps(k):
        x = torch.arange(k, dtype=torch.float32) - k // 2
        g = torch.exp(-x * x / (2.0 * sigma * sigma))
        return g / g.sum()
    C = img.shape[1]
    gx, gy = taps(kx), taps(ky)
    y = F.pad(img, (kx // 2, kx // 2, 0, 0), mode='reflect')
    y = F.conv2d(y, gx.view(1, 1, 1, kx).repeat(C, 1, 1, 1), groups=C)
    y = F.pad(y, (0, 0, ky // 2, ky // 2), mode='reflect')
    return F.conv2d(y, gy.view(1, 1, ky, 1).repeat(C, 1, 1, 1), groups=C)


def blur_kernel_size(n):
    return int(np.floor(np.ceil(0.1 * n) - 0.5 + np.ceil(0.1 * n) % 2))
