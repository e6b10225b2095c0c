"""Kernel-level parity of the HBM-bound kernels against plain torch fp32 (emu on CPU, real lib on GPU)."""
import pytest
import torch
import torch.nn.functional as F

from cmda_amd import ops
from conftest import assert_close, check_le

DT = [(torch.float32, 3e-5), (torch.bfloat16, 1.6e-2)]


@pytest.mark.parametrize('dt,tol', DT)
@pytest.mark.parametrize('C', [32, 64, 128, 160, 320, 1024])
@pytest.mark.parametrize('rows', [37, 9001])
def test_layernorm(tgt, dt, tol, C, rows):
    if rows > 100 and C not in (64, 160):
        pytest.skip('large-row case only for one sub-wave and one full-wave width')
    torch.manual_seed(C)
    x = torch.randn(rows, C).to(dt)
    g, b = torch.randn(C), torch.randn(C)
    dy, dres = torch.randn(rows, C).to(dt), torch.randn(rows, C).to(dt)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    ref.backward(dy.float())
    xd, gd, bd, dyd, dresd = map(tgt.to, (x, g, b, dy, dres))
    y, mean, rstd = ops.layernorm_fwd(xd, gd, bd, 1e-6)
    assert_close(y, ref, tol, name='ln fwd')
    dg, db = torch.zeros(C, device=tgt.device), torch.zeros(C, device=tgt.device)
    dx = ops.layernorm_bwd(dyd, xd, gd, mean, rstd, dg, db, dres=dresd)
    assert_close(dx, xr.grad + dres.float(), tol, name='ln dx')
    assert_close(dg, gr.grad, 2e-5, name='ln dgamma')
    assert_close(db, br.grad, 2e-5, name='ln dbeta')


@pytest.mark.parametrize('dt,tol', DT)
def test_permute_cast_colsum_axpby(tgt, dt, tol):
    torch.manual_seed(0)
    w = torch.randn(6, 5, 3, 3)
    wd = tgt.to(w)
    out = torch.empty(6, 3, 3, 5, dtype=dt, device=tgt.device)
    ops.permute4(wd, out, (6, 5, 3, 3), (0, 2, 3, 1))
    # (bf16: ONE round-to-nearest-even of the fp32 value, i.e. at most 2^-8 of the largest element -- a bound, not a margin)
    assert_close(out, w.permute(0, 2, 3, 1), 4e-3 if dt == torch.bfloat16 else 0, name='permute')
    out = torch.empty(5, 3, 3, 6, dtype=dt, device=tgt.device)
    ops.permute4(wd, out, (6, 5, 3, 3), (1, 2, 3, 0), flipmask=0b1100)
    assert_close(out, w.flip(2, 3).permute(1, 2, 3, 0), 4e-3 if dt == torch.bfloat16 else 0, name='permute+flip')
    g = torch.randn(6, 3, 3, 5)
    acc = tgt.to(w.clone())
    ops.permute4(tgt.to(g), acc, (6, 3, 3, 5), (0, 3, 1, 2), accumulate=True)
    assert_close(acc, w + g.permute(0, 3, 1, 2), 1e-6, name='permute accumulate')
    x = torch.randn(1000, 52).to(dt)
    s = torch.zeros(52, device=tgt.device)
    ops.colsum(tgt.to(x), s, 1000, 52)
    assert_close(s, x.float().sum(0), 1e-5, name='colsum')
    a, b = torch.randn(1003).to(dt), torch.randn(1003).to(dt)
    o = ops.axpby(tgt.to(a), tgt.to(b), 0.5, 0.5)
    assert_close(o, 0.5 * a.float() + 0.5 * b.float(), tol, name='axpby')


@pytest.mark.parametrize('dt,tol', DT)
@pytest.mark.parametrize('L', [24, 256, 280])
def test_softmax(tgt, dt, tol, L):
    torch.manual_seed(L)
    rows = 19
    s = (torch.randn(rows, L) * 3).to(dt)
    sr = s.float().requires_grad_(True)
    ref = torch.softmax(0.125 * sr, -1)
    dp = torch.randn(rows, L).to(dt)
    ref.backward(dp.float())
    p = ops.softmax_fwd_(tgt.to(s.clone()), rows, L, 0.125)
    assert_close(p, ref, tol, name='softmax fwd')
    ds = ops.softmax_bwd_(tgt.to(ref.detach().to(dt)), tgt.to(dp.clone()), rows, L, 0.125)
    assert_close(ds, sr.grad, tol, atol=2e-4 if dt == torch.bfloat16 else 0, name='softmax bwd')


@pytest.mark.parametrize('dt,tol', DT)
@pytest.mark.parametrize('dil,act,shape', [(1, 'gelu', (2, 13, 9, 24)), (1, None, (2, 13, 9, 24)), (6, None, (2, 13, 9, 24)),
                                           (1, 'gelu', (1, 5, 40, 260)), (3, None, (2, 7, 43, 8)), (18, None, (1, 20, 24, 12)),
                                           (12, None, (1, 3, 128, 4))])
def test_dwconv(tgt, dt, tol, dil, act, shape):
    torch.manual_seed(dil)
    B, H, W, C = shape
    x = torch.randn(B, H, W, C).to(dt)
    w, b = torch.randn(C, 1, 3, 3) * 0.3, torch.randn(C)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = F.conv2d(xr, wr, br if act else None, padding=dil, dilation=dil, groups=C)
    ref = F.gelu(z) if act == 'gelu' else z
    dy = torch.randn(B, H, W, C).to(dt)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    xd, wd, bd, dyd = tgt.to(x), tgt.to(w.view(C, 9).t().contiguous()), tgt.to(b) if act else None, tgt.to(dy)  # tap-major [9,C]
    y = ops.dwconv_fwd(xd, wd, bd, B, H, W, C, dil, act)
    assert_close(y, ref.permute(0, 2, 3, 1), tol, name='dw fwd')
    if dil >= 2 and act is None:   # the BatchNorm statistics of the output taken on the way (groups of images: cmda_dwconv3x3_fwd_stats)
        from cmda_amd import _lib as L
        for ipg in (1, B):
            G = B // ipg
            ws = torch.zeros(G * int(L.lib().cmda_bn_ws_floats(C)), device=tgt.device)
            y2 = ops.dwconv_fwd(xd, wd, None, B, H, W, C, dil, None, colstats=(ws, ipg))
            assert torch.equal(y2, y)
            st = ws.view(G, 33, 2, C)[:, :32].sum(1).cpu()
            zr = z.detach().permute(0, 2, 3, 1).reshape(G, ipg * H * W, C)
            assert_close(st[:, 0], zr.sum(1), tol, atol=tol * (ipg * H * W) ** 0.5 * 4, name='dw output column sums')
            assert_close(st[:, 1], (zr * zr).sum(1), tol * 2, name='dw output column sums of squares')
    dz = ops.dwconv_gelu_bwd_prep(xd, wd, bd, dyd, B, H, W, C, dil) if act == 'gelu' else dyd
    dx = ops.dwconv_bwd_data(dz, wd, B, H, W, C, dil)
    assert_close(dx, xr.grad.permute(0, 2, 3, 1), tol * 2, name='dw dx')
    prev = torch.randn(B, H, W, C).to(dt)   # accumulate=True: dx += (the sep-ASPP branches add into one input gradient)
    acc = ops.dwconv_bwd_data(dz, wd, B, H, W, C, dil, out=tgt.to(prev.clone()), accumulate=True)
    assert_close(acc, xr.grad.permute(0, 2, 3, 1) + prev.float(), tol * 3, name='dw dx accumulate')
    dw = torch.zeros(C, 9, device=tgt.device)
    db = torch.zeros(C, device=tgt.device) if act else None
    ops.dwconv_bwd_weight(dz, xd, dw, db, B, H, W, C, dil)
    assert_close(dw, wr.grad.view(C, 9), tol, name='dw dweight')
    if act:
        assert_close(db, br.grad, tol, name='dw dbias')
    if act == 'gelu':   # the fused form of the two calls above (one pass over x / da)
        dw2, db2 = torch.zeros(C, 9, device=tgt.device), torch.zeros(C, device=tgt.device)
        dz2 = ops.dwconv_gelu_bwd_fused(xd, wd, bd, dyd, dw2, db2, B, H, W, C, dil)
        assert_close(dz2, dz, 1e-6 if dt == torch.float32 else 4e-3, name='fused dz')
        assert_close(dw2, wr.grad.view(C, 9), tol, name='fused dweight')
        assert_close(db2, br.grad, tol, name='fused dbias')


@pytest.mark.parametrize('dt,tol', DT)
@pytest.mark.parametrize('IH,IW,OH,OW', [(8, 8, 32, 32), (4, 6, 32, 48), (16, 16, 32, 32), (14, 20, 110, 160), (9, 7, 9, 7)])
def test_bilinear(tgt, dt, tol, IH, IW, OH, OW):
    torch.manual_seed(IH)
    B, C, ld, coff = 2, 8, 24, 12
    x = torch.randn(B, IH, IW, C).to(dt)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.interpolate(xr, size=(OH, OW), mode='bilinear', align_corners=False)
    dy = torch.randn(B, OH, OW, ld).to(dt)
    ref.backward(dy.float()[..., coff:coff + C].permute(0, 3, 1, 2))
    y = torch.zeros(B, OH, OW, ld, dtype=dt, device=tgt.device)
    ops.bilinear_fwd(tgt.to(x), y, B, IH, IW, OH, OW, C, ld, coff)
    assert_close(y[..., coff:coff + C], ref.permute(0, 2, 3, 1), tol if dt == torch.bfloat16 else 2e-6, name='resize fwd')
    assert y[..., :coff].abs().max().item() == 0 and y[..., coff + C:].abs().max().item() == 0
    dx = torch.empty(B, IH, IW, C, dtype=dt, device=tgt.device)
    ops.bilinear_bwd(tgt.to(dy), dx, B, IH, IW, OH, OW, C, ld, coff)
    assert_close(dx, xr.grad.permute(0, 2, 3, 1), tol, name='resize bwd')


@pytest.mark.parametrize('dt,tol', DT)
@pytest.mark.parametrize('relu', [True, False])
def test_batchnorm(tgt, dt, tol, relu):
    torch.manual_seed(3)
    M, C, ld, coff = 600, 36, 48, 8
    x = (torch.randn(M, C) * 2 + 3).to(dt)
    g, b = torch.rand(C) + 0.5, torch.randn(C) * 0.2
    rm, rv = torch.randn(C), torch.rand(C) + 0.5
    bn = torch.nn.BatchNorm1d(C)
    bn.weight.data.copy_(g), bn.bias.data.copy_(b), bn.running_mean.copy_(rm), bn.running_var.copy_(rv)
    xr = x.float().requires_grad_(True)
    ref = bn(xr)
    ref = F.relu(ref) if relu else ref
    dy = torch.randn(M, ld).to(dt)
    ref.backward(dy.float()[:, coff:coff + C])
    xd, gd, bd, rmd, rvd = map(tgt.to, (x, g, b, rm.clone(), rv.clone()))
    y = torch.zeros(M, ld, dtype=dt, device=tgt.device)
    mean, rstd = ops.bn_train_fwd(xd, gd, bd, y, rmd, rvd, M, C, 1e-5, 0.1, relu, ld, coff)
    assert_close(y[:, coff:coff + C], ref, tol, name='bn fwd')
    assert_close(rmd, bn.running_mean, 1e-5, name='running_mean')
    assert_close(rvd, bn.running_var, 1e-5, name='running_var')
    dg, db = torch.zeros(C, device=tgt.device), torch.zeros(C, device=tgt.device)
    dx = ops.bn_train_bwd(tgt.to(dy), xd, mean, rstd, gd, bd, dg, db, M, C, relu, ld, coff)
    assert_close(dx, xr.grad, tol * 2, name='bn dx')
    assert_close(dg, bn.weight.grad, 1e-4, name='bn dgamma')
    assert_close(db, bn.bias.grad, 1e-4, name='bn dbeta')


@pytest.mark.parametrize('h,w,H,W', [(8, 8, 32, 32), (6, 10, 24, 40), (7, 5, 28, 20), (8, 8, 8, 8), (5, 7, 13, 18), (2, 3, 32, 48), (9, 17, 36, 68),
                                     (12, 20, 24, 40), (10, 9, 30, 27), (32, 40, 128, 160), (11, 13, 66, 65)])   # (tiled backward: factors 2 / 3 / 4 / 5-6, several tiles)
@pytest.mark.parametrize('use_weight', [True, False])
def test_ce_upsample(tgt, h, w, H, W, use_weight):
    torch.manual_seed(h * 3 + w)
    B, nc = 2, (19 if (h + w) % 3 else 7)   # (19: the unrolled instance of the tiled backward; 7: its run-time class count)
    logits = torch.randn(B, h, w, nc) * 2
    label = torch.randint(0, nc, (B, H, W))
    label[torch.rand(B, H, W) < 0.1] = 255
    weight = torch.rand(B, H, W) if use_weight else None
    lr = logits.permute(0, 3, 1, 2).clone().requires_grad_(True)
    up = F.interpolate(lr, size=(H, W), mode='bilinear', align_corners=False)
    loss_px = F.cross_entropy(up, label, reduction='none', ignore_index=255)
    if use_weight:
        loss_px = loss_px * weight
    loss = loss_px.mean()
    (loss * 0.7).backward()
    acc_ref = (up.argmax(1) == label).float().sum()
    acc, lse = ops.ce_upsample_fwd(tgt.to(logits), tgt.to(label), tgt.to(weight), H, W)
    n = B * H * W
    assert_close(acc[0] / n, loss, 2e-6, name='ce loss')
    assert acc[1].item() == acc_ref.item()
    gs = tgt.to(torch.tensor([0.7]))
    dl = ops.ce_upsample_bwd(tgt.to(logits), tgt.to(label), tgt.to(weight), lse, gs, 1.0 / n, H, W)
    assert_close(dl, lr.grad.permute(0, 2, 3, 1), 2e-5, name='ce dlogits')


def test_pseudo_label(tgt):
    torch.manual_seed(5)
    B, h, w, H, W, nc = 2, 8, 12, 32, 48, 19
    logits = torch.randn(B, h, w, nc) * 4
    up = F.interpolate(logits.permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=False)
    prob_ref, lab_ref = torch.softmax(up, 1).max(1)
    lab, prob, cnt = ops.pseudo_label(tgt.to(logits), H, W, 0.968)
    top2 = up.topk(2, 1).values
    near_tie = (top2[:, 0] - top2[:, 1]) < 1e-5
    assert bool(((lab.cpu() == lab_ref) | near_tie).all())
    assert_close(prob, prob_ref, 1e-5, name='pseudo prob')
    assert abs(cnt.item() - int((prob_ref >= 0.968).sum())) <= 2
    wgt = ops.pseudo_weight(cnt, B, H, W, top=3, bottom=5)
    exp = torch.full((B, H, W), cnt.item() / (B * H * W))
    exp[:, :3] = 0
    exp[:, H - 5:] = 0
    assert_close(wgt, exp, 1e-7, name='pseudo weight')


def test_classmix_ema_adamw(tgt):
    torch.manual_seed(7)
    B, C, H, W = 2, 3, 8, 10
    src, tg = torch.randn(B, C, H, W), torch.randn(B, C, H, W)
    lab = torch.randint(0, 5, (B, H, W))
    classes = torch.tensor([[1, 3, -1], [0, 2, 4]])
    mask = torch.stack([(lab[i][None] == classes[i][classes[i] >= 0][:, None, None]).sum(0) for i in range(B)]).float()
    out = ops.class_mix(tgt.to(src), tgt.to(tg), tgt.to(lab), tgt.to(classes))
    assert_close(out, mask[:, None] * src + (1 - mask[:, None]) * tg, 0, name='classmix')
    plab = torch.randint(0, 19, (B, H, W))
    ol = ops.class_mix_label(tgt.to(lab), tgt.to(plab), tgt.to(lab), tgt.to(classes))
    assert torch.equal(ol.cpu(), (mask.long() * lab + (1 - mask.long()) * plab))
    p, e = torch.randn(1001), torch.randn(1001)
    ed = tgt.to(e.clone())
    ops.ema_update(ed, tgt.to(p), 0.9)
    assert_close(ed, 0.9 * e + 0.1 * p, 1e-6, name='ema')
    prm = torch.nn.Parameter(torch.randn(777))
    opt = torch.optim.AdamW([prm], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    pd = tgt.to(prm.data.clone())
    m, v = torch.zeros(777, device=tgt.device), torch.zeros(777, device=tgt.device)
    pb = torch.empty(777, dtype=torch.bfloat16, device=tgt.device)
    for step in (1, 2, 3):
        gr = torch.randn(777)
        prm.grad = gr.clone()
        opt.step()
        ops.adamw_step(pd, tgt.to(gr), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, step, p_bf16=pb)
    assert_close(pd, prm.data, 2e-6, name='adamw')
    assert_close(pb, prm.data, 4e-3, name='adamw bf16 copy')   # one rounding of the updated fp32 master: <= 2^-8 of the largest element


def _gold(name):
    import os
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(here, 'golden', name + '.npz')).items()}


def test_isr_golden(tgt):
    """On-device Image Content-Extractor vs the reference's get_image_change_from_pil outputs (tests/golden/isr.npz)."""
    from oracle import uda as ouda
    g = _gold('isr')
    img_u8 = g['img']  # [H,W,3] uint8
    # build the normalised image whose denorm*255 truncates back to img_u8 (as the step does with the mixed image)
    mean = torch.tensor(ouda.IMG_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ouda.IMG_STD).view(1, 3, 1, 1)
    x = ((img_u8.permute(2, 0, 1)[None].float() + 0.5) - mean) / std
    gray = ops.isr_gray(tgt.to(x.contiguous()))
    assert torch.equal(gray.cpu()[0], g['gray'])
    params = {'dsec': dict(val_range=[0.01, 1.01], threshold=0.005, clip_range=0.1, shift_pixel=1),
              'dz': dict(val_range=[1, 100], threshold=0.01, clip_range=0.1, shift_pixel=3)}
    for pn, p in params.items():
        for d in ('rightdown', 'rightup', 'leftdown', 'leftup', 'all'):
            out = ops.isr_from_gray(gray, shift_direction=d, **p)
            assert_close(out[0, 0:1], g[f'{pn}_{d}'], 2e-6, atol=2e-7, name=f'isr {pn} {d}')
            assert torch.equal(out[0, 0], out[0, 2])


def test_voxel_golden(tgt):
    g = _gold('voxel')
    for bins in (1, 5):
        vg = ops.events_to_voxel_grid(*[tgt.to(g[f'{k}{bins}']) for k in 'txyp'], bins, 48, 64)
        assert_close(vg, g[f'vg{bins}'], 2e-6, atol=1e-6, name='voxel grid')
        nrm = ops.events_norm(tgt.to(g[f'vg{bins}']), (5000 / 500000) * 1.5 * 100)
        assert_close(nrm, g[f'norm{bins}'], 1e-5, name='events_norm')


def test_strong_augmentation(tgt):
    """per-sample colour-jitter parameters and the (ky from H, kx from W) blur of dacs_transforms.py:64-98, plus the device-side
    on/off gates a captured launch sequence relies on"""
    from oracle import uda as ouda
    torch.manual_seed(9)
    img = torch.randn(2, 3, 40, 56)
    per_sample = [([2, 0, 3, 1], 1.1, 0.9, 1.15, -0.07), ([1, 3, 0, 2], 0.85, 1.2, 0.8, 0.11)]
    ref = torch.cat([ouda.color_jitter(img[i:i + 1], *per_sample[i]) for i in range(2)])
    prm = tgt.to(ops.jitter_params(per_sample))
    on, off = tgt.to(torch.ones(1, dtype=torch.int32)), tgt.to(torch.zeros(1, dtype=torch.int32))
    out = ops.color_jitter_(tgt.to(img.clone()), prm, on)
    assert_close(out, ref, 1e-4, atol=2e-4, name='color jitter', outlier_frac=1e-3, outlier_rtol=2.0)
    assert torch.equal(ops.color_jitter_(tgt.to(img.clone()), prm, off).cpu(), img), 'gate off must leave the image untouched'
    ky, kx = ouda.blur_kernel_size(40), ouda.blur_kernel_size(56)
    assert (ky, kx) == (ops.blur_kernel_size(40), ops.blur_kernel_size(56)) and ky != kx
    refb = ouda.gaussian_blur_hw(img, ky, kx, 0.8)
    tx, ty = tgt.to(ops.gaussian_taps(kx, 0.8)), tgt.to(ops.gaussian_taps(ky, 0.8))
    outb = ops.gaussian_blur_(tgt.to(img.clone()), tx, ty, on)
    assert_close(outb, refb, 1e-5, atol=1e-6, name='gaussian blur')
    assert torch.equal(ops.gaussian_blur_(tgt.to(img.clone()), tx, ty, off).cpu(), img), 'gate off must leave the image untouched'


def _attention_ref(q, kv, B, N, Nk, heads, C, scale):
    hd = C // heads
    qf = q.view(B, N, heads, hd).permute(0, 2, 1, 3)
    k = kv[:, :C].reshape(B, Nk, heads, hd).permute(0, 2, 1, 3)
    v = kv[:, C:].reshape(B, Nk, heads, hd).permute(0, 2, 1, 3)
    a = (qf @ k.transpose(-1, -2) * scale).softmax(-1)  # mix_transformer.py:97-99
    return (a @ v).permute(0, 2, 1, 3).reshape(B * N, C)


@pytest.mark.parametrize('B,N,Nk,heads', [(1, 64, 256, 1), (2, 200, 256, 2), (1, 70, 37, 1), (1, 300, 130, 5), (2, 520, 256, 1),
                                          (16, 4100, 256, 2)])
def test_fused_attention(tgt, B, N, Nk, heads):
    if B * N > 20000 and tgt.device.type != 'cuda':
        pytest.skip('the 128-queries-per-block launch shape only occurs on grids far too large for the emulator')
    """fused softmax(q k^T) v and its backward (probabilities recomputed in LDS) against autograd on the same bf16 inputs"""
    torch.manual_seed(N + Nk)
    C, scale = heads * 64, 0.125
    q, kv, do = torch.randn(B * N, C).bfloat16(), torch.randn(B * Nk, 2 * C).bfloat16(), torch.randn(B * N, C).bfloat16()
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = _attention_ref(qr, kvr, B, N, Nk, heads, C, scale)
    ref.backward(do.float())
    qd, kvd, dod = tgt.to(q), tgt.to(kv), tgt.to(do)
    assert ops.attention_fused_ok(qd, Nk, heads, C)
    o = ops.attention_fused_fwd(qd, kvd, B, N, Nk, heads, C, scale)
    assert_close(o, ref.detach(), 1.6e-2, name='attention o')
    dkv = torch.zeros(B * Nk, 2 * C, device=tgt.device)
    dq = ops.attention_fused_bwd(qd, kvd, dod, dkv, B, N, Nk, heads, C, scale)   # accumulating form: fp32 workspace + atomics
    assert_close(dq, qr.grad, 2e-2, name='attention dq')
    assert_close(dkv, kvr.grad, 2e-2, name='attention dkv')
    assert ops.attention_bwd_direct(B, N, Nk, heads) == (N <= 1024)
    if ops.attention_bwd_direct(B, N, Nk, heads):   # few queries: one block per key slice, dK | dV stored straight as bf16
        dkv16 = torch.full((B * Nk, 2 * C), float('nan'), dtype=torch.bfloat16, device=tgt.device)
        dq2 = ops.attention_fused_bwd(qd, kvd, dod, None, B, N, Nk, heads, C, scale, dkv16=dkv16)
        assert_close(dq2, qr.grad, 2e-2, name='attention dq (direct)')
        assert_close(dkv16, kvr.grad, 2e-2, name='attention dkv (direct)')


@pytest.mark.parametrize('B,N,Nk,heads', [(1, 64, 256, 1), (2, 200, 256, 2), (1, 70, 37, 1), (1, 300, 130, 5), (2, 1100, 256, 1)])
def test_fused_attention_split_bf16(tgt, B, N, Nk, heads):
    """the split-bf16 instances of the fused attention kernels (fp32 storage, three bf16 MFMAs per product: the tolerance-meeting
    mode's mix_transformer.py:97-101): forward, dq, dK | dV in the accumulating and the direct form, against fp32 autograd -- the
    error of ~16 mantissa bits per product, two orders below the bf16 kernels'"""
    if B * N > 1500 and tgt.device.type != 'cuda':
        pytest.skip('query counts beyond the direct mode: GPU only (emulator run time)')
    torch.manual_seed(N + Nk)
    C, scale = heads * 64, 0.125
    q, kv, do = torch.randn(B * N, C), torch.randn(B * Nk, 2 * C), torch.randn(B * N, C)
    qr, kvr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    ref = _attention_ref(qr, kvr, B, N, Nk, heads, C, scale)
    ref.backward(do)
    qd, kvd, dod = tgt.to(q), tgt.to(kv), tgt.to(do)
    assert ops.attention_fused_ok(qd, Nk, heads, C, x3=True) and not ops.attention_fused_ok(qd, Nk, heads, C)
    o = ops.attention_fused_fwd(qd, kvd, B, N, Nk, heads, C, scale)
    assert o.dtype == torch.float32
    assert_close(o, ref.detach(), 1e-4, name='split-bf16 attention o')
    dkv = torch.zeros(B * Nk, 2 * C, device=tgt.device)
    dq = ops.attention_fused_bwd(qd, kvd, dod, dkv, B, N, Nk, heads, C, scale)   # accumulating form: fp32 atomics
    assert_close(dq, qr.grad, 2e-4, name='split-bf16 attention dq')
    assert_close(dkv, kvr.grad, 2e-4, name='split-bf16 attention dkv')
    if ops.attention_bwd_direct(B, N, Nk, heads):   # few queries: one block per key slice stores dK | dV (fp32)
        dkv2 = torch.full((B * Nk, 2 * C), float('nan'), device=tgt.device)
        dq2 = ops.attention_fused_bwd(qd, kvd, dod, None, B, N, Nk, heads, C, scale, dkv16=dkv2)
        assert_close(dq2, qr.grad, 2e-4, name='split-bf16 attention dq (direct)')
        assert_close(dkv2, kvr.grad, 2e-4, name='split-bf16 attention dkv (direct)')
    # through the block-level composite: runtime.gemm_x3 routes fp32 storage to these kernels; same result as the unfused products
    from cmda_amd import nn as K
    import cmda_amd.runtime as rt
    rt.set_compute_dtype(torch.float32)
    rt.set_gemm_x3(True)
    try:
        o1, P1 = K.attention_fwd(qd, kvd, B, N, Nk, heads, C, scale)
        assert P1 is None
        dq1, dkv1 = K.attention_bwd(dod, qd, kvd, None, B, N, Nk, heads, C, scale)
        ops.ATTN_X3_OFF = True
        o2, P2 = K.attention_fwd(qd, kvd, B, N, Nk, heads, C, scale)
        assert P2 is not None
        dq2, dkv2 = K.attention_bwd(dod, qd, kvd, P2, B, N, Nk, heads, C, scale)
    finally:
        ops.ATTN_X3_OFF = False
        rt.set_gemm_x3(False)
    assert_close(o1, o2, 1e-4, name='fused vs unfused split-bf16 attention')
    assert_close(dq1, dq2, 2e-4, name='fused vs unfused split-bf16 dq')
    assert_close(dkv1, dkv2, 2e-4, name='fused vs unfused split-bf16 dkv')


@pytest.mark.parametrize('B,N,Nk,heads', [(1, 1120, 280, 2), (2, 280, 260, 5), (1, 70, 320, 8), (1, 300, 257, 1)])
def test_fused_attention_eval_keys(tgt, B, N, Nk, heads):
    """inference on 440 x 640 frames leaves 260 / 280 keys after the spatial reduction (encoder_decoder.py:897-936, mix_transformer.py
    sr_ratios): the forward-only instance of the fused kernel holds up to 320 keys; the training kernels stay at 256"""
    torch.manual_seed(N + Nk)
    C, scale = heads * 64, 0.125
    q, kv = torch.randn(B * N, C).bfloat16(), torch.randn(B * Nk, 2 * C).bfloat16()
    ref = _attention_ref(q.float(), kv.float(), B, N, Nk, heads, C, scale)
    qd, kvd = tgt.to(q), tgt.to(kv)
    assert ops.attention_fused_ok(qd, Nk, heads, C, need_grad=False) and not ops.attention_fused_ok(qd, Nk, heads, C)
    o = ops.attention_fused_fwd(qd, kvd, B, N, Nk, heads, C, scale)
    assert_close(o, ref, 1.6e-2, name='attention o (eval keys)')
    # and through the block: save=False (no backward to follow) takes the fused kernel, save=True the materialised path; same output
    from cmda_amd import nn as K
    import cmda_amd.runtime as rt
    rt.set_compute_dtype(torch.bfloat16)
    try:
        o1, P1 = K.attention_fwd(qd, kvd, B, N, Nk, heads, C, scale, need_grad=False)
        o2, P2 = K.attention_fwd(qd, kvd, B, N, Nk, heads, C, scale, need_grad=True)
    finally:
        rt.set_compute_dtype(torch.float32)
    assert P1 is None and P2 is not None
    assert_close(o1, o2.float(), 4e-2, name='fused vs materialised')   # (bf16 outputs one ulp apart: 7.8e-3 of range measured)


# ------------------------------------------------------------------ round-3 entry points, each against torch
@pytest.mark.parametrize('C,rows', [(64, 300), (320, 130), (512, 33), (128, 1000)])
def test_layernorm_fp32_stream_bf16_operands(tgt, C, rows):
    """cmda_layernorm_fwd2 / bwd2: the fp32 residual stream in, the bf16 GEMM operand out; backward: bf16 gradients in and out, the
    statistics recomputed from the fp32 input (mix_transformer.py:141-146 with x kept in fp32)"""
    torch.manual_seed(C + rows)
    x = torch.randn(rows, C) * 3 + 1
    g, b = torch.randn(C), torch.randn(C)
    dy, dres = torch.randn(rows, C).bfloat16(), torch.randn(rows, C).bfloat16()
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    ref.backward(dy.float())
    xd, gd, bd, dyd, dresd = map(tgt.to, (x, g, b, dy, dres))
    y, mean, rstd = ops.layernorm_fwd(xd, gd, bd, 1e-6, out_dtype=torch.bfloat16)
    assert y.dtype == torch.bfloat16
    assert_close(y, ref, 8e-3, name='ln fwd fp32 -> bf16')            # one rounding of the exact fp32 result
    assert_close(mean, x.mean(1), 1e-5, name='ln mean')
    dg, db = torch.zeros(C, device=tgt.device), torch.zeros(C, device=tgt.device)
    sc = torch.rand(4)
    rps = (rows + 3) // 4
    dx, dxs = ops.layernorm_bwd(dyd, xd, gd, mean, rstd, dg, db, dres=dresd, out_scale=tgt.to(sc), rows_per_scale=rps)
    assert dx.dtype == torch.bfloat16
    want = xr.grad + dres.float()
    assert_close(dx, want, 1.6e-2, name='ln dx (bf16 out)')
    assert_close(dxs, want * sc[torch.arange(rows) // rps, None], 1.6e-2, name='ln dx scaled')
    assert_close(dg, gr.grad, 2e-5, name='ln dgamma')
    assert_close(db, br.grad, 2e-5, name='ln dbeta')


@pytest.mark.parametrize('M,N,K', [(300, 64, 64), (130, 320, 1280), (1000, 128, 512)])
def test_gemm_fp32_residual_epilogue(tgt, M, N, K):
    """x1 = x + drop_path(proj(o)) with x and x1 in fp32 and the GEMM operands in bf16 (cmda_gemm_params_t.res_f32): the residual is
    added in fp32 in the epilogue, per-sample DropPath scale on the GEMM term only"""
    torch.manual_seed(M)
    a, w, bias = torch.randn(M, K).bfloat16(), (torch.randn(N, K) * 0.05).bfloat16(), torch.randn(N)
    res = torch.randn(M, N) * 50          # large against the GEMM term: a bf16 residual add would lose ~0.2 absolute here
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0])
    rps = (M + 3) // 4
    ref = res + (a.float() @ w.float().t() + bias) * sc[torch.arange(M) // rps, None]
    out = torch.empty(M, N, dtype=torch.float32, device=tgt.device)
    ad, wd = tgt.to(a), tgt.to(w)
    ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(wd, N, K), out, M, N, K, dtype=1, bias=tgt.to(bias), res=tgt.to(res),
             rowscale=tgt.to(sc), rows_per_scale=rps)
    err = (out.cpu() - ref).abs().max().item()
    check_le('fp32 residual epilogue: max abs err', err, 2e-3, strict=True)


@pytest.mark.parametrize('dt,tol', [(torch.bfloat16, 2e-3), (torch.float32, 2e-5)])
def test_conv_co1(tgt, dt, tol):
    """the generator's last layer, Conv2d(64, 1, 7, padding 3 reflect) + tanh (cyclegan_model.py:372-374), as a dot-product stencil
    (bf16: v_dot2; fp32 storage -- both parity modes --: plain FMAs over channel chunks)"""
    torch.manual_seed(3)
    B, H, W, C, K = 2, 19, 23, 64, 7
    x = torch.randn(B, C, H, W).to(dt)
    w = (torch.randn(1, C, K, K) * 0.05).to(dt)
    bias = torch.randn(1)
    for reflect in (True, False):
        xp = F.pad(x.float(), (3, 3, 3, 3), mode='reflect') if reflect else F.pad(x.float(), (3, 3, 3, 3))
        ref = torch.tanh(F.conv2d(xp, w.float(), bias))[:, 0]
        xd = tgt.to(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous())
        wd = tgt.to(w.permute(0, 2, 3, 1).reshape(-1).contiguous())
        assert ops.conv_co1_ok(xd, C, K, 3)
        out = ops.conv_co1(xd, wd, tgt.to(bias), B, H, W, C, K, 3, reflect, 'tanh')
        assert_close(out, ref, tol, name=f'conv_co1 reflect={reflect}')


def test_rows_fill_cast_pad_nchw_pad(tgt):
    torch.manual_seed(4)
    bias = torch.randn(20)
    out = ops.rows_fill(torch.full((37, 20), float('nan'), device=tgt.device), tgt.to(bias))
    assert torch.equal(out.cpu(), bias.expand(37, 20))
    out = ops.rows_fill(torch.full((5, 8), float('nan'), device=tgt.device), None)
    assert torch.equal(out.cpu(), torch.zeros(5, 8))
    src = torch.randn(33, 27)
    for dt in (torch.float32, torch.bfloat16):
        dst = ops.cast_pad_cols(tgt.to(src), 32, dt)
        assert dst.shape == (33, 32) and dst.dtype == dt
        assert torch.equal(dst[:, :27].cpu().float(), src.to(dt).float()) and not dst[:, 27:].cpu().float().abs().any()
    img = torch.randn(2, 3, 7, 9)
    for dt in (torch.float32, torch.bfloat16):
        dst = torch.full((2 * 63, 8), float('nan'), dtype=dt, device=tgt.device)
        ops.nchw_to_nhwc_pad(tgt.to(img), dst, 2, 3, 63, 8)
        want = torch.zeros(2, 7, 9, 8)
        want[..., :3] = img.permute(0, 2, 3, 1)
        assert torch.equal(dst.cpu().float(), want.view(-1, 8).to(dt).float())
