"""MFMA GEMM / implicit-GEMM conv (cmda_amd/csrc/gemm.hip) against plain torch fp32."""
import pytest
import torch
import torch.nn.functional as F

from cmda_amd import ops
from conftest import assert_close

# tag 2 = CMDA_F32X3: fp32 storage, split-bf16 (bf16 x 3) MFMA -- ~16 mantissa bits per product (gemm_x3.hip)
DT = [(torch.float32, 0, 2e-5), (torch.bfloat16, 1, 1.5e-2), (torch.float32, 2, 1e-4)]


@pytest.mark.parametrize('dt,tag,tol', DT)
@pytest.mark.parametrize('M,N,K', [(70, 50, 40), (130, 19, 147), (256, 128, 64), (33, 200, 8), (300, 260, 96),
                                   (300, 260, 128), (130, 19, 192), (700, 64, 320),
                                   (130, 64, 1024), (257, 96, 768)])   # >= 12 k-tiles on a resident grid: the 4-stage configurations
def test_gemm_layouts(tgt, dt, tag, tol, M, N, K):
    torch.manual_seed(M * 7 + N)
    a, b = torch.randn(M, K).to(dt), torch.randn(N, K).to(dt)
    bias, res = torch.randn(N), torch.randn(M, N).to(dt)
    ref = a.float() @ b.float().t()
    ad, bd, biasd, resd = map(tgt.to, (a, b, bias, res))
    out = torch.empty(M, N, dtype=dt, device=tgt.device)
    ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=tag)
    assert_close(out, ref, tol, name='NT')
    out = torch.empty(M, N, dtype=dt, device=tgt.device)
    ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=tag, bias=biasd, act='gelu', res=resd)
    assert_close(out, F.gelu(ref + bias) + res.float(), tol, name='NT+bias+gelu+res')
    out = torch.empty(M, N, dtype=dt, device=tgt.device)
    ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=tag, bias=biasd, act='relu')
    assert_close(out, F.relu(ref + bias), tol, name='NT+bias+relu')
    btd = tgt.to(b.t().contiguous())
    out = torch.empty(M, N, dtype=dt, device=tgt.device)
    ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(btd, K, N), out, M, N, K, b_kstrided=True, dtype=tag)
    assert_close(out, ref, tol, name='NN')
    atd = tgt.to(a.t().contiguous())
    out = torch.zeros(M, N, dtype=torch.float32, device=tgt.device)
    ops.gemm(ops.plain_view(atd, K, M), ops.plain_view(btd, K, N), out, M, N, K, a_kstrided=True, b_kstrided=True,
             dtype=tag, atomic=True, splits=3)
    assert_close(out, ref, 1e-5 if tag != 2 else tol, name='TN split-K atomic')
    out = torch.ones(M, N, dtype=dt, device=tgt.device)
    ops.gemm(ops.plain_view(atd, K, M), ops.plain_view(bd, N, K), out, M, N, K, a_kstrided=True, dtype=tag, beta=1.0)
    assert_close(out, ref + 1, tol, name='TT-ish (A k-strided, B k-contig) beta=1')


@pytest.mark.parametrize('dt,tag,tol', DT)
def test_gemm_batched_heads(tgt, dt, tag, tol):
    torch.manual_seed(1)
    Bn, Nq, Nk, h, hd = 2, 40, 24, 2, 64  # hd = 64: the LDS-DMA NT kernel's K
    C = h * hd
    q, kv = torch.randn(Bn, Nq, C).to(dt), torch.randn(Bn, Nk, 2 * C).to(dt)
    qd, kvd = tgt.to(q), tgt.to(kv)
    S = torch.empty(Bn, h, Nq, Nk, dtype=dt, device=tgt.device)
    for hh in range(h):
        ops.gemm(ops.plain_view(qd, Nq, hd, ld=C, batch_stride=Nq * C, offset=hh * hd),
                 ops.plain_view(kvd, Nk, hd, ld=2 * C, batch_stride=Nk * 2 * C, offset=hh * hd),
                 S, Nq, Nk, hd, batch=Bn, c_batch_stride=h * Nq * Nk, c_offset=hh * Nq * Nk, dtype=tag, alpha=0.25)
    qr = q.float().view(Bn, Nq, h, hd).permute(0, 2, 1, 3)
    kr = kv.float()[..., :C].reshape(Bn, Nk, h, hd).permute(0, 2, 1, 3)
    assert_close(S, 0.25 * qr @ kr.transpose(-1, -2), tol, name='batched QK^T')
    # same product in ONE launch over (batch, head) = (batch, batch2)
    S2 = torch.empty(Bn, h, Nq, Nk, dtype=dt, device=tgt.device)
    ops.gemm(ops.plain_view(qd, Nq, hd, ld=C, batch_stride=Nq * C, batch2_stride=hd),
             ops.plain_view(kvd, Nk, hd, ld=2 * C, batch_stride=Nk * 2 * C, batch2_stride=hd), S2, Nq, Nk, hd, batch=Bn,
             batch2=h, c_batch_stride=h * Nq * Nk, c_batch2_stride=Nq * Nk, dtype=tag, alpha=0.25)
    assert_close(S2, 0.25 * qr @ kr.transpose(-1, -2), tol, name='batched QK^T (batch2)')


CONVS = [(2, 9, 11, 8, 24, 3, 1, 1, 1), (1, 16, 16, 3, 16, 7, 4, 3, 1), (1, 12, 12, 16, 8, 3, 2, 1, 1),
         (1, 14, 14, 8, 8, 3, 1, 2, 2), (1, 8, 8, 16, 32, 2, 2, 0, 1), (1, 11, 13, 8, 8, 3, 2, 1, 1),
         (2, 6, 70, 64, 16, 3, 1, 1, 1), (1, 10, 66, 72, 8, 3, 1, 6, 6),  # channels / row lengths beyond one 64-wide k-tile
         # kernel == stride (spatial-reduction convs): patch views, filled like plain operands when KW*Ci % 64 == 0
         (2, 8, 12, 32, 24, 2, 2, 0, 1), (1, 16, 8, 64, 16, 4, 4, 0, 1), (3, 8, 8, 96, 40, 2, 2, 0, 1), (2, 32, 16, 8, 16, 8, 8, 0, 1),
         # OW = 16 divides the 64-pixel k-tile and the pixel count is a multiple of it: the K-strided patch view's constant-step DMA source
         (1, 32, 32, 32, 8, 2, 2, 0, 1), (2, 32, 64, 16, 24, 4, 4, 0, 1),
         # stride-1 'same' convolutions whose output rows are whole k-tiles: the row-fast DMA source of the K-strided im2col operand
         # (weight gradient; DmaSrc mode 3), plain and dilated, two images (the pixel index runs across image boundaries)
         (2, 4, 64, 8, 16, 3, 1, 1, 1), (1, 8, 64, 8, 8, 3, 1, 2, 2), (2, 3, 128, 16, 8, 3, 1, 1, 1)]


@pytest.mark.parametrize('dt,tag,tol', DT)
@pytest.mark.parametrize('Bc,H,W,Ci,Co,KH,st,pd,dl', CONVS)
def test_conv_implicit_gemm(tgt, dt, tag, tol, Bc, H, W, Ci, Co, KH, st, pd, dl):
    torch.manual_seed(H + Ci)
    x, w = torch.randn(Bc, H, W, Ci).to(dt), torch.randn(Co, Ci, KH, KH).to(dt)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yref = F.conv2d(xr, wr, stride=st, padding=pd, dilation=dl)
    OH, OW = yref.shape[2:]
    dy = torch.randn(Bc, OH, OW, Co).to(dt)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    xd, dyd = tgt.to(x), tgt.to(dy)
    wg = tgt.to(w.permute(0, 2, 3, 1).contiguous().view(Co, -1))
    K = KH * KH * Ci
    out = torch.empty(Bc, OH, OW, Co, dtype=dt, device=tgt.device)
    ops.gemm(ops.conv_view(xd, Bc, H, W, Ci, KH, KH, st, pd, dl), ops.plain_view(wg, Co, K), out, Bc * OH * OW, Co, K, dtype=tag)
    assert_close(out, yref.permute(0, 2, 3, 1), tol * 2, name='conv fwd')
    dW = torch.zeros(Co, K, device=tgt.device)
    ops.gemm(ops.plain_view(dyd, Bc * OH * OW, Co), ops.conv_view(xd, Bc, H, W, Ci, KH, KH, st, pd, dl), dW, Co, K,
             Bc * OH * OW, a_kstrided=True, b_kstrided=True, dtype=tag, atomic=True, splits=2)
    assert_close(dW.view(Co, KH, KH, Ci).permute(0, 3, 1, 2), wr.grad, 1e-5 if tag != 2 else tol, name='conv wgrad')
    if st == 1:
        wd = tgt.to(w.flip(2, 3).permute(1, 2, 3, 0).contiguous().view(Ci, -1))
        dx = torch.empty(Bc, H, W, Ci, dtype=dt, device=tgt.device)
        ops.gemm(ops.conv_view(dyd, Bc, OH, OW, Co, KH, KH, 1, dl * (KH - 1) - pd, dl, OH=H, OW=W),
                 ops.plain_view(wd, Ci, KH * KH * Co), dx, Bc * H * W, Ci, KH * KH * Co, dtype=tag)
        assert_close(dx, xr.grad.permute(0, 2, 3, 1), tol * 2, name='conv dgrad')


def _run_forced_tile(marker):
    """the tile heuristics only pick the 256x256 (8-wave) kernel for very large problems: force it through
    cmda_gemm_params_t.tile_hint (tests/conftest.py sets ops.GEMM_TILE_HINT from CMDA_TEST_GEMM_TILE) and run the whole GEMM
    suite through it in a child process"""
    import os
    import subprocess
    import sys
    here = os.path.abspath(__file__)
    # 4 | 1024: the ping-pong kernel (gemm_pp.hip) wherever its epilogue / operand modes allow; 4 | 512: gemm_glds_kernel's 8-wave tile
    for hint in ('1028', '516'):
        env = dict(os.environ, CMDA_TEST_GEMM_TILE=hint)
        # (the tests of launches that ignore the forced tile -- pair / row-panel / deferred-grouped -- are left out: they run once, above)
        r = subprocess.run([sys.executable, '-m', 'pytest', here, '-q', '-x', '-m', marker, '-k',
                            'not forced_tile and not pair_launch and not row_panel and not deferred_grouped and not ln_gemm', '-p', 'no:cacheprovider'],
                           env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, f'tile_hint {hint}: ' + r.stdout[-2000:] + r.stderr[-2000:]


def test_forced_tile_256_emu():
    _run_forced_tile('not gpu')


@pytest.mark.gpu
def test_forced_tile_256_gpu():
    _run_forced_tile('gpu')


@pytest.mark.parametrize('dt,tag,tol,B,OH,OW,Co,Ci,k', [(torch.float32, 0, 2e-5, 2, 3, 5, 24, 8, 2),
                                                        # bf16 with Co % 64 == 0: the lean kernel's patch-store epilogue (K-strided weights)
                                                        (torch.bfloat16, 1, 2e-2, 2, 3, 5, 64, 32, 2), (torch.bfloat16, 1, 2e-2, 3, 4, 8, 128, 64, 4),
                                                        (torch.bfloat16, 1, 2e-2, 1, 9, 7, 192, 8, 2)])
def test_gemm_unpatchify_store(tgt, dt, tag, tol, B, OH, OW, Co, Ci, k):
    """c_patch: the data gradient of a kernel == stride convolution is stored straight in NHWC (mix_transformer.py:70-75's sr
    conv backward), compared with conv_transpose2d; also with beta accumulation."""
    import torch.nn.functional as F
    from cmda_amd import ops
    torch.manual_seed(5)
    dy = torch.randn(B, Co, OH, OW).to(dt)
    w = (torch.randn(Co, Ci, k, k) * (1.0 if tag == 0 else 0.2)).to(dt)
    want = F.conv_transpose2d(dy.float(), w.float(), stride=k)       # [B, Ci, OH*k, OW*k]
    prev = torch.randn(B * OH * k * OW * k, Ci).to(dt)
    dy_rows = tgt.to(dy.permute(0, 2, 3, 1).reshape(-1, Co).contiguous())
    w_khwc = tgt.to(w.permute(0, 2, 3, 1).reshape(Co, k * k * Ci).contiguous())
    for beta in (0.0, 1.0):
        dx = tgt.to(prev.clone())
        M, K = B * OH * OW, k * k * Ci
        ops.gemm(ops.plain_view(dy_rows, M, Co), ops.plain_view(w_khwc, Co, K), dx, M, K, Co, b_kstrided=True, dtype=tag,
                 beta=beta, c_patch=(OW, k, k * Ci))
        ref = want.permute(0, 2, 3, 1).reshape(-1, Ci) + beta * prev.float()
        assert_close(dx, ref, tol, name=f'unpatchify beta={beta}')


@pytest.mark.parametrize('hint,M,N', [(3, 600, 1100), (3, 520, 1024), (2, 1100, 1060), (4, 1030, 2100)])
def test_gemm_grouped_tile_walk(tgt, hint, M, N):
    """wide outputs walk their tiles in groups of GM tile rows, column by column (L2 reuse inside an XCD): every tile must still
    be computed exactly once, ragged last group and ragged edges included.  hint = forced tile (3: 64x64, 2: 128x64, 4: 256x256)."""
    K = 64
    torch.manual_seed(hint * 100 + M)
    a, b = torch.randn(M, K).to(torch.bfloat16), torch.randn(N, K).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    out = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=tgt.device)
    ad, bd = tgt.to(a), tgt.to(b)   # (views hold raw pointers: the device copies must outlive the launch)
    ops.GEMM_TILE_HINT = hint
    try:
        ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=1)
        assert_close(out, ref, 1.5e-2, name='grouped walk')
        ops.GEMM_TILE_HINT = hint | 256   # the row-major walk (tuning switch) gives the same matrix
        out2 = torch.empty_like(out)
        ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out2, M, N, K, dtype=1)
        assert torch.equal(out.cpu(), out2.cpu())
    finally:
        ops.GEMM_TILE_HINT = 0


@pytest.mark.parametrize('dt,tag', [(torch.bfloat16, 1), (torch.float32, 0)])
def test_gemm_deferred_grouped_weight_gradients(tgt, dt, tag):
    """ops.gemm(defer=True) inside a deferral scope queues the weight gradients of a backward pass and gemm_flush_deferred()
    launches them as grouped grids (cmda_gemm_grouped: 64x64 ... 128x128 tiles and gemm_wg.hip's 256x256 tile, plain / patch / im2col B views, fused bias
    gradient, accumulation into non-zero gradients, deep contractions split by the planner); the fp32 parity mode and forced
    tiles fall back to single launches inside the same call.  Result == the same GEMMs launched one by one."""
    import torch.nn.functional as Fn
    from cmda_amd import nn as K
    import cmda_amd.runtime as rt
    torch.manual_seed(5)
    rt.set_compute_dtype(dt)
    try:
        shapes = [(520, 64, 64), (4200, 320, 128), (3300, 128, 256), (700, 72, 40), (130, 256, 128), (64, 8, 24), (2000, 256, 64), (9000, 128, 320),
                  (3000, 320, 320), (2100, 640, 320), (1500, 1280, 320), (1100, 320, 1280)]   # (rows M, out N, in K); the last four: 256x256 tiles
        lins = []
        for rows, n, k in shapes:
            lins.append((torch.randn(rows, n).to(dt), torch.randn(rows, k).to(dt), torch.nn.Parameter(tgt.to(torch.randn(n, k))),
                         torch.nn.Parameter(tgt.to(torch.randn(n)))))
        # a spatial-reduction conv (patch view) and a 3x3 conv (im2col view)
        convs = [(2, 16, 16, 32, 24, 2, 2, 0), (1, 12, 20, 16, 16, 3, 1, 1), (1, 12, 20, 32, 272, 3, 1, 1), (2, 16, 16, 320, 320, 2, 2, 0)]
        cvs = []
        for Bc, H, W, Ci, Co, k, st, pd in convs:
            OH, OW = K.conv_out_size(H, W, k, st, pd)
            cvs.append((torch.randn(Bc * OH * OW, Co).to(dt), torch.randn(Bc, H, W, Ci).to(dt), torch.nn.Parameter(tgt.to(torch.randn(Co, Ci, k, k))),
                        torch.nn.Parameter(tgt.to(torch.randn(Co))), (Bc, H, W, st, pd)))

        def run(defer):
            ops.GEMM_DEFER = defer
            grads = []
            for _, _, w, b in lins:
                w.grad, b.grad = torch.full_like(w.data, 0.5), torch.full_like(b.data, -1.0)
            for _, _, w, b, _ in cvs:
                w.grad, b.grad = torch.zeros_like(w.data), torch.zeros_like(b.data)
            with ops.ln_deferral():
                for dy, x, w, b in lins:
                    K.linear_bwd(tgt.to(dy), tgt.to(x), w, b, dy.shape[0], x.shape[1], need_dx=False)
                for dy, x, w, b, (Bc, H, W, st, pd) in cvs:
                    K.conv_bwd(tgt.to(dy), tgt.to(x.reshape(-1, x.shape[-1])), w, b, Bc, H, W, st, pd, need_dx=False)
                assert bool(ops._GD['queues']) == bool(defer)
            assert not ops._GD['queues']
            for _, _, w, b in lins:
                grads += [w.grad.cpu().clone(), b.grad.cpu().clone()]
            for _, _, w, b, _ in cvs:
                grads += [w.grad.cpu().clone(), b.grad.cpu().clone()]
            return grads
        try:
            single, grouped = run(False), run(True)
        finally:
            ops.GEMM_DEFER = True
        tol = 2e-2 if dt == torch.bfloat16 else 1e-4
        for i, (a, b) in enumerate(zip(single, grouped)):
            assert_close(b, a, 1e-5, name=f'grouped vs single {i}')
        # and against torch
        for j, (dy, x, w, b) in enumerate(lins):
            assert_close(grouped[2 * j], dy.float().t() @ x.float() + 0.5, tol, name=f'dW {j}')
            assert_close(grouped[2 * j + 1], dy.float().sum(0) - 1.0, tol, name=f'db {j}')
        for j, (dy, x, w, b, (Bc, H, W, st, pd)) in enumerate(cvs):
            xr = x.float().permute(0, 3, 1, 2)
            wr = w.data.cpu().clone().requires_grad_(True)
            y = Fn.conv2d(xr, wr, None, st, pd)
            g = dy.float().view(Bc, y.shape[2], y.shape[3], -1).permute(0, 3, 1, 2)
            y.backward(g)
            assert_close(grouped[2 * len(lins) + 2 * j], wr.grad, tol, name=f'conv dW {j}')
    finally:
        rt.set_compute_dtype(torch.float32)


@pytest.mark.parametrize('M,N,K', [(70, 320, 64), (200, 640, 320), (130, 300, 200), (64, 1280, 128), (257, 320, 1280)])
def test_gemm_row_panel_tile(tgt, M, N, K):
    """the 64 x 320 row-panel tile (gemm_t4.hip, tile_hint 5): whole rows of the C = 320 stage's Linear layers per workgroup, with
    every epilogue the encoder blocks use on it (bias, GELU, per-sample drop-path scale, fp32 residual into an fp32 output)"""
    old = ops.GEMM_TILE_HINT
    ops.GEMM_TILE_HINT = 5
    try:
        torch.manual_seed(M + N)
        a, b = torch.randn(M, K).bfloat16(), (torch.randn(N, K) * 0.2).bfloat16()
        bias, res = torch.randn(N), torch.randn(M, N)
        ref = a.float() @ b.float().t()
        ad, bd, biasd = tgt.to(a), tgt.to(b), tgt.to(bias)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=tgt.device)
        ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=1, bias=biasd)
        assert_close(out, ref + bias, 1.5e-2, name='row panel NT + bias')
        out = torch.empty(M, N, dtype=torch.bfloat16, device=tgt.device)
        ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=1, bias=biasd, act='gelu')
        assert_close(out, F.gelu(ref + bias), 1.5e-2, name='row panel NT + bias + gelu')
        sc = torch.tensor([0.0, 1.25, 1.25, 0.0])
        rps = (M + 3) // 4
        out = torch.empty(M, N, dtype=torch.float32, device=tgt.device)
        ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=1, bias=biasd, res=tgt.to(res),
                 rowscale=tgt.to(sc), rows_per_scale=rps)
        assert_close(out, res + (ref + bias) * sc[torch.arange(M) // rps, None], 2e-3, atol=1e-3, name='row panel fp32 residual epilogue')
    finally:
        ops.GEMM_TILE_HINT = old


@pytest.mark.parametrize('nn', [False, True], ids=['NT', 'NN'])
def test_gemm_pair_launch(tgt, nn):
    """cmda_gemm_pair: two independent problems in one grid -- a plain Linear next to a patch-view convolution (q and the
    spatial-reduction conv of a MiT block: same input, different shapes and epilogues), and two data gradients (K-strided weights,
    one accumulating into its output).  Must equal the two single launches bit for bit."""
    torch.manual_seed(17)
    Bc, H, W, C, s = 1, 8, 16, 32, 2          # tokens 128, channels 32; sr conv: kernel = stride = 2 -> 32 rows x K = 128
    M = Bc * H * W
    x = tgt.to(torch.randn(M, C).bfloat16())
    if not nn:
        wq, bq = tgt.to((torch.randn(96, C) * 0.2).bfloat16()), tgt.to(torch.randn(96))
        ws, bs = tgt.to((torch.randn(C, s * s * C) * 0.1).bfloat16()), tgt.to(torch.randn(C))
        OH, OW = H // s, W // s

        def build(hold):
            q = torch.empty(M, 96, dtype=torch.bfloat16, device=tgt.device)
            xs = torch.empty(Bc * OH * OW, C, dtype=torch.bfloat16, device=tgt.device)
            h0 = ops.gemm(ops.plain_view(x, M, C), ops.plain_view(wq, 96, C), q, M, 96, C, dtype=1, bias=bq, act='gelu', hold=hold)
            h1 = ops.gemm(ops.conv_view(x, Bc, H, W, C, s, s, s, 0), ops.plain_view(ws, C, s * s * C), xs, Bc * OH * OW, C, s * s * C,
                          dtype=1, bias=bs, hold=hold)
            return q, xs, h0, h1
    else:
        dy0, dy1 = tgt.to(torch.randn(M, 64).bfloat16()), tgt.to(torch.randn(40, 128).bfloat16())
        w0, w1 = tgt.to((torch.randn(64, C) * 0.2).bfloat16()), tgt.to((torch.randn(128, C) * 0.2).bfloat16())   # [N_out, K_in]
        prev = tgt.to(torch.randn(40, C).bfloat16())

        def build(hold):
            d0 = torch.empty(M, C, dtype=torch.bfloat16, device=tgt.device)
            d1 = prev.clone()
            h0 = ops.gemm(ops.plain_view(dy0, M, 64), ops.plain_view(w0, 64, C), d0, M, C, 64, b_kstrided=True, dtype=1, hold=hold)
            h1 = ops.gemm(ops.plain_view(dy1, 40, 128), ops.plain_view(w1, 128, C), d1, 40, C, 128, b_kstrided=True, dtype=1, beta=1.0, hold=hold)
            return d0, d1, h0, h1
    a0, a1, _, _ = build(False)
    b0, b1, h0, h1 = build(True)
    old = ops.GEMM_PAIR
    ops.GEMM_PAIR = True   # (off by default in the step since round 4's eight-wave lean kernel: ops.GEMM_PAIR)
    try:
        ops.gemm_pair(h0, h1)
    finally:
        ops.GEMM_PAIR = old
    assert torch.equal(a0.float().cpu(), b0.float().cpu()) and torch.equal(a1.float().cpu(), b1.float().cpu())
    if not nn:
        ref = torch.nn.functional.gelu(x.float().cpu() @ wq.float().cpu().t() + bq.cpu())
        assert_close(b0, ref, 1.5e-2, name='pair: q')


@pytest.mark.parametrize('M,N,K,xdt', [(130, 320, 320, torch.float32), (64, 64, 64, torch.float32), (200, 96, 128, torch.float32),
                                       (257, 640, 320, torch.bfloat16), (70, 200, 256, torch.float32), (100, 64, 384, torch.float32),
                                       (90, 128, 512, torch.float32)])
def test_ln_gemm_matches_two_launches(tgt, M, N, K, xdt):
    """cmda_ln_gemm (LayerNorm in the prologue of the Linear behind it: norm1 -> q, attn.norm -> kv) against the two separate launches:
    statistics to fp32 round-off, normalised rows within one bf16 rounding step, the output within the bf16 GEMM tolerance; ragged
    M / N included; K = 512 exceeds the resident panel and takes the library's two-launch path (bit-identical then)."""
    torch.manual_seed(M + K)
    x = tgt.to((torch.randn(M, K) * 2 + 0.5).to(xdt))
    gamma, beta = tgt.to(torch.randn(K) * 0.5 + 1), tgt.to(torch.randn(K) * 0.1)
    w, b = tgt.to((torch.randn(N, K) * 0.1).bfloat16()), tgt.to(torch.randn(N))
    xn_ref, m_ref, r_ref = ops.layernorm_fwd(x, gamma, beta, 1e-6, out_dtype=torch.bfloat16)
    y_ref = torch.empty(M, N, dtype=torch.bfloat16, device=tgt.device)
    ops.gemm(ops.plain_view(xn_ref, M, K), ops.plain_view(w, N, K), y_ref, M, N, K, dtype=1, bias=b)
    exact = K > 384
    old = ops.LN_GEMM, ops.GEMM_TILE_HINT
    # (off by default in the step: ops.LN_GEMM; tile_hint bit 15: the fused kernel whatever the grid size -- its shape rule, gemm_ln.hip)
    ops.LN_GEMM, ops.GEMM_TILE_HINT = True, 32768
    try:
        for store in (True, False):
            xn = torch.zeros(M, K, dtype=torch.bfloat16, device=tgt.device)
            y = torch.empty(M, N, dtype=torch.bfloat16, device=tgt.device)
            h = ops.gemm(ops.plain_view(xn, M, K), ops.plain_view(w, N, K), y, M, N, K, dtype=1, bias=b, hold=True, keep=(xn,))
            m, r = ops.ln_gemm(x, gamma, beta, 1e-6, h, store_xn=store)
            if exact:
                assert torch.equal(y.float().cpu(), y_ref.float().cpu()) and torch.equal(xn.float().cpu(), xn_ref.float().cpu())
                assert torch.equal(m.cpu(), m_ref.cpu()) and torch.equal(r.cpu(), r_ref.cpu())
                continue
            assert_close(m, m_ref, 1e-5, name='ln_gemm mean')
            assert_close(r, r_ref, 1e-5, name='ln_gemm rstd')
            assert_close(y, y_ref, 4e-3, name='ln_gemm output vs two launches')
            if store:
                assert_close(xn, xn_ref, 8e-3, name='ln_gemm normalised rows')   # one bf16 step (2^-8) of the largest entry at most
                assert float((xn.float() != xn_ref.float()).float().mean()) < 0.01
            else:
                assert float(xn.float().abs().max()) == 0.0
    finally:
        ops.LN_GEMM, ops.GEMM_TILE_HINT = old
    # against torch (fp32 LayerNorm, bf16-rounded operands)
    ref = torch.nn.functional.layer_norm(x.float().cpu(), (K,), gamma.cpu(), beta.cpu(), 1e-6).bfloat16().float() @ w.float().cpu().t() + b.cpu()
    assert_close(y, ref, 1.5e-2, name='ln_gemm vs torch')
