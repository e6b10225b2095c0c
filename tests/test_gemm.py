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


@pytest.mark.parametrize('Bc,H,W,Ci,Co,k,splits', [(2, 16, 16, 32, 40, 2, 3), (1, 32, 16, 16, 64, 4, 4), (3, 8, 8, 64, 24, 2, 5)])
def test_x3_patch_conv_split_k_forward(tgt, Bc, H, W, Ci, Co, k, splits):
    """split-bf16 mode, kernel == stride convolution of a small token count (the spatial-reduction convolutions of the deep stages,
    mix_transformer.py:70-75): bias rows pre-filled, the K range split over workgroups, fp32 atomics -- the lean split kernel's forward
    form with a split count (patch view as A: a split starts inside a kernel row), and the same with a plain A operand"""
    torch.manual_seed(Ci + k)
    x, w, bias = torch.randn(Bc, H, W, Ci), torch.randn(Co, Ci, k, k), torch.randn(Co)
    yref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=k).permute(0, 2, 3, 1)
    OH, OW = H // k, W // k
    M, K = Bc * OH * OW, k * k * Ci
    xd, wg = tgt.to(x), tgt.to(w.permute(0, 2, 3, 1).contiguous().view(Co, -1))
    prev = ops.GEMM_TILE_HINT
    try:
        for hint in (0, 65536, 8192):   # the library's choice, the lean split kernel whatever the tile choice, the register-staged kernel
            ops.GEMM_TILE_HINT = hint
            out = tgt.to(bias.expand(M, Co).contiguous())
            ops.gemm(ops.conv_view(xd, Bc, H, W, Ci, k, k, k, 0, 1), ops.plain_view(wg, Co, K), out, M, Co, K, dtype=2, atomic=True, splits=splits)
            assert_close(out.view(Bc, OH, OW, Co), yref, 1e-4, name=f'x3 patch conv, split-K atomics (hint {hint})')
    finally:
        ops.GEMM_TILE_HINT = prev
    cols = tgt.to(F.unfold(x.permute(0, 3, 1, 2), k, stride=k).view(Bc, Ci, k * k, OH * OW).permute(0, 3, 2, 1).reshape(M, K).contiguous())
    out2 = tgt.to(bias.expand(M, Co).contiguous())
    ops.gemm(ops.plain_view(cols, M, K), ops.plain_view(wg, Co, K), out2, M, Co, K, dtype=2, atomic=True, splits=splits)
    assert_close(out2.view(Bc, OH, OW, Co), yref, 1e-4, name='x3 plain A, split-K atomics')


@pytest.mark.parametrize('dt,tag,tol', DT)
@pytest.mark.parametrize('hint', [0, 1, 2, 3, 4, 516, 1028])
def test_gemm_fused_column_statistics(tgt, dt, tag, tol, hint):
    """cmda_gemm_params_t.colstats: the batch statistics of the BatchNorm / InstanceNorm behind a convolution (mmcv ConvModule conv -> norm,
    daformer_head.py:46-62; cyclegan_model.py:339-434) accumulated by the GEMM epilogue into the BatchNorm workspace -- sums of the
    stored values per row group, on every tile of the general kernels; then cmda_bn_train_fwd(ws_has_stats) against the BatchNorm that
    takes its own statistics pass, and the workspace handed back zeroed"""
    from cmda_amd import _lib as L
    torch.manual_seed(3 + hint)
    groups, rpg, N, K = 3, 512, 72, 96
    M = groups * rpg
    a, b, bias = torch.randn(M, K).to(dt), torch.randn(N, K).to(dt), torch.randn(N) * 2 + 0.5
    res = torch.randn(M, N).to(dt) if hint != 1028 else None      # (1028: the ping-pong kernel, which takes no residual)
    ref = a.float() @ b.float().t() + bias + (res.float() if res is not None else 0.0)
    ad, bd, bsd, rd = tgt.to(a), tgt.to(b), tgt.to(bias), (tgt.to(res) if res is not None else None)
    wsn = int(L.lib().cmda_bn_ws_floats(N))
    prev = ops.GEMM_TILE_HINT
    try:
        ops.GEMM_TILE_HINT = hint
        for out_dt in (dt, torch.float32):
            ws = torch.zeros(groups * wsn, device=tgt.device)
            out = torch.empty(M, N, dtype=out_dt, device=tgt.device)
            ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=tag, bias=bsd, res=rd, colstats=(ws, rpg))
            assert_close(out, ref, tol, name='output')
            st = ws.view(groups, 33, 2, N)[:, :32].sum(1).cpu()          # [group][sum | sum of squares][channel]
            # (the ping-pong kernel sums the bf16 values it stores, the general epilogue the unrounded ones)
            rg = (out.float().cpu() if (hint == 1028 and tag == 1 and out_dt == dt) else ref).view(groups, rpg, N)
            assert_close(st[:, 0], rg.sum(1), tol, atol=tol * rpg ** 0.5 * 4, name='column sums')
            assert_close(st[:, 1], (rg * rg).sum(1), tol * 2, name='column sums of squares')
            assert ws.view(groups, 33, 2, N)[:, 32].abs().max().item() == 0.0
            # BatchNorm from the workspace == BatchNorm with its own statistics pass over the stored activation
            g, be = tgt.to(torch.rand(N) + 0.5), tgt.to(torch.randn(N))
            outs = []
            for fused in (True, False):
                rm, rv = torch.zeros(N, device=tgt.device), torch.ones(N, device=tgt.device)
                y = torch.empty_like(out)
                mean, rstd = ops.bn_train_fwd(out, g, be, y, rm, rv, rpg, N, 1e-5, 0.1, True, groups=groups, order=[2, 0, 1],
                                              stats_ws=ws if fused else None)
                outs.append((y, mean, rstd, rm, rv))
            # (bf16 storage: the fused statistics are those of the unrounded values, the separate pass sees the rounded ones)
            st_tol = 2e-5 if out_dt == torch.float32 else 4e-3
            for i, nm in enumerate(('y', 'mean', 'rstd', 'running_mean', 'running_var')):
                assert_close(outs[0][i], outs[1][i], st_tol if i else max(st_tol, tol), atol=st_tol, name=f'BatchNorm from epilogue statistics: {nm}')
            assert ws.abs().max().item() == 0.0, 'the workspace comes back zeroed'
            # a group order that is not a permutation would leave a group's sums in the workspace: refused when the workspace carries them
            from cmda_amd._lib import CmdaError
            with pytest.raises(CmdaError):
                ops.bn_train_fwd(out, g, be, torch.empty_like(out), torch.zeros(N, device=tgt.device), torch.ones(N, device=tgt.device), rpg, N,
                                 1e-5, 0.1, True, groups=groups, order=[2, 0, 0], stats_ws=ws)
    finally:
        ops.GEMM_TILE_HINT = prev


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
        # (the tests of launches that ignore the forced tile -- deferred-grouped -- are left out: they run once, above)
        r = subprocess.run([sys.executable, '-m', 'pytest', here, '-q', '-x', '-m', marker, '-k',
                            'not forced_tile and not deferred_grouped', '-p', 'no:cacheprovider'],
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
                                                        (torch.bfloat16, 1, 2e-2, 1, 9, 7, 192, 8, 2),
                                                        # split-bf16 with Co % 32 == 0: the lean split kernel's patch-store epilogue
                                                        (torch.float32, 2, 1e-4, 2, 3, 5, 64, 32, 2), (torch.float32, 2, 1e-4, 3, 4, 8, 128, 64, 4),
                                                        (torch.float32, 2, 1e-4, 1, 9, 7, 96, 8, 2)])
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


@pytest.mark.parametrize('dt,tag', [(torch.bfloat16, 1), (torch.float32, 0), (torch.float32, 2)])
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
    rt.set_gemm_x3(tag == 2)   # split-bf16 mode: the lean split kernel's grouped weight-gradient form (token counts that are whole 32-deep k-tiles)
    try:
        shapes = [(520, 64, 64), (4200, 320, 128), (3300, 128, 256), (700, 72, 40), (130, 256, 128), (64, 8, 24), (2000, 256, 64), (9000, 128, 320),
                  (3000, 320, 320), (2100, 640, 320), (1500, 1280, 320), (1100, 320, 1280),   # (rows M, out N, in K); these four: 256x256 tiles
                 (4096, 320, 320), (2048, 1280, 320), (1024, 64, 512), (8192, 64, 64), (96, 128, 128)]   # whole 32-deep k-tiles
        lins = []
        for rows, n, k in shapes:
            lins.append((torch.randn(rows, n).to(dt), torch.randn(rows, k).to(dt), torch.nn.Parameter(tgt.to(torch.randn(n, k))),
                         torch.nn.Parameter(tgt.to(torch.randn(n)))))
        # a spatial-reduction conv (patch view) and a 3x3 conv (im2col view)
        # (the last one: a patch view whose output rows are exactly one 64-deep k-tile (OW = 64) on the 256 x 256 tile's 32-deep kernel --
        # classed 'fast' by a 64-deep eligibility test, its weight gradient came out wrong)
        convs = [(2, 16, 16, 32, 24, 2, 2, 0), (1, 12, 20, 16, 16, 3, 1, 1), (1, 12, 20, 32, 272, 3, 1, 1), (2, 16, 16, 320, 320, 2, 2, 0),
                 (1, 4, 128, 64, 256, 2, 2, 0)]
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
        if tag == 2:   # the planner does put an eligible problem into the grouped launch of the split kernel (a block map exists)
            import ctypes
            from cmda_amd import _lib as L
            dy, x = tgt.to(lins[-5][0]), tgt.to(lins[-5][1])
            pr = ops.gemm(ops.plain_view(dy, 4096, 320), ops.plain_view(x, 4096, 320), torch.zeros(320, 320, device=tgt.device), 320, 320, 4096,
                          a_kstrided=True, b_kstrided=True, dtype=2, atomic=True, splits=0, hold=True)[0]
            arr = (L.GemmParams * 1)(pr)
            assert int(L.lib().cmda_gemm_grouped_ws_bytes(arr, ctypes.c_int32(1))) > (ctypes.sizeof(L.GemmParams) + 15) // 16 * 16
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
        rt.set_gemm_x3(False)
        rt.set_compute_dtype(torch.float32)


def test_grouped_gemm_eager_arena_recycles_after_a_capture(tgt):
    """ops._gd_arena: once a capture has pinned plans, eager grouped launches build their tables in a second, recyclable arena that is
    rewound when it fills -- no new pinned arena per exhaustion.  Simulated here by marking plans pinned and shrinking
    the eager arena: every flush still produces the right weight gradients and the rewinds are counted."""
    from cmda_amd import nn as K
    import cmda_amd.runtime as rt
    rt.set_compute_dtype(torch.bfloat16)
    saved = (ops._GD['pinned_plans'], ops._GD.get('eager_arena'), ops._GD_EAGER_ARENA_BYTES, ops._GD.get('recycles', 0), dict(ops._GD['plans']))
    try:
        ops._GD['pinned_plans'], ops._GD['eager_arena'], ops._GD_EAGER_ARENA_BYTES = True, None, 1024
        torch.manual_seed(9)
        before = ops._GD.get('recycles', 0)
        for rep in range(6):
            rows, n, k = 256 + 64 * rep, 64, 64    # a new problem (and plan key) every time
            dy, x = torch.randn(rows, n).bfloat16(), torch.randn(rows, k).bfloat16()
            w, b = torch.nn.Parameter(tgt.to(torch.randn(n, k))), torch.nn.Parameter(tgt.to(torch.randn(n)))
            w.grad, b.grad = torch.zeros_like(w.data), torch.zeros_like(b.data)
            with ops.ln_deferral():
                K.linear_bwd(tgt.to(dy), tgt.to(x), w, b, rows, k, need_dx=False)
            assert_close(w.grad, dy.float().t() @ x.float(), 2e-2, name=f'dW after {rep} flushes')
        assert ops._GD.get('recycles', 0) > before, 'the eager arena never filled: shrink it further'
        assert ops._GD["eager_arena"][0].numel() <= 2048
    finally:
        ops._GD['pinned_plans'], ops._GD['eager_arena'], ops._GD_EAGER_ARENA_BYTES = saved[0], saved[1], saved[2]
        ops._GD['recycles'] = saved[3]
        ops._GD['plans'] = saved[4]
        rt.set_compute_dtype(torch.float32)


def test_gemm_x3_big_three_launch_path(tgt):
    """split-bf16 mode, LARGE problems (ops._gemm_x3_big): operands split once into bf16 hi / lo tensors (cmda_split_bf16) and the
    contraction run as three launches of the bf16 kernels accumulated in the fp32 output -- plain NT with bias + fp32 residual + beta,
    the weight-gradient form with atomics and the fused bias gradient, an im2col view.  Forced here by lowering the FLOP threshold."""
    old, old_i = ops.X3_BIG_FLOPS, ops.X3_BIG_INTENSITY
    ops.X3_BIG_FLOPS, ops.X3_BIG_INTENSITY = 1e3, 0.0
    try:
        torch.manual_seed(21)
        M, N, K = 200, 72, 128
        a, b = torch.randn(M, K), torch.randn(N, K)
        bias, res, c0 = torch.randn(N), torch.randn(M, N), torch.randn(M, N)
        ad, bd = tgt.to(a), tgt.to(b)
        out = tgt.to(c0.clone())
        ops.gemm(ops.plain_view(ad, M, K), ops.plain_view(bd, N, K), out, M, N, K, dtype=2, bias=tgt.to(bias), res=tgt.to(res), beta=0.5)
        assert_close(out, a @ b.t() + bias + res + 0.5 * c0, 2e-5, name='x3 big NT + bias + res + beta')
        hi, lo = ops.split_bf16(ad)
        assert_close(hi.float() + lo.float(), a, 2e-5, name='hi + lo')
        assert torch.equal(hi.cpu(), a.bfloat16())
        atd, btd = tgt.to(a.t().contiguous()), tgt.to(b.t().contiguous())
        dW, cs = torch.zeros(M, N, device=tgt.device), torch.zeros(M, device=tgt.device)
        ops.gemm(ops.plain_view(atd, K, M), ops.plain_view(btd, K, N), dW, M, N, K, a_kstrided=True, b_kstrided=True, dtype=2, atomic=True,
                 splits=0, colsum=cs)
        assert_close(dW, a @ b.t(), 2e-5, name='x3 big weight-gradient form')
        assert_close(cs, a.sum(1), 2e-5, name='x3 big fused bias gradient')
        x, w = torch.randn(2, 12, 12, 16), torch.randn(24, 16, 3, 3)
        wg = tgt.to(w.permute(0, 2, 3, 1).contiguous().view(24, -1))
        o = torch.empty(2, 12, 12, 24, device=tgt.device)
        ops.gemm(ops.conv_view(tgt.to(x), 2, 12, 12, 16, 3, 3, 1, 1, 1), ops.plain_view(wg, 24, 144), o, 288, 24, 144, dtype=2)
        assert_close(o, F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1), 2e-5, name='x3 big im2col view')
    finally:
        ops.X3_BIG_FLOPS, ops.X3_BIG_INTENSITY = old, old_i

