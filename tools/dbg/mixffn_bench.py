"""graph-timed fused MixFFN (cmda_mixffn_fwd) against LayerNorm + fc1 + depthwise/GELU + fc2 as four launches at the stage-3 shapes of
the step (us per block-half, one lane; and two lanes side by side: the image encoder's B and the event encoder's 2B)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmda_amd import ops


def make(B, H, W, C, hidden, dev):
    M = B * H * W
    x = torch.randn(M, C, device=dev)
    p = dict(x=x, gamma=torch.randn(C, device=dev), beta=torch.randn(C, device=dev),
             w1=(torch.randn(hidden, C, device=dev) * C ** -0.5).bfloat16(), b1=torch.randn(hidden, device=dev) * 0.1,
             wdw=torch.randn(9, hidden, device=dev) * 0.3, bdw=torch.randn(hidden, device=dev) * 0.1,
             w2=(torch.randn(C, hidden, device=dev) * hidden ** -0.5).bfloat16(), b2=torch.randn(C, device=dev) * 0.1,
             B=B, H=H, W=W, C=C, hidden=hidden, M=M)
    p['xn'] = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    p['h'] = torch.empty(M, hidden, dtype=torch.bfloat16, device=dev)
    p['y'] = torch.empty(M, C, device=dev)
    return p


def chain(p):
    M, C, hidden = p['M'], p['C'], p['hidden']
    ops.layernorm_fwd(p['x'], p['gamma'], p['beta'], 1e-6, out=p['xn'])
    ops.gemm(ops.plain_view(p['xn'], M, C), ops.plain_view(p['w1'], hidden, C), p['h'], M, hidden, C, dtype=1, bias=p['b1'])
    a = ops.dwconv_fwd(p['h'], p['wdw'], p['bdw'], p['B'], p['H'], p['W'], hidden, 1, 'gelu')
    ops.gemm(ops.plain_view(a, M, hidden), ops.plain_view(p['w2'], C, hidden), p['y'], M, C, hidden, dtype=1, bias=p['b2'], res=p['x'])
    return p['y']


def fused(p, save, lines=None):
    return ops.mixffn_fwd(p['x'], p['gamma'], p['beta'], 1e-6, p['w1'], p['b1'], p['wdw'], p['bdw'], p['w2'], p['b2'], None,
                          p['B'], p['H'], p['W'], save=save, lines=lines)[0]


def timeit(fns, iters=20, reps=5):
    """fns: one callable per lane; every lane repeats its callable `iters` times on its own stream inside ONE captured graph"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in fns[1:]]
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for st in side:
                st.wait_stream(s)
            for f, st in zip(fns[1:], side):
                with torch.cuda.stream(st):
                    for _ in range(iters):
                        f()
            for _ in range(iters):
                fns[0]()
            for st in side:
                s.wait_stream(st)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


dev = torch.device('cuda:0')
C, hidden = 320, 1280
print('single lane, 32 x 32 tokens per sample, C = 320, hidden = 1280 (us per MLP half)')
for B in (2, 4, 8):
    p = make(B, 32, 32, C, hidden, dev)
    ref = chain(p).clone()
    out = fused(p, False)
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    t_chain = timeit([lambda: chain(p)])
    row = f'  B = {B} (M = {p["M"]:5d}): four launches {t_chain:6.1f}'
    for lines in (2, 1):
        row += f' | fused R={lines}: no-save {timeit([lambda: fused(p, False, lines)]):6.1f}  save {timeit([lambda: fused(p, True, lines)]):6.1f}'
    gf = 4.0 * p['M'] * C * hidden / 1e9
    print(row + f' | {gf:.2f} GFLOP, max err vs chain {err:.1e}', flush=True)
print('two lanes side by side (image encoder B, event encoder 2B)')
for B in (2, 4):
    pa, pb = make(B, 32, 32, C, hidden, dev), make(2 * B, 32, 32, C, hidden, dev)
    t_chain = timeit([lambda: chain(pa), lambda: chain(pb)])
    t_f = timeit([lambda: fused(pa, False), lambda: fused(pb, False)])
    t_fs = timeit([lambda: fused(pa, True), lambda: fused(pb, True)])
    t_f1 = timeit([lambda: fused(pa, False, 1), lambda: fused(pb, False, 2)])
    print(f'  B = {B} + {2 * B}: four launches {t_chain:6.1f} | fused no-save {t_f:6.1f}  save {t_fs:6.1f}  (R=1 / R=2: {t_f1:6.1f})', flush=True)
print('other shapes')
for B, H, W, C2 in ((4, 16, 16, 256), (2, 28, 40, 320), (8, 16, 16, 320)):
    p = make(B, H, W, C2, 4 * C2, dev)
    ref = chain(p).clone()
    out = fused(p, False)
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print(f'  B={B} {H}x{W} C={C2}: four launches {timeit([lambda: chain(p)]):6.1f} | fused {timeit([lambda: fused(p, False)]):6.1f} | err {err:.1e}', flush=True)
