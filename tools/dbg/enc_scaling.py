"""How the MiT-B5 encoder's forward (one lane, replayed from a hipGraph) scales with the batch: per-launch time at B = 1, 2, 4, 8 --
latency-bound chains would not grow with B.  Per stage, with the launch count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cmda_amd  # noqa
from cmda_amd import ops, runtime as rt
from cmda_amd.registry import build_backbone

dev = torch.device('cuda:0')
rt.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = build_backbone(dict(type='mit_b5', style='pytorch', drop_path_rate=0.0)).to(dev).eval()
calls = [0]
_orig = ops.call
def counting(*a, **k):
    calls[0] += 1
    return _orig(*a, **k)


def time_graph(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


two = len(sys.argv) > 1 and sys.argv[1] == 'two'
only = os.environ.get('ENC_ONLY_B')
for save in ((True,) if only else (False, True)):
    for B in ((int(only),) if only else (1, 2, 4, 8)):
        img = torch.randn(B, 3, 512, 512, device=dev)
        with torch.no_grad():
            ops.call = counting; calls[0] = 0
            net.fwd(img, save=save)
            ops.call = _orig
            n = calls[0]
            t = time_graph(lambda: net.fwd(img, save=save))
        print(f'fwd save={save} B={B}: {t:7.3f} ms, {n} library calls, {t / n * 1e3:6.2f} us per call', flush=True)
