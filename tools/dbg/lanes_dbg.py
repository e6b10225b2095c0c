import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import test_dacs as T
import cmda_amd.runtime as rt
from cmda_amd.registry import build_train_model
from weights import seeded_fill
lanes = set(sys.argv[1].split(',')) if len(sys.argv) > 1 and sys.argv[1] != 'none' else set()
rt.set_compute_dtype(torch.float32)
dev = torch.device('cuda:0')
GEN = not os.environ.get('DBG_NO_GEN')
dacs = build_train_model(T.make_cfg(T.SMALL['dims'], T.SMALL['ch'], generator=GEN))
seeded_fill(dacs.model, 7); seeded_fill(dacs.ema_model, 8)
if GEN: seeded_fill(dacs.cyclegan_itrd2en, 9)
dacs.to(dev).train()
src, tg = T.make_batch(2, 64, 64)
batch = dict(source={k: v.to(dev) for k, v in src.items()}, target={k: v.to(dev) for k, v in tg.items()})
torch.manual_seed(11), random.seed(11), np.random.seed(11)
dacs.graph_lane_set = lanes
dacs.enable_graph(warmup_iters=1)
for it in range(3):
    for p in dacs.model.parameters():
        if p.grad is not None: p.grad.zero_()
    lv = dacs(**batch); torch.cuda.synchronize()
    print(sorted(lanes), it, {k: round(float(v), 5) for k, v in lv.items()}, flush=True)


def verify_cache():
    bad = 0
    for (pid, kind), e in rt._cache.items():
        p = e.ref()
        if p is None:
            continue
        d = list(e.dims) + [1] * (4 - len(e.dims)); pm = list(e.perm) + list(range(len(e.perm), 4))
        src = p.data.reshape(d)
        flips = [ax for ax in range(4) if (e.flip >> ax) & 1]
        if flips: src = src.flip(flips)
        want = src.permute(pm).contiguous().reshape(-1).to(e.dst.dtype)
        if not torch.equal(want, e.dst.reshape(-1)):
            bad += 1
            if bad < 4: print('  BAD entry', kind, tuple(p.shape), 'dst ptr', hex(e.dst.data_ptr()), 'max diff', (want.float() - e.dst.reshape(-1).float()).abs().max().item())
    print('  cache entries', len(rt._cache), 'bad', bad, 'plans', {k: (v['nblocks'], v['desc'].numel() // 64) for k, v in rt._plans.items()})


verify_cache()
