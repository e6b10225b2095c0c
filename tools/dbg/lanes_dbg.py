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
dacs = build_train_model(T.make_cfg(T.SMALL['dims'], T.SMALL['ch']))
seeded_fill(dacs.model, 7); seeded_fill(dacs.ema_model, 8); seeded_fill(dacs.cyclegan_itrd2en, 9)
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
