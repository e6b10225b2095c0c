import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import test_dacs as T
import cmda_amd.runtime as rt
from cmda_amd.registry import build_train_model
from weights import seeded_fill
rt.set_compute_dtype(torch.float32)
dev = torch.device('cuda:0')
dacs = build_train_model(T.make_cfg(T.SMALL['dims'], T.SMALL['ch']))
seeded_fill(dacs.model, 7); seeded_fill(dacs.ema_model, 8); seeded_fill(dacs.cyclegan_itrd2en, 9)
dacs.to(dev).train()
src, tg = T.make_batch(2, 64, 64)
batch = dict(source={k: v.to(dev) for k, v in src.items()}, target={k: v.to(dev) for k, v in tg.items()})
torch.manual_seed(11), random.seed(11), np.random.seed(11)
def zero():
    for p in dacs.model.parameters():
        if p.grad is not None: p.grad.zero_()
def snap():
    out = {k: float(v) for k, v in lv.items()}
    ex = {k: v.detach().clone() for k, v in dacs.last_mix.items() if isinstance(v, torch.Tensor)}
    ex.update({'tl_' + k: v.detach().clone() for k, v in dacs.last_mix['teacher_logits'].items() if v is not None})
    g = torch.cat([p.grad.flatten() for p in dacs.model.parameters()]).clone()
    return out, ex, g
zero(); lv = dacs(**batch); d0 = dacs.last_draws
dacs.inject_draws = d0
zero(); lv = dacs(**batch); e1 = snap()
zero(); lv = dacs(**batch); e2 = snap()
print('eager it1 vs it2 loss', e1[0], e2[0])
dacs.enable_graph(warmup_iters=0)
for it in range(3):
    zero(); lv = dacs(**batch); torch.cuda.synchronize(); r = snap()
    print('replay', it, r[0])
    for k in e2[1]:
        a, b = e2[1][k].float(), r[1][k].float()
        print('   ', k, (a - b).abs().max().item())
    print('    grads', (e2[2] - r[2]).abs().max().item(), e2[2].abs().max().item())
