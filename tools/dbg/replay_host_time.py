"""host-side cost of one hipGraph replay of the DACS iteration vs its GPU duration"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import cmda_amd.runtime as rt
from cmda_amd import optim
rt.set_compute_dtype(torch.bfloat16)
dev = torch.device('cuda:0')
torch.manual_seed(1234)
dacs = bench.build_dacs(dev)
opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01, custom_keys=bench.CUSTOM_KEYS)
dacs.attach_flat_store(opt)
batch = bench.synthetic_pairs(2, 512, 100, dev)
if len(sys.argv) > 1:
    dacs.graph_lane_set = set(x for x in sys.argv[1].split(',') if x and x != 'none')
dacs.enable_graph(warmup_iters=2)
def step():
    opt.zero_grad(); dacs(**batch); opt.step(1.0)
for _ in range(4): step()
torch.cuda.synchronize()
G = dacs._graph['graph']
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); G.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'replay only: host {1e3*(t1-t0):.1f} ms, until done {1e3*(t2-t0):.1f} ms')
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'5 steps: host {1e3*(t1-t0)/5:.1f} ms/step, wall {1e3*(t2-t0)/5:.1f} ms/step')
