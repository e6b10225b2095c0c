"""Flat AdamW vs torch.optim.AdamW with the reference's paramwise rules; schedule vs the oracle restatement."""
import torch

from cmda_amd import optim
from conftest import assert_close
from oracle import uda as ouda


def test_flat_adamw_matches_torch(tgt):
    torch.manual_seed(0)
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm1 = torch.nn.LayerNorm(10)
            self.fc = torch.nn.Linear(10, 7)
            self.decode_head = torch.nn.Linear(7, 5)
    net, ref = Net().to(tgt.device), Net()
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    keys = dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0))
    opt = optim.FlatAdamW(net, lr=1e-3, weight_decay=0.01, custom_keys=keys)
    groups = []
    for n, p in ref.named_parameters():
        lr, wd = ouda.param_group_options(n, 1e-3, 0.01, keys)
        groups.append(dict(params=[p], lr=lr, weight_decay=wd))
    topt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for it in range(3):
        opt.zero_grad()
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            g = torch.randn(q.shape)
            q.grad = g.clone()
            p.grad.copy_(g)
        opt.step()
        topt.step()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        # three fused steps against torch.optim.AdamW: fp32 round-off of a handful of operations per element (1e-6 of the tensor's
        # largest element measured on the MI355X; the absolute term covers a near-zero bias whose whole range is 3e-3)
        assert_close(p.data, q.data, 2e-6, atol=5e-9, name=n)


def test_schedule_matches_oracle():
    for it in (0, 1, 700, 1499, 1500, 20000, 39999):
        assert abs(optim.poly_warm_scale(it) * 6e-5 - ouda.poly_warm_lr(6e-5, it)) < 1e-15
