"""Deterministic, construction-order-independent weights/inputs shared by make_golden.py and the tests: every tensor
is drawn from a torch CPU generator seeded by (seed, crc32(name)), so fixtures need not store any weights."""
import zlib

import torch


def _gen(seed, name):
    return torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))


def seeded_randn(shape, seed, name='x', scale=1.0):
    return torch.randn(tuple(shape), generator=_gen(seed, name)) * scale


def seeded_fill(module, seed):
    """Fill every parameter/buffer of `module` (keyed by state_dict name) with seeded values of sane magnitude."""
    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            g = _gen(seed, name)
            if name.endswith('num_batches_tracked'):
                t.zero_()
            elif name.endswith('running_var'):
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            elif name.endswith('running_mean'):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() == 1 and name.endswith('weight'):
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            elif t.dim() == 1:
                t.copy_(0.05 * torch.randn(t.shape, generator=g))
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (1.0 / fan_in ** 0.5))
    return module


def sample_grad(g, n=2048):
    """Compact fingerprint of a gradient tensor: strided sample + sum + abs-sum."""
    f = g.detach().flatten().float()
    stride = max(1, f.numel() // n)
    return torch.cat([f[::stride][:n], f.sum()[None], f.abs().sum()[None]])


def block_label(B, H, W, seed):
    """labels 0..18 constant on 8 x 8 blocks, 5 % ignore pixels (255)"""
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, 19, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    lab[torch.rand((B, 1, H, W), generator=g) < 0.05] = 255
    return lab


# the reference-generated DACS step fixture (tests/golden/dacs_step.npz, make_golden.py::dacs_step): model widths, classifier scale,
# seeds and the inputs of the step -- rebuilt from seeded generators by the generator script and by the tests alike
DACS_DIMS, DACS_CH, DACS_SEG_SCALE = [32, 64, 160, 256], 64, 12.0
DACS_SEEDS = dict(student=111, teacher=112, generator=113, batch=111)


def dacs_batch(B=1, H=512, W=512, seed=111):
    itr = seeded_randn((B, 1, H, W), seed, 'itr').clamp(-1, 1).repeat(1, 3, 1, 1)
    src = dict(image=seeded_randn((B, 3, H, W), seed, 'img'), img_time_res=itr,
               img_self_res=seeded_randn((B, 3, H, W), seed, 'isr').clamp(-1, 1), label=block_label(B, H, W, seed))
    tg = dict(warp_image=seeded_randn((B, 3, H, W), seed, 'nimg'), events_vg=seeded_randn((B, 3, H, W), seed, 'nev').clamp(-1, 1),
              warp_img_self_res=seeded_randn((B, 3, H, W), seed, 'nisr').clamp(-1, 1))
    return src, tg
