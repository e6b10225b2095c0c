"""CPU model of the generator's bf16 numerics (no GPU): the oracle generator with rounding injected where the HIP path rounds,
to size the error of each storage variant against the fp32 oracle before building it.
variants: 'r3' round-3 path (bf16 everywhere); 'res32' fp32 residual stream; 'res32+conv32' also fp32 conv outputs into the norm;
'x3' split-bf16 operands (hi + lo) for activations and weights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden'))
import torch, torch.nn as nn, torch.nn.functional as F
from oracle import cyclegan as ocg
from weights import seeded_fill, seeded_randn

def bf(x): return x.bfloat16().float()
def split(x):
    hi = bf(x); return hi, bf(x - hi)

def conv(x, m, variant, pad_reflect=0, pad_zero=0, stride=1, transposed=False):
    w, b = m.weight, m.bias
    if pad_reflect: x = F.pad(x, (pad_reflect,) * 4, mode='reflect')
    def op(a, ww):
        if transposed: return F.conv_transpose2d(a, ww, None, stride=2, padding=1, output_padding=1)
        return F.conv2d(a, ww, None, stride=stride, padding=pad_zero)
    if variant.startswith('x3'):
        ah, al = split(x); wh, wl = split(w)
        y = op(ah, wh) + op(ah, wl) + op(al, wh)
    else:
        y = op(bf(x), bf(w))
    return y + b.view(1, -1, 1, 1)

def inorm(x): return F.instance_norm(x, eps=1e-5)

def run(G, inp, variant):
    m = G.model
    conv32 = 'conv32' in variant or variant.startswith('x3')
    res32 = 'res32' in variant or variant.startswith('x3')
    st = (lambda t: t) if conv32 else bf          # conv output storage
    act = (lambda t: t) if variant == 'x3' else bf    # norm output storage (the next GEMM operand)
    x = act(torch.relu(inorm(st(conv(inp, m[1], variant, pad_reflect=3)))))
    x = act(torch.relu(inorm(st(conv(x, m[4], variant, pad_zero=1, stride=2)))))
    x = torch.relu(inorm(st(conv(x, m[7], variant, pad_zero=1, stride=2))))
    x = x if res32 else bf(x)
    for i in range(9):
        cb = m[10 + i].conv_block
        y = act(torch.relu(inorm(st(conv(x, cb[1], variant, pad_reflect=1)))))
        y = inorm(st(conv(y, cb[5], variant, pad_reflect=1)))
        x = x + y
        x = x if res32 else bf(x)
    x = act(torch.relu(inorm(st(conv(x, m[19], variant, transposed=True)))))
    x = act(torch.relu(inorm(st(conv(x, m[22], variant, transposed=True)))))
    return torch.tanh(conv(x, m[26], variant, pad_reflect=3))

if __name__ == '__main__':
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    torch.manual_seed(0)
    G = ocg.ResnetGenerator().eval()
    seeded_fill(G, 9)
    inp = seeded_randn((2, 1, S, S), 7, 'itr').clamp(-1, 1)
    with torch.no_grad():
        ref = G(inp)
        for v in ['r3', 'res32', 'res32+conv32', 'x3', 'x3-actbf16']:
            out = run(G, inp, v)
            d = (out - ref).abs()
            print(f'{v:14s} S={S}: max err {d.max().item():.3e}  99.9th pct {d.flatten().kthvalue(int(0.999 * d.numel())).values.item():.3e}  '
                  f'mean {d.mean().item():.3e}  (range {ref.abs().max().item():.3f})')
