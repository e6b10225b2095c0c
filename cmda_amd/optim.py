"""Flat-buffer parameter store + fused AdamW (the optimiser of configs/_base_/schedules/adamw.py with the paramwise
rules of configs/fusion/*: `head` lr x10, `pos_block` / `norm` weight-decay 0; mmcv DefaultOptimizerConstructor
semantics restated in SURVEY.md appendix D) and the poly(1.0) + linear warm-up schedule (schedules/poly10warm.py).

All trainable parameters of a model are re-homed into ONE fp32 buffer (grouped by (lr_mult, decay_mult)), their
gradients into a second one, so that zero_grad is one memset, the data-parallel gradient exchange is a handful of
large RCCL calls over contiguous memory, and the AdamW update is one kernel launch per group (cmda_adamw_step).
"""
import re

import torch

from . import ops
from . import runtime as rt


def _backward_rank(name):
    """0 = gradients final first.  Stable for everything else (python's sort keeps the original order inside a rank)."""
    m = re.search(r'(?:patch_embed|block|norm)([1-4])\.', name)
    if 'decode_head' in name or 'fusion' in name:
        return 0
    return 5 - int(m.group(1)) if m else 5


def _group_of(name, custom_keys):
    for key in sorted(sorted(custom_keys.keys()), key=len, reverse=True):
        if key in name:
            return (custom_keys[key].get('lr_mult', 1.0), custom_keys[key].get('decay_mult', 1.0))
    return (1.0, 1.0)


class FlatAdamW:
    def __init__(self, model, lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, custom_keys=None,
                 name_prefix=''):
        custom_keys = custom_keys or {}
        named = [(name_prefix + n, p) for n, p in model.named_parameters() if p.requires_grad]
        seen, uniq = set(), []
        for n, p in named:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append((n, p))
        # inside a group, parameters sit in the order their gradients become final in the backward pass (decode head, then
        # backbone stage 4 ... 1), so each stage's weights are ONE contiguous slice the all-reduce can start on early
        uniq.sort(key=lambda np_: _backward_rank(np_[0]))
        groups = {}
        for n, p in uniq:
            groups.setdefault(_group_of(n, custom_keys), []).append(p)
        self._names = {id(p): n for n, p in uniq}
        self._order = [p for _, p in named if id(p) in seen]          # model.named_parameters() order, unique
        self._order = list({id(p): p for p in self._order}.values())
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = uniq[0][1].device
        total = sum((p.numel() + 7) // 8 * 8 for _, p in uniq)  # 8-element slots: 16-byte aligned in fp32 AND in the bf16 mirror
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.segments = []  # (start, end, lr_mult, decay_mult)
        off = 0
        with torch.no_grad():
            for (lm, dm), ps in sorted(groups.items()):
                start = off
                for p in ps:
                    n = p.numel()
                    if _khwc_candidate(p):
                        # CHANNELS-LAST storage: the slot holds [Co,KH,KW,Ci] -- the implicit GEMM's own weight layout -- and the
                        # parameter is a permuted view of it with the reference's logical shape [Co,Ci,KH,KW] (state_dict, copy_,
                        # the oracle comparisons all see that shape).  The bf16 mirror the AdamW kernel maintains IS then the GEMM
                        # operand, and the weight-gradient GEMM accumulates straight into the gradient's memory: neither the per-step
                        # re-layout of ~100 M conv weights (0.7 ms) nor the gradient-shadow drain (0.5 ms) exists for these.
                        Co, Ci, KH, KW = p.shape
                        self.flat_p[off:off + n].copy_(p.data.permute(0, 2, 3, 1).reshape(-1))
                        p.data = self.flat_p[off:off + n].view(Co, KH, KW, Ci).permute(0, 3, 1, 2)
                        p.grad = self.flat_g[off:off + n].view(Co, KH, KW, Ci).permute(0, 3, 1, 2)
                    else:
                        self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                        p.data = self.flat_p[off:off + n].view(p.shape)
                        p.grad = self.flat_g[off:off + n].view(p.shape)
                    off += (n + 7) // 8 * 8
                self.segments.append((start, off, lm, dm))
        # bf16 compute copies of every parameter, kept current by the AdamW kernel itself (no per-step cast launches)
        self.flat_bf16 = None
        if rt.compute_dtype() == torch.bfloat16 and self.flat_p.is_cuda:
            self.flat_bf16 = torch.empty(total, dtype=torch.bfloat16, device=dev)
            base = self.flat_p.data_ptr()
            for _, p in uniq:
                o = (p.data.data_ptr() - base) // 4
                p._cmda_bf16 = _slot_view(self.flat_bf16, o, p)
            self.sync_bf16()
        self.step_count = 0
        rt.invalidate()

    def sync_bf16(self):
        """Refresh the bf16 copies after the masters were written by anything but step() (e.g. load_state_dict)."""
        if hasattr(self, 'segments'):
            self.synchronize()
        if self.flat_bf16 is not None:
            ops.permute4(self.flat_p, self.flat_bf16, (self.flat_p.numel(), 1, 1, 1), (0, 1, 2, 3))

    def ranges_of(self, model, prefixes, min_elems=1 << 20):
        """Contiguous [lo, hi) slices of the flat buffers covering the parameters whose name starts with one of
        `prefixes` (adjacent parameters merged; slices shorter than `min_elems` are dropped -- the caller's final
        `finish()` picks those up)."""
        base = self.flat_p.data_ptr()
        spans = []
        for n, p in model.named_parameters():
            if p.requires_grad and any(n.startswith(pre) for pre in prefixes):
                lo = (p.data.data_ptr() - base) // 4
                spans.append((lo, lo + (p.numel() + 7) // 8 * 8))
        out = []
        for lo, hi in sorted(set(spans)):
            if out and lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return [(lo, hi) for lo, hi in out if hi - lo >= min_elems]

    # -- overlapped update (opt-in: `overlap = True`) ------------------------------------------------------------------------------------
    # The update, the gradient clear behind it and (uda.DACS) the EMA teacher update run on a stream of their own, ordered after
    # everything the calling stream has enqueued.  Whoever reads the parameters next must order itself behind that stream:
    # `synchronize()` on an ordinary stream, or -- the captured DACS iteration -- a `runtime.wait_external` step placed behind the
    # part of the iteration that needs no trainable weight (mixing, the frozen generator), which then runs underneath the update.
    overlap = False

    def update_stream(self):
        s = getattr(self, '_update_stream', None)
        if s is None:
            s = self._update_stream = torch.cuda.Stream(self.flat_p.device)
        return s

    def _on_update_stream(self):
        import contextlib
        if not (self.overlap and self.flat_p.is_cuda) or torch.cuda.is_current_stream_capturing():
            return contextlib.nullcontext()
        s = self.update_stream()
        s.wait_stream(torch.cuda.current_stream(self.flat_p.device))
        return torch.cuda.stream(s)

    def flush(self):
        """launch what an overlapped `step()` / `zero_grad()` postponed (no-op otherwise) -- on the update stream, behind everything the
        calling stream has enqueued so far.  The postponement exists for ONE reason: the captured DACS iteration stages its inputs
        (~60 MB of device copies) right after the step boundary, and next to AdamW's 5 TB/s stream those copies took 0.9 ms instead
        of 0.06 and held back the whole iteration (kernel trace, round 6); uda.DACS.forward_train enqueues the copies first, then flushes."""
        todo, self._todo = getattr(self, '_todo', None), None
        if not todo:
            return
        with self._on_update_stream():
            for what, arg in todo:
                if what == 'step':
                    self._launch_step(arg)
                else:
                    self.flat_g.zero_()

    def synchronize(self):
        """order the current stream behind the overlapped update, launching it first if it is still postponed (no-op otherwise)"""
        self.flush()
        if self.overlap and self.flat_p.is_cuda and getattr(self, '_update_stream', None) is not None:
            torch.cuda.current_stream(self.flat_p.device).wait_stream(self._update_stream)

    def _postpone(self):
        return self.overlap and self.flat_p.is_cuda and not torch.cuda.is_current_stream_capturing()

    def zero_grad(self):
        if self._postpone() and getattr(self, '_todo', None):   # behind a postponed step: AdamW reads these gradients first
            self._todo.append(('zero', None))
            return
        with self._on_update_stream():
            self.flat_g.zero_()

    def _launch_step(self, args):
        lr_scale, step_count = args
        for start, end, lm, dm in self.segments:
            ops.adamw_step(self.flat_p[start:end], self.flat_g[start:end], self.flat_m[start:end], self.flat_v[start:end],
                           self.lr * lm * lr_scale, self.betas[0], self.betas[1], self.eps, self.weight_decay * dm,
                           step_count, p_bf16=self.flat_bf16[start:end] if self.flat_bf16 is not None else None)

    def step(self, lr_scale=1.0):
        self.step_count += 1
        if self._postpone():
            self._todo = (getattr(self, '_todo', None) or []) + [('step', (lr_scale, self.step_count))]
        else:
            self._launch_step((lr_scale, self.step_count))
        rt.invalidate()

    # -- torch.optim.AdamW-compatible state (the layout mmcv's CheckpointHook saves and runner.resume restores) ----------------
    def _slot(self, p):
        o = (p.data.data_ptr() - self.flat_p.data_ptr()) // 4
        return o, o + p.numel()

    def state_dict(self):
        self.synchronize()
        return self._state_dict()

    def _state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} keyed by parameter index in
        model.named_parameters() order, one group per parameter (mmcv's DefaultOptimizerConstructor with paramwise_cfg builds
        one group per parameter), tensors on the CPU."""
        state, groups = {}, []
        seg_of = lambda o: next((lm, dm) for s, e, lm, dm in self.segments if s <= o < e)  # noqa: E731
        for i, p in enumerate(self._order):
            lo, hi = self._slot(p)
            lm, dm = seg_of(lo)
            if self.step_count > 0:
                state[i] = {'step': self.step_count, 'exp_avg': _slot_view(self.flat_m, lo, p).detach().cpu().contiguous().clone(),
                            'exp_avg_sq': _slot_view(self.flat_v, lo, p).detach().cpu().contiguous().clone()}
            groups.append({'lr': self.lr * lm, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.weight_decay * dm,
                           'amsgrad': False, 'params': [i]})
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        """restores the Adam moments and the step count (same parameter order as state_dict / torch.optim.AdamW)"""
        self.synchronize()
        steps = set()
        with torch.no_grad():
            for i, p in enumerate(self._order):
                st = sd['state'].get(i, sd['state'].get(str(i)))
                if st is None:
                    continue
                lo, hi = self._slot(p)
                _slot_view(self.flat_m, lo, p).copy_(st['exp_avg'].view(p.shape))
                _slot_view(self.flat_v, lo, p).copy_(st['exp_avg_sq'].view(p.shape))
                steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): the fused update keeps one step count')
        self.step_count = steps.pop() if steps else 0


def khwc_stored(p):
    """True when the 4-D parameter p ([Co,Ci,KH,KW] logically) is stored [Co,KH,KW,Ci] in memory (see FlatAdamW)"""
    return p.dim() == 4 and not p.is_contiguous() and p.permute(0, 2, 3, 1).is_contiguous()


def _khwc_candidate(p):
    """convolution weights whose implicit-GEMM layout is a plain re-ordering of the parameter: [Co,Ci,KH,KW] with Ci % 8 == 0
    (not depthwise [C,1,3,3], not the 3-channel patch embed, which runs on a padded copy)"""
    return p.dim() == 4 and p.shape[1] % 8 == 0 and p.shape[2] * p.shape[3] > 1


def _slot_view(flat, off, p):
    """p's slot of a flat buffer with p's logical shape: a permuted view of [Co,KH,KW,Ci] memory for khwc-stored parameters"""
    n = p.numel()
    if khwc_stored(p):
        Co, Ci, KH, KW = p.shape
        return flat[off:off + n].view(Co, KH, KW, Ci).permute(0, 3, 1, 2)
    return flat[off:off + n].view(p.shape)


def poly_warm_scale(it, max_iters=40000, power=1.0, warmup_iters=1500, warmup_ratio=1e-6):
    """lr(it) / base_lr for poly(power) decay to 0 with linear warm-up (mmcv PolyLrUpdaterHook + 'linear' warmup)."""
    s = (1 - it / max_iters) ** power
    if it < warmup_iters:
        s *= 1 - (1 - it / warmup_iters) * (1 - warmup_ratio)
    return s
