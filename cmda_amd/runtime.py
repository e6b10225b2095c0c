"""Runtime policy shared by the cmda_amd modules: compute dtype, compute-layout copies of the fp32 master
parameters, and the fp32 gradient buffers the HIP kernels accumulate into.

Parameters keep the reference's names, shapes and fp32 storage (state_dict compatible with mit_b5.pth / CMDA
checkpoints).  Kernels consume *compute copies*:
  Linear  weight [N,K]          -> same layout, compute dtype (no copy in fp32 mode)
  Conv2d  weight [Co,Ci,KH,KW]  -> 'khwc'  [Co,KH,KW,Ci]           (implicit-GEMM forward / wgrad layout)
                                -> 'dgrad' [Ci,KH,KW,Co], taps flipped (implicit-GEMM data gradient)
The copies are cached and must be dropped with `invalidate()` after every optimizer step (the fused AdamW kernel
writes the masters without touching torch's version counters).
Parameter gradients are accumulated by the kernels straight into `param.grad` (fp32, allocated zero on first use):
autograd never sees them, which is what lets two backward passes per step accumulate with no extra traffic.
"""
import warnings
import weakref

import torch

from . import ops

_state = {'dtype': torch.float32}
_cache = {}
_frozen = {}   # compute copies of parameters marked `_cmda_frozen` (the Motion-Extractor generator): never invalidated


def set_compute_dtype(dtype):
    assert dtype in (torch.float32, torch.bfloat16)
    _state['dtype'] = dtype
    _frozen.clear()
    _cache.clear()
    _plans.clear()


def compute_dtype():
    return _state['dtype']


def set_residual_fp32(flag):
    """bf16 mode only: keep the encoders' residual stream (the block input x of mix_transformer.py:134,146, summed over up to 52
    blocks) in fp32 -- every LayerNorm reads fp32 and writes bf16 for the GEMM behind it, proj / fc2 add their fp32 residual in the
    epilogue.  Costs ~2x the bytes of the (small) stream tensors; buys back most of the bf16 mode's logit error."""
    _state['res32'] = bool(flag)


def residual_fp32():
    return _state.get('res32', True) and _state['dtype'] == torch.bfloat16


def stream_dtype():
    """storage type of the encoders' residual stream"""
    return torch.float32 if residual_fp32() else _state['dtype']


def set_gemm_x3(flag):
    """fp32 mode only: run every GEMM / implicit-GEMM convolution on SPLIT-bf16 operands (cmda_gemm_params_t.dtype =
    CMDA_F32X3: x = hi + lo, three bf16 MFMAs per k-step, fp32 accumulate, ~1e-5 relative -- csrc/gemm_x3.hip) instead of the
    exact-fp32 matrix instruction, which runs at 1/16 of the bf16 rate.  Storage, statistics and every other kernel stay fp32: the
    tolerance-meeting mode (logits within 1e-3 of the reference) at a fraction of the exact mode's cost.  Off by default: the exact
    mode is what the kernel-level parity tests pin."""
    _state['x3'] = bool(flag)


def gemm_x3():
    return bool(_state.get('x3')) and _state['dtype'] == torch.float32


def tag():
    """cmda_gemm_params_t.dtype of the GEMMs: 0 exact fp32, 1 bf16, 2 fp32 storage with split-bf16 MFMA"""
    if _state['dtype'] == torch.float32:
        return 2 if _state.get('x3') else 0
    return 1


def invalidate():
    """The fp32 masters changed (optimizer step, EMA update, checkpoint load): every compute copy made so far is stale.
    The copies keep their storage (their addresses may be baked into a captured hipGraph); the next `refresh()` -- the first
    use of any copy, or the explicit call at the head of a training iteration -- rewrites ALL of them with ONE batched launch
    (cmda_permute4_batch); the per-tensor lazy re-layouts this replaces were ~470 launches per iteration of a step that is
    bound by its launch count."""
    _state['stale'] = True


def refresh(force=False):
    """bring the compute copies up to date (one launch); `force`: launch even if nothing was invalidated -- a captured
    iteration starts with this node so that every replay sees the masters of its own iteration"""
    if force or _state.get('stale'):
        _state['stale'] = False
        _refresh()


class _Entry:
    __slots__ = ('ref', 'dst', 'dims', 'perm', 'flip')

    def __init__(self, param, dst, dims, perm, flip):
        # flip: bits 0-3 reversed source axes; bits 8-10 padded source axis + 1, bits 16.. its real extent (cmda_permute_desc_t)
        self.ref, self.dst, self.dims, self.perm, self.flip = weakref.ref(param), dst, dims, perm, flip


def _refresh():
    dead = [k for k, e in _cache.items() if e.ref() is None or e.ref().device != e.dst.device]
    for k in dead:
        del _cache[k]
    if not _cache:
        return
    live = list(_cache.values())
    by_dev = {}
    for e in live:
        by_dev.setdefault(e.dst.device, []).append(e)
    for dev, entries in by_dev.items():
        # the plan of an entry set: descriptor / block tables on the device.  Keyed by what the tables hold (source and
        # destination ADDRESSES and the geometry) -- not by object ids, which CPython reuses -- and kept per key: switching between
        # two entry sets (training / evaluation model) does not rebuild, and a plan whose launch was CAPTURED into a hipGraph is
        # never dropped (`_pinned`: the graph bakes in the table pointers; the plan also holds the copies' and the masters'
        # storage, so a replay never reads or writes recycled memory even if a module of that set died meanwhile).
        key = (dev, tuple((e.ref().data.data_ptr(), e.dst.data_ptr(), tuple(e.dims), tuple(e.perm), e.flip) for e in entries))
        plan = _plans.get(key)
        if plan is None:
            import numpy as np
            desc = np.zeros(len(entries), dtype=[('src', '<u8'), ('dst', '<u8'), ('d', '<i4', 4), ('p', '<i4', 4), ('flip', '<i4'),
                                                   ('bf16', '<i4'), ('total', '<i8')])
            blocks = []
            for t, e in enumerate(entries):
                d = list(e.dims) + [1] * (4 - len(e.dims))
                pm = list(e.perm) + list(range(len(e.perm), 4))
                total = 1
                for v in d:
                    total *= v
                desc[t] = (e.ref().data.data_ptr(), e.dst.data_ptr(), d, pm, e.flip, int(e.dst.dtype == torch.bfloat16), total)
                blocks += [(t, c) for c in range((total + 1023) // 1024)]
            if len(_plans) > 8:   # un-captured plans of entry sets that are gone
                for k in [k for k, pl in _plans.items() if not pl.get('pinned')]:
                    del _plans[k]
            plan = _plans[key] = dict(key=key, nblocks=len(blocks),
                                      desc=torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev),
                                      blocks=torch.tensor(blocks, dtype=torch.int32).to(dev),
                                      hold=[(e.dst, e.ref().data) for e in entries])
        if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            plan['pinned'] = True
        ops.permute4_batch(plan['desc'], plan['blocks'], plan['nblocks'])


def refresh_frozen(module=None):
    """rewrite the compute copies of FROZEN parameters (`_cmda_frozen`: the Motion-Extractor generator) from their masters --
    they are outside invalidate() / refresh(), so a checkpoint load that restores `cyclegan_itrd2en.*` after a forward pass must
    call this (checkpoint.load_state_dict does).  module: only that module's parameters (default: all)."""
    ids = None if module is None else {id(p) for p in module.parameters()}
    for (pid, _kind), e in list(_frozen.items()):
        prm = e.ref()
        if prm is None:
            del _frozen[(pid, _kind)]
            continue
        if ids is not None and pid not in ids:
            continue
        if e.dst.device != prm.device:
            del _frozen[(pid, _kind)]
            continue
        if e.flip >> 8:   # channel-padded copy (the generator's first convolution in the bf16 mode): the batched re-layout only
            _fill_padded(prm, e.dst, e.dims, e.perm, e.flip)
        else:
            ops.permute4(_mem_view(prm), e.dst, e.dims, e.perm, flipmask=e.flip)


_plans = {}


def _mem_view(param):
    """the parameter's storage as a contiguous tensor (channels-last stored conv weights: [Co,KH,KW,Ci])"""
    d = param.data
    return d if d.is_contiguous() else d.permute(0, 2, 3, 1)


def _fill_padded(param, dst, dims, perm, flip):
    """(re)write a channel-padded compute copy through a one-entry cmda_permute4_batch plan -- only the batched re-layout knows the
    padding bits of `flip` (cmda_permute_desc_t.flipmask)"""
    import numpy as np
    d = list(dims) + [1] * (4 - len(dims))
    total = 1
    for v in d:
        total *= v
    desc = np.zeros(1, dtype=[('src', '<u8'), ('dst', '<u8'), ('d', '<i4', 4), ('p', '<i4', 4), ('flip', '<i4'), ('bf16', '<i4'), ('total', '<i8')])
    desc[0] = (param.data.data_ptr(), dst.data_ptr(), d, list(perm) + list(range(len(perm), 4)), flip, int(dst.dtype == torch.bfloat16), total)
    blocks = [(0, c) for c in range((total + 1023) // 1024)]
    dd = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(param.device)
    bb = torch.tensor(blocks, dtype=torch.int32).to(param.device)
    ops.permute4_batch(dd, bb, len(blocks))
    dst._keep = (dd, bb)   # (the launch reads them asynchronously)


def _compute_copy(param, kind, dst_shape, dst_dtype, dims, perm, flip=0):
    key = (id(param), kind)
    store = _frozen if getattr(param, '_cmda_frozen', False) else _cache
    if _state.get('stale') and store is _cache:
        refresh()
    e = store.get(key)
    if e is not None and (e.ref() is not param or e.dst.device != param.device or e.dst.dtype != dst_dtype):
        e = None
    if e is None:
        if flip >> 8:   # channel-padded copy: only the batched re-layout knows the padding -- fill it through a one-entry plan
            dst = torch.zeros(dst_shape, dtype=dst_dtype, device=param.device)
            e = store[key] = _Entry(param, dst, dims, perm, flip)
            _fill_padded(param, dst, dims, perm, flip)
        else:
            dst = torch.empty(dst_shape, dtype=dst_dtype, device=param.device)
            ops.permute4(_mem_view(param), dst, dims, perm, flipmask=flip)
            e = store[key] = _Entry(param, dst, dims, perm, flip)
    return e.dst


def w(param):
    """Linear weight [N,K] in the compute dtype."""
    if _state['dtype'] == torch.float32:
        return param.data
    live = getattr(param, '_cmda_bf16', None)  # maintained by optim.FlatAdamW's fused update
    if live is not None:
        return live
    return _compute_copy(param, 'w', param.shape, _state['dtype'], (param.numel(), 1, 1, 1), (0, 1, 2, 3))


def wconv(param, kind='khwc', ci_pad=0):
    """Conv weight [Co,Ci,KH,KW] repacked for the implicit GEMM (see module docstring).  ci_pad > Ci: the input channels padded
    with zeros ([Co, KH, KW, ci_pad]; the 3-channel patch embed -> 8, conv_channel_pad)."""
    Co, Ci, KH, KW = param.shape
    if not param.is_contiguous() and param.permute(0, 2, 3, 1).is_contiguous():
        # CHANNELS-LAST stored parameter (optim.FlatAdamW): its memory -- in the bf16 mode the mirror the AdamW / EMA kernels keep
        # current -- already IS the implicit-GEMM layout [Co, KH*KW*Ci]
        if kind == 'khwc' and ci_pad <= Ci:
            if _state['dtype'] == torch.float32:
                return param.data.permute(0, 2, 3, 1).reshape(Co, KH * KW * Ci)
            live = getattr(param, '_cmda_bf16', None)
            if live is not None:
                return live.permute(0, 2, 3, 1).reshape(Co, KH * KW * Ci)
            return _compute_copy(param, 'khwc_cl', (Co, KH * KW * Ci), _state['dtype'], (Co * KH * KW * Ci, 1, 1, 1), (0, 1, 2, 3))
        if kind == 'dgrad':   # [Ci,KH,KW,Co] with the taps flipped, from [Co,KH,KW,Ci] memory
            return _compute_copy(param, 'dgrad_cl', (Ci, KH * KW * Co), _state['dtype'], (Co, KH, KW, Ci), (3, 1, 2, 0), flip=0b0110)
    if kind == 'khwc' and ci_pad > Ci:
        return _compute_copy(param, f'khwc{ci_pad}', (Co, KH * KW * ci_pad), _state['dtype'], (Co, ci_pad, KH, KW), (0, 2, 3, 1),
                             flip=(2 << 8) | (Ci << 16))
    if kind == 'khwc':
        return _compute_copy(param, kind, (Co, KH * KW * Ci), _state['dtype'], (Co, Ci, KH, KW), (0, 2, 3, 1))
    return _compute_copy(param, kind, (Ci, KH * KW * Co), _state['dtype'], (Co, Ci, KH, KW), (1, 2, 3, 0), flip=0b1100)


def conv_channel_pad(cin):
    """input channels the encoder's first convolution runs with: 3 -> 8 in the bf16 mode (16-byte im2col chunks: the LDS-DMA GEMM
    path instead of the register-staged one, 141 -> ~25 us for 4 x 512 x 512 images); unchanged otherwise"""
    return 8 if (_state['dtype'] == torch.bfloat16 and cin < 8) else cin


def wdw(param):
    """Depthwise 3x3 weight [C,1,3,3] -> tap-major fp32 [9,C] (coalesced per-tap reads in dwconv.hip)."""
    C = param.shape[0]
    return _compute_copy(param, 'dw', (9, C), torch.float32, (C, 9, 1, 1), (1, 0, 2, 3))


def grad(param):
    if param.grad is None:
        param.grad = torch.zeros_like(param.data)
    return param.grad


def act_empty(*shape, device):
    return torch.empty(*shape, dtype=_state['dtype'], device=device)


# Test taps: when a dict is installed here, the stochastic decisions of a pass drawn ON THE DEVICE (DropPath keep masks,
# Dropout2d masks) are published under a name -- the dict then holds the very tensors the kernels read, so after a hipGraph
# replay it shows that replay's draws (tests/test_dacs.py: replayed segments must draw fresh masks, and the oracle fed the same
# masks must reproduce the replayed iteration).  None (the default): nothing is kept.
taps = None


def tap(name, tensor):
    if taps is not None:
        taps[name] = tensor


_ones = {}


def ones1(device):
    """fp32 [1] holding 1.0: the d(loss) seed of a hand-scheduled backward pass"""
    t = _ones.get(device)
    if t is None:
        t = _ones[device] = torch.ones(1, dtype=torch.float32, device=device)
    return t


# ------------------------------------------------------------------ concurrency lanes + segmented hipGraph capture
# At the reference's 2+2 samples per GPU most kernels are far too small for 256 CUs and a dependent kernel costs ~5 us of
# launch / drain latency whatever its size, so the step time is (number of dependent kernels) x latency.  Independent work is
# therefore issued on side streams ("lanes"): 'enc' = the two encoders of the fusion student side by side; 'T' = teacher ->
# mixing -> mixed forward next to the source pass (uda.DACS._iteration); 'wgrad' = every weight-gradient kernel off the
# critical dgrad chain.  Off by default; DACS switches it on while it captures (eager launches would only pay host time).
#
# Measured on MI355X, full DACS step at 2+2 samples (ms per step, gpurun r02d-f), lanes as forked streams inside ONE captured
# graph: no lanes 146; enc 104; T 128; enc+T 102; wgrad alone 156, enc+wgrad 129-138, wgrad+T 154 -- the fine-grained wgrad
# forks (one cross-stream edge per weight gradient, ~3 k per step) cost more than they hide.
# But: hipGraphLaunch of a graph WITH parallel branches costs the host ~7 us per node (76 ms for the 11 k-node iteration: the
# step was host-bound inside the launch call), while a LINEAR graph launches at ~0.3 us per node (3.5 ms).  Hence the
# segmented form (SegmentedCapture): every stretch of one lane between two fork / join points is captured as its OWN
# single-stream graph, all sharing one memory pool, and the lanes are real streams at replay time -- the recorded program is
# a short list of `replay segment k on stream s` and `stream a waits for stream b` steps (~40 per iteration).  Segmented,
# GPU-bound (host 4 ms per iteration): no lanes 151; enc 104-107; T 124; enc+T 109-114 -- a THIRD concurrent queue costs more
# than it hides (GPU_MAX_HW_QUEUES=8: 176), so the default set is {enc}: two queues, everything that can run in pairs does.
# Also measured: the weight gradients of every transformer block as ONE side-lane entry after the block's dgrad chain
# (lane_batch('wgrad'): 4 queues, ~200 forks per step): 83.0 against 79.8 ms -- kept as an option, off; and the event
# encoder's two input sets as separate passes on a second side lane (3 queues, +50 % event-encoder launches): 95.6 against 79.6;
# the four ASPP branches of the decode head split over the two lanes (HBM-bound stencils / BatchNorm next to the pointwise GEMMs): the
# branches take twice as long side by side -- no gain, and the generator loses its slot next to the teacher's decoder.
# Round 6 (segmented replay, ms per step, same box, three alternating runs each; profiles/r06_lanes_ab.txt): {enc} 55.4-55.9;
# {enc, T} in the EARLY-STUDENT form (uda.DACS._iteration: the teacher on lane T from the start of the iteration, its encoders one after
# the other, joined in front of the decode head's loss) 54.3-54.8; + 'Tenc' (the teacher's event encoder on a FOURTH queue, main/T/enc)
# 52.3-52.7; + 'wq' (the encoders' grouped weight gradients on the teacher's two queues, idle by then: LANE_ALIAS) 52.05-52.3 -- the
# default of the captured DACS iteration (uda.GRAPH_LANES).  Four streams = HIP's four hardware queues: every lane stream is chosen so
# that it really runs beside the others (prepare_lane_streams; two torch streams may share a queue), and GPU_MAX_HW_QUEUES=8 ran the
# same four-lane step at 82-85 ms.
_conc = {'on': False, 'streams': {}, 'stack': ['main'], 'sstack': [], 'used': {}, 'keep': {}, 'enabled': {'enc'}, 'seg': None, 'seen': set()}


class SegmentedCapture:
    """records an iteration as linear hipGraph segments + the stream program that orders them (see above)"""

    def __init__(self, device):
        self.device = device
        self.pool = torch.cuda.graph_pool_handle()
        self.main = torch.cuda.Stream(device)
        self.program = []      # ('replay', graph, stream) | ('wait', waiter, waited) | ('call', fn, stream)
        self.active = None
        self.n_nodes = 0

    def begin(self, stream):
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(self.device)
        ctx = torch.cuda.stream(stream)
        ctx.__enter__()
        g.capture_begin(self.pool)
        self.active = (g, stream, ctx)

    def end(self):
        g, stream, ctx = self.active
        with warnings.catch_warnings():   # a stretch between two joins may hold no kernel at all: an empty segment is fine
            warnings.simplefilter('ignore', UserWarning)
            g.capture_end()
        ctx.__exit__(None, None, None)
        self.program.append(('replay', g, stream))
        self.active = None

    def cut(self, waits=(), resume=None):
        """close the running segment, record `waits` = [(waiter, waited)], open the next segment on `resume`"""
        stream = self.active[1]
        self.end()
        for waiter, waited in waits:
            self.program.append(('wait', waiter, waited))
        self.begin(resume if resume is not None else stream)

    def call(self, fn):
        """close the running segment and record a HOST call at this point of its stream: at replay `fn()` runs with that stream
        current, after everything recorded so far has been enqueued on it (a gradient exchange starts here, on its own stream,
        underneath the segments that follow)"""
        stream = self.active[1]
        self.end()
        self.program.append(('call', fn, stream))
        self.begin(stream)

    def replay(self, timeline=None):
        """timeline: a list to receive (program index, stream id, start event, end event) per segment (tools/lanes_timeline.py)"""
        cur = torch.cuda.current_stream(self.device)
        self.main.wait_stream(cur)
        for i, (op, a, b) in enumerate(self.program):
            if op == 'replay':
                with torch.cuda.stream(b):
                    if timeline is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        a.replay()
                        e1.record()
                        timeline.append((i, id(b), e0, e1))
                    else:
                        a.replay()
            elif op == 'call':
                with torch.cuda.stream(b):
                    a()
            elif op == 'wait_ext':   # a stream outside the program (the overlapped optimizer update), looked up at replay time
                ext = b()
                if ext is not None:
                    a.wait_stream(ext)
            else:
                a.wait_stream(b)
        cur.wait_stream(self.main)


def _runs_beside(a, b, device):
    """does work queued on stream `a` overtake a long run of kernels queued earlier on stream `b`?  (False: the two HIP streams share
    a hardware queue -- HIP maps its streams onto a few of them (GPU_MAX_HW_QUEUES, default 4) in creation order -- and serialise)"""
    import time
    x = torch.empty(64 << 20, dtype=torch.float32, device=device)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(b):
        for _ in range(40):          # ~4 ms of HBM-bound kernels
            x.mul_(1.0)
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        y = torch.zeros(64, dtype=torch.float32, device=device)
        y.add_(1.0)
    a.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize(device)
    del x, y
    return dt < 1.5e-3


def prepare_lane_streams(device, main_stream, lanes):
    """Create the side streams of the named lanes (full names: 'main/enc', 'main/T') BEFORE a capture, each on a hardware queue of
    its own: a lane whose stream shares a queue with another lane's only runs when that one is idle (measured, round 6: the
    generator on 'main/enc' started when the teacher on 'main/T' had finished, 12.5 ms into the iteration, tools/lanes_timeline.py).
    Candidates are drawn from torch's stream pool until one overtakes long-running work on the main stream and on every lane
    stream chosen so far (`_runs_beside`); without such a candidate the last one drawn is kept."""
    if device.type != 'cuda':
        return
    chosen = [main_stream]
    for full in lanes:
        key = (full, str(device))   # (the key of lane.__enter__)
        s = _conc['streams'].get(key)
        if s is not None and all(_runs_beside(s, o, device) and _runs_beside(o, s, device) for o in chosen):
            chosen.append(s)
            continue
        cand = None
        for _ in range(24):
            cand = torch.cuda.Stream(device)
            if all(c is not cand for c in chosen) and all(_runs_beside(cand, o, device) and _runs_beside(o, cand, device) for o in chosen):
                break
        _conc['streams'][key] = cand
        chosen.append(cand)


def set_concurrency(flag, lanes=None, seg=None):
    """lanes: subset of {'enc', 'wgrad', 'T'} to use (default: enc, see above); seg: a SegmentedCapture whose main segment
    is already open -- lanes then cut segments instead of forking streams inside one capture"""
    if lanes is not None:
        _conc['enabled'] = set(lanes)
    _conc['on'] = bool(flag)
    _conc['seg'] = seg if flag else None
    _conc['stack'], _conc['used'], _conc['keep'], _conc['seen'], _conc['batch'] = ['main'], {}, {}, set(), None
    _conc['sstack'] = [seg.main] if (flag and seg is not None) else []
    ops.LN_LANE = 'main'


def concurrency():
    return _conc['on']


def lane_enabled(name):
    return _conc['on'] and name in _conc['enabled'] and _conc['stack'][-1] == 'main'


# Lanes that REUSE another lane's stream (and with it its hardware queue: HIP has four by default and more cost dearly -- GPU_MAX_HW_QUEUES=8
# ran the step at 82 ms against 53): the weight-gradient queues of the encoders' backward passes run on the streams the teacher's two
# encoders used at the head of the iteration, which are joined and idle by then.
# (the decode head's weight gradients on such a queue from the start of the backward phase instead of in the image encoder's tail --
# lane 'hw' aliased to main/T/enc -- measured 51.53-51.62 against 51.22-51.30 ms, three alternating runs: the 256 x 256-tile kernels
# stall the encoders' chains, as in round 4; not aliased, not on)
LANE_ALIAS = {'main/wq': 'main/T', 'main/enc/wq': 'main/T/enc'}


class lane:
    """`with lane('enc', t1, t2...)`: run the body on the side stream `<current lane>/enc`, ordered after everything enqueued
    so far on the current lane (re-entering a lane therefore adds exactly that dependency).  Lanes nest.  The tensors
    named (inputs allocated on another stream that the body reads) are kept alive until the enclosing lane calls
    `join_lanes()` -- the caching allocator would otherwise hand their memory to the producer stream again while the side
    stream is still reading.
    independent=True: the body reads nothing the current lane produced since the side lane was last entered, so re-entering adds
    NO dependency -- the body queues up behind the side lane's earlier work only and overlaps whatever the current lane has
    enqueued in between (the first entry of an iteration still orders the lane after the current one)."""

    def __init__(self, name, *keep, independent=False):
        self.name, self.keep, self.on, self.independent = name, keep, False, independent

    def __enter__(self):
        if not _conc['on'] or self.name not in _conc['enabled']:
            return self
        parent = _conc['stack'][-1]
        if self.name == 'enc' and parent != 'main' and 'Tenc' not in _conc['enabled']:
            return self   # the teacher's encoders on lane T run one after the other ('Tenc': side by side on a fourth queue, main/T/enc)
        seg = _conc['seg']
        full = parent + '/' + self.name
        dev = seg.device if seg is not None else torch.cuda.current_device()
        key = (LANE_ALIAS.get(full, full), str(dev))
        s = _conc['streams'].get(key)
        if s is None:
            s = _conc['streams'][key] = torch.cuda.Stream(dev)
        _conc['used'][full] = s
        _conc['keep'].setdefault(full, []).extend(self.keep)
        _conc['stack'].append(full)
        ops.LN_LANE = full
        self.on = True
        order = not (self.independent and full in _conc['seen'])
        _conc['seen'].add(full)
        if seg is not None:
            seg.cut(waits=[(s, _conc['sstack'][-1])] if order else [], resume=s)
            _conc['sstack'].append(s)
        else:   # forked streams inside the surrounding (single) capture, or plain eager streams
            if order:
                s.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(s)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            seg = _conc['seg']
            _conc['stack'].pop()
            ops.LN_LANE = _conc['stack'][-1]
            if seg is not None:
                _conc['sstack'].pop()
                seg.cut(resume=_conc['sstack'][-1])
            else:
                self.ctx.__exit__(*exc)
        return False


class lane_batch:
    """`with lane_batch('wgrad'):` -- work handed to `side('wgrad', fn, ...)` inside the scope is collected and run at the end of
    the scope in ONE entry of the side lane (one fork per transformer block instead of one per weight gradient: the weight
    gradients leave the dependent dgrad chain without ~3 k cross-stream edges per step).  Without the lane: runs in place."""

    def __init__(self, name):
        self.name, self.active = name, False

    def __enter__(self):
        self.active = _conc['on'] and self.name in _conc['enabled'] and _conc.get('batch') is None
        if self.active:
            _conc['batch'] = (self.name, [], [])
        return self

    def __exit__(self, *exc):
        if self.active:
            name, fns, keep = _conc['batch']
            _conc['batch'] = None
            if fns and exc[0] is None:
                with lane(name, *keep):
                    for fn in fns:
                        fn()
        return False


def side(name, fn, *keep):
    """run fn() now, or -- inside an active lane_batch of that name -- later on the side lane (`keep`: the tensors fn reads)"""
    b = _conc.get('batch')
    if b is not None and b[0] == name:
        b[1].append(fn)
        b[2].extend(keep)
    else:
        fn()


def wait_external(fn):
    """order the current lane behind a stream that is not part of the iteration: fn() -> torch.cuda.Stream or None, evaluated NOW in
    eager mode and at every replay of a segmented capture (the overlapped optimizer / EMA update of uda.DACS: what the iteration has
    enqueued on its lanes before this point runs underneath that update)"""
    seg = _conc['seg']
    if seg is not None and seg.active is not None:
        stream = seg.active[1]
        seg.end()
        seg.program.append(('wait_ext', stream, fn))
        seg.begin(stream)
        return
    ext = fn()
    if ext is not None and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream().wait_stream(ext)


def keep_alive(*tensors):
    if _conc['on']:
        _conc['keep'].setdefault(_conc['stack'][-1], []).extend(tensors)


def join_lanes(name=None):
    """make the current lane wait for the lane `name` forked below it (and everything forked below that), or for every lane
    below it when no name is given; then release what those lanes kept alive"""
    if not _conc['on']:
        return
    me = _conc['stack'][-1]
    root = me + '/' + name if name else None

    def hit(k):
        return (k == root or k.startswith(root + '/')) if root else k.startswith(me + '/')
    joined = [_conc['used'].pop(full) for full in [k for k in _conc['used'] if hit(k)]]
    if joined:
        seg = _conc['seg']
        if seg is not None:
            cur = _conc['sstack'][-1]
            seg.cut(waits=[(cur, s) for s in joined], resume=cur)
        else:
            cur = torch.cuda.current_stream()
            for s in joined:
                cur.wait_stream(s)
    for k in [k for k in _conc['keep'] if hit(k)]:
        del _conc['keep'][k]


_anchors = {}


def anchor(device):
    """A grad-requiring scalar passed into the autograd bridges so that their backward always runs: parameter
    gradients are produced by side effect (see module docstring), so autograd cannot know the outputs depend on them."""
    a = _anchors.get(device)
    if a is None:
        a = torch.zeros(1, device=device, requires_grad=True)
        _anchors[device] = a
    return a


# Optional callback fired by the hand-scheduled backward passes when a group of parameter gradients is final FOR THIS PASS:
# grad_ready_hook(tag, module) with tag 'decode_head' or 'backbone.stage{1..4}' and the module whose gradients they are (the
# fusion student has two encoders).  Whoever sets it decides whether "final for this pass" means final for the step: bench.py's
# supervised step runs one backward pass; DACS arms it only around the LAST of its two passes (uda.DACS.final_pass_grad_hook)
# and only for the encoder that is back-propagated once per pass.
grad_ready_hook = None


def notify_grads_ready(tag, module=None):
    hook = grad_ready_hook
    if hook is None:
        return
    join_lanes('wgrad')      # "final" includes the weight gradients still queued on the side lane ...
    ops.ln_fold_deferred()   # "final" includes the LayerNorm parameter gradients still sitting in their workspaces
    seg = _conc['seg']
    if seg is not None and seg.active is not None:
        seg.call(lambda: hook(tag, module))   # segmented capture: the hook becomes a host step of the replay program
    else:
        hook(tag, module)
