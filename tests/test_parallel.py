"""Data-parallel layer on CPU: world_size-2 gloo process groups."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmda_amd.parallel import GradAllReducer, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, wire, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(rank)
    flat = torch.randn(100003)
    mine = flat.clone()
    red = GradAllReducer(flat, bucket_elems=30000, wire_dtype=wire)
    assert len(red.buckets) == 4
    red.all_reduce_mean()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ref = sum(gathered) / world
    tol = 1e-6 if wire == torch.float32 else 2e-2
    ok = torch.allclose(flat, ref, rtol=tol, atol=tol)
    # data-parallel identity: the gradient of the global-batch mean equals the mean of the per-rank gradients
    w = torch.ones(5, requires_grad=True)
    torch.manual_seed(123)
    data = torch.randn(world * 2, 5)
    lo, hi = shard_range(world * 2, rank, world)
    (data[lo:hi] * w).pow(2).mean().backward()
    g = w.grad.clone()
    GradAllReducer(g).all_reduce_mean()
    w2 = torch.ones(5, requires_grad=True)
    (data * w2).pow(2).mean().backward()
    ok = ok and torch.allclose(g, w2.grad, atol=1e-6)
    # staged exchange: two slices issued early (as the backward pass reports finished stages), the rest in finish()
    torch.manual_seed(100 + rank)
    flat2 = torch.randn(100003)
    mine2 = flat2.clone()
    red2 = GradAllReducer(flat2, bucket_elems=30000, wire_dtype=wire)
    red2.start_range(70000, 100003)
    red2.start_range(10, 45000)
    red2.finish()
    g2 = [torch.empty_like(mine2) for _ in range(world)]
    dist.all_gather(g2, mine2)
    ok = ok and torch.allclose(flat2, sum(g2) / world, rtol=tol, atol=tol)
    ok = ok and red2._started == []
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('wire', [torch.float32, torch.bfloat16])
def test_grad_allreduce_world2(wire):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), wire, out), nprocs=world, join=True)
        assert all(out[r] for r in range(world))


def _rs_ag_worker(rank, world, port, wire, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tol = 1e-6 if wire == torch.float32 else 2e-2
    ok = True
    # bucket sizes that do not divide the buffer, buckets that do not divide by the world size (padding to per * world), a last
    # bucket shorter than the world size
    for n, bucket in ((100003, 30001), (4099, 1025), (10, 7), (world * 512 + 1, world * 256)):
        torch.manual_seed(1000 * n + rank)
        flat = torch.randn(n)
        mine = flat.clone()
        red = GradAllReducer(flat, bucket_elems=bucket, wire_dtype=wire, exchange='rs_ag')
        assert red._rs_ag and not red._native_rs
        red.start_range(n // 3, n // 2 + 1)     # staged: a slice issued early, the rest in finish()
        red.finish()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        ref = sum(gathered) / world
        ok = ok and torch.allclose(flat, ref, rtol=tol, atol=tol)
        if wire == torch.float32:   # the same numbers as the all-reduce path, up to the order of the two-term sums
            flat_b = mine.clone()
            GradAllReducer(flat_b, bucket_elems=bucket, wire_dtype=wire, exchange='all_reduce').all_reduce_mean()
            ok = ok and torch.allclose(flat, flat_b, rtol=1e-6, atol=1e-7)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
@pytest.mark.parametrize('wire', [torch.float32, torch.bfloat16])
def test_grad_exchange_reduce_scatter_all_gather_path(world, wire):
    """GradAllReducer's reduce-scatter + all-gather arithmetic (the branch RCCL takes: padding of a bucket to per * world, mean on the
    owned shard, bf16 wire, all-gather, copy-back) with world > 1 on the CPU -- gloo has no reduce_scatter_tensor, so the collective
    is the stand-in of `_reduce_scatter`; everything around it is the production code (mmseg/core/ddp_wrapper.py:70-89's role)"""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rs_ag_worker, args=(world, _free_port(), wire, out), nprocs=world, join=True)
        assert all(out[r] for r in range(world))


def test_shard_range():
    assert [shard_range(16, r, 8) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]


def test_flat_buffer_is_ordered_by_backward_stage():
    """FlatAdamW lays parameters out so that a backbone stage's weights are one contiguous slice (what the overlapped
    all-reduce starts on) and the slices of different stages are disjoint."""
    import cmda_amd  # noqa: F401
    from cmda_amd import optim
    from cmda_amd.registry import build_backbone
    import torch.nn as nn

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = build_backbone(dict(type='mit_b0', style='pytorch'))
            self.decode_head = nn.Linear(8, 8)
    m = M()
    opt = optim.FlatAdamW(m, custom_keys=dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0)))
    spans = {}
    for s in range(1, 5):
        r = opt.ranges_of(m, [f'backbone.patch_embed{s}.', f'backbone.block{s}.', f'backbone.norm{s}.'], min_elems=0)
        assert 1 <= len(r) <= 2, r  # the weights slice (+ the no-decay norm/bias slice of the other parameter group)
        spans[s] = r
        want = sum((p.numel() + 7) // 8 * 8 for n, p in m.named_parameters()
                   if n.startswith((f'backbone.patch_embed{s}.', f'backbone.block{s}.', f'backbone.norm{s}.')))
        assert sum(hi - lo for lo, hi in r) == want
    flat = sorted(x for r in spans.values() for x in r)
    assert all(a[1] <= b[0] for a, b in zip(flat, flat[1:]))
    # stage 4 first: its weights slice starts before stage 1's in the (lr 1, decay 1) group
    big = {s: max(r, key=lambda t: t[1] - t[0]) for s, r in spans.items()}
    assert big[4][0] < big[3][0] < big[2][0] < big[1][0]


@pytest.mark.gpu
def test_backward_reports_stages_in_order_and_staged_grads_match():
    """the hand-scheduled backward fires runtime.grad_ready_hook once per group, head first then stage 4..1, and at the
    moment a stage is reported its slice of the flat gradient buffer is final (equal to the value after the pass)"""
    import cmda_amd.runtime as rt
    from cmda_amd import optim, segmentors, backbones, decode_heads  # noqa: F401
    from cmda_amd.registry import build_segmentor
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    cfg = dict(type='EncoderDecoder',
               backbone=dict(type='MixVisionTransformer', embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8],
                             qkv_bias=True, depths=[1, 1, 1, 1], sr_ratios=[8, 4, 2, 1], drop_path_rate=0.0),
               decode_head=dict(type='DAFormerHead', in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256,
                                dropout_ratio=0.0, num_classes=19, norm_cfg=dict(type='BN'), align_corners=False,
                                decoder_params=dict(embed_dims=256, embed_cfg=dict(type='mlp'), embed_neck_cfg=dict(type='mlp'),
                                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False))))
    model = build_segmentor(cfg).to(dev).train()
    rt.set_compute_dtype(torch.bfloat16)
    try:
        opt = optim.FlatAdamW(model, custom_keys=dict(head=dict(lr_mult=10.0), norm=dict(decay_mult=0.0)))
        ranges = {f'backbone.stage{s}': opt.ranges_of(model, [f'backbone.patch_embed{s}.', f'backbone.block{s}.', f'backbone.norm{s}.'],
                                                     min_elems=0) for s in range(1, 5)}
        ranges['decode_head'] = opt.ranges_of(model, ['decode_head.'], min_elems=0)
        seen, snaps = [], {}

        def hook(tag, module=None):
            seen.append(tag)
            snaps[tag] = [opt.flat_g[lo:hi].clone() for lo, hi in ranges[tag]]
        rt.grad_ready_hook = hook
        img = torch.randn(2, 3, 64, 64, device=dev)
        gt = torch.randint(0, 19, (2, 1, 64, 64), device=dev)
        opt.zero_grad()
        losses, _ = model.forward_train(img, None, gt)
        losses['decode.loss_seg'].backward()
        torch.cuda.synchronize()
        assert seen == ['decode_head', 'backbone.stage4', 'backbone.stage3', 'backbone.stage2', 'backbone.stage1']
        for tag, parts in snaps.items():
            for (lo, hi), snap in zip(ranges[tag], parts):
                assert torch.equal(snap, opt.flat_g[lo:hi]), f'{tag}: gradients changed after the stage was reported'
            assert sum(p.abs().sum().item() for p in parts) > 0
    finally:
        rt.grad_ready_hook = None
        rt.set_compute_dtype(torch.float32)


@pytest.mark.gpu
def test_dacs_final_pass_hook_reports_final_gradients():
    """DACS arms runtime.grad_ready_hook only around its LAST backward pass; what it reports for the decode head and both
    encoders (each back-propagated once per pass: the event encoder sees events + ISR as one batch) equals the gradient at
    the end of the iteration."""
    import random
    import numpy as np
    import cmda_amd.runtime as rt
    from cmda_amd import optim
    from cmda_amd.registry import build_train_model
    from test_dacs import SMALL, make_cfg
    from weights import seeded_fill, seeded_randn
    dev = torch.device('cuda:0')
    rt.set_compute_dtype(torch.float32)
    B, H, W = 2, 64, 64
    dacs = build_train_model(make_cfg(SMALL['dims'], SMALL['ch'], generator=False))
    seeded_fill(dacs.model, 7)
    seeded_fill(dacs.ema_model, 8)
    dacs.to(dev).train()
    opt = optim.FlatAdamW(dacs.model, custom_keys=dict(head=dict(lr_mult=10.0), norm=dict(decay_mult=0.0)))
    dacs.attach_flat_store(opt)
    student = dacs.model
    ranges = {('decode_head', id(student.decode_head)): opt.ranges_of(student, ['decode_head.'], min_elems=0)}
    for name in ('backbone_image', 'backbone_events'):
        for s in range(1, 5):
            ranges[(f'backbone.stage{s}', id(getattr(student, name)))] = opt.ranges_of(
                student, [f'{name}.patch_embed{s}.', f'{name}.block{s}.', f'{name}.norm{s}.'], min_elems=0)
    seen, snaps = [], {}

    def hook(tag, module=None):
        key = (tag, id(module))
        seen.append((tag, module is student.backbone_image, module is student.backbone_events))
        if key in ranges:
            snaps[key] = [opt.flat_g[lo:hi].clone() for lo, hi in ranges[key]]
    dacs.final_pass_grad_hook = hook
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(0, 6, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    src = dict(image=seeded_randn((B, 3, H, W), 7, 'img'), img_time_res=seeded_randn((B, 3, H, W), 7, 'itr'),
               img_self_res=seeded_randn((B, 3, H, W), 7, 'isr').clamp(-1, 1), label=lab)
    tg = dict(warp_image=seeded_randn((B, 3, H, W), 7, 'nimg'), events_vg=seeded_randn((B, 3, H, W), 7, 'nev').clamp(-1, 1),
              warp_img_self_res=seeded_randn((B, 3, H, W), 7, 'nisr').clamp(-1, 1))
    batch = dict(source={k: v.to(dev) for k, v in src.items()}, target={k: v.to(dev) for k, v in tg.items()})
    torch.manual_seed(11), random.seed(11), np.random.seed(11)
    opt.zero_grad()
    dacs(**batch)
    torch.cuda.synchronize()
    assert rt.grad_ready_hook is None                       # disarmed again after the pass
    assert seen[0][0] == 'decode_head' and len(snaps) == 9  # head + four stages of each encoder, reported once each
    assert [t for t, img, _ in seen if img] == [f'backbone.stage{s}' for s in (4, 3, 2, 1)]
    assert [t for t, _, evt in seen if evt] == [f'backbone.stage{s}' for s in (4, 3, 2, 1)]
    for key, parts in snaps.items():
        for (lo, hi), snap in zip(ranges[key], parts):
            assert torch.equal(snap, opt.flat_g[lo:hi]), f'{key[0]}: changed after being reported'


@pytest.mark.gpu
def test_dacs_final_pass_hook_inside_segmented_graph():
    """the same with the iteration captured as linear hipGraph segments (runtime.SegmentedCapture): the hook is recorded as a
    HOST step of the replay program, runs with the reporting lane's stream current after that lane's earlier segments were
    enqueued, and what it snapshots there (stream-ordered) equals the gradient at the end of the iteration -- eager
    iteration 0, capturing iteration 1, replayed iteration 2."""
    import random
    import numpy as np
    import cmda_amd.runtime as rt
    from cmda_amd import optim
    from cmda_amd.registry import build_train_model
    from test_dacs import SMALL, make_cfg
    from weights import seeded_fill, seeded_randn
    dev = torch.device('cuda:0')
    rt.set_compute_dtype(torch.float32)
    B, H, W = 2, 64, 64
    dacs = build_train_model(make_cfg(SMALL['dims'], SMALL['ch'], generator=False))
    seeded_fill(dacs.model, 7)
    seeded_fill(dacs.ema_model, 8)
    dacs.to(dev).train()
    opt = optim.FlatAdamW(dacs.model, custom_keys=dict(head=dict(lr_mult=10.0), norm=dict(decay_mult=0.0)))
    dacs.attach_flat_store(opt)
    student = dacs.model
    ranges = {('decode_head', id(student.decode_head)): opt.ranges_of(student, ['decode_head.'], min_elems=0)}
    for name in ('backbone_image', 'backbone_events'):
        for s in range(1, 5):
            ranges[(f'backbone.stage{s}', id(getattr(student, name)))] = opt.ranges_of(
                student, [f'{name}.patch_embed{s}.', f'{name}.block{s}.', f'{name}.norm{s}.'], min_elems=0)
    seen, snaps = [], {}

    def hook(tag, module=None):
        key = (tag, id(module))
        seen.append((tag, module is student.backbone_image, module is student.backbone_events))
        if key in ranges:
            snaps[key] = [opt.flat_g[lo:hi].clone() for lo, hi in ranges[key]]
    dacs.final_pass_grad_hook = hook
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(0, 6, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    src = dict(image=seeded_randn((B, 3, H, W), 7, 'img'), img_time_res=seeded_randn((B, 3, H, W), 7, 'itr'),
               img_self_res=seeded_randn((B, 3, H, W), 7, 'isr').clamp(-1, 1), label=lab)
    tg = dict(warp_image=seeded_randn((B, 3, H, W), 7, 'nimg'), events_vg=seeded_randn((B, 3, H, W), 7, 'nev').clamp(-1, 1),
              warp_img_self_res=seeded_randn((B, 3, H, W), 7, 'nisr').clamp(-1, 1))
    batch = dict(source={k: v.to(dev) for k, v in src.items()}, target={k: v.to(dev) for k, v in tg.items()})
    torch.manual_seed(11), random.seed(11), np.random.seed(11)
    dacs.enable_graph(warmup_iters=1)
    for it in range(3):
        seen.clear(), snaps.clear()
        opt.zero_grad()
        dacs(**batch)
        torch.cuda.synchronize()
        assert (dacs._graph is not None) == (it >= 1)
        assert rt.grad_ready_hook is None
        assert seen[0][0] == 'decode_head' and len(snaps) == 9, (it, seen)
        assert [t for t, img, _ in seen if img] == [f'backbone.stage{s}' for s in (4, 3, 2, 1)]
        assert [t for t, _, evt in seen if evt] == [f'backbone.stage{s}' for s in (4, 3, 2, 1)]
        assert opt.flat_g.abs().sum().item() > 0
        for key, parts in snaps.items():
            for (lo, hi), snap in zip(ranges[key], parts):
                assert torch.equal(snap, opt.flat_g[lo:hi]), f'iteration {it}, {key[0]}: changed after being reported'


def _dacs_worker(rank, world, port, out, exchange='auto'):
    """one data-parallel rank of the DACS step on the CPU emulator: own batch, rank-local BatchNorm / ClassMix / pseudo-weight,
    gradients exchanged through GradAllReducer with the final-pass staging (decode head + both encoders start their slices from
    DACS.final_pass_grad_hook inside the LAST backward pass, the rest in finish())"""
    import random
    import sys
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.join(here, 'golden'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(3)
    from cmda_amd import _lib, optim
    from conftest import EMU_LIB
    _lib._bind_for_tests(EMU_LIB)
    import cmda_amd.runtime as rt
    from cmda_amd.registry import build_train_model
    from oracle import dacs_iter
    import test_dacs as T
    from weights import seeded_fill, seeded_randn
    rt.set_compute_dtype(torch.float32)
    dims, ch, B, H, W = T.SMALL['dims'], T.SMALL['ch'], 1, 64, 64
    dacs = build_train_model(T.make_cfg(dims, ch, generator=False))
    seeded_fill(dacs.model, 7)        # identical replicas on every rank
    seeded_fill(dacs.ema_model, 8)
    dacs.train()
    opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01)
    dacs.attach_flat_store(opt)
    g = torch.Generator().manual_seed(40 + rank)   # rank-local shard of the global batch
    lab = torch.randint(0, 6, (B, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    src = dict(image=seeded_randn((B, 3, H, W), 40 + rank, 'img'), img_time_res=seeded_randn((B, 3, H, W), 40 + rank, 'itr').clamp(-1, 1),
               img_self_res=seeded_randn((B, 3, H, W), 40 + rank, 'isr').clamp(-1, 1), label=lab)
    tg = dict(warp_image=seeded_randn((B, 3, H, W), 40 + rank, 'nimg'), events_vg=seeded_randn((B, 3, H, W), 40 + rank, 'nev').clamp(-1, 1),
              warp_img_self_res=seeded_randn((B, 3, H, W), 40 + rank, 'nisr').clamp(-1, 1))
    torch.manual_seed(11 + rank), random.seed(11 + rank), np.random.seed(11 + rank)   # rank-local draws
    # the single-rank oracle step on this rank's shard (its own teacher, its own BatchNorm statistics)
    ref, ema = T.oracle_student(dims, ch), T.oracle_student(dims, ch)
    seeded_fill(ref, 7).train()
    seeded_fill(ema, 8).train()
    # --- the data-parallel step
    # exchange='rs_ag': the reduce-scatter + all-gather branch RCCL takes; a bucket size that does not divide the slices (padding)
    reducer = GradAllReducer(opt.flat_g, bucket_elems=(1 << 20) if exchange == 'auto' else 300007, exchange=exchange)
    assert reducer._rs_ag == (exchange == 'rs_ag')
    student = dacs.model
    ranges = {('decode_head', id(student.decode_head)): opt.ranges_of(student, ['decode_head.'], min_elems=0)}
    for name in ('backbone_image', 'backbone_events'):
        for s in range(1, 5):
            ranges[(f'backbone.stage{s}', id(getattr(student, name)))] = opt.ranges_of(
                student, [f'{name}.patch_embed{s}.', f'{name}.block{s}.', f'{name}.norm{s}.'], min_elems=0)
    started = []

    def hook(tag, module=None):
        for lo, hi in ranges.get((tag, id(module)), ()):
            reducer.start_range(lo, hi)
            started.append((lo, hi))
    dacs.final_pass_grad_hook = hook
    opt.zero_grad()
    lv = dacs(source=src, target=tg)
    reducer.finish()
    # K19 (base.py:736-741): the step's log scalars, mean-reduced over the ranks in one small exchange
    from cmda_amd.parallel import reduce_log_vars
    mine_lv = {k: float(v.reshape(-1)[0]) for k, v in lv.items()}
    red_lv = {k: float(v.reshape(-1)[0]) for k, v in reduce_log_vars(lv).items()}
    all_lv = [None] * world
    dist.all_gather_object(all_lv, mine_lv)
    lv_err = max(abs(red_lv[k] - sum(d[k] for d in all_lv) / world) for k in mine_lv)
    lv_spread = max(abs(all_lv[0][k] - all_lv[1][k]) for k in mine_lv)
    o = dacs_iter.dacs_iteration(ref, ema, None, src, tg, local_iter=0, forward_cfg=T.FCFG, isr_parms=T.ISR, shift_type='random',
                                 draws=T.oracle_draws(dacs.last_draws))
    del o
    # expected: the mean over ranks of the independent single-rank oracle gradients
    worst, n_checked = 0.0, 0
    for (n1, p), (n2, q) in zip(dacs.model.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        gq = q.grad.clone()
        dist.all_reduce(gq)
        gq /= world
        worst = max(worst, ((p.grad - gq).abs().max() / (gq.abs().max() + 1e-12)).item())
        n_checked += 1
    out[rank] = dict(worst=worst, n=n_checked, staged=len(started), staged_elems=sum(hi - lo for lo, hi in started),
                     total=opt.flat_g.numel(), lv_err=lv_err, lv_spread=lv_spread, lv_keys=sorted(mine_lv))
    dist.destroy_process_group()


@pytest.mark.parametrize('exchange', ['rs_ag'])
def test_dacs_data_parallel_world2_matches_mean_of_oracle_steps(exchange):
    """global batch 2 over 2 ranks == 2 independent reference-style steps whose gradients are averaged (BatchNorm
    statistics, ClassMix class draws and the pseudo-weight stay rank-local).  The gradients are exchanged through the reduce-scatter +
    all-gather branch (parallel.py `_reduce`), the one the RCCL backend takes; gloo's plain all-reduce branch ('auto' here, one minute
    of emulator time more) is covered by the bucketed / staged exchange tests above."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_dacs_worker, args=(world, _free_port(), out, exchange), nprocs=world, join=True)
        for r in range(world):
            assert out[r]['worst'] < 5e-2, out[r]
            assert out[r]['staged'] >= 9 and out[r]['staged_elems'] > 0.5 * out[r]['total'], out[r]   # most bytes start inside the last pass
            assert out[r]['n'] > 100
            assert out[r]['lv_err'] < 1e-5 and out[r]['lv_spread'] > 1e-4, out[r]   # reduced == mean of different rank-local values
            assert 'decode.loss_seg' in out[r]['lv_keys'] and 'mix.decode.loss_seg' in out[r]['lv_keys']


def _logvar_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from collections import OrderedDict
    from cmda_amd.parallel import reduce_log_vars
    lv = OrderedDict([('decode.loss_seg', torch.tensor(1.0 + rank)), ('decode.acc_seg', torch.tensor([10.0 * (rank + 1)])),
                      ('mix.decode.loss_seg', torch.tensor(0.25 * (rank + 1))), ('loss', torch.tensor(3.0 - rank))])
    red = reduce_log_vars(lv)
    out[rank] = {k: (v.tolist() if v.dim() else v.item()) for k, v in red.items()}
    out[f'keys{rank}'] = list(red.keys())
    dist.destroy_process_group()


def test_log_vars_are_mean_reduced_world2():
    """K19 / mmseg/models/segmentors/base.py:736-741: under data parallelism every logged scalar is its mean over the ranks;
    shapes ([] and [1]) and key order survive; without a process group the dict comes back untouched."""
    from cmda_amd.parallel import reduce_log_vars
    lv = {'a': torch.tensor(1.0)}
    assert reduce_log_vars(lv) is lv
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_logvar_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        for r in range(world):
            assert out[f'keys{r}'] == ['decode.loss_seg', 'decode.acc_seg', 'mix.decode.loss_seg', 'loss']
            assert out[r]['decode.loss_seg'] == pytest.approx(1.5)
            assert out[r]['decode.acc_seg'] == pytest.approx([15.0])
            assert out[r]['mix.decode.loss_seg'] == pytest.approx(0.375)
            assert out[r]['loss'] == pytest.approx(2.5)


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` with no launcher forks one fresh rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torchrun sets them), rank 0 prints the line, the return code is the ranks' -- checked through --dry-run (gloo, no GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '2'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['ranks'] == 2 and out['config']['global_batch'] == '4 + 4'
    # a mismatching launcher environment is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=dict(env, WORLD_SIZE='1', RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def _bn_worker(rank, world, port, out):
    from cmda_amd.parallel import broadcast_bn_buffers
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1),
                              torch.nn.BatchNorm2d(4), torch.nn.LayerNorm([4, 5, 5])).train()
    torch.manual_seed(100 + rank)                      # rank-local batches: the running statistics drift apart, as in training
    for _ in range(3):
        net(torch.randn(2, 3, 5, 5) * (1 + rank))
    before = [b.clone() for n, b in net.named_buffers() if 'running' in n]
    params = [p.detach().clone() for p in net.parameters()]
    n = broadcast_bn_buffers(net)
    after = torch.cat([b.reshape(-1) for nme, b in net.named_buffers() if 'running' in nme])
    gathered = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    b0 = torch.cat([b.reshape(-1) for b in before])
    g0 = [torch.empty_like(b0) for _ in range(world)]
    dist.all_gather(g0, b0)
    out[rank] = dict(n=n, same=all(torch.equal(gathered[0], g) for g in gathered), src=torch.equal(gathered[0], g0[0]),
                     differed=not torch.equal(g0[0], g0[1]),
                     params=all(torch.equal(p, q) for p, q in zip(params, net.parameters())),
                     nbt=[int(b) for nme, b in net.named_buffers() if 'num_batches' in nme])
    dist.destroy_process_group()


def test_bn_buffers_broadcast_before_distributed_eval_world2():
    """eval_hooks.py:92-100: rank 0's BatchNorm running statistics reach every rank before a distributed evaluation; parameters
    and the batch counters are untouched"""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bn_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        for r in range(world):
            assert out[r]['n'] == 2 and out[r]['differed'] and out[r]['same'] and out[r]['src'] and out[r]['params'], out[r]
            assert out[r]['nbt'] == [3, 3]


class _TinySeg(torch.nn.Module):
    """stand-in segmentor with the evaluation interface of the path (`simple_test(rescale, img=...)` -> list of label maps)"""

    def __init__(self, classes=5):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                       torch.nn.Conv2d(8, classes, 1))

    def simple_test(self, rescale=True, img=None):
        return list(self.net(img).argmax(1).numpy())


def _eval_samples(n, classes=5):
    g = torch.Generator().manual_seed(31)
    out = []
    for i in range(n):
        h, w = 12 + i, 16            # ragged heights: nothing of the exchange may depend on the image size
        gt = torch.randint(0, classes, (h, w), generator=g)
        gt[torch.rand(h, w, generator=g) < 0.1] = 255
        out.append(dict(img=torch.randn(1, 3, h, w, generator=g), gt_semantic_seg=gt))
    return out


def _eval_worker(rank, world, port, out):
    from cmda_amd.parallel import distributed_evaluate
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(11)
    model = _TinySeg().train()
    torch.manual_seed(200 + rank)                      # rank-local batches: running statistics differ at evaluation time
    for _ in range(4):
        model.net(torch.randn(2, 3, 8, 8) * (1 + 2 * rank) + rank)
    res = distributed_evaluate(model, _eval_samples(5), 5)
    out[rank] = {k: v.clone() for k, v in res.items()}
    out[f'training{rank}'] = model.training
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_distributed_evaluate_matches_single_process_with_rank0_statistics(world):
    """eval_hooks.py:86-121 + apis/test.py:216-274: the ranks score disjoint shares of the validation set with rank 0's BatchNorm
    statistics and meet in one histogram all-reduce; result = one process scoring all samples with rank 0's model (the reference's
    `multi_gpu_test` + `dataset.evaluate`), on every rank; 5 samples over 2 / 3 ranks = uneven shares"""
    from cmda_amd import metrics
    from cmda_amd.parallel import distributed_evaluate
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_eval_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        got = {r: out[r] for r in range(world)}
        assert all(out[f'training{r}'] for r in range(world))        # the caller's train / eval state is restored
    torch.manual_seed(11)
    model = _TinySeg().train()
    torch.manual_seed(200)
    for _ in range(4):
        model.net(torch.randn(2, 3, 8, 8))
    samples = _eval_samples(5)
    model.eval()
    with torch.no_grad():
        preds = [model.simple_test(True, img=s['img'])[0] for s in samples]
    ref = metrics.mean_iou([torch.as_tensor(p) for p in preds], [s['gt_semantic_seg'] for s in samples], 5)
    single = distributed_evaluate(model, samples, 5)                 # no process group: the same function, one rank
    for r in range(world):
        for k in ('aAcc', 'mIoU', 'mAcc', 'IoU', 'Acc'):
            assert torch.equal(torch.nan_to_num(got[r][k], nan=-1.0), torch.nan_to_num(ref[k], nan=-1.0)), (r, k)
    for k in ('aAcc', 'mIoU', 'IoU'):
        assert torch.equal(torch.nan_to_num(single[k], nan=-1.0), torch.nan_to_num(ref[k], nan=-1.0))


def _one_rank_group(backend):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1')
    dist.init_process_group(backend, rank=0, world_size=1)


def _virtual_ways_case(dev, wire):
    """one rank, `force`, exchange laid out for 3 virtual ranks: the mean over one rank is the identity (up to the wire rounding)"""
    torch.manual_seed(5)
    for n, bucket in ((100003, 30001), (4099, 1025), (10, 7)):
        flat = torch.randn(n, device=dev)
        want = flat.clone()
        red = GradAllReducer(flat, bucket_elems=bucket, wire_dtype=wire, force=True, exchange='rs_ag', virtual_ways=3)
        assert red.active and red._rs_ag and red.ways == 3
        red.start_range(n // 3, n // 2)
        red.all_reduce_mean()
        if dev != 'cpu':
            torch.cuda.synchronize()
        tol = 0.0 if wire == torch.float32 else 8e-3
        assert torch.allclose(flat, want, rtol=tol, atol=tol), (n, bucket)


@pytest.mark.parametrize('wire', [torch.float32, torch.bfloat16])
def test_grad_exchange_virtual_ways_single_rank_gloo(wire):
    """`bench.py --force-reducer` on one GPU: GradAllReducer.virtual_ways pads / scatters / gathers every bucket as for several
    ranks through the collectives of the world-1 group (mmseg/core/ddp_wrapper.py:70-89's exchange, one rank standing in)"""
    _one_rank_group('gloo')
    try:
        _virtual_ways_case('cpu', wire)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('wire', [torch.float32, torch.bfloat16])
def test_grad_exchange_native_reduce_scatter_all_gather_single_rank_rccl(wire):
    """the RCCL calls themselves -- dist.reduce_scatter_tensor / dist.all_gather_into_tensor on shard-sized views of the wire buffer
    (parallel.py `_reduce_scatter` / `_all_gather`, native branch) -- on one GPU, with the padded layout of a 3-rank exchange"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    _one_rank_group('nccl')
    try:
        red = GradAllReducer(torch.zeros(8, device='cuda'), force=True)
        assert red._native_rs
        _virtual_ways_case('cuda', wire)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_distributed_evaluate_single_rank_rccl_host_label_maps():
    """the default predictor returns HOST label maps, RCCL has no CPU backend -- the histogram vector must be moved to the
    model's device before the all-reduce (and a rank without samples must build it there too)"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    from cmda_amd import metrics
    from cmda_amd.parallel import distributed_evaluate

    class Seg(_TinySeg):
        def simple_test(self, rescale=True, img=None):
            return list(self.net(img.cuda()).argmax(1).cpu().numpy())
    _one_rank_group('nccl')
    try:
        torch.manual_seed(11)
        model = Seg().cuda().eval()
        samples = _eval_samples(4)
        got = distributed_evaluate(model, samples, 5, force=True)
        with torch.no_grad():
            preds = [model.simple_test(True, img=s['img'])[0] for s in samples]
        ref = metrics.mean_iou([torch.as_tensor(p) for p in preds], [s['gt_semantic_seg'] for s in samples], 5)
        for k in ('aAcc', 'mIoU', 'IoU'):   # (float64 on the device: the mean's summation order differs from the host's in the last bit)
            assert torch.allclose(torch.nan_to_num(got[k].cpu(), nan=-1.0), torch.nan_to_num(ref[k], nan=-1.0), rtol=1e-12, atol=0), k
        empty = distributed_evaluate(model, [], 5, force=True)      # no samples on this rank: the zero vector goes to the same device
        assert empty['IoU'].shape == (5,)
    finally:
        dist.destroy_process_group()
