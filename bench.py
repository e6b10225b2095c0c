#!/usr/bin/env python3
"""Headline benchmark: training images/s of the CMDA hot path on MI355X (BASELINE.json metric).

Default workload = BASELINE.json configs[3], the configuration the metric is quoted on ("512x512 image+event"): one
FULL CMDA UDA iteration per step -- Motion-Extractor generator on the source time residual, EMA-teacher update, source
forward/backward of the image+events fusion student (two MiT-B5 encoders, AttentionAvgFusion, shared DAFormer head, four
CE terms), teacher forward + pseudo-labels, ClassMix + colour jitter + blur + ISR of the mixed image, mixed
forward/backward, fused AdamW -- at the reference recipe's 2 source + 2 target samples per GPU, bf16 activations / fp32
accumulate, random-init weights of the real architecture, synthetic inputs shaped as SURVEY.md section 8d prescribes
(events through the voxel-grid kernel, ISR through the Content-Extractor kernels, sparse time residual).
`value` counts (source, target) PAIRS per second (one "image" = one 512x512 image + its event / ISR
companions).  N>1: one process per GPU (torchrun), weak scaling (2+2 per GPU), gradients averaged with RCCL.
`--workload supervised` = configs[1] (MiT-B5 + DAFormer head fwd/bwd), kept as a secondary line.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel family = the MFMA GEMM / implicit-GEMM, HIP-event timed
on the launch stream) and, at N=1, `cpu_baseline` (the oracle restatement timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_TBS = 8.0              # HBM3E, same guide (6.3 TB/s achievable by a float4 copy)
GFLOP_PER_IMAGE_FWD = 252.6     # MiT-B5 138.8 + DAFormer head 113.8
GFLOP_PER_PAIR_UDA = 6430.0     # full UDA step per (source, target) pair

HEAD_CFG = dict(in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256, num_classes=19,
                norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
DECODER = dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
               embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
               fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False, act_cfg=dict(type='ReLU'),
                               norm_cfg=dict(type='BN', requires_grad=True)))
ISR_PARMS = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
FORWARD_CFG = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)
CUSTOM_KEYS = dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0))


def host_cores():
    """(logical CPUs visible to this process, physical cores of the host) -- the latter from /proc/cpuinfo."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    phys = set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pid = line.split(':')[1].strip()
            elif line.startswith('core id'):
                cid = line.split(':')[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return logical, (len(phys) or logical)


# ------------------------------------------------------------------------------------------------ configs[1] (secondary)
def build_model(dev, drop_path_rate=0.1, dropout_ratio=0.1):
    import cmda_amd  # noqa: F401
    from cmda_amd.registry import build_segmentor
    cfg = dict(type='EncoderDecoder', backbone=dict(type='mit_b5', style='pytorch', drop_path_rate=drop_path_rate),
               decode_head=dict(type='DAFormerHead', dropout_ratio=dropout_ratio, decoder_params=dict(DECODER), **HEAD_CFG),
               train_cfg=dict(), test_cfg=dict(mode='whole'))
    model = build_segmentor(cfg)
    model.init_weights()
    return model.to(dev).train()


def synthetic_labels(B, size, g):
    lab = torch.randint(0, 19, (B, 1, size // 32, size // 32), generator=g)
    lab = lab.repeat_interleave(32, 2).repeat_interleave(32, 3)
    lab[torch.rand(B, 1, size, size, generator=g) < 0.05] = 255
    return lab


def synthetic_batch(B, size, seed, dev):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, size, size, generator=g)
    return img.to(dev), synthetic_labels(B, size, g).to(dev)


def cpu_baseline_supervised(size):
    """Oracle (port of the reference algorithm, oracle/) timed on the host cores: one fwd+bwd of MiT-B5 + DAFormer
    head on ONE image (bounded sample of the same workload)."""
    from oracle import head as ohd, mit as omit, segmentor as oseg
    logical, physical = host_cores()
    # 16 threads: torch's CPU kernels at these sizes stop scaling there (256 threads took 737 s for this sample on the
    # GPU box's host, 8 threads take 11 s) -- `cores` reports the threads actually used
    cores = min(logical, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    ref = oseg.EncoderDecoder(omit.mit_b5(drop_path_rate=0.1), ohd.DAFormerHead(dropout_ratio=0.1)).train()
    ref.backbone.init_weights()
    img = torch.randn(1, 3, size, size)
    gt = torch.randint(0, 19, (1, 1, size, size))

    def one():
        for p in ref.parameters():
            p.grad = None
        losses, _ = ref.forward_train(img, gt)
        losses['decode.loss_seg'].backward()

    t0 = time.time()
    one()                                   # warm-up / first-call allocations, also the size estimate
    t1 = time.time() - t0
    n = max(1, min(12, int(round(12.0 / max(t1, 1e-3)))))   # ~10-20 s of CPU work in total
    t0 = time.time()
    for _ in range(n):
        one()
    dt = (time.time() - t0) / n
    return {'value': round(1.0 / dt, 4), 'unit': 'img/s', 'cores': cores, 'host_physical_cores': physical,
            'host_logical_cpus': logical, 'kind': 'port',
            'sample': f'{n} x (1 image {size}x{size}, fwd+bwd of MiT-B5+DAFormerHead, fp32 oracle) on {cores} host '
                      f'threads after 1 warm-up, {dt:.2f} s per image'}


# ------------------------------------------------------------------------------------------------ configs[3] (headline)
def dacs_cfg():
    """configs/fusion/cs2dsec_image+events_together_b5.py + the launcher defaults of SURVEY.md appendix A, restated inline
    (the reference tree is not on the GPU box).  cyclegan_itrd2en_path='random': the Motion-Extractor generator with seeded
    random weights (no checkpoint exists offline) -- the generator RUNS in every step, as in dacs.py:400-404."""
    dims = [64, 128, 320, 512]
    bb = dict(type='mit_b5', style='pytorch', drop_path_rate=0.1)
    head = dict(type='DAFormerHeadFusion', dropout_ratio=0.1,
                decoder_params=dict(DECODER, train_type='cs2dsec_image+events_together', share_decoder=True), **HEAD_CFG)
    model = dict(type='FusionEncoderDecoder', backbone_image=dict(bb), backbone_events=dict(bb),
                 fusion_module=dict(type='AttentionAvgFusion', in_channels=dims, drop_path_rate=0.1), decode_head=head,
                 train_type='cs2dsec_image+events_together', train_cfg=dict(), test_cfg=dict(mode='whole'))
    uda = dict(type='DACS', alpha=0.999, pseudo_threshold=0.968, pseudo_weight_ignore_top=0, pseudo_weight_ignore_bottom=0,
               imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None, imnet_feature_dist_scale_min_ratio=None,
               mix='class', blur=True, color_jitter_strength=0.2, color_jitter_probability=0.2, debug_img_interval=1000,
               print_grad_magnitude=False, train_type='cs2dsec_image+events_together', forward_cfg=dict(FORWARD_CFG),
               cyclegan_itrd2en_path='random', img_self_res_reg='no', mixed_image_to_mixed_isr=True,
               random_choice_thres='0.5', shift_type='random', isr_parms=dict(ISR_PARMS), sky_mask=None)
    return dict(model=model, uda=uda, runner=dict(type='IterBasedRunner', max_iters=40000))


def build_dacs(dev):
    import cmda_amd  # noqa: F401
    from cmda_amd.registry import build_train_model
    dacs = build_train_model(dacs_cfg())
    dacs.init_weights()
    return dacs.to(dev).train()


def synthetic_pairs(B, S, seed, dev, n_events=500000):
    """SURVEY.md section 8d synthetic stream, one batch, resident in HBM.  Source: image N(0,1), block-structured labels with
    5 % ignore, sparse time residual U(-1,1)*Bernoulli(0.1) (3 identical channels), ISR from the Content-Extractor kernels.
    Target: image N(0,1), its ISR from the same kernels, events: `n_events` synthetic events per sample (fractional rectified
    x in [0,640), y in [0,480), sorted t, polarity 0/1) -> voxel-grid kernel (1 bin) -> events_norm -> crop 400x400 -> bilinear
    512x512 -> 3 channels (dsec.py:189-366)."""
    from cmda_amd import ops
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, S, S, generator=g).to(dev)
    label_host = synthetic_labels(B, S, g)
    label = label_host.to(dev)
    label._cmda_classes = torch.unique(label_host)   # the class set ClassMix draws from, taken on the host like the loader does
    itr = (torch.rand(B, 1, S, S, generator=g) * 2 - 1) * (torch.rand(B, 1, S, S, generator=g) < 0.1)
    itr = itr.repeat(1, 3, 1, 1).to(dev)
    warp = torch.randn(B, 3, S, S, generator=g).to(dev)

    def isr(img):
        return ops.isr_from_gray(ops.isr_gray(img), ISR_PARMS['val_range'], ISR_PARMS['_threshold'], ISR_PARMS['_clip_range'],
                                 ISR_PARMS['shift_pixel'], 'rightdown')
    ev = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
    crop = 400
    for b in range(B):
        t = torch.rand(n_events, generator=g).sort().values
        x = torch.rand(n_events, generator=g) * 640
        y = torch.rand(n_events, generator=g) * 480
        p = (torch.rand(n_events, generator=g) < 0.5).float()
        grid = ops.events_to_voxel_grid(t.to(dev), x.to(dev), y.to(dev), p.to(dev), 1, 480, 640)
        grid = ops.events_norm(grid, (n_events - 1) / 500000 * 1.5)
        win = grid[:, 40:40 + crop, 120:120 + crop].contiguous()             # [1,400,400] == NHWC [1,400,400,1]
        ev[b] = ops.upsample_logits_nchw(win.view(1, crop, crop, 1), S, S)[0]   # bilinear, align_corners=False (dsec.py:318-319)
    return dict(source=dict(image=image, img_time_res=itr, img_self_res=isr(image), label=label),
                target=dict(warp_image=warp, events_vg=ev, warp_img_self_res=isr(warp)))


def loader_pairs(B, S, seed, dev, n_events=500000):
    """The same batch schema produced by the on-device loader pipeline (cmda_amd/datasets.py + pipeline.py):
    synthetic RAW streams -- 2048x1024 uint8 Cityscapes-like frames with their previous frame and label maps, 640x480 DSEC-like
    frames with `n_events` raw events per sample and a rectification map -- go through PIL-exact resize, random crop / flip,
    ToTensor + Normalize, the time residual, the Content-Extractor (ISR), event rectification -> voxel grid -> events_norm ->
    crop 400 -> bilinear 512, exactly as configs/fusion/cs2dsec_image+events_together_b5.py wires them (cityscapes_ic.py:147-272,
    dsec.py:189-366).  Built once, outside the timed region; resident in HBM."""
    import random
    from cmda_amd import datasets
    cfg = dict(type='UDADataset',
               source=dict(type='CityscapesICDataset', image_resize_size=(2 * S, S), image_crop_size=(S, S),
                           outputs={'image', 'label', 'img_time_res', 'img_self_res'}, isr_parms=dict(ISR_PARMS),
                           shift_type='random', synthetic_length=64, device=dev),
               target=dict(type='DSECDataset', crop_size=(400, 400), after_crop_resize_size=(S, S), events_bins=1,
                           isr_parms=dict(ISR_PARMS), outputs={'warp_image', 'events_vg', 'warp_img_self_res'},
                           shift_type='random', synthetic_length=64, synthetic_events=n_events, device=dev))
    ds = datasets.build_dataset(cfg)
    random.seed(seed)
    first = (seed * 7919) % (len(ds) - B)
    batch = ds.get_batch(list(range(first, first + B)))
    return dict(source={k: v.contiguous() for k, v in batch['source'].items()},
                target={k: v.contiguous() for k, v in batch['target'].items()})


def cpu_baseline_dacs(size):
    """The oracle's DACS iteration (oracle/dacs_iter.py: generator, EMA, source fwd/bwd, teacher, ClassMix + jitter + blur +
    ISR, mixed fwd/bwd) on ONE (source, target) pair, fp32, timed once on the host cores -- a bounded sample of the same
    workload (the bench's step does this for 2 pairs per GPU); the AdamW update (<1 % of the step) is not in the sample."""
    from oracle import cyclegan as ocg, dacs_iter, fusion as ofu, head as ohd, mit as omit, segmentor as oseg
    logical, physical = host_cores()
    cores = min(logical, 16)   # torch's CPU kernels at these sizes stop scaling there (see cpu_baseline_supervised)
    torch.set_num_threads(cores)
    torch.manual_seed(0)

    def net(dp, do):
        m = oseg.FusionEncoderDecoder(backbone_image=omit.mit_b5(drop_path_rate=dp), backbone_events=omit.mit_b5(drop_path_rate=dp),
                                      fusion_module=ofu.AttentionAvgFusion(drop_path_rate=dp),
                                      decode_head=ohd.DAFormerHeadFusion(dropout_ratio=do, share_decoder=True))
        m.backbone_image.init_weights()
        m.backbone_events.init_weights()
        return m.train()
    student, teacher, G = net(0.1, 0.1), net(0.0, 0.0), ocg.ResnetGenerator().eval()
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    src = dict(image=r(1, 3, size, size), img_time_res=r(1, 3, size, size).clamp(-1, 1), img_self_res=r(1, 3, size, size).clamp(-1, 1),
               label=synthetic_labels(1, size, g))
    tgt = dict(warp_image=r(1, 3, size, size), events_vg=r(1, 3, size, size).clamp(-1, 1),
               warp_img_self_res=r(1, 3, size, size).clamp(-1, 1))
    # thread-pool / allocator warm-up on a tiny problem (a full warm-up iteration would double the bench's CPU time)
    tiny = lambda d: {k: v[..., :64, :64].contiguous() for k, v in d.items()}  # noqa: E731
    dacs_iter.dacs_iteration(student, teacher, G, tiny(src), tiny(tgt), local_iter=0, forward_cfg=FORWARD_CFG, isr_parms=ISR_PARMS)
    for p in student.parameters():
        p.grad = None
    t0 = time.time()
    dacs_iter.dacs_iteration(student, teacher, G, src, tgt, local_iter=1, forward_cfg=FORWARD_CFG, isr_parms=ISR_PARMS,
                             shift_type='random')
    dt = time.time() - t0
    return {'value': round(1.0 / dt, 5), 'unit': 'img/s', 'cores': cores, 'host_physical_cores': physical,
            'host_logical_cpus': logical, 'kind': 'port',
            'sample': f'1 x (full DACS iteration on 1 source + 1 target pair at {size}x{size}: generator, EMA, source fwd/bwd, '
                      f'teacher, ClassMix/jitter/blur/ISR, mixed fwd/bwd; fp32 oracle/dacs_iter.py) on {cores} host threads '
                      f'after a 64x64 warm-up, {dt:.1f} s per pair'}


def profile_json(name):
    """A committed measurement under profiles/ (None if absent): `pmc_traffic_<workload>.json` = HBM bytes per GEMM-family
    launch from the PMC passes of this command (separate rocprofv3 --pmc runs, as MI355X_MICROARCH.md prescribes);
    `micro_peaks.json` = the box's measured MFMA / HBM-stream rates (tools/micro)."""
    path = os.path.join(ROOT, 'profiles', f'{name}.json')
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def accuracy_block():
    """Distance of each mode of the step to the CPU oracle at THIS configuration, read from the latest committed parity record
    (profiles/rNN_parity.json, written by tools/gpu/parity.sh = tests/test_dacs.py::test_dacs_iteration_full_depth_512_gpu run
    on the GPU box with CMDA_PARITY_JSON set): range-relative and element-wise logit error of the teacher's fusion logits,
    pseudo-label agreement.  The north star's tolerance is 1e-3 on the logits and bit-exact pseudo-label argmax."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_parity.json')))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    out = {'source': os.path.relpath(files[-1], ROOT), 'tolerance': 'logits within 1e-3 (north star); pseudo-label argmax equal',
           'what': rec.get('_what')}
    for mode in ('bf16', 'f32x3', 'f32'):
        r = rec.get(mode)
        if r:
            out[mode] = {k: r[k] for k in ('logit_err_range', 'logit_err_elementwise_p999', 'logit_err_elementwise_max',
                                           'pseudo_label_agreement', 'mixed_label_agreement', 'gradient_rel_err_p90') if k in r}
            out[mode]['within_tolerance'] = bool(r.get('logit_err_range', 1.0) <= 1e-3)
    return out


def rocprof_gemm_stats(workload):
    """GEMM-family launches / average duration in the committed rocprofv3 --kernel-trace --stats summary of this command's eager
    launch sequence (profiles/r04_<workload>_eager_kernel_stats.csv, else the latest earlier round's), for the cross-check against the live HIP-event figure"""
    import csv
    path = next((p for p in (os.path.join(ROOT, 'profiles', f'{r}_{workload}_eager_kernel_stats.csv') for r in ('r06', 'r05', 'r04', 'r03', 'r02'))
                 if os.path.exists(p)), os.path.join(ROOT, 'profiles', f'r02_{workload}_eager_kernel_stats.csv'))
    try:
        calls = ns = 0
        with open(path) as f:
            for r in csv.DictReader(f):
                if 'gemm' in r['Name']:
                    calls += int(r['Calls'])
                    ns += int(r['TotalDurationNs'])
        return {'file': os.path.relpath(path, ROOT), 'launches': calls, 'avg_launch_us': round(ns / max(calls, 1) / 1e3, 2)} if calls else None
    except (OSError, KeyError, ValueError):
        return None


def _family(n):
    """kernel family by name (the rule of tools/pmc_traffic.py)"""
    for key, fam in (('gemm', 'gemm'), ('attn', 'attention'), ('ln_', 'layernorm'), ('dw_', 'depthwise'), ('bn_', 'batchnorm')):
        if key in n:
            return fam
    return 'ce/pseudo-label' if ('ce_' in n or 'pseudo' in n) else 'other'


def hbm_families(workload):
    """The HBM-bound kernel families of the step against the 8 TB/s roof, from the COMMITTED measurements of this command (not from
    eager event timers, which mostly measure the launch path of a 5-us kernel): bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE of the
    PMC passes (profiles/pmc_traffic_<workload>.json), duration = the family's average kernel duration in the rocprofv3 kernel trace
    of the eager launch sequence (profiles/rNN_<workload>_eager_kernel_stats.csv).  frac = bytes / duration / 8 TB/s."""
    import csv
    pmc = profile_json(f'pmc_traffic_{workload}')
    path = next((p for p in (os.path.join(ROOT, 'profiles', f'{r}_{workload}_eager_kernel_stats.csv') for r in ('r06', 'r05', 'r04'))
                 if os.path.exists(p)), None)
    if not pmc or not path or 'families' not in pmc:
        return None
    dur = {}
    try:
        with open(path) as f:
            for r in csv.DictReader(f):
                a = dur.setdefault(_family(r['Name']), [0, 0])
                a[0] += int(r['Calls'])
                a[1] += int(r['TotalDurationNs'])
    except (OSError, KeyError, ValueError):
        return None
    out = {'peak': HBM_PEAK_TBS, 'unit': 'TB/s', 'bytes': 'rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE per launch (profiles/pmc_traffic_%s.json)' % workload,
           'durations': os.path.relpath(path, ROOT)}
    for fam in ('depthwise', 'layernorm', 'batchnorm', 'ce/pseudo-label'):
        t, d = pmc['families'].get(fam), dur.get(fam)
        if not t or not d or not d[0]:
            continue
        mb = t['fetch_mb_per_launch'] + t['write_mb_per_launch']
        us = d[1] / d[0] / 1e3
        tbs = mb / us   # MB / us = TB/s
        out[fam] = {'mb_per_launch': round(mb, 2), 'avg_us': round(us, 2), 'achieved': round(tbs, 3), 'frac': round(tbs / HBM_PEAK_TBS, 3),
                    'launches_in_trace': d[0], 'ms_in_trace': round(d[1] / 1e6, 2)}
    return out


def gemm_roofline(step, workload):
    """One more (eager) step with every GEMM launch bracketed by HIP events on the launch stream."""
    from cmda_amd import ops
    ops.GEMM_PROFILE = []
    step()
    torch.cuda.synchronize()
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    everything = prof
    prof = [e for e in everything if e[4][0] != 'attention']          # the GEMM / implicit-GEMM family (the dominant kernel)
    gemm_ms = sum(e0.elapsed_time(e1) for _, e0, e1, _, _ in prof)
    gemm_flops = sum(f for f, _, _, _, _ in prof)
    gemm_bytes = sum(b for _, _, _, b, _ in prof)
    # MFMA work by the part of the path that issued it (ops.site): GEMM launches + the fused attention kernels; a grouped
    # weight-gradient launch is split by the FLOP shares of its problems
    sites = {}
    for f, e0, e1, _, key in everything:
        ms = e0.elapsed_time(e1)
        split = key[-1] if isinstance(key[-1], dict) else {key[-1]: f}
        tot = sum(split.values()) or 1.0
        for name, fl in split.items():
            acc = sites.setdefault(name, [0.0, 0.0, 0])
            acc[0] += fl
            acc[1] += ms * fl / tot
            acc[2] += 1
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    if os.environ.get('CMDA_BENCH_GEMM_HIST'):   # per-shape table for tuning (tools/gpu): where the GEMM time of a step goes
        hist = {}
        for f, e0, e1, _, key in prof:
            key = tuple(v for v in key if isinstance(v, (int, bool, str)) and v not in ('mit', 'head', 'fusion', 'generator', 'other'))
            key = tuple(-1 if isinstance(v, str) else v for v in key)   # ('grouped' / 'pair' markers)
            h = hist.setdefault(key, [0, 0.0, 0.0])
            h[0] += 1
            h[1] += e0.elapsed_time(e1)
            h[2] += f
        with open(os.environ['CMDA_BENCH_GEMM_HIST'], 'w') as fh:
            fh.write('M N K batch splits conv a_kstr b_kstr atomic out_f32 | calls ms_total us_avg TFLOP/s\n')
            for key, (c, ms, f) in sorted(hist.items(), key=lambda kv: -kv[1][1]):
                fh.write(' '.join(str(int(v)) for v in key) + f' | {c} {ms:.3f} {ms / c * 1e3:.1f} {f / ms / 1e9:.1f}\n')
    if os.environ.get('CMDA_BENCH_GEMM_LOG'):   # launch-ordered record of this step's GEMM launches (tools/pmc_gemm_instances.py
        with open(os.environ['CMDA_BENCH_GEMM_LOG'], 'w') as fh:   # matches it with the dispatches of a PMC pass of the same command)
            json.dump([dict(flops=f, bytes=b, key=[v if isinstance(v, (int, bool, str)) else 'grouped' for v in key]) for f, _, _, b, key in prof], fh)
    pmc = profile_json(f'pmc_traffic_{workload}')
    roof = {'bound': 'mfma', 'kernel': 'gemm_glds_kernel + gemm_kernel (bf16 MFMA tile GEMM / implicit-GEMM conv family, all '
                                       'template instances)',
            'achieved': round(achieved, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
            'traffic': pmc['hbm_mb_per_launch'] * 1e6 if pmc else None,
            'launches_per_step': len(prof), 'avg_launch_us': round(gemm_ms * 1e3 / max(len(prof), 1), 2),
            'gemm_ms_per_step': round(gemm_ms, 3), 'algorithmic_gflop_per_step': round(gemm_flops / 1e9, 1),
            'algorithmic_mb_per_launch': round(gemm_bytes / max(len(prof), 1) / 1e6, 2)}
    def _site(v):
        fl, ms, n = v
        t = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {'gflop_per_step': round(fl / 1e9, 1), 'ms_per_step': round(ms, 3), 'launches': n, 'achieved': round(t, 2),
                'frac': round(t / MFMA_BF16_PEAK_TFLOPS, 4)}
    roof['sites'] = {k: _site(v) for k, v in sorted(sites.items())}
    if 'mit' in sites:   # the quantity BASELINE.json's >= 0.60 target is defined on: the MiT encoders' GEMMs + fused attention
        roof['mit_blocks'] = dict(_site(sites['mit']), target_frac=0.60,
                                  note='Linear / sr-conv / patch-embed GEMMs, their data and weight gradients and the fused attention '
                                       'kernels of both encoders, HIP events around eager launches (includes the launch gap)')
    if 'head' in sites:
        roof['decode_head'] = _site(sites['head'])
    if 'generator' in sites:
        roof['generator'] = _site(sites['generator'])
    if pmc:
        roof['traffic_source'] = pmc.get('source')
    hbm = hbm_families(workload)
    if hbm:
        roof['hbm'] = hbm
    micro = profile_json('micro_peaks')
    if micro:
        roof['peak_measured'] = micro
    rp = rocprof_gemm_stats(workload)
    if rp:   # kernel durations alone; the live event pairs also bracket the ~2-3 us dependent-launch gap
        roof['rocprof'] = rp
        t_rp = gemm_flops / (len(prof) * rp['avg_launch_us'] * 1e-6) / 1e12 if prof else 0.0
        roof['achieved_rocprof'] = round(t_rp, 2)
        roof['frac_rocprof'] = round(t_rp / MFMA_BF16_PEAK_TFLOPS, 4)   # (this run's FLOP count over the committed trace's kernel durations)
    return roof


def run_dacs(args, rank, world, dev, dist):
    from cmda_amd import optim
    from cmda_amd.parallel import GradAllReducer
    B = args.batch if args.batch > 0 else 2   # reference recipe: 2 source + 2 target samples per GPU
    torch.manual_seed(1234)                   # identical initial weights on every rank
    dacs = build_dacs(dev)
    opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01, custom_keys=CUSTOM_KEYS)
    # the step boundary (AdamW, gradient clear, EMA) on the optimizer's own stream, underneath the weight-free head of the next captured
    # iteration (optim.FlatAdamW.overlap, uda.DACS._iteration); CMDA_OPT_OVERLAP=0: in stream order (same-box A/B)
    opt.overlap = (not args.no_graph) and os.environ.get('CMDA_OPT_OVERLAP', '1') != '0'
    dacs.attach_flat_store(opt)
    # single rank with --force-reducer: the bucket is padded, scattered and gathered as for TWO ranks (GradAllReducer.virtual_ways), so the
    # RCCL reduce_scatter_tensor / all_gather_into_tensor calls of the multi-GPU exchange run with their shard-sized views on one GPU
    reducer = GradAllReducer(opt.flat_g, wire_dtype=torch.bfloat16, force=args.force_reducer, virtual_ways=2 if world == 1 else 1)
    if reducer.active:
        # overlap inside the LAST backward pass of the iteration: the decode head and both encoders (each back-propagated once
        # per pass -- the event encoder sees events + ISR as one batch) start their slices as they finish; the fusion blocks
        # and the small norm / bias group go in finish()
        student = dacs.model
        ranges = {('decode_head', id(student.decode_head)): opt.ranges_of(student, ['decode_head.'])}
        for name in ('backbone_image', 'backbone_events'):
            for s in range(1, 5):
                ranges[(f'backbone.stage{s}', id(getattr(student, name)))] = opt.ranges_of(
                    student, [f'{name}.patch_embed{s}.', f'{name}.block{s}.', f'{name}.norm{s}.'])

        def _ready(tag, module=None):
            for lo, hi in ranges.get((tag, id(module)), ()):
                reducer.start_range(lo, hi)
        dacs.final_pass_grad_hook = _ready
    batch = (loader_pairs if args.data == 'loader' else synthetic_pairs)(B, args.size, 100 + rank, dev)
    torch.manual_seed(1000 + rank)            # per-rank DropPath / Dropout / ClassMix streams
    it = [0]
    if not args.no_graph:
        dacs.enable_graph(warmup_iters=2)     # iterations 0-1 eager, iteration 2 captures the hipGraph, then replays
        if os.environ.get('CMDA_BENCH_LANES') is not None:   # A/B switch for the concurrency lanes inside the graph (tools/gpu)
            dacs.graph_lane_set = set(x for x in os.environ['CMDA_BENCH_LANES'].split(',') if x)

    def step():
        opt.zero_grad()
        log_vars = dacs(**batch)
        reducer.all_reduce_mean()
        opt.step(optim.poly_warm_scale(it[0]))
        it[0] += 1
        return log_vars

    def fence():
        opt.synchronize()                     # an overlapped update that is still postponed is LAUNCHED here: the timed region holds exactly K optimizer steps
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if not args.no_graph:
        for _ in range(3):                    # set-up, not warm-up: two eager iterations + the capturing one
            step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lv = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    losses = {k: round(float(v), 5) for k, v in lv.items() if 'loss' in k}
    graph_mode = ('linear hipGraph segments replayed on two streams (runtime.SegmentedCapture); EMA update, control-block copy, '
                  'AdamW outside') if dacs._graph is not None else 'eager'
    dacs.disable_graph()                      # the roofline leg brackets every GEMM launch with events: eager
    roofline = gemm_roofline(step, 'dacs') if rank == 0 else None
    if rank == 0:
        nparam = sum(p.numel() for p in dacs.model.parameters())
        value = B * world * args.steps / dt
        out = {'metric': 'training images/sec (512x512 image+event, MiT-B5)', 'value': round(value, 3),
               'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': 'BASELINE.json configs[3]: full CMDA UDA step (Motion-Extractor generator, EMA teacher, '
                                      'source CE fwd/bwd, teacher pseudo-labels, ClassMix + colour jitter + blur + ISR of the '
                                      'mixed image, mixed fwd/bwd, AdamW) on the image+events fusion student (two MiT-B5 '
                                      'encoders, AttentionAvgFusion, shared DAFormer head)',
                          'per_gpu_batch': f'{B} source + {B} target', 'global_batch': f'{B * world} + {B * world}',
                          'images_counted': 'one image = one (source, target) pair with its event / ISR companions '
                                            '(SURVEY.md 8d)', 'image_size': args.size, 'parallelism': f'dp{world}',
                          'rccl_ranks': dist.get_world_size() if dist is not None else 1,
                          'student_parameters_M': round(nparam / 1e6, 1), 'launch': graph_mode, 'generator': 'ResnetGenerator 9 blocks, in the step',
                          'synthetic_inputs': ('synthetic RAW streams (2048x1024 frames + previous frames + labels; 640x480 frames + 500k raw '
                                               'events/sample + rectification map) through the on-device loader pipeline of SURVEY.md 8 f3: '
                                               'PIL-exact resize, crop / flip, time residual, ISR, event rectification -> voxel grid -> '
                                               'events_norm -> crop 400 -> 512') if args.data == 'loader' else
                                              ('SURVEY.md 8d: 500k events/sample through the voxel kernel, ISR through the ISR '
                                               'kernels, sparse time residual, block labels with 5 % ignore')},
               'losses': losses,
               'model_gflop_per_pair': GFLOP_PER_PAIR_UDA,
               'model_tflops_achieved': round(GFLOP_PER_PAIR_UDA * value / 1e3 / world, 2),
               'roofline': roofline}
        out['schema'] = 3   # 2: x3_* (split-bf16) and exact_f32_* carry the fp32-storage modes; parity_mode_* = alias of x3_* (ADVICE r04); 3: + accuracy
        out['accuracy'] = accuracy_block()   # per mode: distance to the oracle at this configuration (committed parity run)
        if world == 1 and args.dtype == 'bf16' and not args.no_parity_mode:
            # the children allocate a whole fp32 model each: hand the parent's cached activation blocks back first
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out.update(parity_mode_line(args))
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_dacs(args.size)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def parity_mode_line(args):
    """The SAME step in the two fp32-storage modes, each timed in a CHILD process after the bf16 measurement, so the 1e-3 parity
    claim has a throughput attached: `parity_mode_*` = split-bf16 (bf16 x 3) GEMMs on fp32 storage (runtime.set_gemm_x3: the
    tolerance-meeting mode, tests/test_dacs.py::test_dacs_iteration_full_depth_512_gpu[x3]); `exact_f32_*` = the exact-fp32 matrix
    instruction (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate: what rounds 1-3 reported as parity_mode).  The bf16 line's own
    distance to the oracle is in the line's `accuracy` block (accuracy_block: the committed parity run, profiles/rNN_parity.json)."""
    import subprocess
    out = {}
    for key, dt, what in (('x3', 'f32x3', 'dtype f32x3: fp32 storage, split-bf16 (bf16 x 3) MFMA GEMMs, same step, 3 timed steps in a child process'),
                          ('exact_f32', 'f32', 'dtype f32: exact-fp32 MFMA GEMMs and fp32 storage, same step, 3 timed steps in a child process')):
        cmd = [sys.executable, os.path.abspath(__file__), '--dtype', dt, '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
               '--no-parity-mode', '--data', args.data, '--size', str(args.size)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            out.update({key + '_ms_per_step': d['ms_per_step'], key + '_img_per_s': d['value'], key: what})
            if key == 'x3':   # the keys rounds 1-4 used: exact fp32 until round 3, split-bf16 since round 4 -- `parity_mode_kind` says which
                out.update({'parity_mode_ms_per_step': d['ms_per_step'], 'parity_mode_img_per_s': d['value'], 'parity_mode_kind': 'f32x3'})
        except Exception as e:   # noqa: BLE001  (the headline must not die with the secondary figures)
            out.update({key + '_ms_per_step': None, key + '_error': repr(e)[:200]})
    return out


def run_supervised(args, rank, world, dev, dist):
    import cmda_amd.runtime as rt
    from cmda_amd import optim
    from cmda_amd.parallel import GradAllReducer
    B = args.batch if args.batch > 0 else 64
    torch.manual_seed(1234)  # identical initial weights on every rank
    model = build_model(dev)
    opt = optim.FlatAdamW(model, lr=6e-5, weight_decay=0.01, custom_keys=CUSTOM_KEYS)
    reducer = GradAllReducer(opt.flat_g, wire_dtype=torch.bfloat16, force=args.force_reducer)  # no-op at world size 1
    if reducer.active:
        # overlap: a stage's weight gradients (one contiguous slice of the flat buffer) start their all-reduce on the
        # side stream as soon as that stage's backward is done; the step's reducer.all_reduce_mean() does the rest
        stage_ranges = {f'backbone.stage{s}': opt.ranges_of(model, [f'backbone.patch_embed{s}.', f'backbone.block{s}.',
                                                                     f'backbone.norm{s}.']) for s in range(1, 5)}
        stage_ranges['decode_head'] = opt.ranges_of(model, ['decode_head.'])

        def _ready(tag, module=None):
            for lo, hi in stage_ranges.get(tag, ()):
                reducer.start_range(lo, hi)
        rt.grad_ready_hook = _ready
    torch.manual_seed(1000 + rank)  # per-rank DropPath / Dropout streams
    img, gt = synthetic_batch(B, args.size, rank, dev)
    it = [0]

    def step():
        opt.zero_grad()
        losses, _ = model.forward_train(img, None, gt)
        losses['decode.loss_seg'].backward()
        reducer.all_reduce_mean()
        opt.step(optim.poly_warm_scale(it[0]))
        it[0] += 1
        return losses['decode.loss_seg']

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(loss.item())
    roofline = gemm_roofline(step, 'supervised') if rank == 0 else None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        out = {'metric': 'training images/sec (512x512, MiT-B5 + DAFormer head fwd/bwd + AdamW step)', 'value': round(value, 3),
               'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': 'BASELINE.json configs[1]: MiT-B5 + DAFormer(sep-ASPP) head fwd/bwd, random '
                                      f'{args.size}x{args.size}, HIP kernels', 'global_batch': B * world,
                          'per_gpu_batch': B, 'image_size': args.size, 'parallelism': f'dp{world}',
                          'drop_path_rate': 0.1, 'dropout_ratio': 0.1, 'optimizer': 'AdamW (fused, flat buffers)'},
               'final_loss': round(loss_val, 5),
               'model_gflop_per_image_fwd_bwd': round(3 * GFLOP_PER_IMAGE_FWD, 1),
               'model_tflops_achieved': round(3 * GFLOP_PER_IMAGE_FWD * value / 1e3 / world, 2),
               'roofline': roofline}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_supervised(args.size)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def self_launch(n):
    """one child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1` would (tools/train.py:104 of the reference delegates this to its
    launcher); the children are FRESH interpreters (never fork / exec a process that has initialised the GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:      # POLL: a rank that dies (OOM, RCCL init failure) must not leave the launcher blocked on a
            for p in list(live):     # peer that sits in a collective waiting for it
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                rc = rc or r
            if live and rc == 0:
                time.sleep(0.2)
    finally:
        for p in procs:          # a rank died (or we were interrupted): end its peers
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def dry_run(args, rank, world):
    """`--dry-run`: the launch / rendezvous / max-over-ranks / one-JSON-line plumbing with gloo on the host and no GPU work --
    what the CPU test-suite can check of the N > 1 path (tests/test_parallel.py)."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from cmda_amd.parallel import shard_range
    lo, hi = shard_range(2 * world, rank, world)       # rank r owns pairs [2r, 2r+1] of the global batch
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0 + 1e-9])
    pairs = torch.tensor([float(hi - lo)])
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(pairs)
    if rank == 0:
        print(json.dumps({'metric': 'training images/sec (512x512 image+event, MiT-B5)', 'value': None, 'unit': 'img/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'dry_run': True, 'scaling': 'weak',
                          'config': {'global_batch': f'{int(pairs.item())} + {int(pairs.item())}', 'parallelism': f'dp{world}',
                                     'ranks': dist.get_world_size() if world > 1 else 1, 'backend': 'gloo (dry run, no GPU work)'}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--data', choices=['loader', 'direct'], default='loader',
                    help='dacs workload: batch built by the on-device loader pipeline from synthetic raw streams (default) or directly '
                         'from the voxel / ISR kernels (section 8d)')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('CMDA_BENCH_BATCH', 0)),
                    help='per-GPU batch (default: 2 source + 2 target for dacs, 64 images for supervised)')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'f32x3'],
                    help='bf16 = the speed mode (the headline); f32 = exact-fp32 MFMA GEMMs; f32x3 = fp32 storage with split-bf16 (bf16 x 3) GEMMs, '
                         'the tolerance-meeting mode (runtime.set_gemm_x3)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='dacs', choices=['dacs', 'supervised'],
                    help="dacs = BASELINE.json configs[3]/[4] (the bench line the driver reads): one full CMDA UDA iteration per "
                         "step; supervised = configs[1] (MiT-B5 + DAFormer head fwd/bwd)")
    ap.add_argument('--no-graph', action='store_true', help='dacs: launch every kernel from Python instead of replaying the hipGraph')
    ap.add_argument('--no-parity-mode', action='store_true', help='dacs: skip the second figure (the same step in the exact-fp32 mode)')
    ap.add_argument('--dry-run', action='store_true', help='launch / rendezvous plumbing only (gloo, no GPU work): CPU tests')
    ap.add_argument('--force-reducer', action='store_true',
                    help='testing: run the gradient all-reduce path (RCCL, side stream, staged ranges) even with one rank')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  This process has not touched the GPU (no HIP call,
        # no torch.cuda.* so far) and never will: it starts N fresh children (one rank per GPU, the env torchrun would set),
        # waits, and exits with the worst return code; rank 0 prints the JSON line.
        return sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f'bench.py --gpus {args.gpus}: WORLD_SIZE is {world} (start it plain, or under torch.distributed.run with '
                 f'--nproc-per-node {args.gpus})')
    if args.gpus == 1 and world != 1:
        sys.exit(f'--gpus 1 but WORLD_SIZE={world}')
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_reducer:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=dev)

    import cmda_amd.runtime as rt
    rt.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    rt.set_gemm_x3(args.dtype == 'f32x3')
    if os.environ.get('CMDA_BENCH_GEMM_HINT'):   # tuning A/B (tools/gpu): cmda_gemm_params_t.tile_hint for every GEMM of the run
        from cmda_amd import ops
        ops.GEMM_TILE_HINT = int(os.environ['CMDA_BENCH_GEMM_HINT'])
    if args.workload == 'dacs':
        return run_dacs(args, rank, world, dev, dist)
    return run_supervised(args, rank, world, dev, dist)


if __name__ == '__main__':
    main()
