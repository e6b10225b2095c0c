#!/usr/bin/env python3
"""Headline benchmark: training images/s of the CMDA hot path on MI355X (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1]: MiT-B5 + DAFormer (sep-ASPP) head forward/backward (+ fused AdamW step),
synthetic 512x512 inputs, bf16 activations / fp32 accumulate, random-init weights of the real architecture.
N>1: one process per GPU (torchrun), weak scaling (per-GPU batch fixed), gradients all-reduced (mean) with RCCL.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel family = the MFMA GEMM, HIP-event timed on the launch
stream) and, at N=1, `cpu_baseline` (the oracle restatement timed on the host cores, bounded sample).
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
GFLOP_PER_IMAGE_FWD = 252.6     # MiT-B5 138.8 + DAFormer head 113.8


def build_model(dev, drop_path_rate=0.1, dropout_ratio=0.1):
    import cmda_amd  # noqa: F401
    from cmda_amd.registry import build_segmentor
    cfg = dict(type='EncoderDecoder',
               backbone=dict(type='mit_b5', style='pytorch', drop_path_rate=drop_path_rate),
               decode_head=dict(type='DAFormerHead', in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=256,
                                dropout_ratio=dropout_ratio, num_classes=19, norm_cfg=dict(type='BN', requires_grad=True),
                                align_corners=False,
                                decoder_params=dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                                    embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18),
                                                                    pool=False, act_cfg=dict(type='ReLU'),
                                                                    norm_cfg=dict(type='BN', requires_grad=True))),
                                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
               train_cfg=dict(), test_cfg=dict(mode='whole'))
    model = build_segmentor(cfg)
    model.init_weights()
    return model.to(dev).train()


def synthetic_batch(B, size, seed, dev):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, size, size, generator=g)
    lab = torch.randint(0, 19, (B, 1, size // 32, size // 32), generator=g)
    lab = lab.repeat_interleave(32, 2).repeat_interleave(32, 3)
    lab[torch.rand(B, 1, size, size, generator=g) < 0.05] = 255
    return img.to(dev), lab.to(dev)


def cpu_baseline(size):
    """Oracle (port of the reference algorithm, oracle/) timed on the host cores: one fwd+bwd of MiT-B5 + DAFormer
    head on ONE image (bounded sample of the same workload)."""
    from oracle import head as ohd, mit as omit, segmentor as oseg
    # 16 threads: torch's CPU kernels at these sizes stop scaling there (256 threads took 737 s for this sample on the
    # GPU box's host, 8 threads take 11 s) -- `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 16)
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
    return {'value': round(1.0 / dt, 4), 'unit': 'img/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} x (1 image {size}x{size}, fwd+bwd of MiT-B5+DAFormerHead, fp32 oracle) on {cores} host '
                      f'threads after 1 warm-up, {dt:.2f} s per image'}


def build_dacs(dev):
    """configs/fusion/cs2dsec_image+events_together_b5.py restated inline (the reference tree is not on the GPU box): two
    MiT-B5 encoders, AttentionAvgFusion, shared DAFormerHeadFusion, DACS with ClassMix + ISR of the mixed image."""
    import functools
    import cmda_amd  # noqa: F401
    from cmda_amd.registry import build_train_model
    dims = [64, 128, 320, 512]
    bb = dict(type='mit_b5', style='pytorch', drop_path_rate=0.1)
    head = dict(type='DAFormerHeadFusion', in_channels=dims, in_index=[0, 1, 2, 3], channels=256, dropout_ratio=0.1,
                num_classes=19, norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                decoder_params=dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                                    fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False,
                                                    act_cfg=dict(type='ReLU'), norm_cfg=dict(type='BN', requires_grad=True)),
                                    train_type='cs2dsec_image+events_together', share_decoder=True),
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    model = dict(type='FusionEncoderDecoder', backbone_image=dict(bb), backbone_events=dict(bb),
                 fusion_module=dict(type='AttentionAvgFusion', in_channels=dims, drop_path_rate=0.1), decode_head=head,
                 train_type='cs2dsec_image+events_together', train_cfg=dict(), test_cfg=dict(mode='whole'))
    fcfg = dict(loss_weight={'image': 0.5, 'events': 0.5, 'fusion': 0.5, 'img_self_res': 0.25}, gradual_rate=0.0)
    uda = dict(type='DACS', alpha=0.999, pseudo_threshold=0.968, pseudo_weight_ignore_top=15, pseudo_weight_ignore_bottom=120,
               imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None, imnet_feature_dist_scale_min_ratio=None,
               mix='class', blur=True, color_jitter_strength=0.2, color_jitter_probability=0.2, debug_img_interval=1000,
               print_grad_magnitude=False, train_type='cs2dsec_image+events_together', forward_cfg=fcfg,
               cyclegan_itrd2en_path='', img_self_res_reg='no', mixed_image_to_mixed_isr=True, random_choice_thres='0.5',
               shift_type='rightdown', isr_parms=dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1),
               sky_mask=None)
    dacs = build_train_model(dict(model=model, uda=uda, runner=dict(type='IterBasedRunner', max_iters=40000)))
    del functools
    return dacs.to(dev).train()


def run_dacs(args, rank, world, dev, dist):
    import cmda_amd.runtime as rt
    from cmda_amd import optim
    from cmda_amd.parallel import GradAllReducer
    B = args.batch if '--batch' in sys.argv else 2   # reference recipe: 2 source + 2 target samples per GPU
    torch.manual_seed(1234)
    dacs = build_dacs(dev)
    opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01,
                          custom_keys=dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0)))
    dacs.attach_flat_store(opt)
    reducer = GradAllReducer(opt.flat_g, wire_dtype=torch.bfloat16, force=args.force_reducer)
    if reducer.active:
        # overlap inside the LAST backward pass of the iteration: the decode head and the image encoder (back-propagated once
        # per pass) start their slices as they finish; the event encoder (twice per pass) and the rest go in finish()
        student = dacs.model
        ranges = {('decode_head', id(student.decode_head)): opt.ranges_of(student, ['decode_head.'])}
        for s in range(1, 5):
            ranges[(f'backbone.stage{s}', id(student.backbone_image))] = opt.ranges_of(
                student, [f'backbone_image.patch_embed{s}.', f'backbone_image.block{s}.', f'backbone_image.norm{s}.'])

        def _ready(tag, module=None):
            for lo, hi in ranges.get((tag, id(module)), ()):
                reducer.start_range(lo, hi)
        dacs.final_pass_grad_hook = _ready
    g = torch.Generator().manual_seed(100 + rank)
    S = args.size
    lab = torch.randint(0, 19, (B, 1, S // 32, S // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3)
    r = lambda: torch.randn(B, 3, S, S, generator=g)  # noqa: E731
    batch = dict(source=dict(image=r().to(dev), img_time_res=r().clamp(-1, 1).to(dev), img_self_res=r().clamp(-1, 1).to(dev),
                             label=lab.to(dev)),
                 target=dict(warp_image=r().to(dev), events_vg=r().clamp(-1, 1).to(dev), warp_img_self_res=r().clamp(-1, 1).to(dev)))
    torch.manual_seed(1000 + rank)
    it = [0]

    def step():
        opt.zero_grad()
        log_vars = dacs(**batch)
        reducer.all_reduce_mean()
        opt.step(optim.poly_warm_scale(it[0]))
        it[0] += 1
        return log_vars

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    if rank == 0:
        nparam = sum(p.numel() for p in dacs.model.parameters())
        out = {'metric': 'training images/sec (512x512 image+event, full CMDA UDA step)', 'value': round(2 * B * world * args.steps / dt, 3),
               'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'bf16' if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE.json configs[3]: full CMDA UDA step (source CE + EMA-teacher pseudo-labels, ClassMix, '
                                      'ISR of the mixed image, colour jitter / blur, two student passes, AdamW), image+events fusion '
                                      'student with two MiT-B5 encoders',
                          'per_gpu_batch': f'{B} source + {B} target', 'images_counted': 'source + target samples',
                          'image_size': S, 'parallelism': f'dp{world}', 'student_parameters_M': round(nparam / 1e6, 1)},
               'losses': {k: round(float(v), 5) for k, v in lv.items() if 'loss' in k}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    del rt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('CMDA_BENCH_BATCH', 64)), help='images per GPU')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='supervised', choices=['supervised', 'dacs'],
                    help="supervised = BASELINE.json configs[1] (the bench line the driver reads); dacs = configs[3]/[4], one "
                         "full CMDA UDA iteration (EMA teacher, pseudo-labels, ClassMix, ISR, two student passes) per step")
    ap.add_argument('--force-reducer', action='store_true',
                    help='testing: run the gradient all-reduce path (RCCL, side stream, staged ranges) even with one rank')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
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
    from cmda_amd import ops, optim
    rt.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    if args.workload == 'dacs':
        return run_dacs(args, rank, world, dev, dist)
    torch.manual_seed(1234)  # identical initial weights on every rank
    model = build_model(dev)
    opt = optim.FlatAdamW(model, lr=6e-5, weight_decay=0.01,
                          custom_keys=dict(head=dict(lr_mult=10.0), pos_block=dict(decay_mult=0.0), norm=dict(decay_mult=0.0)))
    from cmda_amd.parallel import GradAllReducer
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
    img, gt = synthetic_batch(args.batch, args.size, rank, dev)
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

    # roofline leg: one more step with every GEMM launch bracketed by events on the launch stream
    ops.GEMM_PROFILE = []
    step()
    torch.cuda.synchronize()
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    gemm_ms = sum(e0.elapsed_time(e1) for _, e0, e1, _ in prof)
    gemm_flops = sum(f for f, _, _, _ in prof)
    gemm_bytes = sum(b for _, _, _, b in prof)
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    roofline = {'bound': 'mfma', 'kernel': 'gemm_glds_kernel + gemm_kernel (bf16 MFMA 16x16x32 tile GEMM / implicit-GEMM conv family, all template instances)',
                'achieved': round(achieved, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                'launches_per_step': len(prof), 'avg_launch_us': round(gemm_ms * 1e3 / max(len(prof), 1), 2),
                'gemm_ms_per_step': round(gemm_ms, 3), 'algorithmic_gflop_per_step': round(gemm_flops / 1e9, 1),
                'algorithmic_mb_per_launch': round(gemm_bytes / max(len(prof), 1) / 1e6, 2),
                'traffic_note': 'PMC FETCH_SIZE/WRITE_SIZE of the same command are in profiles/README.md (separate rocprofv3 passes)'}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        out = {'metric': 'training images/sec (512x512, MiT-B5 + DAFormer head fwd/bwd + AdamW step)', 'value': round(value, 3),
               'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'bf16' if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE.json configs[1]: MiT-B5 + DAFormer(sep-ASPP) head fwd/bwd, random '
                                      f'{args.size}x{args.size}, HIP kernels', 'global_batch': args.batch * world,
                          'per_gpu_batch': args.batch, 'image_size': args.size, 'parallelism': f'dp{world}',
                          'drop_path_rate': 0.1, 'dropout_ratio': 0.1, 'optimizer': 'AdamW (fused, flat buffers)'},
               'final_loss': round(loss_val, 5),
               'model_gflop_per_image_fwd_bwd': round(3 * GFLOP_PER_IMAGE_FWD, 1),
               'model_tflops_achieved': round(3 * GFLOP_PER_IMAGE_FWD * value / 1e3 / world, 2),
               'roofline': roofline}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.size)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
