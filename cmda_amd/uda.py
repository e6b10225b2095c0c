"""DACS self-training step of CMDA on the HIP kernels -- registry key `DACS`.

Mirrors mmseg/models/uda/dacs.py::DACS (__init__ :55-242, _init_ema_weights/_update_ema :250-272, train_step :274-315,
forward_train :357-1099) and uda_decorator.py::UDADecoratorFusion :178-252 for the two train types of configs/fusion/*:
'cs2dsec_image+events_together' and 'cs2dz_image+raw-isr' (plus 'cs2dsec_image+events').  What changes is where the
work runs, not what is computed:
  * EMA teacher update: one fused kernel per parameter tensor (or one over the flat buffer when the student's
    parameters live in cmda_amd.optim.FlatAdamW's flat store) instead of a Python loop of torch ops;
  * teacher soft-max / max / threshold count / pseudo-weight: one fused up-sample+argmax kernel on the 1/4-resolution
    logits (no 19 x H x W tensors, no `.cpu()` sync: the confident-pixel count stays on the device);
  * ClassMix of image / events / label / weight: batched kernels (no per-sample Python loop);
  * ISR of the mixed image: on the device (the reference round-trips every sample through PIL on the host);
  * log values stay device scalars (`_parse_losses`' `.item()` syncs are gone); call `log_vars_to_float` when logging.
Out of scope here: the other six train types, OrgDACS, ImageNet feature distance, the
matplotlib debug panels, sky-mask / flare / cow-mask augmentations.
"""
import os
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import runtime as rt
from .cyclegan import define_G
from .parallel import reduce_log_vars
from .registry import UDA, build_segmentor
from .segmentors import add_prefix, parse_losses

_DIRECT = [['leftdown', 'leftup'], ['rightdown', 'rightup']]


def set_stochastic(model, flag):
    """DropPath / Dropout2d on or off without touching BatchNorm's train mode (dacs.py:458-462 for the teacher)."""
    for m in model.modules():
        if hasattr(m, 'drop_path_rate') or hasattr(m, 'dropout_ratio'):
            m.stochastic = flag


def tgt_shape(tgt):
    """shape of the target batch's image tensor (key name differs between the train types)"""
    t = tgt['warp_image'] if 'warp_image' in tgt else tgt['image']
    return t.shape


def log_vars_to_float(log_vars):
    return {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in log_vars.items()}


def _null_ctx():
    import contextlib
    return contextlib.nullcontext()


# concurrency lanes of the captured iteration (DACS._capture; CMDA_GRAPH_LANES / DACS.graph_lane_set override): runtime.py documents them
GRAPH_LANES = tuple(x for x in os.environ.get('CMDA_GRAPH_LANES', 'enc,T,Tenc,wq').split(',') if x)


@UDA.register_module()
class DACS(nn.Module):
    SUPPORTED = {'cs2dsec_image+events', 'cs2dz_image+raw-isr', 'cs2dsec_image+events_together'}

    def __init__(self, **cfg):
        super().__init__()
        self.model = build_segmentor(deepcopy(cfg['model']))
        self.train_cfg = cfg['model'].get('train_cfg')
        self.test_cfg = cfg['model'].get('test_cfg')
        self.num_classes = cfg['model']['decode_head']['num_classes']
        self.local_iter = 0
        self.max_iters = cfg['max_iters']
        self.alpha = cfg['alpha']
        self.pseudo_threshold = cfg['pseudo_threshold']
        self.psweight_ignore_top = cfg['pseudo_weight_ignore_top']
        self.psweight_ignore_bottom = cfg['pseudo_weight_ignore_bottom']
        self.fdist_lambda = cfg['imnet_feature_dist_lambda']
        assert not self.fdist_lambda > 0, 'ImageNet feature distance is off in configs/fusion/* and not implemented'
        self.mix = cfg['mix']
        assert self.mix == 'class'
        self.blur = cfg['blur']
        self.color_jitter_s = cfg['color_jitter_strength']
        self.color_jitter_p = cfg['color_jitter_probability']
        self.debug_img_interval = cfg['debug_img_interval']
        self.ema_model = build_segmentor(deepcopy(cfg['model']))
        self.train_type = cfg['train_type']
        assert self.train_type in self.SUPPORTED, f'train_type {self.train_type} is outside the accelerated hot path'
        self.forward_cfg = dict(cfg['forward_cfg'])
        self.img_self_res_reg = cfg.get('img_self_res_reg', 'no')
        path = cfg.get('cyclegan_itrd2en_path', '')
        self.cyclegan_itrd2en = None
        if path and self.train_type in {'cs2dsec_image+events', 'cs2dsec_image+events_together'}:
            self.cyclegan_itrd2en = define_G()
            if path != 'random':  # 'random' = seeded random init (no checkpoint exists offline; bench / tests)
                self.cyclegan_itrd2en.load_state_dict(torch.load(path, map_location='cpu'))
            self.cyclegan_itrd2en.eval()
            for p in self.cyclegan_itrd2en.parameters():
                p.requires_grad_(False)
                p._cmda_frozen = True   # runtime: its re-laid-out compute copies survive optimizer steps
        assert cfg.get('sky_mask') is None, 'sky-mask augmentation is off in configs/fusion/* and not implemented'
        self.mixed_image_to_mixed_isr = bool(cfg.get('mixed_image_to_mixed_isr'))
        self.isr_parms = {'val_range': (1, 10 ** 2), '_threshold': 0.04, '_clip_range': 0.2, 'shift_pixel': 3}
        if cfg.get('isr_parms'):
            self.isr_parms = dict(cfg['isr_parms'])
        assert self.mixed_image_to_mixed_isr, 'configs/fusion/* recompute the ISR from the mixed image'
        self.isr_another_fusion = bool(cfg.get('isr_another_fusion'))
        self.fuse_both_ice_and_e = bool(cfg.get('fuse_both_ice_and_e'))
        self.without_events = bool(cfg.get('without_events'))
        self.without_isd = bool(cfg.get('without_isd'))
        self.isr_no_fusion = bool(cfg.get('isr_no_fusion'))
        rct = cfg.get('random_choice_thres', '')
        self.random_choice_thres = float(rct) if rct in {'0.25', '0.75', '0.5'} else 0.5
        self.shift_type = cfg.get('shift_type') or 'rightdown'
        assert self.shift_type in {'all', 'random', 'rightdown'}
        lfc = cfg.get('lambda_feature_consistency', -1)
        self.forward_cfg['lambda_feature_consistency'] = lfc if lfc != -1 else 0.25
        self._flat = None
        self._graph, self._graphs = None, {}
        self._graph_warmup = None
        for p in self.ema_model.parameters():
            p.requires_grad_(False)

    # -- UDADecoratorFusion ------------------------------------------------------------------------------------------
    def get_model(self):
        return self.model

    def get_ema_model(self):
        return self.ema_model

    def extract_feat(self, img):
        return self.get_model().extract_feat(img)

    def encode_decode(self, img, events, **kw):
        return self.get_model().encode_decode(img, events, **kw)

    def simple_test(self, rescale=True, **kwargs):
        return self.get_model().simple_test(rescale, **kwargs)

    def init_weights(self):
        self.model.init_weights()
        self.ema_model.init_weights()

    # -- EMA teacher -----------------------------------------------------------------------------------------------------
    def attach_flat_store(self, opt):
        """Re-home the teacher's parameters into a flat buffer laid out like the student's (cmda_amd.optim.FlatAdamW),
        so that the EMA update is ONE kernel over 177.8 M floats."""
        offsets = {}
        flat_p = opt.flat_p
        base = flat_p.data_ptr()
        for n, p in self.model.named_parameters():
            offsets[n] = (p.data_ptr() - base) // 4
        ema_flat = torch.zeros_like(flat_p)
        from .optim import _slot_view
        student = dict(self.model.named_parameters())
        with torch.no_grad():
            for n, p in self.ema_model.named_parameters():
                off = offsets[n]
                view = _slot_view(ema_flat, off, student[n])   # same storage order as the student's slot (channels-last conv weights)
                view.copy_(p.data)
                p.data = view
        self._flat = (flat_p, ema_flat)
        self._opt = opt   # (an overlapped update -- FlatAdamW.overlap -- is waited for inside the iteration, see _iteration)
        # bf16 compute copies of the teacher in one flat mirror too, refreshed by ONE cast after each EMA update (without
        # it every teacher Linear weight is cast on its own: ~1100 extra launches per iteration)
        self._ema_bf16 = None
        if rt.compute_dtype() == torch.bfloat16 and ema_flat.is_cuda:
            self._ema_bf16 = torch.empty(ema_flat.numel(), dtype=torch.bfloat16, device=ema_flat.device)
            for n, p in self.ema_model.named_parameters():
                p._cmda_bf16 = _slot_view(self._ema_bf16, offsets[n], student[n])
            self._sync_ema_bf16()

    def _sync_ema_bf16(self):
        if getattr(self, '_ema_bf16', None) is not None:
            n = self._flat[1].numel()
            ops.permute4(self._flat[1], self._ema_bf16, (n, 1, 1, 1), (0, 1, 2, 3))

    @staticmethod
    def _ema_one(e, p, alpha):
        if e.data.is_contiguous() and p.data.is_contiguous():
            ops.ema_update(e.data.view(-1), p.data.view(-1), alpha)
        else:   # differently stored tensors (a channels-last student slot against a plain teacher tensor): by logical index
            e.data.mul_(alpha).add_(p.data, alpha=1.0 - alpha)

    def _init_ema_weights(self):
        if self._flat is not None:
            ops.ema_update(self._flat[1], self._flat[0], 0.0, mirror=getattr(self, '_ema_bf16', None))
        else:
            for e, p in zip(self.ema_model.parameters(), self.model.parameters()):
                self._ema_one(e, p, 0.0)
        rt.invalidate()

    def _update_ema(self, it):
        alpha_teacher = min(1 - 1 / (it + 1), self.alpha)
        if self._flat is not None:
            ops.ema_update(self._flat[1], self._flat[0], alpha_teacher, mirror=getattr(self, '_ema_bf16', None))
        else:
            for e, p in zip(self.ema_model.parameters(), self.model.parameters()):
                self._ema_one(e, p, alpha_teacher)
        rt.invalidate()

    # -- one UDA iteration ---------------------------------------------------------------------------------------------------
    def train_step(self, data_batch, optimizer, **kwargs):
        optimizer.zero_grad()
        log_vars = self(**data_batch)
        optimizer.step()
        # base.py:736-741: under data parallelism every logged scalar is its mean over the ranks (one small all-reduce)
        log_vars = reduce_log_vars(log_vars)
        log_vars.pop('loss', None)
        src = data_batch['source']
        n = src['image'].shape[0] if 'image' in src else data_batch['target']['warp_image'].shape[0]
        return dict(log_vars=log_vars, num_samples=n)

    def forward(self, **kwargs):
        return self.forward_train(**kwargs)

    def _choose_classes(self, labels):
        """get_class_masks (dacs_transforms.py:101-112): classes = unique over the WHOLE batch, ceil(n/2) drawn per
        sample with np.random.choice.  One small device->host read (<= 20 class ids), as in the reference.  Returns a CPU
        int64 [B, Kmax] tensor padded with -1 (Kmax fixed by num_classes, so the launch shapes never change)."""
        host = getattr(labels, '_cmda_classes', None)
        if host is not None and getattr(labels, '_cmda_classes_key', None) != (labels.data_ptr(), labels._version):
            host = None   # the label buffer was refilled in place after the loader attached its class set: recompute (dacs_transforms.py:103)
        if host is not None:
            # the loader took the class set on the host from the cropped labels before the H2D copy (datasets.CityscapesICDataset):
            # no device read, no sync, nothing to order
            classes = host
        elif labels.is_cuda and getattr(self, '_graph_warmup', None) is not None:
            # graph mode: the host runs an iteration ahead of the GPU, so the read goes through a side stream instead of draining
            # the main one -- ORDERED after the label's producer: the event the loader recorded behind its copy (`_cmda_ready`),
            # else everything enqueued so far on the current stream (safe for any producer; costs the run-ahead).  record_stream
            # keeps the caching allocator from recycling the buffer under the reader.
            side = getattr(self, '_side_stream', None) or torch.cuda.Stream(labels.device)
            self._side_stream = side
            ready = getattr(labels, '_cmda_ready', None)
            if ready is not None:
                side.wait_event(ready)
            else:
                side.wait_stream(torch.cuda.current_stream(labels.device))
            with torch.cuda.stream(side):
                classes = torch.unique(labels).cpu()
            labels.record_stream(side)
        else:
            classes = torch.unique(labels).cpu()
        n = classes.shape[0]
        k = int((n + n % 2) / 2)
        out = torch.full((labels.shape[0], self._kmax()), -1, dtype=torch.int64)
        for i in range(labels.shape[0]):
            pick = np.random.choice(n, k, replace=False)
            out[i, :k] = classes[torch.as_tensor(pick).long()]
        return out

    def _kmax(self):
        return (self.num_classes + 2) // 2   # labels 0..num_classes-1 plus the ignore index -> at most ceil((nc+1)/2) classes

    # -- host decisions of one iteration (everything random that the reference draws on the host) -------------------------------
    def _draw(self, day_label, H, W):
        """dacs.py:417 (events / ISR choice), :446-456 (strong_parameters), dacs_transforms.py:101-112 (class draw), and the
        per-sample kornia ColorJitter draws of strong_transform (one call per sample, dacs.py:721-724)."""
        B = day_label.shape[0]
        d = {}
        if self.train_type == 'cs2dz_image+raw-isr':
            d['choice'] = 0.0
        elif self.without_events:
            d['choice'] = -1.0
        elif self.without_isd:
            d['choice'] = 2.0
        else:
            d['choice'] = float(torch.rand(1))   # CPU generator: no device sync
        d['color_jitter'] = random.uniform(0, 1)
        d['blur'] = random.uniform(0, 1) if self.blur else 0
        d['sigma'] = random.uniform(0.15, 1.15)
        d['classes'] = self._choose_classes(day_label)
        d['jitter'] = None
        if d['color_jitter'] > self.color_jitter_p:
            s_ = self.color_jitter_s
            lo = max(0.0, 1 - s_)
            d['jitter'] = [([int(v) for v in np.random.permutation(4)], random.uniform(lo, 1 + s_), random.uniform(lo, 1 + s_),
                            random.uniform(lo, 1 + s_), random.uniform(-s_, s_)) for _ in range(B)]
        if self.shift_type == 'random':
            cj = d['color_jitter']
            d['direction'] = _DIRECT[int(cj * 10) % 2][int(cj * 100) % 2]
        else:
            d['direction'] = self.shift_type
        return d

    # -- device-resident control block: what the host decided, in buffers whose addresses never change ----------------------
    def _control_block(self, dev, B, H, W):
        key = (str(dev), B, H, W)
        cb = getattr(self, '_ctl', None)
        if cb is not None and cb['key'] == key:
            return cb
        kx, ky = ops.blur_kernel_size(W), ops.blur_kernel_size(H)
        ndir = 4 if self.shift_type == 'all' else 2
        al = lambda n: (n + 3) // 4 * 4  # noqa: E731  (16-byte aligned sections)
        sizes = [('classes', 2 * B * self._kmax()), ('jitter', 8 * B), ('taps_x', kx), ('taps_y', ky), ('dirs', 2 * ndir), ('flags', 4)]
        off, o = {}, 0
        for name, n in sizes:
            off[name] = (o, n)
            o += al(n)
        # ring of pinned staging buffers: the host may run several iterations ahead of the GPU (graph replay), so a buffer is
        # rewritten only after the copy that read it has executed (event per slot)
        nring = 4 if dev.type == 'cuda' else 1
        hosts = [torch.zeros(o, dtype=torch.int32).pin_memory() if dev.type == 'cuda' else torch.zeros(o, dtype=torch.int32)
                 for _ in range(nring)]
        devbuf = torch.zeros(o, dtype=torch.int32, device=dev)

        def views(buf):
            v = {n: buf[a:a + ln] for n, (a, ln) in off.items()}
            return dict(classes=v['classes'].view(torch.int64).view(B, self._kmax()), jitter=v['jitter'].view(torch.float32).view(B, 8),
                        taps_x=v['taps_x'].view(torch.float32), taps_y=v['taps_y'].view(torch.float32),
                        dirs=v['dirs'].view(ndir, 2), jitter_on=v['flags'][0:1], blur_on=v['flags'][1:2])
        self._ctl = dict(key=key, hosts=hosts, hviews=[views(h) for h in hosts], events=[None] * nring, slot=0, dev=devbuf,
                         d=views(devbuf), kx=kx, ky=ky)
        return self._ctl

    def _stage(self, cb, draws):
        """host decisions -> the device control block: ONE asynchronous copy per iteration"""
        slot = cb['slot']
        cb['slot'] = (slot + 1) % len(cb['hosts'])
        if cb['events'][slot] is not None:
            cb['events'][slot].synchronize()
        h = cb['hviews'][slot]
        h['classes'].copy_(draws['classes'])
        on = draws['jitter'] is not None
        h['jitter_on'][0] = int(on)
        if on:
            h['jitter'].copy_(ops.jitter_params(draws['jitter']))
        blur_on = draws['blur'] > 0.5
        h['blur_on'][0] = int(blur_on)
        if blur_on:
            h['taps_x'].copy_(ops.gaussian_taps(cb['kx'], draws['sigma']))
            h['taps_y'].copy_(ops.gaussian_taps(cb['ky'], draws['sigma']))
        h['dirs'].copy_(torch.tensor(ops.isr_dirs(draws['direction'], self.isr_parms['shift_pixel']), dtype=torch.int32))
        cb['dev'].copy_(cb['hosts'][slot], non_blocking=True)
        if cb['dev'].is_cuda:
            ev = cb['events'][slot] or torch.cuda.Event()
            ev.record()
            cb['events'][slot] = ev

    def _cfg_student(self, use_events):
        tt = self.train_type
        if tt == 'cs2dsec_image+events_together':
            if self.fuse_both_ice_and_e:
                return dict(self.forward_cfg, fusion_all=True)
            if self.isr_another_fusion and not use_events:
                return dict(self.forward_cfg, fusion_isr=True)
        elif tt == 'cs2dsec_image+events':
            if self.isr_no_fusion and not use_events:
                return dict(self.forward_cfg, no_fusion=True)
            if self.isr_another_fusion and not use_events:
                return dict(self.forward_cfg, fusion_isr=True)
        return self.forward_cfg

    # -- the device work of one iteration: no host reads, no host-dependent launch shapes (capturable: hipGraph segments) ----------
    def _iteration(self, src, tgt, ctl, use_events, teacher_second, direction):
        """dacs.py:397-860 minus the host decisions (`_draw`), the EMA update and the optimizer step.  `ctl` = device views of
        the control block; `teacher_second` = the teacher's second input (events or ISR, already chosen); `use_events` /
        `direction` only select code paths that are fixed per configuration (student inputs of 'cs2dsec_image+events')."""
        tt = self.train_type
        opt = getattr(self, '_opt', None)
        ext_wait = (lambda: getattr(opt, '_update_stream', None)) if (opt is not None and getattr(opt, 'overlap', False)) else None
        day_image, day_isr, day_label = src['image'], src['img_self_res'], src['label']
        day_events = night_events = None
        if tt == 'cs2dz_image+raw-isr':
            night_image = tgt['warp_image'] if 'warp_image' in tgt else tgt['image']
        else:
            night_image, night_events = tgt['warp_image'], tgt['events_vg']
        B, _, H, W = day_image.shape
        dev = day_image.device
        log_vars = {}
        if not self.ema_model.training or not getattr(self, '_teacher_mode_set', False):
            self.ema_model.train()          # BatchNorm keeps batch statistics (and updates its running stats) ...
            set_stochastic(self.ema_model, False)  # ... but DropPath / Dropout2d are off in the teacher (dacs.py:458-462)
            self._teacher_mode_set = True
        student, teacher = self.get_model(), self.get_ema_model()
        cfg_s = self._cfg_student(use_events)
        one = rt.ones1(dev)
        lab = day_label.view(B, H, W)
        classes = ctl['classes']
        # Schedule.  The reference runs source step, teacher, mixing, mixed step one after the other (dacs.py:489-860), but the
        # only data dependencies are: mixing needs the teacher's pseudo-labels and the generator's events; the student's
        # gradients are the sum over both steps; its BatchNorm running statistics see the source step before the mixed step.
        # Round-5 schedule (lane 'T' off): teacher -> mixing, the generator queued on the side lane behind the teacher's event encoder,
        # then the student ONCE over source + mixed samples.  Default since round 6 (lanes 'T', 'Tenc', 'wq' on, uda.GRAPH_LANES): the
        # EARLY-STUDENT schedule below.  With the lanes switched off (eager launches) the same code simply runs in program order.

        def teacher_labels():
            """teacher forward -> pseudo-labels -> ClassMix'd labels / weights (dacs.py:653-711, :716-771 for the targets)"""
            if tt != 'cs2dz_image+raw-isr' and self.fuse_both_ice_and_e:
                ema = teacher.encode_decode_lowres(night_image, night_events, tgt['warp_img_self_res'], dict(self.forward_cfg, fusion_all=True))
            elif tt != 'cs2dz_image+raw-isr' and self.isr_another_fusion and not use_events:
                ema = teacher.encode_decode_lowres(night_image, teacher_second, test_cfg=dict(self.forward_cfg, fusion_isr=True))
            else:
                ema = teacher.encode_decode_lowres(night_image, teacher_second, test_cfg=self.forward_cfg)
            pseudo_label, _, count = ops.pseudo_label(ema['fusion_output'], H, W, self.pseudo_threshold, want_prob=False)
            pseudo_weight = ops.pseudo_weight(count, B, H, W, self.psweight_ignore_top, self.psweight_ignore_bottom)
            gt_pixel_weight = torch.ones(B, H, W, dtype=torch.float32, device=dev)
            mixed_lbl = ops.class_mix_label(lab, pseudo_label, lab, classes).view(B, 1, H, W)
            mixed_weight = ops.class_mix(gt_pixel_weight.view(B, 1, H, W), pseudo_weight.view(B, 1, H, W), lab, classes).view(B, H, W)
            return ema, pseudo_label, count, mixed_lbl, mixed_weight

        def mixed_inputs():
            """ClassMix + strong augmentation + ISR of the mixed image (dacs.py:716-771; dacs_transforms.py:64-98, kornia
            semantics): the on/off gates and the per-sample parameters are read from the control block by the kernels"""
            mixed_img = ops.class_mix(day_image, night_image, lab, classes)
            if self.color_jitter_p < 1.0:
                ops.color_jitter_(mixed_img, ctl['jitter'], ctl['jitter_on'])
            if self.blur:
                ops.gaussian_blur_(mixed_img, ctl['taps_x'], ctl['taps_y'], ctl['blur_on'])
            gray = ops.isr_gray(mixed_img)
            mixed_isr = ops.isr_from_gray(gray, self.isr_parms['val_range'], self.isr_parms['_threshold'],
                                          self.isr_parms['_clip_range'], self.isr_parms['shift_pixel'], direction, dirs_dev=ctl['dirs'])
            return mixed_img, mixed_isr

        # EARLY-STUDENT schedule (lane 'T' enabled): the student's forward pass needs the MIXED INPUTS, which depend on the source labels,
        # the class draw and the generator only -- the teacher's pseudo-labels enter at the loss (mixed label / weight).  So the teacher
        # runs on its own lane from the start of the iteration (its two encoders one after the other), the mixed image / ISR are built
        # on this lane at once, the generator on the side lane, and the student's encoders start as soon as the generator is done; lane
        # 'T' is joined in front of the decode head's loss (train_fwd_passes before_head).  The reference's order (teacher before the
        # mixed step, dacs.py:653-860) only matters through these data dependencies.
        early = (getattr(self, 'early_student', True) and rt.lane_enabled('T') and getattr(self, 'fused_student_passes', True)
                 and hasattr(student, 'train_fwd_passes'))
        # OVERLAPPED UPDATE (FlatAdamW.overlap): AdamW, the gradient clear and the EMA update of the step boundary run on the optimizer's
        # own stream; the part of the iteration that reads no trainable weight -- the mixed inputs and the frozen generator -- is enqueued
        # FIRST and runs underneath them, then this lane waits for that stream (`wait_external`), refreshes the weight copies and goes on.
        pre = early and ext_wait is not None and tt != 'cs2dz_image+raw-isr' and self.cyclegan_itrd2en is not None
        day_events_pre = None
        if pre:
            mixed_img, mixed_isr = mixed_inputs()
            with rt.lane('enc', src['img_time_res'], independent=True):
                day_events_pre = self.cyclegan_itrd2en.forward_mean3(src['img_time_res'])
        if ext_wait is not None:
            rt.wait_external(ext_wait)
        rt.refresh(force=True)   # all re-laid-out weight copies follow this iteration's masters (first node without the overlapped update)
        if early:
            with rt.lane('T', night_image, teacher_second, lab, classes, *[v for v in tgt.values() if isinstance(v, torch.Tensor)]):
                ema, pseudo_label, count, mixed_lbl, mixed_weight = teacher_labels()
            if not pre:
                mixed_img, mixed_isr = mixed_inputs()
        else:
            # ---- teacher pseudo-labels (dacs.py:653-711) -------------------------------------------------------------------------
            with rt.lane('T', night_image, teacher_second, *[v for v in tgt.values() if isinstance(v, torch.Tensor)]):
                ema, pseudo_label, count, mixed_lbl, mixed_weight = teacher_labels()
                mixed_img, mixed_isr = mixed_inputs()

        # ---- Image Motion-Extractor (dacs.py:400-404), frozen, no grad ------------------------------------------------------------
        if tt != 'cs2dz_image+raw-isr':
            if self.cyclegan_itrd2en is not None:
                # frozen weights, static input: queued behind the teacher's event encoder on the side lane, next to the
                # teacher's fusion / decoder / pseudo-label / mixing work on this one
                # (EARLY-STUDENT: letting the image encoder start before the generator is done -- generator, event mix and event encoder on
                # the side lane without the join -- measured the same, 52.22 against 52.25 ms over three alternating runs: not kept)
                if day_events_pre is not None:
                    day_events = day_events_pre
                else:
                    with rt.lane('enc', src['img_time_res'], independent=True):
                        day_events = self.cyclegan_itrd2en.forward_mean3(src['img_time_res'])
                rt.join_lanes('enc')
            else:
                day_events = src['img_time_res']
        mixed_events = None
        if day_events is not None:
            if early:
                mixed_events = ops.class_mix(day_events, night_events, lab, classes)
            else:
                with rt.lane('T', day_events):
                    mixed_events = ops.class_mix(day_events, night_events, lab, classes)

        # ---- student inputs: source (dacs.py:489-523) and mixed (dacs.py:820-860) ---------------------------------------------------
        if tt == 'cs2dz_image+raw-isr':
            in_src = {'image': day_image, 'events': day_isr}
            in_mix = {'image': mixed_img, 'events': mixed_isr}
        elif tt == 'cs2dsec_image+events_together':
            in_src = {'image': day_image, 'events': day_events, 'img_self_res': day_isr}
            in_mix = {'image': mixed_img, 'events': mixed_events, 'img_self_res': mixed_isr}
        else:
            in_src = {'image': day_image, 'events': day_events if use_events else day_isr}
            in_mix = {'image': mixed_img, 'events': mixed_events if use_events else mixed_isr}
        hook = getattr(self, 'final_pass_grad_hook', None)
        if hook is not None and dev.type == 'cuda' and torch.cuda.is_current_stream_capturing() and rt._conc['seg'] is None:
            hook = None   # one monolithic capture cannot call out; the segmented capture records the hook as a host step
        prev_hook = rt.grad_ready_hook

        if (getattr(self, 'fused_student_passes', True) and hasattr(student, 'train_fwd_passes')
                and student._joint_ok(in_src['image'], in_src['events'], cfg_s)):
            # ONE student pass over the source and the mixed samples: both steps run the same weights (the optimizer steps after
            # both, dacs.py:523 / :860 only accumulate gradients), so the 2B samples travel as one batch -- half the launches,
            # twice the rows per GEMM.  What is per step in the reference stays per step: BatchNorm batch statistics and the order
            # of the running-statistic updates (source first), the loss normalisation, DropPath / Dropout2d draws per sample.
            if not early:
                rt.join_lanes('T')
            ((l_src, d_src), (l_mix, d_mix)), saved = student.train_fwd_passes(
                [(in_src, day_label, None), (in_mix, mixed_lbl, mixed_weight)], cfg_s,
                before_head=(lambda: rt.join_lanes('T')) if early else None)
            log_vars['decode.loss_seg'], log_vars['decode.acc_seg'] = l_src, d_src['acc_seg']
            log_vars['mix.decode.loss_seg'], log_vars['mix.decode.acc_seg'] = l_mix, d_mix['acc_seg']
            log_vars['loss'] = l_mix   # _parse_losses of the mixed step overwrites 'loss' (dacs.py:851-857)
            # the only backward pass of the iteration: gradients it reports final are final for the step, so a data-parallel
            # driver may start their all-reduce underneath the rest of the pass (runtime.grad_ready_hook)
            if hook is not None:
                rt.grad_ready_hook = hook
            try:
                student.train_bwd(saved, one)
            finally:
                rt.grad_ready_hook = prev_hook
            del saved
        else:
            # two passes (routes the joint pass does not cover): the mixed forward runs after the source forward (BatchNorm
            # running statistics), next to the source backward; the mixed backward follows the join (gradients accumulate)
            loss, (losses, _, _), saved_src = student.train_fwd(in_src, day_label, None, cfg_s)
            log_vars['decode.loss_seg'], log_vars['decode.acc_seg'], log_vars['loss'] = loss, losses['acc_seg'], loss
            with rt.lane('T'):
                loss, (losses, _, _), saved_mix = student.train_fwd(in_mix, mixed_lbl, mixed_weight, cfg_s)
            log_vars['mix.decode.loss_seg'], log_vars['mix.decode.acc_seg'] = loss, losses['acc_seg']
            log_vars['loss'] = loss   # _parse_losses of the mixed step overwrites 'loss' (dacs.py:851-857)
            student.train_bwd(saved_src, one)
            del saved_src
            rt.join_lanes('T')
            if hook is not None:   # the second (last) backward pass: see above
                rt.grad_ready_hook = hook
            try:
                student.train_bwd(saved_mix, one)
            finally:
                rt.grad_ready_hook = prev_hook
            del saved_mix
        extras = dict(mixed_img=mixed_img, mixed_lbl=mixed_lbl, mixed_isr=mixed_isr, pseudo_weight=mixed_weight,
                      pseudo_label=pseudo_label, classes=classes, mixed_events=mixed_events, day_events=day_events,
                      teacher_logits=ema, pseudo_count=count)
        return log_vars, extras

    # -- hipGraph replay of the iteration ------------------------------------------------------------------------------------------
    def enable_graph(self, warmup_iters=2):
        """Capture `_iteration` after `warmup_iters` eager iterations -- as a chain of linear hipGraph segments replayed on the
        concurrency lanes' streams (runtime.SegmentedCapture) -- and replay it from then on: at the reference's 2+2 samples per
        GPU the eager step is bound by the host's launch rate, not by the GPU.  Only the EMA update, the control-block copy and
        the optimizer step stay outside."""
        self._graph_warmup = warmup_iters
        self._graph, self._graphs = None, {}

    def disable_graph(self):
        self._graph_warmup = None
        self._graph, self._graphs = None, {}

    def _capture(self, src, tgt, cb, use_events_struct, direction):
        dev = src['image'].device
        st_src = {k: v.clone() for k, v in src.items() if isinstance(v, torch.Tensor)}
        st_tgt = {k: v.clone() for k, v in tgt.items() if isinstance(v, torch.Tensor)}
        second = torch.empty_like(st_src['image'])
        ws_lanes = ('main', 'main/enc', 'main/T', 'main/T/enc')   # (the per-lane persistent workspaces exist before the capture starts)
        ops.ln_ws_prealloc(dev, ws_lanes)
        ops.zero_ws_prealloc(dev, ws_lanes)
        ops.bn_ws_prealloc(dev, ws_lanes)
        torch.cuda.synchronize(dev)
        rt.refresh(force=True)   # every copy exists and is current before the capture starts
        lanes = getattr(self, 'graph_lane_set', None)
        if lanes is None and os.environ.get('CMDA_LANES'):   # tuning: comma-separated lane set (runtime.set_concurrency)
            lanes = set(os.environ['CMDA_LANES'].split(','))
        if lanes is None:
            # the default set of the captured iteration (round 6, same-box A/B 55.4-55.9 -> 52.1-52.3 ms): 'enc' the two encoders side by
            # side; 'T' + 'Tenc' the EARLY-STUDENT schedule -- the teacher on two queues of its own from the start of the iteration, joined
            # in front of the decode head's loss; 'wq' the encoders' grouped weight gradients on those queues once the teacher is done
            lanes = set(GRAPH_LANES)
        seg = rt.SegmentedCapture(dev)
        g = seg
        if os.environ.get('CMDA_LANE_QUEUES', '1') != '0':   # every lane stream on a hardware queue of its own (runtime.prepare_lane_streams)
            en = lanes if lanes is not None else rt._conc['enabled']
            rt.prepare_lane_streams(dev, seg.main, ['main/' + n for n in sorted(en) if n in ('enc', 'T', 'hw')] +
                                    (['main/T/enc'] if 'T' in en and 'Tenc' in en else []))
            # (running an overlapped optimizer update on the TEACHER lane's stream -- idle at the step boundary, where the update's own stream
            # shares a hardware queue with the generator's lane -- let the generator start beside AdamW instead of behind it, and measured
            # 51.5-51.9 against 51.0-51.5 ms, three alternating runs: the generator then takes 3.7 instead of 2.3 ms.  Not kept.)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        rt.set_concurrency(bool(getattr(self, 'graph_lanes', True)), lanes, seg=seg)
        try:
            seg.begin(seg.main)
            out = self._iteration(st_src, st_tgt, cb['d'], use_events_struct, second, direction)
            rt.join_lanes()
            seg.end()
        finally:
            rt.set_concurrency(False)
        torch.cuda.synchronize(dev)
        key = (use_events_struct, direction, tuple(src['image'].shape), tuple(tgt_shape(tgt)))
        # one captured iteration per launch structure / batch shape, all kept: configurations whose structure follows a
        # per-iteration random choice ('cs2dsec_image+events', isr_another_fusion) alternate between two of them
        self._graphs[key] = self._graph = dict(graph=g, src=st_src, tgt=st_tgt, second=second, out=out, key=key)

    def forward_train(self, **kwargs):
        src, tgt = kwargs['source'], kwargs['target']
        tt = self.train_type
        day_label = src['label']
        B, _, H, W = src['image'].shape
        dev = src['image'].device
        draws = getattr(self, 'inject_draws', None) or self._draw(day_label, H, W)
        self.last_draws = draws
        self.forward_cfg['isr_events_fusion_choice'] = draws['choice']
        use_events = tt != 'cs2dz_image+raw-isr' and draws['choice'] > self.random_choice_thres
        cb = self._control_block(dev, B, H, W)
        self._stage(cb, draws)
        opt = getattr(self, '_opt', None)

        def boundary():
            """the step boundary's device work that precedes this iteration: the (possibly postponed) optimizer update, then the EMA
            teacher update -- with an overlapped update both on the optimizer's stream"""
            if opt is not None:
                opt.flush()
            with (opt._on_update_stream() if opt is not None else _null_ctx()):
                if self.local_iter == 0:
                    self._init_ema_weights()
                if self.local_iter > 0:
                    self._update_ema(self.local_iter)
        if tt == 'cs2dz_image+raw-isr':
            second = tgt['warp_img_self_res'] if 'warp_image' in tgt else tgt['night_isr']
        else:
            second = tgt['events_vg'] if (use_events or self.isr_no_fusion) else tgt['warp_img_self_res']
        # launch-shape-relevant part of the draws: only 'cs2dsec_image+events' / isr_another_fusion route differently by choice,
        # and only shift_type 'all' changes the number of ISR directions (the direction itself lives in the control block)
        struct_events = use_events if (tt == 'cs2dsec_image+events' or self.isr_another_fusion) else True
        ndir_key = 'all' if self.shift_type == 'all' else 'rightdown'
        graph_on = (getattr(self, '_graph_warmup', None) is not None and dev.type == 'cuda'
                    and self.local_iter >= self._graph_warmup)
        if graph_on:
            key = (struct_events, ndir_key, tuple(src['image'].shape), tuple(tgt_shape(tgt)))
            if not hasattr(self, '_graphs'):
                self._graphs = {}
            if key not in self._graphs:
                # (first replay, another launch structure, or another batch shape: the captured launches are shape-specific)
                boundary()
                boundary = lambda: None   # noqa: E731  (done: the capture synchronises the device)
                self._capture(src, tgt, cb, struct_events, ndir_key)
            G = self._graph = self._graphs[key]
            # the inputs are staged FIRST, the step boundary's HBM-bound passes (AdamW, EMA) are enqueued behind them on their own stream
            for k, v in G['src'].items():
                if v.data_ptr() != src[k].data_ptr():
                    v.copy_(src[k])
            for k, v in G['tgt'].items():
                if v.data_ptr() != tgt[k].data_ptr():
                    v.copy_(tgt[k])
            G['second'].copy_(second)
            boundary()
            G['graph'].replay()
            log_vars, extras = G['out']
        else:
            boundary()
            if opt is not None:
                opt.synchronize()   # eager launches: everything behind the (possibly overlapped) update
            log_vars, extras = self._iteration(src, tgt, cb['d'], struct_events, second, ndir_key)
        self.local_iter += 1
        self.last_mix = extras
        return dict(log_vars)
