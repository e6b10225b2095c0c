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
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import runtime as rt
from .cyclegan import define_G
from .registry import UDA, build_segmentor
from .segmentors import add_prefix, parse_losses

_DIRECT = [['leftdown', 'leftup'], ['rightdown', 'rightup']]


def set_stochastic(model, flag):
    """DropPath / Dropout2d on or off without touching BatchNorm's train mode (dacs.py:458-462 for the teacher)."""
    for m in model.modules():
        if hasattr(m, 'drop_path_rate') or hasattr(m, 'dropout_ratio'):
            m.stochastic = flag


def log_vars_to_float(log_vars):
    return {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in log_vars.items()}


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
        with torch.no_grad():
            for n, p in self.ema_model.named_parameters():
                off = offsets[n]
                ema_flat[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = ema_flat[off:off + p.numel()].view(p.shape)
        self._flat = (flat_p, ema_flat)
        # bf16 compute copies of the teacher in one flat mirror too, refreshed by ONE cast after each EMA update (without
        # it every teacher Linear weight is cast on its own: ~1100 extra launches per iteration)
        self._ema_bf16 = None
        if rt.compute_dtype() == torch.bfloat16 and ema_flat.is_cuda:
            self._ema_bf16 = torch.empty(ema_flat.numel(), dtype=torch.bfloat16, device=ema_flat.device)
            for n, p in self.ema_model.named_parameters():
                off = offsets[n]
                p._cmda_bf16 = self._ema_bf16[off:off + p.numel()].view(p.shape)
            self._sync_ema_bf16()

    def _sync_ema_bf16(self):
        if getattr(self, '_ema_bf16', None) is not None:
            n = self._flat[1].numel()
            ops.permute4(self._flat[1], self._ema_bf16, (n, 1, 1, 1), (0, 1, 2, 3))

    def _init_ema_weights(self):
        if self._flat is not None:
            ops.ema_update(self._flat[1], self._flat[0], 0.0)
            self._sync_ema_bf16()
        else:
            for e, p in zip(self.ema_model.parameters(), self.model.parameters()):
                ops.ema_update(e.data.view(-1), p.data.view(-1), 0.0)
        rt.invalidate()

    def _update_ema(self, it):
        alpha_teacher = min(1 - 1 / (it + 1), self.alpha)
        if self._flat is not None:
            ops.ema_update(self._flat[1], self._flat[0], alpha_teacher)
            self._sync_ema_bf16()
        else:
            for e, p in zip(self.ema_model.parameters(), self.model.parameters()):
                ops.ema_update(e.data.view(-1), p.data.view(-1), alpha_teacher)
        rt.invalidate()

    # -- one UDA iteration ---------------------------------------------------------------------------------------------------
    def train_step(self, data_batch, optimizer, **kwargs):
        optimizer.zero_grad()
        log_vars = self(**data_batch)
        optimizer.step()
        log_vars.pop('loss', None)
        src = data_batch['source']
        n = src['image'].shape[0] if 'image' in src else data_batch['target']['warp_image'].shape[0]
        return dict(log_vars=log_vars, num_samples=n)

    def forward(self, **kwargs):
        return self.forward_train(**kwargs)

    def _choose_classes(self, labels):
        """get_class_masks (dacs_transforms.py:101-112): classes = unique over the WHOLE batch, ceil(n/2) drawn per
        sample with np.random.choice.  One small device->host read (<= 20 class ids), as in the reference."""
        classes = torch.unique(labels).cpu()
        n = classes.shape[0]
        k = int((n + n % 2) / 2)
        out = torch.full((labels.shape[0], max(k, 1)), -1, dtype=torch.int64)
        for i in range(labels.shape[0]):
            pick = np.random.choice(n, k, replace=False)
            out[i, :k] = classes[torch.as_tensor(pick).long()]
        return out.to(labels.device)

    def forward_train(self, **kwargs):
        src, tgt = kwargs['source'], kwargs['target']
        tt = self.train_type
        day_events = night_events = None
        if tt == 'cs2dz_image+raw-isr':
            day_image, day_isr, day_label = src['image'], src['img_self_res'], src['label']
            if 'warp_image' in tgt:
                night_image, night_isr = tgt['warp_image'], tgt['warp_img_self_res']
            else:
                night_image, night_isr = tgt['image'], tgt['night_isr']
        else:
            day_image, day_isr, day_label = src['image'], src['img_self_res'], src['label']
            if self.cyclegan_itrd2en is not None:
                itr = src['img_time_res'].mean(dim=1, keepdim=True)
                day_events = self.cyclegan_itrd2en(itr).repeat(1, 3, 1, 1)
            else:
                day_events = src['img_time_res']
            night_image, night_events, night_isr = tgt['warp_image'], tgt['events_vg'], tgt['warp_img_self_res']
            if self.without_events:
                self.forward_cfg['isr_events_fusion_choice'] = -1
            elif self.without_isd:
                self.forward_cfg['isr_events_fusion_choice'] = 2
            else:
                self.forward_cfg['isr_events_fusion_choice'] = torch.rand(1).detach()  # CPU tensor: no device sync
        use_events = tt != 'cs2dz_image+raw-isr' and bool(self.forward_cfg['isr_events_fusion_choice'] > self.random_choice_thres)
        B, _, H, W = day_image.shape
        log_vars = {}

        if self.local_iter == 0:
            self._init_ema_weights()
        if self.local_iter > 0:
            self._update_ema(self.local_iter)
        strong = {'color_jitter': random.uniform(0, 1), 'blur': random.uniform(0, 1) if self.blur else 0,
                  'sigma': random.uniform(0.15, 1.15)}
        self.ema_model.train()          # BatchNorm keeps batch statistics (and updates its running stats) ...
        set_stochastic(self.ema_model, False)  # ... but DropPath / Dropout2d are off in the teacher

        # ---- source ------------------------------------------------------------------------------------------------
        student = self.get_model()
        if tt == 'cs2dz_image+raw-isr':
            inputs, cfg_s = {'image': day_image, 'events': day_isr}, self.forward_cfg
        elif tt == 'cs2dsec_image+events_together':
            inputs = {'image': day_image, 'events': day_events, 'img_self_res': day_isr}
            if self.fuse_both_ice_and_e:
                cfg_s = dict(self.forward_cfg, fusion_all=True)
            elif self.isr_another_fusion and not use_events:
                cfg_s = dict(self.forward_cfg, fusion_isr=True)
            else:
                cfg_s = self.forward_cfg
        else:
            inputs = {'image': day_image, 'events': day_events if use_events else day_isr}
            if self.isr_no_fusion and not use_events:
                cfg_s = dict(self.forward_cfg, no_fusion=True)
            elif self.isr_another_fusion and not use_events:
                cfg_s = dict(self.forward_cfg, fusion_isr=True)
            else:
                cfg_s = self.forward_cfg
        source_losses, _ = student.forward_train(inputs, day_label, return_feat=True, cfg=cfg_s)
        source_losses.pop('features')
        source_loss, clean_log = parse_losses(source_losses)
        log_vars.update(clean_log)
        source_loss.backward()

        # ---- teacher pseudo-labels -------------------------------------------------------------------------------------
        teacher = self.get_ema_model()
        if tt == 'cs2dz_image+raw-isr':
            ema = teacher.encode_decode_lowres(night_image, night_isr, test_cfg=self.forward_cfg)
        else:
            if self.fuse_both_ice_and_e:
                ema = teacher.encode_decode_lowres(night_image, night_events, night_isr, dict(self.forward_cfg, fusion_all=True))
            elif self.isr_another_fusion and not use_events:
                ema = teacher.encode_decode_lowres(night_image, night_isr, test_cfg=dict(self.forward_cfg, fusion_isr=True))
            elif self.isr_no_fusion:
                ema = teacher.encode_decode_lowres(night_image, night_events, test_cfg=self.forward_cfg)
            else:
                ema = teacher.encode_decode_lowres(night_image, night_events if use_events else night_isr, test_cfg=self.forward_cfg)
        pseudo_label, _, count = ops.pseudo_label(ema['fusion_output'], H, W, self.pseudo_threshold, want_prob=False)
        pseudo_weight = ops.pseudo_weight(count, B, H, W, self.psweight_ignore_top, self.psweight_ignore_bottom)
        gt_pixel_weight = torch.ones(B, H, W, dtype=torch.float32, device=day_image.device)

        # ---- ClassMix (+ ISR of the mixed image) ---------------------------------------------------------------------------
        lab = day_label.view(B, H, W)
        classes = self._choose_classes(day_label)
        mixed_img = ops.class_mix(day_image, night_image, lab, classes)
        # strong_transform's colour jitter / Gaussian blur of the mixed image (dacs_transforms.py:64-98; kornia semantics)
        if strong['color_jitter'] > self.color_jitter_p:
            s_ = self.color_jitter_s
            order = list(np.random.permutation(4))
            ops.color_jitter_(mixed_img, order, random.uniform(max(0.0, 1 - s_), 1 + s_), random.uniform(max(0.0, 1 - s_), 1 + s_),
                              random.uniform(max(0.0, 1 - s_), 1 + s_), random.uniform(-s_, s_))
        if strong['blur'] > 0.5:
            k = int(np.floor(np.ceil(0.1 * H) - 0.5 + np.ceil(0.1 * H) % 2))
            ops.gaussian_blur_(mixed_img, k, strong['sigma'])
        mixed_events = ops.class_mix(day_events, night_events, lab, classes) if day_events is not None else None
        gray = ops.isr_gray(mixed_img)
        if self.shift_type == 'random':
            cj = strong['color_jitter']
            direction = _DIRECT[int(cj * 10) % 2][int(cj * 100) % 2]
        else:
            direction = self.shift_type
        mixed_isr = ops.isr_from_gray(gray, self.isr_parms['val_range'], self.isr_parms['_threshold'],
                                      self.isr_parms['_clip_range'], self.isr_parms['shift_pixel'], direction)
        mixed_lbl = ops.class_mix_label(lab, pseudo_label, lab, classes).view(B, 1, H, W)
        mixed_weight = ops.class_mix(gt_pixel_weight.view(B, 1, H, W), pseudo_weight.view(B, 1, H, W), lab, classes).view(B, H, W)

        # ---- mixed ---------------------------------------------------------------------------------------------------------
        if tt == 'cs2dz_image+raw-isr':
            inputs = {'image': mixed_img, 'events': mixed_isr}
        elif tt == 'cs2dsec_image+events_together':
            inputs = {'image': mixed_img, 'events': mixed_events, 'img_self_res': mixed_isr}
        else:
            inputs = {'image': mixed_img, 'events': mixed_events if use_events else mixed_isr}
        mix_losses, _ = student.forward_train(inputs, mixed_lbl, seg_weight=mixed_weight, return_feat=True, cfg=cfg_s)
        mix_losses.pop('features')
        mix_loss, mix_log = parse_losses(add_prefix(mix_losses, 'mix'))
        log_vars.update(mix_log)
        # the second (last) backward pass of the iteration: gradients reported final by this pass are final for the step, so a
        # data-parallel driver may start their all-reduce underneath the rest of the pass (runtime.grad_ready_hook)
        prev_hook = rt.grad_ready_hook
        if getattr(self, 'final_pass_grad_hook', None) is not None:
            rt.grad_ready_hook = self.final_pass_grad_hook
        try:
            mix_loss.backward()
        finally:
            rt.grad_ready_hook = prev_hook
        self.local_iter += 1
        self.last_mix = dict(mixed_img=mixed_img, mixed_lbl=mixed_lbl, mixed_isr=mixed_isr, pseudo_weight=mixed_weight,
                             pseudo_label=pseudo_label, classes=classes)
        return log_vars
