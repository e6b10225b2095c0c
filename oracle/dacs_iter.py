"""ORACLE (test infrastructure): one whole DACS iteration composed from the oracle's pieces.

Follows mmseg/models/uda/dacs.py::DACS.forward_train :357-860 for train_type 'cs2dsec_image+events_together' (and
'cs2dz_image+raw-isr') with the launcher defaults of SURVEY.md appendix A:
  :397-417  unpack the batch; Motion-Extractor G(mean_c(img_time_res)) under no_grad, x3 channels; events/ISR choice u
  :437-442  EMA teacher init (it 0) / update (it > 0)
  :446-456  strong_parameters (python `random`): colour-jitter gate, blur gate, sigma
  :489-523  source forward_train + backward
  :653-711  teacher encode_decode (BN in train mode, DropPath / Dropout off) -> softmax/max of the fusion logits,
            pseudo_weight = mean(prob >= thr) * ones, optional top / bottom rows zeroed
  :716-771  get_class_masks; per sample: strong_transform(image) = mix -> ColorJitter -> GaussianBlur
            (models/utils/dacs_transforms.py:11-35,64-98), mix of events / label / weight, ISR of the mixed image (:729-744)
  :820-860  mixed forward_train (seg_weight = mixed pseudo-weight) + backward
Used by tests/test_dacs.py (parity of the HIP step) and bench.py's cpu_baseline leg (timed on the host cores).
Every stochastic decision can be injected through `draws` so that the HIP step and this restatement see the same ones.
"""
import random

import numpy as np
import torch

from . import uda as U


def draw_jitter(s):
    """kornia ColorJitter(brightness=s, contrast=s, saturation=s, hue=s) parameter draw for ONE sample: the order of the
    four ops and the factors (brightness / contrast / saturation in [max(0,1-s), 1+s], hue in [-s, s])."""
    lo = max(0.0, 1.0 - s)
    order = [int(v) for v in np.random.permutation(4)]
    return (order, random.uniform(lo, 1 + s), random.uniform(lo, 1 + s), random.uniform(lo, 1 + s), random.uniform(-s, s))


def dacs_iteration(student, teacher, generator, source, target, *, local_iter, forward_cfg, alpha=0.999,
                   pseudo_threshold=0.968, ignore_top=0, ignore_bottom=0, isr_parms=None, shift_type='rightdown',
                   blur=True, color_jitter_s=0.2, color_jitter_p=0.2, random_choice_thres=0.5,
                   train_type='cs2dsec_image+events_together', draws=None):
    """Runs source fwd/bwd, teacher, mixing, mixed fwd/bwd on the CPU; parameter gradients accumulate in student.grad.
    Returns a dict with the log values and the intermediate tensors the parity test compares."""
    draws = dict(draws or {})
    isr_parms = isr_parms or dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
    day_image, day_isr, day_label = source['image'], source['img_self_res'], source['label']
    B, _, H, W = day_image.shape
    if train_type == 'cs2dz_image+raw-isr':
        day_events = night_events = None
        night_image = target['warp_image'] if 'warp_image' in target else target['image']
        night_isr = target['warp_img_self_res'] if 'warp_img_self_res' in target else target['night_isr']
        choice = 0.0
    else:
        if generator is not None:
            with torch.no_grad():
                day_events = generator(source['img_time_res'].mean(dim=1, keepdim=True)).repeat(1, 3, 1, 1)
        else:
            day_events = source['img_time_res']
        night_image, night_events, night_isr = target['warp_image'], target['events_vg'], target['warp_img_self_res']
        choice = draws['choice'] if 'choice' in draws else float(torch.rand(1))
    use_events = train_type != 'cs2dz_image+raw-isr' and choice > random_choice_thres

    U.update_ema(list(teacher.parameters()), list(student.parameters()), local_iter, alpha)
    cj = draws['color_jitter'] if 'color_jitter' in draws else random.uniform(0, 1)
    bl = draws['blur'] if 'blur' in draws else (random.uniform(0, 1) if blur else 0)
    sigma = draws['sigma'] if 'sigma' in draws else random.uniform(0.15, 1.15)

    # ---- source
    if train_type == 'cs2dz_image+raw-isr':
        inputs = {'image': day_image, 'events': day_isr}
    else:
        inputs = {'image': day_image, 'events': day_events, 'img_self_res': day_isr}
    l_s, _ = student.forward_train(inputs, day_label, cfg=forward_cfg)
    l_s['decode.loss_seg'].backward()

    # ---- teacher (train-mode BN; its Dropout / DropPath must already be off: build it with rates 0 or .eval() them)
    with torch.no_grad():
        second = night_isr if train_type == 'cs2dz_image+raw-isr' else (night_events if use_events else night_isr)
        out = teacher.encode_decode(night_image, second, output_features=True, test_cfg=forward_cfg)
        plabel, prob, pweight, count = U.pseudo_labels_fullres(out['fusion_output'], pseudo_threshold, ignore_top, ignore_bottom)

    # ---- mixing
    chosen = draws['classes'] if 'classes' in draws else U.choose_classes(day_label, np.random)
    jit = draws.get('jitter')
    if shift_type == 'random':
        direction = U.random_shift_direction(cj)
    else:
        direction = shift_type
    mixed_img, mixed_ev, mixed_lbl, mixed_w, mixed_isr, jitter_used = [], [], [], [], [], []
    for i in range(B):
        m = U.class_mask(day_label[i], chosen[i])
        img = U.one_mix(m, day_image[i], night_image[i])[None]
        if cj > color_jitter_p:
            prm = jit[i] if jit is not None else draw_jitter(color_jitter_s)
            jitter_used.append(prm)
            img = U.color_jitter(img, *prm)
        if bl > 0.5:
            img = U.gaussian_blur_hw(img, U.blur_kernel_size(H), U.blur_kernel_size(W), sigma)
        mixed_img.append(img)
        if day_events is not None:
            mixed_ev.append(U.one_mix(m, day_events[i], night_events[i])[None])
        mixed_lbl.append(U.one_mix(m, day_label[i][0], plabel[i])[None])
        mixed_w.append(U.one_mix(m, torch.ones(H, W), pweight[i]))
        mixed_isr.append(U.mixed_image_to_isr(img, isr_parms['shift_pixel'], isr_parms['val_range'], isr_parms['_threshold'],
                                              isr_parms['_clip_range'], direction))
    mixed_img, mixed_lbl = torch.cat(mixed_img), torch.cat(mixed_lbl)
    mixed_w, mixed_isr = torch.cat(mixed_w), torch.cat(mixed_isr)
    mixed_ev = torch.cat(mixed_ev) if mixed_ev else None
    if train_type == 'cs2dz_image+raw-isr':
        inputs = {'image': mixed_img, 'events': mixed_isr}
    else:
        inputs = {'image': mixed_img, 'events': mixed_ev, 'img_self_res': mixed_isr}
    l_m, _ = student.forward_train(inputs, mixed_lbl, seg_weight=mixed_w, cfg=forward_cfg)
    l_m['decode.loss_seg'].backward()
    return {'decode.loss_seg': l_s['decode.loss_seg'].detach(), 'decode.acc_seg': l_s['decode.acc_seg'],
            'mix.decode.loss_seg': l_m['decode.loss_seg'].detach(), 'mix.decode.acc_seg': l_m['decode.acc_seg'],
            'day_events': day_events, 'pseudo_label': plabel, 'pseudo_prob': prob, 'pseudo_count': count,
            'mixed_img': mixed_img, 'mixed_events': mixed_ev, 'mixed_lbl': mixed_lbl, 'mixed_weight': mixed_w,
            'mixed_isr': mixed_isr, 'classes': chosen, 'use_events': use_events, 'teacher_logits': out,
            'draws': dict(choice=choice, color_jitter=cj, blur=bl, sigma=sigma, classes=chosen, jitter=jitter_used or None)}
