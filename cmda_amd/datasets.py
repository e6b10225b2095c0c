"""Dataset registry keys of configs/fusion/* -- `UDADataset`, `CityscapesICDataset`, `DSECDataset`, `DarkZurichICDataset` -- and
`build_dataset` / `build_dataloader` (mmseg/datasets/builder.py:26-27,66-177).

The reference's datasets decode PNG / HDF5 files on CPU workers (out of scope: SURVEY.md section 2 rows 18-23; neither the files nor
h5py / torchvision exist here).  What IS part of the drop-in boundary is (a) that the registry keys resolve with the reference's
constructor arguments, (b) the batch-dict schema the training step consumes (uda_dataset.py:37-143: {'source': {...}, 'target':
{...}} with the per-dataset keys and shapes), and (c) the arithmetic between the raw frame / raw events and those tensors (SURVEY
8 row f3).  So each dataset here owns a SYNTHETIC RAW STREAM -- seeded uint8 frames, label maps and event lists of the real
geometries (Cityscapes 1024x2048, DSEC 480x640 + ~500 k events, Dark Zurich 1080x1920) held on the device -- and turns it into the
reference's sample dict with the on-device pipeline of cmda_amd/pipeline.py (PIL-exact resize / crop / flip, real-time ISR, event
rectification + voxel grid + events_norm, time residual), drawing the loaders' random crop / flip decisions from Python's `random`
exactly where the reference draws them.  Batched (`get_batch`) and per-sample (`__getitem__`) access return the same tensors.
"""
import json
import os
import random

import numpy as np
import torch

from . import ops
from . import pipeline as pl
from .registry import DATASETS, PIPELINES, build_from_cfg  # noqa: F401

CLASSES = ('road', 'sidewalk', 'building', 'wall', 'fence', 'pole', 'traffic light', 'traffic sign', 'vegetation', 'terrain',
           'sky', 'person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle')
PALETTE = [[128, 64, 128], [244, 35, 232], [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153], [250, 170, 30],
           [220, 220, 0], [107, 142, 35], [152, 251, 152], [70, 130, 180], [220, 20, 60], [255, 0, 0], [0, 0, 142], [0, 0, 70],
           [0, 60, 100], [0, 80, 100], [0, 0, 230], [119, 11, 32]]
_DIRECT = [['leftdown', 'leftup'], ['rightdown', 'rightup']]
_DEFAULT_ISR = {'val_range': (1, 10 ** 2), '_threshold': 0.04, '_clip_range': 0.2, 'shift_pixel': 3}   # dsec.py:180, cityscapes_ic.py


def _device(device):
    if device is not None:
        return torch.device(device)
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def _stack_dev(tensors, dev):
    """batch of host tensors -> one device tensor: every sample is copied to the device on its own and stacked THERE (stacking
    2048x1024 uint8 frames on the host first cost 9 ms per stack -- 46 of the 49 ms a 2 + 2 batch spent in the loader)"""
    return torch.stack([t.to(dev, non_blocking=True) for t in tensors])


def _blocky_u8(g, h, w, c, cell=16):
    """seeded synthetic frame: piecewise-constant regions + noise (enough structure for ISR / time residual to be non-trivial)"""
    base = torch.rand((h // cell + 1, w // cell + 1, c), generator=g).repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:h, :w]
    return ((base * 0.8 + 0.2 * torch.rand((h, w, c), generator=g)) * 255).to(torch.uint8)


def _blocky_labels(g, h, w, cell=32, ignore=0.05):
    lab = torch.randint(0, 19, (h // cell + 1, w // cell + 1), generator=g).repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:h, :w]
    lab = lab.clone()
    lab[torch.rand((h, w), generator=g) < ignore] = 255
    return lab


_WARNED = set()


def _warn_synthetic(cls, path):
    """The registry keys are the reference's, the data is NOT: a config that points at real files must not train on seeded noise
    silently.  One loud warning per dataset class; CMDA_STRICT_DATA=1 turns it into an error."""
    if not path:
        return
    msg = (f'{cls.__name__}: path {path!r} was given, but cmda_amd has no file-backed reader (PNG / HDF5 decoding is outside the '
           f'accelerated hot path, SURVEY.md section 2): this dataset produces SEEDED SYNTHETIC frames / labels / events of the real '
           f'geometry.  Pass dataset_path="" to acknowledge, or feed real tensors to DACS.train_step directly.')
    if os.environ.get('CMDA_STRICT_DATA') == '1':
        raise FileNotFoundError(msg)
    if cls.__name__ not in _WARNED:
        _WARNED.add(cls.__name__)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class _SyntheticBase:
    CLASSES, PALETTE, ignore_index = CLASSES, PALETTE, 255

    def _gen(self, idx):
        return torch.Generator().manual_seed(self.seed * 1000003 + int(idx))

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        batch = self.get_batch([idx])
        return {k: (v[0] if isinstance(v, torch.Tensor) else v[0] if isinstance(v, list) else v) for k, v in batch.items()}


@DATASETS.register_module()
class CityscapesICDataset(_SyntheticBase):
    """mmseg/datasets/cityscapes_ic.py:23-272 (source domain): outputs ⊆ {'image', 'label', 'img_time_res', 'img_self_res'}.
    resize 2048x1024 -> image_resize_size -> random crop image_crop_size -> random flip; 'img_time_res' = the log-intensity change
    against the previous sequence frame (create_cityscapes_image_change.py:16-35, computed on the fly), 'img_self_res' = ISR of
    the cropped image (isr_parms / shift_type as in the reference)."""

    def __init__(self, dataset_path='', image_resize_size=(1024, 512), image_crop_size=(512, 512), image_change_range=1,
                 classes=CLASSES, palette=PALETTE, return_GI_or_IC='image_change', isr_shift_pixel=4, enforce_3_channels=True,
                 outputs={'image', 'label'}, isr_noise=False, isr_cow_mask=False, high_resolution_isr=False, random_flare=None,
                 cs_isr_data_type='day', sky_mask=None, shift_3_channel=False, isr_parms='', shift_type='rightdown',
                 synthetic_length=2975, raw_size=(2048, 1024), seed=0, device=None):
        _warn_synthetic(type(self), dataset_path)
        assert image_crop_size[0] <= image_resize_size[0] and image_crop_size[1] <= image_resize_size[1]
        assert not (isr_noise or isr_cow_mask or high_resolution_isr or shift_3_channel) and random_flare is None and sky_mask is None, \
            'augmentations that are off in configs/fusion/* are not implemented'
        assert shift_type in {'all', 'random', 'rightdown'}
        self.image_resize_size, self.image_crop_size = tuple(image_resize_size), tuple(image_crop_size)
        self.outputs, self.CLASSES, self.PALETTE = set(outputs), classes, palette
        self.isr_parms = dict(isr_parms) if isr_parms != '' else dict(_DEFAULT_ISR, shift_pixel=isr_shift_pixel)
        self.shift_type, self.enforce_3_channels = shift_type, enforce_3_channels
        self.length, self.raw_size, self.seed, self.device = synthetic_length, tuple(raw_size), seed, _device(device)
        self.file_path = {'label': [f'synthetic/{i:06d}_gtFine_labelTrainIds.png' for i in range(min(synthetic_length, 4096))]}

    def raw(self, idx):
        """(frame uint8 [H,W,3], previous frame uint8 [H,W,3], label int64 [rh,rw] at the resized resolution)"""
        g = self._gen(idx)
        W, H = self.raw_size
        now = _blocky_u8(g, H, W, 3)
        prev = (now.int() + (torch.randn((H, W, 3), generator=g) * 18).int()).clamp(0, 255).to(torch.uint8)
        return now, prev, _blocky_labels(g, self.image_resize_size[1], self.image_resize_size[0])

    def draw_decisions(self):
        """the loader's random decisions of ONE sample, in the reference's order (cityscapes_ic.py:149-151): (flip, x, y)"""
        (rw, rh), (cw, ch) = self.image_resize_size, self.image_crop_size
        return int(random.random() < 0.5), random.randint(0, rw - cw), random.randint(0, rh - ch)

    def label_crop(self, idx, decision):
        """the cropped / flipped label of sample idx under `decision` (host tensor [ch, cw]): what Rare-Class-Sampling inspects
        before it accepts a crop (uda_dataset.py:96-107) -- without running the image pipeline for a crop it may reject"""
        f, x, y = decision
        cw, ch = self.image_crop_size
        lab = self.raw_label(idx)[y:y + ch, x:x + cw]
        return torch.flip(lab, dims=[-1]) if f else lab

    def raw_label(self, idx):
        g = self._gen(idx)
        W, H = self.raw_size
        _blocky_u8(g, H, W, 3)                      # keep the generator stream of raw(): frame, previous-frame noise, label
        torch.randn((H, W, 3), generator=g)
        return _blocky_labels(g, self.image_resize_size[1], self.image_resize_size[0])

    def get_batch(self, indices, decisions=None):
        """decisions: optional per-sample (flip, x, y) drawn by the caller (UDADataset's Rare-Class-Sampling re-crop loop)"""
        dev = self.device
        B = len(indices)
        (rw, rh), (cw, ch) = self.image_resize_size, self.image_crop_size
        if decisions is None:
            decisions = [self.draw_decisions() for _ in indices]
        flips, xs, ys = [d[0] for d in decisions], [d[1] for d in decisions], [d[2] for d in decisions]
        raws = [self.raw(i) for i in indices]
        now = _stack_dev([r[0] for r in raws], dev)
        out = {}
        samp = pl.make_samp(B, dev, out_x0=xs, out_y0=ys, flip_out=flips)
        W, H = self.raw_size
        need_isr = 'img_self_res' in self.outputs
        if 'image' in self.outputs or need_isr:
            r = pl.pil_resize_u8(now, samp, (W, H), (rw, rh), (cw, ch), norm=(pl.IMAGENET_MEAN, pl.IMAGENET_STD), want_gray=need_isr)
            out['image'] = r['f']
        if 'label' in self.outputs:
            labs = []
            for (_, _, lab), x, y, f in zip(raws, xs, ys, flips):
                lab = lab[y:y + ch, x:x + cw]
                labs.append((torch.flip(lab, dims=[-1]) if f else lab)[None])
            # the class set ClassMix draws from (dacs_transforms.py:103-104: unique over the whole batch), taken HERE on the host
            # from the cropped labels: the training step then needs no device read at all for it (the reference's
            # `torch.unique(labels)` + `.cpu()` is a sync; a side-stream read would have to be ordered after the copy below)
            # (out-of-band attributes of the label tensor: the batch-dict schema stays the reference's)
            classes = torch.unique(torch.stack(labs))
            out['label'] = _stack_dev(labs, dev)
            out['label']._cmda_classes = classes
            # (validated by the consumer: an in-place refill of this buffer bumps _version and the stale set is dropped)
            out['label']._cmda_classes_key = (out['label'].data_ptr(), out['label']._version)
            if dev.type == 'cuda':
                out['label']._cmda_ready = torch.cuda.Event()
                out['label']._cmda_ready.record()
        if 'img_time_res' in self.outputs:
            prev = _stack_dev([r[1] for r in raws], dev)
            tr = pl.time_residual_u8(pl.luma_u8(now), pl.luma_u8(prev))
            t = pl.pil_resize_u8(tr.view(B, H, W, 1), samp, (W, H), (rw, rh), (cw, ch), norm=((0.5,) * 3, (0.5,) * 3),
                                 rep3=self.enforce_3_channels)
            out['img_time_res'] = t['f']
        if need_isr:
            isr = []
            for b in range(B):   # the shift direction is drawn per sample (cityscapes_ic.py: direct[x % 2][y % 2])
                d = _DIRECT[xs[b] % 2][ys[b] % 2] if self.shift_type == 'random' else self.shift_type
                isr.append(ops.isr_from_gray(r['gray'][b:b + 1], self.isr_parms['val_range'], self.isr_parms['_threshold'],
                                             self.isr_parms['_clip_range'], self.isr_parms['shift_pixel'], d))
            out['img_self_res'] = torch.cat(isr)
        return out

    def sample_class_stats(self, n=256):
        """per-sample class pixel counts of the synthetic labels (the role of sample_class_stats.json for Rare-Class-Sampling)"""
        stats = []
        for i in range(min(n, self.length)):
            lab = self.raw(i)[2]
            ids, cnt = torch.unique(lab[lab != 255], return_counts=True)
            stats.append({'file': self.file_path['label'][i], **{str(int(c)): int(k) for c, k in zip(ids, cnt)}})
        return stats


@DATASETS.register_module()
class DSECDataset(_SyntheticBase):
    """mmseg/datasets/dsec.py:124-366 (target domain).  Training samples (no 'label' in outputs): random crop `crop_size` of the
    480x640 frame -> random flip -> PIL resize to `after_crop_resize_size`; 'warp_img_self_res' = real-time ISR of the resized
    frame; 'events_vg' = rectified events -> voxel grid (events_bins) -> events_norm -> the same crop / flip -> bilinear -> x3.
    Test samples ('label' in outputs): the first 440 rows, no augmentation, + 'label' [440,640] and 'img_metas'."""

    def __init__(self, dataset_txt_path='', events_num=-1, events_bins=5, events_clip_range=None, crop_size=(400, 400),
                 after_crop_resize_size=(512, 512), image_change_range=1, outputs={'events_vg', 'image'}, output_num=1,
                 classes=CLASSES, palette=PALETTE, isr_shift_pixel=4, test_mode=False, events_bins_5_avg_1=False, isr_parms='',
                 isr_type='real_time', enforce_3_channels=True, shift_type='rightdown', synthetic_length=1692,
                 synthetic_events=500000, seed=1, device=None):
        _warn_synthetic(type(self), dataset_txt_path)
        assert output_num == 1 and not events_bins_5_avg_1 and isr_type == 'real_time'
        assert shift_type in {'all', 'random', 'rightdown'}
        self.outputs = set(outputs)
        train = 'label' not in self.outputs
        self.crop_size = (crop_size[1], crop_size[0]) if train else tuple(crop_size)                      # (W, H), dsec.py:151
        self.after_crop_resize_size = (after_crop_resize_size[1], after_crop_resize_size[0]) if train else tuple(after_crop_resize_size)
        self.events_bins, self.events_clip_range, self.events_num = events_bins, events_clip_range, events_num
        self.CLASSES, self.PALETTE = classes, palette
        self.events_height, self.events_width = 480, 640
        self.isr_parms = dict(isr_parms) if isr_parms != '' else dict(_DEFAULT_ISR)
        self.shift_type, self.enforce_3_channels = shift_type, enforce_3_channels
        self.length, self.n_events, self.seed, self.device = synthetic_length, synthetic_events, seed, _device(device)
        yy, xx = np.meshgrid(np.arange(480, dtype=np.float32), np.arange(640, dtype=np.float32), indexing='ij')
        rect = np.stack([np.clip(xx + 1.5 * np.sin(yy / 60.0), 0, 638.99), np.clip(yy + 1.0 * np.cos(xx / 80.0), 0, 478.99)], -1)
        self.rectify_map = torch.from_numpy(rect.astype(np.float32)).to(self.device)   # rectify_map[y, x] = (x_rect, y_rect)

    def raw(self, idx):
        g = self._gen(idx)
        n = self.n_events if self.events_num == -1 else self.events_num
        t = torch.sort(torch.randint(0, 50000, (n,), generator=g))[0] + 1_000_000
        return (_blocky_u8(g, 480, 640, 3), t, torch.randint(0, 640, (n,), generator=g, dtype=torch.int32),
                torch.randint(0, 480, (n,), generator=g, dtype=torch.int32), torch.randint(0, 2, (n,), generator=g, dtype=torch.uint8),
                _blocky_labels(g, 480, 640))

    def _voxel(self, raw):
        dev = self.device
        _, t, x, y, p, _ = raw
        tn, xr, yr, pol = pl.event_prep(t.to(dev), x.to(dev), y.to(dev), p.to(dev), self.rectify_map, 480, 640)
        vg = ops.events_to_voxel_grid(tn, xr, yr, pol, self.events_bins, 480, 640)
        clip = random.uniform(*self.events_clip_range) if self.events_clip_range is not None else (t.numel() - 1) / 500000 * 1.5
        return ops.events_norm(vg, clip)

    def get_batch(self, indices):
        dev = self.device
        B = len(indices)
        train = 'label' not in self.outputs
        raws = [self.raw(i) for i in indices]
        frames = _stack_dev([r[0] for r in raws], dev)
        out = {}
        if train:
            cw, ch = self.crop_size
            flips, xs, ys = [], [], []
            for _ in indices:    # dsec.py:203-206
                flips.append(int(random.random() < 0.5))
                xs.append(random.randint(0, 640 - cw))
                ys.append(random.randint(0, 480 - ch))
            samp = pl.make_samp(B, dev, src_x0=xs, src_y0=ys, flip_src=flips)
            need_isr = 'warp_img_self_res' in self.outputs
            if {'warp_image', 'warp_img_self_res'} & self.outputs:
                r = pl.pil_resize_u8(frames, samp, (cw, ch), self.after_crop_resize_size, norm=(pl.IMAGENET_MEAN, pl.IMAGENET_STD),
                                     want_gray=need_isr)
                if 'warp_image' in self.outputs:
                    out['warp_image'] = r['f']
                if need_isr:
                    isr = []
                    for b in range(B):
                        d = _DIRECT[xs[b] % 2][ys[b] % 2] if self.shift_type == 'random' else self.shift_type
                        isr.append(ops.isr_from_gray(r['gray'][b:b + 1], self.isr_parms['val_range'], self.isr_parms['_threshold'],
                                                     self.isr_parms['_clip_range'], self.isr_parms['shift_pixel'], d))
                    out['warp_img_self_res'] = torch.cat(isr)
            if 'events_vg' in self.outputs:
                vg = torch.stack([self._voxel(r) for r in raws])
                out['events_vg'] = pl.crop_flip_resize_f32(vg, samp, (cw, ch), self.after_crop_resize_size,
                                                           rep=3 if (self.enforce_3_channels and self.events_bins == 1) else 1)
            return out
        # test mode (dsec.py:229-230, 324-325): no augmentation, the first 440 rows
        samp = pl.make_samp(B, dev)
        if 'warp_image' in self.outputs:
            r = pl.pil_resize_u8(frames, samp, (640, 480), (640, 480), norm=(pl.IMAGENET_MEAN, pl.IMAGENET_STD))
            out['warp_image'] = r['f'][:, :, :440].contiguous()
        if 'events_vg' in self.outputs:
            vg = torch.stack([self._voxel(r) for r in raws])[:, :, :440, :]
            out['events_vg'] = (vg.repeat(1, 3, 1, 1) if (self.enforce_3_channels and self.events_bins == 1) else vg).contiguous()
        out['label'] = _stack_dev([r[5][:440] for r in raws], dev)
        if 'img_metas' in self.outputs:
            out['img_metas'] = [dict(img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
                                     img_shape=(440, 640), pad_shape=(440, 640), ori_shape=(440, 640),
                                     ori_filename=f'synthetic_{int(i):06d}.png', flip=False) for i in indices]
        return out

    def get_gt_seg_maps(self, efficient_test=False):
        return [self.raw(i)[5][:440].numpy() for i in range(self.length)]


@DATASETS.register_module()
class DarkZurichICDataset(_SyntheticBase):
    """mmseg/datasets/dark_zurich_ic.py:21-330 (target of configs/fusion/cs2dz_image+raw-isr_b5.py): outputs ⊆ {'image',
    'night_isr', 'label'}; resize 1920x1080 -> image_resize_size, ISR of the RESIZED frame, then the same random crop / flip for
    both (train) or the whole resized frame + label (test)."""

    def __init__(self, dataset_path='', image_resize_size=(960, 540), image_crop_size=(512, 512), image_resize_size2=None,
                 test_mode=False, split_train=False, dz_isr_data_type='night', shift_pixel=3, enforce_3_channels=True,
                 classes=CLASSES, palette=PALETTE, outputs={'image', 'night_isr'}, submit_to_website=False, auto_threshold=False,
                 high_resolution_isr=False, isr_parms='', shift_3_channel=False, shift_type='rightdown', synthetic_length=2416,
                 raw_size=(1920, 1080), seed=2, device=None):
        _warn_synthetic(type(self), dataset_path)
        assert image_resize_size2 is None and not (auto_threshold or high_resolution_isr or shift_3_channel or submit_to_website)
        self.image_resize_size, self.image_crop_size = tuple(image_resize_size), tuple(image_crop_size)
        self.test_mode, self.outputs = test_mode, set(outputs)
        self.CLASSES, self.PALETTE = classes, palette
        self.isr_parms = dict(isr_parms) if isr_parms != '' else dict(_DEFAULT_ISR, shift_pixel=shift_pixel)
        self.shift_type, self.enforce_3_channels = shift_type, enforce_3_channels
        self.length, self.raw_size, self.seed, self.device = synthetic_length, tuple(raw_size), seed, _device(device)

    def raw(self, idx):
        g = self._gen(idx)
        return _blocky_u8(g, self.raw_size[1], self.raw_size[0], 3), _blocky_labels(g, self.image_resize_size[1], self.image_resize_size[0])

    def get_batch(self, indices):
        dev = self.device
        B = len(indices)
        (rw, rh), (cw, ch) = self.image_resize_size, self.image_crop_size
        W, H = self.raw_size
        raws = [self.raw(i) for i in indices]
        frames = _stack_dev([r[0] for r in raws], dev)
        flips, xs, ys = [0] * B, [0] * B, [0] * B
        if not self.test_mode:
            flips, xs, ys = [], [], []
            for _ in indices:    # dark_zurich_ic.py:140-143
                flips.append(int(random.random() < 0.5))
                xs.append(random.randint(0, rw - cw))
                ys.append(random.randint(0, rh - ch))
        out = {}
        full = pl.pil_resize_u8(frames, pl.make_samp(B, dev), (W, H), (rw, rh), want_u8=True, want_gray=True,
                                norm=(pl.IMAGENET_MEAN, pl.IMAGENET_STD))
        if self.test_mode:
            out['image'] = full['f']
            out['label'] = _stack_dev([r[1][None] for r in raws], dev)
        else:
            samp = pl.make_samp(B, dev, out_x0=xs, out_y0=ys, flip_out=flips)
            out['image'] = pl.pil_resize_u8(frames, samp, (W, H), (rw, rh), (cw, ch), norm=(pl.IMAGENET_MEAN, pl.IMAGENET_STD))['f']
        if 'night_isr' in self.outputs:
            isr = []
            for b in range(B):
                d = _DIRECT[xs[b] % 2][ys[b] % 2] if (self.shift_type == 'random' and not self.test_mode) else \
                    ('rightdown' if self.shift_type == 'random' else self.shift_type)
                v = ops.isr_from_gray(full['gray'][b:b + 1], self.isr_parms['val_range'], self.isr_parms['_threshold'],
                                      self.isr_parms['_clip_range'], self.isr_parms['shift_pixel'], d)
                if not self.test_mode:   # the ISR is computed on the whole resized frame, THEN cropped and flipped (:252-256)
                    v = v[:, :, ys[b]:ys[b] + ch, xs[b]:xs[b] + cw]
                    v = torch.flip(v, dims=[-1]) if flips[b] else v
                isr.append(v.contiguous())
            out['night_isr'] = torch.cat(isr)
        return out


def get_rcs_class_probs(stats, temperature):
    """uda_dataset.py:12-34: class frequencies over the source labels -> softmax((1 - freq) / T).  `stats` = the parsed
    sample_class_stats.json (list of {'file', '<class id>': pixels})."""
    overall = {}
    for s in stats:
        for c, n in s.items():
            if c == 'file':
                continue
            overall[int(c)] = overall.get(int(c), 0) + n
    overall = {k: v for k, v in sorted(overall.items(), key=lambda item: item[1])}
    freq = torch.tensor(list(overall.values()), dtype=torch.float32)
    freq = freq / torch.sum(freq)
    freq = 1 - freq
    freq = torch.softmax(freq / temperature, dim=-1)
    return list(overall.keys()), freq.numpy()


@DATASETS.register_module()
class UDADataset:
    """uda_dataset.py:37-143: pairs a source and a target sample -> {'source': ..., 'target': ...}; optional Rare-Class-Sampling
    of the source index (class drawn from `get_rcs_class_probs`, then a file that contains it)."""

    def __init__(self, source, target, cfg):
        self.source, self.target = source, target
        self.ignore_index, self.CLASSES, self.PALETTE = target.ignore_index, target.CLASSES, target.PALETTE
        assert target.ignore_index == source.ignore_index and target.CLASSES == source.CLASSES and target.PALETTE == source.PALETTE
        rcs_cfg = cfg.get('rare_class_sampling')
        self.rcs_enabled = rcs_cfg is not None
        if self.rcs_enabled:
            self.rcs_class_temp = rcs_cfg['class_temp']
            self.rcs_min_crop_ratio, self.rcs_min_pixels = rcs_cfg['min_crop_ratio'], rcs_cfg['min_pixels']
            root = cfg.get('source_json_root') or ''
            path = os.path.join(root, 'sample_class_stats.json')
            stats = json.load(open(path)) if os.path.exists(path) else source.sample_class_stats()
            self.rcs_classes, self.rcs_classprob = get_rcs_class_probs(stats, self.rcs_class_temp)
            swc_path = os.path.join(root, 'samples_with_class.json')
            if os.path.exists(swc_path):   # uda_dataset.py:63-76: {class: [[file, pixels], ...]}
                swc = {int(k): v for k, v in json.load(open(swc_path)).items()}
                self.samples_with_class = {c: [f for f, px in swc.get(c, []) if px > self.rcs_min_pixels] for c in self.rcs_classes}
            else:
                self.samples_with_class = {c: [s['file'] for s in stats if s.get(str(c), 0) > self.rcs_min_pixels] for c in self.rcs_classes}
            self.rcs_classes = [c for c in self.rcs_classes if self.samples_with_class[c]]
            self.rcs_classprob = np.array([p for c, p in zip(get_rcs_class_probs(stats, self.rcs_class_temp)[0], self.rcs_classprob)
                                           if c in self.rcs_classes])
            self.rcs_classprob = self.rcs_classprob / self.rcs_classprob.sum()
            self.file_to_idx = {f: i for i, f in enumerate(source.file_path['label'])}

    def _source_index(self, idx):
        """-> (source index, accepted crop decision or None).  Rare-Class-Sampling (uda_dataset.py:89-107): draw a class, a file
        that holds it, then re-crop that file up to 10 times until the crop shows more than min_pixels * min_crop_ratio pixels of
        the class (each re-crop = one more `self.source[i1]` in the reference, i.e. one more (flip, x, y) draw)."""
        if not self.rcs_enabled:
            return idx // len(self.target), None
        c = np.random.choice(self.rcs_classes, p=self.rcs_classprob)
        i1 = self.file_to_idx[np.random.choice(self.samples_with_class[c])]
        if not hasattr(self.source, 'draw_decisions'):
            return i1, None
        dec = self.source.draw_decisions()
        if self.rcs_min_crop_ratio > 0:
            for _ in range(10):
                n_class = int((self.source.label_crop(i1, dec) == int(c)).sum())
                if n_class > self.rcs_min_pixels * self.rcs_min_crop_ratio:
                    break
                dec = self.source.draw_decisions()
        return i1, dec

    def _target_index(self, idx):
        return int(np.random.choice(range(len(self.target)))) if self.rcs_enabled else idx % len(self.target)

    def __getitem__(self, idx):
        b = self.get_batch([idx])
        pick = lambda d: {k: (v[0] if isinstance(v, (torch.Tensor, list)) else v) for k, v in d.items()}   # noqa: E731
        return {'source': pick(b['source']), 'target': pick(b['target'])}

    def get_batch(self, indices):
        # per sample, in the reference's order (get_rare_class_sample): source draws, then the target index
        src, tgt = [], []
        for i in indices:
            src.append(self._source_index(i))
            tgt.append(self._target_index(i))
        if all(d is None for _, d in src):
            source = self.source.get_batch([i for i, _ in src])
        else:
            source = self.source.get_batch([i for i, _ in src], decisions=[d for _, d in src])
        return {'source': source, 'target': self.target.get_batch(tgt)}

    def __len__(self):
        return len(self.source) * len(self.target)


def build_dataset(cfg, default_args=None):
    """mmseg/datasets/builder.py:66-91 (the UDADataset special case; other wrappers are out of scope)"""
    if cfg['type'] == 'UDADataset':
        return UDADataset(source=build_dataset(cfg['source'], default_args), target=build_dataset(cfg['target'], default_args), cfg=cfg)
    return build_from_cfg(cfg, DATASETS, default_args)


class _Loader:
    """Iterable of collated batches.  Each rank of a data-parallel job walks its own shard of a seeded permutation of the indices
    (the role of DistributedSampler in builder.py:137-138); batches come from `dataset.get_batch` (device-resident tensors)."""

    def __init__(self, dataset, samples_per_gpu, rank, world, shuffle, seed, drop_last):
        self.dataset, self.bs, self.rank, self.world = dataset, samples_per_gpu, rank, world
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed or 0, drop_last, 0

    def __len__(self):
        per_rank = len(self.dataset) // self.world
        return per_rank // self.bs if self.drop_last else -(-per_rank // self.bs)

    def __iter__(self):
        n = len(self.dataset)
        per_rank = n // self.world
        if self.shuffle:   # a permutation of a (possibly huge: len(source) * len(target)) range without materialising it
            g = np.random.RandomState(self.seed + self.epoch)
            a, b = int(g.randint(1, 1 << 30)) | 1, int(g.randint(0, 1 << 30))
            while np.gcd(a, n) != 1:
                a += 2
            order = lambda k: (a * k + b) % n  # noqa: E731
        else:
            order = lambda k: k  # noqa: E731
        self.epoch += 1
        for s in range(0, per_rank - (self.bs - 1 if self.drop_last else 0), self.bs):
            idx = [order(self.rank * per_rank + k) for k in range(s, min(s + self.bs, per_rank))]
            yield self.dataset.get_batch(idx)


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu=0, num_gpus=1, dist=True, shuffle=True, seed=None, drop_last=False,
                     pin_memory=True, dataloader_type='PoolDataLoader', **kwargs):
    """mmseg/datasets/builder.py:94-177 with the CPU worker pool replaced by the on-device pipeline: `workers_per_gpu`,
    `pin_memory` and `dataloader_type` are accepted and ignored (there is no host-side decoding to parallelise)."""
    rank = world = None
    if dist and torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    return _Loader(dataset, samples_per_gpu, rank or 0, world or 1, shuffle, seed, drop_last)
