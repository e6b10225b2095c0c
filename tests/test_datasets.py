"""Dataset registry keys of configs/fusion/* and the batch-dict schema the training step consumes (reference:
mmseg/datasets/builder.py:26-27,66-177, uda_dataset.py:37-143, cityscapes_ic.py:147-272, dsec.py:189-339, dark_zurich_ic.py)."""
import os
import random

import numpy as np
import pytest
import torch

import cmda_amd  # noqa: F401
from cmda_amd import datasets as D
from cmda_amd.config import Config
from cmda_amd.registry import DATASETS

REF = '/root/reference/configs/fusion'
SMALL_SRC = dict(raw_size=(256, 128), image_resize_size=(128, 64), image_crop_size=(64, 64), synthetic_length=6)


def test_registry_keys_resolve():
    for key in ('UDADataset', 'CityscapesICDataset', 'DSECDataset', 'DarkZurichICDataset'):
        assert key in DATASETS, key


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference configs are only present in the authoring container')
@pytest.mark.parametrize('name', ['cs2dsec_image+events_together_b5.py', 'cs2dz_image+raw-isr_b5.py'])
def test_reference_data_configs_build(name):
    """cfg.data.train / val of the reference's own configs go through build_dataset unchanged (constructor arguments accepted)"""
    cfg = Config.fromfile(os.path.join(REF, name))
    train = D.build_dataset(cfg.data.train)
    assert type(train).__name__ == 'UDADataset' and len(train) == len(train.source) * len(train.target)
    assert type(train.source).__name__ == 'CityscapesICDataset'
    assert type(train.target).__name__ in ('DSECDataset', 'DarkZurichICDataset')
    val = D.build_dataset(cfg.data.val)
    assert 'label' in val.outputs
    loader = D.build_dataloader(train, cfg.data.samples_per_gpu, cfg.data.workers_per_gpu, 1, dist=False, seed=0, drop_last=True)
    assert len(loader) == len(train) // cfg.data.samples_per_gpu


def test_uda_batch_schema(tgt):
    """{'source': {image, label, img_time_res, img_self_res}, 'target': {warp_image, events_vg, warp_img_self_res}} with the
    reference's shapes / dtypes / value ranges, batched == per-sample (same random draws), and Rare-Class-Sampling wiring"""
    dev = tgt.device
    isr = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
    cfg = dict(type='UDADataset',
               source=dict(type='CityscapesICDataset', outputs={'image', 'label', 'img_time_res', 'img_self_res'}, isr_parms=isr,
                           shift_type='random', device=dev, **SMALL_SRC),
               target=dict(type='DSECDataset', crop_size=(400, 400), after_crop_resize_size=(64, 64), events_bins=1, isr_parms=isr,
                           outputs={'warp_image', 'events_vg', 'warp_img_self_res'}, shift_type='random', synthetic_length=5,
                           synthetic_events=3000, device=dev))
    ds = D.build_dataset(cfg)
    assert len(ds) == 30
    random.seed(3)
    b = ds.get_batch([0, 7])
    src, tg = b['source'], b['target']
    assert set(src) == {'image', 'label', 'img_time_res', 'img_self_res'} and set(tg) == {'warp_image', 'events_vg', 'warp_img_self_res'}
    assert src['image'].shape == (2, 3, 64, 64) and src['image'].dtype == torch.float32
    assert src['label'].shape == (2, 1, 64, 64) and src['label'].dtype == torch.int64
    assert src['img_time_res'].shape == (2, 3, 64, 64) and src['img_self_res'].shape == (2, 3, 64, 64)
    assert tg['warp_image'].shape == (2, 3, 64, 64) and tg['events_vg'].shape == (2, 3, 64, 64) and tg['warp_img_self_res'].shape == (2, 3, 64, 64)
    for k in ('img_time_res', 'img_self_res'):
        assert src[k].abs().max().item() <= 1.0 + 1e-6 and torch.equal(src[k][:, 0], src[k][:, 1])
    for k in ('events_vg', 'warp_img_self_res'):
        assert tg[k].abs().max().item() <= 1.0 + 1e-6 and tg[k].abs().sum().item() > 0
    lab = src['label']
    assert ((lab >= 0) & (lab < 19) | (lab == 255)).all()
    # per-sample access consumes the same random draws in the same order
    random.seed(3)
    s0 = ds.source[0]
    assert s0['image'].shape == (3, 64, 64) and torch.equal(s0['image'], src['image'][0])
    # Rare-Class-Sampling (uda_dataset.py:12-34,86-108) on the synthetic class statistics
    cfg_rcs = dict(cfg, rare_class_sampling=dict(min_pixels=10, class_temp=0.01, min_crop_ratio=0.5))
    rcs = D.build_dataset(cfg_rcs)
    assert rcs.rcs_enabled and abs(float(np.sum(rcs.rcs_classprob)) - 1.0) < 1e-6
    np.random.seed(0)
    assert set(rcs.get_batch([0, 1])['source']) == set(src)


def test_dsec_test_mode_schema(tgt):
    ds = D.build_dataset(dict(type='DSECDataset', outputs={'warp_image', 'events_vg', 'label', 'img_metas'}, events_bins=1,
                              synthetic_length=3, synthetic_events=2000, device=tgt.device))
    s = ds[1]
    assert s['warp_image'].shape == (3, 440, 640) and s['events_vg'].shape == (3, 440, 640) and s['label'].shape == (440, 640)
    assert s['img_metas']['ori_shape'] == (440, 640) and len(ds.get_gt_seg_maps()) == 3


@pytest.mark.gpu
def test_loader_feeds_dacs_step_at_full_geometry():
    """the synthetic stream at the real geometries (Cityscapes 2048x1024 -> 512, DSEC 480x640 + 500 k events -> 512) drives a DACS
    iteration (reduced-width student): the schema is exactly what DACS.train_step consumes"""
    from conftest import Target
    from cmda_amd import _lib, optim
    import cmda_amd.runtime as rt
    from cmda_amd.registry import build_train_model
    from test_dacs import SMALL, make_cfg
    _lib._unbind_for_tests()
    if not torch.cuda.is_available():
        pytest.skip('no GPU on this machine')
    dev = Target('gpu').device
    rt.set_compute_dtype(torch.bfloat16)
    isr = dict(val_range=[0.01, 1.01], _threshold=0.005, _clip_range=0.1, shift_pixel=1)
    ds = D.build_dataset(dict(
        type='UDADataset',
        source=dict(type='CityscapesICDataset', outputs={'image', 'label', 'img_time_res', 'img_self_res'}, isr_parms=isr,
                    shift_type='random', synthetic_length=8, device=dev),
        target=dict(type='DSECDataset', crop_size=(400, 400), after_crop_resize_size=(512, 512), events_bins=1, isr_parms=isr,
                    outputs={'warp_image', 'events_vg', 'warp_img_self_res'}, shift_type='random', synthetic_length=8, device=dev)))
    loader = D.build_dataloader(ds, 2, 4, 1, dist=False, seed=0, drop_last=True)
    dacs = build_train_model(make_cfg(SMALL['dims'], SMALL['ch'])).to(dev).train()
    opt = optim.FlatAdamW(dacs.model, lr=6e-5, weight_decay=0.01)
    dacs.attach_flat_store(opt)

    class _Opt:   # the optimizer object DACS.train_step drives (zero_grad / step)
        zero_grad, step = opt.zero_grad, lambda self=None: opt.step(1.0)
    random.seed(0), np.random.seed(0), torch.manual_seed(0)
    it = iter(loader)
    for _ in range(2):
        batch = next(it)
        assert batch['source']['image'].shape == (2, 3, 512, 512) and batch['target']['events_vg'].shape == (2, 3, 512, 512)
        out = dacs.train_step(batch, _Opt())
        assert out['num_samples'] == 2 and np.isfinite(float(out['log_vars']['decode.loss_seg']))
        assert np.isfinite(float(out['log_vars']['mix.decode.loss_seg']))
    rt.set_compute_dtype(torch.float32)
