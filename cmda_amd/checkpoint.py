"""Checkpoint I/O for the path's modules ("next" row).

The reference goes through mmcv (third-party, absent from /root/reference): `mmcv.runner.load_checkpoint(model, filename,
map_location, strict, revise_keys=[(r'^module\\.', '')])` in mmseg/apis/inference.py:42-45 and mmseg/apis/train.py:130,
`_load_checkpoint` + `load_state_dict(strict=False)` in mix_transformer.py:343-357 (ImageNet weights: the file holds either
the bare state dict or {'state_dict': ...} / {'model': ...}), and `save_checkpoint` from the runner's checkpoint hook
({'meta': ..., 'state_dict': ..., 'optimizer': ...}, weights moved to the CPU).  Restated from that documented behaviour:
same file layout, same key handling, so checkpoints written by either side load in the other (the state-dict keys of every
module here equal the reference's: tests/golden/*_keys.json).

After loading, the kernels' cached compute copies of the parameters are invalidated (runtime.invalidate) and every live
bf16 mirror of a loaded parameter (FlatAdamW's flat mirror of the student, DACS's flat mirror of the EMA teacher) is
re-cast from the fp32 master in place, so a load is self-healing whatever store the parameters were re-homed into.
`strip_for_release` restates function.py:28-37 (drop `ema_model.*` / `cyclegan*` keys from a training checkpoint).
"""
import re
import time
from collections import OrderedDict

import torch

from . import runtime as rt


def _state_dict_of(ckpt):
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'no state dict found in checkpoint of type {type(ckpt)}')
    for key in ('state_dict', 'model'):
        if key in ckpt and isinstance(ckpt[key], dict):
            return ckpt[key]
    return ckpt


def load_state_dict(module, state_dict, strict=False, logger=None):
    """mmcv.runner.load_state_dict: never raises on shape/key mismatch unless strict; returns (missing, unexpected)."""
    own = module.state_dict()
    missing = [k for k in own if k not in state_dict and 'num_batches_tracked' not in k]
    unexpected = [k for k in state_dict if k not in own]
    mismatch = [k for k in state_dict if k in own and tuple(state_dict[k].shape) != tuple(own[k].shape)]
    usable = {k: v for k, v in state_dict.items() if k in own and k not in mismatch}
    with torch.no_grad():
        for k, v in usable.items():
            own[k].copy_(v)   # in place: parameters re-homed into flat optimizer buffers keep their storage
        # live bf16 compute mirrors (runtime.w returns them without consulting the cache): refresh from the new masters
        for p in module.parameters():
            mirror = getattr(p, '_cmda_bf16', None)
            if mirror is not None:
                mirror.copy_(p.data)
    rt.invalidate()
    rt.refresh_frozen(module)   # compute copies of frozen parameters (the Motion-Extractor generator) are outside invalidate()
    msg = []
    if unexpected:
        msg.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msg.append('missing keys in source state_dict: ' + ', '.join(missing))
    if mismatch:
        msg.append('size mismatch for: ' + ', '.join(mismatch))
    if msg:
        text = 'The model and loaded state dict do not match exactly\n' + '\n'.join(msg)
        if strict:
            raise RuntimeError(text)
        if logger is not None:
            logger.warning(text)
    return missing, unexpected + mismatch


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None, revise_keys=((r'^module\.', ''),)):
    """Load `filename` into `model` (a bare module or one wrapped as `.module`); returns the checkpoint dict."""
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    sd = _state_dict_of(ckpt)
    meta = getattr(sd, '_metadata', None)
    for pat, rep in revise_keys:
        sd = OrderedDict((re.sub(pat, rep, k), v) for k, v in sd.items())
    if meta is not None:
        sd._metadata = meta
    load_state_dict(getattr(model, 'module', model), sd, strict, logger)
    return ckpt


def weights_to_cpu(state_dict):
    out = OrderedDict((k, v.detach().cpu()) for k, v in state_dict.items())
    out._metadata = getattr(state_dict, '_metadata', OrderedDict())
    return out


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """{'meta', 'state_dict'[, 'optimizer']} with CPU weights, as the runner's CheckpointHook writes it."""
    meta = dict(meta or {})
    meta.setdefault('time', time.asctime())
    module = getattr(model, 'module', model)
    if hasattr(module, 'CLASSES') and module.CLASSES is not None:
        meta.setdefault('CLASSES', module.CLASSES)
    if optimizer is not None and hasattr(optimizer, 'synchronize'):
        optimizer.synchronize()   # an overlapped (postponed) update lands before the weights are read (optim.FlatAdamW.overlap)
    opt_of_model = getattr(module, '_opt', None)
    if opt_of_model is not None and opt_of_model is not optimizer and hasattr(opt_of_model, 'synchronize'):
        opt_of_model.synchronize()
    ckpt = {'meta': meta, 'state_dict': weights_to_cpu(module.state_dict())}
    if optimizer is not None:
        if not hasattr(optimizer, 'state_dict'):
            raise TypeError(f'{type(optimizer).__name__} has no state_dict(): refusing to pickle the optimizer object')
        ckpt['optimizer'] = optimizer.state_dict()
    torch.save(ckpt, filename)
    return ckpt


def strip_for_release(state_dict):
    """function.py:28-37 (`--function convert_pth`): the released checkpoint is the training state dict without the EMA
    teacher and the Motion-Extractor generator -- every key containing 'ema_model' or 'cyclegan' is dropped."""
    return OrderedDict((k, v) for k, v in state_dict.items() if 'ema_model' not in k and 'cyclegan' not in k)
