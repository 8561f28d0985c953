"""Checkpoints in the mmcv layout the reference's released weights use: ``{'meta', 'state_dict', 'optimizer'}``
(mmcv CheckpointHook / load_checkpoint as used by depth/apis/train.py:117-120 and tools/test.py:129)."""
import os
import re
import warnings

import torch


def _state_dict_cpu(model):
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def save_checkpoint(model, path, optimizer=None, meta=None):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = dict(meta=dict(meta or {}), state_dict=_state_dict_cpu(model))
    if optimizer is not None:
        def cpu(o):
            if torch.is_tensor(o):
                return o.detach().cpu()
            if isinstance(o, dict):
                return {k: cpu(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(cpu(v) for v in o)
            return o
        ckpt['optimizer'] = cpu(optimizer.state_dict())
    tmp = path + '.tmp'
    torch.save(ckpt, tmp)
    os.replace(tmp, path)


def load_checkpoint(model, path, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),), logger=None):
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {path}')
    state = ckpt.get('state_dict', ckpt.get('model', ckpt))
    for pat, rep in revise_keys:
        state = {re.sub(pat, rep, k): v for k, v in state.items()}
    result = model.load_state_dict(state, strict=strict)
    ckpt['_load_result'] = result
    # mmcv's load_checkpoint reports what did not line up; a checkpoint that matches NOTHING (e.g. a stray key prefix) would
    # otherwise be evaluated / trained from random weights without a word
    own = set(model.state_dict().keys())
    matched = own & set(state.keys())
    benign = ('num_batches_tracked', 'relative_position_index')
    missing = [k for k in result.missing_keys if not k.endswith(benign)]
    unexpected = [k for k in result.unexpected_keys if not k.endswith(benign)]
    if own and not matched:
        raise RuntimeError(f'{path}: none of the {len(state)} checkpoint keys matches the model '
                           f'(e.g. {next(iter(state), None)!r} vs {next(iter(own))!r})')
    log = logger or (lambda msg: warnings.warn(msg))
    if missing:
        log(f'load_checkpoint({path}): {len(missing)} missing keys, e.g. {missing[:5]}')
    if unexpected:
        log(f'load_checkpoint({path}): {len(unexpected)} unexpected keys, e.g. {unexpected[:5]}')
    return ckpt
