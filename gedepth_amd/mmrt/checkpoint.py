"""Checkpoints in the mmcv layout the reference's released weights use: ``{'meta', 'state_dict', 'optimizer'}``
(mmcv CheckpointHook / load_checkpoint as used by depth/apis/train.py:117-120 and tools/test.py:129)."""
import os
import re

import torch


def _state_dict_cpu(model):
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def save_checkpoint(model, path, optimizer=None, meta=None):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = dict(meta=dict(meta or {}), state_dict=_state_dict_cpu(model))
    if optimizer is not None:
        osd = optimizer.state_dict()
        ckpt['optimizer'] = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in osd.items()}
    tmp = path + '.tmp'
    torch.save(ckpt, tmp)
    os.replace(tmp, path)


def load_checkpoint(model, path, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),)):
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {path}')
    state = ckpt.get('state_dict', ckpt.get('model', ckpt))
    for pat, rep in revise_keys:
        state = {re.sub(pat, rep, k): v for k, v in state.items()}
    result = model.load_state_dict(state, strict=strict)
    ckpt['_load_result'] = result
    return ckpt
