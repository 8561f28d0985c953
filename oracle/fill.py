"""Deterministic weight fill shared by the golden-fixture generator, the tests and smoke().

TEST INFRASTRUCTURE (see oracle/gedepth_oracle.py header).  ``fill_tensor(name, shape)``
depends only on the parameter *name* and *shape*, so the reference (in the build container),
the oracle and the product model can all be given identical weights without committing
hundreds of MB of checkpoints.  numpy ``RandomState`` is used because its stream is frozen
across numpy versions.
"""
import zlib

import numpy as np
import torch


def _rng(name, salt=''):
    return np.random.RandomState(zlib.crc32((salt + '|' + name).encode()) & 0x7FFFFFFF)


def fill_tensor(name, shape, dtype=torch.float32, salt=''):
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if leaf == 'relative_position_index':
        return None  # structural buffer: keep the module's own
    r = _rng(name, salt)
    x = r.standard_normal(shape).astype(np.float32) if len(shape) else np.float32(r.standard_normal())
    if leaf == 'running_var':
        x = 1.0 + 0.2 * np.abs(x)
    elif leaf == 'running_mean':
        x = 0.1 * x
    elif leaf == 'relative_position_bias_table':
        x = 0.5 * x
    elif leaf == 'level_embed':
        x = 0.5 * x
    elif leaf == 'bias':
        x = (1.5 if 'sampling_offsets' in name else 0.1) * x
    elif leaf == 'weight' and len(shape) == 1:      # LayerNorm / BatchNorm scale
        x = 1.0 + 0.1 * x
    elif len(shape) >= 2:                            # Linear / Conv: ~unit-gain fan-in scaling
        fan_in = int(np.prod(shape[1:]))
        x = x * (1.0 / np.sqrt(fan_in))
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def fill_state_dict(spec, salt=''):
    """spec: iterable of (name, shape) or a module state_dict -> {name: tensor} (index buffers skipped)."""
    if isinstance(spec, dict):
        spec = [(k, tuple(v.shape)) for k, v in spec.items()]
    out = {}
    for name, shape in spec:
        t = fill_tensor(name, shape, salt=salt)
        if t is not None:
            out[name] = t
    return out


def load_filled(module, salt=''):
    """Overwrite every parameter/buffer of ``module`` with the rule; returns the state dict used."""
    sd = module.state_dict()
    new = fill_state_dict(sd, salt)
    for k, v in new.items():
        new[k] = v.to(sd[k].dtype)
    missing = module.load_state_dict(new, strict=False)
    assert not missing.unexpected_keys, missing
    return module.state_dict()
