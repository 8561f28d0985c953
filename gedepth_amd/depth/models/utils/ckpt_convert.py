"""Official Swin checkpoint -> DepthFormerSwin key/weight layout (depth/models/utils/ckpt_convert.py:5-56).

``nn.Unfold`` orders the 4C merged features channel-major (c, kh, kw) while the official model
concatenates the four strided sub-grids [x0,x1,x2,x3] = (kh,kw) in (0,0),(1,0),(0,1),(1,1) order, hence
the [0,2,1,3] swap + transpose of ``downsample.reduction`` / ``downsample.norm``.
"""
from collections import OrderedDict

_RENAMES = (('attn.', 'attn.w_msa.'), ('mlp.fc1.', 'ffn.layers.0.0.'), ('mlp.fc2.', 'ffn.layers.1.'))


def _reorder_merge(v):
    """(…, 4*C) official sub-grid-major order -> unfold channel-major order along the last dim."""
    lead = v.shape[:-1]
    c = v.shape[-1] // 4
    v = v.reshape(*lead, 4, c)[..., [0, 2, 1, 3], :]
    return v.transpose(-1, -2).reshape(*lead, 4 * c)


def swin_convert(ckpt):
    out = OrderedDict()
    for k, v in ckpt.items():
        if k.startswith('head'):
            continue
        if k.startswith('layers'):
            nk = k
            if 'attn.' in k:
                nk = k.replace(*_RENAMES[0])
            elif 'mlp.' in k:
                if 'mlp.fc1.' in k:
                    nk = k.replace(*_RENAMES[1])
                elif 'mlp.fc2.' in k:
                    nk = k.replace(*_RENAMES[2])
                else:
                    nk = k.replace('mlp.', 'ffn.')
            elif 'downsample' in k and ('reduction.' in k or 'norm.' in k):
                v = _reorder_merge(v)
            nk = nk.replace('layers', 'stages', 1)
        elif k.startswith('patch_embed'):
            nk = k.replace('proj', 'projection') if 'proj' in k else k
        else:
            nk = k
        out[nk] = v
    return out
