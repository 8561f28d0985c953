# GEDepth-Vanilla on a Swin-T DepthFormer (BASELINE.json configs[0]/[1]; not expressible with the reference's
# hard-coded neck widths — SURVEY.md S3 — hence the explicit in_channels of the PE neck).
_base_ = ['./depthformer_v.py']
_swin_t = [64, 96, 192, 384, 768]
model = dict(
    pretrained=None,
    backbone=dict(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24]),
    neck=dict(in_channels=_swin_t, out_channels=_swin_t),
    pe_mask_neck=dict(in_channels=[768, 384, 192, 96, 64]),
    decode_head=dict(in_channels=_swin_t, up_sample_channels=_swin_t))
data = dict(samples_per_gpu=8)
