# GEDepth-Vanilla, DepthFormer Swin-L, DDAD (resolves to the same model / data / schedule as the reference's
# configs/depthformer/depthformer_v_ddad.py).
_base_ = ['../_base_/models/depthformer_swin.py', '../_base_/default_runtime.py',
          '../_base_/datasets/ddad_gedepth.py', '../_base_/schedules/gedepth_adamw_cosine.py']
_swin_l = [64, 192, 384, 768, 1536]
model = dict(
    pretrained=None,
    depth_scale=250,
    backbone=dict(embed_dims=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7, USEPE=True),
    neck=dict(type='HAHIHeteroNeck', positional_encoding=dict(type='SinePositionalEncoding', num_feats=256),
              in_channels=_swin_l, out_channels=_swin_l, embedding_dim=512, scales=[1, 1, 1, 1, 1]),
    pe_mask_neck=dict(type='LightPEMASKNeck'),
    decode_head=dict(type='DenseDepthHead', act_cfg=dict(type='LeakyReLU', inplace=True), in_channels=_swin_l,
                     up_sample_channels=_swin_l, channels=64, min_depth=1e-3, max_depth=200))
lr_config = dict(policy='CosineAnnealing', min_lr_ratio=1e-8, by_epoch=False)
runner = dict(type='IterBasedRunner', max_iters=38400)
log_config = dict(_delete_=True, interval=50,
                  hooks=[dict(type='TextLoggerHook', by_epoch=False), dict(type='TensorboardLoggerHook')])
