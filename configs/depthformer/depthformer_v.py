# GEDepth-Vanilla, DepthFormer Swin-L, KITTI (resolves to the same model/optimizer/schedule as the reference's
# configs/depthformer/depthformer_v.py).
_base_ = ['../_base_/models/depthformer_swin.py', '../_base_/default_runtime.py',
          '../_base_/datasets/kitti_gedepth.py', '../_base_/schedules/gedepth_adamw_cosine.py']
_swin_l = [64, 192, 384, 768, 1536]
model = dict(
    pretrained="ckpt/swin_large_patch4_window7_224_22k.pth",
    backbone=dict(embed_dims=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7, num_stages=0,
                  USEPE=True),
    neck=dict(type='HAHIHeteroNeck', positional_encoding=dict(type='SinePositionalEncoding', num_feats=256),
              in_channels=_swin_l, out_channels=_swin_l, embedding_dim=512, scales=[1, 1, 1, 1, 1]),
    pe_mask_neck=dict(type='LightPEMASKNeck'),
    decode_head=dict(type='DenseDepthHead', act_cfg=dict(type='LeakyReLU', inplace=True), in_channels=_swin_l,
                     up_sample_channels=_swin_l, channels=64, min_depth=1e-3, max_depth=80))
lr_config = dict(policy='CosineAnnealing', warmup='linear', warmup_iters=16 * 1600, warmup_ratio=1.0 / 1000,
                 min_lr_ratio=1e-8, by_epoch=False)
runner = dict(type='IterBasedRunner', max_iters=1600 * 48)
log_config = dict(_delete_=True, interval=10,
                  hooks=[dict(type='TextLoggerHook', by_epoch=False), dict(type='TensorboardLoggerHook')])
