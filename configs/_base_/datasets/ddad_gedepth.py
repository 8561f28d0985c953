# DDAD (cameras 1, 5, 6, 9 at 384 x 640) with the ground-embedding channels, per-camera heights and slope classes.
USEPE_FLAG = True
depth_scale = 250
dataset_type = 'DDADDataset'
data_root = 'data/DDAD'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
_meta_keys = ('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip', 'flip_direction',
              'img_norm_cfg',)
train_pipeline = [
    dict(type='LoadDDADImageFromFile', USEPE=USEPE_FLAG, USE_DYNAMIC_PE=True),
    dict(type='DDADDepthLoadAnnotations', USE_DYNAMIC_PE=True),
    dict(type='DDADResize', shape=(384, 640), USE_DYNAMIC_PE=True),
    dict(type='Resize', ratio_range=(0.5, 2.0)),
    dict(type='Padding', img_padding_value=(0, 0, 0), depth_padding_value=255, pe_k=True, ori_h=384, ori_w=640),
    dict(type='RandomRotate', prob=0.5, degree=2.5),
    dict(type='RandomFlip', prob=0.0),
    dict(type='RandomCrop', crop_size=(384, 640)),
    dict(type='ColorAug', prob=0.5, gamma_range=[0.9, 1.1], brightness_range=[0.9, 1.1], color_range=[0.9, 1.1]),
    dict(type='Normalize', depth_scale=depth_scale, **img_norm_cfg),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'depth_gt', 'pe_k_gt', 'height'], meta_keys=_meta_keys),
]
test_pipeline = [
    dict(type='LoadDDADImageFromFile', USEPE=USEPE_FLAG, USE_DYNAMIC_PE=True),
    dict(type='DDADResize', shape=(384, 640), depth=False),
    dict(type='MultiScaleFlipAug', img_scale=(384, 640), flip=False, flip_direction='horizontal',
         transforms=[
             dict(type='Normalize', depth_scale=depth_scale, **img_norm_cfg),
             dict(type='ImageToTensor', keys=['img']),
             dict(type='Collect', keys=['img', 'height', 'test'], meta_keys=_meta_keys),
         ])
]
_cams = ['CAMERA_%02d' % idx for idx in [1, 5, 6, 9]]
_split = dict(type=dataset_type, min_depth=1e-3, max_depth=200, cameras=_cams)
data = dict(
    samples_per_gpu=4,
    workers_per_gpu=4,
    train_dataloader=dict(shuffle=True, drop_last=True, persistent_workers=False),
    val_dataloader=dict(shuffle=False, persistent_workers=True),
    test_dataloader=dict(shuffle=False, persistent_workers=False),
    train=dict(split='splits/ddad_train_split.txt', pipeline=train_pipeline, **_split),
    val=dict(split='splits/ddad_test_split.txt', pipeline=test_pipeline, **_split),
    test=dict(split='splits/ddad_test_split.txt', pipeline=test_pipeline, **_split))
