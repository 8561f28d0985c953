# KITTI Eigen split with the 5-channel ground-embedding loader (RGB + filtered pe + raw pe).
USEPE_FLAGS = True
dataset_type = 'KITTIDataset'
data_root = 'data/kitti'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
crop_size = (352, 704)
_meta_keys = ('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor',
              'flip', 'flip_direction', 'img_norm_cfg', 'cam_intrinsic')
train_pipeline = [
    dict(type='LoadImageFromFile', USEPE=USEPE_FLAGS),
    dict(type='DepthLoadAnnotations'),
    dict(type='LoadKITTICamIntrinsic'),
    dict(type='KBCrop', depth=True, pe_k=True),
    dict(type='Resize', ratio_range=(0.5, 2.0)),
    dict(type='Padding', img_padding_value=(0, 0, 0), depth_padding_value=255, pe_k=True),
    dict(type='RandomRotate', prob=0.5, degree=2.5),
    dict(type='RandomFlip', prob=0.5),
    dict(type='RandomCrop', crop_size=(352, 704)),
    dict(type='ColorAug', prob=0.5, gamma_range=[0.9, 1.1], brightness_range=[0.9, 1.1], color_range=[0.9, 1.1]),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'depth_gt', 'pe_ori_point', 'pe_k_gt'], meta_keys=_meta_keys),
]
test_pipeline = [
    dict(type='LoadImageFromFile', USEPE=USEPE_FLAGS),
    dict(type='LoadKITTICamIntrinsic'),
    dict(type='KBCrop', depth=False, pe_k=False),
    dict(type='MultiScaleFlipAug', img_scale=(1216, 352), flip=True, flip_direction='horizontal',
         transforms=[
             dict(type='RandomFlip', direction='horizontal'),
             dict(type='Normalize', **img_norm_cfg),
             dict(type='ImageToTensor', keys=['img']),
             dict(type='Collect', keys=['img', 'pe_ori_point'], meta_keys=_meta_keys),
         ])
]
_split = dict(type=dataset_type, data_root=data_root, img_dir='input', ann_dir='gt_depth', depth_scale=256,
              garg_crop=True, eigen_crop=False, min_depth=1e-3, max_depth=80)
data = dict(
    samples_per_gpu=2,
    workers_per_gpu=2,
    train=dict(split='splits/kitti_eigen_train.txt', pipeline=train_pipeline, **_split),
    val=dict(split='splits/kitti_eigen_test.txt', pipeline=test_pipeline, **_split),
    test=dict(split='splits/kitti_eigen_test.txt', pipeline=test_pipeline, **_split))
