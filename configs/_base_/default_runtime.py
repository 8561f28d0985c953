log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook', by_epoch=True),
                                      dict(type='TensorboardImageLoggerHook', by_epoch=True)])
dist_params = dict(backend='nccl')      # == RCCL on ROCm
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]
cudnn_benchmark = True
