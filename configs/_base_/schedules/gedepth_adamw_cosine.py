# AdamW + cosine schedule shared by the GEDepth configs (iteration based).
max_lr = 1e-4
optimizer = dict(
    type='AdamW', lr=max_lr, betas=(0.9, 0.999), weight_decay=0.01,
    paramwise_cfg=dict(custom_keys={
        'absolute_pos_embed': dict(decay_mult=0.),
        'relative_position_bias_table': dict(decay_mult=0.),
        'norm': dict(decay_mult=0.)}))
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
evaluation = dict(by_epoch=False, start=0, interval=800, pre_eval=True, rule='less', save_best='abs_rel',
                  greater_keys=("a1", "a2", "a3"), less_keys=("abs_rel", "rmse"))
checkpoint_config = dict(by_epoch=False, max_keep_ckpts=2, interval=800)
