_base_ = ['./depthformer_swint_v.py']
model = dict(dynamic_pe_neck=dict(type='DynamicPENeckSOFT', in_channels=[768, 384, 192, 96, 64]))
