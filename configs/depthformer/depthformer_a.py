# GEDepth-Adaptive = Vanilla + the 11-way slope-logit neck.
_base_ = ['./depthformer_v.py']
model = dict(dynamic_pe_neck=dict(type='DynamicPENeckSOFT'))
