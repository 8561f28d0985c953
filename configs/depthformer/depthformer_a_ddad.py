# GEDepth-Adaptive on DDAD = Vanilla + the 11-way slope-logit neck (reference: configs/depthformer/depthformer_a_ddad.py).
_base_ = ['./depthformer_v_ddad.py']
model = dict(dynamic_pe_neck=dict(type='DynamicPENeckSOFT'))
