from .hahi import HAHIHeteroNeck, MultiScaleDeformableAttention
from .pe_necks import DynamicPENeckSOFT, LightPEMASKNeck

__all__ = ['HAHIHeteroNeck', 'MultiScaleDeformableAttention', 'LightPEMASKNeck', 'DynamicPENeckSOFT']
