from .compose import Compose
from .formating import Collect, DefaultFormatBundle, ImageToTensor, to_tensor
from .loading import (DDADDepthLoadAnnotations, DepthLoadAnnotations, LoadDDADCamIntrinsic, LoadDDADImageFromFile,
                      LoadImageFromFile, LoadKITTICamIntrinsic)
from .test_time_aug import MultiScaleFlipAug
from .transforms import ColorAug, DDADResize, KBCrop, Normalize, Padding, RandomCrop, RandomFlip, RandomRotate, Resize

__all__ = ['Compose', 'Collect', 'DefaultFormatBundle', 'ImageToTensor', 'to_tensor', 'DepthLoadAnnotations',
           'LoadImageFromFile', 'LoadKITTICamIntrinsic', 'DDADDepthLoadAnnotations', 'LoadDDADCamIntrinsic',
           'LoadDDADImageFromFile', 'DDADResize', 'MultiScaleFlipAug', 'ColorAug', 'KBCrop', 'Normalize', 'Padding',
           'RandomCrop', 'RandomFlip', 'RandomRotate', 'Resize']
