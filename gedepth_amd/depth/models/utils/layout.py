"""Channels-last execution of the convolutional part of the depther (HAHI neck, PE necks, DenseDepth head, conv stem).

MIOpen's bf16 implicit-GEMM convolutions are NHWC inside: on NCHW tensors every call is bracketed by batched_transpose
kernels (4.9 ms of the KITTI step).  ``to_channels_last(model)`` stores the convolution weights channels-last and makes the
backbone hand its stage outputs over as channels-last maps — which are views of its token matrices — so that 1x1 / 3x3
convolutions, BatchNorm + ReLU, bias + LeakyReLU, bilinear resizes, the HAHI token <-> map glue (now views or row
concatenations) and the deformable attention all work on ONE layout (gedepth_amd/csrc/nhwc.hip).  Results are the same
function of the same parameters; state-dict keys and shapes are unchanged (memory format is not part of a checkpoint).

Call it BEFORE ``build_optimizer``: the flat parameter arena keeps each convolution weight's channels-last order."""
import torch
import torch.nn as nn


def to_channels_last(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d) and m.weight.shape[1] > 1:
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    backbone = getattr(model, 'backbone', None)
    if backbone is not None and hasattr(backbone, 'channels_last'):
        backbone.channels_last = True
    return model
