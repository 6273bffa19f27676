"""Dice_fn (utils/metrics2d.py:8-29): hard Dice summed over the batch, computed on device by the
fused statistics kernels; returns a 0-dim device tensor (no host sync until the caller's .item())."""
import torch

from . import _seg


def Dice_fn(inputs, targets, threshold=0.5):
    if threshold != 0.5:
        raise NotImplementedError('aide_amd.Dice_fn implements the reference default threshold 0.5')
    with torch.no_grad():
        _, extra = _seg.seg_loss(inputs.detach(), targets, 1.0, 1.0, 255, 2, 1.0, 1.0, 1.0)
    return extra['hard_dice']
