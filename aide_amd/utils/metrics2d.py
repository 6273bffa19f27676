"""Binary segmentation metrics of utils/metrics2d.py:8-84 on device: one pass of the fused statistics kernel gives, per
image, the hard-prediction count (softmax p1 >= 0.5), the target count and their intersection; every metric below is a
handful of scalar operations on those three numbers.  Results stay on the device (no host sync) unless the reference
itself returns Python numbers (Dice_fn_Nozero)."""
import torch

from . import _seg


def _counts(inputs, targets, threshold):
    if threshold != 0.5:
        raise NotImplementedError('aide_amd metrics implement the reference default threshold 0.5')
    with torch.no_grad():
        _, extra = _seg.seg_loss(inputs.detach(), targets, 1.0, 1.0, 255, 2, 1.0, 1.0, 1.0)
    st = extra['stats']
    hw = float(inputs.shape[2] * inputs.shape[3])
    return extra, st[:, _seg.S_HP], st[:, _seg.S_T], st[:, _seg.S_HI], hw


def Dice_fn(inputs, targets, threshold=0.5):
    """metrics2d.py:8-29: hard Dice summed over the batch (empty target: 1 if the prediction is empty too, else 0)."""
    extra, _, _, _, _ = _counts(inputs, targets, threshold)
    return extra['hard_dice']


def Dice_fn_Nozero(inputs, targets, threshold=0.5):
    """metrics2d.py:31-52: (Dice sum as a Python float, number of images that are not empty in both target and
    prediction).  Syncs, like the reference's .item()."""
    extra, p, t, _, _ = _counts(inputs, targets, threshold)
    count = int(((t != 0) | (p != 0)).sum().item())
    return extra['hard_dice'].item(), count


def TP_TN_FP_FN(inputs, targets, threshold=0.5):
    """metrics2d.py:54-70: confusion counts of the LAST image of the batch (the reference's loop overwrites them per
    image), as 0-dim float tensors."""
    _, p, t, i, hw = _counts(inputs, targets, threshold)
    tp, fp, fn = i[-1], p[-1] - i[-1], t[-1] - i[-1]
    tn = hw - p[-1] - t[-1] + i[-1]
    return tp.float(), tn.float(), fp.float(), fn.float()


def IoU_fn(inputs, targets, threshold=0.5):
    """metrics2d.py:72-84: sum over the batch of |P & T| / |P | T| (NaN for an image empty in both, as the reference)."""
    _, p, t, i, _ = _counts(inputs, targets, threshold)
    return (i.float() / (p + t - i).float()).sum()
