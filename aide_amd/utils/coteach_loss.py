"""Co-teaching loss operators with the reference's names and signatures
(utils/coteach_loss.py:94-161), plus the fused form of the inline selection used by the real AIDE
loop (train_files/trainchaos_proposed_30cases1labeled.py:274-321)."""
import torch
from torch import nn

from . import _seg
from .._lib import lib, check
from ..ops import stream_ptr, ptr
import ctypes


def _keep_count(forget_rate, n):
    return int((1 - forget_rate) * n)          # coteach_loss.py:107-108 (truncation)


class _CoteachBase(nn.Module):
    variant = 1

    def __init__(self, weight=1.0, reduction='mean'):
        super(_CoteachBase, self).__init__()
        if reduction != 'none':
            # the reference raises IndexError from torch.mean(dim=[1,2]) on a reduced scalar
            raise IndexError("Coteachingloss_* only work with reduction='none' (reference behaviour, "
                             "utils/coteach_loss.py:102)")
        self.weight = float(weight)

    def forward(self, inputs1, inputs2, targets, forget_rate):
        n = inputs1.shape[0]
        keep = _keep_count(forget_rate, n)
        if self.variant == 2:
            drop = n - keep
            if drop > 0 and not (keep == drop or keep == 1 or drop == 1):
                raise RuntimeError('Coteachingloss_weightimage: keep (%d) and drop (%d) sets do not broadcast '
                                   '(same failure as the reference, utils/coteach_loss.py:142-145)' % (keep, drop))
        l1, l2, self.last = _seg.coteach_loss(inputs1, inputs2, targets, targets, self.variant, keep,
                                              self.weight, 1.0)
        return l1, l2


class Coteachingloss_dropimage(_CoteachBase):
    variant = 1


class Coteachingloss_weightimage(_CoteachBase):
    variant = 2


def pseudo_label_ensemble(aug_logits, temperature=1.0):
    """mean softmax over the (reverse-augmented) passes -> sharpen -> weightmap
    (trainchaos_proposed_30cases1labeled.py:274-292). Returns (pseudo_label [N,2,H,W], weightmap [N,1,H,W])."""
    lgs = [_seg._logits(t.detach()) for t in aug_logits]
    n, _, h, w = lgs[0].shape
    pl = torch.empty(n, 2, h, w, device=lgs[0].device, dtype=torch.float32)
    wm = torch.empty(n, 1, h, w, device=lgs[0].device, dtype=torch.float32)
    arr = (ctypes.c_void_p * len(lgs))(*[t.data_ptr() for t in lgs])
    check(lib.aide_pseudo_label(arr, len(lgs), 2 * h * w, n, h * w, float(temperature), ptr(pl), ptr(wm),
                                stream_ptr()), 'pseudo_label')
    return pl, wm


class CoTeachingProposedLoss(nn.Module):
    """Fused form of trainchaos_proposed_30cases1labeled.py:303-321: both nets' per-image CE+Dice,
    the two ascending sorts, keep/drop split, consistency MSE on the dropped set and the composite
    losses come out of two statistics passes and one finalize launch.

    forward(outputs1, outputs2, targets1, targets2, pseudo1, wmap1, pseudo2, wmap2, rate)
        -> (loss1, loss2, indx1, indx2)     net1 is scored against targets2 / pseudo2 / wmap2 (:303,:311)
    """

    def __init__(self, cediceweight=None, ceclassweight=None, segcor_weight=(1.0, 10.0), keep=2):
        super(CoTeachingProposedLoss, self).__init__()
        self.w_ce, self.w_dice = (1.0, 1.0) if cediceweight is None else (float(cediceweight[0]), float(cediceweight[1]))
        self.class_w = _seg.class_weights(ceclassweight)
        self.w_seg, self.w_cor = float(segcor_weight[0]), float(segcor_weight[1])
        self.keep = keep

    def forward(self, outputs1, outputs2, targets1, targets2, pseudo1, wmap1, pseudo2, wmap2, rate):
        l1, l2, self.last = _seg.coteach_loss(outputs1, outputs2, targets2, targets1, 0, self.keep, self.w_ce,
                                              self.w_dice, 1.0, float(rate), self.w_seg, self.w_cor,
                                              pseudo2, wmap2, pseudo1, wmap1, self.class_w)
        return l1, l2, self.last['argsort1'], self.last['argsort2']
