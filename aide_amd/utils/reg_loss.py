"""Pixelcoreg_Focalloss / Pixelcoreg_Focalloss_twomodel (utils/reg_loss.py:58-193) on device: one map kernel,
the per-image "k smallest" selection (aide_select_smallest, no host argsort) and one backward kernel."""
import torch
from torch import nn

from . import _seg
from .coteach_loss import _select
from .._lib import lib, check
from ..ops import stream_ptr, ptr


class _PixelCoregFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, z3, tg, t_bs, forget_rate, kd, reduction):
        n, _, h, w = z1.shape
        hw = h * w
        dev = z1.device
        key = torch.empty(n * hw, device=dev, dtype=torch.float32)
        val = torch.empty(n * hw, device=dev, dtype=torch.float32) if z3 is not None else key
        tf = torch.empty(n * hw, device=dev, dtype=torch.float32)
        check(lib.aide_pixelcoreg_map(ptr(z1), ptr(z2), ptr(z3), ptr(tg), t_bs, n, hw, float(kd), ptr(key),
                                      ptr(val) if z3 is not None else None, ptr(tf), stream_ptr()), 'pixelcoreg_map')
        keep = int((1 - forget_rate) * hw)                 # reg_loss.py:104-105
        mask, sums, _ = _select(key, val, n, hw, k_host=keep)
        _, tsel, _ = _select(key, tf, n, hw, k_host=keep)
        ctx.save_for_backward(z1, z2, z3 if z3 is not None else z1, tg, mask)
        ctx.three, ctx.t_bs, ctx.kd = z3 is not None, t_bs, float(kd)
        ctx.scale = 1.0 / (n * keep) if (reduction == 'mean' and keep > 0) else (float('nan') if reduction == 'mean' else 1.0)
        frac = (tsel.sum() / tf.double().sum()).float()     # kept foreground fraction (:127), a metric
        ctx.mark_non_differentiable(frac)
        return (sums.sum() * ctx.scale).float(), frac

    @staticmethod
    def backward(ctx, g, _gf):
        z1, z2, z3, tg, mask = ctx.saved_tensors
        n, _, h, w = z1.shape
        coeff = (g.reshape(1).float() * ctx.scale).contiguous()
        if ctx.three:
            g3 = torch.empty_like(z3)
            check(lib.aide_pixelcoreg_bwd(ptr(z1), ptr(z2), ptr(z3), ptr(tg), ctx.t_bs, n, h * w, ctx.kd, ptr(mask),
                                          ptr(coeff), None, None, ptr(g3), stream_ptr()), 'pixelcoreg_bwd')
            return None, None, g3, None, None, None, None, None
        g1, g2 = torch.empty_like(z1), torch.empty_like(z2)
        check(lib.aide_pixelcoreg_bwd(ptr(z1), ptr(z2), None, ptr(tg), ctx.t_bs, n, h * w, ctx.kd, ptr(mask), ptr(coeff),
                                      ptr(g1), ptr(g2), None, stream_ptr()), 'pixelcoreg_bwd')
        return g1, g2, None, None, None, None, None, None


class _Base(nn.Module):
    def __init__(self, smooth=1.0, reduction='mean'):
        super(_Base, self).__init__()
        if reduction not in ('mean', 'sum'):
            raise NotImplementedError("aide_amd.Pixelcoreg_Focalloss*: reduction 'mean' or 'sum' (the reference's 'none' "
                                      "returns the kept values in sorted order)")
        self.smooth, self.reduction = smooth, reduction


class Pixelcoreg_Focalloss(_Base):
    """utils/reg_loss.py:58-131.  forward(inputs1, inputs2, inputs3, targets, forget_rate, kdweight, device)
    -> (loss, kept foreground fraction); only inputs3 receives a gradient, as in the reference."""

    def forward(self, inputs1, inputs2, inputs3, targets, forget_rate, kdweight, device=None):
        z1, z2, z3 = (_seg._logits2(x, 'Pixelcoreg_Focalloss') for x in (inputs1, inputs2, inputs3))
        tg, t_bs = _seg._targets(targets, z1)
        return _PixelCoregFn.apply(z1, z2, z3, tg, t_bs, forget_rate, kdweight, self.reduction)


class Pixelcoreg_Focalloss_twomodel(_Base):
    """utils/reg_loss.py:133-193.  forward(inputs1, inputs2, targets, forget_rate, kdweight, device)."""

    def forward(self, inputs1, inputs2, targets, forget_rate, kdweight, device=None):
        z1, z2 = _seg._logits2(inputs1, 'Pixelcoreg_Focalloss_twomodel'), _seg._logits2(inputs2, 'Pixelcoreg_Focalloss_twomodel')
        tg, t_bs = _seg._targets(targets, z1)
        return _PixelCoregFn.apply(z1, z2, None, tg, t_bs, forget_rate, kdweight, self.reduction)
