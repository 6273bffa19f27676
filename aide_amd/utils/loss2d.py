"""Drop-in loss modules with the reference's names, constructor and forward signatures
(utils/loss2d.py:5-154), computed by the fused HIP loss kernels (csrc/loss.hip).

Scope: int64 index targets [N,H,W]; two classes (the reference's hot-path configuration) on the specialised kernels of
csrc/loss.hip, 3 .. 8 classes on the general form in csrc/loss_mc.hip (same statistics, same finalize kernels).
`cediceweight` / `ceclassweight` may be CPU tensors exactly as the reference scripts pass them
(trainchaos_comparison_1case.py:157-166); they are read as host scalars at construction."""
import torch
from torch import nn

from . import _seg
from .._lib import lib, check
from ..ops import stream_ptr, ptr

_RED = {'mean': 0, 'sum': 1}


class _CEMap(torch.autograd.Function):
    """reduction='none' cross-entropy map (utils/loss2d.py:8 with reduction='none')."""

    @staticmethod
    def forward(ctx, logits, targets, t_bs, w0, w1, ignore):
        n, c, h, w = logits.shape
        out = torch.empty(n, h, w, device=logits.device, dtype=torch.float32)
        if c == 2:
            check(lib.aide_ce_map(ptr(logits), 2 * h * w, ptr(targets), t_bs, w0, w1, ignore, n, h * w, ptr(out),
                                  None, None, 0, stream_ptr()), 'ce_map')
        else:
            check(lib.aide_ce_map_mc(ptr(logits), c * h * w, ptr(targets), t_bs, _seg.class_w_array(w0, w1, c), c, ignore,
                                     n, h * w, ptr(out), None, None, 0, stream_ptr()), 'ce_map_mc')
        ctx.save_for_backward(logits, targets)
        ctx.cfg = (t_bs, w0, w1, ignore)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, targets = ctx.saved_tensors
        t_bs, w0, w1, ignore = ctx.cfg
        n, c, h, w = logits.shape
        dl = torch.empty_like(logits)
        if c == 2:
            check(lib.aide_ce_map(ptr(logits), 2 * h * w, ptr(targets), t_bs, w0, w1, ignore, n, h * w, None,
                                  ptr(g.contiguous()), ptr(dl), 2 * h * w, stream_ptr()), 'ce_map_bwd')
        else:
            check(lib.aide_ce_map_mc(ptr(logits), c * h * w, ptr(targets), t_bs, _seg.class_w_array(w0, w1, c), c, ignore,
                                     n, h * w, None, ptr(g.contiguous()), ptr(dl), c * h * w, stream_ptr()), 'ce_map_mc_bwd')
        return dl, None, None, None, None, None


class _MSEMap(torch.autograd.Function):
    """(softmax(input) - target)^2, elementwise (utils/loss2d.py:115-117 with reduction='none')."""

    @staticmethod
    def forward(ctx, logits, target):
        n, c, h, w = logits.shape
        out = torch.empty_like(logits)
        if c == 2:
            check(lib.aide_mse_map(ptr(logits), 2 * h * w, ptr(target), 2 * h * w, n, h * w, ptr(out), None, None,
                                   0, stream_ptr()), 'mse_map')
        else:
            check(lib.aide_mse_map_mc(ptr(logits), c * h * w, ptr(target), c * h * w, c, n, h * w, ptr(out), None, None,
                                      0, stream_ptr()), 'mse_map_mc')
        ctx.save_for_backward(logits, target)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        n, c, h, w = logits.shape
        dl = torch.empty_like(logits)
        if c == 2:
            check(lib.aide_mse_map(ptr(logits), 2 * h * w, ptr(target), 2 * h * w, n, h * w, None,
                                   ptr(g.contiguous()), ptr(dl), 2 * h * w, stream_ptr()), 'mse_map_bwd')
        else:
            check(lib.aide_mse_map_mc(ptr(logits), c * h * w, ptr(target), c * h * w, c, n, h * w, None,
                                      ptr(g.contiguous()), ptr(dl), c * h * w, stream_ptr()), 'mse_map_mc_bwd')
        return dl, None


class _DiceTerms(torch.autograd.Function):
    """K = 1: DiceLoss on a probability input [N,H,W] (utils/loss2d.py:47-61); K = 2: MulticlassDiceLoss with one-hot
    targets [N,2,H,W] and class weights on logits [N,2,H,W] (utils/loss2d.py:96-104)."""

    @staticmethod
    def forward(ctx, x, t, k, w0, w1, smooth, red):
        n = x.shape[0]
        hw = x.shape[-1] * x.shape[-2]
        ws = torch.empty(lib.aide_dice_terms_ws_bytes(n, hw) // 8, device=x.device, dtype=torch.float64)
        per = torch.empty(n, device=x.device, dtype=torch.float32)
        out = torch.empty(n if red == 2 else 1, device=x.device, dtype=torch.float32)
        check(lib.aide_dice_terms_fwd(ptr(x), k * hw, ptr(t), k * hw, n, hw, k, w0, w1, smooth, red, ptr(ws), ptr(per),
                                      ptr(out), stream_ptr()), 'dice_terms_fwd')
        ctx.save_for_backward(x, t, ws)
        ctx.cfg = (k, w0, w1, smooth, red, hw)
        return out if red == 2 else out[0]

    @staticmethod
    def backward(ctx, g):
        x, t, ws = ctx.saved_tensors
        k, w0, w1, smooth, red, hw = ctx.cfg
        g = g.contiguous().float().reshape(-1)
        dx = torch.empty_like(x)
        check(lib.aide_dice_terms_bwd(ptr(x), k * hw, ptr(t), k * hw, x.shape[0], hw, k, w0, w1, smooth, red, ptr(ws),
                                      ptr(g), ptr(dx), k * hw, stream_ptr()), 'dice_terms_bwd')
        return dx, None, None, None, None, None, None


class _DiceTermsMC(torch.autograd.Function):
    """MulticlassDiceLoss with one-hot targets [N,C,H,W], C = 3 .. 8 (utils/loss2d.py:96-104): csrc/loss_mc.hip."""

    @staticmethod
    def forward(ctx, x, t, cw, smooth, red):
        n, c, h, w = x.shape
        hw = h * w
        ws = torch.empty(lib.aide_dice_terms_mc_ws_bytes(n, hw, c) // 8, device=x.device, dtype=torch.float64)
        per = torch.empty(n, device=x.device, dtype=torch.float32)
        out = torch.empty(n if red == 2 else 1, device=x.device, dtype=torch.float32)
        check(lib.aide_dice_terms_mc_fwd(ptr(x), c * hw, ptr(t), c * hw, n, hw, c, cw, smooth, red, ptr(ws), ptr(per),
                                         ptr(out), stream_ptr()), 'dice_terms_mc_fwd')
        ctx.save_for_backward(x, t, ws)
        ctx.cfg = (cw, smooth, red)
        return out if red == 2 else out[0]

    @staticmethod
    def backward(ctx, g):
        x, t, ws = ctx.saved_tensors
        cw, smooth, red = ctx.cfg
        n, c, h, w = x.shape
        g = g.contiguous().float().reshape(-1)
        dx = torch.empty_like(x)
        check(lib.aide_dice_terms_mc_bwd(ptr(x), c * h * w, ptr(t), c * h * w, n, h * w, c, cw, smooth, red, ptr(ws),
                                         ptr(g), ptr(dx), c * h * w, stream_ptr()), 'dice_terms_mc_bwd')
        return dx, None, None, None, None


def _dense_f32(t, shape, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('aide_amd losses run on a HIP device only; there is no CPU fallback')
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError('aide_amd: %s shape %s, expected %s' % (what, tuple(t.shape), tuple(shape)))
    return t.float().contiguous()          # the reference's `.float()` (loss2d.py:50)


class CrossEntropyLoss2d(nn.Module):
    def __init__(self, weight=None, reduction='mean', ignore_index=255):
        super(CrossEntropyLoss2d, self).__init__()
        self.w0, self.w1 = _seg.class_weights(weight)
        self.reduction, self.ignore_index = reduction, ignore_index

    def forward(self, inputs, targets):
        if self.reduction == 'none':
            logits = _seg._logits(inputs)
            t, t_bs = _seg._targets(targets, logits)
            return _CEMap.apply(logits, t, t_bs, self.w0, self.w1, self.ignore_index)
        loss, self.last = _seg.seg_loss(inputs, targets, self.w0, self.w1, self.ignore_index,
                                        _RED[self.reduction], 1.0, 0.0, 1.0)
        return loss


class DiceLoss(nn.Module):
    def __init__(self, weight=None, smooth=1.0, reduction='mean'):
        super(DiceLoss, self).__init__()
        self.weight, self.smooth, self.reduction = weight, smooth, reduction

    def forward(self, input, target):
        red = 2 if self.reduction == 'none' else _RED[self.reduction]
        if input.dim() <= 3:                # utils/loss2d.py:47-48: a probability map, no softmax
            if input.dtype != torch.float32 or not input.is_cuda:
                raise RuntimeError('aide_amd.DiceLoss: probability input must be an fp32 tensor on a HIP device')
            # the reference flattens both per image (`input.view(N, -1)`, `target.view(N, -1)`, N = target.size(0)): any
            # rank <= 3 input and any target with the same number of elements per image, e.g. [N,1,H,W] beside [N,H,W]
            n = target.shape[0]
            if input.numel() % max(n, 1) or target.numel() != input.numel():
                raise RuntimeError('aide_amd.DiceLoss: input %s and target %s do not flatten to the same [N, -1]'
                                   % (tuple(input.shape), tuple(target.shape)))
            x = input.contiguous().view(n, 1, -1)
            t = _dense_f32(target.reshape(n, 1, -1), x.shape, 'DiceLoss target')
            out = _DiceTerms.apply(x, t, 1, 1.0, 1.0, float(self.smooth), red)
            return out
        loss, self.last = _seg.seg_loss(input, target, 1.0, 1.0, 255, red, 0.0, 1.0 if red else 1.0,
                                        float(self.smooth))
        return loss


class Dice_Loss(nn.Module):
    """utils/loss2d.py:63-85: the logits branch of DiceLoss under a second name."""

    def __init__(self, smooth=1.0, reduction='mean'):
        super(Dice_Loss, self).__init__()
        self.smooth, self.reduction = smooth, reduction

    def forward(self, inputs, targets):
        red = 2 if self.reduction == 'none' else _RED[self.reduction]
        loss, self.last = _seg.seg_loss(inputs, targets, 1.0, 1.0, 255, red, 0.0, 1.0, float(self.smooth))
        return loss


class MulticlassDiceLoss(nn.Module):
    def __init__(self, weight=None, smooth=1.0, reduction='mean'):
        super(MulticlassDiceLoss, self).__init__()
        self.weight, self.smooth, self.reduction = weight, smooth, reduction

    def forward(self, input, target):
        red = 2 if self.reduction == 'none' else _RED[self.reduction]
        if target.dim() > 3:                # utils/loss2d.py:98-104: one Dice term per class, weighted
            logits = _seg._logits(input)
            w0, w1 = _seg.class_weights(self.weight)
            tgt = _dense_f32(target, logits.shape, 'MulticlassDiceLoss one-hot target')
            if logits.shape[1] > 2:
                return _DiceTermsMC.apply(logits, tgt, _seg.class_w_array(w0, w1, logits.shape[1]), float(self.smooth), red)
            if w1 is None:
                raise RuntimeError('aide_amd: %d class weights for logits with 2 classes' % len(w0))
            return _DiceTerms.apply(logits, tgt, 2, w0, w1, float(self.smooth), red)
        # index targets: class-1 Dice only, class weights ignored (utils/loss2d.py:105-106)
        loss, self.last = _seg.seg_loss(input, target, 1.0, 1.0, 255, red, 0.0, 1.0, float(self.smooth))
        return loss


class MulticlassMSELoss(nn.Module):
    def __init__(self, reduction='mean'):
        super(MulticlassMSELoss, self).__init__()
        self.reduction = reduction

    def forward(self, input, target):
        logits = _seg._logits(input)
        tgt = _seg._dense(target, logits.shape, 'MSE target')
        m = _MSEMap.apply(logits, tgt)
        if self.reduction == 'none':
            return m
        return m.mean() if self.reduction == 'mean' else m.sum()    # off the hot path


def _pair(w):
    if w is None:
        return 1.0, 1.0
    return float(w[0]), float(w[1])


class CEMDiceLoss(nn.Module):
    def __init__(self, cediceweight=None, ceclassweight=None, diceclassweight=None, reduction='mean'):
        super(CEMDiceLoss, self).__init__()
        self.w_ce, self.w_dice = _pair(cediceweight)
        self.w0, self.w1 = _seg.class_weights(ceclassweight)
        if reduction not in _RED:
            raise ValueError("CEMDiceLoss: reduction must be 'mean' or 'sum' (the reference's 'none' mixes a "
                             "[N,H,W] map with a [N] vector)")
        self.reduction = reduction

    def forward(self, inputs, targets):
        loss, self.last = _seg.seg_loss(inputs, targets, self.w0, self.w1, 255, _RED[self.reduction],
                                        self.w_ce, self.w_dice, 1.0)
        return loss


class CEMDiceLossImage(nn.Module):
    def __init__(self, cediceweight=None, ceclassweight=None, diceclassweight=None, reduction='mean'):
        super(CEMDiceLossImage, self).__init__()
        self.w_ce, self.w_dice = _pair(cediceweight)
        self.w0, self.w1 = _seg.class_weights(ceclassweight)

    def forward(self, inputs, targets):
        loss, self.last = _seg.seg_loss(inputs, targets, self.w0, self.w1, 255, 2, self.w_ce, self.w_dice, 1.0)
        return loss


class CEDiceLoss(nn.Module):
    """utils/loss2d.py:156-171: CrossEntropyLoss2d(weight=classweight) * cediceweight[0] + DiceLoss * cediceweight[1]
    (DiceLoss ignores its weight argument, loss2d.py:35-60) -- one fused statistics pass here."""

    def __init__(self, cediceweight=None, classweight=None, reduction='mean'):
        super(CEDiceLoss, self).__init__()
        self.w_ce, self.w_dice = _pair(cediceweight)
        self.w0, self.w1 = _seg.class_weights(classweight)
        if reduction not in _RED:
            raise ValueError("CEDiceLoss: reduction must be 'mean' or 'sum' (the reference's 'none' adds a [N,H,W] map "
                             "to a [N] vector)")
        self.reduction = reduction

    def forward(self, inputs, targets):
        loss, self.last = _seg.seg_loss(inputs, targets, self.w0, self.w1, 255, _RED[self.reduction],
                                        self.w_ce, self.w_dice, 1.0)
        return loss
