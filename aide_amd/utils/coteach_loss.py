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


# ---- remaining operators of utils/coteach_loss.py -------------------------------------
class _KLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2):
        n, c, h, w = z1.shape
        out = torch.empty(n, h, w, device=z1.device, dtype=torch.float32)
        if c == 2:
            check(lib.aide_kl_map(ptr(z1), 2 * h * w, ptr(z2), 2 * h * w, n, h * w, ptr(out), None, None, 0, None, 0,
                                  stream_ptr()), 'kl_map')
        else:
            check(lib.aide_kl_map_mc(ptr(z1), c * h * w, ptr(z2), c * h * w, c, n, h * w, ptr(out), None, None, 0, None, 0,
                                     stream_ptr()), 'kl_map_mc')
        ctx.save_for_backward(z1, z2)
        return out

    @staticmethod
    def backward(ctx, g):
        z1, z2 = ctx.saved_tensors
        n, c, h, w = z1.shape
        g = g.contiguous().float()
        g1, g2 = torch.empty_like(z1), torch.empty_like(z2)
        if c == 2:
            check(lib.aide_kl_map(ptr(z1), 2 * h * w, ptr(z2), 2 * h * w, n, h * w, None, ptr(g), ptr(g1), 2 * h * w,
                                  ptr(g2), 2 * h * w, stream_ptr()), 'kl_map bwd')
        else:
            check(lib.aide_kl_map_mc(ptr(z1), c * h * w, ptr(z2), c * h * w, c, n, h * w, None, ptr(g), ptr(g1), c * h * w,
                                     ptr(g2), c * h * w, stream_ptr()), 'kl_map_mc bwd')
        return g1, g2


def KLbidirection(inputs1, inputs2):
    """utils/coteach_loss.py:85-92: per-pixel KL(p1||p2) + KL(p2||p1) of the two softmax maps -> [N,H,W]."""
    z1, z2 = _seg._logits(inputs1), _seg._logits(inputs2)
    if z1.shape != z2.shape:
        raise RuntimeError('KLbidirection: logits of different shapes %s / %s' % (tuple(z1.shape), tuple(z2.shape)))
    return _KLFn.apply(z1, z2)


def _select(sel_vals, sum_vals, nseg, m, k_host=-1, rr=-1.0, k_in=None, only_positive=False):
    dev = sel_vals.device
    mask = torch.empty(nseg * m, device=dev, dtype=torch.uint8)
    sums = torch.empty(nseg, device=dev, dtype=torch.float64)
    ks = torch.empty(nseg, device=dev, dtype=torch.int64)
    check(lib.aide_select_smallest(ptr(sel_vals), ptr(sum_vals), m, nseg, m, int(k_host), float(rr), ptr(k_in),
                                   int(only_positive), ptr(mask), ptr(sums), ptr(ks), stream_ptr()), 'select_smallest')
    return mask, sums, ks


class _RegionCEFn(torch.autograd.Function):
    """mean over images and kept regions of the region CE of `z`, the kept set chosen by the OTHER net's losses."""

    @staticmethod
    def forward(ctx, z, aux, loss_self, loss_other, keep, win=None):
        n, _, h, w = z.shape
        p = loss_self.numel() // n
        mask, sums, _ = _select(loss_other, loss_self, n, p, k_host=keep)
        ctx.save_for_backward(z, aux, mask)
        ctx.denom = float(n * keep)
        ctx.win = win
        return (sums.sum() / ctx.denom).float() if keep > 0 else sums.sum().float() * float('nan')

    @staticmethod
    def backward(ctx, g):
        z, aux, mask = ctx.saved_tensors
        n, c, h, w = z.shape
        coeff = (g.reshape(1).float() / ctx.denom).contiguous()
        dz = torch.empty_like(z)
        if ctx.win is not None:
            check(lib.aide_region_ce_bwd_win(ptr(z), c * h * w, ptr(aux), ptr(mask), ptr(coeff), c, n, h, w, ctx.win[0],
                                             ctx.win[1], ptr(dz), c * h * w, stream_ptr()), 'region_ce_bwd_win')
        elif c == 2:
            check(lib.aide_region_ce_bwd(ptr(z), 2 * h * w, ptr(aux), ptr(mask), ptr(coeff), n, h, w, ptr(dz), 2 * h * w,
                                         stream_ptr()), 'region_ce_bwd')
        else:
            check(lib.aide_region_ce_bwd_mc(ptr(z), c * h * w, ptr(aux), ptr(mask), ptr(coeff), c, n, h, w, ptr(dz),
                                            c * h * w, stream_ptr()), 'region_ce_bwd_mc')
        return dz, None, None, None, None, None


class Coteachingloss_dropregionce(nn.Module):
    """utils/coteach_loss.py:163-196.  Cross entropy on max-pooled regions (logits per class, targets): window = stride =
    (int(H / int(H * scale)), int(W / int(W * scale))), ceil_mode (:171-174; the default scale 0.5 gives 2x2 windows, which
    have kernels of their own); per image the `int((1 - forget_rate) * P)` regions with the smallest loss of the other net
    are kept; mean over all kept."""

    def __init__(self, scale=0.5, reduction='none'):
        super(Coteachingloss_dropregionce, self).__init__()
        if reduction != 'none':
            raise RuntimeError("Coteachingloss_dropregionce needs reduction='none' (the reference's .view(N, -1), "
                               "utils/coteach_loss.py:178, fails on a reduced loss)")
        self.scale = scale

    def forward(self, inputs1, inputs2, targets, forget_rate):
        z1, z2 = _seg._logits(inputs1), _seg._logits(inputs2)
        if z1.shape != z2.shape:
            raise RuntimeError('Coteachingloss_dropregionce: logits of different shapes')
        tg, t_bs = _seg._targets(targets, z1)
        n, c, h, w = z1.shape
        ph_, pw_ = int(h * self.scale), int(w * self.scale)              # the reference's patch_w / patch_h (:172)
        if ph_ < 1 or pw_ < 1:
            raise ZeroDivisionError('Coteachingloss_dropregionce: scale %r leaves no patch (the reference divides by zero)' % (self.scale,))
        kh, kw = int(h / ph_), int(w / pw_)
        dev = z1.device
        if (kh, kw) != (2, 2) or h % 2 or w % 2:
            # any other window (and odd sizes, whose last window is clipped): the general kernels
            p = ((h + kh - 1) // kh) * ((w + kw - 1) // kw)
            keep = int((1 - forget_rate) * p)
            with torch.no_grad():
                l1, l2 = torch.empty(n * p, device=dev), torch.empty(n * p, device=dev)
                a1 = torch.empty(n * p * (c + 1), device=dev, dtype=torch.int32)
                a2 = torch.empty_like(a1)
                for z, l, a in ((z1, l1, a1), (z2, l2, a2)):
                    check(lib.aide_region_ce_fwd_win(ptr(z), c * h * w, ptr(tg), t_bs, c, n, h, w, kh, kw, 255, ptr(l), ptr(a),
                                                     stream_ptr()), 'region_ce_fwd_win')
            return (_RegionCEFn.apply(z1, a1, l1, l2, keep, (kh, kw)), _RegionCEFn.apply(z2, a2, l2, l1, keep, (kh, kw)))
        p = (h // 2) * (w // 2)
        keep = int((1 - forget_rate) * p)
        with torch.no_grad():
            l1, l2 = torch.empty(n * p, device=dev), torch.empty(n * p, device=dev)
            adt = torch.uint8 if c == 2 else torch.int32           # arg-max bookkeeping: a byte for two classes, a word for C
            a1, a2 = torch.empty(n * p, device=dev, dtype=adt), torch.empty(n * p, device=dev, dtype=adt)
            for z, l, a in ((z1, l1, a1), (z2, l2, a2)):
                if c == 2:
                    check(lib.aide_region_ce_fwd(ptr(z), 2 * h * w, ptr(tg), t_bs, n, h, w, 255, ptr(l), ptr(a),
                                                 stream_ptr()), 'region_ce_fwd')
                else:
                    check(lib.aide_region_ce_fwd_mc(ptr(z), c * h * w, ptr(tg), t_bs, c, n, h, w, 255, ptr(l), ptr(a),
                                                    stream_ptr()), 'region_ce_fwd_mc')
        return _RegionCEFn.apply(z1, a1, l1, l2, keep), _RegionCEFn.apply(z2, a2, l2, l1, keep)


class _DropPixelFn(torch.autograd.Function):
    """mean of the `keep2` smallest positive values of target * (KL(z1, z2) + CE(z_which, target)) over the dropped
    images idx (device int64); keep2 = int(rr * #positive) or the count handed over from the other branch."""

    @staticmethod
    def forward(ctx, z1, z2, tg, t_bs, idx, which, rr, k_in):
        n, c, h, w = z1.shape
        hw, nd = h * w, idx.numel()
        v = torch.empty(nd * hw, device=z1.device, dtype=torch.float32)
        if c == 2:
            check(lib.aide_droppixel_map(ptr(z1), 2 * hw, ptr(z2), 2 * hw, ptr(tg), t_bs, ptr(idx), nd, hw, which, ptr(v),
                                         stream_ptr()), 'droppixel_map')
        else:
            check(lib.aide_droppixel_map_mc(ptr(z1), c * hw, ptr(z2), c * hw, ptr(tg), t_bs, ptr(idx), nd, c, hw, which, 255,
                                            ptr(v), stream_ptr()), 'droppixel_map_mc')
        mask, sums, ks = _select(v, v, 1, nd * hw, rr=(-1.0 if k_in is not None else rr), k_in=k_in,
                                 k_host=0, only_positive=True)
        ctx.save_for_backward(z1, z2, tg, idx, mask, ks)
        ctx.t_bs, ctx.which = t_bs, which
        ctx.mark_non_differentiable(ks)
        return (sums[0] / ks[0].double()).float(), ks          # 0 / 0 -> nan like torch.mean of an empty tensor

    @staticmethod
    def backward(ctx, g, _gk):
        z1, z2, tg, idx, mask, ks = ctx.saved_tensors
        n, c, h, w = z1.shape
        hw, nd = h * w, idx.numel()
        coeff = (g.reshape(1).double() / ks[0].double()).float().contiguous()
        g1, g2 = torch.zeros_like(z1), torch.zeros_like(z2)
        if c == 2:
            check(lib.aide_droppixel_bwd(ptr(z1), 2 * hw, ptr(z2), 2 * hw, ptr(tg), ctx.t_bs, ptr(idx), nd, hw, ctx.which,
                                         ptr(mask), ptr(coeff), ptr(g1), ptr(g2), stream_ptr()), 'droppixel_bwd')
        else:
            check(lib.aide_droppixel_bwd_mc(ptr(z1), c * hw, ptr(z2), c * hw, ptr(tg), ctx.t_bs, ptr(idx), nd, c, hw,
                                            ctx.which, 255, ptr(mask), ptr(coeff), ptr(g1), ptr(g2), stream_ptr()),
                  'droppixel_bwd_mc')
        return g1, g2, None, None, None, None, None, None


class Coteachingloss_dropimagedroppixel(_CoteachBase):
    """utils/coteach_loss.py:198-254: Coteachingloss_dropimage plus 0.25 x the pixel-level term on the dropped
    images.  Reference quirks kept: branch 2 keeps as many pixels as branch 1 (`num_remember2`, :249); a dropped
    set without positive foreground values gives nan (torch.mean of an empty tensor)."""
    variant = 1

    def forward(self, inputs1, inputs2, targets, forget_rate):
        l1, l2 = super(Coteachingloss_dropimagedroppixel, self).forward(inputs1, inputs2, targets, forget_rate)
        n = inputs1.shape[0]
        keep = _keep_count(forget_rate, n)
        if keep >= n:                                       # nothing dropped: both extra terms are 0.0
            return l1, l2
        z1, z2 = _seg._logits(inputs1), _seg._logits(inputs2)
        tg, t_bs = _seg._targets(targets, z1)
        rr = 1 - forget_rate
        d1 = self.last['argsort1'][keep:].contiguous()      # images dropped by net 1's ranking
        d2 = self.last['argsort2'][keep:].contiguous()
        drop1, k2 = _DropPixelFn.apply(z1, z2, tg, t_bs, d2, 0, rr, None)
        drop2, _ = _DropPixelFn.apply(z1, z2, tg, t_bs, d1, 1, rr, k2)
        return l1 + 0.25 * drop1, l2 + 0.25 * drop2


def pseudo_label_ensemble(aug_logits, temperature=1.0):
    """mean softmax over the (reverse-augmented) passes -> sharpen -> weightmap
    (trainchaos_proposed_30cases1labeled.py:274-292). Returns (pseudo_label [N,C,H,W], weightmap [N,1,H,W])."""
    lgs = [_seg._logits(t.detach()) for t in aug_logits]
    n, c, h, w = lgs[0].shape
    if any(t.shape != lgs[0].shape for t in lgs):
        raise RuntimeError('aide_amd: the passes of a pseudo-label ensemble differ in shape')
    pl = torch.empty(n, c, h, w, device=lgs[0].device, dtype=torch.float32)
    wm = torch.empty(n, 1, h, w, device=lgs[0].device, dtype=torch.float32)
    arr = (ctypes.c_void_p * len(lgs))(*[t.data_ptr() for t in lgs])
    if c == 2:
        check(lib.aide_pseudo_label(arr, len(lgs), 2 * h * w, n, h * w, float(temperature), ptr(pl), ptr(wm),
                                    stream_ptr()), 'pseudo_label')
    else:
        check(lib.aide_pseudo_label_mc(arr, len(lgs), c, c * h * w, n, h * w, float(temperature), ptr(pl), ptr(wm),
                                       stream_ptr()), 'pseudo_label_mc')
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
