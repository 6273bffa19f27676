"""Shared plumbing for the fused segmentation losses: one statistics pass + one finalize launch in
the forward, one elementwise launch in the backward (csrc/loss.hip)."""
import ctypes

import torch

from .._lib import lib, check
from ..ops import stream_ptr, ptr

NS = 8
S_CE, S_W, S_I, S_P, S_T, S_M, S_HP, S_HI = range(8)
MAXC = 8                # aide_seg_max_classes()


def _logits(x):
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError('aide_amd losses run on a HIP device only (got %s); there is no CPU fallback'
                           % (x.device if isinstance(x, torch.Tensor) else type(x)))
    if x.dim() != 4 or not 2 <= x.shape[1] <= MAXC:
        raise NotImplementedError('aide_amd fused losses take logits [N, C, H, W] with 2 <= C <= %d (two classes: the '
                                  'reference configuration, train_files/trainchaos_comparison_1case.py:121, on the '
                                  'specialised kernels of csrc/loss.hip; 3 .. %d on csrc/loss_mc.hip); got %s'
                                  % (MAXC, MAXC, tuple(x.shape)))
    if x.dtype != torch.float32:
        raise RuntimeError('aide_amd: logits must be fp32')
    return x if x.is_contiguous() else x.contiguous()


def _logits2(x, what):
    """Logits for the operators that are binary in the reference itself (Pixelcoreg_Focalloss: utils/reg_loss.py:70-99 reads
    soft-max channels 0 and 1 and weights them with t and 1 - t)."""
    x = _logits(x)
    if x.shape[1] != 2:
        raise NotImplementedError('aide_amd: %s is implemented for num_classes == 2 (got logits %s)'
                                  % (what, tuple(x.shape)))
    return x


def _targets(t, logits):
    """int64 index targets [N,H,W]; a non-contiguous batch stride (the reference's mask[:,1] view,
    SURVEY.md §A.3 item 10) is consumed in place."""
    if t.device != logits.device:
        raise RuntimeError('aide_amd: targets on %s but logits on %s' % (t.device, logits.device))
    if t.dim() > 3:          # one-hot -> arg-max first (utils/loss2d.py:11-12); not on the hot path
        oh = t.float().contiguous()
        n_, c_, h_, w_ = oh.shape
        t = torch.empty(n_, h_, w_, device=oh.device, dtype=torch.int64)
        check(lib.aide_onehot_argmax(ptr(oh), c_ * h_ * w_, n_, c_, h_ * w_, ptr(t), stream_ptr()), 'onehot_argmax')
    if t.dtype != torch.int64:
        raise RuntimeError('aide_amd: targets must be int64 class indices')
    if t.device != logits.device:
        raise RuntimeError('aide_amd: targets on %s but logits on %s' % (t.device, logits.device))
    n, h, w = t.shape
    if (n, h, w) != (logits.shape[0], logits.shape[2], logits.shape[3]):
        raise RuntimeError('aide_amd: targets shape %s does not match logits %s' % (tuple(t.shape), tuple(logits.shape)))
    if not (t.stride(2) == 1 and t.stride(1) == w):
        t = t.contiguous()
    return t, (t.stride(0) if n > 1 else h * w)


def class_weights(weight):
    """The class-weight argument of a loss module as the pair (w0, w1) the two-class entry points take; more than two
    weights travel as (tuple of C floats, None).  None = unweighted, for any class count."""
    if weight is None:
        return 1.0, 1.0
    w = [float(v) for v in weight]
    if len(w) == 2:
        return w[0], w[1]
    if not 3 <= len(w) <= MAXC:
        raise NotImplementedError('aide_amd fused losses support 2 .. %d classes (got %d class weights)' % (MAXC, len(w)))
    return tuple(w), None


def class_w_array(w0, w1, c):
    """HOST float array of the C class weights for the *_mc entry points (csrc/loss_mc.hip)."""
    if w1 is None:
        vals = list(w0)
    elif (w0, w1) == (1.0, 1.0):
        vals = [1.0] * c
    else:
        vals = [w0, w1]
    if len(vals) != c:           # nn.CrossEntropyLoss raises for a weight vector that does not match the class count
        raise RuntimeError('aide_amd: %d class weights for logits with %d classes' % (len(vals), c))
    return (ctypes.c_float * c)(*vals)


def stats_pass(logits, targets, t_bs, w0, w1, ignore_index, pseudo=None, wmap=None):
    n, c, h, w = logits.shape
    hw = h * w
    partials = torch.empty(lib.aide_seg_loss_ws_bytes(n, hw) // 8, device=logits.device, dtype=torch.float64)
    p_bs = c * hw if pseudo is not None else 0
    w_bs = hw if wmap is not None else 0
    if c == 2:
        if w1 is None:
            raise RuntimeError('aide_amd: %d class weights for logits with 2 classes' % len(w0))
        check(lib.aide_seg_stats(ptr(logits), 2 * hw, ptr(targets), t_bs, w0, w1, ignore_index, ptr(pseudo), p_bs,
                                 ptr(wmap), w_bs, n, hw, ptr(partials), stream_ptr()), 'seg_stats')
    else:
        check(lib.aide_seg_stats_mc(ptr(logits), c * hw, ptr(targets), t_bs, class_w_array(w0, w1, c), c, ignore_index,
                                    ptr(pseudo), p_bs, ptr(wmap), w_bs, n, hw, ptr(partials), stream_ptr()),
              'seg_stats_mc')
    return partials


def _dense(x, shape, what):
    if x is None:
        return None
    if tuple(x.shape) != tuple(shape) or x.dtype != torch.float32:
        raise RuntimeError('aide_amd: %s must be fp32 of shape %s, got %s' % (what, tuple(shape), tuple(x.shape)))
    return x if x.is_contiguous() else x.contiguous()


class SegBackward(torch.autograd.Function):
    """Attaches the fused backward kernel to a loss value that was computed by the finalize kernel."""

    @staticmethod
    def forward(ctx, logits, value, pack):
        ctx.pack = pack
        ctx.save_for_backward(logits)
        if callable(value):           # the finalize launch itself: its output tensor is born here, no copy
            return value()
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        pk = ctx.pack
        n, c, h, w = logits.shape
        hw = h * w
        g = g.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        g_stride = 0 if g.numel() == 1 else 1
        dl = torch.empty_like(logits)
        pseudo, wmap = pk.get('pseudo'), pk.get('wmap')
        if c == 2:
            check(lib.aide_seg_loss_bwd(ptr(logits), 2 * hw, ptr(pk['targets']), pk['t_bs'], pk['w0'], pk['w1'],
                                        pk['ignore'], ptr(pseudo), 2 * hw if pseudo is not None else 0, ptr(wmap),
                                        hw if wmap is not None else 0, n, hw, ptr(pk['stats']), ptr(pk['coef']),
                                        pk['smooth'], ptr(g), g_stride, ptr(dl), 2 * hw, stream_ptr()),
                  'seg_loss_bwd')
        else:
            check(lib.aide_seg_loss_bwd_mc(ptr(logits), c * hw, ptr(pk['targets']), pk['t_bs'],
                                           class_w_array(pk['w0'], pk['w1'], c), c, pk['ignore'], ptr(pseudo),
                                           c * hw if pseudo is not None else 0, ptr(wmap),
                                           hw if wmap is not None else 0, n, hw, ptr(pk['stats']), ptr(pk['coef']),
                                           pk['smooth'], ptr(g), g_stride, ptr(dl), c * hw, stream_ptr()),
                  'seg_loss_bwd_mc')
        return dl, None, None


def seg_loss(logits, targets, w0, w1, ignore_index, reduction, w_ce, w_dice, smooth):
    """reduction: 0 mean, 1 sum, 2 per-image. Returns (loss, extras dict)."""
    logits = _logits(logits)
    targets, t_bs = _targets(targets, logits)
    n, _, h, w = logits.shape
    hw = h * w
    dev = logits.device
    with torch.no_grad():
        partials = stats_pass(logits, targets, t_bs, w0, w1, ignore_index)
        stats = torch.empty(n * NS, device=dev, dtype=torch.float64)
        per_image = torch.empty(n, device=dev, dtype=torch.float32)
        idx = torch.empty(n, device=dev, dtype=torch.int64)
        coef = torch.empty(3 * n, device=dev, dtype=torch.float32)
        hard = torch.empty(1, device=dev, dtype=torch.float32)

    def finalize():                   # runs inside SegBackward.forward: the loss tensor is that node's own output
        out = torch.empty(n if reduction == 2 else 1, device=dev, dtype=torch.float32)
        check(lib.aide_seg_loss_finalize(ptr(partials), n, hw, reduction, w_ce, w_dice, smooth, ptr(stats),
                                         ptr(out), ptr(per_image), ptr(idx), ptr(coef), ptr(hard),
                                         stream_ptr()), 'seg_loss_finalize')
        return out if reduction == 2 else out.view(())
    pack = dict(targets=targets, t_bs=t_bs, w0=w0, w1=w1, ignore=ignore_index, stats=stats, coef=coef,
                smooth=smooth)
    loss = SegBackward.apply(logits, finalize, pack)
    return loss, dict(per_image=per_image, argsort=idx, hard_dice=hard.view(()), stats=stats.view(n, NS))


def coteach_loss(logits1, logits2, targets1, targets2, variant, keep, w_ce, w_dice, smooth=1.0, rate=0.0,
                 w_seg=1.0, w_cor=0.0, pseudo1=None, wmap1=None, pseudo2=None, wmap2=None,
                 class_w=(1.0, 1.0), ignore_index=255):
    """Two-network cross-selected losses. `targetsK`/`pseudoK`/`wmapK` are what net K is scored against.
    variant: 0 proposed inline step, 1 Coteachingloss_dropimage, 2 Coteachingloss_weightimage."""
    logits1, logits2 = _logits(logits1), _logits(logits2)
    targets1, t1_bs = _targets(targets1, logits1)
    targets2, t2_bs = _targets(targets2, logits2)
    n, c, h, w = logits1.shape
    if logits2.shape != logits1.shape:
        raise RuntimeError('aide_amd: the two networks\' logits differ in shape: %s vs %s'
                           % (tuple(logits1.shape), tuple(logits2.shape)))
    hw = h * w
    dev = logits1.device
    w0, w1 = class_w
    pseudo1 = _dense(pseudo1, (n, c, h, w), 'pseudo label')
    pseudo2 = _dense(pseudo2, (n, c, h, w), 'pseudo label')
    wmap1 = _dense(wmap1, (n, 1, h, w), 'weight map')
    wmap2 = _dense(wmap2, (n, 1, h, w), 'weight map')
    with torch.no_grad():
        pa1 = stats_pass(logits1, targets1, t1_bs, w0, w1, ignore_index, pseudo1, wmap1)
        pa2 = stats_pass(logits2, targets2, t2_bs, w0, w1, ignore_index, pseudo2, wmap2)
        f64 = dict(device=dev, dtype=torch.float64)
        f32 = dict(device=dev, dtype=torch.float32)
        st1, st2 = torch.empty(n * NS, **f64), torch.empty(n * NS, **f64)
        loss = torch.empty(2, **f32)
        pi1, pi2 = torch.empty(n, **f32), torch.empty(n, **f32)
        i1 = torch.empty(n, device=dev, dtype=torch.int64)
        i2 = torch.empty(n, device=dev, dtype=torch.int64)
        c1, c2 = torch.empty(3 * n, **f32), torch.empty(3 * n, **f32)
        hard = torch.empty(2, **f32)
        if c == 2:
            check(lib.aide_coteach_finalize(ptr(pa1), ptr(pa2), n, hw, variant, keep, w_ce, w_dice, smooth, rate,
                                            w_seg, w_cor, ptr(st1), ptr(st2), ptr(loss), ptr(pi1), ptr(pi2),
                                            ptr(i1), ptr(i2), ptr(c1), ptr(c2), ptr(hard), stream_ptr()),
                  'coteach_finalize')
        else:
            check(lib.aide_coteach_finalize_mc(ptr(pa1), ptr(pa2), n, hw, c, variant, keep, w_ce, w_dice, smooth, rate,
                                               w_seg, w_cor, ptr(st1), ptr(st2), ptr(loss), ptr(pi1), ptr(pi2),
                                               ptr(i1), ptr(i2), ptr(c1), ptr(c2), ptr(hard), stream_ptr()),
                  'coteach_finalize_mc')
    pk1 = dict(targets=targets1, t_bs=t1_bs, w0=w0, w1=w1, ignore=ignore_index, stats=st1, coef=c1,
               smooth=smooth, pseudo=pseudo1, wmap=wmap1)
    pk2 = dict(targets=targets2, t_bs=t2_bs, w0=w0, w1=w1, ignore=ignore_index, stats=st2, coef=c2,
               smooth=smooth, pseudo=pseudo2, wmap=wmap2)
    l1 = SegBackward.apply(logits1, loss[0], pk1)
    l2 = SegBackward.apply(logits2, loss[1], pk2)
    return l1, l2, dict(per_image1=pi1, per_image2=pi2, argsort1=i1, argsort2=i2, hard_dice=hard)
