"""Tensor-level wrappers over the C ABI (include/aide_hip.h).

PyTorch is used here only for device memory and the current HIP stream; every function below
launches hand-written gfx950 kernels from libaide_hip.so and nothing else. Inputs must live on a
HIP device: there is no CPU path.
"""
import ctypes

import torch

from ._lib import lib, check


_STREAM = [None]        # explicit launch stream (use_stream); None = torch's current stream


def stream_ptr():
    if _STREAM[0] is not None:
        return _STREAM[0]
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class use_stream(object):
    """Launch the enclosed ops on the given HIP stream (raw pointer) without switching torch's current stream."""

    def __init__(self, sptr):
        self.sptr = sptr if isinstance(sptr, ctypes.c_void_p) else ctypes.c_void_p(sptr)

    def __enter__(self):
        self.prev = _STREAM[0]
        _STREAM[0] = self.sptr
        return self

    def __exit__(self, *a):
        _STREAM[0] = self.prev
        return False


class Event(ctypes.c_void_p):
    """a HIP event owned by a plan (aide_event_create); destroyed with its owner"""

    def __del__(self):
        try:
            if self.value:
                lib.load().aide_event_destroy(ctypes.c_void_p(self.value))
                self.value = None
        except Exception:       # interpreter shutdown
            pass


class WgradQueue(ctypes.c_void_p):
    """a caller-owned queue of weight-gradient slab reduces (aide_wgrad_queue_*): the wgrad wrappers given one only record
    their reduce, flush(stream) runs everything recorded as one launch.  Host memory; destroyed with its owner."""

    def __del__(self):
        try:
            if self.value:
                lib.load().aide_wgrad_queue_destroy(ctypes.c_void_p(self.value))
                self.value = None
        except Exception:       # interpreter shutdown
            pass

    def pending(self):
        return lib.load().aide_wgrad_queue_pending(self)      # (raw entry: a host-state query stays off the launch tape)

    def flush(self, stream):
        check(lib.aide_wgrad_queue_flush(self, stream), 'wgrad_queue_flush')

    def discard(self):
        return lib.load().aide_wgrad_queue_discard(self)


def new_wgrad_queue():
    q = WgradQueue()
    check(lib.load().aide_wgrad_queue_create(ctypes.byref(q)), 'wgrad_queue_create')
    return q


def _qh(queue):
    return queue if queue is not None else None


def new_event():
    ev = Event()
    check(lib.aide_event_create(ctypes.byref(ev)), 'event_create')
    return ev


def order(ev, src, dst):
    """work enqueued on stream `dst` from now on waits for the work enqueued on `src` so far"""
    check(lib.aide_stream_order(ev, src, dst), 'stream_order')


def record(ev, src):
    """mark the work enqueued on stream `src` so far (first half of order())"""
    check(lib.aide_event_record(ev, src), 'event_record')


def wait(dst, ev):
    """work enqueued on stream `dst` from now on waits for the work marked by record(ev, .) (second half of order())"""
    check(lib.aide_stream_wait_event(dst, ev), 'stream_wait_event')


def _req(t, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('aide_amd ops need HIP device tensors (got %s); there is no CPU fallback'
                           % (t.device if isinstance(t, torch.Tensor) else type(t)))
    if t.dtype != dtype:
        raise RuntimeError('aide_amd: expected dtype %s, got %s' % (dtype, t.dtype))
    return t


def is_bf16(t):
    return t.dtype == torch.bfloat16


def planes(t, bf16_ok=False):
    """(data_ptr, batch_stride) of an NCHW fp32 tensor whose channel planes are dense
    (a channel slice of a contiguous buffer qualifies).  bf16_ok: the tensor may also be a bf16-stored conv output /
    conv-output gradient of the precision='bf16' mode (strides are in elements either way)."""
    _req(t, t.dtype if (bf16_ok and isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16) else torch.float32)
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ok = (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1)
    if not ok:
        raise RuntimeError('aide_amd: tensor must be NCHW with dense channel planes, got strides %s'
                           % (t.stride(),))
    if n == 1:
        sn = c * h * w
    return ctypes.c_void_p(t.data_ptr()), sn


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# ------------------------------------------------------------------------------- conv 3x3
def conv_chunk(cin):
    return lib.aide_conv3x3_chunk(cin)


def pad_to(c, g):
    return (c + g - 1) // g * g


def pack_weights(w, need_dgrad=True):
    """w [Co,Ci,3,3] -> (wf [ci_pad,9,Co], wd [co_pad,9,Ci] | None)."""
    _req(w)
    co, ci = w.shape[0], w.shape[1]
    ci_pad, co_pad = pad_to(ci, conv_chunk(ci)), pad_to(co, conv_chunk(co))
    wf = torch.empty(ci_pad, 9, co, device=w.device, dtype=torch.float32)
    wd = torch.empty(co_pad, 9, ci, device=w.device, dtype=torch.float32) if need_dgrad else None
    check(lib.aide_conv3x3_pack_weights(ptr(w), ptr(wf), ptr(wd), co, ci, ci_pad, co_pad, stream_ptr()),
          'conv3x3_pack_weights')
    return wf, wd


def pack_table(entries, device):
    """entries: list of (w, wf|None, wd|None) -> (device table, n, total_blocks) for
    aide_conv3x3_pack_weights_multi (48-byte records, see include/aide_hip.h)."""
    import struct
    rec, start = b'', 0
    for w, wf, wd in entries:
        co, ci = w.shape[0], w.shape[1]
        elems = (wf.numel() if wf is not None else 0) + (wd.numel() if wd is not None else 0)
        rec += struct.pack('<QQQiiiiq', w.data_ptr(), wf.data_ptr() if wf is not None else 0,
                           wd.data_ptr() if wd is not None else 0, co, ci,
                           wf.shape[0] if wf is not None else 0, wd.shape[0] if wd is not None else 0, start)
        start += (elems + 255) // 256
    return torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(device), len(entries), start


def pack_weights_into(w, wf, wd):
    co, ci = w.shape[0], w.shape[1]
    check(lib.aide_conv3x3_pack_weights(ptr(w), ptr(wf), ptr(wd), co, ci, wf.shape[0],
                                        wd.shape[0] if wd is not None else 0, stream_ptr()),
          'conv3x3_pack_weights')


def conv3x3_igemm(x, wp, bias, y, accumulate=False, plan=-1, ws=None, epi_scale=None, epi_relu=True):
    """y (+)= conv3x3(x) with packed weights wp [cin_pad, 9, cout] (forward pack or dgrad pack).
    epi_scale [cout]: y = relu?(acc * epi_scale + bias) (eval-mode BatchNorm folded into the conv; non-split only)."""
    xp, xbs = planes(x)
    yp, ybs = planes(y)
    n, cin, h, w = x.shape
    cout = y.shape[1]
    assert wp.shape[2] == cout and wp.shape[0] >= cin and y.shape[0] == n and y.shape[2:] == x.shape[2:]
    if plan < 0:
        plan = lib.aide_conv3x3_plan(n, cin, h, w, cout)
    splitk = plan >> 8
    if splitk > 1 and ws is None:
        ws = torch.empty(lib.aide_conv3x3_ws_bytes(n, h, w, cout, splitk) // 4, device=x.device,
                         dtype=torch.float32)
    check(lib.aide_conv3x3_igemm(xp, xbs, ptr(wp), cout, ptr(bias), yp, ybs, n, cin, h, w, cout,
                                 int(accumulate), plan, ptr(ws), ptr(epi_scale), int(bool(epi_relu)), stream_ptr()),
          'conv3x3_igemm')
    return y


def conv3x3_wgrad(dz, a, dw, ws=None, queue=None):
    """queue (every weight-gradient wrapper): None = the slab reduce is launched behind the kernel, or a WgradQueue whose
    flush() runs the reduces of several layers as one launch (ws must then stay untouched until that flush)"""
    dp, dbs = planes(dz)
    ap, abs_ = planes(a)
    n, co, h, w = dz.shape
    ci = a.shape[1]
    assert tuple(dw.shape) == (co, ci, 3, 3) and dw.is_contiguous()
    if ws is None:
        ws = torch.empty(lib.aide_conv3x3_wgrad_ws_bytes(n, co, ci, h, w) // 4, device=dz.device,
                         dtype=torch.float32)
    check(lib.aide_conv3x3_wgrad(dp, dbs, ap, abs_, ptr(dw), n, co, ci, h, w, ptr(ws), _qh(queue), stream_ptr()),
          'conv3x3_wgrad')
    return dw


# ------------------------------------------------------------------------------- conv transpose 2x2
def convT2x2_fwd(x, w, b, y):
    xp, xbs = planes(x)
    yp, ybs = planes(y)
    n, ci, h, wd = x.shape
    co = w.shape[1]
    check(lib.aide_convT2x2_fwd(xp, xbs, ptr(w), ptr(b), yp, ybs, n, ci, co, h, wd, stream_ptr()),
          'convT2x2_fwd')
    return y


def convT2x2_dgrad(dy, w, dx):
    gp, gbs = planes(dy)
    dp, dbs = planes(dx)
    n, ci, h, wd = dx.shape
    co = w.shape[1]
    check(lib.aide_convT2x2_dgrad(gp, gbs, ptr(w), dp, dbs, n, ci, co, h, wd, stream_ptr()),
          'convT2x2_dgrad')
    return dx


def convT2x2_wgrad(x, dy, dw, ws=None):
    xp, xbs = planes(x)
    gp, gbs = planes(dy)
    n, ci, h, wd = x.shape
    co = dy.shape[1]
    if ws is None:
        ws = torch.empty(lib.aide_convT2x2_wgrad_ws_bytes(n, ci, co, h, wd) // 4, device=x.device,
                         dtype=torch.float32)
    check(lib.aide_convT2x2_wgrad(xp, xbs, gp, gbs, ptr(dw), n, ci, co, h, wd, ptr(ws), stream_ptr()),
          'convT2x2_wgrad')
    return dw


# ------------------------------------------------------------------------------- batch norm
def bn_ws(c, device):
    """workspace of the training-mode BatchNorm calls: zero-filled (arrival counters of the one-pass kernels), one stream at a time"""
    return torch.zeros((lib.aide_bn_ws_bytes(c) + 7) // 8, device=device, dtype=torch.float64)


def bn_train_fwd(z, a, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean, rstd, scale, shift,
                 ws, relu=True):
    """a = relu(bn_train(z)); updates running stats / num_batches_tracked; saves mean/rstd/scale/shift."""
    zp, zbs = planes(z, bf16_ok=True)
    ap, abs_ = planes(a, bf16_ok=True)
    n, c, h, w = z.shape
    check(lib.aide_bn_train_fwd_mixed(zp, int(is_bf16(z)), zbs, ap, int(is_bf16(a)), abs_, n, c, h, w, ptr(gamma), ptr(beta), eps,
                                      momentum, ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean), ptr(rstd),
                                      ptr(scale), ptr(shift), int(relu), ptr(ws), stream_ptr()), 'bn_train_fwd')
    return a


def bn_train_fwd_slabs(slabs, splitk, split_stride, bias, z, a, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                       mean, rstd, scale, shift, ws, relu=True):
    """bn_train_fwd whose input is the split-K slabs [splitk][N_total][C][H][W] of the convolution before it (launched with
    accumulate=2): sums them (+ bias) in split order, writes z, normalises.  `slabs` points at the first image of this
    (group) batch inside slab 0; `split_stride` is the full slab size in elements."""
    zp, zbs = planes(z, bf16_ok=True)
    ap, abs_ = planes(a, bf16_ok=True)
    n, c, h, w = z.shape
    check(lib.aide_bn_train_fwd_slabs(slabs, splitk, split_stride, ptr(bias), zp, int(is_bf16(z)), zbs, ap, int(is_bf16(a)),
                                      abs_, n, c, h, w, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
                                      ptr(running_var), ptr(nbt), ptr(mean), ptr(rstd), ptr(scale), ptr(shift), int(relu),
                                      ptr(ws), stream_ptr()), 'bn_train_fwd_slabs')
    return a


def bn_train_fwd_groups(z, a, groups, bn, mean, rstd, scale, shift, ws, relu=True, slabs=None, splitk=0, split_stride=0,
                        slab_bias=None, parts=None, nparts=0, parts_stride=0, conv_bias=None):
    """BatchNorm(train)+ReLU of a stacked batch: z / a hold `groups` runs of N / groups images; per-group statistics, the
    running statistics updated once per group in order -- `groups` sequential forwards in one launch sequence.
    Input: z, or the split-K slabs of the conv before it (slabs = pointer, z is written), or the conv epilogue's
    statistics (parts: group g's nparts entries at g * nparts of every channel's parts_stride)."""
    zp, zbs = planes(z, bf16_ok=True)
    ap, abs_ = planes(a, bf16_ok=True)
    n, c, h, w = z.shape
    assert n % groups == 0
    check(lib.aide_bn_train_fwd_groups(zp, int(is_bf16(z)), zbs, ap, int(is_bf16(a)), abs_, n // groups, groups, c, h, w,
                                       slabs, splitk, split_stride, ptr(slab_bias), ptr(parts), nparts, parts_stride,
                                       ptr(conv_bias), ptr(bn.weight), ptr(bn.bias), bn.eps, bn.momentum,
                                       ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked), ptr(mean),
                                       ptr(rstd), ptr(scale), ptr(shift), int(relu), ptr(ws), stream_ptr()),
          'bn_train_fwd_groups')
    return a


def bn_train_fwd_pool(z, a, pooled, groups, bn, mean, rstd, scale, shift, ws, relu=True):
    """bn_train_fwd(_groups) that also writes max_pool2d(a, 2) into `pooled` (this layer's channel slice of the pooled buffer): fp32,
    lib.aide_bn_relu_bwd_pool_supported(n // groups, c, h, w) shapes"""
    zp, zbs = planes(z)
    ap, abs_ = planes(a)
    pp, pbs = planes(pooled)
    n, c, h, w = z.shape
    check(lib.aide_bn_train_fwd_pool(zp, zbs, ap, abs_, pp, pbs, n // groups, groups, c, h, w, ptr(bn.weight), ptr(bn.bias), bn.eps,
                                     bn.momentum, ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked), ptr(mean),
                                     ptr(rstd), ptr(scale), ptr(shift), int(relu), ptr(ws), stream_ptr()), 'bn_train_fwd_pool')
    return a


def bn_finalize_groups(n_per_group, groups, c, h, w, bn, parts, nparts, parts_stride, conv_bias, mean, rstd, scale, shift,
                       tab, tab_c0):
    """statistics -> per-group (scale, shift) entries [tab_c0, tab_c0 + c) of tab [groups, tab_C, 2] + running statistics;
    no pass over z (the consumer conv applies the table in its loader: conv3x3_wino4(..., in_tab=))"""
    assert tab.dim() == 3 and tab.shape[0] == groups and tab.shape[2] == 2 and tab.is_contiguous()
    check(lib.aide_bn_finalize_groups(n_per_group, groups, c, h, w, ptr(parts), nparts, parts_stride, ptr(conv_bias),
                                      ptr(bn.weight), ptr(bn.bias), bn.eps, bn.momentum, ptr(bn.running_mean),
                                      ptr(bn.running_var), ptr(bn.num_batches_tracked), ptr(mean), ptr(rstd), ptr(scale),
                                      ptr(shift), ptr(tab), tab.shape[1], tab_c0, stream_ptr()), 'bn_finalize_groups')


def bn_train_fwd_parts(z, a, parts, nparts, conv_bias, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean,
                       rstd, scale, shift, relu=True, first=0, stride=None):
    """bn_train_fwd with the statistics emitted by the conv epilogue (conv3x3_wino4(..., stats=)): one pass over z.
    A group of a stacked batch: first = index of the group's first entry, nparts = its entry count, stride = entries per
    channel of the whole launch."""
    zp, zbs = planes(z, bf16_ok=True)
    ap, abs_ = planes(a, bf16_ok=True)
    n, c, h, w = z.shape
    pp = ctypes.c_void_p(parts.data_ptr() + 8 * first)
    check(lib.aide_bn_train_fwd_parts_strided(zp, int(is_bf16(z)), zbs, ap, int(is_bf16(a)), abs_, n, c, h, w, pp, nparts,
                                              nparts if stride is None else stride, ptr(conv_bias), ptr(gamma), ptr(beta), eps,
                                              momentum, ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean), ptr(rstd),
                                              ptr(scale), ptr(shift), int(relu), stream_ptr()), 'bn_train_fwd_parts')
    return a


def bn_eval_fold(bn, conv_bias, scale, shift, fbias):
    """eval-mode coefficients of `bn` and the bias of the convolution before it folded through them"""
    check(lib.aide_bn_eval_fold(scale.numel(), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                                bn.eps, ptr(conv_bias), ptr(scale), ptr(shift), ptr(fbias), stream_ptr()), 'bn_eval_fold')


def bn_relu_apply(z, a, scale, shift, relu=True):
    zp, zbs = planes(z, bf16_ok=True)
    ap, abs_ = planes(a, bf16_ok=True)
    n, c, h, w = z.shape
    check(lib.aide_bn_relu_apply_mixed(zp, int(is_bf16(z)), zbs, ap, int(is_bf16(a)), abs_, n, c, h, w, ptr(scale), ptr(shift),
                                       int(relu), stream_ptr()), 'bn_relu_apply')
    return a


def bn_relu_bwd(dA, z, dz, mean, rstd, scale, shift, dgamma, dbeta, dbias, ws, relu=True, done=None):
    """done: event recorded when dz is complete (attached to the last dispatch; pair with wait(stream, done))"""
    gp, gbs = planes(dA, bf16_ok=True)
    zp, zbs = planes(z, bf16_ok=True)
    dp, dbs = planes(dz, bf16_ok=True)
    n, c, h, w = z.shape
    check(lib.aide_bn_relu_bwd_mixed(gp, int(is_bf16(dA)), gbs, zp, int(is_bf16(z)), zbs, dp, int(is_bf16(dz)), dbs, n, c, h, w,
                                     ptr(mean), ptr(rstd), ptr(scale), ptr(shift), int(relu), ptr(dgamma), ptr(dbeta),
                                     ptr(dbias), ptr(ws), done, stream_ptr()), 'bn_relu_bwd')
    return dz


def bn_relu_bwd_pool(dA, pdy, z, dz, mean, rstd, scale, shift, dgamma, dbeta, dbias, ws, relu=True, done=None):
    """bn_relu_bwd of a layer whose activation also fed a 2x2 max-pooling: dA + the pooled gradient `pdy` routed to the arg-max of
    every window (no max-pooling backward pass); fp32, lib.aide_bn_relu_bwd_pool_supported(n, c, h, w) shapes"""
    gp, gbs = planes(dA)
    pp, pbs = planes(pdy)
    zp, zbs = planes(z)
    dp, dbs = planes(dz)
    n, c, h, w = z.shape
    check(lib.aide_bn_relu_bwd_pool(gp, gbs, pp, pbs, zp, zbs, dp, dbs, n, c, h, w, ptr(mean), ptr(rstd), ptr(scale), ptr(shift),
                                    int(relu), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(ws), done, stream_ptr()), 'bn_relu_bwd_pool')
    return dz


def bn_relu_bwd_head(dlogits, head_w, z, dz, mean, rstd, scale, shift, dgamma, dbeta, dbias, ws, relu=True, done=None):
    """bn_relu_bwd of the layer under the 1x1 head: dA = head_w^T dlogits formed inside the kernel (no head data-gradient pass);
    head_w [K, C] fp32, dlogits [N, K, H, W] fp32"""
    lp, lbs = planes(dlogits)
    zp, zbs = planes(z)
    dp, dbs = planes(dz)
    n, c, h, w = z.shape
    k = dlogits.shape[1]
    assert head_w.shape == (k, c) and head_w.is_contiguous()
    check(lib.aide_bn_relu_bwd_head(lp, lbs, ptr(head_w), k, zp, zbs, dp, dbs, n, c, h, w, ptr(mean), ptr(rstd), ptr(scale),
                                    ptr(shift), int(relu), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(ws), done, stream_ptr()),
          'bn_relu_bwd_head')
    return dz


def bn_relu_bwd_slabs(slabs, splitk, z, dz, mean, rstd, scale, shift, dgamma, dbeta, dbias, ws, relu=True, done=None):
    """bn_relu_bwd whose dA is the split-K slabs [splitk][N][C][H][W] (fp32 tensor `slabs`, at its start) left by the
    data-gradient convolution (accumulate=2); fp32 z / dz, one-pass shapes (lib.aide_bn_one_pass(n, c, h, w) == 1)."""
    zp, zbs = planes(z)
    dp, dbs = planes(dz)
    n, c, h, w = z.shape
    check(lib.aide_bn_relu_bwd_slabs(ptr(slabs), splitk, n * c * h * w, zp, zbs, dp, dbs, n, c, h, w, ptr(mean), ptr(rstd),
                                     ptr(scale), ptr(shift), int(relu), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(ws), done,
                                     stream_ptr()),
          'bn_relu_bwd_slabs')
    return dz


# ------------------------------------------------------------------------------- pool / upsample
def maxpool2x2_fwd(x, y):
    xp, xbs = planes(x, bf16_ok=True)
    yp, ybs = planes(y, bf16_ok=True)
    n, c, h, w = x.shape
    if is_bf16(x) or is_bf16(y):       # bf16-stored activations (precision='bf16')
        check(lib.aide_maxpool2x2_fwd_mixed(xp, int(is_bf16(x)), xbs, yp, int(is_bf16(y)), ybs, n, c, h, w, stream_ptr()),
              'maxpool2x2_fwd')
    else:
        check(lib.aide_maxpool2x2_fwd(xp, xbs, yp, ybs, n, c, h, w, stream_ptr()), 'maxpool2x2_fwd')
    return y


def maxpool2x2_bwd(x, dy, dx, accumulate=False):
    xp, xbs = planes(x, bf16_ok=True)
    gp, gbs = planes(dy, bf16_ok=True)
    dp, dbs = planes(dx, bf16_ok=True)
    n, c, h, w = x.shape
    if is_bf16(x) or is_bf16(dy) or is_bf16(dx):
        check(lib.aide_maxpool2x2_bwd_mixed(xp, int(is_bf16(x)), xbs, gp, int(is_bf16(dy)), gbs, dp, int(is_bf16(dx)), dbs,
                                            n, c, h, w, int(accumulate), stream_ptr()), 'maxpool2x2_bwd')
    else:
        check(lib.aide_maxpool2x2_bwd(xp, xbs, gp, gbs, dp, dbs, n, c, h, w, int(accumulate), stream_ptr()),
              'maxpool2x2_bwd')
    return dx


def upsample2x_fwd(x, y):
    xp, xbs = planes(x, bf16_ok=True)
    yp, ybs = planes(y, bf16_ok=True)
    n, c, h, w = x.shape
    if is_bf16(x) or is_bf16(y):
        check(lib.aide_upsample2x_bilinear_fwd_mixed(xp, int(is_bf16(x)), xbs, yp, int(is_bf16(y)), ybs, n, c, h, w,
                                                     stream_ptr()), 'upsample2x_fwd')
    else:
        check(lib.aide_upsample2x_bilinear_fwd(xp, xbs, yp, ybs, n, c, h, w, stream_ptr()), 'upsample2x_fwd')
    return y


def upsample2x_bwd(dy, dx, accumulate=False):
    gp, gbs = planes(dy, bf16_ok=True)
    dp, dbs = planes(dx, bf16_ok=True)
    n, c, h, w = dx.shape
    if is_bf16(dy) or is_bf16(dx):
        check(lib.aide_upsample2x_bilinear_bwd_mixed(gp, int(is_bf16(dy)), gbs, dp, int(is_bf16(dx)), dbs, n, c, h, w,
                                                     int(accumulate), stream_ptr()), 'upsample2x_bwd')
    else:
        check(lib.aide_upsample2x_bilinear_bwd(gp, gbs, dp, dbs, n, c, h, w, int(accumulate), stream_ptr()),
              'upsample2x_bwd')
    return dx


def fill_zero(t):
    p, bs = planes(t, bf16_ok=True)
    n, c, h, w = t.shape
    if is_bf16(t):                       # as pairs: zero is zero in either type
        assert bs % 2 == 0 and (c * h * w) % 2 == 0
        check(lib.aide_fill_zero(p, bs // 2, n, 1, 1, c * h * w // 2, stream_ptr()), 'fill_zero')
    else:
        check(lib.aide_fill_zero(p, bs, n, c, h, w, stream_ptr()), 'fill_zero')
    return t


# ------------------------------------------------------------------------------- head 1x1
def head1x1_fwd(x, w, b, y):
    xp, xbs = planes(x, bf16_ok=True)
    yp, ybs = planes(y)
    n, c, h, wd = x.shape
    k = w.shape[0]
    check(lib.aide_head1x1_fwd_mixed(xp, int(is_bf16(x)), xbs, ptr(w), ptr(b), yp, ybs, n, c, k, h, wd, stream_ptr()),
          'head1x1_fwd')
    return y


def head1x1_fwd_bn(z, scale, shift, w, b, y):
    """logits of the head on relu(z * scale[c] + shift[c]) formed while the raw conv output z is read (fp32)"""
    zp, zbs = planes(z)
    yp, ybs = planes(y)
    n, c, h, wd = z.shape
    k = w.shape[0]
    check(lib.aide_head1x1_fwd_bn(zp, zbs, ptr(scale), ptr(shift), ptr(w), ptr(b), yp, ybs, n, c, k, h, wd, stream_ptr()),
          'head1x1_fwd_bn')
    return y


def head1x1_wgrad_bn(dy, z, scale, shift, dw, db, ws=None):
    """weight / bias gradient of the head on relu(z * scale[c] + shift[c]) (the activation recomputed from z)"""
    gp, gbs = planes(dy)
    zp, zbs = planes(z)
    n, c, h, wd = z.shape
    k = dw.shape[0]
    if ws is None:
        ws = torch.empty(lib.aide_head1x1_ws_bytes(c, k) // 8, device=z.device, dtype=torch.float64)
    check(lib.aide_head1x1_wgrad_bn(gp, gbs, zp, zbs, ptr(scale), ptr(shift), ptr(dw), ptr(db), n, c, k, h, wd, ptr(ws),
                                    stream_ptr()), 'head1x1_wgrad_bn')


def head1x1_bwd(dy, x, w, dx, dw, db, ws=None):
    gp, gbs = planes(dy)
    xp, xbs = planes(x, bf16_ok=True)
    n, c, h, wd = x.shape
    k = w.shape[0]
    if dx is not None:
        dp, dbs = planes(dx, bf16_ok=True)
    else:
        dp, dbs = ctypes.c_void_p(0), 0
    if ws is None:
        ws = torch.empty(lib.aide_head1x1_ws_bytes(c, k) // 8, device=x.device, dtype=torch.float64)
    check(lib.aide_head1x1_bwd_mixed(gp, gbs, xp, int(is_bf16(x)), xbs, ptr(w), dp, int(dx is not None and is_bf16(dx)),
                                     dbs, ptr(dw), ptr(db), n, c, k, h, wd, ptr(ws), stream_ptr()), 'head1x1_bwd')


# ------------------------------------------------------------------------------- Winograd F(2x2,3x3)
def wino_supported(cin, h, w, cout):
    return bool(lib.aide_conv3x3_wino_supported(cin, h, w, cout))


def wino_pack_table(entries, device):
    """entries: list of (w, uf, ud|None). -> (device table, n, total_blocks) for aide_conv3x3_wino_pack_multi."""
    import struct
    rec, start = b'', 0
    for w, uf, ud in entries:
        co, ci = w.shape[0], w.shape[1]
        rec += struct.pack('<QQQiiiiq', w.data_ptr(), uf.data_ptr() if uf is not None else 0,
                           ud.data_ptr() if ud is not None else 0, co, ci,
                           uf.shape[0] if uf is not None else 0, ud.shape[0] if ud is not None else 0, start)
        start += lib.aide_conv3x3_wino_pack_blocks(co, ci)
    return torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(device), len(entries), start


def wino_pack(w, need_dgrad=True):
    """w [Co,Ci,3,3] -> (uf [ci_pad,16,Co], ud [co_pad,16,Ci] | None): G g G^T of the (rotated) filters."""
    _req(w)
    co, ci = w.shape[0], w.shape[1]
    uf = torch.empty(pad_to(ci, 8), 16, co, device=w.device, dtype=torch.float32)
    ud = torch.empty(pad_to(co, 8), 16, ci, device=w.device, dtype=torch.float32) if need_dgrad else None
    tab, n, blocks = wino_pack_table([(w, uf, ud)], w.device)
    check(lib.aide_conv3x3_wino_pack_multi(ptr(tab), n, blocks, stream_ptr()), 'conv3x3_wino_pack_multi')
    return uf, ud


def conv3x3_wino(x, u, bias, y, accumulate=False, splitk=-1, ws=None):
    xp, xbs = planes(x)
    yp, ybs = planes(y)
    n, cin, h, w = x.shape
    cout = y.shape[1]
    assert u.shape[1] == 16 and u.shape[2] == cout and u.shape[0] >= cin
    if splitk < 0:
        splitk = lib.aide_conv3x3_wino_splitk(n, cin, h, w, cout)
    if splitk > 1 and ws is None:
        ws = torch.empty(lib.aide_conv3x3_ws_bytes(n, h, w, cout, splitk) // 4, device=x.device, dtype=torch.float32)
    check(lib.aide_conv3x3_wino(xp, xbs, ptr(u), ptr(bias), yp, ybs, n, cin, h, w, cout, int(accumulate), splitk,
                                ptr(ws), stream_ptr()), 'conv3x3_wino')
    return y


# ---- Winograd F(4x4,3x3) (large layers) -------------------------------------------------------------
def wino4_supported(cin, h, w, cout):
    return bool(lib.aide_conv3x3_wino4_supported(cin, h, w, cout))


def wino4_pack_table(entries, device):
    """entries: list of (w, uf|None, ud|None). -> (device table, n, total_blocks) for aide_conv3x3_wino4_pack_multi."""
    import struct
    rec, start = b'', 0
    for w, uf, ud in entries:
        co, ci = w.shape[0], w.shape[1]
        rec += struct.pack('<QQQiiiiq', w.data_ptr(), uf.data_ptr() if uf is not None else 0,
                           ud.data_ptr() if ud is not None else 0, co, ci, 0, 0, start)
        start += lib.aide_conv3x3_wino4_pack_blocks(co, ci)
    return torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(device), len(entries), start


def wino4_pack(w, need_dgrad=True):
    """w [Co,Ci,3,3] -> (uf [Ci,36,Co], ud [Co,36,Ci] | None): 6x6 G g G^T of the (rotated) filters, channel-blocked by 4."""
    _req(w)
    co, ci = w.shape[0], w.shape[1]
    uf = torch.empty(ci, 36, co, device=w.device, dtype=torch.float32)
    ud = torch.empty(co, 36, ci, device=w.device, dtype=torch.float32) if need_dgrad else None
    tab, n, blocks = wino4_pack_table([(w, uf, ud)], w.device)
    check(lib.aide_conv3x3_wino4_pack_multi(ptr(tab), n, blocks, stream_ptr()), 'conv3x3_wino4_pack_multi')
    return uf, ud


def conv3x3_wino4(x, u, bias, y, accumulate=False, splitk=-1, ws=None, stats=None, epi_scale=None, epi_relu=True,
                  in_tab=None, in_group_images=0):
    """stats [cout * wino4_stats_parts * 2]: the launch also writes its BatchNorm statistics partials (non-split,
    accumulate = 0, W >= 32); epi_scale [cout]: y = relu?(acc * epi_scale + bias) (eval-mode BatchNorm folded in);
    in_tab [n / in_group_images, cin, 2]: x is the raw output of the layer before, the loader applies relu(x * scale + shift)"""
    xp, xbs = planes(x)
    yp, ybs = planes(y)
    n, cin, h, w = x.shape
    cout = y.shape[1]
    assert u.shape[1] == 36 and u.shape[2] == cout and u.shape[0] == cin
    if splitk < 0:
        splitk = lib.aide_conv3x3_wino4_splitk(n, cin, h, w, cout)
    if splitk > 1 and ws is None:
        ws = torch.empty(lib.aide_conv3x3_ws_bytes(n, h, w, cout, splitk) // 4, device=x.device, dtype=torch.float32)
    check(lib.aide_conv3x3_wino4(xp, xbs, ptr(u), ptr(bias), yp, ybs, n, cin, h, w, cout, int(accumulate), splitk,
                                 ptr(ws), ptr(stats), ptr(epi_scale), int(bool(epi_relu)), ptr(in_tab), int(in_group_images),
                                 stream_ptr()), 'conv3x3_wino4')
    return y


# ------------------------------------------------------------------------------- bf16-MFMA conv mode
def bf16_supported(cin, h, w, cout):
    return bool(lib.aide_conv3x3_bf16_supported(cin, h, w, cout))


def bf16_pack_alloc(cout, cin, device):
    """Uninitialised bf16 pack of one direction: [ceil(cin/16)][9][2][cout][8] (as int16 storage)."""
    return torch.empty(lib.aide_conv3x3_bf16_pack_elems(cout, cin), device=device, dtype=torch.int16)


def bf16_pack_table(entries, device):
    """entries: list of (w, uf|None, ud|None) -> (device table, n, total_blocks) for aide_conv3x3_bf16_pack_multi."""
    import struct
    rec, start = b'', 0
    for w, uf, ud in entries:
        co, ci = w.shape[0], w.shape[1]
        rec += struct.pack('<QQQiiiiq', w.data_ptr(), uf.data_ptr() if uf is not None else 0,
                           ud.data_ptr() if ud is not None else 0, co, ci, 0, 0, start)
        start += lib.aide_conv3x3_bf16_pack_blocks(co, ci)
    return torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(device), len(entries), start


def bf16_pack(w, need_dgrad=True):
    """w [Co,Ci,3,3] fp32 -> (uf, ud | None) bf16 packs (RNE)."""
    _req(w)
    co, ci = w.shape[0], w.shape[1]
    uf = bf16_pack_alloc(co, ci, w.device)
    ud = bf16_pack_alloc(ci, co, w.device) if need_dgrad else None
    tab, n, blocks = bf16_pack_table([(w, uf, ud)], w.device)
    check(lib.aide_conv3x3_bf16_pack_multi(ptr(tab), n, blocks, stream_ptr()), 'conv3x3_bf16_pack_multi')
    return uf, ud


def conv3x3_bf16(x, u, bias, y, accumulate=False, splitk=-1, ws=None):
    """y (+)= conv3x3(x) on the bf16 MFMA path (bf16 operands, fp32 accumulation).  x, y are fp32 tensors; y may be a
    bf16-stored z (forward) or x a bf16-stored dz (dgrad)."""
    xp, xbs = planes(x, bf16_ok=True)
    yp, ybs = planes(y, bf16_ok=True)
    n, cin, h, w = x.shape
    cout = y.shape[1]
    assert u.dtype == torch.int16 and u.numel() == lib.aide_conv3x3_bf16_pack_elems(cout, cin)
    if splitk < 0:
        splitk = lib.aide_conv3x3_bf16_splitk(n, cin, h, w, cout)
    if splitk > 1 and ws is None:
        ws = torch.empty(lib.aide_conv3x3_ws_bytes(n, h, w, cout, splitk) // 4, device=x.device, dtype=torch.float32)
    check(lib.aide_conv3x3_bf16_mixed(xp, int(is_bf16(x)), xbs, ptr(u), ptr(bias), yp, int(is_bf16(y)), ybs, n, cin, h,
                                      w, cout, int(accumulate), splitk, ptr(ws), stream_ptr()), 'conv3x3_bf16')
    return y


def wgrad_bf16_supported(co, ci, h, w):
    return bool(lib.aide_conv3x3_wgrad_bf16_supported(co, ci, h, w))


def conv3x3_wgrad_bf16(dz, a, dw, ws=None, queue=None, co_blocks=0):
    """dw [Co,Ci,3,3] fp32 <- weight gradient on the bf16 MFMA path (dz fp32 or bf16-stored).  co_blocks: 2 / 4 = the
    64 / 128 co x 64 ci workgroup tile, 0 = the library's rule (ws is sized for the same choice)."""
    dzp, dzbs = planes(dz, bf16_ok=True)
    ap, abs_ = planes(a, bf16_ok=True)
    n, co, h, w = dz.shape
    ci = a.shape[1]
    assert tuple(dw.shape) == (co, ci, 3, 3) and dw.is_contiguous()
    if ws is None:
        ws = torch.empty(lib.aide_conv3x3_wgrad_bf16_ws_bytes(n, co, ci, h, w, co_blocks) // 4, device=dz.device,
                         dtype=torch.float32)
    check(lib.aide_conv3x3_wgrad_bf16_mixed(dzp, int(is_bf16(dz)), dzbs, ap, int(is_bf16(a)), abs_, ptr(dw), n, co, ci,
                                            h, w, ptr(ws), co_blocks, _qh(queue), stream_ptr()), 'conv3x3_wgrad_bf16')
    return dw


def wgrad_wino4_supported(co, ci, h, w):
    return bool(lib.aide_conv3x3_wgrad_wino4_supported(co, ci, h, w))


def conv3x3_wgrad_wino4(dz, a, dw, ws=None, target_wgs=0, queue=None):
    """dw [Co,Ci,3,3] <- weight gradient via the transposed Winograd F(4x4,3x3) kernel.  target_wgs: workgroups of the
    launch (0: the library's default, half of the chip); ws must hold aide_conv3x3_wgrad_wino4_ws_bytes_t() for it."""
    dzp, dzbs = planes(dz)
    ap, abs_ = planes(a)
    n, co, h, w = dz.shape
    ci = a.shape[1]
    if ws is None:
        ws = torch.empty(lib.aide_conv3x3_wgrad_wino4_ws_bytes_t(n, co, ci, h, w, target_wgs) // 4, device=dz.device,
                         dtype=torch.float32)
    elif target_wgs and ws.numel() * 4 < lib.load().aide_conv3x3_wgrad_wino4_ws_bytes_t(n, co, ci, h, w, target_wgs):   # (raw entry: a host-side size query stays off the launch tape)
        raise RuntimeError('aide_amd: weight-gradient workspace too small for %d workgroups' % target_wgs)
    check(lib.aide_conv3x3_wgrad_wino4_t(dzp, dzbs, ap, abs_, ptr(dw), n, co, ci, h, w, ptr(ws), target_wgs, _qh(queue),
                                         stream_ptr()), 'conv3x3_wgrad_wino4')
    return dw


def wgrad_wino_supported(co, ci, h, w):
    return bool(lib.aide_conv3x3_wgrad_wino_supported(co, ci, h, w))


def conv3x3_wgrad_wino(dz, a, dw, ws=None, queue=None):
    dp, dbs = planes(dz)
    ap, abs_ = planes(a)
    n, co, h, w = dz.shape
    ci = a.shape[1]
    assert tuple(dw.shape) == (co, ci, 3, 3) and dw.is_contiguous()
    if ws is None:
        ws = torch.empty(lib.aide_conv3x3_wgrad_wino_ws_bytes(n, co, ci, h, w) // 4, device=dz.device,
                         dtype=torch.float32)
    check(lib.aide_conv3x3_wgrad_wino(dp, dbs, ap, abs_, ptr(dw), n, co, ci, h, w, ptr(ws), _qh(queue), stream_ptr()),
          'conv3x3_wgrad_wino')
    return dw


# ------------------------------------------------------------------------------- Spatial_Attention branch
def _dense(t):
    _req(t)
    if not t.is_contiguous():
        raise RuntimeError('aide_amd: dense contiguous tensor expected')
    return ptr(t)


def pwconv_fwd(x, w, b, y):
    """y [N,R,H,W] = conv1x1(x [N,C,H,W]; w [R,C], b [R])."""
    xp, xbs = planes(x)
    yp, ybs = planes(y)
    n, c, h, wd = x.shape
    r = y.shape[1]
    assert w.numel() == r * c
    check(lib.aide_pwconv_fwd(xp, xbs, ptr(w), ptr(b), yp, ybs, n, c, r, h * wd, stream_ptr()), 'pwconv_fwd')
    return y


def pwconv_dgrad(dt, w, dx, gate=None, dout=None, accumulate=False):
    """dx [N,C,H,W] (+)= gate * dout + w^T dt   (dt [N,R,H,W], w [R,C])."""
    tp, tbs = planes(dt)
    xp, xbs = planes(dx)
    n, c, h, wd = dx.shape
    r = dt.shape[1]
    assert w.numel() == r * c
    dp, dbs = planes(dout) if dout is not None else (ctypes.c_void_p(0), 0)
    check(lib.aide_pwconv_dgrad(tp, tbs, ptr(w), ptr(gate), dp, dbs, xp, xbs, n, c, r, h * wd, int(accumulate),
                                stream_ptr()), 'pwconv_dgrad')
    return dx


def pwconv_wgrad(dt, x, dw, db):
    tp, tbs = planes(dt)
    xp, xbs = planes(x)
    n, c, h, wd = x.shape
    r = dt.shape[1]
    assert dw.numel() == r * c
    check(lib.aide_pwconv_wgrad(tp, tbs, xp, xbs, ptr(dw), ptr(db), n, c, r, h * wd, stream_ptr()), 'pwconv_wgrad')


def dconv_small(x, w, b, y, dilation, transposed=False):
    """Dilated 3x3 conv (padding = dilation) on dense small-channel tensors; transposed=True is its dgrad."""
    n, cx, h, wd = x.shape
    cout, cin = w.shape[0], w.shape[1]
    assert (cx, y.shape[1]) == ((cout, cin) if transposed else (cin, cout))
    check(lib.aide_dconv3x3_small(_dense(x), _dense(w), ptr(b), _dense(y), n, cin, cout, h, wd, dilation,
                                  int(transposed), stream_ptr()), 'dconv3x3_small')
    return y


def dconv_small_wgrad(dy, x, dw, db, dilation):
    n, cout, h, wd = dy.shape
    cin = x.shape[1]
    check(lib.aide_dconv3x3_small_wgrad(_dense(dy), _dense(x), ptr(dw), ptr(db), n, cout, cin, h, wd, dilation,
                                        stream_ptr()), 'dconv3x3_small_wgrad')


def sa_gate_fwd(t4, bn, training, stat, gate):
    check(lib.aide_sa_gate_fwd(_dense(t4), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                               ptr(bn.num_batches_tracked), float(bn.eps), float(bn.momentum), int(training),
                               ptr(stat), _dense(gate), t4.numel(), stream_ptr()), 'sa_gate_fwd')


def sa_mul(gate, y, out):
    yp, ybs = planes(y)
    op_, obs = planes(out)
    n, c, h, wd = y.shape
    check(lib.aide_sa_mul(_dense(gate), yp, ybs, op_, obs, n, c, h * wd, stream_ptr()), 'sa_mul')


def sa_gate_bwd(dout, y, gate, t4, stat, gamma, dgamma, dbeta, dt4, ws):
    dp, dbs = planes(dout)
    yp, ybs = planes(y)
    n, c, h, wd = y.shape
    assert ws.numel() >= n * h * wd + 4
    check(lib.aide_sa_gate_bwd(dp, dbs, yp, ybs, _dense(gate), _dense(t4), ptr(stat), ptr(gamma), ptr(dgamma),
                               ptr(dbeta), _dense(dt4), n, c, h * wd, ptr(ws), stream_ptr()), 'sa_gate_bwd')
