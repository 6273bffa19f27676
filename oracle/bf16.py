"""bf16-operand emulation of the 3x3 convolutions of an oracle network (TEST INFRASTRUCTURE ONLY).

BASELINE config 5 asks for a bf16 MFMA path.  The product's contract (aide_amd/csrc/conv3x3_bf16.hip) is:
conv operands -- activations, incoming gradients, filters -- are rounded to bf16 (round-to-nearest-even) where
they enter a convolution, products are accumulated in fp32, the conv output z (= accumulators + bias) is stored as
bf16 (what torch.autocast does as well; BatchNorm then reads the bf16 values in fp32), activations (BatchNorm+ReLU
outputs, pooled and up-sampled tensors) of the planes the bf16 kernels cover are stored as bf16, and everything else
(BatchNorm, pooling, up-sampling, head arithmetic, gradients of activations, loss, Adam) is the fp32 arithmetic of the
reference.  (The gradient dz is stored as bf16
too, which is not a separate rounding: dz is only ever read as a conv operand.)  `emulate_bf16(net)` rewires the
nn.Conv2d(.., 3, padding=1) layers of an oracle network (oracle/nets.py, which follows
models_twomodalinputs/netblocks.py:24-27 and models_singlemodalinput/UNet.py:19-22) to exactly that arithmetic with
stock aten CPU ops, for the layers / directions the product runs in bf16 (same shape predicates as
include/aide_hip.h: aide_conv3x3_bf16_supported / aide_conv3x3_wgrad_bf16_supported, restated here in Python so
the oracle never calls into the HIP library)."""
import types

import torch
import torch.nn.functional as F


STORE_Z_BF16 = [True]       # False: emulate the A-B mode with an fp32-stored conv output (engine config.store_bf16)
STORE_A_BF16 = [True]       # False: ... with fp32-stored activations (engine.STORE_A_BF16)


def rb(t):
    """fp32 -> bf16 (RNE) -> fp32"""
    return t.bfloat16().float()


def conv_bf16_supported(cin, h, w, cout):
    return w >= 32 and w % 32 == 0 and cout % 32 == 0


def wgrad_bf16_supported(co, ci, h, w):
    return co % 32 == 0 and w % 32 == 0 and h % 4 == 0


class _ConvBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        n, ci, h, wd = x.shape
        co = w.shape[0]
        if conv_bf16_supported(ci, h, wd, co):
            y = F.conv2d(rb(x), rb(w), b, padding=1)
            return rb(y) if STORE_Z_BF16[0] else y
        return F.conv2d(x, w, b, padding=1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        n, ci, h, wd = x.shape
        co = w.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            if conv_bf16_supported(co, h, wd, ci):           # dgrad = the same kernel with the channel roles swapped
                dx = F.conv_transpose2d(rb(dy), rb(w), padding=1)
            else:
                dx = F.conv_transpose2d(dy, w, padding=1)
        if wgrad_bf16_supported(co, ci, h, wd):
            dw = torch.nn.grad.conv2d_weight(rb(x), w.shape, rb(dy), padding=1)
        else:
            dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
        return dx, dw, dy.sum((0, 2, 3))


STORE_G_BF16 = [True]       # gradients of the bf16-stored activations are stored as bf16 as well (engine.STORE_G_BF16)


class _Store(torch.autograd.Function):
    """A bf16-stored activation: the value is rounded on the way forward, its (summed) gradient on the way back."""

    @staticmethod
    def forward(ctx, t):
        return rb(t)

    @staticmethod
    def backward(ctx, g):
        return rb(g) if STORE_G_BF16[0] else g


def _round_st(t):
    return _Store.apply(t)


def _store_hook(mod, inp, out):
    # the product stores an activation buffer as bf16 when all its readers are bf16 convolutions (forward and weight
    # gradient), pooling, up-sampling or the head: in FuseUNet / UNet that is every plane whose width is a multiple of 32
    # (and height of 4).  Rounding commutes with the ReLU that follows a BatchNorm and with max-pooling.
    n, c, h, w = out.shape
    if STORE_A_BF16[0] and w % 32 == 0 and h % 4 == 0:
        return _round_st(out)
    return out


def emulate_bf16(net):
    """Rewire every 3x3/pad-1 nn.Conv2d of `net` (in place; parameters and state_dict keys unchanged) and round the
    activations the product stores as bf16 (outputs of the conv BatchNorms and of the bilinear up-sampling)."""
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.padding == (1, 1):
            m.forward = types.MethodType(lambda self, x: _ConvBf16.apply(x, self.weight, self.bias), m)
        elif isinstance(m, torch.nn.BatchNorm2d) and m.num_features > 1:
            m.register_forward_hook(_store_hook)
        elif isinstance(m, torch.nn.Upsample):
            m.register_forward_hook(_store_hook)
    return net
