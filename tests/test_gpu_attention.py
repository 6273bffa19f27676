"""-m gpu: kernels of the Spatial_Attention branch (aide_amd/csrc/attention.hip) against the aten CPU ops they replace
(netblocks.py:68-89: 1x1 conv, dilated 3x3 conv, BatchNorm2d(1) + sigmoid gate, gate multiply) and their autograd
backward; fp32, 2e-5 of the tensor scale.  Whole-network parity of fuseunetsa / UNetsa against the reference's
golden vectors is in tests/test_gpu_models.py (g1_fuseunetsa.npz, g1_unetsa.npz)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol=2e-5, what=''):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, '%s: max abs err %.3e > %.3e (scale %.3e)' % (what, err, rtol * scale, scale)


@pytest.mark.parametrize('case', [(2, 32, 2, 16, 16), (1, 64, 4, 8, 12), (2, 100, 6, 5, 3), (1, 512, 32, 4, 4)])
def test_pwconv(dev, case):
    from aide_amd import ops
    n, c, r, h, w = case
    g = torch.Generator().manual_seed(c + r)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(r, c, 1, 1, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(r, generator=g, requires_grad=True)
    dy = torch.randn(n, r, h, w, generator=g)
    y = F.conv2d(x, wt, b)
    y.backward(dy)
    xd, wd, bd, dyd = x.detach().to(dev), wt.detach().to(dev), b.detach().to(dev), dy.to(dev)
    yd = torch.empty(n, r, h, w, device=dev)
    ops.pwconv_fwd(xd, wd, bd, yd)
    _close(yd, y, what='pw fwd %s' % (case,))
    dx = torch.empty_like(xd)
    ops.pwconv_dgrad(dyd, wd, dx)
    _close(dx, x.grad, what='pw dgrad')
    gate = torch.rand(n, h, w, generator=g)
    dout = torch.randn(n, c, h, w, generator=g)
    base = torch.randn(n, c, h, w, generator=g)
    dx2 = base.to(dev)
    ops.pwconv_dgrad(dyd, wd, dx2, gate=gate.to(dev), dout=dout.to(dev), accumulate=True)
    _close(dx2, base + x.grad + gate[:, None] * dout, what='pw dgrad fused')
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    ops.pwconv_wgrad(dyd, xd, dw, db)
    _close(dw, wt.grad, what='pw wgrad')
    _close(db, b.grad, what='pw bias grad')


@pytest.mark.parametrize('case', [(2, 2, 32, 32, 4), (1, 8, 16, 20, 4), (2, 4, 6, 6, 4), (1, 32, 4, 4, 4), (1, 3, 12, 9, 2)])
def test_dilated_conv_small(dev, case):
    from aide_amd import ops
    n, r, h, w, dil = case
    g = torch.Generator().manual_seed(r * 7 + h)
    x = torch.randn(n, r, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(r, r, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(r, generator=g, requires_grad=True)
    dy = torch.randn(n, r, h, w, generator=g)
    y = F.conv2d(x, wt, b, padding=dil, dilation=dil)
    y.backward(dy)
    xd, wd, bd, dyd = x.detach().to(dev), wt.detach().to(dev), b.detach().to(dev), dy.to(dev)
    yd = torch.empty_like(xd)
    ops.dconv_small(xd, wd, bd, yd, dil)
    _close(yd, y, what='dconv fwd %s' % (case,))
    dx = torch.empty_like(xd)
    ops.dconv_small(dyd, wd, None, dx, dil, transposed=True)
    _close(dx, x.grad, what='dconv dgrad')
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    ops.dconv_small_wgrad(dyd, xd, dw, db, dil)
    _close(dw, wt.grad, what='dconv wgrad')
    _close(db, b.grad, what='dconv bias grad')


@pytest.mark.parametrize('training', [True, False])
def test_gate_and_multiply(dev, training):
    """gate = sigmoid(BatchNorm2d(1)(t4)), out = gate * y, and the backward w.r.t. t4 / gamma / beta / y."""
    from aide_amd import ops
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 2, 8, 12, 10
    bn = torch.nn.BatchNorm2d(1)
    with torch.no_grad():
        bn.weight.fill_(1.3); bn.bias.fill_(-0.2); bn.running_mean.fill_(0.1); bn.running_var.fill_(0.7)
    bn.train(training)
    t4 = torch.randn(n, 1, h, w, generator=g, requires_grad=True)
    y = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    dout = torch.randn(n, c, h, w, generator=g)
    import copy
    bnd = copy.deepcopy(bn).to(dev)
    out = torch.sigmoid(bn(t4)) * y
    gate_d, stat = torch.empty(n, h, w, device=dev), torch.empty(2, device=dev)
    ops.sa_gate_fwd(t4.detach().to(dev), bnd, training, stat, gate_d)
    outd = torch.empty(n, c, h, w, device=dev)
    ops.sa_mul(gate_d, y.detach().to(dev), outd)
    _close(outd, out, what='gate * y')
    _close(bnd.running_mean, bn.running_mean, what='running mean')
    _close(bnd.running_var, bn.running_var, what='running var')
    assert int(bnd.num_batches_tracked) == int(bn.num_batches_tracked)
    if training:
        out.backward(dout)
        dgam, dbet = torch.empty(1, device=dev), torch.empty(1, device=dev)
        dt4 = torch.empty(n, 1, h, w, device=dev)
        ws = torch.empty(n * h * w + 4, device=dev)
        ops.sa_gate_bwd(dout.to(dev), y.detach().to(dev), gate_d, t4.detach().to(dev), stat, bnd.weight, dgam, dbet,
                        dt4, ws)
        _close(dt4, t4.grad, rtol=1e-4, what='d t4')
        _close(dgam, bn.weight.grad, rtol=1e-4, what='d gamma')
        _close(dbet, bn.bias.grad, rtol=1e-4, what='d beta')


def test_attention_models_train(dev):
    """fuseunetsa / UNetsa: same state_dict keys as the oracle (== reference), training reduces the loss, eval runs."""
    import oracle
    from aide_amd import utils as U
    from aide_amd.optim import Adam
    from aide_amd.models_twomodalinputs import fuseunetsa
    from aide_amd.models_singlemodalinput import UNetsa
    for ours, ref, nin in ((fuseunetsa, oracle.fuseunetsa, 2), (UNetsa, oracle.UNetsa, 1)):
        torch.manual_seed(2)
        net = ours(2)
        assert list(net.state_dict().keys()) == list(ref(2).state_dict().keys())
        net = net.to(dev)
        g = torch.Generator().manual_seed(3)
        xs = [torch.randn(2, 3, 48, 32, generator=g).to(dev) for _ in range(nin)]
        t = (torch.rand(2, 48, 32, generator=g) > 0.7).long().to(dev)
        w = torch.tensor([1.0, 1.0])
        crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
        opt = Adam(net.parameters(), lr=1e-3, amsgrad=True)
        losses = []
        for _ in range(5):
            opt.zero_grad()
            loss = crit(net(*xs), t)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        assert losses[-1] < losses[0], losses
        net.eval()
        with torch.no_grad():
            assert torch.isfinite(net(*xs)).all()
