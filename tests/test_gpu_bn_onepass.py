"""One-pass BatchNorm kernels (-m gpu): S workgroups per channel hold the channel in registers and exchange fp64 partial sums
through the workspace (csrc/bn.hip, bn_fwd_coop_kernel / bn_bwd_coop_kernel).  Replaces nn.BatchNorm2d(train) + nn.ReLU and
their autograd backward (reference: models_twomodalinputs/netblocks.py:24-28, models_singlemodalinput/UNet.py:19-23).
Checked here: parity with aten (CPU) on shapes that hit every (units per thread, workgroups per channel) instantiation and
ragged tails; the inter-workgroup exchange is bit-reproducible launch after launch on one workspace, also beside a kernel that fills the
chip on another stream; a stacked batch equals sequential forwards bit for bit; the slab-fed forms equal reduce-then-normalise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol=2e-5, what=''):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, '%s: max abs err %.3e > %.3e (ref scale %.3e)' % (what, err, rtol * scale, scale)


def _epochs(ws, c):
    """generations completed per channel: the first int64 of every channel's workspace block (8 + 4 * 256 doubles)"""
    return ws.view(torch.int64)[::8 + 4 * 256][:c]


def _stats(c, dev):
    return tuple(torch.empty(c, device=dev) for _ in range(4))


# (N, C, H, W); workgroups per channel / values per thread: coop_plan (bn.hip)
SHAPES = [
    (4, 32, 256, 256),
    (4, 64, 128, 128),
    (4, 128, 64, 64),
    (4, 256, 32, 32),
    (4, 512, 16, 16),
    (4, 1024, 16, 16),
    (4, 64, 320, 320),
    (3, 40, 36, 28),        # ragged: 756 units, last workgroup partly empty
    (5, 8, 20, 20),         # 500 units, C * S small -> Q 1, S 2
    (1, 16, 48, 32),
    (2, 3, 16, 16),
    (8, 16, 512, 256),      # 262144 units: Q 8, S 128 (the largest one-pass channel)
]


@pytest.mark.parametrize('shape', SHAPES)
def test_onepass_fwd_bwd_vs_aten(dev, shape):
    from aide_amd import ops
    from aide_amd._lib import lib
    n, c, h, w = shape
    assert lib.aide_bn_one_pass(n, c, h, w) == 1
    g = torch.Generator().manual_seed(n * 1000 + c + h)
    z = torch.randn(n, c, h, w, generator=g) * 1.7 + torch.randn(1, c, 1, 1, generator=g)
    dA = torch.randn(n, c, h, w, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    bn = torch.nn.BatchNorm2d(c).double()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    zr = z.double().requires_grad_(True)
    ar = F.relu(bn(zr))
    ar.backward(dA.double())

    zd, dAd = z.to(dev), dA.to(dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    mean, rstd, scale, shift = _stats(c, dev)
    ws = ops.bn_ws(c, dev)
    a = torch.empty_like(zd)
    ops.bn_train_fwd(zd, a, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rm, rv, nbt, mean, rstd, scale, shift, ws, True)
    a2 = torch.empty_like(zd)
    ops.bn_relu_apply(zd, a2, scale, shift, True)
    assert torch.equal(a, a2)
    _close(a, ar, what='fwd')
    _close(rm, bn.running_mean, what='running_mean')
    _close(rv, bn.running_var, what='running_var')
    assert int(nbt.item()) == 1
    dz = torch.empty_like(zd)
    dg, db, dbias = (torch.empty(c, device=dev) for _ in range(3))
    ops.bn_relu_bwd(dAd, zd, dz, mean, rstd, scale, shift, dg, db, dbias, ws, True)
    # (the ReLU mask of an element within rounding of zero may differ from the float64 reference: compare where it agrees)
    mask_ref = (ar > 0).cpu()
    mask_got = (a > 0).cpu()
    assert (mask_ref != mask_got).sum().item() <= 2
    if torch.equal(mask_ref, mask_got):
        _close(dz, zr.grad, rtol=5e-5, what='dz')
        _close(dg, bn.weight.grad, rtol=5e-5, what='dgamma')
        _close(db, bn.bias.grad, rtol=5e-5, what='dbeta')
    assert dbias.abs().max().item() < 1e-3
    ep = _epochs(ws, c)
    assert int(ep.min().item()) == int(ep.max().item()) and int(ep[0].item()) in (0, 2)     # (one-workgroup channels exchange nothing)


def test_onepass_repeatable_beside_a_full_chip(dev):
    """the same workspace re-used back to back, alone and while another stream keeps every CU busy: identical bits"""
    from aide_amd import ops
    n, c, h, w = 4, 128, 64, 64
    g = torch.Generator().manual_seed(5)
    z = torch.randn(n, c, h, w, generator=g).to(dev) * 3.0
    dA = torch.randn(n, c, h, w, generator=g).to(dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    ws = ops.bn_ws(c, dev)
    mean, rstd, scale, shift = _stats(c, dev)

    def run():
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        a, dz = torch.empty_like(z), torch.empty_like(z)
        outs = [torch.empty(c, device=dev) for _ in range(3)]
        ops.bn_train_fwd(z, a, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, rstd, scale, shift, ws, True)
        ops.bn_relu_bwd(dA, z, dz, mean, rstd, scale, shift, outs[0], outs[1], outs[2], ws, True)
        return [a, dz, rm, rv] + outs

    ref = run()
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device=dev)
    side = torch.cuda.Stream(device=dev)
    for it in range(12):
        if it % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    big @ big
        got = run()
        for x, y in zip(got, ref):
            assert torch.equal(x, y)


@pytest.mark.parametrize('case', [(4, 4, 128, 64, 64, 0), (2, 4, 256, 32, 32, 0), (4, 4, 32, 128, 128, 0), (3, 2, 512, 16, 16, 0),
                                  (4, 4, 128, 64, 64, 2), (4, 4, 512, 32, 32, 4), (2, 3, 256, 32, 32, 3)])
def test_onepass_groups_equal_sequential_forwards(dev, case):
    """a stacked batch (groups walked inside the kernel) against one launch per group: activations, running statistics, the
    saved coefficients -- bit for bit; z as it is (splitk = 0) or summed from split-K slabs"""
    from aide_amd import ops
    m, groups, c, h, w, splitk = case
    n = m * groups
    g = torch.Generator().manual_seed(c + h + groups)
    bn = torch.nn.BatchNorm2d(c).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
    bias = torch.randn(c, generator=g).to(dev)
    if splitk:
        slabs = torch.randn(splitk, n, c, h, w, generator=g).to(dev)
        zsum = slabs[0].clone()
        for s in range(1, splitk):
            zsum += slabs[s]
        zsum += bias.view(1, c, 1, 1)
    else:
        zsum = (torch.randn(n, c, h, w, generator=g) * 2.0 + 0.5).to(dev)
    import ctypes
    # sequential reference
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    ws = ops.bn_ws(c, dev)
    st_a = _stats(c, dev)
    a_ref = torch.empty_like(zsum)
    z_ref = torch.empty_like(zsum) if splitk else zsum
    for gi in range(groups):
        sl = slice(gi * m, (gi + 1) * m)
        if splitk:
            ptr = ctypes.c_void_p(slabs.data_ptr() + 4 * gi * m * c * h * w)
            ops.bn_train_fwd_slabs(ptr, splitk, n * c * h * w, bias, z_ref[sl], a_ref[sl], bn.weight, bn.bias, bn.eps, bn.momentum,
                                   bn.running_mean, bn.running_var, bn.num_batches_tracked, st_a[0], st_a[1], st_a[2], st_a[3],
                                   ws, True)
        else:
            ops.bn_train_fwd(zsum[sl], a_ref[sl], bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                             bn.num_batches_tracked, st_a[0], st_a[1], st_a[2], st_a[3], ws, True)
    rm_ref, rv_ref, nbt_ref = bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked.item())
    if splitk:
        assert torch.equal(z_ref, zsum)
    # stacked
    with torch.no_grad():
        bn.running_mean.copy_(rm0)
        bn.running_var.copy_(rv0)
        bn.num_batches_tracked.zero_()
    st_b = _stats(c, dev)
    a = torch.empty_like(zsum)
    zz = torch.empty_like(zsum) if splitk else zsum
    if splitk:
        ops.bn_train_fwd_groups(zz, a, groups, bn, st_b[0], st_b[1], st_b[2], st_b[3], ws, slabs=ctypes.c_void_p(slabs.data_ptr()),
                                splitk=splitk, split_stride=n * c * h * w, slab_bias=bias)
        assert torch.equal(zz, zsum)
    else:
        ops.bn_train_fwd_groups(zz, a, groups, bn, st_b[0], st_b[1], st_b[2], st_b[3], ws)
    assert torch.equal(a, a_ref)
    assert torch.equal(bn.running_mean, rm_ref) and torch.equal(bn.running_var, rv_ref)
    assert int(bn.num_batches_tracked.item()) == nbt_ref == groups
    for x, y in zip(st_a, st_b):
        assert torch.equal(x, y)


@pytest.mark.parametrize('case', [(4, 128, 64, 64, 2), (4, 512, 32, 32, 4), (4, 1024, 16, 16, 16), (4, 64, 128, 128, 3)])
def test_onepass_fwd_from_splitk_slabs(dev, case):
    """forward fed by split-K slabs == slab reduce (split order, + bias) followed by the plain forward, bit for bit"""
    import ctypes
    from aide_amd import ops
    n, c, h, w, splitk = case
    g = torch.Generator().manual_seed(c + splitk)
    slabs = torch.randn(splitk, n, c, h, w, generator=g).to(dev)
    bias = torch.randn(c, generator=g).to(dev)
    zsum = slabs[0].clone()
    for s in range(1, splitk):
        zsum += slabs[s]
    zsum += bias.view(1, c, 1, 1)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    ws = ops.bn_ws(c, dev)
    outs = []
    for fed in (False, True):
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        st = _stats(c, dev)
        a = torch.empty_like(zsum)
        if fed:
            zz = torch.empty_like(zsum)
            ops.bn_train_fwd_slabs(ctypes.c_void_p(slabs.data_ptr()), splitk, n * c * h * w, bias, zz, a, gamma, beta, 1e-5, 0.1,
                                   rm, rv, nbt, st[0], st[1], st[2], st[3], ws, True)
            assert torch.equal(zz, zsum)
        else:
            ops.bn_train_fwd(zsum, a, gamma, beta, 1e-5, 0.1, rm, rv, nbt, st[0], st[1], st[2], st[3], ws, True)
        outs.append([a, rm, rv] + list(st))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize('shape', [(8, 32, 512, 512), (4, 64, 64, 64), (2, 256, 16, 16), (4, 32, 72, 56)])
def test_onepass_bf16_storage(dev, shape):
    """bf16-stored z / a / dA / dz (precision='bf16'): the arithmetic is that of the fp32 kernels on the widened tensors --
    bit-identical statistics and coefficient vectors, outputs equal to the fp32 outputs rounded to nearest even"""
    from aide_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h + c)
    zb = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev).bfloat16()
    db_ = torch.randn(n, c, h, w, generator=g).to(dev).bfloat16()
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    res = []
    for narrow in (False, True):
        z = zb if narrow else zb.float()
        dA = db_ if narrow else db_.float()
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        st = _stats(c, dev)
        ws = ops.bn_ws(c, dev)
        a, dz = torch.empty_like(z), torch.empty_like(z)
        outs = [torch.empty(c, device=dev) for _ in range(3)]
        ops.bn_train_fwd(z, a, gamma, beta, 1e-5, 0.1, rm, rv, nbt, st[0], st[1], st[2], st[3], ws, True)
        ops.bn_relu_bwd(dA, z, dz, st[0], st[1], st[2], st[3], outs[0], outs[1], outs[2], ws, True)
        res.append((a, dz, [rm, rv] + list(st) + outs))
    (a32, dz32, v32), (a16, dz16, v16) = res
    for x, y in zip(v32, v16):
        assert torch.equal(x, y)
    assert torch.equal(a32.bfloat16(), a16) and torch.equal(dz32.bfloat16(), dz16)


@pytest.mark.parametrize('shape', [(4, 64, 256, 256), (4, 128, 64, 64), (2, 512, 32, 32), (3, 24, 40, 48), (1, 8, 64, 32)])
def test_onepass_bwd_with_pooled_gradient(dev, shape):
    """aide_bn_relu_bwd_pool: dA + the gradient of MaxPool2d(2, 2)(relu(bn(z))) routed to the window arg-max inside the BatchNorm
    backward == aide_maxpool2x2_bwd(accumulate) into dA followed by the plain backward, bit for bit -- on activations with many
    tied windows (half of the values are clipped to zero by the ReLU: the first maximum in row-major order takes the gradient, as in
    nn.MaxPool2d's backward, fuseunet.py:13-31) and with exact ties between positive values."""
    from aide_amd import ops
    from aide_amd._lib import lib
    n, c, h, w = shape
    assert lib.aide_bn_relu_bwd_pool_supported(n, c, h, w) == 1
    g = torch.Generator().manual_seed(h * 7 + c)
    z = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev)
    z[:, :, 0::2, 0::2] = z[:, :, 1::2, 1::2] * torch.where(torch.rand(n, c, h // 2, w // 2, generator=g) > 0.7, 1.0, 0.3).to(dev)
    dA = torch.randn(n, c, h, w, generator=g).to(dev)
    pdy = torch.randn(n, c, h // 2, w // 2, generator=g).to(dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    ws = ops.bn_ws(c, dev)
    st = _stats(c, dev)
    a = torch.empty_like(z)
    ops.bn_train_fwd(z, a, gamma, beta, 1e-5, 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev),
                     torch.zeros((), dtype=torch.int64, device=dev), st[0], st[1], st[2], st[3], ws, True)
    # reference sequence
    dsum = dA.clone()
    ops.maxpool2x2_bwd(a, pdy, dsum, accumulate=True)
    ref = [torch.empty_like(z)] + [torch.empty(c, device=dev) for _ in range(3)]
    ops.bn_relu_bwd(dsum, z, ref[0], st[0], st[1], st[2], st[3], ref[1], ref[2], ref[3], ws, True)
    out = [torch.empty_like(z)] + [torch.empty(c, device=dev) for _ in range(3)]
    ops.bn_relu_bwd_pool(dA, pdy, z, out[0], st[0], st[1], st[2], st[3], out[1], out[2], out[3], ws, True)
    for x, y, what in zip(out, ref, ('dz', 'dgamma', 'dbeta', 'dbias')):
        assert torch.equal(x, y), what
    # ... and against autograd: relu(bn(z)) feeding a max-pooling and a second reader
    zr = z.detach().cpu().double().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(c).double()
    with torch.no_grad():
        bn.weight.copy_(gamma.cpu())
        bn.bias.copy_(beta.cpu())
    ar = F.relu(bn(zr))
    (F.max_pool2d(ar, 2) * pdy.cpu().double()).sum().backward(retain_graph=True)
    (ar * dA.cpu().double()).sum().backward()
    if torch.equal((ar > 0).cpu(), (a > 0).cpu()):
        _close(out[0], zr.grad, rtol=1e-4, what='dz vs autograd')


@pytest.mark.parametrize('case', [(4, 64, 256, 256, 2), (2, 64, 64, 48, 3), (3, 32, 20, 20, 8), (1, 16, 32, 32, 1)])
def test_onepass_bwd_with_head_gradient(dev, case):
    """aide_bn_relu_bwd_head: the 1x1 head's data gradient sum_k w[k][c] dlogits[k] formed inside the BatchNorm backward of the layer
    under the head == aide_head1x1_bwd's dx followed by the plain backward, bit for bit (2 .. 8 classes; last_conv1, fuseunet.py:41)."""
    from aide_amd import ops
    n, c, h, w, k = case
    g = torch.Generator().manual_seed(c + k)
    z = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev)
    dl = torch.randn(n, k, h, w, generator=g).to(dev)
    hw_ = (torch.randn(k, c, generator=g) * 0.2).to(dev)
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    ws = ops.bn_ws(c, dev)
    st = _stats(c, dev)
    ops.bn_train_fwd(z, torch.empty_like(z), gamma, beta, 1e-5, 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev),
                     torch.zeros((), dtype=torch.int64, device=dev), st[0], st[1], st[2], st[3], ws, True)
    dA = torch.empty_like(z)
    ops.head1x1_bwd(dl, x, hw_, dA, None, None)
    ref = [torch.empty_like(z)] + [torch.empty(c, device=dev) for _ in range(3)]
    ops.bn_relu_bwd(dA, z, ref[0], st[0], st[1], st[2], st[3], ref[1], ref[2], ref[3], ws, True)
    out = [torch.empty_like(z)] + [torch.empty(c, device=dev) for _ in range(3)]
    ops.bn_relu_bwd_head(dl, hw_, z, out[0], st[0], st[1], st[2], st[3], out[1], out[2], out[3], ws, True)
    for a, b, what in zip(out, ref, ('dz', 'dgamma', 'dbeta', 'dbias')):
        assert torch.equal(a, b), what
    _close(dA, torch.einsum('nkhw,kc->nchw', dl.cpu().double(), hw_.cpu().double()), rtol=1e-6, what='head dx')


def test_onepass_two_kernels_at_once(dev):
    """two one-pass launches in flight together (the two lanes of a forward pass, the two networks of the co-teaching step): each on
    its own stream and workspace, many workgroups per channel on both, 40 rounds back to back -- every result equals the one the launch
    gives alone, bit for bit (the waits of one kernel must never starve the other's missing workgroups)"""
    from aide_amd import ops
    shapes = [(4, 32, 256, 256), (4, 64, 128, 128)]
    streams = [torch.cuda.Stream(device=dev) for _ in shapes]
    jobs = []
    for k, (n, c, h, w) in enumerate(shapes):
        g = torch.Generator().manual_seed(90 + k)
        z = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev)
        dA = torch.randn(n, c, h, w, generator=g).to(dev)
        gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
        jobs.append(dict(z=z, dA=dA, gamma=gamma, beta=beta, ws=ops.bn_ws(c, dev), st=_stats(c, dev), c=c,
                         a=torch.empty_like(z), dz=torch.empty_like(z), outs=[torch.empty(c, device=dev) for _ in range(3)]))

    def launch(j):
        c = j['c']
        ops.bn_train_fwd(j['z'], j['a'], j['gamma'], j['beta'], 1e-5, 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev),
                         torch.zeros((), dtype=torch.int64, device=dev), j['st'][0], j['st'][1], j['st'][2], j['st'][3], j['ws'], True)
        ops.bn_relu_bwd(j['dA'], j['z'], j['dz'], j['st'][0], j['st'][1], j['st'][2], j['st'][3], j['outs'][0], j['outs'][1],
                        j['outs'][2], j['ws'], True)
    refs = []
    for j in jobs:                                   # alone
        launch(j)
        torch.cuda.synchronize()
        refs.append([j['a'].clone(), j['dz'].clone()] + [o.clone() for o in j['outs']])
    for _ in range(40):
        for j in jobs:
            j['a'].fill_(float('nan'))
            j['dz'].fill_(float('nan'))
        torch.cuda.synchronize()
        for j, s in zip(jobs, streams):
            with torch.cuda.stream(s):
                launch(j)
        torch.cuda.synchronize()
        for j, ref in zip(jobs, refs):
            for x, y in zip([j['a'], j['dz']] + j['outs'], ref):
                assert torch.equal(x, y)


@pytest.mark.parametrize('case', [(4, 1, 32, 256, 256), (4, 4, 64, 128, 128), (2, 3, 24, 40, 48), (4, 1, 128, 64, 64)])
def test_onepass_fwd_with_pooled_output(dev, case):
    """aide_bn_train_fwd_pool: the activation AND max_pool2d(activation, 2) from one launch == the plain forward (per group, in order)
    followed by aide_maxpool2x2_fwd, bit for bit -- activations, pooled tensor, running statistics, saved coefficients."""
    from aide_amd import ops
    from aide_amd._lib import lib
    m, groups, c, h, w = case
    n = m * groups
    assert lib.aide_bn_relu_bwd_pool_supported(m, c, h, w) == 1
    g = torch.Generator().manual_seed(c + h + groups)
    z = (torch.randn(n, c, h, w, generator=g) * 2.0 + 0.3).to(dev)
    res = []
    for fused in (False, True):
        bn = torch.nn.BatchNorm2d(c).to(dev)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=torch.Generator().manual_seed(7)) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=torch.Generator().manual_seed(8)) * 0.2)
        st = _stats(c, dev)
        ws = ops.bn_ws(c, dev)
        a = torch.empty_like(z)
        pooled = torch.full((n, c + 8, h // 2, w // 2), float('nan'), device=dev)[:, 8:]        # a channel slice of a wider buffer
        if fused:
            ops.bn_train_fwd_pool(z, a, pooled, groups, bn, st[0], st[1], st[2], st[3], ws)
        else:
            ops.bn_train_fwd_groups(z, a, groups, bn, st[0], st[1], st[2], st[3], ws)
            ops.maxpool2x2_fwd(a, pooled)
        res.append([a, pooled.clone(), bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()] + list(st))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert torch.equal(res[1][1], F.max_pool2d(res[1][0], 2))
