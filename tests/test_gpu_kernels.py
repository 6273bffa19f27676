"""Kernel-level parity (-m gpu): every C-ABI kernel family against the stock-aten CPU op it replaces
(the same ops the oracle in oracle/nets.py is built from). Tolerances: fp32, 1e-3 relative is the
north-star bound; these kernels are exact-fp32 FMA chains so we hold them to ~1e-5."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol=2e-5, atol=None, what=''):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    tol = rtol * scale if atol is None else atol
    assert err <= tol, '%s: max abs err %.3e > %.3e (ref scale %.3e)' % (what, err, tol, scale)


CONV_CASES = [
    # N, Cin, Cout, H, W
    (2, 3, 32, 32, 32), (1, 32, 32, 64, 64), (2, 64, 64, 32, 32), (1, 128, 64, 64, 64),
    (2, 64, 128, 16, 16), (1, 256, 256, 16, 16), (2, 24, 96, 20, 20), (1, 40, 32, 40, 24),
    (1, 64, 128, 80, 80), (4, 512, 512, 16, 16),
    # narrow layers of the UNet16 ... UNet2 variants (UNet.py:274-400): partial 32-channel tile, masked channel chunk
    (2, 3, 2, 32, 32), (2, 2, 2, 32, 32), (1, 4, 2, 64, 32), (2, 8, 4, 16, 16), (1, 16, 8, 20, 20), (2, 32, 16, 32, 32),
    (1, 16, 16, 64, 64), (2, 16, 32, 16, 16), (1, 2, 4, 40, 24),
    # stem layers on the folded-tap weight-gradient kernel (Ci <= 3, W % 64 == 0, H % 4 == 0)
    (2, 3, 32, 64, 64), (1, 3, 64, 8, 128), (3, 2, 40, 4, 64), (1, 1, 32, 12, 192), (4, 3, 32, 128, 128),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv3x3_fwd_dgrad_wgrad(dev, case):
    from aide_amd import ops
    n, ci, co, h, w = case
    g = torch.Generator().manual_seed(ci * 131 + co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3.0 * ci ** 0.5))
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=1)
    yr.backward(dy)

    xd, wd_, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wf, wdg = ops.pack_weights(wd_, need_dgrad=(ci % 32 == 0 or ci < 32 and ci != 3))
    y = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_igemm(xd, wf, bd, y)
    _close(y, yr, what='fwd %s' % (case,))
    for variant in range(6):                       # every tile variant must agree
        if variant in (2, 5) and co % 128:
            continue
        if variant != 0 and co % 64:
            continue
        if ci < 8 and variant > 1:
            continue
        for splitk in (1, 2):
            y2 = torch.zeros_like(y)
            ops.conv3x3_igemm(xd, wf, bd, y2, plan=variant | (splitk << 8))
            _close(y2, yr, what='fwd variant %d splitk %d %s' % (variant, splitk, case))
    if wdg is not None:
        dx = torch.empty(n, ci, h, w, device=dev)
        ops.conv3x3_igemm(dyd, wdg, None, dx)
        _close(dx, xr.grad, what='dgrad %s' % (case,))
        dx2 = torch.ones_like(dx)
        ops.conv3x3_igemm(dyd, wdg, None, dx2, accumulate=True)
        _close(dx2, xr.grad + 1.0, what='dgrad accumulate %s' % (case,))
    dw = torch.empty(co, ci, 3, 3, device=dev)
    ops.conv3x3_wgrad(dyd, xd, dw)
    _close(dw, wr.grad, what='wgrad %s' % (case,))


def test_conv3x3_channel_slices(dev):
    """inputs/outputs that are channel slices of concatenation buffers (torch.cat elimination)."""
    from aide_amd import ops
    g = torch.Generator().manual_seed(5)
    big_in = torch.randn(2, 96, 32, 32, generator=g)
    wt = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.conv2d(big_in[:, 32:64], wt, b, padding=1)
    bi = big_in.to(dev)
    big_out = torch.full((2, 160, 32, 32), 7.0, device=dev)
    wf, _ = ops.pack_weights(wt.to(dev))
    ops.conv3x3_igemm(bi[:, 32:64], wf, b.to(dev), big_out[:, 64:128])
    _close(big_out[:, 64:128], ref, what='slice conv')
    assert (big_out[:, :64] == 7.0).all() and (big_out[:, 128:] == 7.0).all()


@pytest.mark.parametrize('shape', [(4, 32, 64, 64), (2, 64, 16, 16), (3, 8, 20, 20)])
def test_bn_relu_fwd_bwd(dev, shape):
    from aide_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c)
    z = torch.randn(n, c, h, w, generator=g) * 2.0 + torch.randn(1, c, 1, 1, generator=g)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    dA = torch.randn(n, c, h, w, generator=g)
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    zr = z.clone().requires_grad_(True)
    ar = F.relu(bn(zr))
    ar.backward(dA)

    zd = z.to(dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    mean, rstd, scale, shift = (torch.empty(c, device=dev) for _ in range(4))
    ws = ops.bn_ws(c, dev)
    a = torch.empty_like(zd)
    ops.bn_train_fwd(zd, a, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rm, rv, nbt, mean, rstd, scale, shift, ws, True)
    a2 = torch.empty_like(zd)
    ops.bn_relu_apply(zd, a2, scale, shift, True)          # eval-path apply kernel with the same coefficients
    assert torch.equal(a, a2)
    _close(a, ar, what='bn+relu fwd')
    _close(rm, bn.running_mean, what='running_mean')
    _close(rv, bn.running_var, what='running_var')
    assert int(nbt.item()) == 1
    dz = torch.empty_like(zd)
    dg, db, dbias = (torch.empty(c, device=dev) for _ in range(3))
    ops.bn_relu_bwd(dA.to(dev), zd, dz, mean, rstd, scale, shift, dg, db, dbias, ws, True)
    _close(dz, zr.grad, rtol=5e-5, what='bn bwd dz')
    _close(dg, bn.weight.grad, rtol=5e-5, what='dgamma')
    _close(db, bn.bias.grad, rtol=5e-5, what='dbeta')
    assert dbias.abs().max().item() < 1e-3      # mathematically zero (dead conv bias)


def test_bn_more_planes_than_a_grid_dimension(dev):
    """N * C >= 65536 planes (a stacked batch of 4 x 33 images at C = 512 on the deepest level): the plane index of the BatchNorm
    apply kernels rides on gridDim.x, the one-pass kernels take channels * splits workgroups -- every form against aten"""
    from aide_amd import ops
    n, c, h, w = 132, 512, 2, 2
    g = torch.Generator().manual_seed(3)
    z = torch.randn(n, c, h, w, generator=g)
    bn = torch.nn.BatchNorm2d(c)
    bn.train()
    ar = F.relu(bn(z))
    zd = z.to(dev)
    st = [torch.empty(c, device=dev) for _ in range(4)]
    a = torch.empty_like(zd)
    ops.bn_train_fwd(zd, a, torch.ones(c, device=dev), torch.zeros(c, device=dev), 1e-5, 0.1, torch.zeros(c, device=dev),
                     torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev), st[0], st[1], st[2], st[3],
                     ops.bn_ws(c, dev), True)
    _close(a, ar, what='train forward')
    a2 = torch.empty_like(zd)
    ops.bn_relu_apply(zd, a2, st[2], st[3], True)          # grid = (planes, chunks)
    assert torch.equal(a, a2)


@pytest.mark.parametrize('shape', [(8, 64, 256, 256), (4, 128, 64, 64), (3, 8, 20, 20)])
def test_bn_relu_bwd_completion_event(dev, shape):
    """`done` of aide_bn_relu_bwd*: an event recorded when dz is complete, attached to the call's last dispatch (two-pass,
    single-kernel and scalar forms).  Another stream that waits for it must see all of dz: the consumer copies a NaN-primed
    dz while the producing stream goes on to overwrite the inputs; ten rounds back to back."""
    import ctypes
    from aide_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    z = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev)
    dA = torch.randn(n, c, h, w, generator=g).to(dev)
    mean, rstd, scale, shift = (torch.empty(c, device=dev) for _ in range(4))
    ws = ops.bn_ws(c, dev)
    ops.bn_train_fwd(z, torch.empty_like(z), torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3, 1e-5, 0.1,
                     torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev),
                     mean, rstd, scale, shift, ws, True)
    dg, db, dbias = (torch.empty(c, device=dev) for _ in range(3))
    ref = torch.empty_like(z)
    ops.bn_relu_bwd(dA, z, ref, mean, rstd, scale, shift, dg, db, dbias, ws, True)
    side = torch.cuda.Stream(device=dev)
    ev = ops.new_event()
    dz = torch.empty_like(z)
    outs = []
    torch.cuda.synchronize()
    for _ in range(10):
        dz.fill_(float('nan'))
        side.wait_stream(torch.cuda.current_stream())          # (the consumer's previous copy is done before the fill: separate check)
        ops.bn_relu_bwd(dA, z, dz, mean, rstd, scale, shift, dg, db, dbias, ws, True, done=ev)
        ops.wait(ctypes.c_void_p(side.cuda_stream), ev)
        with torch.cuda.stream(side):
            outs.append(dz.clone())
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


@pytest.mark.parametrize('case', [(4, 512, 16, 16, 8), (4, 256, 32, 32, 4), (2, 128, 64, 64, 2)])
def test_bn_relu_bwd_from_splitk_slabs(dev, case):
    """BatchNorm backward that reads its dA from the split-K slabs of the data-gradient convolution (the engine's
    conv2 -> conv1 pairs at the 64x64 .. 16x16 levels): bit-identical to the split reduce followed by bn_relu_bwd, and the
    F(4x4) / F(2x2) kernels leave exactly those slabs with accumulate = 2."""
    from aide_amd import ops
    from aide_amd._lib import lib
    n, c, h, w, splitk = case
    assert lib.aide_bn_two_pass(n, c, h, w) == 0
    g = torch.Generator().manual_seed(c + h)
    z = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev)
    slabs = torch.randn(splitk, n, c, h, w, generator=g).to(dev)
    dA = slabs[0].clone()
    for s in range(1, splitk):
        dA += slabs[s]                                     # the order of the split reduce
    mean, rstd, scale, shift = (torch.empty(c, device=dev) for _ in range(4))
    ws = ops.bn_ws(c, dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    ops.bn_train_fwd(z, torch.empty_like(z), torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3, 1e-5, 0.1,
                     rm, rv, nbt, mean, rstd, scale, shift, ws, True)
    ref = [torch.empty_like(z)] + [torch.empty(c, device=dev) for _ in range(3)]
    ops.bn_relu_bwd(dA, z, ref[0], mean, rstd, scale, shift, ref[1], ref[2], ref[3], ws, True)
    out = [torch.empty_like(z)] + [torch.empty(c, device=dev) for _ in range(3)]
    ops.bn_relu_bwd_slabs(slabs, splitk, z, out[0], mean, rstd, scale, shift, out[1], out[2], out[3], ws, True)
    for a, b, what in zip(out, ref, ('dz', 'dgamma', 'dbeta', 'dbias')):
        assert torch.equal(a, b), what
    # the convolution side: accumulate = 2 leaves the slabs, their sum in split order is the reduced result
    ci = c
    x = torch.randn(n, ci, h, w, generator=g).to(dev)
    wt = (torch.randn(c, ci, 3, 3, generator=g) * 0.05).to(dev)
    for pack, conv, sk in ((ops.wino4_pack, ops.conv3x3_wino4, lib.aide_conv3x3_wino4_splitk(n, ci, h, w, c)),
                           (ops.wino_pack, ops.conv3x3_wino, lib.aide_conv3x3_wino_splitk(n, ci, h, w, c))):
        if conv is ops.conv3x3_wino4 and not ops.wino4_supported(ci, h, w, c):
            continue
        if sk <= 1:
            continue
        u, _ = pack(wt)
        wsk = torch.zeros(sk * n * c * h * w, device=dev)
        y = conv(x, u, None, torch.empty(n, c, h, w, device=dev), splitk=sk, ws=wsk)
        wsk2 = torch.zeros_like(wsk)
        conv(x, u, None, torch.empty(n, c, h, w, device=dev), accumulate=2, splitk=sk, ws=wsk2)
        sl = wsk2.view(sk, n, c, h, w)
        acc = sl[0].clone()
        for s in range(1, sk):
            acc += sl[s]
        assert torch.equal(acc, y)


def test_maxpool_ties_and_backward(dev):
    from aide_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 16, 24, generator=g)
    x[:, :, :6, :] = 0.25                  # constant background -> exact ties in every window
    x[0, 0, 8:10, 4:6] = 1.5
    dy = torch.randn(2, 8, 8, 12, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    yr.backward(dy)
    xd = x.to(dev)
    y = torch.empty(2, 8, 8, 12, device=dev)
    ops.maxpool2x2_fwd(xd, y)
    assert torch.equal(y.cpu(), yr.detach())
    dx = torch.empty_like(xd)
    ops.maxpool2x2_bwd(xd, dy.to(dev), dx, accumulate=False)
    assert torch.equal(dx.cpu(), xr.grad)          # first-max tie rule, bit exact
    dx2 = torch.ones_like(xd)
    ops.maxpool2x2_bwd(xd, dy.to(dev), dx2, accumulate=True)
    assert torch.equal(dx2.cpu(), xr.grad + 1.0)


@pytest.mark.parametrize('shape', [(2, 8, 16, 16), (1, 4, 20, 12), (2, 3, 1, 1),
                                   (1, 3, 64, 64), (2, 2, 16, 128), (1, 2, 48, 192),      # whole 32 x 128 output tiles
                                   (132, 512, 2, 2)])      # 67 584 planes: more than gridDim.y holds (a stacked batch at C = 512)
def test_upsample_bilinear(dev, shape):
    from aide_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h)
    x = torch.randn(n, c, h, w, generator=g)
    dy = torch.randn(n, c, 2 * h, 2 * w, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
    yr.backward(dy)
    y = torch.empty(n, c, 2 * h, 2 * w, device=dev)
    ops.upsample2x_fwd(x.to(dev), y)
    _close(y, yr, rtol=5e-6, what='upsample fwd')
    dx = torch.empty(n, c, h, w, device=dev)
    ops.upsample2x_bwd(dy.to(dev), dx)
    _close(dx, xr.grad, rtol=1e-5, what='upsample bwd')
    # the tiled backward reads aligned 16-byte pieces when it can: a gradient that starts 4 bytes into an allocation takes
    # the other loader and must give the same values bit for bit, as must the accumulating form on top of ones
    if (2 * w) % 2 == 0:
        raw = torch.empty(dy.numel() + 4, device=dev)
        dyo = raw[1:1 + dy.numel()].view(dy.shape)
        dyo.copy_(dy.to(dev))
        dx2 = torch.empty(n, c, h, w, device=dev)
        ops.upsample2x_bwd(dyo, dx2)
        assert torch.equal(dx2, dx)
        dx3 = torch.ones(n, c, h, w, device=dev)
        ops.upsample2x_bwd(dy.to(dev), dx3, accumulate=True)
        assert torch.equal(dx3, dx + 1.0)


@pytest.mark.parametrize('case', [(2, 64, 32, 16, 16), (1, 128, 64, 8, 12), (2, 24, 40, 10, 10)])
def test_convT2x2(dev, case):
    from aide_amd import ops
    n, ci, co, h, w = case
    g = torch.Generator().manual_seed(ci)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(ci, co, 2, 2, generator=g) * 0.1
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, 2 * h, 2 * w, generator=g)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, b, stride=2)
    yr.backward(dy)
    y = torch.empty(n, co, 2 * h, 2 * w, device=dev)
    ops.convT2x2_fwd(x.to(dev), wt.to(dev), b.to(dev), y)
    _close(y, yr, what='convT fwd')
    dx = torch.empty(n, ci, h, w, device=dev)
    ops.convT2x2_dgrad(dy.to(dev), wt.to(dev), dx)
    _close(dx, xr.grad, what='convT dgrad')
    dw = torch.empty(ci, co, 2, 2, device=dev)
    ops.convT2x2_wgrad(x.to(dev), dy.to(dev), dw)
    _close(dw, wr.grad, what='convT wgrad')


def test_head1x1(dev):
    from aide_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 32, 32, generator=g)
    wt = torch.randn(2, 64, 1, 1, generator=g) * 0.2
    b = torch.randn(2, generator=g)
    dy = torch.randn(2, 2, 32, 32, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    yr.backward(dy)
    y = torch.empty(2, 2, 32, 32, device=dev)
    w2 = wt.view(2, 64).to(dev)
    ops.head1x1_fwd(x.to(dev), w2, b.to(dev), y)
    _close(y, yr, what='head fwd')
    dx = torch.empty(2, 64, 32, 32, device=dev)
    dw, db = torch.empty(2, 64, device=dev), torch.empty(2, device=dev)
    ops.head1x1_bwd(dy.to(dev), x.to(dev), w2, dx, dw, db)
    _close(dx, xr.grad, what='head dgrad')
    _close(dw, wr.grad.view(2, 64), what='head wgrad')
    _close(db, br.grad, what='head dbias')


WINO_CASES = [(2, 64, 64, 32, 32), (1, 128, 64, 64, 64), (2, 64, 128, 16, 16), (1, 256, 256, 16, 16),
              (1, 64, 128, 80, 80), (4, 512, 512, 16, 16), (1, 32, 64, 40, 24), (2, 8, 64, 6, 12)]


@pytest.mark.parametrize('case', WINO_CASES)
def test_conv3x3_winograd_fwd_dgrad(dev, case):
    """Winograd F(2x2,3x3) forward / dgrad vs aten: same 2e-5 bound as the direct kernels."""
    from aide_amd import ops
    n, ci, co, h, w = case
    assert ops.wino_supported(ci, h, w, co)
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3.0 * ci ** 0.5))
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wt, b, padding=1)
    yr.backward(dy)
    uf, ud = ops.wino_pack(wt.to(dev), need_dgrad=(ci % 64 == 0))
    for splitk in (1, 2):
        if splitk > ci // 8:
            continue
        y = torch.full((n, co, h, w), 3.0, device=dev)
        ops.conv3x3_wino(x.to(dev), uf, b.to(dev), y, splitk=splitk)
        _close(y, yr, what='wino fwd splitk %d %s' % (splitk, case))
    y = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wino(x.to(dev), uf, b.to(dev), y)                 # auto split-K
    _close(y, yr, what='wino fwd auto %s' % (case,))
    if ud is not None:
        dx = torch.ones(n, ci, h, w, device=dev)
        ops.conv3x3_wino(dy.to(dev), ud, None, dx, accumulate=True)
        _close(dx, xr.grad + 1.0, what='wino dgrad accumulate %s' % (case,))


@pytest.mark.parametrize('case', [(2, 64, 64, 32, 32), (1, 128, 64, 64, 64), (2, 16, 128, 16, 32), (1, 256, 256, 32, 32),
                                  (1, 64, 128, 80, 80), (1, 8, 64, 20, 44), (1, 96, 192, 40, 48), (2, 40, 64, 48, 36),
                                  (2, 32, 32, 32, 64), (1, 64, 96, 16, 32),
                                  (3, 64, 64, 20, 20), (1, 32, 96, 24, 28), (2, 64, 32, 36, 24),   # 16 < W < 32: one masked tile column
                                  (2, 64, 64, 40, 40), (1, 64, 32, 60, 100), (2, 32, 64, 44, 40),    # 20 x 20 canvas tiles (fewer slots than 16 x 32)
                                  (4, 64, 64, 16, 16), (2, 128, 96, 32, 16), (6, 32, 64, 20, 16)])   # 16-wide: image pairs per tile
def test_conv3x3_winograd4_fwd_dgrad(dev, case):
    """Winograd F(4x4,3x3) forward / dgrad vs aten, incl. ragged block edges (H % 16, W % 32 != 0), split-K,
    accumulate; 16-pixel-wide images run two per workgroup tile (the 16 x 16 bottleneck level).  F(4x4) transform constants reach 8: the bound is 1e-4 of the output scale (north star: 1e-3)."""
    from aide_amd import ops
    n, ci, co, h, w = case
    assert ops.wino4_supported(ci, h, w, co)
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3.0 * ci ** 0.5))
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wt, b, padding=1)
    yr.backward(dy)
    uf, ud = ops.wino4_pack(wt.to(dev), need_dgrad=(ci % 32 == 0))
    for splitk in (1, 2, 4):
        if (ci // 8) % splitk:
            continue
        y = torch.full((n, co, h, w), 3.0, device=dev)
        ops.conv3x3_wino4(x.to(dev), uf, b.to(dev), y, splitk=splitk)
        _close(y, yr, rtol=1e-4, what='wino4 fwd splitk %d %s' % (splitk, case))
    y = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wino4(x.to(dev), uf, b.to(dev), y)                # auto split-K
    _close(y, yr, rtol=1e-4, what='wino4 fwd auto %s' % (case,))
    if ud is not None:
        dx = torch.ones(n, ci, h, w, device=dev)
        ops.conv3x3_wino4(dy.to(dev), ud, None, dx, accumulate=True)
        _close(dx, xr.grad + 1.0, rtol=1e-4, what='wino4 dgrad accumulate %s' % (case,))


@pytest.mark.parametrize('case', [(4, 64, 64, 32, 32, 2), (2, 128, 64, 64, 64, 1), (6, 16, 128, 16, 32, 3), (4, 64, 128, 80, 80, 2),
                                  (2, 96, 192, 40, 48, 2), (4, 64, 64, 40, 40, 4), (2, 64, 32, 60, 100, 1),      # canvas tiles
                                  (3, 32, 96, 24, 28, 1), (2, 1024, 64, 16, 32, 2)])
def test_conv3x3_winograd4_input_batchnorm(dev, case):
    """F(4x4) forward whose loader applies relu(x * scale + shift) with a per-group, per-input-channel table (in_bn_tab:
    the producing layer's BatchNorm + ReLU on the way in) == the convolution of the materialised activation, BIT FOR BIT
    (same fmaf / max per element, same kernel after the staging) -- zero padding included: BN(0) != 0, the halo must stay 0.
    Negative scales, split-K, both tile shapes, a 1024-channel input (the table limit), several image groups."""
    from aide_amd import ops
    n, ci, co, h, w, groups = case
    g = torch.Generator().manual_seed(ci + 13 * co + h)
    z = torch.randn(n, ci, h, w, generator=g).to(dev)
    wt = (torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3.0 * ci ** 0.5))).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    tab = torch.randn(groups, ci, 2, generator=g).to(dev)          # (scale, shift): both signs
    tab[0, 0, 0] = 0.0
    ng = n // groups
    act = torch.empty_like(z)
    for gi in range(groups):
        sl = slice(gi * ng, (gi + 1) * ng)
        # the apply kernel's arithmetic: fmaf(z, scale, shift), then max(., 0)
        act[sl] = torch.clamp_min(torch.addcmul(tab[gi, :, 1].view(1, ci, 1, 1), z[sl], tab[gi, :, 0].view(1, ci, 1, 1)), 0.0)
    from aide_amd import ops as O
    a2 = torch.empty_like(z)                                       # ... through the library's own apply kernel, per group
    for gi in range(groups):
        sl = slice(gi * ng, (gi + 1) * ng)
        O.bn_relu_apply(z[sl], a2[sl], tab[gi, :, 0].contiguous(), tab[gi, :, 1].contiguous(), True)
    uf, _ = ops.wino4_pack(wt, need_dgrad=False)
    for splitk in (1, 2):
        if (ci // 8) % splitk:
            continue
        y_ref = torch.empty(n, co, h, w, device=dev)
        ops.conv3x3_wino4(a2, uf, b, y_ref, splitk=splitk)
        y = torch.full((n, co, h, w), 7.0, device=dev)
        ops.conv3x3_wino4(z, uf, b, y, splitk=splitk, in_tab=tab, in_group_images=ng)
        assert torch.equal(y, y_ref), 'input BatchNorm in the loader differs from the materialised form (splitk %d) %s' % (splitk, case)
    yr = F.conv2d(act.cpu(), wt.cpu(), b.cpu(), padding=1)
    _close(y, yr, rtol=1e-4, what='wino4 with input BatchNorm vs aten %s' % (case,))


@pytest.mark.parametrize('case', [(2, 64, 64, 32, 32), (1, 128, 64, 64, 64), (2, 64, 128, 16, 16), (1, 256, 256, 16, 16),
                                  (1, 64, 128, 80, 80), (4, 512, 512, 16, 16), (1, 96, 160, 40, 24), (2, 64, 64, 6, 12)])
def test_conv3x3_wgrad_winograd(dev, case):
    """Winograd weight gradient (transposed F(2x2,3x3)) vs aten, incl. ragged widths and partial channel tiles."""
    from aide_amd import ops
    n, ci, co, h, w = case
    assert ops.wgrad_wino_supported(co, ci, h, w)
    g = torch.Generator().manual_seed(ci + 3 * co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).requires_grad_(True)
    dy = torch.randn(n, co, h, w, generator=g)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    ops.conv3x3_wgrad_wino(dy.to(dev), x.to(dev), dw)
    _close(dw, wt.grad, what='wino wgrad %s' % (case,))
    # channel-slice operands (concatenation buffers)
    big = torch.randn(n, ci + 32, h, w, generator=g)
    wt2 = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).requires_grad_(True)
    F.conv2d(big[:, 32:], wt2, None, padding=1).backward(dy)
    ops.conv3x3_wgrad_wino(dy.to(dev), big.to(dev)[:, 32:], dw)
    _close(dw, wt2.grad, what='wino wgrad slice %s' % (case,))


@pytest.mark.parametrize('case', [(2, 64, 64, 32, 32), (1, 128, 64, 64, 64), (2, 32, 128, 16, 16), (1, 256, 256, 32, 32),
                                  (1, 64, 128, 80, 80), (3, 32, 64, 8, 16), (1, 96, 192, 40, 24), (2, 64, 64, 12, 36),
                                  (2, 32, 32, 32, 32), (1, 64, 96, 16, 48), (1, 32, 160, 8, 16)])   # trailing half co tile
def test_conv3x3_wgrad_winograd4(dev, case):
    """Weight gradient via the transposed Winograd F(4x4,3x3) vs aten, incl. ragged column blocks (W % 16 != 0),
    odd chunk counts per split and channel-slice operands.  Bound 1e-4 of the gradient scale (north star: 1e-3)."""
    from aide_amd import ops
    n, ci, co, h, w = case
    assert ops.wgrad_wino4_supported(co, ci, h, w)
    g = torch.Generator().manual_seed(ci + 3 * co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).requires_grad_(True)
    dy = torch.randn(n, co, h, w, generator=g)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    ops.conv3x3_wgrad_wino4(dy.to(dev), x.to(dev), dw)
    _close(dw, wt.grad, rtol=1e-4, what='wino4 wgrad %s' % (case,))
    big = torch.randn(n, ci + 32, h, w, generator=g)
    wt2 = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).requires_grad_(True)
    F.conv2d(big[:, 32:], wt2, None, padding=1).backward(dy)
    ops.conv3x3_wgrad_wino4(dy.to(dev), big.to(dev)[:, 32:], dw)
    _close(dw, wt2.grad, rtol=1e-4, what='wino4 wgrad slice %s' % (case,))


FULL_LAYERS = [(1024, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 64, 256)]   # Ci, Co, H (= W), N = 4


@pytest.mark.parametrize('layer', FULL_LAYERS)
def test_full_size_layers_winograd_vs_direct(dev, layer):
    """BASELINE config 2 layer sizes (N = 4): the three independent implementations of the same operator — F(4x4,3x3),
    F(2x2,3x3) and the direct implicit GEMM — must agree (forward, dgrad, wgrad; 1e-4 of the result scale), the F(4x4) kernels
    must be linear in their input, and image 0 of every direction is held to aten's float64 convolution on the CPU."""
    from aide_amd import ops
    ci, co, h = layer
    n = 4
    g = torch.Generator(device='cpu').manual_seed(ci + h)
    x = torch.randn(n, ci, h, h, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3.0 * ci ** 0.5))).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    dy = torch.randn(n, co, h, h, generator=g).to(dev)
    wf, wd = ops.pack_weights(w)
    u2f, u2d = ops.wino_pack(w)
    u4f, u4d = ops.wino4_pack(w)
    y0 = ops.conv3x3_igemm(x, wf, b, torch.empty(n, co, h, h, device=dev))
    y2 = ops.conv3x3_wino(x, u2f, b, torch.empty_like(y0))
    y4 = ops.conv3x3_wino4(x, u4f, b, torch.empty_like(y0))
    _close(y2, y0, rtol=1e-4, what='F2 vs direct fwd %s' % (layer,))
    _close(y4, y0, rtol=1e-4, what='F4 vs direct fwd %s' % (layer,))
    # linearity: conv(2 x1 - x2) == 2 conv(x1) - conv(x2) (no bias)
    x2 = torch.randn(n, ci, h, h, generator=g).to(dev)
    ya = ops.conv3x3_wino4(x, u4f, None, torch.empty_like(y0))
    yb = ops.conv3x3_wino4(x2, u4f, None, torch.empty_like(y0))
    yc = ops.conv3x3_wino4(2 * x - x2, u4f, None, torch.empty_like(y0))
    _close(yc, 2 * ya - yb, rtol=1e-4, what='F4 linearity %s' % (layer,))
    d0 = ops.conv3x3_igemm(dy, wd, None, torch.empty(n, ci, h, h, device=dev))
    d4 = ops.conv3x3_wino4(dy, u4d, None, torch.empty_like(d0))
    _close(d4, d0, rtol=1e-4, what='F4 vs direct dgrad %s' % (layer,))
    g0 = ops.conv3x3_wgrad(dy, x, torch.empty_like(w))
    g2 = ops.conv3x3_wgrad_wino(dy, x, torch.empty_like(w))
    g4 = ops.conv3x3_wgrad_wino4(dy, x, torch.empty_like(w))
    _close(g2, g0, rtol=1e-4, what='F2 vs direct wgrad %s' % (layer,))
    _close(g4, g0, rtol=1e-4, what='F4 vs direct wgrad %s' % (layer,))
    # determinism: a second launch is bit-identical (fixed-order split reductions, no atomics)
    assert torch.equal(ops.conv3x3_wgrad_wino4(dy, x, torch.empty_like(w)), g4)
    assert torch.equal(ops.conv3x3_wino4(x, u4f, b, torch.empty_like(y0)), y4)
    # ... and an anchor OUTSIDE this library, so that the comparison above is not a self-comparison: image 0 of the forward, the
    # data gradient and the weight gradient against aten's float64 convolution on the CPU (one image bounds the CPU time)
    xc, wc, bc, dyc = x[:1].double().cpu(), w.double().cpu(), b.double().cpu(), dy[:1].double().cpu()
    ry = F.conv2d(xc, wc, bc, padding=1)
    _close(y4[:1], ry, rtol=1e-4, what='F4 vs aten float64 fwd %s' % (layer,))
    _close(y0[:1], ry, rtol=2e-5, what='direct vs aten float64 fwd %s' % (layer,))
    rd = torch.nn.grad.conv2d_input(xc.shape, wc, dyc, padding=1)
    _close(d4[:1], rd, rtol=1e-4, what='F4 vs aten float64 dgrad %s' % (layer,))
    rg = torch.nn.grad.conv2d_weight(xc, wc.shape, dyc, padding=1)
    g4_1 = ops.conv3x3_wgrad_wino4(dy[:1], x[:1], torch.empty_like(w))
    _close(g4_1, rg, rtol=1e-4, what='F4 vs aten float64 wgrad %s' % (layer,))
