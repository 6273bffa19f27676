"""Generate tests/golden/*.npz by importing the REAL reference (/root/reference).

Run only in the build container (the reference does not travel to the GPU box):

    python -B oracle/gen_golden.py

For every fixture it (1) runs the imported reference, (2) runs this repo's oracle
restatement on the same seeds/inputs and asserts bit-equality on CPU, and (3) stores the
reference's inputs/outputs as data.  Only numbers are stored — no reference source.
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')


def _import_reference():
    """Import reference packages under private names so they do not shadow ours."""
    import importlib
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, REF)
    ref_f = importlib.import_module('models_twomodalinputs')
    ref_u = importlib.import_module('models_singlemodalinput')
    ref_utils = importlib.import_module('utils')
    sys.path.remove(REF)
    return ref_f, ref_u, ref_utils


def _np(t):
    return t.detach().cpu().numpy()


def _sub(t, limit=8192):
    """Full tensor when small, else the deterministic flat subsample t.flatten()[::stride]
    with stride = ceil(numel/limit) (tests recompute the same stride)."""
    a = _np(t)
    if a.size <= limit:
        return a
    stride = -(-a.size // limit)
    return a.reshape(-1)[::stride].copy()


def _same(a, b, what):
    assert torch.equal(a, b), 'oracle != reference for %s (max abs diff %g)' % (
        what, (a.double() - b.double()).abs().max().item())


def tiny_inputs(two_modal, n=2, size=32):
    g = torch.Generator().manual_seed(1234)
    xs = [torch.randn(n, 3, size, size, generator=g) for _ in range(2 if two_modal else 1)]
    t = (torch.rand(n, size, size, generator=g) > 0.7).long()
    return xs, t


def g1_model(name, ref_ctor, ora_ctor, kwargs, two_modal, ref_utils):
    import oracle
    fx = {}
    torch.manual_seed(2)
    rnet = ref_ctor(2, **kwargs)
    torch.manual_seed(2)
    onet = ora_ctor(2, **kwargs)
    rsd, osd = rnet.state_dict(), onet.state_dict()
    assert list(rsd.keys()) == list(osd.keys()), 'state_dict keys differ for ' + name
    for k in rsd:
        _same(rsd[k], osd[k], name + ' init ' + k)
    xs, t = tiny_inputs(two_modal)
    w = torch.tensor([1.0, 1.0])
    res = {}
    for tag, net, crit_mod in (('ref', rnet, ref_utils), ('ora', onet, oracle)):
        net.train()
        out = net(*xs)
        crit = crit_mod.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
        crit_i = crit_mod.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)
        loss = crit(out, t)
        per_img = crit_i(out, t)
        net.zero_grad()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, amsgrad=True)
        opt.step()
        after1 = {k: p.detach().clone() for k, p in net.named_parameters()}
        bufs = {k: b.clone() for k, b in net.named_buffers()}
        net.eval()
        with torch.no_grad():
            ev = net(*xs)
        res[tag] = dict(out=out.detach(), loss=loss.detach(), per_img=per_img.detach(),
                        grads=grads, after1=after1, bufs=bufs, ev=ev)
    r, o = res['ref'], res['ora']
    for k in ('out', 'loss', 'per_img', 'ev'):
        _same(r[k], o[k], name + ' ' + k)
    for k in r['grads']:
        _same(r['grads'][k], o['grads'][k], name + ' grad ' + k)
        _same(r['after1'][k], o['after1'][k], name + ' adam1 ' + k)
    for k in r['bufs']:
        _same(r['bufs'][k], o['bufs'][k], name + ' buf ' + k)
    names = list(r['grads'].keys())
    for i, x in enumerate(xs):
        fx['x%d' % i] = _np(x)
    fx['targets'] = _np(t)
    fx['logits'] = _np(r['out'])
    fx['loss'] = _np(r['loss'])
    fx['per_image_loss'] = _np(r['per_img'])
    fx['eval_logits'] = _np(r['ev'])
    fx['param_names'] = np.array(names)
    fx['grad_norms'] = np.array([r['grads'][k].double().norm().item() for k in names])
    fx['grad_absmax'] = np.array([r['grads'][k].abs().max().item() for k in names])
    fx['param_sum'] = np.array(sum(p.double().sum().item() for p in rnet.state_dict().values()
                                   if p.dtype.is_floating_point))
    first = names[0]
    head_w = 'last_conv1.weight'
    full = [first, names[1], head_w, 'last_conv1.bias',
            [k for k in names if k.startswith('up_block4.block.conv2.weight')][0],
            [k for k in names if k.startswith('up_block1.bilinear_up') and k.endswith('weight')][0]]
    for k in full:
        fx['grad/' + k] = _sub(r['grads'][k])
        fx['adam1/' + k] = _sub(r['after1'][k])
    for k, b in r['bufs'].items():
        if k.startswith('up_block4.block.bn2') or k.startswith(first.rsplit('.', 2)[0] + '.bn1'):
            fx['buf/' + k] = _np(b)
    np.savez_compressed(os.path.join(OUT, 'g1_%s.npz' % name), **fx)
    print('g1', name, 'loss', float(r['loss']), 'sum logits', float(r['out'].double().sum()),
          'gradL2', float(np.sqrt((fx['grad_norms'] ** 2).sum())))


def g3_losses(ref_utils):
    import oracle
    from aide_amd.synthetic import chaos_batch
    fx = {}
    g = torch.Generator().manual_seed(77)
    n, s = 4, 64
    _, _, t = chaos_batch(n, s, seed=5)
    t[0, 10:30, 12:40] = 1                      # guarantee non-empty images too
    z1 = torch.randn(n, 2, s, s, generator=g) * 2.0
    z2 = torch.randn(n, 2, s, s, generator=g) * 2.0
    # make the per-image losses well separated so that the argsort is well defined
    z1 += torch.tensor([0.0, 0.6, -0.5, 1.1]).view(n, 1, 1, 1) * (2 * t.unsqueeze(1).float() - 1) * \
        torch.tensor([-1.0, 1.0]).view(1, 2, 1, 1)
    z2 += torch.tensor([0.9, -0.4, 0.3, -1.0]).view(n, 1, 1, 1) * (2 * t.unsqueeze(1).float() - 1) * \
        torch.tensor([-1.0, 1.0]).view(1, 2, 1, 1)
    pseudo = torch.softmax(torch.randn(n, 2, s, s, generator=g), dim=1)
    wmap = (1.0 - 4.0 * pseudo[:, 0] * pseudo[:, 1]).unsqueeze(1)
    fx.update(z1=_np(z1), z2=_np(z2), targets=_np(t), pseudo=_np(pseudo), wmap=_np(wmap))
    for wname, cw in (('w11', torch.tensor([1.0, 1.0])), ('w13', torch.tensor([1.0, 3.0]))):
        cdw = torch.tensor([1.0, 1.0]) if wname == 'w11' else torch.tensor([0.7, 1.6])
        for lname, kw in (('CrossEntropyLoss2d', dict(weight=cw)),
                          ('MulticlassDiceLoss', dict(weight=cw)),
                          ('DiceLoss', dict()),
                          ('CEMDiceLoss', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw)),
                          ('CEMDiceLossImage', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw))):
            vals = {}
            for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                zz = z1.clone().requires_grad_(True)
                v = getattr(mod, lname)(**kw)(zz, t)
                (v.sum() if v.dim() else v).backward()
                vals[tag] = (v.detach(), zz.grad.clone())
            _same(vals['ref'][0], vals['ora'][0], lname + wname)
            _same(vals['ref'][1], vals['ora'][1], lname + wname + ' grad')
            fx['%s/%s' % (lname, wname)] = _np(vals['ref'][0])
            fx['%s/%s/grad' % (lname, wname)] = _np(vals['ref'][1])
    # CE 'none' reduction (per-pixel map) and 'sum'
    for red in ('none', 'sum'):
        a = ref_utils.CrossEntropyLoss2d(weight=torch.tensor([1.0, 3.0]), reduction=red)(z1, t)
        b = oracle.CrossEntropyLoss2d(weight=torch.tensor([1.0, 3.0]), reduction=red)(z1, t)
        _same(a, b, 'CE ' + red)
        fx['CrossEntropyLoss2d/w13/' + red] = _np(a)
    # MSE consistency (a13) with weightmap, caller-side mean
    vals = {}
    for tag, mod in (('ref', ref_utils), ('ora', oracle)):
        zz = z1.clone().requires_grad_(True)
        v = (wmap * mod.MulticlassMSELoss(reduction='none')(zz, pseudo)).mean()
        v.backward()
        vals[tag] = (v.detach(), zz.grad.clone())
    _same(vals['ref'][0], vals['ora'][0], 'mse')
    _same(vals['ref'][1], vals['ora'][1], 'mse grad')
    fx['mse_wm_mean'] = _np(vals['ref'][0])
    fx['mse_wm_mean/grad'] = _np(vals['ref'][1])
    # co-teaching operators (a16, a17-weightimage)
    for cname in ('Coteachingloss_dropimage', 'Coteachingloss_weightimage'):
        for fr in (0.0, 0.25, 0.5):
            vals = {}
            for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                a1 = z1.clone().requires_grad_(True)
                a2 = z2.clone().requires_grad_(True)
                l1, l2 = getattr(mod, cname)(weight=1.0, reduction='none')(a1, a2, t, fr)
                (l1 + l2).backward()
                vals[tag] = (l1.detach(), l2.detach(), a1.grad.clone(), a2.grad.clone())
            for i in range(4):
                _same(vals['ref'][i], vals['ora'][i], '%s fr=%g #%d' % (cname, fr, i))
            key = '%s/fr%g' % (cname, fr)
            fx[key + '/loss1'], fx[key + '/loss2'] = _np(vals['ref'][0]), _np(vals['ref'][1])
            fx[key + '/grad1'], fx[key + '/grad2'] = _np(vals['ref'][2]), _np(vals['ref'][3])
    crit = ref_utils.Coteachingloss_dropimage(weight=1.0, reduction='none')
    l1 = crit.weight * torch.mean(crit.ce(z1, t), dim=[1, 2]) + crit.dice(z1, t)
    l2 = crit.weight * torch.mean(crit.ce(z2, t), dim=[1, 2]) + crit.dice(z2, t)
    fx['coteach/per_image1'], fx['coteach/per_image2'] = _np(l1), _np(l2)
    fx['coteach/argsort1'] = np.asarray(np.argsort(l1.cpu().data))
    fx['coteach/argsort2'] = np.asarray(np.argsort(l2.cpu().data))
    fx['coteach/min_gap1'] = np.diff(np.sort(_np(l1))).min()
    fx['coteach/min_gap2'] = np.diff(np.sort(_np(l2))).min()
    # hard Dice metric (sum over the batch)
    d_ref = ref_utils.Dice_fn(z1.clone(), t)
    d_ora = oracle.Dice_fn(z1.clone(), t)
    _same(torch.as_tensor(d_ref), torch.as_tensor(d_ora), 'Dice_fn')
    fx['Dice_fn'] = _np(torch.as_tensor(d_ref))
    np.savez_compressed(os.path.join(OUT, 'g3_losses.npz'), **fx)
    print('g3 losses ok; coteach gaps', fx['coteach/min_gap1'], fx['coteach/min_gap2'])


def g4_proposed(ref_f, ref_utils):
    """Proposed co-teaching step on tiny inputs: reference modules/losses driven by the restated
    step (oracle.steps), compared with the all-oracle run."""
    import oracle
    from oracle import steps
    fx = {}
    n, s = 4, 32
    g = torch.Generator().manual_seed(4321)
    xin = torch.randn(n, 3, s, s, generator=g)
    xout = torch.randn(n, 3, s, s, generator=g)
    t1 = (torch.rand(n, s, s, generator=g) > 0.7).long()
    t2 = (torch.rand(n, s, s, generator=g) > 0.65).long()
    augs = [(xin + 0.05 * torch.randn(n, 3, s, s, generator=g),
             xout + 0.05 * torch.randn(n, 3, s, s, generator=g)) for _ in range(4)]
    fx.update(xin=_np(xin), xout=_np(xout), t1=_np(t1), t2=_np(t2))
    for i, (a, b) in enumerate(augs):
        fx['aug%d_in' % i], fx['aug%d_out' % i] = _np(a), _np(b)
    w = torch.tensor([1.0, 1.0])
    for rate in (0.0, 0.25, 1.0):
        res = {}
        for tag, fmod, umod in (('ref', ref_f, ref_utils), ('ora', oracle, oracle)):
            torch.manual_seed(2)
            net1 = fmod.fuseunet(2)
            net2 = fmod.fuseunet(2)
            net1.train(), net2.train()
            crit = umod.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)
            corr = umod.MulticlassMSELoss(reduction='none')
            o1 = torch.optim.Adam(net1.parameters(), lr=1e-4, amsgrad=True)
            o2 = torch.optim.Adam(net2.parameters(), lr=1e-4, amsgrad=True)
            r = steps.proposed_step(net1, net2, crit, corr, o1, o2, xin, xout, augs, t1, t2, rate)
            r['g1'] = torch.stack([p.grad.double().norm() for p in net1.parameters()])
            r['g2'] = torch.stack([p.grad.double().norm() for p in net2.parameters()])
            r['nbt'] = net1.modal1_downblock1.block.bn1.num_batches_tracked.clone()
            r['head1'] = net1.last_conv1.weight.detach().clone()
            res[tag] = r
        for k in ('outputs1', 'outputs2', 'loss1', 'loss2', 'indx1', 'indx2', 'g1', 'g2', 'head1'):
            _same(res['ref'][k], res['ora'][k], 'proposed r=%g %s' % (rate, k))
        r = res['ref']
        key = 'r%g/' % rate
        for k in ('outputs1', 'outputs2', 'loss1', 'loss2', 'indx1', 'indx2', 'loss1_pre',
                  'loss2_pre', 'g1', 'g2', 'nbt', 'head1'):
            fx[key + k] = _np(r[k])
        fx[key + 'min_gap1'] = np.diff(np.sort(_np(r['loss1_pre']))).min()
        fx[key + 'min_gap2'] = np.diff(np.sort(_np(r['loss2_pre']))).min()
        print('g4 r=%g' % rate, float(r['loss1']), float(r['loss2']), r['indx1'].tolist(),
              r['indx2'].tolist(), 'gaps', fx[key + 'min_gap1'], fx[key + 'min_gap2'])
    np.savez_compressed(os.path.join(OUT, 'g4_proposed.npz'), **fx)


def g5_adam(ref_f, ref_utils):
    """Three comparison steps (trainchaos_comparison_1case.py:190-199) on the tiny inputs."""
    import oracle
    from oracle import steps
    fx = {}
    xs, t = tiny_inputs(True)
    w = torch.tensor([1.0, 1.0])
    res = {}
    for tag, fmod, umod in (('ref', ref_f, ref_utils), ('ora', oracle, oracle)):
        torch.manual_seed(2)
        net = fmod.fuseunet(2)
        net.train()
        crit = umod.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, amsgrad=True)
        losses, snaps = [], []
        for _ in range(3):
            _, loss = steps.comparison_step(net, crit, opt, xs[0], xs[1], t)
            losses.append(loss)
            snaps.append({k: p.detach().clone() for k, p in net.named_parameters()})
        res[tag] = (torch.stack(losses), snaps)
    _same(res['ref'][0], res['ora'][0], 'adam losses')
    keys = ['last_conv1.weight', 'last_conv1.bias', 'up_block4.block.conv2.weight',
            'up_block4.block.bn2.weight', 'up_block4.block.bn2.bias',
            'modal1_downblock1.block.conv1.weight']
    for i in (0, 2):
        for k in keys:
            _same(res['ref'][1][i][k], res['ora'][1][i][k], 'adam step%d %s' % (i + 1, k))
            fx['step%d/%s' % (i + 1, k)] = _sub(res['ref'][1][i][k])
    fx['losses'] = _np(res['ref'][0])
    np.savez_compressed(os.path.join(OUT, 'g5_adam.npz'), **fx)
    print('g5 losses', fx['losses'])


# element-wise gradient fixtures at full size: the first conv, a mid-decoder conv, its BatchNorm, the up conv, the head
C2_GRAD_KEYS = ('modal1_downblock1.block.conv1.weight', 'modal2_downblock3.block.conv2.weight',
                'up_block1.bilinear_up.1.weight', 'up_block1.block.conv1.weight',
                'up_block1.block.bn1.weight', 'up_block1.block.bn1.bias', 'up_block4.block.conv2.weight',
                'last_conv1.weight', 'last_conv1.bias')
C4_GRAD_KEYS = ('down_block1.block.conv1.weight', 'down_block3.block.conv2.weight',
                'up_block1.bilinear_up.1.weight', 'up_block1.block.conv1.weight',
                'up_block1.block.bn1.weight', 'up_block1.block.bn1.bias', 'up_block4.block.conv2.weight',
                'last_conv1.weight', 'last_conv1.bias')


def _fp64_truth(ctor, ref_utils, inputs, t, keys):
    """The SAME reference network evaluated in float64 (`.double()`): the ground truth both fp32 implementations round
    differently around.  A ReLU pre-activation that is ~1e-7 from zero lands on either side of it in two fp32
    evaluations; one such flip changes a gradient element by far more than 1e-3 of its value, so the reference's own
    fp32 gradients differ from this truth by up to 2e-3 element-wise at 256x256.  Stored: per-parameter norms and the
    element-wise slices of `keys`, so that a test can bound |ours - truth| by the reference's own |fp32 - truth|."""
    torch.manual_seed(2)
    net = ctor().double()
    net.train()
    w = torch.tensor([1.0, 1.0], dtype=torch.float64)
    out = net(*[x.double() for x in inputs])
    loss = ref_utils.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
    loss.backward()
    fx = {'grad_norms64': _np(torch.stack([p.grad.norm() for p in net.parameters()])),
          'logits_rows64': _np(out[:, :, ::37, :]), 'loss64': _np(loss.detach())}
    named = dict(net.named_parameters())
    for k in keys:
        fx['grad64/' + k] = _sub(named[k].grad)
    return fx


def g2_config(ref_f, ref_utils):
    """BASELINE config 2 digests: FuseUNet N=4, 256x256, synthetic CHAOS-shaped batch."""
    import oracle
    from aide_amd.synthetic import chaos_batch
    fx = {}
    xin, xout, t = chaos_batch(4, 256, seed=1234)
    w = torch.tensor([1.0, 1.0])
    res = {}
    for tag, fmod, umod in (('ref', ref_f, ref_utils), ('ora', oracle, oracle)):
        torch.manual_seed(2)
        net = fmod.fuseunet(2)
        net.train()
        out = net(xin, xout)
        loss = umod.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
        per = umod.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
        loss.backward()
        res[tag] = (out.detach(), loss.detach(), per.detach(),
                    torch.stack([p.grad.double().norm() for p in net.parameters()]),
                    [k for k, _ in net.named_parameters()],
                    {k: p.grad.detach().clone() for k, p in net.named_parameters() if k in C2_GRAD_KEYS})
    for i in range(4):
        _same(res['ref'][i], res['ora'][i], 'g2 #%d' % i)
    for k in C2_GRAD_KEYS:
        _same(res['ref'][5][k], res['ora'][5][k], 'g2 grad ' + k)
    out, loss, per, gn, names, grads = res['ref']
    for k in C2_GRAD_KEYS:                      # element-wise gradients of the real reference (full tensor or _sub stride)
        fx['grad/' + k] = _sub(grads[k])
    fx.update(_fp64_truth(lambda: ref_f.fuseunet(2), ref_utils, (xin, xout), t, C2_GRAD_KEYS))
    fx['logits_sum'] = np.array(out.double().sum().item())
    fx['logits_abs_sum'] = np.array(out.double().abs().sum().item())
    fx['logits_rows'] = _np(out[:, :, ::37, :])          # 7 rows per image/class
    fx['loss'], fx['per_image_loss'] = _np(loss), _np(per)
    fx['grad_norms'], fx['param_names'] = _np(gn), np.array(names)
    fx['seed'] = np.array(1234)
    np.savez_compressed(os.path.join(OUT, 'g2_config2.npz'), **fx)
    print('g2 loss', float(loss), 'per-image', per.tolist())



def g13_config4(ref_u, ref_utils):
    """BASELINE config 4 at its own size: UNet(2), N=4, 3x320x320 (prostate-shaped synthetic, single modality) --
    digests and element-wise gradient slices of the real reference (models_singlemodalinput/UNet.py:152-165)."""
    import oracle
    from aide_amd.synthetic import chaos_batch
    fx = {}
    xin, _, t = chaos_batch(4, 320, seed=1234, single_modal=True)
    w = torch.tensor([1.0, 1.0])
    res = {}
    for tag, umod, lmod in (('ref', ref_u, ref_utils), ('ora', oracle, oracle)):
        torch.manual_seed(2)
        net = umod.UNet(2)
        net.train()
        out = net(xin)
        loss = lmod.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
        per = lmod.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
        loss.backward()
        res[tag] = (out.detach(), loss.detach(), per.detach(),
                    torch.stack([p.grad.double().norm() for p in net.parameters()]),
                    [k for k, _ in net.named_parameters()],
                    {k: p.grad.detach().clone() for k, p in net.named_parameters() if k in C4_GRAD_KEYS})
    for i in range(4):
        _same(res['ref'][i], res['ora'][i], 'g13 #%d' % i)
    out, loss, per, gn, names, grads = res['ref']
    assert set(grads) == set(C4_GRAD_KEYS), sorted(set(C4_GRAD_KEYS) - set(grads))
    for k in C4_GRAD_KEYS:
        _same(grads[k], res['ora'][5][k], 'g13 grad ' + k)
        fx['grad/' + k] = _sub(grads[k])
    fx.update(_fp64_truth(lambda: ref_u.UNet(2), ref_utils, (xin,), t, C4_GRAD_KEYS))
    fx['logits_sum'] = np.array(out.double().sum().item())
    fx['logits_abs_sum'] = np.array(out.double().abs().sum().item())
    fx['logits_rows'] = _np(out[:, :, ::37, :])
    fx['loss'], fx['per_image_loss'] = _np(loss), _np(per)
    fx['grad_norms'], fx['param_names'] = _np(gn), np.array(names)
    fx['seed'] = np.array(1234)
    np.savez_compressed(os.path.join(OUT, 'g13_config4.npz'), **fx)
    print('g13 (C4 UNet 320) loss', float(loss), 'per-image', per.tolist())


def c3_inputs(n=4, size=256):
    """Inputs of the full-size proposed-step fixture (shared by the generator and tests/test_gpu_fullsize.py):
    CHAOS-shaped batch, the labels of the two networks from two different label draws, four intensity-augmented
    copies (identity reverse-aug, as the parity runs of SURVEY 8d prescribe)."""
    from aide_amd.synthetic import chaos_batch
    xin, xout, t1 = chaos_batch(n, size, seed=1234)
    _, _, t2 = chaos_batch(n, size, seed=977)
    augs = [(xin * (1 + 0.05 * (k + 1)), xout * (1 - 0.04 * (k + 1))) for k in range(4)]
    return xin, xout, t1, t2, augs


def g14_config3(ref_f, ref_utils):
    """BASELINE config 3 at its own size: the AIDE proposed co-teaching step (two FuseUNets, 4 augmented train-mode
    forwards + forward + backward + Adam each) at N=4, 256x256, rate 0.25
    (train_files/trainchaos_proposed_30cases1labeled.py:260-325 restated in oracle/steps.py, reference modules / losses
    plugged in)."""
    import oracle
    from oracle import steps
    fx = {}
    xin, xout, t1, t2, augs = c3_inputs()
    w = torch.tensor([1.0, 1.0])
    rate = 0.25
    res = {}
    for tag, fmod, umod in (('ref', ref_f, ref_utils), ('ora', oracle, oracle)):
        torch.manual_seed(2)
        net1 = fmod.fuseunet(2)
        net2 = fmod.fuseunet(2)
        net1.train(), net2.train()
        crit = umod.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)
        corr = umod.MulticlassMSELoss(reduction='none')
        o1 = torch.optim.Adam(net1.parameters(), lr=1e-4, amsgrad=True)
        o2 = torch.optim.Adam(net2.parameters(), lr=1e-4, amsgrad=True)
        r = steps.proposed_step(net1, net2, crit, corr, o1, o2, xin, xout, augs, t1, t2, rate)
        r['g1'] = torch.stack([p.grad.double().norm() for p in net1.parameters()])
        r['g2'] = torch.stack([p.grad.double().norm() for p in net2.parameters()])
        r['nbt'] = net1.modal1_downblock1.block.bn1.num_batches_tracked.clone()
        r['rm'] = net1.up_block4.block.bn2.running_mean.clone()
        res[tag] = r
    for k in ('outputs1', 'outputs2', 'loss1', 'loss2', 'indx1', 'indx2', 'g1', 'g2', 'rm'):
        _same(res['ref'][k], res['ora'][k], 'g14 %s' % k)
    r = res['ref']
    for k in ('loss1', 'loss2', 'indx1', 'indx2', 'loss1_pre', 'loss2_pre', 'g1', 'g2', 'nbt', 'rm'):
        fx[k] = _np(r[k])
    fx['outputs1_rows'] = _np(r['outputs1'][:, :, ::37, :])
    fx['outputs2_rows'] = _np(r['outputs2'][:, :, ::37, :])
    fx['min_gap1'] = np.diff(np.sort(_np(r['loss1_pre']))).min()
    fx['min_gap2'] = np.diff(np.sort(_np(r['loss2_pre']))).min()
    fx['rate'] = np.array(rate)
    np.savez_compressed(os.path.join(OUT, 'g14_config3.npz'), **fx)
    print('g14 (C3 256) loss', float(r['loss1']), float(r['loss2']), r['indx1'].tolist(), r['indx2'].tolist(),
          'per-image', r['loss1_pre'].tolist(), r['loss2_pre'].tolist(), 'gaps', fx['min_gap1'], fx['min_gap2'])


def g15_config5(ref_f, ref_utils):
    """BASELINE config 5 at its own size: FuseUNet N=8, 2 x 3x512x512.  The reference has no bf16 mode, so this fixture
    holds (i) the real reference's fp32 forward (logit rows, loss: the anchor) and (ii) the digests of the bf16-operand
    oracle (oracle/bf16.py: the product's rounding contract applied to the bit-equal restatement) -- logits, loss,
    per-parameter gradient norms."""
    import oracle
    from oracle import bf16 as OB
    from aide_amd.synthetic import chaos_batch
    fx = {}
    xin, xout, t = chaos_batch(8, 512, seed=1234)
    w = torch.tensor([1.0, 1.0])
    torch.manual_seed(2)
    ref = ref_f.fuseunet(2)
    ref.train()
    with torch.no_grad():
        out_ref = ref(xin, xout)
        loss_ref = ref_utils.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)(out_ref, t)
    torch.manual_seed(2)
    ora32 = oracle.fuseunet(2)
    ora32.train()
    with torch.no_grad():
        _same(out_ref, ora32(xin, xout), 'g15 fp32 forward')
    del ora32
    fx['ref_fp32_logits_rows'] = _np(out_ref[:, :, ::73, :])
    fx['ref_fp32_loss'] = _np(loss_ref)
    torch.manual_seed(2)
    net = OB.emulate_bf16(oracle.fuseunet(2))
    net.train()
    out = net(xin, xout)
    loss = oracle.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
    per = oracle.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t)
    loss.backward()
    fx['bf16_logits_rows'] = _np(out[:, :, ::73, :])
    fx['bf16_logits_sum'] = np.array(out.double().sum().item())
    fx['bf16_logits_abs_sum'] = np.array(out.double().abs().sum().item())
    fx['bf16_loss'], fx['bf16_per_image_loss'] = _np(loss), _np(per)
    fx['bf16_grad_norms'] = _np(torch.stack([p.grad.double().norm() for p in net.parameters()]))
    fx['param_names'] = np.array([k for k, _ in net.named_parameters()])
    fx['bf16_vs_fp32_logits'] = np.array(((out - out_ref).abs().max() / out_ref.abs().max()).item())
    fx['seed'] = np.array(1234)
    np.savez_compressed(os.path.join(OUT, 'g15_config5.npz'), **fx)
    print('g15 (C5 512 bs8) fp32 loss', float(loss_ref), 'bf16-oracle loss', float(loss), 'bf16 vs fp32 logits',
          float(fx['bf16_vs_fp32_logits']))


def g16_config5_autocast(ref_f, ref_utils):
    """A second, reference-held anchor for the bf16 mode (BASELINE config 5 at its own size): the REAL reference's modules run
    under PyTorch's own CPU bf16 autocast (convolutions on bf16 operands with fp32 accumulation, bf16-stored activations) --
    not this repo's oracle -- next to the same network in fp32: logit rows, loss, per-parameter gradient norms of both.  The
    GPU test requires the HIP bf16 path to be no further from the reference's fp32 results than the reference under autocast is."""
    from aide_amd.synthetic import chaos_batch
    fx = {}
    xin, xout, t = chaos_batch(8, 512, seed=1234)
    w = torch.tensor([1.0, 1.0])
    crit = ref_utils.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    for tag, ctx in (('fp32', None), ('autocast', torch.autocast('cpu', dtype=torch.bfloat16))):
        torch.manual_seed(2)
        net = ref_f.fuseunet(2)
        net.train()
        if ctx is None:
            out = net(xin, xout)
        else:
            with ctx:
                out = net(xin, xout)
        out32 = out.float()
        loss = crit(out32, t)                      # the loss in fp32 on the network's logits (as the HIP path computes it)
        loss.backward()
        fx[tag + '_logits_rows'] = _np(out32[:, :, ::73, :])
        fx[tag + '_loss'] = _np(loss.detach())
        fx[tag + '_grad_norms'] = _np(torch.stack([p.grad.double().norm() for p in net.parameters()]))
        fx['param_names'] = np.array([k for k, _ in net.named_parameters()])
        print('g16', tag, 'loss', float(loss))
    a, b = torch.from_numpy(fx['autocast_logits_rows']), torch.from_numpy(fx['fp32_logits_rows'])
    fx['autocast_vs_fp32_logits'] = np.array(((a - b).abs().max() / b.abs().max()).item())
    live = fx['fp32_grad_norms'] > 1e-5
    e = np.abs(fx['autocast_grad_norms'][live] - fx['fp32_grad_norms'][live]) / fx['fp32_grad_norms'][live]
    fx['seed'] = np.array(1234)
    np.savez_compressed(os.path.join(OUT, 'g16_config5_autocast.npz'), **fx)
    print('g16 autocast vs fp32: logits %.3e, loss %.3e, gradient norms median %.3e worst %.3e' % (
        float(fx['autocast_vs_fp32_logits']), abs(float(fx['autocast_loss']) - float(fx['fp32_loss'])) / float(fx['fp32_loss']),
        np.median(e), e.max()))


def g6_inference(ref_f, ref_u, ref_utils):
    """Per-case inference (trainchaos_comparison_1case.py:233-273): two training steps move the BN running
    statistics, then eval-mode bs=1 slices -> softmax -> argmax -> [H,W,S] volume and Dice3d_fn."""
    import oracle
    from oracle import steps
    fx = {}
    w = torch.tensor([1.0, 1.0])
    for name, two_modal, mods in (('fuseunet', True, (ref_f, oracle)), ('unet', False, (ref_u, oracle))):
        xs, t = tiny_inputs(two_modal)
        g = torch.Generator().manual_seed(77)
        sl = [torch.randn(6, 3, 48, 32, generator=g) for _ in range(2 if two_modal else 1)]
        tgt = (torch.rand(48, 32, 6, generator=g) > 0.6).numpy().astype(np.int64)
        res = []
        for m in mods:
            torch.manual_seed(2)
            net = m.fuseunet(2) if two_modal else m.UNet(2)
            net.train()
            crit = (ref_utils if m is not oracle else oracle).CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
            opt = torch.optim.Adam(net.parameters(), lr=1e-4, amsgrad=True)
            for _ in range(2):
                steps.comparison_step(net, crit, opt, xs[0], xs[1] if two_modal else None, t)
            net.eval()
            vol, lg = steps.predict_case(net, sl[0], sl[1] if two_modal else None)
            # an untrained net predicts one class everywhere: move the head bias to the median margin so
            # that the label volume is a real mixture (same state for reference and oracle; recorded)
            if not res:
                shift = float((lg[:, 1] - lg[:, 0]).median())
            with torch.no_grad():
                net.last_conv1.bias[1] -= shift
            vol, lg = steps.predict_case(net, sl[0], sl[1] if two_modal else None)
            res.append((vol, lg, steps.Dice3d_fn(vol, tgt)))
        assert np.array_equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), name
        vol, lg, dice = res[0]
        for i, x in enumerate(sl):
            fx['%s/slices%d' % (name, i)] = _np(x)
        fx[name + '/labels'] = vol.astype(np.uint8)                 # [H,W,S]
        fx[name + '/margin'] = _np(lg[:, 1] - lg[:, 0])             # [S,H,W]
        fx[name + '/targets'] = tgt.astype(np.uint8)
        fx[name + '/dice3d'] = np.array(dice)
        fx[name + '/head_bias1_shift'] = np.array(shift, dtype=np.float64)
        print('g6', name, 'foreground fraction %.3f' % vol.mean(), 'dice3d %.6f' % dice,
              'min |margin| %.2e' % float((lg[:, 1] - lg[:, 0]).abs().min()))
    np.savez_compressed(os.path.join(OUT, 'g6_inference.npz'), **fx)


def g7_coteach_ext(ref_utils):
    """a17: KLbidirection, Coteachingloss_dropregionce, Coteachingloss_dropimagedroppixel on the g3 logits."""
    import importlib
    import oracle
    ref_ct = importlib.import_module('utils.coteach_loss')
    g3 = np.load(os.path.join(OUT, 'g3_losses.npz'))
    z1, z2, t = torch.from_numpy(g3['z1']), torch.from_numpy(g3['z2']), torch.from_numpy(g3['targets'])
    fx = {}
    vals = {}
    for tag, fn in (('ref', ref_ct.KLbidirection), ('ora', oracle.KLbidirection)):
        a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        v = fn(a1, a2)
        gw = torch.linspace(0.5, 1.5, v.numel()).view_as(v)
        (v * gw).sum().backward()
        vals[tag] = (v.detach(), a1.grad.clone(), a2.grad.clone())
    for i in range(3):
        _same(vals['ref'][i], vals['ora'][i], 'KLbidirection #%d' % i)
    fx['KL/map'], fx['KL/grad1'], fx['KL/grad2'] = (_np(x) for x in vals['ref'])
    for cname, kw in (('Coteachingloss_dropregionce', dict(scale=0.5, reduction='none')),
                      ('Coteachingloss_dropimagedroppixel', dict(weight=1.0, reduction='none'))):
        for fr in (0.0, 0.25, 0.5):
            vals = {}
            for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                out = []
                for which in (0, 1):                    # the two losses are back-propagated separately
                    a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                    ls = getattr(mod, cname)(**kw)(a1, a2, t, fr)
                    ls[which].backward()
                    out += [ls[which].detach(), a1.grad.clone() if a1.grad is not None else torch.zeros_like(z1),
                            a2.grad.clone() if a2.grad is not None else torch.zeros_like(z2)]
                vals[tag] = out
            for i in range(6):
                _same(vals['ref'][i], vals['ora'][i], '%s fr=%g #%d' % (cname, fr, i))
            key = '%s/fr%g' % (cname, fr)
            r = vals['ref']
            fx[key + '/loss1'], fx[key + '/l1_grad1'], fx[key + '/l1_grad2'] = _np(r[0]), _np(r[1]), _np(r[2])
            fx[key + '/loss2'], fx[key + '/l2_grad1'], fx[key + '/l2_grad2'] = _np(r[3]), _np(r[4]), _np(r[5])
            print('g7', key, float(r[0]), float(r[3]), 'cross-grad norms', float(r[2].norm()), float(r[4].norm()))
    np.savez_compressed(os.path.join(OUT, 'g7_coteach_ext.npz'), **fx)


def g8_pixelcoreg(ref_utils):
    """Pixelcoreg_Focalloss / _twomodel (utils/reg_loss.py:58-193) on the g3 logits."""
    import oracle
    g3 = np.load(os.path.join(OUT, 'g3_losses.npz'))
    z1, z2, t = torch.from_numpy(g3['z1']), torch.from_numpy(g3['z2']), torch.from_numpy(g3['targets'])
    z3 = torch.randn(z1.shape, generator=torch.Generator().manual_seed(99)) * 2.0
    fx = {'z3': _np(z3)}
    for cname, three in (('Pixelcoreg_Focalloss', True), ('Pixelcoreg_Focalloss_twomodel', False)):
        for fr, kd, red in ((0.0, 0.3, 'mean'), (0.25, 0.3, 'mean'), (0.5, 0.7, 'sum')):
            vals = {}
            for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                a = [z.clone().requires_grad_(True) for z in ((z1, z2, z3) if three else (z1, z2))]
                loss, frac = getattr(mod, cname)(reduction=red)(*a, t, fr, kd, torch.device('cpu'))
                loss.backward()
                vals[tag] = [loss.detach(), torch.as_tensor(frac).detach().float()] + \
                    [x.grad.clone() if x.grad is not None else torch.zeros_like(z1) for x in a]
            for i in range(len(vals['ref'])):
                _same(vals['ref'][i], vals['ora'][i], '%s fr=%g #%d' % (cname, fr, i))
            key = '%s/fr%g_kd%g_%s' % (cname, fr, kd, red)
            fx[key + '/loss'], fx[key + '/frac'] = _np(vals['ref'][0]), _np(vals['ref'][1])
            for i, g in enumerate(vals['ref'][2:]):
                fx[key + '/grad%d' % (i + 1)] = _np(g)
            print('g8', key, float(vals['ref'][0]), float(vals['ref'][1]))
    np.savez_compressed(os.path.join(OUT, 'g8_pixelcoreg.npz'), **fx)


def g10_variants(ref_f, ref_u, ref_utils, oracle):
    g1_model('fuseunetsaseparate', ref_f.fuseunetsaseparate, oracle.fuseunetsaseparate, {}, True, ref_utils)
    for w in (128, 32, 16, 2):
        g1_model('unet%d' % w, getattr(ref_u, 'UNet%d' % w), getattr(oracle, 'UNet%d' % w), {}, False, ref_utils)


def g11_polylr(ref_utils):
    """utils/poly_lr_scheduler.py:27-47: the learning rates of 25 epochs for two (lr, max_epoch, power) settings."""
    fx = {}
    for tag, (lr, max_epoch, power) in (('a', (1e-4, 20, 0.9)), ('b', (3e-3, 7, 2.0))):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=lr)
        sch = ref_utils.PolyLR(opt, max_epoch=max_epoch, power=power)
        seq = [opt.param_groups[0]['lr']]
        for _ in range(24):
            sch.step()
            seq.append(opt.param_groups[0]['lr'])
        fx['cfg_' + tag] = np.array([lr, max_epoch, power])
        fx['lr_' + tag] = np.array(seq, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'g11_polylr.npz'), **fx)
    print('g11', fx['lr_a'][:3], fx['lr_b'][:3])


def g12_metrics(ref_utils):
    """metrics2d.py:8-84 (Dice_fn, Dice_fn_Nozero, TP_TN_FP_FN, IoU_fn) and loss2d.py:63-85,156-171 (Dice_Loss,
    CEDiceLoss) on a batch with the edge cases: an image empty in target and prediction, one empty in the target only."""
    import oracle
    from aide_amd.synthetic import chaos_batch
    g = torch.Generator().manual_seed(91)
    n, s = 5, 48
    _, _, t = chaos_batch(n, s, seed=9)
    t[0, 8:30, 10:40] = 1
    t[1] = 0
    t[2] = 0
    t[4, 5:20, 5:20] = 1
    z = torch.randn(n, 2, s, s, generator=g) * 2.0
    z[1, 0] += 30.0                              # image 1: empty target, empty prediction
    z[2, 1, 4:9, 4:9] += 30.0                    # image 2: empty target, non-empty prediction
    fx = dict(z=_np(z), targets=_np(t))
    for batch, tag in ((slice(0, n), 'all'), (slice(0, 1), 'first'), (slice(3, 5), 'nonempty')):
        zz, tt = z[batch], t[batch]
        vals = {}
        for key, mod in (('ref', ref_utils), ('ora', oracle)):
            d = mod.Dice_fn(zz.clone(), tt)
            dn = mod.Dice_fn_Nozero(zz.clone(), tt)
            c = mod.TP_TN_FP_FN(zz.clone(), tt)
            vals[key] = [torch.as_tensor(d).float(), torch.tensor([dn[0], float(dn[1])]),
                         torch.stack([torch.as_tensor(v).float() for v in c])]
            if tag == 'nonempty':                # IoU of an image empty in both is 0/0
                vals[key].append(torch.as_tensor(mod.IoU_fn(zz.clone(), tt)).float())
        for a, b in zip(vals['ref'], vals['ora']):
            _same(a, b, 'metrics ' + tag)
        fx[tag + '/Dice_fn'], fx[tag + '/Dice_fn_Nozero'], fx[tag + '/TP_TN_FP_FN'] = [_np(v) for v in vals['ref'][:3]]
        if tag == 'nonempty':
            fx[tag + '/IoU_fn'] = _np(vals['ref'][3])
    assert torch.isnan(torch.as_tensor(ref_utils.IoU_fn(z.clone(), t)))          # the reference's 0/0
    cw, cdw = torch.tensor([1.0, 3.0]), torch.tensor([0.7, 1.6])
    for lname, kw in (('Dice_Loss', dict(smooth=1.0, reduction='mean')), ('Dice_Loss', dict(smooth=0.5, reduction='sum')),
                      ('Dice_Loss', dict(reduction='none')), ('CEDiceLoss', dict(cediceweight=cdw, classweight=cw)),
                      ('CEDiceLoss', dict(reduction='sum')), ('CEDiceLoss', dict())):
        vals = {}
        for key, mod in (('ref', ref_utils), ('ora', oracle)):
            zz = z.clone().requires_grad_(True)
            v = getattr(mod, lname)(**kw)(zz, t)
            (v.sum() if v.dim() else v).backward()
            vals[key] = (v.detach(), zz.grad.clone())
        _same(vals['ref'][0], vals['ora'][0], lname)
        _same(vals['ref'][1], vals['ora'][1], lname + ' grad')
        key = '%s/%s' % (lname, '_'.join('%s=%s' % (k, 'w' if torch.is_tensor(v) else v) for k, v in sorted(kw.items())))
        fx[key], fx[key + '/grad'] = _np(vals['ref'][0]), _np(vals['ref'][1])
    # branches no shipped script reaches: one-hot targets in CrossEntropyLoss2d (loss2d.py:11-12) and MulticlassDiceLoss
    # (:98-104, class weights), DiceLoss on a probability input (:47-48)
    onehot = torch.nn.functional.one_hot(t, 2).permute(0, 3, 1, 2).float().contiguous()
    prob = torch.rand(n, s, s, generator=g)
    fx['onehot'], fx['prob'] = _np(onehot), _np(prob)
    mw = torch.tensor([0.3, 1.7])
    cases = [('CrossEntropyLoss2d', dict(weight=cw), 'z', 'onehot'), ('CrossEntropyLoss2d', dict(reduction='sum'), 'z', 'onehot'),
             ('MulticlassDiceLoss', dict(weight=mw), 'z', 'onehot'), ('MulticlassDiceLoss', dict(reduction='none'), 'z', 'onehot'),
             ('MulticlassDiceLoss', dict(weight=mw, smooth=0.5, reduction='sum'), 'z', 'onehot'),
             ('DiceLoss', dict(), 'prob', 't'), ('DiceLoss', dict(smooth=0.25, reduction='none'), 'prob', 't'),
             ('DiceLoss', dict(reduction='sum'), 'prob', 'onehot1')]
    srcs = dict(z=z, prob=prob)
    tgts = dict(onehot=onehot, t=t, onehot1=onehot[:, 1].contiguous())
    for lname, kw, xin, tin in cases:
        vals = {}
        for key, mod in (('ref', ref_utils), ('ora', oracle)):
            xx = srcs[xin].clone().requires_grad_(True)
            v = getattr(mod, lname)(**kw)(xx, tgts[tin])
            # weighted upstream gradient so that the per-image ('none') backward is exercised with distinct factors
            (v * torch.arange(1, v.numel() + 1).float()).sum().backward() if v.dim() else v.backward()
            vals[key] = (v.detach(), xx.grad.clone())
        _same(vals['ref'][0], vals['ora'][0], 'branch ' + lname)
        _same(vals['ref'][1], vals['ora'][1], 'branch grad ' + lname)
        key = 'branch/%s/%s/%s' % (lname, tin, '_'.join('%s=%s' % (k, 'w' if torch.is_tensor(v) else v) for k, v in sorted(kw.items())))
        fx[key], fx[key + '/grad'] = _np(vals['ref'][0]), _np(vals['ref'][1])
    np.savez_compressed(os.path.join(OUT, 'g12_metrics.npz'), **fx)
    print('g12', [k for k in fx if '/' in k and not k.endswith('grad')])


def g17_multiclass(ref_f, ref_u, ref_utils):
    """num_classes > 2: the reference's modules on 3 / 4 / 5 / 8 classes -- tiny networks (logits,
    loss, per-image loss, gradients, eval logits) and every loss form that takes index or one-hot targets, with class
    weights, an ignored pixel, the consistency term and the two image-level co-teaching operators.  The oracle
    restatement is asserted bit-equal to the reference before anything is written."""
    import oracle
    fx = {}
    # ---- networks
    for name, rc, oc, two_modal, C in (('fuseunet3', ref_f.fuseunet, oracle.fuseunet, True, 3),
                                       ('unet4', ref_u.UNet, oracle.UNet, False, 4)):
        torch.manual_seed(2)
        rnet = rc(C)
        torch.manual_seed(2)
        onet = oc(C)
        for (k, a), (_, b) in zip(rnet.state_dict().items(), onet.state_dict().items()):
            _same(a, b, name + ' init ' + k)
        g = torch.Generator().manual_seed(4321)
        xs = [torch.randn(2, 3, 32, 32, generator=g) for _ in range(2 if two_modal else 1)]
        t = torch.randint(0, C, (2, 32, 32), generator=g)
        cw = torch.tensor([1.0, 2.0, 0.5, 1.5][:C])
        cdw = torch.tensor([0.8, 1.3])
        res = {}
        for tag, net, mod in (('ref', rnet, ref_utils), ('ora', onet, oracle)):
            net.train()
            out = net(*xs)
            loss = mod.CEMDiceLoss(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw)(out, t)
            per_img = mod.CEMDiceLossImage(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw)(out, t)
            net.zero_grad()
            loss.backward()
            grads = {k: q.grad.clone() for k, q in net.named_parameters()}
            net.eval()
            with torch.no_grad():
                ev = net(*xs)
            res[tag] = dict(out=out.detach(), loss=loss.detach(), per_img=per_img.detach(), grads=grads, ev=ev)
        r, o = res['ref'], res['ora']
        for k in ('out', 'loss', 'per_img', 'ev'):
            _same(r[k], o[k], name + ' ' + k)
        for k in r['grads']:
            _same(r['grads'][k], o['grads'][k], name + ' grad ' + k)
        names = list(r['grads'].keys())
        for i, x in enumerate(xs):
            fx['%s/x%d' % (name, i)] = _np(x)
        fx[name + '/targets'] = _np(t)
        fx[name + '/class_w'] = _np(cw)
        fx[name + '/cedice_w'] = _np(cdw)
        fx[name + '/logits'] = _np(r['out'])
        fx[name + '/loss'] = _np(r['loss'])
        fx[name + '/per_image_loss'] = _np(r['per_img'])
        fx[name + '/eval_logits'] = _np(r['ev'])
        fx[name + '/labels'] = _np(torch.argmax(torch.softmax(r['ev'], dim=1), dim=1))
        fx[name + '/param_names'] = np.array(names)
        fx[name + '/grad_norms'] = np.array([r['grads'][k].double().norm().item() for k in names])
        for k in ('last_conv1.weight', 'last_conv1.bias', names[0]):
            fx['%s/grad/%s' % (name, k)] = _np(r['grads'][k])
        print('g17', name, 'loss', float(r['loss']))
    # ---- loss forms
    for C in (3, 5, 8):
        g = torch.Generator().manual_seed(100 + C)
        n, s = 4, 24
        t = torch.randint(0, C, (n, s, s), generator=g)
        z1 = torch.randn(n, C, s, s, generator=g) * 2.0
        z2 = torch.randn(n, C, s, s, generator=g) * 2.0
        onehot = torch.nn.functional.one_hot(t, C).permute(0, 3, 1, 2).float()
        # separate the per-image losses (well-defined argsort): push image i towards / away from its target
        z1 = z1 + torch.tensor([0.0, 0.7, -0.6, 1.2]).view(n, 1, 1, 1) * (2 * onehot - 1)
        z2 = z2 + torch.tensor([1.0, -0.5, 0.4, -1.1]).view(n, 1, 1, 1) * (2 * onehot - 1)
        t_ign = t.clone()
        t_ign[0, 3:9, 5:17] = 255
        pseudo = torch.softmax(torch.randn(n, C, s, s, generator=g), dim=1)
        wmap = (1.0 - 4.0 * pseudo[:, 0] * pseudo[:, 1]).unsqueeze(1)
        cw = torch.tensor([1.0, 3.0, 0.5, 2.0, 1.5, 0.25, 1.0, 4.0][:C])
        cdw = torch.tensor([0.7, 1.6])
        pre = 'c%d/' % C
        fx.update({pre + 'z1': _np(z1), pre + 'z2': _np(z2), pre + 'targets': _np(t), pre + 'targets_ignore': _np(t_ign),
                   pre + 'pseudo': _np(pseudo), pre + 'wmap': _np(wmap), pre + 'class_w': _np(cw), pre + 'cedice_w': _np(cdw)})
        cases = (('CrossEntropyLoss2d', dict(weight=cw), t_ign),
                 ('CrossEntropyLoss2d_sum', dict(weight=cw, reduction='sum'), t_ign),
                 ('CrossEntropyLoss2d_none', dict(weight=cw, reduction='none'), t_ign),
                 ('CrossEntropyLoss2d_onehot', dict(weight=cw), onehot),
                 ('DiceLoss', dict(), t), ('DiceLoss_none', dict(reduction='none'), t), ('Dice_Loss_sum', dict(reduction='sum'), t),
                 ('MulticlassDiceLoss', dict(weight=cw), t),
                 ('MulticlassDiceLoss_onehot', dict(weight=cw), onehot),
                 ('MulticlassDiceLoss_onehot_none', dict(weight=cw, reduction='none'), onehot),
                 ('CEMDiceLoss', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw), t),
                 ('CEMDiceLoss_sum', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw, reduction='sum'), t),
                 ('CEMDiceLossImage', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw), t),
                 ('CEDiceLoss', dict(cediceweight=cdw, classweight=cw), t))
        for key, kw, tgt in cases:
            lname = key.split('_')[0] if not key.startswith('Dice_Loss') else 'Dice_Loss'
            vals = {}
            for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                zz = z1.clone().requires_grad_(True)
                v = getattr(mod, lname)(**kw)(zz, tgt)
                gsel = torch.linspace(0.5, 1.5, v.numel()).view(v.shape) if v.dim() else None
                ((v * gsel).sum() if v.dim() else v).backward()
                vals[tag] = (v.detach(), zz.grad.clone())
            _same(vals['ref'][0], vals['ora'][0], pre + key)
            _same(vals['ref'][1], vals['ora'][1], pre + key + ' grad')
            fx[pre + key] = _np(vals['ref'][0])
            fx[pre + key + '/grad'] = _np(vals['ref'][1])
        vals = {}
        for tag, mod in (('ref', ref_utils), ('ora', oracle)):
            zz = z1.clone().requires_grad_(True)
            v = (wmap * mod.MulticlassMSELoss(reduction='none')(zz, pseudo)).mean()
            v.backward()
            vals[tag] = (v.detach(), zz.grad.clone())
        _same(vals['ref'][0], vals['ora'][0], pre + 'mse')
        _same(vals['ref'][1], vals['ora'][1], pre + 'mse grad')
        fx[pre + 'mse_wm_mean'], fx[pre + 'mse_wm_mean/grad'] = _np(vals['ref'][0]), _np(vals['ref'][1])
        for cname in ('Coteachingloss_dropimage', 'Coteachingloss_weightimage'):
            for fr in (0.25, 0.5):
                vals = {}
                for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                    a1 = z1.clone().requires_grad_(True)
                    a2 = z2.clone().requires_grad_(True)
                    l1, l2 = getattr(mod, cname)(weight=1.0, reduction='none')(a1, a2, t, fr)
                    (l1 + l2).backward()
                    vals[tag] = (l1.detach(), l2.detach(), a1.grad.clone(), a2.grad.clone())
                for i in range(4):
                    _same(vals['ref'][i], vals['ora'][i], '%s%s fr=%g #%d' % (pre, cname, fr, i))
                key = '%s%s/fr%g' % (pre, cname, fr)
                fx[key + '/loss1'], fx[key + '/loss2'] = _np(vals['ref'][0]), _np(vals['ref'][1])
                fx[key + '/grad1'], fx[key + '/grad2'] = _np(vals['ref'][2]), _np(vals['ref'][3])
        # the composite loss of the proposed loop (trainchaos_proposed_30cases1labeled.py:303-321) on C classes, keep = 2
        crit = ref_utils.CEMDiceLossImage(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw)
        corr = ref_utils.MulticlassMSELoss(reduction='none')
        rate, segcor = 0.3, (1.0, 10.0)
        a1 = z1.clone().requires_grad_(True)
        a2 = z2.clone().requires_grad_(True)
        pre1, pre2 = crit(a1, t), crit(a2, t)
        _, indx1 = pre1.sort()
        _, indx2 = pre2.sort()
        l1 = segcor[0] * (crit(a1[indx2[0:2]], t[indx2[0:2]]).mean() + (1.0 - rate) * crit(a1[indx2[2:]], t[indx2[2:]]).mean()) + \
            segcor[1] * rate * (wmap[indx2[2:]] * corr(a1[indx2[2:]], pseudo[indx2[2:]])).mean()
        l2 = segcor[0] * (crit(a2[indx1[0:2]], t[indx1[0:2]]).mean() + (1.0 - rate) * crit(a2[indx1[2:]], t[indx1[2:]]).mean()) + \
            segcor[1] * rate * (wmap[indx1[2:]] * corr(a2[indx1[2:]], pseudo[indx1[2:]])).mean()
        (l1 + l2).backward()
        fx[pre + 'proposed/loss1'], fx[pre + 'proposed/loss2'] = _np(l1.detach()), _np(l2.detach())
        fx[pre + 'proposed/grad1'], fx[pre + 'proposed/grad2'] = _np(a1.grad), _np(a2.grad)
        fx[pre + 'proposed/indx1'], fx[pre + 'proposed/indx2'] = _np(indx1), _np(indx2)
        fx[pre + 'proposed/min_gap'] = min(np.diff(np.sort(_np(pre1.detach()))).min(), np.diff(np.sort(_np(pre2.detach()))).min())
        # pseudo-label ensemble (:274-288): mean softmax of 4 passes, sharpen, weight map
        passes = [torch.randn(n, C, s, s, generator=g) * 1.5 for _ in range(4)]
        pl = sum(torch.softmax(q, dim=1) for q in passes) / 4.0
        tmp = torch.pow(pl, 2.0)
        pl = tmp / tmp.sum(dim=1).unsqueeze(dim=1)
        fx[pre + 'ensemble/passes'] = _np(torch.stack(passes))
        fx[pre + 'ensemble/pseudo'] = _np(pl)
        fx[pre + 'ensemble/wmap'] = _np((1.0 - 4.0 * pl[:, 0] * pl[:, 1]).unsqueeze(1))
        # the generic rank-4 co-teaching operators (utils/coteach_loss.py:85-92, 163-254) on C classes: KL map with both
        # gradients; region CE / drop-pixel losses, each back-propagated separately, gradients w.r.t. BOTH logit tensors
        if C <= 5:
            import importlib
            ref_ct = importlib.import_module('utils.coteach_loss')
            vals = {}
            for tag, fn in (('ref', ref_ct.KLbidirection), ('ora', oracle.KLbidirection)):
                a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                v = fn(a1, a2)
                (v * torch.linspace(0.5, 1.5, v.numel()).view_as(v)).sum().backward()
                vals[tag] = (v.detach(), a1.grad.clone(), a2.grad.clone())
            for i in range(3):
                _same(vals['ref'][i], vals['ora'][i], pre + 'KLbidirection #%d' % i)
            fx[pre + 'KL/map'], fx[pre + 'KL/grad1'], fx[pre + 'KL/grad2'] = (_np(x) for x in vals['ref'])
            for cname, kw in (('Coteachingloss_dropregionce', dict(scale=0.5, reduction='none')),
                              ('Coteachingloss_dropimagedroppixel', dict(weight=1.0, reduction='none'))):
                for fr in (0.25, 0.5):
                    vals = {}
                    for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                        out = []
                        for which in (0, 1):
                            a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                            ls = getattr(mod, cname)(**kw)(a1, a2, t, fr)
                            ls[which].backward()
                            out += [ls[which].detach(), a1.grad.clone() if a1.grad is not None else torch.zeros_like(z1),
                                    a2.grad.clone() if a2.grad is not None else torch.zeros_like(z2)]
                        vals[tag] = out
                    for i in range(6):
                        _same(vals['ref'][i], vals['ora'][i], '%s%s fr=%g #%d' % (pre, cname, fr, i))
                    key = '%s%s/fr%g' % (pre, cname, fr)
                    r = vals['ref']
                    fx[key + '/loss1'], fx[key + '/l1_grad1'], fx[key + '/l1_grad2'] = _np(r[0]), _np(r[1]), _np(r[2])
                    fx[key + '/loss2'], fx[key + '/l2_grad1'], fx[key + '/l2_grad2'] = _np(r[3]), _np(r[4]), _np(r[5])
                    print('g17', key, float(r[0]), float(r[3]))
        d_ref = ref_utils.Dice_fn(z1.clone(), t)
        _same(torch.as_tensor(d_ref), torch.as_tensor(oracle.Dice_fn(z1.clone(), t)), pre + 'Dice_fn')
        fx[pre + 'Dice_fn'] = _np(torch.as_tensor(d_ref))
        fx[pre + 'labels'] = _np(torch.argmax(torch.softmax(z1, dim=1), dim=1))
    np.savez_compressed(os.path.join(OUT, 'g17_multiclass.npz'), **fx)
    print('g17 multiclass ok')


def g18_dropregionce_scale(ref_utils):
    """Coteachingloss_dropregionce for pooling windows other than the default 2 x 2 (utils/coteach_loss.py:163-196: window =
    stride = int(H / int(H * scale)), ceil_mode): scale 0.25 (4 x 4 windows) and 0.3 (3 x 3, the last row / column of windows
    clipped) on the two-class g3 logits, and a three-class case on a ragged plane.  Both losses, back-propagated
    separately, with their gradients w.r.t. the own logits (the selection is not differentiated)."""
    import oracle
    g3 = np.load(os.path.join(OUT, 'g3_losses.npz'))
    fx = {}
    g = torch.Generator().manual_seed(77)
    cases = {
        'c2': (torch.from_numpy(g3['z1']), torch.from_numpy(g3['z2']), torch.from_numpy(g3['targets'])),      # [4, 2, 64, 64]
        'c3': (torch.randn(2, 3, 36, 44, generator=g), torch.randn(2, 3, 36, 44, generator=g),
               torch.randint(0, 3, (2, 36, 44), generator=g)),
    }
    for cname, (z1, z2, t) in cases.items():
        if cname != 'c2':                      # (c2 = the inputs of g3_losses.npz)
            fx[cname + '/z1'], fx[cname + '/z2'], fx[cname + '/targets'] = _np(z1), _np(z2), _np(t)
        for scale in (0.25, 0.3):
            for fr in (0.25, 0.5):
                vals = {}
                for tag, mod in (('ref', ref_utils), ('ora', oracle)):
                    out = []
                    for which in (0, 1):
                        a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                        ls = mod.Coteachingloss_dropregionce(scale=scale, reduction='none')(a1, a2, t, fr)
                        ls[which].backward()
                        own = a1 if which == 0 else a2
                        out += [ls[which].detach(), own.grad.clone()]
                    vals[tag] = out
                for i in range(4):
                    _same(vals['ref'][i], vals['ora'][i], 'dropregionce %s scale=%g fr=%g #%d' % (cname, scale, fr, i))
                key = '%s/s%g/fr%g' % (cname, scale, fr)
                r = vals['ref']
                fx[key + '/loss1'], fx[key + '/grad1'], fx[key + '/loss2'], fx[key + '/grad2'] = (_np(x) for x in r)
                print('g18', key, float(r[0]), float(r[2]), 'nonzero grads', int((r[1] != 0).sum()), int((r[3] != 0).sum()))
    np.savez_compressed(os.path.join(OUT, 'g18_dropregionce_scale.npz'), **fx)


def _ref_functions(path, names, glb):
    """The named top-level functions of a reference script that cannot be imported (the train scripts import skimage / pydicom
    and run argparse / mkdir at module top): their definitions are taken out of the script's syntax tree and compiled
    IN MEMORY from the reference's own text -- nothing of it is written anywhere."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), (path, [n.name for n in body])
    ns = dict(glb)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'), ns)
    return [ns[n] for n in names]


def g19_reverseaug():
    """Pins the LOOP-LEVEL restatements to the reference's own function text: `reverseaug` and `sharpen` of
    train_files/trainchaos_proposed_30cases1labeled.py:81-101 and the p^(1/T) `sharpen` of
    train_files/trainkidney_proposed_mask1.py:113-117 are executed here (see _ref_functions) on non-identity flips / rotations
    (PIL's 0 / 90 / 180 / 270 fast paths, square and non-square planes, augno < 4) and oracle.steps.reverseaug /
    oracle.losses.sharpen / sharpen_root must reproduce them bit for bit."""
    import warnings
    from PIL import Image
    from oracle import steps, losses
    glb = dict(torch=torch, np=np, Image=Image)
    chaos = os.path.join(REF, 'train_files', 'trainchaos_proposed_30cases1labeled.py')
    kidney = os.path.join(REF, 'train_files', 'trainkidney_proposed_mask1.py')
    ref_rev, ref_sharpen = _ref_functions(chaos, ['reverseaug', 'sharpen'], glb)
    kid_rev, kid_sharpen = _ref_functions(kidney, ['reverseaugbatch', 'sharpen'], glb)     # (:97-111: the batched form this step calls, :274-275)
    out = {}
    g = torch.Generator().manual_seed(21)
    augsets = [
        {'augno': [4, 4, 4, 3],
         'hflip1': [0, 1, 0, 1], 'degree1': [0.0, 37.5, -60.0, 12.25],
         'hflip2': [1, 0, 1, 0], 'degree2': [90.0, 180.0, 270.0, -90.0],
         'hflip3': [0, 0, 1, 1], 'degree3': [59.99, -0.5, 360.0, 45.0],
         'hflip4': [1, 1, 0, 0], 'degree4': [-33.0, 5.0, 120.0, 77.0]},
        {'augno': [2, 4, 1, 4],
         'hflip1': [1, 1, 0, 0], 'degree1': [-90.0, 60.0, 180.0, -17.0],
         'hflip2': [0, 1, 1, 0], 'degree2': [23.0, -180.0, 3.0, 270.0],
         'hflip3': [1, 0, 0, 1], 'degree3': [1.0, 90.0, 2.0, -45.0],
         'hflip4': [0, 0, 1, 1], 'degree4': [9.0, 0.0, 4.0, 33.3]}]
    case = 0
    for (h, w) in ((16, 16), (12, 20), (20, 12)):
        for augset in augsets:
            outs = [torch.randn(4, 2, h, w, generator=g) for _ in range(4)]
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                ref = ref_rev(augset, [o.clone() for o in outs], 2)
                kid = kid_rev(augset, [o.clone() for o in outs], 2)
                ora = steps.reverseaug(augset, [o.clone() for o in outs], 2)
            for k in range(4):
                _same(ora[k], ref[k], 'reverseaug case %d pass %d' % (case, k))
                _same(kid[k], ref[k], 'kidney reverseaug case %d pass %d' % (case, k))
                out['rev%d/in%d' % (case, k)] = _np(outs[k])
                out['rev%d/out%d' % (case, k)] = _np(ref[k])
            for key, v in augset.items():
                out['rev%d/%s' % (case, key)] = np.asarray(v, dtype=np.float64)
            case += 1
    out['rev_cases'] = np.asarray(case)
    p = torch.softmax(torch.randn(4, 2, 16, 16, generator=g) * 2.0, dim=1)
    out['sharpen/p'] = _np(p)
    for T in (0.5, 1.0, 2.0):
        a, b = ref_sharpen(p.clone(), T), kid_sharpen(p.clone(), T)
        _same(losses.sharpen(p.clone(), T), a, 'sharpen p^T, T = %g' % T)
        _same(losses.sharpen_root(p.clone(), T), b, 'sharpen p^(1/T), T = %g' % T)
        out['sharpen/pow_T_%g' % T] = _np(a)
        out['sharpen/pow_invT_%g' % T] = _np(b)
    np.savez_compressed(os.path.join(OUT, 'g19_reverseaug.npz'), **out)
    print('g19_reverseaug.npz: %d reverseaug cases and both sharpen flavours reproduced bit for bit by the oracle' % case)


def _ref_train_loop_body(path):
    """The statements of the TRAINING loop body of a reference `*_proposed_*` script -- the first `for batch_idx, (...) in
    tqdm(enumerate(train_loader), ...)` inside its `for epoch in ...` loop -- compiled in memory from the script's own syntax
    tree, to be exec'd in a namespace that supplies what the surrounding function would (nets, criteria, optimizers, a batch)."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    epoch_loop = next(n for n in ast.walk(tree) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == 'epoch')
    inner = next(n for n in epoch_loop.body if isinstance(n, ast.For) and 'train_loader' in ast.dump(n.iter))
    first, last = inner.body[0].lineno, inner.body[-1].end_lineno
    return compile(ast.Module(body=inner.body, type_ignores=[]), path, 'exec'), (first, last)


def g20_proposed_variants(ref_f, ref_u, ref_utils):
    """The co-teaching step in the three forms of the reference's `*_proposed_*` scripts, each produced by EXECUTING the
    script's own training-loop body (see _ref_train_loop_body) with the imported reference modules, non-identity flips /
    rotations through its own reverseaug / reverseaugbatch and its own sharpen:
      chaos   trainchaos_proposed_30cases1labeled.py:262-330   fuseunet, bs 4, train-mode passes, p^T, keep 2
      kidney  trainkidney_proposed_mask1.py:266-338            UNet, bs 4, eval-mode passes, p^(1/T), keep 2
      breast  trainbreast_dataset3_proposed_272cases25labeled.py:258-336   UNet, bs 8, eval-mode passes, p^(1/T), keep 4
      prostate trainprostate_proposed_isbi3ttransferisbidx.py:253-327     UNet, bs 4, train-mode passes, p^T, keep 2
    and oracle.steps.proposed_step (oracle nets and losses, same seeds) must reproduce loss / indices / gradients bit for bit."""
    import types
    import warnings
    import oracle
    from PIL import Image
    from oracle import steps, losses
    tf = os.path.join(REF, 'train_files')
    cases = [('chaos', os.path.join(tf, 'trainchaos_proposed_30cases1labeled.py'), 'reverseaug', True, 4, 0.5, 2, False, losses.sharpen),
             ('kidney', os.path.join(tf, 'trainkidney_proposed_mask1.py'), 'reverseaugbatch', False, 4, 0.5, 2, True, losses.sharpen_root),
             ('breast', os.path.join(tf, 'trainbreast_dataset3_proposed_272cases25labeled.py'), 'reverseaugbatch', False, 8, 2.0, 4, True,
              losses.sharpen_root),
             # round 6: the prostate scripts are a fourth form -- UNet like kidney / breast, but NO eval() around the
             # augmentation passes (they update the BatchNorm statistics) and p^T like chaos
             ('prostate', os.path.join(tf, 'trainprostate_proposed_isbi3ttransferisbidx.py'), 'reverseaug', False, 4, 0.5, 2, False,
              losses.sharpen)]
    fx = {}
    s = 32
    w = torch.tensor([1.0, 1.0])
    for name, path, revname, two_modal, n, temp, keep, eval_aug, sharpen_fn in cases:
        body, lines = _ref_train_loop_body(path)
        rev, shp = _ref_functions(path, [revname, 'sharpen'], dict(torch=torch, np=np, Image=Image))
        g = torch.Generator().manual_seed(977 + n)
        nin = 2 if two_modal else 1
        xs = [torch.randn(n, 3, s, s, generator=g) for _ in range(nin)]
        t1 = (torch.rand(n, s, s, generator=g) > 0.7).long()
        t2 = (torch.rand(n, s, s, generator=g) > 0.65).long()
        augs = [[x + 0.05 * torch.randn(n, 3, s, s, generator=g) for x in xs] for _ in range(4)]
        augset = {'augno': [4] * n}
        for k in range(4):
            augset['hflip%d' % (k + 1)] = [int(v) for v in torch.randint(0, 2, (n,), generator=g)]
            deg = (torch.rand(n, generator=g) * 120.0 - 60.0).tolist()
            if k == 1:
                deg[0], deg[1] = 90.0, 180.0                     # PIL's exact fast paths, too
            augset['degree%d' % (k + 1)] = [float(d) for d in deg]
        rate = 0.36                                               # (epoch 12 of a 20-epoch warm-up)
        key = name + '/'
        for i, x in enumerate(xs):
            fx[key + 'x%d' % i] = _np(x)
        fx[key + 't1'], fx[key + 't2'] = _np(t1), _np(t2)
        for k in range(4):
            for i in range(nin):
                fx[key + 'aug%d_%d' % (k, i)] = _np(augs[k][i])
            fx[key + 'hflip%d' % (k + 1)] = np.asarray(augset['hflip%d' % (k + 1)], dtype=np.float64)
            fx[key + 'degree%d' % (k + 1)] = np.asarray(augset['degree%d' % (k + 1)], dtype=np.float64)
        fx[key + 'meta'] = np.asarray([n, temp, keep, int(eval_aug), rate, nin], dtype=np.float64)

        def nets(fmod, umod):
            torch.manual_seed(2)
            ctor = fmod.fuseunet if two_modal else umod.UNet
            a, b = ctor(2), ctor(2)
            a.train(), b.train()
            return a, b

        # ---- the reference: its own loop body, its own modules ----
        net1, net2 = nets(ref_f, ref_u)
        ns = dict(torch=torch, np=np, Image=Image, device=torch.device('cpu'), num_classes=2, net1=net1, net2=net2,
                  criterion=ref_utils.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w),
                  corrlosscriterion=ref_utils.MulticlassMSELoss(reduction='none'),
                  optimizer1=torch.optim.Adam(net1.parameters(), lr=1e-4, amsgrad=True),
                  optimizer2=torch.optim.Adam(net2.parameters(), lr=1e-4, amsgrad=True),
                  args=types.SimpleNamespace(batch_size=n, temperature=temp, segcor_weight=[1.0, 10.0]),
                  rate_schedule=np.full(20, rate), epoch=12, Dice_fn=ref_utils.Dice_fn, sharpen=shp,
                  train_count=0, train_loss1=0., train_loss2=0., train_dice1=0., train_dice2=0., batch_idx=0)
        ns[revname] = rev
        aug_ns = dict((kk, list(v)) for kk, v in augset.items())
        for k in range(4):
            if two_modal:
                aug_ns['imgmodal1%d' % (k + 1)], aug_ns['imgmodal2%d' % (k + 1)] = augs[k][0].clone(), augs[k][1].clone()
            else:
                aug_ns['img%d' % (k + 1)] = augs[k][0].clone()
        ns['augset'] = aug_ns
        if two_modal:          # the CHAOS loader hands one-hot targets; the loop takes channel 1 (:294-295)
            oh = lambda t: torch.stack([1 - t, t], 1)
            ns.update(inphase=xs[0].clone(), outphase=xs[1].clone(), targets=oh(t1), targets1=oh(t1), targets2=oh(t2))
        else:
            ns.update(inputs=xs[0].clone(), targets=t1.clone(), targets1=t1.clone(), targets2=t2.clone())
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            exec(body, ns)
        ref = dict(loss1=ns['loss1'].detach(), loss2=ns['loss2'].detach(), indx1=ns['indx1'], indx2=ns['indx2'],
                   loss1_pre=ns['loss1_segpre'].detach(), loss2_pre=ns['loss2_segpre'].detach(),
                   outputs1=ns['outputs1'].detach(), outputs2=ns['outputs2'].detach(),
                   pl1=ns['pseudo_label1'], pl2=ns['pseudo_label2'], wm1=ns['weightmap1'], wm2=ns['weightmap2'],
                   g1=torch.stack([p.grad.double().norm() for p in net1.parameters()]),
                   g2=torch.stack([p.grad.double().norm() for p in net2.parameters()]),
                   head1=net1.last_conv1.weight.detach().clone(), head2=net2.last_conv1.weight.detach().clone())
        bn_ref = [m for m in net1.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
        ref['nbt'] = bn_ref.num_batches_tracked.clone()
        ref['rm'] = bn_ref.running_mean.clone()
        assert net1.training and net2.training
        assert ns['train_count'] == n

        # ---- the oracle restatement ----
        o1n, o2n = nets(oracle, oracle)
        crit = oracle.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)
        corr = oracle.MulticlassMSELoss(reduction='none')
        oa = torch.optim.Adam(o1n.parameters(), lr=1e-4, amsgrad=True)
        ob = torch.optim.Adam(o2n.parameters(), lr=1e-4, amsgrad=True)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            r = steps.proposed_step(o1n, o2n, crit, corr, oa, ob, xs[0], xs[1] if two_modal else None,
                                    [tuple(a) for a in augs], t1, t2, rate, temperature=temp,
                                    reverse=lambda lst: steps.reverseaug(augset, lst, 2), keep=keep, eval_aug=eval_aug,
                                    sharpen_fn=sharpen_fn)
        r['g1'] = torch.stack([p.grad.double().norm() for p in o1n.parameters()])
        r['g2'] = torch.stack([p.grad.double().norm() for p in o2n.parameters()])
        r['head1'], r['head2'] = o1n.last_conv1.weight.detach().clone(), o2n.last_conv1.weight.detach().clone()
        bn_o = [m for m in o1n.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
        r['nbt'], r['rm'] = bn_o.num_batches_tracked.clone(), bn_o.running_mean.clone()
        for k in ('outputs1', 'outputs2', 'loss1', 'loss2', 'indx1', 'indx2', 'loss1_pre', 'loss2_pre', 'pl1', 'pl2', 'wm1', 'wm2',
                  'g1', 'g2', 'head1', 'head2', 'nbt', 'rm'):
            _same(r[k], ref[k], 'g20 %s %s' % (name, k))
        for k in ('loss1', 'loss2', 'indx1', 'indx2', 'loss1_pre', 'loss2_pre', 'g1', 'g2', 'head1', 'head2', 'nbt', 'rm'):
            fx[key + k] = _np(ref[k])
        for k in ('outputs1', 'outputs2', 'pl1', 'pl2', 'wm1', 'wm2'):
            fx[key + k] = _sub(ref[k], 2048)
        fx[key + 'min_gap1'] = np.diff(np.sort(_np(ref['loss1_pre']))).min()
        fx[key + 'min_gap2'] = np.diff(np.sort(_np(ref['loss2_pre']))).min()
        print('g20 %-6s (loop body lines %d-%d executed) loss %.6f %.6f indx %s %s gaps %.3g %.3g nbt %d' % (
            name, lines[0], lines[1], float(ref['loss1']), float(ref['loss2']), ref['indx1'].tolist(), ref['indx2'].tolist(),
            fx[key + 'min_gap1'], fx[key + 'min_gap2'], int(ref['nbt'])))
    np.savez_compressed(os.path.join(OUT, 'g20_proposed_variants.npz'), **fx)


CLI_SCRIPTS = ('trainchaos_comparison_1case.py', 'trainchaos_proposed_30cases1labeled.py', 'trainkidney_proposed_mask1.py',
               'trainbreast_dataset3_proposed_272cases25labeled.py', 'trainprostate_proposed_isbi3ttransferisbidx.py')


def g21_cli_defaults():
    """The flag tables of the reference train scripts the package mirrors, read from their `parser.add_argument(...)` lines
    (syntax tree; nothing is executed): per script the flag names and the JSON text of their defaults and types -- strings
    and numbers only.  tests/test_abi_and_host.py compares the mirrors' parse_args([]) against it."""
    import ast
    import json
    fx = {}
    for fn in CLI_SCRIPTS:
        tree = ast.parse(open(os.path.join(REF, 'train_files', fn)).read())
        flags, defaults, types = [], [], []
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'add_argument':
                name = ast.literal_eval(node.args[0])
                kw = {k.arg: k.value for k in node.keywords}
                flags.append(name)
                defaults.append(json.dumps(ast.literal_eval(kw['default']) if 'default' in kw else None))
                types.append(kw['type'].id if 'type' in kw else 'str')
        key = fn[:-3]
        fx[key + '/flags'], fx[key + '/defaults'], fx[key + '/types'] = np.asarray(flags), np.asarray(defaults), np.asarray(types)
        print('g21 %-55s %d flags' % (fn, len(flags)))
    np.savez_compressed(os.path.join(OUT, 'g21_cli_defaults.npz'), **fx)


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ['g21']:
        return g21_cli_defaults()
    if sys.argv[1:] == ['g20']:
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return g20_proposed_variants(ref_f, ref_u, ref_utils)
    if sys.argv[1:] == ['g19']:
        return g19_reverseaug()
    if sys.argv[1:] == ['g18']:
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return g18_dropregionce_scale(ref_utils)
    if sys.argv[1:] == ['g8']:
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return g8_pixelcoreg(ref_utils)
    if sys.argv[1:] == ['g7']:
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return g7_coteach_ext(ref_utils)
    if sys.argv[1:] == ['g9']:           # attention variants only
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        import oracle
        g1_model('fuseunetsa', ref_f.fuseunetsa, oracle.fuseunetsa, {}, True, ref_utils)
        return g1_model('unetsa', ref_u.UNetsa, oracle.UNetsa, {}, False, ref_utils)
    if sys.argv[1:] == ['g10']:          # separate-encoder attention net and the UNet width variants
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        import oracle
        return g10_variants(ref_f, ref_u, ref_utils, oracle)
    if sys.argv[1:] == ['g11']:          # PolyLR learning-rate sequence
        ref_f, ref_u, ref_utils = _import_reference()
        return g11_polylr(ref_utils)
    if sys.argv[1:] == ['g12']:          # binary metrics and the two remaining loss classes
        ref_f, ref_u, ref_utils = _import_reference()
        return g12_metrics(ref_utils)
    if sys.argv[1:] and sys.argv[1] in ('g2', 'g13', 'g14', 'g15', 'g16'):      # full-size digests of BASELINE configs 2, 4, 3, 5
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return {'g2': lambda: g2_config(ref_f, ref_utils), 'g13': lambda: g13_config4(ref_u, ref_utils),
                'g14': lambda: g14_config3(ref_f, ref_utils), 'g15': lambda: g15_config5(ref_f, ref_utils), 'g16': lambda: g16_config5_autocast(ref_f, ref_utils)}[sys.argv[1]]()
    if sys.argv[1:] == ['g17']:         # num_classes > 2
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return g17_multiclass(ref_f, ref_u, ref_utils)
    if sys.argv[1:] == ['g6']:
        torch.set_num_threads(8)
        ref_f, ref_u, ref_utils = _import_reference()
        return g6_inference(ref_f, ref_u, ref_utils)
    torch.set_num_threads(8)
    ref_f, ref_u, ref_utils = _import_reference()
    import oracle
    g1_model('fuseunet', ref_f.fuseunet, oracle.fuseunet, {}, True, ref_utils)
    g1_model('fuseunet_learned', ref_f.fuseunet, oracle.fuseunet, dict(learned_bilinear=True), True, ref_utils)
    g1_model('unet', ref_u.UNet, oracle.UNet, {}, False, ref_utils)
    g1_model('unet_learned', ref_u.UNet, oracle.UNet, dict(learned_bilinear=True), False, ref_utils)
    g1_model('fuseunetsa', ref_f.fuseunetsa, oracle.fuseunetsa, {}, True, ref_utils)
    g1_model('unetsa', ref_u.UNetsa, oracle.UNetsa, {}, False, ref_utils)
    g3_losses(ref_utils)
    g4_proposed(ref_f, ref_utils)
    g5_adam(ref_f, ref_utils)
    g2_config(ref_f, ref_utils)
    g13_config4(ref_u, ref_utils)
    g14_config3(ref_f, ref_utils)
    g15_config5(ref_f, ref_utils)
    g16_config5_autocast(ref_f, ref_utils)
    g6_inference(ref_f, ref_u, ref_utils)
    g7_coteach_ext(ref_utils)
    g8_pixelcoreg(ref_utils)
    g10_variants(ref_f, ref_u, ref_utils, oracle)
    g11_polylr(ref_utils)
    g12_metrics(ref_utils)
    g17_multiclass(ref_f, ref_u, ref_utils)
    g18_dropregionce_scale(ref_utils)
    g19_reverseaug()
    g20_proposed_variants(ref_f, ref_u, ref_utils)
    print('all golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
