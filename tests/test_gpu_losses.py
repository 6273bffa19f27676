"""-m gpu: fused loss / co-teaching kernels through the drop-in modules, against the golden vectors
produced by the real reference (tests/golden/g3_losses.npz) and against the oracle on fresh inputs.
fp32 tolerance 1e-4 relative (north star: 1e-3); selection indices bit-exact."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def close(a, b, rtol=1e-4, what=''):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b)).double()
    err = (a - b).abs().max().item()
    assert err <= rtol * (b.abs().max().item() + 1e-12) + 1e-9, '%s: err %.3e (scale %.3e)' % (what, err, b.abs().max().item())


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(GOLD, 'g3_losses.npz'))


def test_golden_losses(dev, fx):
    from aide_amd import utils as U
    z1 = torch.from_numpy(fx['z1']).to(dev)
    t = torch.from_numpy(fx['targets']).to(dev)
    for wname, cw, cdw in (('w11', [1.0, 1.0], [1.0, 1.0]), ('w13', [1.0, 3.0], [0.7, 1.6])):
        cw, cdw = torch.tensor(cw), torch.tensor(cdw)            # CPU tensors, as the reference scripts pass them
        for lname, kw in (('CrossEntropyLoss2d', dict(weight=cw)), ('MulticlassDiceLoss', dict(weight=cw)),
                          ('DiceLoss', {}), ('CEMDiceLoss', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw)),
                          ('CEMDiceLossImage', dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw))):
            zz = z1.clone().requires_grad_(True)
            v = getattr(U, lname)(**kw)(zz, t)
            (v.sum() if v.dim() else v).backward()
            close(v, fx['%s/%s' % (lname, wname)], what=lname + wname)
            close(zz.grad, fx['%s/%s/grad' % (lname, wname)], what=lname + wname + ' grad')
    for red in ('none', 'sum'):
        v = U.CrossEntropyLoss2d(weight=torch.tensor([1.0, 3.0]), reduction=red)(z1, t)
        close(v, fx['CrossEntropyLoss2d/w13/' + red], what='CE ' + red)
    close(U.Dice_fn(z1, t), fx['Dice_fn'], what='Dice_fn')


def test_golden_mse_consistency(dev, fx):
    from aide_amd import utils as U
    zz = torch.from_numpy(fx['z1']).to(dev).requires_grad_(True)
    pseudo, wmap = torch.from_numpy(fx['pseudo']).to(dev), torch.from_numpy(fx['wmap']).to(dev)
    m = U.MulticlassMSELoss(reduction='none')(zz, pseudo)
    v = (wmap * m).mean()                       # the caller-side reduction of the reference loop (:311-313)
    v.backward()
    close(v, fx['mse_wm_mean'], what='mse')
    close(zz.grad, fx['mse_wm_mean/grad'], what='mse grad')


@pytest.mark.parametrize('cname', ['Coteachingloss_dropimage', 'Coteachingloss_weightimage'])
@pytest.mark.parametrize('fr', [0.0, 0.25, 0.5])
def test_golden_coteaching(dev, fx, cname, fr):
    from aide_amd import utils as U
    a1 = torch.from_numpy(fx['z1']).to(dev).requires_grad_(True)
    a2 = torch.from_numpy(fx['z2']).to(dev).requires_grad_(True)
    t = torch.from_numpy(fx['targets']).to(dev)
    op = getattr(U, cname)(weight=1.0, reduction='none')
    l1, l2 = op(a1, a2, t, fr)
    (l1 + l2).backward()
    key = '%s/fr%g' % (cname, fr)
    close(l1, fx[key + '/loss1'], what=key + ' loss1')
    close(l2, fx[key + '/loss2'], what=key + ' loss2')
    close(a1.grad, fx[key + '/grad1'], what=key + ' grad1')
    close(a2.grad, fx[key + '/grad2'], what=key + ' grad2')
    # the argsort mask is bit-exact (adjacent-loss gaps recorded in the fixture are >> fp32 noise)
    assert op.last['argsort1'].cpu().tolist() == fx['coteach/argsort1'].tolist()
    assert op.last['argsort2'].cpu().tolist() == fx['coteach/argsort2'].tolist()
    close(op.last['per_image1'], fx['coteach/per_image1'])
    assert float(fx['coteach/min_gap1']) > 1e-3 and float(fx['coteach/min_gap2']) > 1e-3


def test_coteaching_edge_cases(dev, fx):
    from aide_amd import utils as U
    a1, a2 = torch.from_numpy(fx['z1']).to(dev), torch.from_numpy(fx['z2']).to(dev)
    t = torch.from_numpy(fx['targets']).to(dev)
    l1, l2 = U.Coteachingloss_dropimage(reduction='none')(a1, a2, t, 0.9)      # num_remember == 0
    assert torch.isnan(l1).item() and torch.isnan(l2).item()                    # reference: mean of empty -> NaN
    # ties: identical images -> stable ascending order, lower index first
    z = a1[:1].repeat(4, 1, 1, 1).contiguous()
    tt = t[:1].repeat(4, 1, 1).contiguous()
    op = U.Coteachingloss_dropimage(reduction='none')
    op(z, z.clone(), tt, 0.5)
    assert op.last['argsort1'].cpu().tolist() == [0, 1, 2, 3]
    # non-contiguous targets view (mask[:, 1] of a one-hot [N,5,H,W] tensor)
    onehot = torch.zeros(4, 5, 64, 64, dtype=torch.int64, device=dev)
    onehot[:, 1] = t
    view = onehot[:, 1, :, :]
    assert not view.is_contiguous()
    w = torch.tensor([1.0, 1.0])
    v1 = U.CEMDiceLoss(w, w, w)(a1, view)
    v2 = U.CEMDiceLoss(w, w, w)(a1, t)
    assert torch.equal(v1, v2)
    # ignore_index pixels drop out of the CE term
    ti = t.clone()
    ti[:, :8] = 255
    ref = oracle.CrossEntropyLoss2d()(a1.cpu(), ti.cpu())
    close(U.CrossEntropyLoss2d()(a1, ti), ref, what='ignore_index')


def test_pseudo_label_and_proposed_selection(dev):
    """Inline selection of trainchaos_proposed_30cases1labeled.py:274-321 vs the oracle restatement."""
    from aide_amd import utils as U
    from oracle import steps
    g = torch.Generator().manual_seed(11)
    n, s = 4, 64
    augs = [torch.randn(n, 2, s, s, generator=g) for _ in range(4)]
    o1 = torch.randn(n, 2, s, s, generator=g) * 1.5
    o2 = torch.randn(n, 2, s, s, generator=g) * 1.5
    o1 += torch.tensor([0.0, 0.7, -0.6, 1.3]).view(n, 1, 1, 1) * torch.tensor([-1.0, 1.0]).view(1, 2, 1, 1)
    o2 += torch.tensor([1.0, -0.5, 0.4, -1.2]).view(n, 1, 1, 1) * torch.tensor([-1.0, 1.0]).view(1, 2, 1, 1)
    t1 = (torch.rand(n, s, s, generator=g) > 0.7).long()
    t2 = (torch.rand(n, s, s, generator=g) > 0.6).long()
    for temp in (1.0, 0.5):
        pl_r, wm_r = steps.pseudo_labels(augs, temp)
        pl, wm = U.pseudo_label_ensemble([a.to(dev) for a in augs], temp)
        close(pl, pl_r, what='pseudo label')
        close(wm, wm_r, rtol=1e-4, what='weightmap')
    pl1, wm1 = steps.pseudo_labels(augs, 1.0)
    pl2, wm2 = steps.pseudo_labels(list(reversed(augs)), 1.0)
    pl2 = pl2.flip(0).contiguous(); wm2 = wm2.flip(0).contiguous()
    w = torch.tensor([1.0, 1.0])
    for rate in (0.0, 0.25, 1.0):
        r1, r2 = o1.clone().requires_grad_(True), o2.clone().requires_grad_(True)
        l1r, l2r, i1r, i2r, p1r, p2r = steps.proposed_losses(oracle.CEMDiceLossImage(w, w, w), oracle.MulticlassMSELoss('none'),
                                                             r1, r2, t1, t2, pl1, wm1, pl2, wm2, rate)
        l1r.backward(); l2r.backward()
        d1, d2 = o1.to(dev).requires_grad_(True), o2.to(dev).requires_grad_(True)
        op = U.CoTeachingProposedLoss(cediceweight=w, ceclassweight=w, segcor_weight=(1.0, 10.0), keep=2)
        l1, l2, i1, i2 = op(d1, d2, t1.to(dev), t2.to(dev), pl1.to(dev), wm1.to(dev), pl2.to(dev), wm2.to(dev), rate)
        assert i1.cpu().tolist() == i1r.tolist() and i2.cpu().tolist() == i2r.tolist()      # bit-exact mask
        close(l1, l1r, what='loss1 r=%g' % rate); close(l2, l2r, what='loss2 r=%g' % rate)
        l1.backward()                      # loss1 only touches net 1's logits
        assert d2.grad is None
        l2.backward()
        close(d1.grad, r1.grad, what='dlogits1 r=%g' % rate)
        close(d2.grad, r2.grad, what='dlogits2 r=%g' % rate)


def test_full_size_properties(dev):
    """BASELINE size (4 x 2 x 256 x 256): size-independent identities instead of an oracle run."""
    from aide_amd import utils as U
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(4, 2, 256, 256, generator=g) * 2).to(dev)
    t = (torch.rand(4, 256, 256, generator=g) > 0.9).long().to(dev)
    w = torch.tensor([1.0, 1.0])
    per = U.CEMDiceLossImage(w, w, w)(z, t)
    # permuting the batch permutes the per-image losses bit-exactly (fixed-order reductions) ...
    perm = torch.tensor([2, 0, 3, 1], device=dev)
    per_p = U.CEMDiceLossImage(w, w, w)(z[perm].contiguous(), t[perm].contiguous())
    assert torch.equal(per_p, per[perm])
    # ... CE-mean + Dice-mean equals the mean of the per-image values for unit class weights
    tot = U.CEMDiceLoss(w, w, w)(z, t)
    close(tot, per.mean(), rtol=1e-5)
    # ... shifting both logits by a constant changes nothing (softmax invariance)
    close(U.CEMDiceLossImage(w, w, w)(z + 3.0, t), per, rtol=1e-5)
    # ... and two runs are bit-identical (no atomics)
    assert torch.equal(U.CEMDiceLossImage(w, w, w)(z, t), per)


def test_reverseaug_matches_pil(dev):
    """GPU reverseaug vs the reference's own reverseaug (fixture g19_reverseaug.npz: the function text of
    trainchaos_proposed_30cases1labeled.py:81-95 executed in the build container), incl. PIL's exact fast paths
    (0 / 90 / 180 / 270 degrees), non-square planes and augno < 4 -- and, on larger planes than the fixture holds, vs the
    oracle restatement that the fixture pins bit for bit."""
    import warnings
    from aide_amd import utils as U
    from oracle import steps
    from test_oracle_golden import g19_cases
    fx = np.load(os.path.join(GOLD, 'g19_reverseaug.npz'))
    for c, (augset, ins, outs) in enumerate(g19_cases(fx)):
        got = U.reverseaug(augset, [t.to(dev) for t in ins], 2)
        for k in range(4):
            assert (got[k].cpu() - outs[k]).abs().max().item() < 2e-6, (c, k)
        for b in range(4):                       # the passes k >= augno[b] of an element are left untouched
            for k in range(augset['augno'][b], 4):
                assert torch.equal(got[k][b].cpu(), ins[k][b]), (c, b, k)
    g = torch.Generator().manual_seed(21)
    for (h, w) in ((32, 32), (48, 64)):
        nb = 4
        outs = [torch.randn(nb, 2, h, w, generator=g) for _ in range(4)]
        augset = {'augno': [4, 4, 4, 3],
                  'hflip1': [0, 1, 0, 1], 'degree1': [0.0, 37.5, -60.0, 12.25],
                  'hflip2': [1, 0, 1, 0], 'degree2': [90.0, 180.0, 270.0, -90.0],
                  'hflip3': [0, 0, 1, 1], 'degree3': [59.99, -0.5, 360.0, 45.0],
                  'hflip4': [1, 1, 0, 0], 'degree4': [-33.0, 5.0, 120.0, 77.0]}
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = steps.reverseaug(augset, [o.clone() for o in outs], 2)
        got = U.reverseaug(augset, [o.to(dev) for o in outs], 2)
        for k in range(4):
            assert (got[k].cpu() - ref[k]).abs().max().item() < 2e-6, (h, w, k)
        # element 3 has augno == 3: its 4th output is left untouched by the reference
        assert torch.equal(got[3][3].cpu(), outs[3][3])


def test_kl_bidirection_g7(dev):
    """utils/coteach_loss.py:85-92 vs the reference's map and both gradients (g7_coteach_ext.npz)."""
    from aide_amd import utils as U
    g3, fx = np.load(os.path.join(GOLD, 'g3_losses.npz')), np.load(os.path.join(GOLD, 'g7_coteach_ext.npz'))
    a1 = torch.from_numpy(g3['z1']).to(dev).requires_grad_(True)
    a2 = torch.from_numpy(g3['z2']).to(dev).requires_grad_(True)
    v = U.KLbidirection(a1, a2)
    ref = fx['KL/map']
    assert np.abs(v.detach().cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    gw = torch.linspace(0.5, 1.5, v.numel()).view_as(v).to(dev)
    (v * gw).sum().backward()
    for g, key in ((a1.grad, 'KL/grad1'), (a2.grad, 'KL/grad2')):
        assert np.abs(g.cpu().numpy() - fx[key]).max() < 2e-5 * np.abs(fx[key]).max()


@pytest.mark.parametrize('cname', ['Coteachingloss_dropregionce', 'Coteachingloss_dropimagedroppixel'])
def test_coteach_ext_g7(dev, cname):
    """Coteachingloss_dropregionce (:163-196) / _dropimagedroppixel (:198-254) vs the real reference: both losses
    and, back-propagated separately, their gradients w.r.t. BOTH logit tensors (the KL term of the pixel branch
    reaches the other net), forget rates 0 / 0.25 / 0.5."""
    from aide_amd import utils as U
    g3, fx = np.load(os.path.join(GOLD, 'g3_losses.npz')), np.load(os.path.join(GOLD, 'g7_coteach_ext.npz'))
    z1, z2, t = (torch.from_numpy(g3[k]).to(dev) for k in ('z1', 'z2', 'targets'))
    kw = dict(scale=0.5, reduction='none') if 'region' in cname else dict(weight=1.0, reduction='none')
    for fr in (0.0, 0.25, 0.5):
        key = '%s/fr%g' % (cname, fr)
        for which in (0, 1):
            a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
            ls = getattr(U, cname)(**kw)(a1, a2, t, fr)
            ref = float(fx[key + '/loss%d' % (which + 1)])
            assert abs(ls[which].item() - ref) < 1e-5 * abs(ref), (key, which, ls[which].item(), ref)
            ls[which].backward()
            for g, gk in ((a1.grad, '/l%d_grad1' % (which + 1)), (a2.grad, '/l%d_grad2' % (which + 1))):
                gr = fx[key + gk]
                got = np.zeros_like(gr) if g is None else g.cpu().numpy()
                # selections may differ from the reference only between values closer than fp32 noise
                bad = np.abs(got - gr) > 1e-4 * np.abs(gr).max() + 1e-12
                assert bad.mean() < 2e-4, (key, gk, bad.sum(), np.abs(got - gr).max(), np.abs(gr).max())
    with pytest.raises((RuntimeError, IndexError)):
        getattr(U, cname)(reduction='mean')


def test_dropregionce_scale_g18(dev):
    """Coteachingloss_dropregionce(scale) for the pooling windows the default kernels do not cover (scale 0.25: 4 x 4; scale
    0.3: 3 x 3 with clipped border windows; two and three classes) vs the real reference (g18_dropregionce_scale.npz):
    both losses and their gradients, forget rates 0.25 / 0.5.  (Round 3 raised NotImplementedError for scale != 0.5.)"""
    from aide_amd import utils as U
    g3, fx = np.load(os.path.join(GOLD, 'g3_losses.npz')), np.load(os.path.join(GOLD, 'g18_dropregionce_scale.npz'))
    cases = {'c2': tuple(torch.from_numpy(g3[k]).to(dev) for k in ('z1', 'z2', 'targets')),
             'c3': tuple(torch.from_numpy(fx['c3/' + k]).to(dev) for k in ('z1', 'z2', 'targets'))}
    for cname, (z1, z2, t) in cases.items():
        for scale in (0.25, 0.3):
            for fr in (0.25, 0.5):
                key = '%s/s%g/fr%g' % (cname, scale, fr)
                for which in (0, 1):
                    a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                    ls = U.Coteachingloss_dropregionce(scale=scale, reduction='none')(a1, a2, t, fr)
                    ref = float(fx[key + '/loss%d' % (which + 1)])
                    assert abs(ls[which].item() - ref) < 1e-5 * abs(ref), (key, which, ls[which].item(), ref)
                    ls[which].backward()
                    gr = fx[key + '/grad%d' % (which + 1)]
                    got = (a1 if which == 0 else a2).grad.cpu().numpy()
                    # selections may differ from the reference only between values closer than fp32 noise
                    bad = np.abs(got - gr) > 1e-4 * np.abs(gr).max() + 1e-12
                    assert bad.mean() < 2e-4, (key, bad.sum(), np.abs(got - gr).max(), np.abs(gr).max())
                    other = (a2 if which == 0 else a1).grad
                    assert other is None or float(other.abs().max()) == 0.0
    with pytest.raises(ZeroDivisionError):                        # the reference divides by int(H * scale) = 0
        U.Coteachingloss_dropregionce(scale=0.001, reduction='none')(cases['c2'][0], cases['c2'][1], cases['c2'][2], 0.25)


def test_select_smallest_ties_and_edges(dev):
    """aide_select_smallest: stable ties (lower index first), the > 0 candidate filter, k = int(rr * count),
    k handed over on the device, and the empty selection (-> nan mean like torch.mean of an empty tensor)."""
    from aide_amd.utils.coteach_loss import _select
    g = torch.Generator().manual_seed(3)
    v = torch.rand(5000, generator=g)
    v[100:140] = 0.25                                           # 40 tied values
    v[::7] = 0.0                                                # non-candidates for the positive filter
    v[3] = -1.0
    vd = v.to(dev)
    for only_pos, k in ((False, 1300), (True, 1300), (True, 0), (False, 5000)):
        cand = (v > 0) if only_pos else torch.ones_like(v, dtype=torch.bool)
        order = torch.from_numpy(np.argsort(np.where(cand.numpy(), v.numpy(), np.inf), kind='stable'))
        kk = min(k, int(cand.sum()))
        ref = torch.zeros(5000, dtype=torch.uint8); ref[order[:kk]] = 1
        mask, sums, ks = _select(vd, vd, 1, 5000, k_host=k, only_positive=only_pos)
        assert int(ks[0]) == kk and torch.equal(mask.cpu(), ref)
        assert abs(float(sums[0]) - float(v.double()[ref.bool()].sum())) < 1e-9
    mask, sums, ks = _select(vd, vd, 1, 5000, rr=0.75, only_positive=True)
    assert int(ks[0]) == int(0.75 * int((v > 0).sum()))
    mask2, _, ks2 = _select(vd, vd, 1, 5000, k_in=ks, only_positive=True)
    assert torch.equal(mask, mask2) and int(ks2[0]) == int(ks[0])
    z = torch.zeros(64, device=dev)
    _, s0, k0 = _select(z, z, 1, 64, rr=0.5, only_positive=True)
    assert int(k0[0]) == 0 and torch.isnan(s0[0] / k0[0].double())


@pytest.mark.parametrize('cname', ['Pixelcoreg_Focalloss', 'Pixelcoreg_Focalloss_twomodel'])
def test_pixelcoreg_g8(dev, cname):
    """utils/reg_loss.py:58-193 vs the real reference (g8_pixelcoreg.npz): loss, kept foreground fraction and the
    gradients of every input (the three-net form differentiates net 3 only)."""
    from aide_amd import utils as U
    g3, fx = np.load(os.path.join(GOLD, 'g3_losses.npz')), np.load(os.path.join(GOLD, 'g8_pixelcoreg.npz'))
    three = cname == 'Pixelcoreg_Focalloss'
    zs = [torch.from_numpy(g3['z1']), torch.from_numpy(g3['z2'])] + ([torch.from_numpy(fx['z3'])] if three else [])
    t = torch.from_numpy(g3['targets']).to(dev)
    for fr, kd, red in ((0.0, 0.3, 'mean'), (0.25, 0.3, 'mean'), (0.5, 0.7, 'sum')):
        key = '%s/fr%g_kd%g_%s' % (cname, fr, kd, red)
        a = [z.clone().to(dev).requires_grad_(True) for z in zs]
        loss, frac = getattr(U, cname)(reduction=red)(*a, t, fr, kd, dev)
        assert abs(loss.item() - float(fx[key + '/loss'])) < 2e-5 * abs(float(fx[key + '/loss'])), (key, loss.item())
        assert abs(frac.item() - float(fx[key + '/frac'])) < 2e-3
        loss.backward()
        for i, x in enumerate(a):
            gr = fx[key + '/grad%d' % (i + 1)]
            got = np.zeros_like(gr) if x.grad is None else x.grad.cpu().numpy()
            bad = np.abs(got - gr) > 1e-4 * max(np.abs(gr).max(), 1e-30) + 1e-12
            assert bad.mean() < 2e-4, (key, i, bad.sum())
    with pytest.raises(NotImplementedError):
        getattr(U, cname)(reduction='none')


G12_LOSSES = [('Dice_Loss', dict(smooth=1.0, reduction='mean'), 'Dice_Loss/reduction=mean_smooth=1.0'),
              ('Dice_Loss', dict(smooth=0.5, reduction='sum'), 'Dice_Loss/reduction=sum_smooth=0.5'),
              ('Dice_Loss', dict(reduction='none'), 'Dice_Loss/reduction=none'),
              ('CEDiceLoss', dict(cediceweight=[0.7, 1.6], classweight=[1.0, 3.0]), 'CEDiceLoss/cediceweight=w_classweight=w'),
              ('CEDiceLoss', dict(reduction='sum'), 'CEDiceLoss/reduction=sum'), ('CEDiceLoss', dict(), 'CEDiceLoss/')]
G12_BATCHES = [(slice(0, 5), 'all'), (slice(0, 1), 'first'), (slice(3, 5), 'nonempty')]


def test_golden_metrics_and_remaining_losses(dev):
    """Dice_fn / Dice_fn_Nozero / TP_TN_FP_FN / IoU_fn (metrics2d.py:8-84) from the fused statistics kernel and Dice_Loss /
    CEDiceLoss (loss2d.py:63-85,156-171) against values of the reference (tests/golden/g12_metrics.npz), including the
    images that are empty in the target (and in the prediction)."""
    from aide_amd import utils as U
    fx = np.load(os.path.join(GOLD, 'g12_metrics.npz'))
    z, t = torch.from_numpy(fx['z']).to(dev), torch.from_numpy(fx['targets']).to(dev)
    for batch, tag in G12_BATCHES:
        zz, tt = z[batch].contiguous(), t[batch].contiguous()
        close(U.Dice_fn(zz, tt), fx[tag + '/Dice_fn'], what='Dice_fn ' + tag)
        d, c = U.Dice_fn_Nozero(zz, tt)
        assert isinstance(d, float) and isinstance(c, int)
        close(torch.tensor([d, float(c)]), fx[tag + '/Dice_fn_Nozero'], what='Dice_fn_Nozero ' + tag)
        got = torch.stack([v.cpu() for v in U.TP_TN_FP_FN(zz, tt)])
        assert torch.equal(got, torch.from_numpy(fx[tag + '/TP_TN_FP_FN'])), 'confusion counts are integers: exact'
    close(U.IoU_fn(z[3:5].contiguous(), t[3:5].contiguous()), fx['nonempty/IoU_fn'], what='IoU_fn')
    assert torch.isnan(U.IoU_fn(z, t))            # an image empty in both: 0/0, as the reference
    with pytest.raises(NotImplementedError):
        U.Dice_fn(z, t, threshold=0.3)
    for lname, kw, key in G12_LOSSES:
        kw = {k: torch.tensor(v) if isinstance(v, list) else v for k, v in kw.items()}
        zz = z.clone().requires_grad_(True)
        v = getattr(U, lname)(**kw)(zz, t)
        (v.sum() if v.dim() else v).backward()
        close(v, fx[key], what=key)
        close(zz.grad, fx[key + '/grad'], what=key + ' grad')


G12_BRANCHES = [('CrossEntropyLoss2d', dict(weight=[1.0, 3.0]), 'z', 'onehot', 'weight=w'),
                ('CrossEntropyLoss2d', dict(reduction='sum'), 'z', 'onehot', 'reduction=sum'),
                ('MulticlassDiceLoss', dict(weight=[0.3, 1.7]), 'z', 'onehot', 'weight=w'),
                ('MulticlassDiceLoss', dict(reduction='none'), 'z', 'onehot', 'reduction=none'),
                ('MulticlassDiceLoss', dict(weight=[0.3, 1.7], smooth=0.5, reduction='sum'), 'z', 'onehot',
                 'reduction=sum_smooth=0.5_weight=w'),
                ('DiceLoss', dict(), 'prob', 't', ''), ('DiceLoss', dict(smooth=0.25, reduction='none'), 'prob', 't',
                                                       'reduction=none_smooth=0.25'),
                ('DiceLoss', dict(reduction='sum'), 'prob', 'onehot1', 'reduction=sum')]


def test_loss_branches_off_the_hot_path(dev):
    """One-hot targets in CrossEntropyLoss2d (utils/loss2d.py:11-12) and MulticlassDiceLoss with class weights (:98-104),
    DiceLoss on a probability input (:47-48): values and input gradients of the reference (g12_metrics.npz 'branch/...')."""
    from aide_amd import utils as U
    fx = np.load(os.path.join(GOLD, 'g12_metrics.npz'))
    src = dict(z=torch.from_numpy(fx['z']).to(dev), prob=torch.from_numpy(fx['prob']).to(dev))
    onehot = torch.from_numpy(fx['onehot']).to(dev)
    tgt = dict(onehot=onehot, t=torch.from_numpy(fx['targets']).to(dev), onehot1=onehot[:, 1].contiguous())
    for lname, kw, xin, tin, tag in G12_BRANCHES:
        kw = {k: torch.tensor(v) if isinstance(v, list) else v for k, v in kw.items()}
        x = src[xin].clone().requires_grad_(True)
        v = getattr(U, lname)(**kw)(x, tgt[tin])
        if v.dim():
            (v * torch.arange(1, v.numel() + 1, device=dev).float()).sum().backward()
        else:
            v.backward()
        key = 'branch/%s/%s/%s' % (lname, tin, tag)
        close(v, fx[key], what=key)
        close(x.grad, fx[key + '/grad'], what=key + ' grad')
        if lname == 'DiceLoss':
            # the reference flattens per image (`view(N, -1)`, loss2d.py:47-50): a 2-D [N, H*W] input and an [N,1,H,W]
            # target are the same problem and must give the same value and gradient
            n = x.shape[0]
            x2 = src[xin].clone().reshape(n, -1).requires_grad_(True)
            v2 = getattr(U, lname)(**kw)(x2, tgt[tin].unsqueeze(1))
            if v2.dim():
                (v2 * torch.arange(1, v2.numel() + 1, device=dev).float()).sum().backward()
            else:
                v2.backward()
            close(v2, fx[key], what=key + ' (flattened)')
            close(x2.grad.reshape(x.shape), fx[key + '/grad'], what=key + ' grad (flattened)')
