"""-m gpu: whole-network parity through the drop-in nn.Modules (C-ABI underneath).

Forward quantities (logits, loss, BN running statistics, eval-mode logits) are compared tightly with
the golden vectors of the real reference.  Gradients: two correct fp32 implementations differ by
~1e-6 in a pre-activation, which occasionally puts it on the other side of zero; one such ReLU-mask
flip changes the gradients of that layer and of everything upstream by O(1/pixels) — far above 1e-3 on
a 32x32 problem, irrelevant at 256x256 (see test_config2_digest_256).  The gradient check is therefore
made flip-free BY CONSTRUCTION: the oracle's backward is evaluated with the ReLU masks of our forward
(F.relu temporarily replaced by `x * mask`, a <= 1e-5 perturbation of its forward), and then every
parameter gradient is held to 1e-3.  The number of flipped mask elements is asserted to be tiny."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
RTOL = 1e-3


def rel(a, b):
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a, b = a.detach().cpu().double(), torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def sub(a, limit=8192):
    a = a.detach().cpu().numpy()
    if a.size <= limit:
        return a
    return a.reshape(-1)[::-(-a.size // limit)]


def build_pair(kind, learned, dev):
    from aide_amd.models_twomodalinputs import fuseunet, fuseunetsa, fuseunetsaseparate
    from aide_amd.models_singlemodalinput import UNet, UNetsa, UNet128, UNet32, UNet16, UNet2
    ours_c, ref_c = {'fuseunet': (fuseunet, oracle.fuseunet), 'unet': (UNet, oracle.UNet),
                     'fuseunetsa': (fuseunetsa, oracle.fuseunetsa), 'unetsa': (UNetsa, oracle.UNetsa),
                     'fuseunetsaseparate': (fuseunetsaseparate, oracle.fuseunetsaseparate),
                     'unet128': (UNet128, oracle.UNet128), 'unet32': (UNet32, oracle.UNet32),
                     'unet16': (UNet16, oracle.UNet16), 'unet2': (UNet2, oracle.UNet2)}[kind]
    if kind.endswith('sa') or kind.endswith('separate'):
        assert not learned
        torch.manual_seed(2)
        ref = ref_c(2)
        torch.manual_seed(2)
        return ours_c(2).to(dev), ref
    torch.manual_seed(2)
    ref = ref_c(2, learned_bilinear=learned)
    torch.manual_seed(2)
    net = ours_c(2, learned_bilinear=learned).to(dev)
    return net, ref


class forced_relu_masks(object):
    """Context manager: torch.nn.functional.relu := x * (our post-ReLU mask of the BN that ran last)."""

    def __init__(self, net, ref, plan):
        name_of = {id(m): n for n, m in net.named_modules()}
        self.masks = {}
        for st in plan.steps:
            if st['kind'] in ('conv', 'convT'):
                if st.get('head_lazy'):
                    # the layer under the head never stores its activation (config.lazy_head): the sign of fma(z, scale, shift) is the
                    # sign of the exact value, which float64 evaluates exactly
                    a = st['z'].detach().cpu().double() * st['scale'].cpu().double().view(1, -1, 1, 1) + \
                        st['shift'].cpu().double().view(1, -1, 1, 1)
                    self.masks[name_of[id(st['bn'])]] = (a > 0).float()
                    continue
                self.masks[name_of[id(st['bn'])]] = (plan.view(st['dst']).detach().cpu() > 0).float()
        # max-pool arg-max: near-ties (two window elements within the fp32 noise) are the same kind of
        # discrete event; the oracle's pooling is evaluated with OUR window winners as well
        self.pool_idx = {}
        for st in plan.steps:
            if st['kind'] == 'pool':
                src = plan.view(st['src']).detach().cpu().float()        # (a bf16-stored buffer in the bf16 mode)
                _, idx = torch.nn.functional.max_pool2d(src, 2, 2, return_indices=True)
                self.pool_idx[tuple(src.shape[2:])] = idx
        self.pool_flips = 0
        self.pool_calls = {}
        self.ref, self.cur, self.flips, self.hooks = ref, [None], {}, []
        self.total = sum(m.numel() for m in self.masks.values())

    def __enter__(self):
        import torch.nn.functional as F
        names = {id(m): n for n, m in self.ref.named_modules()}

        def hook(mod, inp, out):
            if names[id(mod)] not in self.masks:
                return                          # Spatial_Attention's BatchNorm2d(1): sigmoid follows, not a ReLU
            self.cur[0] = names[id(mod)]
            n = int(((out.detach() > 0).float() != self.masks[self.cur[0]]).sum())
            if n:
                self.flips[self.cur[0]] = n
        for m in self.ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                self.hooks.append(m.register_forward_hook(hook))
        self.orig = F.relu
        F.relu = lambda x, inplace=False: x * self.masks[self.cur[0]]
        self.orig_pool = F.max_pool2d

        def forced_pool(x, kernel_size, stride=None, *a, **k):
            idx = self.pool_idx[tuple(x.shape[2:])]
            calls = self.pool_calls[tuple(x.shape[2:])] = self.pool_calls.get(tuple(x.shape[2:]), 0) + 1
            if calls == 1:
                idx = idx[:, :x.shape[1]]                      # modal-1: the whole fused tensor (fuseunetsaseparate: its own leading slice)
            else:
                idx = idx[:, idx.shape[1] - x.shape[1]:]       # modal-2 pools the trailing channel slice
            _, own = self.orig_pool(x.detach(), 2, 2, return_indices=True)
            self.pool_flips += int((own != idx).sum())
            n, c, h, w = x.shape
            return x.reshape(n, c, h * w).gather(2, idx.reshape(n, c, -1)).reshape(n, c, h // 2, w // 2)
        F.max_pool2d = forced_pool
        return self

    def __exit__(self, *a):
        import torch.nn.functional as F
        F.relu = self.orig
        F.max_pool2d = self.orig_pool
        for h in self.hooks:
            h.remove()


CASES = [('fuseunet', False, 'g1_fuseunet.npz'), ('fuseunet', True, 'g1_fuseunet_learned.npz'),
         ('unet', False, 'g1_unet.npz'), ('unet', True, 'g1_unet_learned.npz'),
         ('fuseunetsa', False, 'g1_fuseunetsa.npz'), ('unetsa', False, 'g1_unetsa.npz'),      # attention variants
         ('fuseunetsaseparate', False, 'g1_fuseunetsaseparate.npz'),                         # fuseunet.py:210-322
         ('unet128', False, 'g1_unet128.npz'), ('unet32', False, 'g1_unet32.npz'),           # UNet.py:210-272
         ('unet16', False, 'g1_unet16.npz'), ('unet2', False, 'g1_unet2.npz')]               # narrow: partial channel tiles


@pytest.mark.parametrize('kind,learned,gold', CASES)
def test_golden_forward_backward_adam(dev, kind, learned, gold):
    from aide_amd import utils as U
    from aide_amd.optim import Adam
    fx = np.load(os.path.join(GOLD, gold))
    net, ref = build_pair(kind, learned, dev)
    nin = 2 if kind.startswith('fuseunet') else 1
    xs = [torch.from_numpy(fx['x%d' % i]) for i in range(nin)]
    t = torch.from_numpy(fx['targets'])
    w = torch.tensor([1.0, 1.0])
    net.train(); ref.train()
    out = net(*[x.to(dev) for x in xs])
    assert rel(out, fx['logits']) < RTOL
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    loss = crit(out, t.to(dev))
    assert abs(loss.item() - float(fx['loss'])) < RTOL * float(fx['loss'])
    per = U.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)(out.detach(), t.to(dev))
    assert rel(per, fx['per_image_loss']) < RTOL
    loss.backward()
    plan = list(net.engine.plans.values())[0]
    with forced_relu_masks(net, ref, plan) as fm:
        out_r = ref(*xs)
        oracle.CEMDiceLoss(w, w, w)(out_r, t).backward()
    flips = fm.flips
    assert sum(flips.values()) <= 4 + 2e-5 * fm.total, 'implausibly many ReLU mask flips: %s' % flips
    assert rel(out, out_r) < RTOL
    names = [str(n) for n in fx['param_names']]
    assert names == [k for k, _ in net.named_parameters()]
    for (k, p), (_, q), gabs in zip(net.named_parameters(), ref.named_parameters(), fx['grad_absmax']):
        err = (p.grad.cpu().double() - q.grad.double()).abs().max().item()
        if (k.endswith('.bias') and float(gabs) < 1e-6) or k.endswith('.conv4.bias'):
            # mathematically zero: a conv bias that feeds a BatchNorm (Spatial_Attention.conv4 -> bn too)
            assert err < 1e-5, 'dead conv bias %s: abs err %.2e' % (k, err)
        else:
            scale = q.grad.abs().max().item()
            assert err <= RTOL * scale, '%s: grad err %.2e scale %.2e (flips %s)' % (k, err, scale, flips)
    taint = set()
    if not flips:                      # then the reference's own gradients (golden) must match as well
        for k in fx.files:
            if k.startswith('grad/') and float(fx['grad_absmax'][names.index(k[5:])]) >= 1e-6:
                p = dict(net.named_parameters())[k[5:]]
                assert rel(torch.from_numpy(sub(p.grad)), fx[k]) < RTOL, k
    # Adam(amsgrad) step on the live parameters, BN running statistics, eval-mode forward
    torch.optim.Adam(ref.parameters(), lr=1e-4, amsgrad=True).step()
    Adam(net.parameters(), lr=1e-4, amsgrad=True).step()
    # (the first amsgrad step is lr * g / (|g| + eps): elements whose gradient is comparable to
    # eps = 1e-8 move by an ill-conditioned fraction of lr in ANY fp32 implementation -> masked out)
    ref_grads = {k: q.grad.clone() for k, q in ref.named_parameters()}
    for (k, p), (_, q), gabs in zip(net.named_parameters(), ref.named_parameters(), fx['grad_absmax']):
        if float(gabs) < 1e-6 or id(p) in taint or k.endswith('.conv4.bias'):
            continue                                   # dead biases random-walk in the reference too
        well = ref_grads[k].abs() > 1e-5
        d = (p.detach().cpu() - q.detach()).abs()
        assert (not well.any()) or d[well].max().item() < 2e-6, k          # lr = 1e-4 sized steps
        assert d.max().item() <= 2.1e-4, k
    for (k, b), (_, c) in zip(net.named_buffers(), ref.named_buffers()):
        if 'num_batches_tracked' in k:
            assert int(b) == int(c) == 1
        else:
            assert rel(b, c) < RTOL, k
    net.eval(); ref.eval()
    with torch.no_grad():
        ev, ev_r = net(*[x.to(dev) for x in xs]), ref(*xs)
    assert rel(ev, ev_r) < 5e-3
    if not flips:
        assert rel(ev, fx['eval_logits']) < 5e-3
    with pytest.raises(RuntimeError):
        net(*[x.to(dev).requires_grad_(False) for x in xs]).sum().backward()     # eval-mode backward


def test_reference_checkpoint_roundtrip(dev):
    """state_dict produced by the oracle (== reference format) loads and reproduces its eval output."""
    net, ref = build_pair('fuseunet', False, dev)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.01 * torch.randn_like(p))
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(0)
    x1, x2 = torch.randn(1, 3, 48, 32, generator=g), torch.randn(1, 3, 48, 32, generator=g)   # 3x2 at the bottom
    net.eval(); ref.eval()
    with torch.no_grad():
        assert rel(net(x1.to(dev), x2.to(dev)), ref(x1, x2)) < RTOL


def test_config2_digest_256(dev):
    """BASELINE config 2: FuseUNet N=4, 256x256 on the synthetic CHAOS batch vs digests of the real
    reference's run (tests/golden/g2_config2.npz)."""
    from aide_amd import utils as U
    from aide_amd.synthetic import chaos_batch
    from aide_amd.models_twomodalinputs import fuseunet
    fx = np.load(os.path.join(GOLD, 'g2_config2.npz'))
    xin, xout, t = chaos_batch(4, 256, seed=int(fx['seed']))
    torch.manual_seed(2)
    net = fuseunet(2).to(dev)
    net.train()
    out = net(xin.to(dev), xout.to(dev))
    assert rel(out[:, :, ::37, :], fx['logits_rows']) < RTOL
    assert abs(out.double().sum().item() - float(fx['logits_sum'])) < RTOL * float(fx['logits_abs_sum'])
    w = torch.tensor([1.0, 1.0])
    loss = U.CEMDiceLoss(w, w, w)(out, t.to(dev))
    assert abs(loss.item() - float(fx['loss'])) < 1e-4 * float(fx['loss'])
    per = U.CEMDiceLossImage(w, w, w)(out.detach(), t.to(dev))
    assert rel(per, fx['per_image_loss']) < 1e-4
    loss.backward()
    # at 256x256 individual mask flips average out: per-parameter gradient norms within 1e-3
    gn = np.array([p.grad.double().norm().item() for p in net.parameters()])
    live = fx['grad_norms'] > 1e-5
    assert np.max(np.abs(gn[live] - fx['grad_norms'][live]) / fx['grad_norms'][live]) < 5e-3
    # determinism: a second forward/backward is bit-identical (fixed-order reductions, no atomics)
    g1 = [p.grad.clone() for p in net.parameters()]
    with torch.no_grad():                      # undo the running-stat update so inputs are identical
        pass
    net.zero_grad()
    out2 = net(xin.to(dev), xout.to(dev))
    assert torch.equal(out2, out)
    U.CEMDiceLoss(w, w, w)(out2, t.to(dev)).backward()
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))


def test_backward_after_newer_forward_is_refused(dev):
    net, _ = build_pair('fuseunet', False, dev)
    x = torch.randn(1, 3, 32, 32, device=dev)
    a = net(x, x)
    net(x, x)
    with pytest.raises(RuntimeError, match='newer forward'):
        a.sum().backward()
    with pytest.raises(RuntimeError, match='multiples of 16'):
        net(torch.randn(1, 3, 40, 40, device=dev), torch.randn(1, 3, 40, 40, device=dev))


def test_side_stream_wgrad_is_bit_identical(dev):
    """Weight gradients run on a side stream; serial and overlapped schedules must agree bit for bit
    (same kernels, fixed-order reductions) — also a race detector for the shared slab workspace."""
    from aide_amd import utils as U
    for kind, learned in (('fuseunet', False), ('unet', True)):
        net, _ = build_pair(kind, learned, dev)
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(2, 3, 64, 64, generator=g).to(dev) for _ in range(2 if kind == 'fuseunet' else 1)]
        t = (torch.rand(2, 64, 64, generator=g) > 0.8).long().to(dev)
        w = torch.tensor([1.0, 1.0])
        grads = []
        for overlap in (True, False, True):
            net.zero_grad()
            out = net(*xs)
            list(net.engine.plans.values())[0].overlap = overlap
            U.CEMDiceLoss(w, w, w)(out, t).backward()
            grads.append([p.grad.clone() for p in net.parameters()])
        for a, b, c in zip(*grads):
            assert torch.equal(a, b) and torch.equal(a, c)


def test_engine_assigned_gradients_follow_autograd_semantics(dev):
    """Parameter gradients are assigned by the engine (one autograd anchor, persistent arena): the values equal the plain
    autograd form bit for bit, a second backward ACCUMULATES (as AccumulateGrad does), zero_grad(set_to_none=False) keeps
    working, a frozen parameter gets no gradient, and the fused Adam steps identically from either form."""
    from aide_amd import engine as E, utils as U
    from aide_amd.optim import Adam
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(2, 3, 32, 32, generator=g).to(dev) for _ in range(2)]
    t = (torch.rand(2, 32, 32, generator=g) > 0.7).long().to(dev)
    w = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(w, w, w)

    def grads_of(direct, passes=1, zero_in_place=False, freeze=False):
        net, _ = build_pair('fuseunet', False, dev)
        net.engine.config.direct_grads = direct
        params = list(net.parameters())
        if freeze:
            params[5].requires_grad_(False)
        opt = Adam([p for p in params if p.requires_grad], lr=1e-3, amsgrad=True)
        if zero_in_place:                       # gradients exist and are zero: the backward must add into them
            crit(net(*xs), t).backward()
            opt.zero_grad(set_to_none=False)
            assert all(float(p.grad.abs().max()) == 0.0 for p in params if p.requires_grad)
        for _ in range(passes):
            crit(net(*xs), t).backward()
        out = [None if p.grad is None else p.grad.clone() for p in params]
        opt.step()
        torch.cuda.synchronize()
        return out, [p.detach().clone() for p in params]
    ref, pref = grads_of(False)
    got, pgot = grads_of(True)
    assert all(torch.equal(a, b) for a, b in zip(ref, got)), 'engine-assigned gradients differ from the autograd form'
    assert all(torch.equal(a, b) for a, b in zip(pref, pgot)), 'Adam step differs'
    twice, _ = grads_of(True, passes=2)
    twice_ref, _ = grads_of(False, passes=2)
    for a, b, c in zip(ref, twice, twice_ref):
        assert torch.equal(b, c), 'accumulated gradients differ from the autograd form'
        assert torch.allclose(b, 2 * a, rtol=1e-6, atol=1e-12)
    inplace, _ = grads_of(True, zero_in_place=True)
    assert all(torch.equal(a, b) for a, b in zip(ref, inplace)), 'zero_grad(set_to_none=False) path'
    frozen, _ = grads_of(True, freeze=True)
    assert frozen[5] is None and all(torch.equal(a, b) for k, (a, b) in enumerate(zip(ref, frozen)) if k != 5)


def test_engine_assigned_gradients_keep_torch_contracts(dev):
    """What the engine-assigned gradient path must not break (round-3 advisor findings): (1) a gradient tensor the caller
    still holds after zero_grad() stays intact -- the next backward takes a new arena instead of overwriting it;
    (2) tensor hooks on parameters fire (the module falls back to the plain autograd form while any are registered);
    (3) a forward issued on a non-current stream: after backward() the CALLER's stream is ordered behind the whole pass
    (autograd's leaf-stream synchronisation, through the anchor's AccumulateGrad) -- no hand-written wait_stream."""
    from aide_amd import utils as U
    g = torch.Generator().manual_seed(12)
    w = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(w, w, w)

    def batch(n, size):
        xs = [torch.randn(n, 3, size, size, generator=g).to(dev) for _ in range(2)]
        return xs, (torch.rand(n, size, size, generator=g) > 0.7).long().to(dev)
    # (1) held gradients
    net, _ = build_pair('fuseunet', False, dev)
    params = list(net.parameters())
    xs, t = batch(2, 32)
    crit(net(*xs), t).backward()
    held = [p.grad for p in params]                    # the caller keeps the tensors ...
    snap = [h.clone() for h in held]
    net.zero_grad()                                    # ... across zero_grad (set_to_none)
    xs2, t2 = batch(2, 32)
    crit(net(*xs2), t2).backward()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(held, snap)), 'a held gradient was overwritten by the next backward'
    assert all(p.grad is not h for p, h in zip(params, held))
    assert any(not torch.equal(p.grad, h) for p, h in zip(params, held))
    del held
    # ... and so does an ALIAS of one (detach / view / slice share the arena's storage; round-4 advisor finding)
    alias = [params[0].grad.detach(), params[3].grad.view(-1)[:5], params[7].grad.data]
    snap = [a.clone() for a in alias]
    net.zero_grad()
    crit(net(*xs), t).backward()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(alias, snap)), 'an alias of a held gradient was overwritten'
    del alias
    net.zero_grad()
    crit(net(*xs2), t2).backward()                     # nobody holds the views any more: the arena is reused in place
    a0 = net.engine._arena
    net.zero_grad()
    crit(net(*xs2), t2).backward()
    assert net.engine._arena is a0
    # ... and the reference's own loop (zero_grad, forward, backward, optimizer.step) holds nothing either: one arena for
    # all steps, the optimizer on its steady-state path (no pointer-table upload), no 100 MB zero fill per step
    from aide_amd.optim import Adam
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    arenas, fast_steps = set(), []
    for _ in range(5):
        opt.zero_grad()
        crit(net(*xs2), t2).backward()
        opt.step()
        arenas.add(net.engine._arena.data_ptr())
        fast_steps.append((id(opt._fast[0]), opt._fast[0]['step']) if opt._fast else None)      # (the slow path builds a new record)
    assert len(arenas) == 1 and len(set(f[0] for f in fast_steps)) == 1 and [f[1] for f in fast_steps] == [1, 2, 3, 4, 5], \
        (arenas, fast_steps)
    # (2) parameter hooks
    ref, _ = build_pair('fuseunet', False, dev)
    hk, _ = build_pair('fuseunet', False, dev)
    calls = []
    wp = list(hk.parameters())[0]
    wp.register_hook(lambda gr: (calls.append(1), gr * 2.0)[1])
    crit(ref(*xs), t).backward()
    crit(hk(*xs), t).backward()
    assert calls == [1], 'the parameter hook did not fire'
    rp = list(ref.parameters())
    assert torch.equal(wp.grad, 2.0 * rp[0].grad)
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(list(hk.parameters())[1:], rp[1:]))
    # (3) forward on a non-current stream, gradients read on the caller's stream straight after backward()
    net2, _ = build_pair('fuseunet', False, dev)
    xs3, t3 = batch(4, 256)                            # a backward pass of several ms: the host is far ahead of it
    s2 = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(2):
        net2.zero_grad()
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            out = net2(*xs3)
            loss = crit(out, t3)
        loss.backward()
        early = [p.grad.clone() for p in net2.parameters()]      # caller's stream, no explicit wait
        torch.cuda.synchronize()
        assert all(torch.equal(a, p.grad) for a, p in zip(early, net2.parameters())), \
            "the caller's stream read gradients before the backward pass had written them"


def test_two_lane_schedules_are_bit_identical(dev):
    """The second encoder's chains on their own stream in the forward pass (config.dual_fwd / free_lane) and the last weight
    gradient on the main stream (config.tail_wgrad_main) must give the single-lane
    results bit for bit -- same kernels, own workspaces -- over several steps (also a race detector for the lane's
    BatchNorm / split-K workspaces and the fork / join points), with and without launch tapes' replays."""
    from aide_amd import engine, utils as U
    net, _ = build_pair('fuseunet', False, dev)
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(4, 3, 128, 128, generator=g).to(dev) for _ in range(2)]
    t = (torch.rand(4, 128, 128, generator=g) > 0.8).long().to(dev)
    w = torch.tensor([1.0, 1.0])
    cfg = net.engine.config
    results = []
    # (free: lane 1 pools its own channels and runs from level to level without a fork / join per level)
    # tailm: the last op's weight gradient on the main stream instead of behind the weight-gradient stream's backlog
    for fwd, free, tailm in ((False, False, False), (True, False, True), (True, True, False), (True, True, True),
                             (True, False, False)):
        cfg.dual_fwd, cfg.free_lane, cfg.tail_wgrad_main = fwd, free, tailm    # (an assignment drops the engine's plans and tapes)
        outs = []
        for _ in range(3):                                 # recorded pass + two replays (BatchNorm running stats move on)
            net.zero_grad()
            out = net(*xs)
            U.CEMDiceLoss(w, w, w)(out, t).backward()
            outs.append([out.detach().clone()] + [p.grad.clone() for p in net.parameters()])
        results.append(outs[-1])
        torch.cuda.synchronize()
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize('kind', ['fuseunet', 'unet'])
def test_pool_backward_inside_batchnorm_is_bit_identical(dev, kind):
    """config.fuse_pool_bwd: the max-pooling backward of every level folded into the BatchNorm backward of the layer(s) it pooled
    (aide_bn_relu_bwd_pool) against the plain sequence (aide_maxpool2x2_bwd accumulating into the skip gradient, then the BatchNorm
    backward): every gradient bit for bit, over a recorded pass and two tape replays; the fused plan launches no pooling backward."""
    from aide_amd import utils as U
    net, _ = build_pair(kind, False, dev)
    g = torch.Generator().manual_seed(21)
    nin = 2 if kind == 'fuseunet' else 1
    xs = [torch.randn(4, 3, 128, 128, generator=g).to(dev) for _ in range(nin)]
    t = (torch.rand(4, 128, 128, generator=g) > 0.8).long().to(dev)
    w = torch.tensor([1.0, 1.0])
    nfwd = 1 if kind == 'fuseunet' else 0          # at least the 32-channel first level of the FuseUNet: one-pass BatchNorm on z as it is (no slabs, no epilogue statistics)
    results = []
    for fuse in (False, True):
        net.engine.config.fuse_pool_bwd = fuse
        net.engine.config.fuse_head_bwd = fuse              # (... and the head's data gradient inside the last BatchNorm backward)
        net.engine.config.lazy_head = fuse                  # (... and that layer's BatchNorm + ReLU inside the head's loaders)
        net.engine.config.fuse_pool_fwd = fuse              # (... and the pooling forward inside the BatchNorm forward, levels 0-1)
        outs = None
        for _ in range(3):
            net.zero_grad()
            out = net(*xs)
            U.CEMDiceLoss(w, w, w)(out, t).backward()
            outs = [out.detach().clone()] + [p.grad.clone() for p in net.parameters()]
        results.append(outs)
        plan = list(net.engine.plans.values())[-1]
        fused = [st for st in plan.steps if st['kind'] == 'pool' and st.get('bwd_fused')]
        # (128 x 128 inputs: the 16 x 16 level holds 1024 values per channel -- units of 4, which the fused form does not take)
        assert len(fused) == (3 if fuse else 0)
        assert sum(1 for st in plan.steps if st.get('head_fuse') is not None) == (1 if fuse else 0)
        nf = sum(1 for st in plan.steps if st['kind'] == 'pool' and st.get('fwd_fused'))
        assert nf == 0 if not fuse else nf >= nfwd
        torch.cuda.synchronize()
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_unet_ragged_sizes(dev):
    """UNet at 160x176 (levels 160x176 ... 10x11): exercises the PT_W = 16 / 8 tiles, widths that are
    not multiples of 4 (dword loaders), ragged wgrad tiles and the scalar BN / pool kernels."""
    from aide_amd import utils as U
    net, ref = build_pair('unet', False, dev)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 3, 160, 176, generator=g)
    t = (torch.rand(1, 160, 176, generator=g) > 0.8).long()
    w = torch.tensor([1.0, 1.0])
    net.train(); ref.train()
    out = net(x.to(dev))
    loss = U.CEMDiceLoss(w, w, w)(out, t.to(dev))
    loss.backward()
    plan = list(net.engine.plans.values())[0]
    with forced_relu_masks(net, ref, plan) as fm:
        out_r = ref(x)
        loss_r = oracle.CEMDiceLoss(w, w, w)(out_r, t)
        loss_r.backward()
    assert rel(out, out_r) < RTOL and abs(loss.item() - loss_r.item()) < RTOL * loss_r.item()
    assert sum(fm.flips.values()) <= 4 + 2e-5 * fm.total, 'implausibly many ReLU mask flips: %s' % fm.flips
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        scale = q.grad.abs().max().item()
        err = (p.grad.cpu() - q.grad).abs().max().item()
        if k.endswith('.bias') and scale < 1e-6:
            assert err < 1e-5, k
        else:
            assert err <= RTOL * scale, '%s: err %.2e scale %.2e' % (k, err, scale)


def test_large_plane_1024(dev):
    """Maximum-size edge: one 1024x1024 pair (16x the pixels of a BASELINE image; 32-bit offsets reach 0.5 GiB per operand,
    every conv runs its large-grid variant, no split-K anywhere) -- training forward + backward against the oracle."""
    from aide_amd import utils as U
    from aide_amd.synthetic import chaos_batch
    net, ref = build_pair('fuseunet', False, dev)
    xin, xout, t = chaos_batch(1, 1024, seed=31)
    w = torch.tensor([1.0, 1.0])
    net.train(); ref.train()
    out = net(xin.to(dev), xout.to(dev))
    loss = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)(out, t.to(dev))
    loss.backward()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    out_r = ref(xin, xout)
    loss_r = oracle.CEMDiceLoss(w, w, w)(out_r, t)
    loss_r.backward()
    assert rel(out, out_r) < RTOL
    assert abs(loss.item() - loss_r.item()) < 1e-4 * abs(loss_r.item())
    gn = torch.tensor([p.grad.double().norm().item() for p in net.parameters()])
    gr = torch.tensor([q.grad.double().norm().item() for q in ref.parameters()])
    names = [k for k, _ in net.named_parameters()]
    bns = {k.rsplit('.', 1)[0] for k in names if '.bn' in k or 'bilinear_up.2' in k}
    for k, a, b in zip(names, gn.tolist(), gr.tolist()):
        # conv biases that feed a BatchNorm have zero true gradient (rounding residue only, which grows with the plane)
        stem = k.rsplit('.', 1)[0]
        dead = k.endswith('.bias') and ('conv' in stem or stem.endswith('bilinear_up.1')) and k != 'last_conv1.bias'
        if dead:
            assert a < 1e-3 and b < 1e-3, (k, a, b)
        else:
            # per-parameter gradient norms (ReLU-mask flips at 1e-6 forward differences are not forced here)
            assert abs(a - b) <= 5e-3 * b + 1e-7, (k, a, b)


def test_eval_fold_tracks_parameters_and_statistics(dev):
    """Eval mode folds BatchNorm(running statistics) + ReLU into the F(4x4) conv epilogue and caches the per-layer
    coefficients across forwards (the per-case inference loop, trainchaos_comparison_1case.py:233-273).  The folded form
    must agree with the unfolded kernels and with the oracle, and must follow every way the coefficients can change: a
    training step (running statistics through raw pointers, weights through the fused Adam), an in-place edit of a buffer,
    load_state_dict -- over recorded and replayed launch tapes."""
    from aide_amd import engine, utils as U
    from aide_amd.optim import Adam
    net, ref = build_pair('fuseunet', False, dev)
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(4, 3, 128, 128, generator=g) for _ in range(2)]
    t = (torch.rand(4, 128, 128, generator=g) > 0.8).long()
    xd = [x.to(dev) for x in xs]
    w = torch.tensor([1.0, 1.0])
    opt = Adam(net.parameters(), lr=1e-3, amsgrad=True)

    def evals():
        net.eval(); ref.eval()
        with torch.no_grad():
            outs = [net(*xd).clone() for _ in range(3)]          # recorded pass + two replays (cached coefficients)
            out_r = ref(*xs)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert rel(outs[0], out_r) < 5e-3
        return outs[0]

    def unfolded():
        eng = net.engine
        keep, keep_packs = dict(eng.plans), eng._shared_packs
        try:
            eng.config.fold_eval_bn = False                        # (drops the plans: the choice is made when a plan is built)
            with torch.no_grad():
                return net(*xd).clone()
        finally:
            eng.config.fold_eval_bn = True
            eng.plans, eng._shared_packs = keep, keep_packs        # back to the plans whose cached coefficients are under test

    a0 = evals()
    plan = [p for p in net.engine.plans.values() if not p.training][0]
    assert any(st.get('fold') for st in plan.steps), 'no layer took the folded epilogue'
    assert rel(a0, unfolded()) < 1e-5
    # a training step moves weights and running statistics
    net.train()
    for _ in range(2):
        opt.zero_grad(); U.CEMDiceLoss(w, w, w)(net(*xd), t.to(dev)).backward(); opt.step()
    ref.load_state_dict(net.state_dict())        # (the oracle follows the device's trajectory: only the eval arithmetic is under test)
    a1 = evals()
    assert rel(a1, a0) > 1e-3, 'eval output did not move with the training steps (stale coefficients?)'
    assert rel(a1, unfolded()) < 1e-5
    # an in-place edit of running statistics (tensor._version)
    with torch.no_grad():
        for model in (net, ref):
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_var.mul_(1.5)
    a2 = evals()
    assert rel(a2, a1) > 1e-3
    assert rel(a2, unfolded()) < 1e-5
    # load_state_dict
    with torch.no_grad():
        for mr in ref.modules():
            if isinstance(mr, torch.nn.BatchNorm2d):
                mr.running_mean.add_(0.05); mr.weight.mul_(0.9)
    net.load_state_dict(ref.state_dict())
    a3 = evals()
    assert rel(a3, a2) > 1e-3
    assert rel(a3, unfolded()) < 1e-5
