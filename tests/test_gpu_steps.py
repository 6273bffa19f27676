"""-m gpu: the two training inner steps end to end (models + fused losses + fused Adam) against the
golden trajectories recorded from the real reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_comparison_three_steps(dev):
    """trainchaos_comparison_1case.py:195-199 x3: loss trajectory 1.3329 -> 1.2752 -> 1.2272."""
    from aide_amd import utils as U
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.optim import Adam
    fx, g1 = np.load(os.path.join(GOLD, 'g5_adam.npz')), np.load(os.path.join(GOLD, 'g1_fuseunet.npz'))
    x1, x2 = torch.from_numpy(g1['x0']).to(dev), torch.from_numpy(g1['x1']).to(dev)
    t = torch.from_numpy(g1['targets']).to(dev)
    w = torch.tensor([1.0, 1.0])
    torch.manual_seed(2)
    net = fuseunet(2).to(dev)
    net.train()
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = crit(net(x1, x2), t)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # Tolerance per step: this 2x3x32x32 fixture normalises over as few as 8 values per channel at the
    # bottom level and Adam's first steps are ~lr*sign(g), so rounding differences are amplified step by
    # step.  tests/test_oracle_golden.py::test_trajectory_noise_floor measures it on the oracle itself:
    # one-ulp (1e-7 relative) weight noise moves its losses by up to 2e-4 at step 2 and 1.5e-3 at step 3.
    rel = np.abs(np.array(losses) - fx['losses']) / fx['losses']
    assert rel[0] < 1e-5 and rel[1] < 5e-4 and rel[2] < 3e-3, (losses, fx['losses'])
    assert losses[0] > losses[1] > losses[2]
    hw = net.last_conv1.weight.detach().cpu().numpy()
    assert np.abs(hw - fx['step3/last_conv1.weight']).max() < 1e-4     # three lr=1e-4 steps


def test_proposed_coteaching_step(dev):
    """trainchaos_proposed_30cases1labeled.py:260-325 vs the reference run (g4_proposed.npz)."""
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.optim import Adam
    from aide_amd.utils import CoTeachingProposedLoss
    from aide_amd.train_files.trainchaos_proposed_30cases1labeled import coteach_step
    fx = np.load(os.path.join(GOLD, 'g4_proposed.npz'))
    T = lambda k: torch.from_numpy(fx[k]).to(dev)
    augs = [(T('aug%d_in' % i), T('aug%d_out' % i)) for i in range(4)]
    for rate in (0.0, 0.25, 1.0):
        key = 'r%g/' % rate
        torch.manual_seed(2)
        n1, n2 = fuseunet(2).to(dev), fuseunet(2).to(dev)
        n1.train(); n2.train()
        o1, o2 = Adam(n1.parameters(), lr=1e-4, amsgrad=True), Adam(n2.parameters(), lr=1e-4, amsgrad=True)
        op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
        r = coteach_step(n1, n2, o1, o2, op, T('xin'), T('xout'), augs, T('t1'), T('t2'), rate)
        pre1, pre2 = r['extra']['per_image1'].cpu().numpy(), r['extra']['per_image2'].cpu().numpy()
        assert np.abs(pre1 - fx[key + 'loss1_pre']).max() < 1e-3 * np.abs(fx[key + 'loss1_pre']).max()
        assert np.abs(pre2 - fx[key + 'loss2_pre']).max() < 1e-3 * np.abs(fx[key + 'loss2_pre']).max()
        # the sort is bit-exact whenever the reference's adjacent-loss gap exceeds fp32 noise; this
        # fixture's gaps (5.5e-4, 7.7e-4) are ~100x our per-image loss error
        err = max(np.abs(pre1 - fx[key + 'loss1_pre']).max(), np.abs(pre2 - fx[key + 'loss2_pre']).max())
        assert err * 10 < min(float(fx[key + 'min_gap1']), float(fx[key + 'min_gap2']))
        assert r['indx1'].cpu().tolist() == fx[key + 'indx1'].tolist()
        assert r['indx2'].cpu().tolist() == fx[key + 'indx2'].tolist()
        assert abs(r['loss1'].item() - float(fx[key + 'loss1'])) < 1e-3 * abs(float(fx[key + 'loss1']))
        assert abs(r['loss2'].item() - float(fx[key + 'loss2'])) < 1e-3 * abs(float(fx[key + 'loss2']))
        assert int(n1.modal1_downblock1.block.bn1.num_batches_tracked) == int(fx[key + 'nbt']) == 5
        live = fx[key + 'g1'] > 1e-5
        gn = np.array([p.grad.double().norm().item() for p in n1.parameters()])
        assert np.median(np.abs(gn[live] - fx[key + 'g1'][live]) / fx[key + 'g1'][live]) < 1e-3


@pytest.mark.parametrize('name', ['chaos', 'kidney', 'breast', 'prostate'])
def test_proposed_step_variants_g20(dev, name):
    """The co-teaching step in the four forms of the reference's nine `*_proposed_*` scripts against fixture g20, which was
    produced by EXECUTING the scripts' own loop bodies (oracle/gen_golden.py::g20_proposed_variants):
      chaos   fuseunet, bs 4, train-mode augmentation passes, p^T, keep 2     (trainchaos_proposed_30cases1labeled.py:262-330)
      kidney  UNet, bs 4, eval-mode passes, p^(1/T), keep 2                    (trainkidney_proposed_mask1.py:266-338)
      breast  UNet, bs 8, eval-mode passes, p^(1/T), keep int(bs / 2) = 4      (trainbreast_dataset3_proposed_272cases25labeled.py:258-336)
      prostate UNet, bs 4, TRAIN-mode passes (no eval() in the step), p^T, keep 2 (trainprostate_proposed_isbi3ttransferisbidx.py:253-327)
    with non-identity flips / rotations through the on-device reverse augmentation.  Index vectors bit-exact (the recorded
    adjacent-loss gaps exceed the per-image loss error tenfold), losses / pseudo labels / weight maps / gradient norms <= 1e-3."""
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNet
    from aide_amd.optim import Adam
    from aide_amd.utils import CoTeachingProposedLoss
    from aide_amd.train_files.trainchaos_proposed_30cases1labeled import coteach_step, join_networks, VARIANTS
    from test_oracle_golden import g20_case, sub
    fx = np.load(os.path.join(GOLD, 'g20_proposed_variants.npz'))
    c = g20_case(fx, name)
    key = name + '/'
    var = VARIANTS[name]
    assert var['two_modal'] == c['two_modal'] and var['eval_aug'] == c['eval_aug'] and var['keep'](c['n']) == c['keep']
    D = lambda t: t.to(dev)
    torch.manual_seed(2)
    ctor = fuseunet if c['two_modal'] else UNet
    n1, n2 = ctor(2).to(dev), ctor(2).to(dev)
    n1.train(); n2.train()
    o1, o2 = Adam(n1.parameters(), lr=1e-4, amsgrad=True), Adam(n2.parameters(), lr=1e-4, amsgrad=True)
    op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0], keep=c['keep'])
    augs = [tuple(D(a) for a in pair) for pair in c['augs']]
    r = coteach_step(n1, n2, o1, o2, op, D(c['xs'][0]), D(c['xs'][1]) if c['two_modal'] else None, augs, D(c['t1']), D(c['t2']),
                     c['rate'], temperature=c['temp'], augset=c['augset'], eval_aug=var['eval_aug'], sharpen=var['sharpen'])
    join_networks()
    assert n1.training and n2.training
    pre1, pre2 = r['extra']['per_image1'].cpu().numpy(), r['extra']['per_image2'].cpu().numpy()
    err = max(np.abs(pre1 - fx[key + 'loss1_pre']).max(), np.abs(pre2 - fx[key + 'loss2_pre']).max())
    assert err < 1e-3 * np.abs(fx[key + 'loss1_pre']).max()
    assert err * 10 < min(float(fx[key + 'min_gap1']), float(fx[key + 'min_gap2'])), (err, fx[key + 'min_gap1'], fx[key + 'min_gap2'])
    assert r['indx1'].cpu().tolist() == fx[key + 'indx1'].tolist()              # bit-exact selection, bs 8 / keep 4 included
    assert r['indx2'].cpu().tolist() == fx[key + 'indx2'].tolist()
    for k in ('loss1', 'loss2'):
        assert abs(r[k].item() - float(fx[key + k])) < 1e-3 * abs(float(fx[key + k])), k
    for k in ('pl1', 'pl2', 'wm1', 'wm2'):
        assert np.abs(sub(r[k].cpu().numpy(), 2048) - fx[key + k]).max() < 1e-3, k
    bn = [m for m in n1.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
    assert int(bn.num_batches_tracked) == int(fx[key + 'nbt']) == (5 if name in ('chaos', 'prostate') else 1)
    assert np.abs(bn.running_mean.cpu().numpy() - fx[key + 'rm']).max() < 1e-5
    for net, gk in ((n1, 'g1'), (n2, 'g2')):
        live = fx[key + gk] > 1e-5
        gn = np.array([p.grad.double().norm().item() for p in net.parameters()])
        assert np.median(np.abs(gn[live] - fx[key + gk][live]) / fx[key + gk][live]) < 1e-3, gk
    assert np.abs(n1.last_conv1.weight.detach().cpu().numpy() - fx[key + 'head1']).max() < 1e-5


def test_grouped_forward_in_eval_mode(dev):
    """forward_groups under net.eval() (the augmentation passes of the UNet co-teaching scripts, trainkidney_proposed_mask1.py:
    265-272): the stacked pass uses the running statistics, updates nothing and equals the sequential eval-mode forwards."""
    from aide_amd.models_singlemodalinput import UNet
    torch.manual_seed(2)
    net = UNet(2).to(dev)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, 3, 64, 64, generator=g).to(dev) for _ in range(4)]
    net.train()
    with torch.no_grad():
        net(xs[0])                                 # running statistics that are not the initial (0, 1)
    net.eval()
    bufs = [b.clone() for b in net.buffers()]
    with torch.no_grad():
        seq = [net(x).clone() for x in xs]
    grp = net.forward_groups(xs)
    for a, b in zip(seq, grp):
        assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()
    assert all(torch.equal(a, b) for a, b in zip(bufs, net.buffers()))


def test_coteaching_two_streams_is_bit_identical(dev):
    """Network 2 on its own stream (TWO_NET_STREAMS, the default) and packed filters shared between a network's plans
    (config.shared_packs, the default) are schedules, not different computations: three steps from the same initial state give
    bit-identical losses, selections and parameters of BOTH networks as the single-stream order with private packs -- any
    missing cross-stream dependency (inputs, optimizer update, engine-assigned gradients) or a stale shared pack would show
    here."""
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.optim import Adam
    from aide_amd.utils import CoTeachingProposedLoss
    from aide_amd.train_files import trainchaos_proposed_30cases1labeled as M
    fx = np.load(os.path.join(GOLD, 'g4_proposed.npz'))
    T = lambda k: torch.from_numpy(fx[k]).to(dev)
    augs = [(T('aug%d_in' % i), T('aug%d_out' % i)) for i in range(4)]
    n = T('xin').shape[0]
    augset = {'augno': [4] * n}
    for k in range(4):
        augset['hflip%d' % (k + 1)] = [(k + b) % 2 for b in range(n)]
        augset['degree%d' % (k + 1)] = [15.0 * (k + 1) - 7.0 * b for b in range(n)]
    res = {}
    old = M.TWO_NET_STREAMS[0]
    try:
        for two in (False, True, 'pipelined'):
            M.TWO_NET_STREAMS[0] = bool(two)
            torch.manual_seed(2)
            n1, n2 = fuseunet(2).to(dev), fuseunet(2).to(dev)
            n1.engine.config.shared_packs = n2.engine.config.shared_packs = bool(two)   # ... and with every plan packing its own filters (both directions) vs shared packs
            n1.train(); n2.train()
            o1, o2 = Adam(n1.parameters(), lr=1e-3, amsgrad=True), Adam(n2.parameters(), lr=1e-3, amsgrad=True)
            op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
            trace = []
            for _ in range(3):
                # 'pipelined': network 2's backward pass and optimizer step stay on its stream, the next step's network-1
                # forwards run beside them (coteach_step(pipeline=True), what the train script and bench.py use)
                r = M.coteach_step(n1, n2, o1, o2, op, T('xin'), T('xout'), augs, T('t1'), T('t2'), 0.25, augset=augset,
                                   pipeline=(two == 'pipelined'))
                trace.append((r['loss1'].clone(), r['loss2'].clone(), r['indx1'].clone(), r['indx2'].clone()))
                # the batch of the step is dropped here while (pipelined) network 2's backward pass may still read it on its own
                # stream: what the caller allocates next on the main stream -- the next batch -- must not land in that memory
                # (round 6: the engine record_stream()s its inputs; without it both stems' weight gradients were corrupted)
                junk = [torch.full_like(T('xin'), float('nan')) for _ in range(14)]
                del junk
            M.join_networks()
            torch.cuda.synchronize()
            res[two] = (trace, [p.detach().clone() for p in list(n1.parameters()) + list(n2.parameters())],
                        [b.detach().clone() for b in list(n1.buffers()) + list(n2.buffers())])
    finally:
        M.TWO_NET_STREAMS[0] = old
    for other in (True, 'pipelined'):
        for a, b in zip(res[False][0], res[other][0]):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
        assert all(torch.equal(x, y) for x, y in zip(res[False][1], res[other][1]))
        assert all(torch.equal(x, y) for x, y in zip(res[False][2], res[other][2]))


def test_pipelined_network2_reads_join_by_themselves(dev):
    """coteach_step(pipeline=True) returns with network 2's backward pass and optimizer step still running on its own stream
    (round-4 advisor finding: every caller had to remember join_networks()).  A forward of network 2 on the caller's stream
    and its state_dict() now order themselves behind that stream: what they see equals what a full device synchronisation
    shows afterwards."""
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.optim import Adam
    from aide_amd.utils import CoTeachingProposedLoss
    from aide_amd.train_files import trainchaos_proposed_30cases1labeled as M
    fx = np.load(os.path.join(GOLD, 'g4_proposed.npz'))
    T = lambda k: torch.from_numpy(fx[k]).to(dev)
    augs = [(T('aug%d_in' % i), T('aug%d_out' % i)) for i in range(4)]
    torch.manual_seed(2)
    n1, n2 = fuseunet(2).to(dev), fuseunet(2).to(dev)
    n1.train(); n2.train()
    o1, o2 = Adam(n1.parameters(), lr=1e-3, amsgrad=True), Adam(n2.parameters(), lr=1e-3, amsgrad=True)
    op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
    for _ in range(3):
        M.coteach_step(n1, n2, o1, o2, op, T('xin'), T('xout'), augs, T('t1'), T('t2'), 0.25, pipeline=True)
    assert n2.engine.pending_stream is not None
    early = {k: v.clone() for k, v in n2.state_dict().items()}          # no join_networks(), no synchronize
    n2.eval()
    with torch.no_grad():
        out_early = n2(T('xin'), T('xout')).clone()                       # a forward on the caller's stream
    torch.cuda.synchronize()
    late = n2.state_dict()
    assert all(torch.equal(early[k], late[k]) for k in late), 'state_dict() read network 2 before its optimizer step had finished'
    with torch.no_grad():
        assert torch.equal(out_early, n2(T('xin'), T('xout')))
    assert n2.engine.pending_stream is None


def test_cli_smoke(dev, tmp_path):
    """The restated CLI (--model_name / --batch_size as in README.md:32) trains, the loss falls, every epoch evaluates a
    case and the best checkpoint is written in the reference's format ({'net': state_dict, ...}, :329-345) and loads back."""
    from aide_amd.train_files.trainchaos_comparison_1case import parse_args, Train, build_model
    args = parse_args(['--model_name', 'fuseunet', '--batch_size', '2', '--img_size', '64', '--num_epoch', '3',
                       '--steps_per_epoch', '4', '--checkpoint', str(tmp_path / 'ck')])
    assert args.lr == 1e-4 and args.torch_seed == 2 and args.loss == 'cedice'
    net, hist = Train(args)
    assert hist['train_loss'][-1] < hist['train_loss'][0]
    assert len(hist['traincase_dice']) == 3 and all(0.0 <= d <= 1.0 for d in hist['traincase_dice'])
    files = os.listdir(str(tmp_path / 'ck')) if os.path.isdir(str(tmp_path / 'ck')) else []
    if max(hist['traincase_dice']) <= 0.0:         # best starts at 0.0 as in the reference (:188): nothing to save
        assert files == []
        return
    assert files == ['fuseunet_r%d_besttraincasedice.pkl' % args.repetition]          # the reference's name (:125, :343-344)
    state = torch.load(os.path.join(str(tmp_path / 'ck'), files[0]), map_location='cpu', weights_only=False)
    assert set(state) >= {'net', 'loss', 'dice', 'epoch', 'history'}
    build_model('fuseunet', 2).load_state_dict(state['net'])
    with pytest.raises(ValueError, match='Model not implemented'):
        build_model('resnet', 2)


def test_cli_smoke_default_size(dev):
    """BASELINE config 1 literally: `--model_name fuseunet --batch_size 4` at the script's default 256 x 256
    (trainchaos_comparison_1case.py:21-49,186-202), three steps through the CLI on the device; the loss trajectory against the
    oracle's comparison_step (the restated :195-199 on aten CPU ops) run on the SAME chaos_batch seeds the CLI draws."""
    import oracle
    from oracle import steps
    from aide_amd.synthetic import chaos_batch
    from aide_amd.train_files.trainchaos_comparison_1case import parse_args, Train
    args = parse_args(['--model_name', 'fuseunet', '--batch_size', '4', '--img_size', '256', '--num_epoch', '1',
                       '--steps_per_epoch', '3', '--checkpoint', ''])
    assert args.img_size == parse_args([]).img_size == 256 and args.batch_size == parse_args([]).batch_size == 4
    net, hist = Train(args)
    got = hist['step_loss']
    assert len(got) == 3
    torch.manual_seed(args.torch_seed)
    ref_net = oracle.fuseunet(2)
    ref_net.train()
    w = torch.tensor([1.0, 1.0])
    crit = oracle.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = torch.optim.Adam(ref_net.parameters(), lr=args.lr, amsgrad=True)
    want = []
    for it in range(3):
        xin, xout, t = chaos_batch(4, 256, seed=args.torch_seed * 100003 + it)
        _, loss = steps.comparison_step(ref_net, crit, opt, xin, xout, t)
        want.append(float(loss))
    rel = np.abs(np.array(got) - np.array(want)) / np.array(want)
    # step 1 sees identical weights; later steps inherit Adam's lr * sign(g) amplification of rounding differences (see
    # test_comparison_three_steps and tests/test_oracle_golden.py::test_trajectory_noise_floor)
    assert rel[0] < 1e-4 and rel[1] < 1e-3 and rel[2] < 3e-3, (got, want)


def test_cli_unet_proposed_smoke(dev, tmp_path):
    """The single-modal co-teaching CLIs (flag tables of trainkidney_proposed_mask1.py:28-60,
    trainbreast_dataset3_proposed_272cases25labeled.py and trainprostate_proposed_isbi3ttransferisbidx.py:26-58): eval-mode
    augmentation passes + p^(1/T) (kidney, breast) or train-mode passes + p^T (prostate), keep 2 resp. int(batch_size / 2); the
    kidney script initialises BOTH networks from --resumefile (:180-182); best checkpoints under the reference's file names
    (kidney / breast :173-176, :451, :461; prostate :173-174, :489-503)."""
    import oracle
    from aide_amd.train_files import (trainkidney_proposed_mask1 as K, trainbreast_dataset3_proposed_272cases25labeled as B,
                                      trainprostate_proposed_isbi3ttransferisbidx as P)
    torch.manual_seed(11)
    init = oracle.UNet(2).state_dict()                       # a checkpoint as the reference's UNet writes it
    resume = str(tmp_path / 'init.pkl')
    torch.save({'net': init, 'loss': 0.0, 'epoch': 1}, resume)
    for mod, bs in ((K, 2), (B, 4), (P, 2)):
        ck = tmp_path / mod.__name__.split('.')[-1]
        args = mod.parse_args(['--batch_size', str(bs), '--img_size', '64', '--num_epoch', '2', '--steps_per_epoch', '2', '--lr', '0.0',
                               '--warmup_epoch', '2', '--temperature', '0.5', '--checkpoint', str(ck)] +
                              (['--resumefile', resume] if mod is K else []))
        n1, n2 = mod.Train(args)
        assert all(torch.isfinite(p).all() for p in list(n1.parameters()) + list(n2.parameters()))
        assert n1.training and n2.training
        if mod is K:                                         # lr 0: both networks still hold the resumed weights
            for net in (n1, n2):
                sd = net.state_dict()
                assert all(torch.equal(sd[k].cpu(), init[k]) for k in init if 'running_' not in k and 'num_batches' not in k)
        files = sorted(os.listdir(str(ck))) if os.path.isdir(str(ck)) else []
        if mod is P:
            want = ['UNet_temp0.5_r100_net1_besttraincasedice.pkl', 'UNet_temp0.5_r100_net2_besttraincasedice.pkl']
        else:
            want = ['UNet_warmup2_temp0.5_r%d_net%d_besttraindice.pkl' % (args.repetition, k) for k in (1, 2)]
        assert files in ([], want), files                   # (best starts at 0.0: a case Dice of 0 writes nothing)
    with pytest.raises(ValueError, match='Model not implemented'):
        K.Train(K.parse_args(['--model1_name', 'fuseunet']))
    with pytest.raises(ValueError, match='Model not implemented'):
        P.Train(P.parse_args(['--model_name', 'fuseunet']))


def test_cli_proposed_smoke(dev, tmp_path):
    """The proposed co-teaching CLI: two epochs with the loader-style augmentation dict (on-device reverse augmentation
    with random flips / rotations), per-case evaluation of both networks, both checkpoints written (:495-526)."""
    from aide_amd.train_files.trainchaos_proposed_30cases1labeled import parse_args, Train
    args = parse_args(['--batch_size', '2', '--img_size', '64', '--num_epoch', '2', '--steps_per_epoch', '2',
                       '--warmup_epoch', '2', '--checkpoint', str(tmp_path / 'ck')])
    n1, n2 = Train(args)
    files = sorted(os.listdir(str(tmp_path / 'ck'))) if os.path.isdir(str(tmp_path / 'ck')) else []
    # the reference's names (:178-179, :512-513, :524-525), its spelling of the second one included; best starts at 0.0 (:244)
    stem = 'fuseunet_temp%s_r%d' % (args.temperature, args.repetition)
    assert files in ([], [stem + '_net1_besttraincasedice.pkl', stem + '_net2_besttraincasedicde.pkl'])
    assert all(torch.isfinite(p).all() for p in n1.parameters())


def test_rccl_gradient_allreduce_single_rank(dev):
    """The bucketed RCCL all-reduce path of bench.py --gpus N, exercised on a 1-rank NCCL(=RCCL) group:
    side streams, events and ReduceOp.AVG must run and leave the gradients bit-identical."""
    import os
    import torch.distributed as dist
    from aide_amd import utils as U
    from aide_amd.distributed import GradAllReduce, broadcast_module
    from aide_amd.models_twomodalinputs import fuseunet
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    # (as init_from_env does it)
    from aide_amd import streams
    streams.reserve_queue(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(2)
        net = fuseunet(2).to(dev)
        net.train()
        broadcast_module(net)
        g = torch.Generator().manual_seed(1)
        x1, x2 = torch.randn(2, 3, 64, 64, generator=g).to(dev), torch.randn(2, 3, 64, 64, generator=g).to(dev)
        t = (torch.rand(2, 64, 64, generator=g) > 0.8).long().to(dev)
        w = torch.tensor([1.0, 1.0])
        U.CEMDiceLoss(w, w, w)(net(x1, x2), t).backward()
        ref = [p.grad.clone() for p in net.parameters()]
        net.zero_grad()
        red = GradAllReduce(net, bucket_mb=8.0, force=True)
        U.CEMDiceLoss(w, w, w)(net(x1, x2), t).backward()
        torch.cuda.synchronize()
        assert len(red.sched.buckets) >= 5 and all(p == 0 for p in red.sched.pending)
        for a, p in zip(ref, net.parameters()):
            assert torch.equal(a, p.grad)
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_survives_plan_eviction(dev):
    """Plans are dropped and rebuilt under the reducer (MAX_PLANS eviction with two alternating input shapes): every
    rebuilt plan's ops must map to their own parameters in the bucket schedule (the per-op index cache lives on the step
    dict, not in a table keyed on id(step)) -- gradients stay bit-identical to a replica without the reducer."""
    import copy
    import os
    import torch.distributed as dist
    from aide_amd import utils as U
    from aide_amd.distributed import GradAllReduce
    from aide_amd.models_twomodalinputs import fuseunet
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    from aide_amd import streams
    streams.reserve_queue(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(2)
        net = fuseunet(2).to(dev)
        ref = copy.deepcopy(net)
        net.train(); ref.train()
        net.engine.MAX_PLANS = 1
        red = GradAllReduce(net, bucket_mb=8.0, force=True)
        w = torch.tensor([1.0, 1.0])
        g = torch.Generator().manual_seed(3)
        for it in range(3):
            for size in (64, 32):
                x1 = torch.randn(2, 3, size, size, generator=g).to(dev)
                x2 = torch.randn(2, 3, size, size, generator=g).to(dev)
                t = (torch.rand(2, size, size, generator=g) > 0.8).long().to(dev)
                net.zero_grad(); ref.zero_grad()
                U.CEMDiceLoss(w, w, w)(net(x1, x2), t).backward()
                U.CEMDiceLoss(w, w, w)(ref(x1, x2), t).backward()
                torch.cuda.synchronize()
                assert len(net.engine.plans) == 1 and all(p == 0 for p in red.sched.pending)
                for (k, a), b in zip(net.named_parameters(), ref.parameters()):
                    assert torch.equal(a.grad, b.grad), (it, size, k)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['fuseunet', 'unetsa'])
def test_grouped_forward_equals_sequential(dev, kind):
    """net.forward_groups([...]) (one stacked pass, per-group BatchNorm statistics) == the sequential train-mode forwards
    of the augmentation loop (trainchaos_proposed_30cases1labeled.py:265-269): outputs, running statistics,
    num_batches_tracked.  (Convolutions over 4x the pixels may pick another split-K plan: 2e-5, not bitwise.)"""
    import copy
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNetsa
    torch.manual_seed(2)
    net = (fuseunet(2) if kind == 'fuseunet' else UNetsa(2)).to(dev)
    ref = copy.deepcopy(net)
    g = torch.Generator().manual_seed(5)
    nin = 2 if kind == 'fuseunet' else 1
    groups = [tuple(torch.randn(2, 3, 32, 48, generator=g).to(dev) for _ in range(nin)) for _ in range(4)]
    net.train(); ref.train()
    seq = [ref(*grp).detach() for grp in groups]
    outs = net.forward_groups(groups)
    assert len(outs) == 4
    for a, b in zip(outs, seq):
        assert not a.requires_grad
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    for (k, p), (_, q) in zip(net.named_buffers(), ref.named_buffers()):
        if 'num_batches_tracked' in k:
            assert int(p) == int(q) == 4
        else:
            assert (p - q).abs().max().item() <= 2e-5 * (q.abs().max().item() + 1e-6), k
    with pytest.raises(RuntimeError):
        net.forward_groups([groups[0], tuple(t[:1] for t in groups[1])])


@pytest.mark.parametrize('kind', ['fuseunet', 'unetsa'])
def test_grouped_batchnorm_launches_equal_per_group_launches(dev, kind, monkeypatch):
    """The stacked pass normalises all groups of a layer in one launch sequence (config.grouped_bn,
    aide_bn_train_fwd_groups: plain / split-K-slab / epilogue-statistics input, the small-plane single-kernel form and
    the two-pass form) == one launch sequence per group: outputs, running statistics (updated in group order),
    num_batches_tracked.  (The two-pass form sums its partials in fewer splits: 1e-6, not bitwise.)"""
    import copy
    from aide_amd import engine, ops
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNetsa
    torch.manual_seed(2)
    net = (fuseunet(2) if kind == 'fuseunet' else UNetsa(2)).to(dev)
    ref = copy.deepcopy(net)
    g = torch.Generator().manual_seed(6)
    nin = 2 if kind == 'fuseunet' else 1
    groups = [tuple(torch.randn(2, 3, 128, 160, generator=g).to(dev) for _ in range(nin)) for _ in range(4)]
    net.train(); ref.train()
    calls = []
    real = ops.bn_train_fwd_groups
    monkeypatch.setattr(ops, 'bn_train_fwd_groups', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    net.engine.config.lazy_bn = ref.engine.config.lazy_bn = False
    net.engine.config.grouped_bn = True
    outs = net.forward_groups(groups) + net.forward_groups(groups)
    assert len(calls) >= 20, len(calls)               # (the second pass replays the launch tape)
    ref.engine.config.grouped_bn = False
    n0 = len(calls)
    refs = ref.forward_groups(groups) + ref.forward_groups(groups)
    assert len(calls) == n0
    torch.cuda.synchronize()
    for a, b in zip(outs, refs):
        assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item()
    for (k, p), (_, q) in zip(net.named_buffers(), ref.named_buffers()):
        if 'num_batches_tracked' in k:
            assert int(p) == int(q) == 8
        else:
            assert (p - q).abs().max().item() <= 1e-6 * (q.abs().max().item() + 1e-12), k


@pytest.mark.parametrize('kind,size', [('fuseunet', 256), ('unet', 320)])
def test_lazy_batchnorm_in_the_reader_is_bit_identical(dev, kind, size):
    """Forward-only stacked passes apply the BatchNorm + ReLU of a layer whose only reader is an F(4x4) convolution in that
    convolution's loader (config.lazy_bn; conv3x3_wino4 in_bn_tab) instead of a pass over the tensor: same arithmetic
    element for element (fmaf, max), so outputs, running statistics and num_batches_tracked equal the materialised form
    bit for bit -- also for border tiles (zero padding must stay zero after the affine map) and the 20 x 20 canvas tiles."""
    import copy
    from aide_amd import engine
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNet
    torch.manual_seed(2)
    net = (fuseunet(2) if kind == 'fuseunet' else UNet(2)).to(dev)
    with torch.no_grad():                              # non-trivial affine parameters, negative scales included
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.randn(m.weight.shape, device=dev) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, device=dev) * 0.3)
    ref = copy.deepcopy(net)
    g = torch.Generator().manual_seed(9)
    nin = 2 if kind == 'fuseunet' else 1
    groups = [tuple(torch.randn(2, 3, size, size, generator=g).to(dev) for _ in range(nin)) for _ in range(4)]
    net.train(); ref.train()
    net.engine.config.lazy_bn = True
    outs = net.forward_groups(groups)
    plan = [p for p in net.engine.plans.values() if p.groups == 4][0]
    assert plan.lazy_bn >= 4, plan.lazy_bn
    outs2 = net.forward_groups(groups)             # replayed tape, running statistics move on
    ref.engine.config.lazy_bn = False
    refs = ref.forward_groups(groups)
    assert [p for p in ref.engine.plans.values() if p.groups == 4][0].lazy_bn == 0
    refs2 = ref.forward_groups(groups)
    torch.cuda.synchronize()
    for a, b in zip(outs + outs2, refs + refs2):
        assert torch.equal(a, b)
    for (k, p), (_, q) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.equal(p, q), k


def test_rank_streams_avoid_shared_hardware_queues(dev):
    """A data-parallel rank measures which streams share a hardware queue (aide_amd/streams.py) before its engine builds a
    plan: the weight-gradient stream it hands the engine runs beside the main stream AND beside RCCL's own stream, the lane
    stream beside the two compute streams; the step on those streams gives the gradients of the plain step bit for bit."""
    import os
    import torch.distributed as dist
    from aide_amd import utils as U, streams
    from aide_amd.distributed import GradAllReduce, broadcast_module
    from aide_amd.models_twomodalinputs import fuseunet
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    g = torch.Generator().manual_seed(1)
    x1, x2 = torch.randn(2, 3, 64, 64, generator=g).to(dev), torch.randn(2, 3, 64, 64, generator=g).to(dev)
    t = (torch.rand(2, 64, 64, generator=g) > 0.8).long().to(dev)
    w = torch.tensor([1.0, 1.0])
    torch.manual_seed(2)
    plain = fuseunet(2).to(dev)
    plain.train()
    U.CEMDiceLoss(w, w, w)(plain(x1, x2), t).backward()
    ref = [p.grad.clone() for p in plain.parameters()]
    saved = dict(streams.PREFERRED)
    streams.PREFERRED.clear()
    streams.reserve_queue(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(2)
        net = fuseunet(2).to(dev)
        net.train()
        broadcast_module(net)
        red = GradAllReduce(net, bucket_mb=8.0, force=True)          # no plan yet: the rank picks its streams here
        sp = red.stream_plan
        assert sp is not None and len(sp['classes']) >= 3, sp
        assert sp['side'] is not None and sp['side'] != sp['main_class'] and sp['side'] not in (sp['rccl_class'] or [])
        assert sp['lane'] is not None and sp['lane'] not in (sp['main_class'], sp['side'])
        pref = streams.PREFERRED[dev.index if dev.index is not None else torch.cuda.current_device()]
        for _ in range(2):                                            # recorded pass + a replay
            net.zero_grad()
            U.CEMDiceLoss(w, w, w)(net(x1, x2), t).backward()
        torch.cuda.synchronize()
        plan = list(net.engine.plans.values())[0]
        assert plan.side is pref['side'] and plan.lane_b is pref['lane']
        assert red.describe()['hw_queues']['side'] == sp['side']
        for a, p in zip(ref, net.parameters()):
            assert torch.equal(a, p.grad)
    finally:
        dist.destroy_process_group()
        streams.PREFERRED.clear()
        streams.PREFERRED.update(saved)
