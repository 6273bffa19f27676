"""AIDE proposed co-teaching loop on MI355X with the reference's CLI
(train_files/trainchaos_proposed_30cases1labeled.py: parse_args :21-57, inner step :260-330).

Two FuseUNets are co-trained; per step each net does 4 augmented forwards (train-mode BatchNorm, as
in the reference) + 1 training forward + 1 backward.  Pseudo-label ensemble, sharpening, weight
maps, both per-image CE+Dice vectors, the two ascending sorts, the keep/drop split and the composite
losses run as fused HIP kernels (aide_amd.utils.coteach_loss).  The PIL reverse-augmentation (:81-95) runs on
the device (`aide_reverse_aug`) from the loader-style `augset` dict; per-case evaluation and the best-checkpoint rule
(:495-526) are mirrored on synthetic cases.
"""
import argparse
import logging
import os
import random
import time

import numpy as np
import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='CHAOS segmentation, AIDE proposed (MI355X HIP engine)')
    p.add_argument('--model_name', default='fuseunet', type=str)
    p.add_argument('--data_mean', default=None, nargs='+', type=float)
    p.add_argument('--data_std', default=None, nargs='+', type=float)
    p.add_argument('--batch_size', default=4, type=int)
    p.add_argument('--gpu_order', default='1', type=str)
    p.add_argument('--torch_seed', default=2, type=int)
    p.add_argument('--lr', default=1e-4, type=float)
    p.add_argument('--num_epoch', default=100, type=int)
    p.add_argument('--loss', default='cedice', type=str)
    p.add_argument('--img_size', default=256, type=int)
    p.add_argument('--lr_policy', default='StepLR', type=str)
    p.add_argument('--cedice_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--ceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--diceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--rotation', default=60.0, type=float)
    p.add_argument('--warmup_epoch', default=20, type=int)
    p.add_argument('--temperature', default=1.0, type=float)
    p.add_argument('--segcor_weight', default=[1.0, 10.0], nargs='+', type=float)
    p.add_argument('--checkpoint', default='checkpoint_chaos_proposed30cases1label/')
    p.add_argument('--history', default='history_chaos_proposed30cases1label')
    p.add_argument('--cudnn', default=0, type=int)
    p.add_argument('--repetition', default=200, type=int)
    # not in the reference: size of the synthetic epoch (there is no dataset on this path)
    p.add_argument('--steps_per_epoch', default=8, type=int)
    return p.parse_args(argv)


# The two networks of a step are independent until the cross-scored loss: network 2's forwards (the stacked
# augmentation pass, its reverse augmentation and pseudo labels, the training forward) and its backward run on a second
# stream, so that the HBM-bound passes of one network (BatchNorm, pooling, up-sampling, reverse augmentation) overlap the
# matrix-bound convolutions of the other and the launch gaps of one chain are filled by the other.
TWO_NET_STREAMS = [True]
_NET2_STREAM = {}


def _net2_stream(device):
    s = _NET2_STREAM.get(device)
    if s is None:
        from ..engine import _preferred
        pref = _preferred(device)                     # (a data-parallel rank: a stream whose hardware queue was measured)
        s = _NET2_STREAM[device] = (pref.get('aux') if pref is not None else None) or torch.cuda.Stream(device=device)
    return s


def join_networks():
    """after a run of coteach_step(..., pipeline=True): order the caller's stream behind network 2's stream (its last
    backward pass and optimizer step) before anything else touches network 2.  Forwards of network 2 (evaluation) and its
    `state_dict()` (checkpoints) do this by themselves -- the engine orders a forward issued on another stream behind the
    pending one, a state-dict pre-hook joins -- so the explicit call is only needed before raw reads of its parameters
    (`p.data`, `.item()` on a buffer)."""
    for dev, s in _NET2_STREAM.items():
        torch.cuda.current_stream(dev).wait_stream(s)


def _guard_pipelined(net, stream):
    """network `net` keeps its backward pass / optimizer step on `stream` past the end of coteach_step: mark its engine (a
    forward on any other stream waits first) and make state_dict() join"""
    net.engine.pending_stream = stream
    if not getattr(net, '_aide_join_hook', False):
        def _join(module, prefix, keep_vars):
            s = module.engine.pending_stream
            if s is not None:
                torch.cuda.current_stream(s.device).wait_stream(s)
        net.register_state_dict_pre_hook(_join)
        net._aide_join_hook = True


def coteach_step(net1, net2, opt1, opt2, loss_op, inphase, outphase, aug_pairs, targets1, targets2, rate,
                 temperature=1.0, augset=None, pipeline=False, eval_aug=False, sharpen='pow'):
    """One step of trainchaos_proposed_30cases1labeled.py:260-325 on device tensors. `augset` (the
    loader's dict with 'augno', 'hflip{k}', 'degree{k}') triggers the on-device reverseaug (:271-272).
    The single-modal forms of the eight UNet `*_proposed_*` scripts: outphase=None, aug_pairs a list of tensors (or 1-tuples);
    kidney / breast: eval_aug=True (the nets are in eval() for the augmentation passes and back in train() for the step:
    trainkidney_proposed_mask1.py:265-266,290-291), sharpen='root' (p^(1/T), :113-117); prostate: eval_aug=False, sharpen='pow'
    (trainprostate_proposed_isbi3ttransferisbidx.py:96-100 -- no eval() in its step); the keep count is the loss
    operator's (`CoTeachingProposedLoss(keep=...)`: 2, or int(batch_size / 2) in the breast scripts,
    trainbreast_dataset3_proposed_272cases25labeled.py:304).
    pipeline: network 2's backward pass and optimizer step (:324-325) stay on network 2's stream and the call returns
    with the caller's stream free after network 1's update -- the NEXT step's network-1 forwards (:265-269, which need
    only network 1's new weights) then run beside them.  Same arithmetic, same order per network; the caller owes a
    join_networks() before it reads network 2 outside this function."""
    from aide_amd.utils import pseudo_label_ensemble, reverseaug
    if sharpen not in ('pow', 'root'):
        raise ValueError("sharpen must be 'pow' (p^T) or 'root' (p^(1/T))")
    expo = float(temperature) if sharpen == 'pow' else 1.0 / float(temperature)

    def fwd(net, x, y):
        return net(x, y) if y is not None else net(x)
    cur = torch.cuda.current_stream(inphase.device)
    two = TWO_NET_STREAMS[0]
    s2 = _net2_stream(inphase.device) if two else cur
    if two:
        s2.wait_stream(cur)                                       # the inputs (and last step's optimizer) are cur's work
    if eval_aug:
        net1.eval()
        net2.eval()
    # :265-269 -- the four augmented forwards of a network as ONE stacked pass (train mode: per-group BatchNorm statistics and
    # running-stat updates, in order: the semantics of the sequential forwards; eval mode: running statistics, nothing
    # updated; either way 4x the pixels per conv launch)
    a1 = net1.forward_groups(aug_pairs)
    with torch.cuda.stream(s2):
        a2 = net2.forward_groups(aug_pairs)
        if augset is not None:
            a2 = reverseaug(augset, a2, a2[0].shape[1])           # :271-272, no host round trip
        pl2, wm2 = pseudo_label_ensemble(a2, expo)                # :274-292
    if augset is not None:
        a1 = reverseaug(augset, a1, a1[0].shape[1])
    pl1, wm1 = pseudo_label_ensemble(a1, expo)
    if eval_aug:
        net1.train()
        net2.train()
    opt1.zero_grad()
    opt2.zero_grad()
    o1 = fwd(net1, inphase, outphase)                             # :301-302
    with torch.cuda.stream(s2):
        o2 = fwd(net2, inphase, outphase)
    if two:
        cur.wait_stream(s2)
    loss1, loss2, indx1, indx2 = loss_op(o1, o2, targets1, targets2, pl1, wm1, pl2, wm2, rate)   # :303-321
    loss1.backward()                                              # :322-325 (graphs are disjoint)
    opt1.step()
    if two and pipeline:
        with torch.cuda.stream(s2):                               # (autograd's end-of-pass join then lands on s2, not on cur)
            loss2.backward()
            opt2.step()
        _guard_pipelined(net2, s2)
    else:
        loss2.backward()                                          # (network 2's node runs on the stream of its forward)
        if two:
            cur.wait_stream(s2)
        opt2.step()
    return dict(outputs1=o1.detach(), outputs2=o2.detach(), loss1=loss1.detach(), loss2=loss2.detach(),
                indx1=indx1, indx2=indx2, extra=loss_op.last, pl1=pl1, pl2=pl2, wm1=wm1, wm2=wm2)


# The four forms of the reference's nine `*_proposed_*` scripts (oracle.steps.proposed_step; fixture g20 executes their loop
# bodies).  names: which flags name the two networks; ckpt: the best-checkpoint file names --
#   chaos    '{model}_temp{T}_r{R}_net1_besttraincasedice.pkl' / '..._net2_besttraincasedicde.pkl' (the reference's own spelling,
#            which its test scripts open: trainchaos_proposed_30cases1labeled.py:178-179, :512-513, :524-525)
#   kidney / breast  '{model k}_warmup{W}_temp{T}_r{R}_net{k}_besttraindice.pkl' (trainkidney_proposed_mask1.py:173-176, :451, :461)
#   prostate '{model}_temp{T}_r{R}_net{k}_besttraincasedice.pkl' (trainprostate_proposed_isbi3ttransferisbidx.py:173-174, :489-503)
# resume: kidney initialises BOTH networks from --resumefile (trainkidney_proposed_mask1.py:180-182).
def _ckpt_chaos(args, names, k):
    return '%s_temp%s_r%d_net%d_%s.pkl' % (names[0], args.temperature, args.repetition, k,
                                           'besttraincasedice' if k == 1 else 'besttraincasedicde')


def _ckpt_warmup(args, names, k):
    return '%s_warmup%s_temp%s_r%d_net%d_besttraindice.pkl' % (names[k - 1], args.warmup_epoch, args.temperature, args.repetition, k)


def _ckpt_prostate(args, names, k):
    return '%s_temp%s_r%d_net%d_besttraincasedice.pkl' % (names[0], args.temperature, args.repetition, k)


VARIANTS = {
    'chaos': dict(two_modal=True, eval_aug=False, sharpen='pow', keep=lambda bs: min(2, bs), ckpt=_ckpt_chaos, resume=False),
    'kidney': dict(two_modal=False, eval_aug=True, sharpen='root', keep=lambda bs: min(2, bs), ckpt=_ckpt_warmup, resume=True),
    'breast': dict(two_modal=False, eval_aug=True, sharpen='root', keep=lambda bs: int(bs / 2), ckpt=_ckpt_warmup, resume=False),
    # the prostate scripts never call eval() inside the step (their augmentation passes run in train mode and move the
    # BatchNorm statistics) and sharpen with p^T (trainprostate_proposed_isbi3ttransferisbidx.py:96-100, :253-327)
    'prostate': dict(two_modal=False, eval_aug=False, sharpen='pow', keep=lambda bs: min(2, bs), ckpt=_ckpt_prostate, resume=False),
}


def load_resumefile(path, nets):
    """trainkidney_proposed_mask1.py:180-182: `torch.load(args.resumefile)['net']` into BOTH networks before the loop.  The
    reference fails when the file is missing; the mirror (whose default path names a file only the reference's authors have)
    warns and keeps the seeded random initialisation -> True when loaded."""
    if not path:
        return False
    if not os.path.exists(path):
        logging.warning('resumefile %s not found: both networks keep their random initialisation', path)
        return False
    state = torch.load(path, map_location='cpu')['net']
    for net in nets:
        net.load_state_dict(state)
    return True


def Train(args=None, variant='chaos'):
    from aide_amd.optim import Adam
    from aide_amd.synthetic import chaos_batch
    from aide_amd.utils import CoTeachingProposedLoss
    from aide_amd.utils.poly_lr_scheduler import make_scheduler
    from aide_amd.distributed import init_from_env, attach
    from aide_amd.train_files.trainchaos_comparison_1case import build_model, evaluate_case
    args = args or parse_args()
    var = VARIANTS[variant]
    if args.loss != 'cedice':
        # the reference selects its criterion from --loss (:217-225), but its loop sorts the criterion's PER-IMAGE vector (:303-306):
        # only 'cedice' (CEMDiceLossImage) yields one -- 'ce' / 'dice' return scalars and the script fails at .sort()
        raise ValueError("--loss %s: the proposed loop needs the per-image criterion ('cedice')" % args.loss)
    if var['two_modal']:
        if args.model_name != 'fuseunet':                                # :75-78
            raise ValueError('Model not implemented')
        names = (args.model_name, args.model_name)
    else:
        # kidney / breast name the two networks separately, the prostate scripts share --model_name
        names = (args.model1_name, args.model2_name) if hasattr(args, 'model1_name') else (args.model_name, args.model_name)
        if any(nm not in ('UNet', 'UNetsa') for nm in names):            # trainkidney_proposed_mask1.py:73-80
            raise ValueError('Model not implemented')
    torch.manual_seed(args.torch_seed)
    torch.cuda.manual_seed_all(args.torch_seed)
    np.random.seed(args.torch_seed)
    random.seed(args.torch_seed)
    # reference: nn.DataParallel over --gpu_order (:183-186); here one process per GPU, rank r on gpu_order[r], per-replica
    # BatchNorm statistics and small-loss selection, both networks' gradients mean-all-reduced over RCCL
    rank, world, device = init_from_env([int(d) for d in args.gpu_order.split(',')])
    net1, net2 = build_model(names[0], 2), build_model(names[1], 2)
    if var['resume']:
        load_resumefile(getattr(args, 'resumefile', None), (net1, net2))
    net1, net2 = net1.to(device), net2.to(device)
    reducers = (attach(net1), attach(net2))          # noqa: F841
    loss_op = CoTeachingProposedLoss(cediceweight=args.cedice_weight, ceclassweight=args.ceclass_weight,
                                     segcor_weight=args.segcor_weight, keep=var['keep'](args.batch_size))
    opt1 = Adam(net1.parameters(), lr=args.lr, amsgrad=True)
    opt2 = Adam(net2.parameters(), lr=args.lr, amsgrad=True)
    sch1 = make_scheduler(args.lr_policy, opt1, args.num_epoch)        # :236-240
    sch2 = make_scheduler(args.lr_policy, opt2, args.num_epoch)
    g = torch.Generator(device='cpu').manual_seed(args.torch_seed)
    single = not var['two_modal']
    best = 0.0                                                        # :244
    for epoch in range(args.num_epoch):
        ts = time.time()
        rate = min((float(epoch) / float(args.warmup_epoch)) ** 2, 1.0)          # :248
        net1.train()
        net2.train()
        l1 = torch.zeros((), device=device)
        l2 = torch.zeros((), device=device)
        for it in range(args.steps_per_epoch):
            xin, xout, t = chaos_batch(args.batch_size, args.img_size,
                                       seed=(args.torch_seed * 100003 + epoch * 1009 + it) * world + rank)
            if single:
                augs = [(xin * (1 + 0.1 * torch.randn(1, generator=g))).to(device) for _ in range(4)]
                xin, xout, t = xin.to(device), None, t.to(device)
            else:
                augs = [((xin * (1 + 0.1 * torch.randn(1, generator=g))).to(device),
                         (xout * (1 + 0.1 * torch.randn(1, generator=g))).to(device)) for _ in range(4)]
                xin, xout, t = xin.to(device), xout.to(device), t.to(device)
            # augmentation bookkeeping as the loader's dict (:81-95): 4 augmentations per sample, random flips and rotations
            # within +-args.rotation; the logits are mapped back on the device (aide_reverse_aug)
            augset = {'augno': [4] * args.batch_size}
            for k in range(4):
                augset['hflip%d' % (k + 1)] = [int(torch.randint(0, 2, (1,), generator=g)) for _ in range(args.batch_size)]
                augset['degree%d' % (k + 1)] = [float((torch.rand(1, generator=g) * 2 - 1) * args.rotation)
                                                for _ in range(args.batch_size)]
            r = coteach_step(net1, net2, opt1, opt2, loss_op, xin, xout, augs, t, t, rate, args.temperature, augset=augset,
                             pipeline=True, eval_aug=var['eval_aug'], sharpen=var['sharpen'])
            l1 += r['loss1']
            l2 += r['loss2']
        join_networks()
        if sch1 is not None:
            sch1.step()
            sch2.step()
        # per-case evaluation of both networks and the best-checkpoint rule of :495-526 (average of the two case Dice values)
        cd1 = evaluate_case(net1, args, device, single, epoch)
        cd2 = evaluate_case(net2, args, device, single, epoch)
        if rank == 0:
            logging.info('epoch %d loss1 %.4f loss2 %.4f traincase_dice %.3f %.3f time %.1fs', epoch + 1,
                         float(l1) / args.steps_per_epoch, float(l2) / args.steps_per_epoch, cd1, cd2, time.time() - ts)
            if args.checkpoint and (cd1 + cd2) / 2.0 > best:
                best = (cd1 + cd2) / 2.0
                os.makedirs(args.checkpoint, exist_ok=True)
                for k, net in ((1, net1), (2, net2)):       # file names per variant: see VARIANTS
                    torch.save({'net': net.state_dict(), 'loss': float(l1 if k == 1 else l2) / args.steps_per_epoch,
                                'epoch': epoch + 1},
                               os.path.join(args.checkpoint, var['ckpt'](args, names, k)))
    return net1, net2


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(message)s')
    Train()
