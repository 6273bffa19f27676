"""Comparison training loop on MI355X with the reference's CLI.

Mirrors train_files/trainchaos_comparison_1case.py: the flag names and defaults of parse_args()
(:21-49), build_model() and its ValueError (:61-66), seeding (:108-112), criterion selection
(:157-168), Adam(amsgrad)+StepLR (:170-176) and the inner step (:190-202).  Dataset I/O, per-case
evaluation and checkpointing are mirrored on synthetic cases (the dataset itself is out of scope): batches
come from the synthetic CHAOS-shaped generator, and unlike the reference nothing runs at import time.

    python -m aide_amd.train_files.trainchaos_comparison_1case --model_name fuseunet --batch_size 4
"""
import argparse
import logging
import os
import random
import time

import numpy as np
import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='CHAOS segmentation, comparison model (MI355X HIP engine)')
    p.add_argument('--model_name', default='fuseunet', type=str, help='fuseunet')
    p.add_argument('--data_mean', default=None, nargs='+', type=float)
    p.add_argument('--data_std', default=None, nargs='+', type=float)
    p.add_argument('--batch_size', default=4, type=int)
    p.add_argument('--gpu_order', default='0', type=str)
    p.add_argument('--torch_seed', default=2, type=int)
    p.add_argument('--lr', default=1e-4, type=float)
    p.add_argument('--num_epoch', default=100, type=int)
    p.add_argument('--loss', default='cedice', type=str, help='ce, dice, cedice')
    p.add_argument('--img_size', default=256, type=int)
    p.add_argument('--lr_policy', default='StepLR', type=str)
    p.add_argument('--cedice_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--ceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--diceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--checkpoint', default='checkpoint_chaos_comparison1case/')
    p.add_argument('--history', default='history_chaos_comparison1case')
    p.add_argument('--cudnn', default=0, type=int)
    p.add_argument('--repetition', default=2, type=int)
    # not in the reference: size of the synthetic epoch (there is no dataset on this path)
    p.add_argument('--steps_per_epoch', default=16, type=int)
    return p.parse_args(argv)


def build_model(model_name, num_classes):
    """'fuseunet' (trainchaos_*.py:61-65); the single-modality scripts accept 'UNet' / 'UNetsa'
    (trainkidney_comparison_mask1.py:62-68); the other model classes of the two model files by their class names."""
    import aide_amd.models_twomodalinputs as two
    import aide_amd.models_singlemodalinput as one
    for mod in (two, one):
        ctor = getattr(mod, model_name, None)
        if isinstance(ctor, type):
            return ctor(num_classes=num_classes)
    raise ValueError('Model not implemented')


def Train(args=None):
    from aide_amd import utils as U
    from aide_amd.optim import Adam
    from aide_amd.synthetic import chaos_batch
    from aide_amd.utils.poly_lr_scheduler import make_scheduler
    from aide_amd.distributed import init_from_env, attach
    args = args or parse_args()
    torch.manual_seed(args.torch_seed)
    torch.cuda.manual_seed_all(args.torch_seed)
    np.random.seed(args.torch_seed)
    random.seed(args.torch_seed)
    # the reference wraps the net in nn.DataParallel over --gpu_order (:131-134); here: one process per GPU
    # (python -m torch.distributed.run --nproc-per-node N -m aide_amd.train_files...), rank r on gpu_order[r]
    rank, world, device = init_from_env([int(g) for g in args.gpu_order.split(',')])
    num_classes = 2
    net = build_model(args.model_name, num_classes).to(device)
    reducer = attach(net)          # noqa: F841  (bucketed RCCL gradient mean all-reduce, installed on the engine)
    cedice_weight = torch.tensor(args.cedice_weight)
    ceclass_weight = torch.tensor(args.ceclass_weight)
    diceclass_weight = torch.tensor(args.diceclass_weight)
    if args.loss == 'ce':
        criterion = U.CrossEntropyLoss2d(weight=ceclass_weight)
    elif args.loss == 'dice':
        criterion = U.MulticlassDiceLoss(weight=diceclass_weight)
    elif args.loss == 'cedice':
        criterion = U.CEMDiceLoss(cediceweight=cedice_weight, ceclassweight=ceclass_weight,
                                  diceclassweight=diceclass_weight)
    else:
        raise ValueError('Do not have this loss')
    optimizer = Adam(net.parameters(), lr=args.lr, amsgrad=True)
    scheduler = make_scheduler(args.lr_policy, optimizer, args.num_epoch)
    single = not args.model_name.startswith('fuseunet')
    history = {'train_loss': [], 'train_dice': [], 'step_loss': []}
    best_casedice = 0.0                      # :188 (a case Dice of 0 never saves, as in the reference)
    for epoch in range(args.num_epoch):
        ts = time.time()
        net.train()
        loss_sum = torch.zeros((), device=device)
        dice_sum = torch.zeros((), device=device)
        count = 0
        step_losses = []                                      # device scalars, read once per epoch
        for it in range(args.steps_per_epoch):
            inphase, outphase, targets = chaos_batch(args.batch_size, args.img_size,
                                                     seed=(args.torch_seed * 100003 + epoch * 1009 + it) * world + rank,
                                                     single_modal=single)
            inphase, targets = inphase.to(device), targets.to(device)
            optimizer.zero_grad()
            outputs = net(inphase) if single else net(inphase, outphase.to(device))
            loss = criterion(outputs, targets)
            loss.backward()
            optimizer.step()
            count += inphase.shape[0]
            step_losses.append(loss.detach())
            loss_sum += loss.detach() * inphase.shape[0]      # device-side accumulation: one host
            dice_sum += U.Dice_fn(outputs, targets)           # sync per epoch instead of two per step
        if scheduler is not None:
            scheduler.step()
        history['train_loss'].append(float(loss_sum) / count)
        history['step_loss'] += [float(v) for v in torch.stack(step_losses).cpu()]
        history['train_dice'].append(float(dice_sum) / count)
        # per-case evaluation (reference :232-315: every slice of a case through the eval-mode net, 3-D Dice of the label
        # volume) on a synthetic case, and the best-checkpoint rule of :329-345 ({'net': state_dict, ...})
        casedice = evaluate_case(net, args, device, single, epoch)
        history.setdefault('traincase_dice', []).append(casedice)
        if rank == 0:
            logging.info('epoch %d train_loss %.4f train_dice %.4f traincase_dice %.3f time %.1fs', epoch + 1,
                         history['train_loss'][-1], history['train_dice'][-1], casedice, time.time() - ts)
            if args.checkpoint and casedice > best_casedice:
                best_casedice = casedice
                os.makedirs(args.checkpoint, exist_ok=True)
                # file name of :125, :343-344 ('{model}_r{rep}.pkl' -> '{model}_r{rep}_besttraincasedice.pkl'): what the
                # reference's evaluation scripts look for
                name = '%s_r%d_besttraincasedice.pkl' % (args.model_name, args.repetition)
                torch.save({'net': net.state_dict(), 'loss': history['train_loss'][-1], 'dice': history['train_dice'][-1],
                            'epoch': epoch + 1, 'history': history}, os.path.join(args.checkpoint, name))
    return net, history


def evaluate_case(net, args, device, single, epoch, slices=8):
    """3-D Dice of one synthetic case predicted slice-batch-wise in eval mode (aide_amd.inference.predict_case)."""
    from aide_amd.inference import predict_case, Dice3d_fn, keep_largest_connected_components
    from aide_amd.synthetic import chaos_batch
    inphase, outphase, targets = chaos_batch(slices, args.img_size, seed=args.torch_seed * 7919 + 13, single_modal=single)
    net.eval()
    pred = predict_case(net, inphase, batch_size=slices) if single else predict_case(net, inphase, outphase, batch_size=slices)
    net.train()
    pred = keep_largest_connected_components(pred)                   # :267-268 (CPU post-processing, as in the reference)
    tgt = targets.permute(1, 2, 0).contiguous().numpy()
    if tgt.sum() == 0 and pred.sum() == 0:
        return 1.0
    return float(Dice3d_fn(pred, tgt))


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(message)s')
    Train()
