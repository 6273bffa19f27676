"""AIDE proposed co-teaching loop, prostate cross-domain form, with the reference's CLI
(train_files/trainprostate_proposed_isbi3ttransferisbidx.py: parse_args :26-58, inner step :253-327; its sibling
trainprostate_proposed_isbidxtransferisbi3t.py differs in data paths only).

Two U-Nets of ONE `--model_name` are co-trained.  Unlike the kidney / breast scripts the step never calls eval(): the four
augmentation passes run in train mode and move the BatchNorm statistics (as in the CHAOS script), the pseudo labels are
sharpened with p^T (:96-100), the small-loss selection keeps two images (:301-304), and both best checkpoints are named
'{model}_temp{T}_r{R}_net{k}_besttraincasedice.pkl' (:173-174, :489-503).  Everything else -- stacked augmentation pass,
on-device reverse augmentation, fused selection and losses, Adam, evaluation -- is the machinery of
aide_amd.train_files.trainchaos_proposed_30cases1labeled (variant 'prostate', pinned by fixture g20 'prostate').
"""
import argparse
import logging

from aide_amd.train_files import trainchaos_proposed_30cases1labeled as _core


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Prostate segmentation, AIDE proposed (MI355X HIP engine)')
    p.add_argument('--model_name', default='UNet', type=str)
    p.add_argument('--data_mean', default=None, nargs='+', type=float)
    p.add_argument('--data_std', default=None, nargs='+', type=float)
    p.add_argument('--rotation', default=60, type=float)
    p.add_argument('--batch_size', default=4, type=int)
    p.add_argument('--gpu_order', default='0', type=str)
    p.add_argument('--torch_seed', default=2, type=int)
    p.add_argument('--lr', default=1e-4, type=float)
    p.add_argument('--warmup_epoch', default=20, type=int)
    p.add_argument('--num_epoch', default=100, type=int)
    p.add_argument('--loss', default='cedice', type=str)
    p.add_argument('--img_size', default=256, type=int)
    p.add_argument('--temperature', default=1.0, type=float)
    p.add_argument('--lr_policy', default='StepLR', type=str)
    p.add_argument('--cedice_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--segcor_weight', default=[1.0, 10.0], nargs='+', type=float)
    p.add_argument('--ceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--diceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--checkpoint', default='checkpoint_train3tgeneratedx_comparisoncrossdomain/')
    p.add_argument('--history', default='history_train3tgeneratedx_comparisoncrossdomain')
    p.add_argument('--cudnn', default=0, type=int)
    p.add_argument('--repetition', default=100, type=int)
    # not in the reference: size of the synthetic epoch (there is no dataset on this path)
    p.add_argument('--steps_per_epoch', default=8, type=int)
    return p.parse_args(argv)


coteach_step = _core.coteach_step
join_networks = _core.join_networks


def Train(args=None):
    return _core.Train(args or parse_args(), variant='prostate')


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(message)s')
    Train()
