"""AIDE proposed co-teaching loop, single-modal U-Net form, with the reference's CLI
(train_files/trainkidney_proposed_mask1.py: parse_args :28-60, inner step :262-333).

Two U-Nets (`--model1_name / --model2_name`: UNet | UNetsa) are co-trained; per step the nets are put in eval() for the four
augmentation passes (:265-266, back to train() :290-291), the pseudo labels are sharpened with p^(1/T) (:113-117) and the small-loss
selection keeps two images; both networks start from `--resumefile` (:180-182) and the best checkpoints are named
'{model k}_warmup{W}_temp{T}_r{R}_net{k}_besttraindice.pkl' (:173-176, :451, :461).  Everything else -- stacked augmentation pass, on-device reverse augmentation, fused
selection and losses, Adam, evaluation, checkpoints -- is the machinery of aide_amd.train_files.trainchaos_proposed_30cases1labeled.
"""
import argparse
import logging

from aide_amd.train_files import trainchaos_proposed_30cases1labeled as _core


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Kidney segmentation, AIDE proposed (MI355X HIP engine)')
    p.add_argument('--model1_name', default='UNet', type=str)
    p.add_argument('--model2_name', default='UNet', type=str)
    p.add_argument('--data_mean', default=None, nargs='+', type=float)
    p.add_argument('--data_std', default=None, nargs='+', type=float)
    p.add_argument('--rotation', default=60, type=float)
    p.add_argument('--batch_size', default=4, type=int)
    p.add_argument('--gpu_order', default='0,1', type=str)
    p.add_argument('--torch_seed', default=2, type=int)
    p.add_argument('--lr', default=1e-5, type=float)
    p.add_argument('--warmup_epoch', default=20, type=int)
    p.add_argument('--num_epoch', default=100, type=int)
    p.add_argument('--loss', default='cedice', type=str)
    p.add_argument('--img_size', default=512, type=int)
    p.add_argument('--temperature', default=1.0, type=float)
    p.add_argument('--lr_policy', default='StepLR', type=str)
    p.add_argument('--cedice_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--segcor_weight', default=[1.0, 10.0], nargs='+', type=float)
    p.add_argument('--ceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--diceclass_weight', default=[1.0, 1.0], nargs='+', type=float)
    p.add_argument('--update_percent', default=0.25, type=float)
    p.add_argument('--resumefile', default='initializemodel/UNet_besttraindice_Task1Mask1.pkl')
    p.add_argument('--maskidentity', default=1, type=int)
    p.add_argument('--checkpoint', default='checkpoint_kidney_proposedmask1')
    p.add_argument('--history', default='history_kidney_proposedmask1')
    p.add_argument('--cudnn', default=0, type=int)
    p.add_argument('--repetition', default=100, type=int)
    # not in the reference: size of the synthetic epoch (there is no dataset on this path)
    p.add_argument('--steps_per_epoch', default=8, type=int)
    return p.parse_args(argv)


coteach_step = _core.coteach_step
join_networks = _core.join_networks


def Train(args=None):
    return _core.Train(args or parse_args(), variant='kidney')


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(message)s')
    Train()
