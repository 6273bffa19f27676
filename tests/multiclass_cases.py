"""The loss forms of tests/golden/g17_multiclass.npz (written by oracle/gen_golden.py::g17_multiclass from the real
reference): (fixture key, module name, constructor keywords from the fixture's weights, which target array)."""
import torch

CLASS_COUNTS = (3, 5, 8)


def loss_cases(fx, pre):
    cw = torch.from_numpy(fx[pre + 'class_w'])
    cdw = torch.from_numpy(fx[pre + 'cedice_w'])
    full = dict(cediceweight=cdw, ceclassweight=cw, diceclassweight=cw)
    return (('CrossEntropyLoss2d', 'CrossEntropyLoss2d', dict(weight=cw), 'ignore'),
            ('CrossEntropyLoss2d_sum', 'CrossEntropyLoss2d', dict(weight=cw, reduction='sum'), 'ignore'),
            ('CrossEntropyLoss2d_none', 'CrossEntropyLoss2d', dict(weight=cw, reduction='none'), 'ignore'),
            ('CrossEntropyLoss2d_onehot', 'CrossEntropyLoss2d', dict(weight=cw), 'onehot'),
            ('DiceLoss', 'DiceLoss', dict(), 'index'),
            ('DiceLoss_none', 'DiceLoss', dict(reduction='none'), 'index'),
            ('Dice_Loss_sum', 'Dice_Loss', dict(reduction='sum'), 'index'),
            ('MulticlassDiceLoss', 'MulticlassDiceLoss', dict(weight=cw), 'index'),
            ('MulticlassDiceLoss_onehot', 'MulticlassDiceLoss', dict(weight=cw), 'onehot'),
            ('MulticlassDiceLoss_onehot_none', 'MulticlassDiceLoss', dict(weight=cw, reduction='none'), 'onehot'),
            ('CEMDiceLoss', 'CEMDiceLoss', dict(full), 'index'),
            ('CEMDiceLoss_sum', 'CEMDiceLoss', dict(full, reduction='sum'), 'index'),
            ('CEMDiceLossImage', 'CEMDiceLossImage', dict(full), 'index'),
            ('CEDiceLoss', 'CEDiceLoss', dict(cediceweight=cdw, classweight=cw), 'index'))


def targets_of(fx, pre, kind, C):
    t = torch.from_numpy(fx[pre + 'targets'])
    if kind == 'ignore':
        return torch.from_numpy(fx[pre + 'targets_ignore'])
    if kind == 'onehot':
        return torch.nn.functional.one_hot(t, C).permute(0, 3, 1, 2).float().contiguous()
    return t


def upstream(v):
    """the upstream gradient the fixture was generated with: linspace(0.5, 1.5) over a non-scalar loss"""
    return torch.linspace(0.5, 1.5, v.numel()).view(v.shape).to(v.device) if v.dim() else None
