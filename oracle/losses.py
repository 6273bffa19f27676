"""Oracle (test infrastructure): restatement of the reference's segmentation
losses and co-teaching selection on CPU with stock aten ops.

Reference:
  utils/loss2d.py:5-13     CrossEntropyLoss2d
  utils/loss2d.py:35-61    DiceLoss
  utils/loss2d.py:87-107   MulticlassDiceLoss (index targets -> class-1 Dice only)
  utils/loss2d.py:109-117  MulticlassMSELoss
  utils/loss2d.py:119-135  CEMDiceLoss
  utils/loss2d.py:137-154  CEMDiceLossImage
  utils/coteach_loss.py:94-119   Coteachingloss_dropimage
  utils/coteach_loss.py:121-161  Coteachingloss_weightimage
  utils/coteach_loss.py:85-92    KLbidirection
  utils/coteach_loss.py:163-196  Coteachingloss_dropregionce
  utils/coteach_loss.py:198-254  Coteachingloss_dropimagedroppixel
  utils/reg_loss.py:58-193       Pixelcoreg_Focalloss, Pixelcoreg_Focalloss_twomodel
  utils/metrics2d.py:8-29  Dice_fn
  train_files/trainchaos_proposed_30cases1labeled.py:97-101 sharpen
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def _ce(inputs, targets, weight, reduction, ignore_index=255):
    # loss2d.py:8,10-13 : one-hot (4-D) targets are arg-maxed first
    if targets.dim() > 3:
        targets = torch.argmax(targets.float(), dim=1)
    return F.cross_entropy(inputs, targets, weight=weight, reduction=reduction,
                           ignore_index=ignore_index)


def _reduce(per_image, n, reduction):
    # loss2d.py:53-60
    if reduction == 'mean':
        return per_image.sum() / n
    if reduction == 'sum':
        return per_image.sum()
    if reduction == 'none':
        return per_image
    raise ValueError(reduction)


def _dice_from_prob(prob_fg, target, smooth, reduction):
    # loss2d.py:42-52 : operands are down-cast with .float()
    n = target.size(0)
    p = prob_fg.reshape(n, -1).float()
    t = target.reshape(n, -1).float()
    per = 1.0 - (2.0 * (p * t).sum(1) + smooth) / (p.sum(1) + t.sum(1) + smooth)
    return _reduce(per, n, reduction)


class CrossEntropyLoss2d(nn.Module):
    def __init__(self, weight=None, reduction='mean', ignore_index=255):
        super().__init__()
        self.register_buffer('weight', weight)
        self.reduction, self.ignore_index = reduction, ignore_index

    def forward(self, inputs, targets):
        return _ce(inputs, targets, self.weight, self.reduction, self.ignore_index)


class DiceLoss(nn.Module):
    def __init__(self, weight=None, smooth=1.0, reduction='mean'):
        super().__init__()
        self.weight, self.smooth, self.reduction = weight, smooth, reduction

    def forward(self, input, target):
        # loss2d.py:44-48 : 4-D input = logits -> softmax -> class 1; 3-D input = probability
        if input.dim() > 3:
            input = F.softmax(input, dim=1)[:, 1]
        return _dice_from_prob(input, target, self.smooth, self.reduction)


class MulticlassDiceLoss(nn.Module):
    def __init__(self, weight=None, smooth=1.0, reduction='mean'):
        super().__init__()
        self.weight, self.smooth, self.reduction = weight, smooth, reduction

    def forward(self, input, target):
        prob = F.softmax(input, dim=1)                      # loss2d.py:96
        if target.dim() > 3:                                # loss2d.py:98-104 one-hot targets
            total = 0
            for c in range(target.shape[1]):
                d = _dice_from_prob(prob[:, c], target[:, c], self.smooth, self.reduction)
                if self.weight is not None:
                    d = d * self.weight[c]
                total = total + d
            return total
        return _dice_from_prob(prob[:, 1], target, self.smooth, self.reduction)  # :106


class MulticlassMSELoss(nn.Module):
    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, input, target):
        return F.mse_loss(F.softmax(input, dim=1), target, reduction=self.reduction)


class CEMDiceLoss(nn.Module):
    def __init__(self, cediceweight=None, ceclassweight=None, diceclassweight=None,
                 reduction='mean'):
        super().__init__()
        self.cediceweight = cediceweight
        self.ce = CrossEntropyLoss2d(ceclassweight, reduction)
        self.multidice = MulticlassDiceLoss(diceclassweight, reduction=reduction)

    def forward(self, inputs, targets):
        ce, dice = self.ce(inputs, targets), self.multidice(inputs, targets)
        if self.cediceweight is not None:                   # loss2d.py:131-134
            return ce * self.cediceweight[0] + dice * self.cediceweight[1]
        return ce + dice


class CEMDiceLossImage(nn.Module):
    def __init__(self, cediceweight=None, ceclassweight=None, diceclassweight=None,
                 reduction='mean'):
        super().__init__()
        self.cediceweight = cediceweight
        self.ce = CrossEntropyLoss2d(ceclassweight, 'none')
        self.multidice = MulticlassDiceLoss(diceclassweight, reduction='none')

    def forward(self, inputs, targets):
        ce = self.ce(inputs, targets).mean(dim=[1, 2])      # loss2d.py:147-148 plain pixel mean
        dice = self.multidice(inputs, targets)
        if self.cediceweight is not None:
            return ce * self.cediceweight[0] + dice * self.cediceweight[1]
        return ce + dice


def _ct_image_loss(weight, inputs, targets):
    # coteach_loss.py:102 : w * mean_hw(NLL(log_softmax)) + Dice_i ; reduction='none' variant
    ce = F.nll_loss(F.log_softmax(inputs, dim=1), targets, reduction='none', ignore_index=255)
    prob = F.softmax(inputs, dim=1)[:, 1]
    return weight * ce.mean(dim=[1, 2]) + _dice_from_prob(prob, targets, 1.0, 'none')


def _argsort_host(loss):
    # coteach_loss.py:104-105 : numpy argsort of the per-image losses on the host (stable for ties
    # at these sizes: lower index first)
    return torch.from_numpy(np.argsort(loss.detach().cpu().numpy(), kind='stable'))


class Coteachingloss_dropimage(nn.Module):
    """coteach_loss.py:94-119. Only meaningful with reduction='none' (SURVEY §2.1 #5)."""

    def __init__(self, weight=1.0, reduction='mean'):
        super().__init__()
        if reduction != 'none':
            # reference raises IndexError from torch.mean(dim=[1,2]) on a scalar; restate loudly
            raise IndexError("Coteachingloss_* require reduction='none' (reference behaviour)")
        self.weight = weight

    def forward(self, inputs1, inputs2, targets, forget_rate):
        l1 = _ct_image_loss(self.weight, inputs1, targets)
        l2 = _ct_image_loss(self.weight, inputs2, targets)
        i1, i2 = _argsort_host(l1), _argsort_host(l2)
        keep = int((1 - forget_rate) * l1.shape[0])         # :107-108 truncation
        k1, k2 = i1[:keep], i2[:keep]
        u1 = _ct_image_loss(self.weight, inputs1[k2], targets[k2])   # cross selection :114-117
        u2 = _ct_image_loss(self.weight, inputs2[k1], targets[k1])
        return u1.mean(dim=0), u2.mean(dim=0)


class Coteachingloss_weightimage(nn.Module):
    """coteach_loss.py:121-161: dropped images kept with weight 0.1 (shape-broadcast as in the
    reference: valid when |keep| == |drop| or one of them has one element)."""

    def __init__(self, weight=1.0, reduction='mean'):
        super().__init__()
        if reduction != 'none':
            raise IndexError("Coteachingloss_* require reduction='none' (reference behaviour)")
        self.weight = weight

    def forward(self, inputs1, inputs2, targets, forget_rate):
        l1 = _ct_image_loss(self.weight, inputs1, targets)
        l2 = _ct_image_loss(self.weight, inputs2, targets)
        i1, i2 = _argsort_host(l1), _argsort_host(l2)
        keep = int((1 - forget_rate) * l1.shape[0])
        k1, d1, k2, d2 = i1[:keep], i1[keep:], i2[:keep], i2[keep:]
        u1 = _ct_image_loss(self.weight, inputs1[k2], targets[k2])
        if len(d1) > 0:                                      # :141 (quirk: tests ind_1_drop)
            u1 = u1 + 0.1 * _ct_image_loss(self.weight, inputs1[d2], targets[d2])
        u2 = _ct_image_loss(self.weight, inputs2[k1], targets[k1])
        if len(d2) > 0:
            u2 = u2 + 0.1 * _ct_image_loss(self.weight, inputs2[d1], targets[d1])
        return u1.mean(dim=0), u2.mean(dim=0)


def KLbidirection(inputs1, inputs2):
    """coteach_loss.py:85-92: per-pixel KL(p1||p2) + KL(p2||p1) of the two softmax maps -> [N,H,W]."""
    p1 = F.softmax(inputs1, dim=1)
    p2 = F.softmax(inputs2, dim=1)
    kl12 = torch.sum(p1 * torch.log(p1 / p2), dim=1)
    kl21 = torch.sum(p2 * torch.log(p2 / p1), dim=1)
    return kl12 + kl21


class Coteachingloss_dropregionce(nn.Module):
    """coteach_loss.py:163-196: CE on max-pooled regions (logits pooled per class, targets pooled), per image the
    `num_remember` regions with the smallest loss of the OTHER net are kept; mean over all kept regions."""

    def __init__(self, scale=0.5, reduction='none'):
        super().__init__()
        self.scale = scale
        self.reduction = reduction

    def forward(self, inputs1, inputs2, targets, forget_rate):
        total_w, total_h = inputs1.shape[2], inputs1.shape[3]
        patch_w, patch_h = int(total_w * self.scale), int(total_h * self.scale)
        kernel = (int(total_w / patch_w), int(total_h / patch_h))
        pool = lambda x: F.max_pool2d(x, kernel_size=kernel, stride=kernel, padding=0, ceil_mode=True)
        z1, z2 = pool(inputs1), pool(inputs2)
        tp = pool(targets.float()).long()
        n = z1.shape[0]
        loss1 = F.nll_loss(F.log_softmax(z1, dim=1), tp, reduction=self.reduction, ignore_index=255).view(n, -1)
        loss2 = F.nll_loss(F.log_softmax(z2, dim=1), tp, reduction=self.reduction, ignore_index=255).view(n, -1)
        i1 = torch.from_numpy(np.argsort(loss1.detach().cpu().numpy(), axis=-1, kind='stable'))
        i2 = torch.from_numpy(np.argsort(loss2.detach().cpu().numpy(), axis=-1, kind='stable'))
        keep = int((1 - forget_rate) * loss1.shape[1])
        u1 = torch.cat([loss1[i, i2[i, :keep]] for i in range(n)], dim=0)      # :186-194
        u2 = torch.cat([loss2[i, i1[i, :keep]] for i in range(n)], dim=0)
        return torch.mean(u1), torch.mean(u2)


class Coteachingloss_dropimagedroppixel(nn.Module):
    """coteach_loss.py:198-254: image-level small-loss selection as dropimage, plus 0.25 x a pixel-level term on
    the dropped images: (KL + CE) on foreground pixels, the smallest remember_rate fraction kept.  Quirks kept:
    branch 2 reuses num_remember2 of branch 1 (:249) and tests len(ind_2_drop) (:240)."""

    def __init__(self, weight=1.0, reduction='mean'):
        super().__init__()
        if reduction != 'none':
            raise IndexError("Coteachingloss_* require reduction='none' (reference behaviour)")
        self.weight = weight

    def forward(self, inputs1, inputs2, targets, forget_rate):
        ce = lambda z, t: F.nll_loss(F.log_softmax(z, dim=1), t, reduction='none', ignore_index=255)
        l1 = _ct_image_loss(self.weight, inputs1, targets)
        l2 = _ct_image_loss(self.weight, inputs2, targets)
        i1, i2 = _argsort_host(l1), _argsort_host(l2)
        rr = 1 - forget_rate
        keep = int(rr * l1.shape[0])
        k1, d1, k2, d2 = i1[:keep], i1[keep:], i2[:keep], i2[keep:]
        u1 = _ct_image_loss(self.weight, inputs1[k2], targets[k2])
        u2 = _ct_image_loss(self.weight, inputs2[k1], targets[k1])
        drop1 = drop2 = 0.0
        keep2 = None
        if len(d1) > 0:
            t1 = targets[d2]
            v = (KLbidirection(inputs1[d2], inputs2[d2]) + ce(inputs1[d2], t1)).view(-1) * t1.view(-1).float()
            fore = v[v > 0]
            order = torch.from_numpy(np.argsort(fore.detach().cpu().numpy(), kind='stable'))
            keep2 = int(rr * len(order))
            drop1 = torch.mean(fore[order[:keep2]])
        if len(d2) > 0:
            t2 = targets[d1]
            v = (KLbidirection(inputs1[d1], inputs2[d1]) + ce(inputs2[d1], t2)).view(-1) * t2.view(-1).float()
            fore = v[v > 0]
            order = torch.from_numpy(np.argsort(fore.detach().cpu().numpy(), kind='stable'))
            drop2 = torch.mean(fore[order[:keep2]])          # :249 num_remember2 of branch 1
        return u1.mean(dim=0) + 0.25 * drop1, u2.mean(dim=0) + 0.25 * drop2


def _focal_kd_maps(targets, *inputs):
    """reg_loss.py:69-97: one log_softmax / softmax per net, shared by the focal(gamma=2) terms (lossweight is
    overwritten with 1, :67) and by KL(p1||p2) + KL(p2||p1) of nets 1 and 2 — same expression order as the
    reference, so the autograd accumulation order (and with it the last bits of the gradients) is the same."""
    n = targets.shape[0]
    ls = [F.log_softmax(z, dim=1) for z in inputs]
    sm = [F.softmax(z, dim=1) for z in inputs]
    focal = [(-targets.float() * torch.pow(1 - q[:, 1, :, :], 2) * l[:, 1, :, :]
              - 1 * (1 - targets).float() * torch.pow(1 - q[:, 0, :, :], 2) * l[:, 0, :, :]).view(n, -1)
             for l, q in zip(ls, sm)]
    p1, p2 = sm[0], sm[1]
    k12 = (p1[:, 0, :, :] * torch.log(p1[:, 0, :, :] / p2[:, 0, :, :]) +
           p1[:, 1, :, :] * torch.log(p1[:, 1, :, :] / p2[:, 1, :, :])).view(n, -1)
    k21 = (p2[:, 0, :, :] * torch.log(p2[:, 0, :, :] / p1[:, 0, :, :]) +
           p2[:, 1, :, :] * torch.log(p2[:, 1, :, :] / p1[:, 1, :, :])).view(n, -1)
    return focal, k12, k21


def _pixel_select(key, value, targets, forget_rate, reduction):
    # reg_loss.py:101-129: per image the int((1 - forget_rate) * HW) pixels with the smallest key
    n = key.shape[0]
    order = torch.from_numpy(np.argsort(key.detach().cpu().numpy(), axis=-1, kind='stable'))
    keep = int((1 - forget_rate) * key.shape[1])
    upd = torch.cat([value[i, order[i, :keep]] for i in range(n)], dim=0).view(n, -1)
    if reduction == 'mean':
        upd = torch.mean(upd)
    elif reduction == 'sum':
        upd = torch.sum(upd)
    ts = targets.view(n, -1)
    tsel = torch.cat([ts[i, order[i, :keep]] for i in range(n)], dim=0)
    return upd, tsel.sum() / targets.sum()


class Pixelcoreg_Focalloss(nn.Module):
    """reg_loss.py:58-131: three nets; pixels ranked by (1-kd)(focal1+focal2+focal3) + kd KL(1,2); the loss is the
    mean of net 3's focal loss over the kept pixels; second output = kept foreground fraction."""

    def __init__(self, smooth=1.0, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, inputs1, inputs2, inputs3, targets, forget_rate, kdweight, device=None):
        (l1, l2, l3), k12, k21 = _focal_kd_maps(targets, inputs1, inputs2, inputs3)
        key = (1 - kdweight) * (l1 + l2 + l3) + kdweight * (k12 + k21)
        return _pixel_select(key, l3, targets, forget_rate, self.reduction)


class Pixelcoreg_Focalloss_twomodel(nn.Module):
    """reg_loss.py:133-193: two nets; the combined map is both the ranking key and the value that is averaged."""

    def __init__(self, smooth=1.0, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, inputs1, inputs2, targets, forget_rate, kdweight, device=None):
        (l1, l2), k12, k21 = _focal_kd_maps(targets, inputs1, inputs2)
        key = (1 - kdweight) * (l1 + l2) + kdweight * (k12 + k21)
        return _pixel_select(key, key, targets, forget_rate, self.reduction)


def Dice_fn(inputs, targets, threshold=0.5):
    """metrics2d.py:8-29 — hard Dice, returns the SUM over the batch."""
    fg = (F.softmax(inputs, dim=1)[:, 1] >= threshold).float()
    total = 0.0
    for p, t in zip(fg, targets):
        p, t = p.reshape(-1), t.reshape(-1).float()
        if t.sum() == 0:
            d = torch.tensor(1.0) if p.sum() == 0 else torch.tensor(0.0)
        else:
            d = 2.0 * (p * t).sum() / (p.sum() + t.sum())
        total = total + d
    return total


def _hard_fg(inputs, threshold):
    return (F.softmax(inputs, dim=1)[:, 1] >= threshold).float()


def Dice_fn_Nozero(inputs, targets, threshold=0.5):
    """metrics2d.py:31-52 — (Dice sum as a float, images that are not empty in both target and prediction)."""
    count = 0
    for p, t in zip(_hard_fg(inputs, threshold), targets):
        if t.sum() != 0 or p.sum() != 0:
            count += 1
    return float(Dice_fn(inputs, targets, threshold)), count


def TP_TN_FP_FN(inputs, targets, threshold=0.5):
    """metrics2d.py:54-70 — the loop overwrites the counts per image: the LAST image's values are returned."""
    p, t = _hard_fg(inputs, threshold)[-1].reshape(-1), targets[-1].reshape(-1).float()
    return (p * t).sum(), ((1 - p) * (1 - t)).sum(), (p * (1 - t)).sum(), ((1 - p) * t).sum()


def IoU_fn(inputs, targets, threshold=0.5):
    """metrics2d.py:72-84 — sum over the batch of intersection / union (0/0 = NaN is kept)."""
    total = 0.0
    for p, t in zip(_hard_fg(inputs, threshold), targets):
        p, t = p.reshape(-1), t.reshape(-1).float()
        inter = (p * t).sum()
        total = total + inter / (p.sum() + t.sum() - inter)
    return total


class Dice_Loss(nn.Module):
    """loss2d.py:63-85 — DiceLoss's logits branch."""

    def __init__(self, smooth=1.0, reduction='mean'):
        super().__init__()
        self.inner = DiceLoss(smooth=smooth, reduction=reduction)

    def forward(self, inputs, targets):
        return self.inner(inputs, targets)


class CEDiceLoss(nn.Module):
    """loss2d.py:156-171."""

    def __init__(self, cediceweight=None, classweight=None, reduction='mean'):
        super().__init__()
        self.cediceweight = cediceweight
        self.ce = CrossEntropyLoss2d(weight=classweight, reduction=reduction)
        self.dice = DiceLoss(weight=classweight, reduction=reduction)

    def forward(self, inputs, targets):
        a, b = self.ce(inputs, targets), self.dice(inputs, targets)
        if self.cediceweight is not None:
            return a * self.cediceweight[0] + b * self.cediceweight[1]
        return a + b


def sharpen(mask, temperature):
    """trainchaos_proposed_30cases1labeled.py:97-101 — p^T / sum_c p^T."""
    m = torch.pow(mask, temperature)
    return m / m.sum(dim=1).unsqueeze(dim=1)


def sharpen_root(mask, temperature):
    """trainkidney_proposed_mask1.py:113-117 (and the breast scripts) — p^(1/T) / sum_c p^(1/T): the flavour of the eight
    UNet `*_proposed_*` scripts."""
    m = torch.pow(mask, 1.0 / temperature)
    return m / m.sum(dim=1).unsqueeze(dim=1)
