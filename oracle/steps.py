"""Oracle (test infrastructure): the two training inner steps, restated.

The reference's train scripts cannot be imported (they import skimage/pydicom
and run argparse/mkdir at import — SURVEY.md §8c), so the inner loops are
restated here line by line and validated in ``oracle/gen_golden.py`` by running
them with the *imported reference* modules/losses plugged in.

Reference:
  train_files/trainchaos_comparison_1case.py:190-202          comparison step
  train_files/trainchaos_proposed_30cases1labeled.py:260-330  AIDE proposed step
"""
import torch
import torch.nn.functional as F

from .losses import sharpen


def comparison_step(net, criterion, optimizer, inphase, outphase, targets):
    """trainchaos_comparison_1case.py:195-199. ``outphase=None`` -> single-modal net(x)."""
    optimizer.zero_grad()
    outputs = net(inphase, outphase) if outphase is not None else net(inphase)
    loss = criterion(outputs, targets)
    loss.backward()
    optimizer.step()
    return outputs.detach(), loss.detach()


def reverseaug(augset, augoutput, classno):
    """trainchaos_proposed_30cases1labeled.py:81-95, restated: per (batch, aug, class) plane -> PIL 'F'
    image -> optional FLIP_LEFT_RIGHT -> rotate(0 - degree, BILINEAR) -> back.  PIL (pillow, unpinned in
    requirements.txt:3) is the third-party arithmetic here; this oracle calls it directly."""
    import numpy as np
    from PIL import Image
    for b in range(len(augset['augno'])):
        for k in range(int(augset['augno'][b])):
            flip = augset['hflip%d' % (k + 1)][b]
            rot = 0 - float(augset['degree%d' % (k + 1)][b])
            for c in range(classno):
                m = Image.fromarray(augoutput[k][b, c].cpu().numpy(), mode='F')
                if flip:
                    m = m.transpose(Image.FLIP_LEFT_RIGHT)
                m = m.rotate(rot, Image.BILINEAR)
                augoutput[k][b, c] = torch.from_numpy(np.array(m))
    return augoutput


def pseudo_labels(aug_logits, temperature, sharpen_fn=None):
    """trainchaos_proposed_30cases1labeled.py:274-292 — mean softmax over the (already
    reverse-augmented) passes, sharpen, weightmap = 1 - 4 p0 p1.  sharpen_fn: losses.sharpen (p^T, the CHAOS script) or
    losses.sharpen_root (p^(1/T), trainkidney_proposed_mask1.py:113-117 and the other UNet scripts)."""
    acc = None
    for lg in aug_logits:
        sm = F.softmax(lg, dim=1)
        acc = sm if acc is None else acc + sm
    pl = (sharpen_fn or sharpen)(acc / float(len(aug_logits)), temperature)
    wm = (1.0 - 4.0 * pl[:, 0] * pl[:, 1]).unsqueeze(1)
    return pl, wm


def proposed_losses(criterion, corr, outputs1, outputs2, targets1, targets2,
                    pl1, wm1, pl2, wm2, rate, segcor_weight=(1.0, 10.0), keep=2):
    """trainchaos_proposed_30cases1labeled.py:303-321. ``criterion`` = CEMDiceLossImage,
    ``corr`` = MulticlassMSELoss('none'). Returns (loss1, loss2, indx1, indx2, seg1pre, seg2pre).
    keep: 2 in the CHAOS / kidney scripts (:307-310), int(batch_size / 2) in the breast scripts
    (trainbreast_dataset3_proposed_272cases25labeled.py:303-312)."""
    l1pre = criterion(outputs1, targets2)          # net1 scored against net2's labels (:303)
    l2pre = criterion(outputs2, targets1)
    _, indx1 = l1pre.sort()
    _, indx2 = l2pre.sort()
    k2, d2, k1, d1 = indx2[:keep], indx2[keep:], indx1[:keep], indx1[keep:]
    l1_s1 = criterion(outputs1[k2], targets2[k2]).mean()
    l2_s1 = criterion(outputs2[k1], targets1[k1]).mean()
    l1_s2 = criterion(outputs1[d2], targets2[d2]).mean()
    l2_s2 = criterion(outputs2[d1], targets1[d1]).mean()
    l1_cor = (wm2[d2] * corr(outputs1[d2], pl2[d2])).mean()
    l2_cor = (wm1[d1] * corr(outputs2[d1], pl1[d1])).mean()
    loss1 = segcor_weight[0] * (l1_s1 + (1.0 - rate) * l1_s2) + segcor_weight[1] * rate * l1_cor
    loss2 = segcor_weight[0] * (l2_s1 + (1.0 - rate) * l2_s2) + segcor_weight[1] * rate * l2_cor
    return loss1, loss2, indx1, indx2, l1pre.detach(), l2pre.detach()


def proposed_step(net1, net2, criterion, corr, opt1, opt2, inphase, outphase, aug_pairs,
                  targets1, targets2, rate, temperature=1.0, segcor_weight=(1.0, 10.0),
                  reverse=None, keep=2, eval_aug=False, sharpen_fn=None):
    """One AIDE co-teaching step. ``aug_pairs`` = list of (imgmodal1_k, imgmodal2_k) -- or of single tensors / 1-tuples for a
    single-modal net (``outphase=None``); ``reverse`` = callable(list_of_logits)->list_of_logits (identity when None).
    The three forms the reference's nine ``*_proposed_*`` scripts take (pinned by fixture g20, which EXECUTES their loop bodies):
      CHAOS (trainchaos_proposed_30cases1labeled.py:260-325): fuseunet, train-mode augmentation passes, sharpen p^T, keep 2;
      kidney / prostate (trainkidney_proposed_mask1.py:262-333): UNet, nets in eval() for the augmentation passes (:265-266)
        and back to train() (:290-291), sharpen p^(1/T), keep 2;
      breast (trainbreast_dataset3_proposed_272cases25labeled.py:258-331): as kidney with keep = int(batch_size / 2) (:304)."""
    def fwd(net, x, y):
        return net(x, y) if y is not None else net(x)
    if eval_aug:
        net1.eval()
        net2.eval()
    a1, a2 = [], []
    for pair in aug_pairs:
        pair = tuple(pair) if isinstance(pair, (tuple, list)) else (pair,)
        xin, xout = pair[0], (pair[1] if len(pair) > 1 else None)
        a1.append(fwd(net1, xin, xout).detach())
        a2.append(fwd(net2, xin, xout).detach())
    if reverse is not None:
        a1, a2 = reverse(a1), reverse(a2)
    pl1, wm1 = pseudo_labels(a1, temperature, sharpen_fn)
    pl2, wm2 = pseudo_labels(a2, temperature, sharpen_fn)
    if eval_aug:
        net1.train()
        net2.train()
    opt1.zero_grad()
    opt2.zero_grad()
    o1 = fwd(net1, inphase, outphase)
    o2 = fwd(net2, inphase, outphase)
    loss1, loss2, indx1, indx2, l1pre, l2pre = proposed_losses(
        criterion, corr, o1, o2, targets1, targets2, pl1, wm1, pl2, wm2, rate, segcor_weight, keep)
    loss1.backward(retain_graph=True)                             # :322-325
    opt1.step()
    loss2.backward()
    opt2.step()
    return dict(outputs1=o1.detach(), outputs2=o2.detach(), loss1=loss1.detach(),
                loss2=loss2.detach(), indx1=indx1, indx2=indx2, loss1_pre=l1pre, loss2_pre=l2pre,
                pl1=pl1, pl2=pl2, wm1=wm1, wm2=wm2)


def predict_case(net, inphase, outphase=None):
    """trainchaos_comparison_1case.py:257-267 — per slice, bs=1, eval mode: softmax -> argmax -> numpy,
    slices stacked on the last axis.  Returns (generatedtarget int64 [H,W,S], logits [S,2,H,W])."""
    import numpy as np
    gen, lg = [], []
    for i in range(inphase.shape[0]):
        with torch.no_grad():
            a = torch.unsqueeze(inphase[i], 0)
            output = net(a, torch.unsqueeze(outphase[i], 0)) if outphase is not None else net(a)
            lg.append(output[0].clone())
            output = F.softmax(output, dim=1)
            output = torch.argmax(output, dim=1)
            gen.append(output.squeeze().cpu().numpy())
    return np.stack(gen, axis=-1), torch.stack(lg, 0)


def Dice3d_fn(inputs, targets):
    """trainchaos_comparison_1case.py:88-95."""
    import numpy as np
    iflat = inputs.reshape(-1)
    tflat = targets.reshape(-1)
    intersection = 2 * np.sum(iflat * tflat)
    union = np.sum(iflat) + np.sum(tflat)
    return intersection / union
