"""reverseaug on the GPU — same signature and in-place behaviour as the reference's
reverseaug(augset, augoutput, classno) (train_files/trainchaos_proposed_30cases1labeled.py:81-95), which
moves every (batch, aug, class) logit plane to the host, runs PIL transpose + rotate(BILINEAR) and copies
it back: 32 planes per network per step, the only host round trip inside the co-teaching hot loop
(SURVEY.md §8f row 1).  Here one HIP kernel per augmented output does flip + PIL-exact rotation."""
import math

import torch

from .._lib import lib, check
from ..ops import stream_ptr, ptr, planes


def pil_rotate_params(angle, w, h):
    """The inverse affine matrix PIL's Image.rotate(angle) builds (expand=False, centre = image centre),
    and the fast-path mode PIL takes for multiples of 90 degrees."""
    ang = float(angle) % 360.0
    if ang == 0:
        return [1, 0, 0, 0, 1, 0], 1
    if ang == 180:
        return [1, 0, 0, 0, 1, 0], 2
    if ang in (90, 270) and w == h:
        return [1, 0, 0, 0, 1, 0], 3 if ang == 90 else 4
    a = -math.radians(ang)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return m, 0


def reverse_aug_tensor(logits, hflips, degrees):
    """logits [N,C,H,W] (HIP) -> new tensor with, per image n: optional horizontal flip, then rotation
    by -degrees[n] (the reference rotates by 0 - degree, :86)."""
    xp, xbs = planes(logits)
    n, c, h, w = logits.shape
    rows = []
    for i in range(n):
        m, mode = pil_rotate_params(0 - float(degrees[i]), w, h)
        rows.append(m + [1.0 if bool(hflips[i]) else 0.0, float(mode)])
    par = torch.tensor(rows, dtype=torch.float64).to(logits.device)
    out = torch.empty_like(logits)
    op, obs = planes(out)
    check(lib.aide_reverse_aug(xp, xbs, op, obs, ptr(par), n, c, h, w, stream_ptr()), 'reverse_aug')
    return out


def reverseaug(augset, augoutput, classno):
    """Drop-in for the reference function: augoutput[k] is the [N,classno,H,W] output of augmented pass k;
    augset carries 'augno', 'hflip{k}', 'degree{k}' per batch element (datasetchaos_proposed/dataset.py)."""
    nb = len(augset['augno'])
    naug = int(augset['augno'][0])
    for k in range(naug):
        assert augoutput[k].shape[1] == classno
        flips = [augset['hflip%d' % (k + 1)][b] for b in range(nb)]
        degs = [augset['degree%d' % (k + 1)][b] for b in range(nb)]
        # the reference leaves an element untouched when k >= augno[b]
        act = [k < int(augset['augno'][b]) for b in range(nb)]
        flips = [f if a else 0 for f, a in zip(flips, act)]
        degs = [d if a else 0.0 for d, a in zip(degs, act)]
        augoutput[k] = reverse_aug_tensor(augoutput[k].contiguous(), flips, degs)
    return augoutput
