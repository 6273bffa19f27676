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


def _aug_rows(n, w, h, hflips, degrees):
    rows = []
    for i in range(n):
        m, mode = pil_rotate_params(0 - float(degrees[i]), w, h)
        rows.append(m + [1.0 if bool(hflips[i]) else 0.0, float(mode)])
    return rows


def _upload(rows, device):
    """[.., 8] float64 parameter rows -> device, through pinned memory and without blocking the host.  (A plain `.to(device)`
    of a pageable tensor waits for everything queued on the stream: four such waits per network per step kept the host from
    running ahead in the co-teaching loop -- 19 of its 29 ms per step were spent inside them.)"""
    host = torch.tensor(rows, dtype=torch.float64).pin_memory()
    return host.to(device, non_blocking=True)


def reverse_aug_tensor(logits, hflips, degrees, par=None):
    """logits [N,C,H,W] (HIP) -> new tensor with, per image n: optional horizontal flip, then rotation
    by -degrees[n] (the reference rotates by 0 - degree, :86).  par: the [N, 8] device parameter rows, if already uploaded."""
    xp, xbs = planes(logits)
    n, c, h, w = logits.shape
    if par is None:
        par = _upload(_aug_rows(n, w, h, hflips, degrees), logits.device)
    out = torch.empty_like(logits)
    op, obs = planes(out)
    check(lib.aide_reverse_aug(xp, xbs, op, obs, ptr(par), n, c, h, w, stream_ptr()), 'reverse_aug')
    return out


def reverseaug(augset, augoutput, classno):
    """Drop-in for the reference function: augoutput[k] is the [N,classno,H,W] output of augmented pass k;
    augset carries 'augno', 'hflip{k}', 'degree{k}' per batch element (datasetchaos_proposed/dataset.py)."""
    nb = len(augset['augno'])
    # (the reference walks range(augno[b]) per element, :82-83; its caller makes augno[0] passes, :265)
    naug = min(len(augoutput), max(int(a) for a in augset['augno']))
    if naug == 0:
        return augoutput
    h, w = augoutput[0].shape[2], augoutput[0].shape[3]
    rows = []
    for k in range(naug):
        assert augoutput[k].shape[1] == classno
        flips = [augset['hflip%d' % (k + 1)][b] for b in range(nb)]
        degs = [augset['degree%d' % (k + 1)][b] for b in range(nb)]
        # the reference leaves an element untouched when k >= augno[b]
        act = [k < int(augset['augno'][b]) for b in range(nb)]
        flips = [f if a else 0 for f, a in zip(flips, act)]
        degs = [d if a else 0.0 for d, a in zip(degs, act)]
        rows.append(_aug_rows(nb, w, h, flips, degs))
    par = _upload(rows, augoutput[0].device)               # one upload for all passes: [naug, N, 8]
    for k in range(naug):
        augoutput[k] = reverse_aug_tensor(augoutput[k].contiguous(), None, None, par=par[k])
    return augoutput
