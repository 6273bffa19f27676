"""Per-case inference of the training scripts, on device and batched.

Reference (train_files/trainchaos_comparison_1case.py:233-273, :275-314; the same loop in
trainchaos_proposed_30cases1labeled.py:373-493 and evalchaos_comparison_1cases.py:143-243): for every
slice of a case, bs=1: `argmax(softmax(net(inphase, outphase), dim=1), dim=1)` under `net.eval()` +
`no_grad`, `.cpu().numpy()` per slice, `np.stack(..., axis=-1)`, then skimage's largest connected
component (CPU, out of scope here) and `Dice3d_fn`.

Here the slices of a case go through the eval-mode kernels in batches (eval BatchNorm uses the running
statistics, so slices are independent and batching changes nothing but the launch count), the label
map is one kernel (`aide_label_map`) and the volume leaves the device once.
"""
import numpy as np
import torch

from ._lib import lib, check
from .ops import ptr, stream_ptr


def label_map(logits):
    """[N,C,H,W] fp32 logits (C = 2 .. 8) -> [N,H,W] int64 labels = torch.argmax(F.softmax(logits, 1), 1)."""
    if logits.dim() != 4 or not 2 <= logits.shape[1] <= 8 or logits.dtype != torch.float32 or not logits.is_cuda:
        raise RuntimeError('label_map expects a [N,C,H,W] fp32 HIP tensor with 2 <= C <= 8')
    logits = logits.detach()
    if not logits.is_contiguous():
        logits = logits.contiguous()
    n, c, h, w = logits.shape
    out = torch.empty(n, h, w, device=logits.device, dtype=torch.int64)
    if c == 2:
        check(lib.aide_label_map(ptr(logits), 2 * h * w, n, h * w, ptr(out), stream_ptr()), 'label_map')
    else:
        check(lib.aide_label_map_mc(ptr(logits), c * h * w, c, n, h * w, ptr(out), stream_ptr()), 'label_map_mc')
    return out


def predict_labels(net, *modal_inputs, **kw):
    """Labels [S,H,W] (int64, on device) for S slices; `modal_inputs` = (inphase[, outphase]) each
    [S,3,H,W].  The network must be in eval mode, as in the reference loop (:210 `net.eval()`)."""
    batch_size = kw.pop('batch_size', 16)
    if kw:
        raise TypeError('unexpected arguments %r' % sorted(kw))
    if net.training:
        raise RuntimeError('predict_labels needs net.eval(): train-mode BatchNorm at bs=1 is not what the '
                           'reference loop runs')
    s = modal_inputs[0].shape[0]
    dev = next(net.parameters()).device
    out = []
    with torch.no_grad():
        for i in range(0, s, batch_size):
            xs = [m[i:i + batch_size].to(dev, non_blocking=True) for m in modal_inputs]
            out.append(label_map(net(*xs)))
    return torch.cat(out, 0) if len(out) > 1 else out[0]


def predict_case(net, *modal_inputs, **kw):
    """The reference's `generatedtarget`: numpy int64 [H,W,S] (slices stacked on the last axis, :267)."""
    return predict_labels(net, *modal_inputs, **kw).permute(1, 2, 0).contiguous().cpu().numpy()


def Dice3d_fn(inputs, targets):
    """trainchaos_comparison_1case.py:88-95 for label volumes (numpy arrays or tensors): 2·Σ(i·t)/(Σi+Σt).
    Integer sums, one float64 division — identical to the numpy original (0/0 -> nan like numpy)."""
    if isinstance(inputs, np.ndarray) and isinstance(targets, np.ndarray):
        i, t = inputs.reshape(-1), targets.reshape(-1)
        return 2 * np.sum(i * t) / (np.sum(i) + np.sum(t))
    i = torch.as_tensor(inputs).reshape(-1).to(torch.int64)
    t = torch.as_tensor(targets).reshape(-1).to(i.device, torch.int64)
    inter, union = 2 * int((i * t).sum().item()), int(i.sum().item()) + int(t.sum().item())
    return np.float64(inter) / np.float64(union) if union else np.float64('nan')


def keep_largest_connected_components(mask):
    """trainchaos_comparison_1case.py:68-77 / trainchaos_proposed_30cases1labeled.py:103-112: the largest connected blob of a
    label volume (connectivity 1: face neighbours), uint8.  CPU post-processing in the reference (skimage.measure.label +
    regionprops).  skimage connects neighbours of EQUAL value (0 = background) and numbers the blobs in raster order of
    their first voxel; scipy.ndimage.label connects every non-zero voxel, so it is run once per label value and the blobs
    are ordered by their first voxel -- the same blob as the reference's `np.argmax(area)` on ties, and for a multi-class
    volume (num_classes up to 8 here) blobs of different classes stay separate as they do there."""
    from scipy import ndimage
    mask = np.asarray(mask)
    out = np.zeros(mask.shape, dtype=np.uint8)
    if mask.size == 0 or mask.max() <= 0:
        return out
    best = None                                   # (area, -first voxel, value, blob labels, blob id)
    for v in np.unique(mask):
        if v == 0:
            continue
        blobs, count = ndimage.label(mask == v)
        flat = blobs.reshape(-1)
        area = np.bincount(flat, minlength=count + 1)[1:]
        first = np.full(count + 1, flat.size, dtype=np.int64)
        idx = np.flatnonzero(flat)
        np.minimum.at(first, flat[idx], idx)
        for b in range(count):
            cand = (int(area[b]), -int(first[b + 1]))
            if best is None or cand > best[0]:
                best = (cand, blobs, b + 1)
    out[best[1] == best[2]] = 1
    return out
