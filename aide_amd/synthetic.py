"""Synthetic CHAOS-shaped batches (no dataset I/O on the hot path).

Reproduces the *tensor contract* of the reference loader, not its file I/O:
  datasetchaos_comparison/dataset.py:24-25   grayscale slice replicated to 3 channels
  datasetchaos_comparison/transform.py:128   /255
  datasetchaos_comparison/transform.py:160-163  per-image per-channel standardise, unbiased std
  datasetchaos_comparison/dataset.py:63-67 + trainchaos_comparison_1case.py:194
      targets = one-hot mask channel 1 -> int64 {0,1} [N,H,W]
Statistics follow SURVEY.md §8(d): ~28 % exact-zero background, ~12 % saturated pixels,
one smooth foreground blob (0-14 % of the slice), ~40 % of slices empty.
"""
import numpy as np
import torch


def _smooth_field(rng, size, coarse=8):
    """Bilinear-upsampled coarse noise in [0,1] (numpy only)."""
    g = rng.rand(coarse + 1, coarse + 1)
    xs = np.linspace(0, coarse, size, endpoint=False)
    i0 = np.floor(xs).astype(np.int64)
    f = (xs - i0)[None, :]
    rows = g[:, i0] * (1 - f) + g[:, i0 + 1] * f            # [coarse+1, size]
    fy = (xs - i0)[:, None]
    return rows[i0, :] * (1 - fy) + rows[i0 + 1, :] * fy     # [size, size]


def chaos_slice(rng, size):
    """One (in-phase u8, out-phase u8, liver mask {0,1}) triple, HxW."""
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    cy, cx = size * (0.5 + 0.04 * rng.randn()), size * (0.5 + 0.04 * rng.randn())
    ry, rx = size * 0.46, size * 0.50
    body = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0      # ~72 % of the slice
    f = _smooth_field(rng, size)
    tex = 0.06 * rng.randn(size, size)
    g1 = np.clip(255.0 * (0.10 + 1.33 * f + tex), 0, 255)            # ~12 % saturate
    g2 = np.clip(255.0 * (0.05 + 1.10 * f + 0.08 * rng.randn(size, size)), 0, 255)
    g1 = np.where(body, g1, 0.0).round().astype(np.uint8)
    g2 = np.where(body, g2, 0.0).round().astype(np.uint8)
    mask = np.zeros((size, size), np.int64)
    if rng.rand() >= 0.4:                                            # ~40 % empty slices
        frac = rng.uniform(0.005, 0.14)
        r = np.sqrt(frac / np.pi) * size
        by = cy + rng.uniform(-0.15, 0.15) * size
        bx = cx - rng.uniform(0.05, 0.25) * size
        ang = rng.uniform(0, np.pi)
        dy, dx = yy - by, xx - bx
        u = dy * np.cos(ang) + dx * np.sin(ang)
        v = -dy * np.sin(ang) + dx * np.cos(ang)
        mask = ((u / (1.3 * r)) ** 2 + (v / (0.77 * r)) ** 2 <= 1.0) & body
        mask = mask.astype(np.int64)
    return g1, g2, mask


def _to_tensor_norm(g):
    """u8 HxW -> float32 [3,H,W]: replicate, /255, standardise with unbiased std."""
    x = torch.from_numpy(g.astype(np.float32) / 255.0)
    x = x.unsqueeze(0).expand(3, -1, -1).contiguous()
    mean = x.mean(dim=(1, 2), keepdim=True)
    std = x.std(dim=(1, 2), keepdim=True)           # unbiased (transform.py:160-163)
    return (x - mean) / std


def chaos_batch(n, size=256, seed=1234, single_modal=False):
    """Returns (inphase[N,3,H,W] f32, outphase[N,3,H,W] f32, targets[N,H,W] i64) on CPU."""
    rng = np.random.RandomState(seed)
    a, b, t = [], [], []
    for _ in range(n):
        g1, g2, m = chaos_slice(rng, size)
        a.append(_to_tensor_norm(g1))
        b.append(_to_tensor_norm(g2))
        t.append(torch.from_numpy(m))
    inphase, outphase, targets = torch.stack(a), torch.stack(b), torch.stack(t)
    if single_modal:
        return inphase, None, targets
    return inphase, outphase, targets
