"""Up-sampling / pooling kernels at the FuseUNet decoder / encoder shapes, alone: ms and effective TB/s.
python tools/bench_spatial.py [c2|c5] [fp32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd import ops          # noqa: E402


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    dt = torch.bfloat16 if (sys.argv[2] if len(sys.argv) > 2 else 'fp32') == 'bf16' else torch.float32
    n, size = (8, 512) if cfg == 'c5' else (4, 256)
    es = 2 if dt == torch.bfloat16 else 4
    dev = torch.device('cuda:0')
    tot = [0.0] * 4
    for c, lv in ((1024, 4), (512, 3), (256, 2), (128, 1)):          # up-sampling inputs (fuseunet.py:85-88)
        h = size >> lv
        x = torch.randn(n, c, h, h, device=dev).to(dt)
        y = torch.empty(n, c, 2 * h, 2 * h, device=dev, dtype=dt)
        dy = torch.randn(n, c, 2 * h, 2 * h, device=dev).to(dt)
        dx = torch.empty_like(x)
        tf = timeit(lambda: ops.upsample2x_fwd(x, y))
        tb = timeit(lambda: ops.upsample2x_bwd(dy, dx))
        nb = 5 * x.numel() * es
        tot[0] += tf; tot[1] += tb
        print('upsample C=%4d %3d->%3d  fwd %.3f ms (%.2f TB/s)  bwd %.3f ms (%.2f TB/s)'
              % (c, h, 2 * h, tf, nb / tf * 1e-9, tb, nb / tb * 1e-9), flush=True)
    for c, lv in ((64, 0), (128, 1), (256, 2), (512, 3)):             # pooled skip tensors
        h = size >> lv
        x = torch.randn(n, c, h, h, device=dev).to(dt)
        y = torch.empty(n, c, h // 2, h // 2, device=dev, dtype=dt)
        dyp = torch.randn_like(y)
        dx = torch.empty_like(x)
        tf = timeit(lambda: ops.maxpool2x2_fwd(x, y))
        tb = timeit(lambda: ops.maxpool2x2_bwd(x, dyp, dx))
        tot[2] += tf; tot[3] += tb
        print('maxpool  C=%4d %3d->%3d  fwd %.3f ms (%.2f TB/s)  bwd %.3f ms (%.2f TB/s)'
              % (c, h, h // 2, tf, 1.25 * x.numel() * es / tf * 1e-9, tb, 2.25 * x.numel() * es / tb * 1e-9), flush=True)
    print('totals: upsample fwd %.3f bwd %.3f, pool fwd %.3f bwd %.3f ms' % tuple(tot))


if __name__ == '__main__':
    main()
