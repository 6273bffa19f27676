"""BatchNorm+ReLU forward / backward kernels at the FuseUNet level shapes (HBM-bound): ms and effective TB/s.
python tools/bench_bn.py [c5|c2] [bf16|fp32] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd import ops          # noqa: E402


def timeit(fn, reps):
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
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
    dt = torch.bfloat16 if (sys.argv[2] if len(sys.argv) > 2 else 'bf16') == 'bf16' else torch.float32
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    n, size = (8, 512) if cfg == 'c5' else (4, 256)
    dev = torch.device('cuda:0')
    es = 2 if dt == torch.bfloat16 else 4
    tot = [0.0, 0.0]
    for c, lv in ((32, 0), (64, 0), (64, 1), (128, 1), (128, 2), (256, 2), (256, 3), (512, 3), (512, 4)):
        h = size >> lv
        z = torch.randn(n, c, h, h, device=dev).to(dt)
        dA = torch.randn(n, c, h, h, device=dev).to(dt)
        a, dz = torch.empty_like(z), torch.empty_like(z)
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros((), device=dev, dtype=torch.int64)
        mean, rstd, sc, sh = (torch.empty(c, device=dev) for _ in range(4))
        dg, db, dbias = (torch.empty(c, device=dev) for _ in range(3))
        ws = ops.bn_ws(c, dev)
        tf = timeit(lambda: ops.bn_train_fwd(z, a, g, b, 1e-5, 0.1, rm, rv, nbt, mean, rstd, sc, sh, ws), reps)
        tb = timeit(lambda: ops.bn_relu_bwd(dA, z, dz, mean, rstd, sc, sh, dg, db, dbias, ws), reps)
        nb = z.numel() * es
        tot[0] += tf; tot[1] += tb
        print('C=%4d @%3d  fwd %.3f ms (%.2f TB/s over 3 passes)  bwd %.3f ms (%.2f TB/s over 5 passes)'
              % (c, h, tf, 3 * nb / tf * 1e-9, tb, 5 * nb / tb * 1e-9), flush=True)
    print('total fwd %.3f ms, bwd %.3f ms' % tuple(tot))


if __name__ == '__main__':
    main()
