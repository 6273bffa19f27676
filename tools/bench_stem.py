"""Stem forward layers alone: the stem kernel against the general kernels it replaces. usage: bench_stem.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aide_amd import ops
dev = torch.device('cuda:0')


def timeit(f, reps=50):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, ci, co, h) in [(4, 3, 32, 256), (4, 3, 64, 256), (16, 3, 32, 256), (8, 3, 32, 512), (4, 3, 64, 320)]:
    x = torch.randn(n, ci, h, h, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.1; b = torch.randn(co, device=dev)
    y = torch.empty(n, co, h, h, device=dev); y16 = torch.empty(n, co, h, h, device=dev, dtype=torch.bfloat16)
    wf, _ = ops.pack_weights(w, need_dgrad=False)
    t_ig = timeit(lambda: ops.conv3x3_igemm(x, wf, b, y))
    t_st = timeit(lambda: ops.conv3x3_stem_fwd(x, w, b, y))
    u16 = ops.bf16_pack(w, need_dgrad=False)[0]
    t_b = timeit(lambda: ops.conv3x3_bf16(x, u16, b, y16)) if u16 is not None else float('nan')
    t_s16 = timeit(lambda: ops.conv3x3_stem_fwd(x, w, b, y16, round_bf16=True))
    by32, by16 = (n * co * h * h * 4 + n * ci * h * h * 4) / 1e3, (n * co * h * h * 2 + n * ci * h * h * 4) / 1e3
    print('N%d %d->%d @%d  fp32: general %.1f us, stem %.1f us (%.2f TB/s)   bf16 out: general %.1f us, stem %.1f us (%.2f TB/s)'
          % (n, ci, co, h, t_ig, t_st, by32 / t_st / 1e3, t_b, t_s16, by16 / t_s16 / 1e3))
