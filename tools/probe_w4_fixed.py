"""Fixed vs per-stage cost of the F(4x4) forward kernel: time Cin in {32..512} -> 64 @ 256x256 for N in {2, 4, 8}
(256 / 512 / 1024 workgroups = 1 / 2 / 4 rounds of the 256 CUs).  usage: probe_w4_fixed.py  (AIDE_HIP_LIB selects the build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aide_amd import ops
dev = torch.device('cuda:0')
ws = torch.empty(1 << 26, device=dev)
tag = os.path.basename(os.environ.get('AIDE_HIP_LIB', 'current'))
for N in (2, 4, 8):
    for ci in (32, 64, 128, 256, 512):
        co, h = 64, 256
        x = torch.randn(N, ci, h, h, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        y = torch.empty(N, co, h, h, device=dev)
        u, _ = ops.wino4_pack(w); f = lambda: ops.conv3x3_wino4(x, u, None, y, ws=ws)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 30 * 1e3
        rounds = N * 16 * 8 / 256
        print('%-16s N=%d ci=%3d  %7.1f us  per round %6.1f us  per stage-round %.3f us' % (tag, N, ci, t, t / rounds, t / rounds / (ci / 4)))
