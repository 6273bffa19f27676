"""Time one conv entry point on one shape (A/B of library builds via AIDE_HIP_LIB). usage: ab_one.py mode Ci Co H"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aide_amd import ops
mode, ci, co, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device('cuda:0'); N = 4
x = torch.randn(N, ci, h, h, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
y = torch.empty(N, co, h, h, device=dev); dy = torch.randn(N, co, h, h, device=dev); dw = torch.empty_like(w)
ws = torch.empty(1 << 26, device=dev)
if mode == 'wino4':
    u, _ = ops.wino4_pack(w); f = lambda: ops.conv3x3_wino4(x, u, None, y, ws=ws)
elif mode == 'wino':
    u, _ = ops.wino_pack(w); f = lambda: ops.conv3x3_wino(x, u, None, y, ws=ws)
elif mode == 'wgrad4':
    f = lambda: ops.conv3x3_wgrad_wino4(dy, x, dw, ws=ws)
else:
    f = lambda: ops.conv3x3_wgrad_wino(dy, x, dw, ws=ws)
for _ in range(5): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(30): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 30
print('%-60s %s %d->%d@%d  %.4f ms  %.1f TF' % (os.path.basename(os.environ.get('AIDE_HIP_LIB', 'current')), mode, ci, co, h, t, 2.0 * N * h * h * ci * co * 9 / 1e9 / t))
