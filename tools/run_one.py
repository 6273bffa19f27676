"""Run ONE conv kernel family on one shape a few times (target for rocprofv3 --pmc).
usage: run_one.py {fwd|dgrad|wgrad|wino|wino4|wino_wgrad} Ci Co H [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aide_amd import ops
from aide_amd._lib import lib
mode, ci, co, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device('cuda:0'); N = 4
x = torch.randn(N, ci, h, h, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
dy = torch.randn(N, co, h, h, device=dev); y = torch.empty(N, co, h, h, device=dev)
dx = torch.empty(N, ci, h, h, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
wf, wd = ops.pack_weights(w)
uf, ud = ops.wino_pack(w)
uf4, ud4 = ops.wino4_pack(w) if ops.wino4_supported(ci, h, h, co) else (None, None)
ws = torch.empty(1 << 26, device=dev)
for _ in range(iters):
    if mode == 'fwd': ops.conv3x3_igemm(x, wf, None, y, ws=ws)
    elif mode == 'dgrad': ops.conv3x3_igemm(dy, wd, None, dx, ws=ws)
    elif mode == 'wino': ops.conv3x3_wino(x, uf, None, y, ws=ws)
    elif mode == 'wino4': ops.conv3x3_wino4(x, uf4, None, y, ws=ws)
    elif mode == 'wino_wgrad': ops.conv3x3_wgrad_wino(dy, x, dw, ws=ws)
    else: ops.conv3x3_wgrad(dy, x, dw, ws=ws)
torch.cuda.synchronize()
