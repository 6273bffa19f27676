import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from aide_amd import ops
from aide_amd._lib import lib
dev = torch.device('cuda:0')
for (n, ci, co, h) in [(2, 256, 128, 16), (2, 128, 64, 32), (2, 512, 256, 8), (2, 1024, 512, 4), (2,256,128,16)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, h, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) * 0.05
    dy = torch.randn(n, co, h, h, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=1).backward(dy)
    wf, wd = ops.pack_weights(w.to(dev))
    plan = lib.aide_conv3x3_plan(n, co, h, h, ci)
    ws = torch.empty(1 << 22, device=dev)
    dx = torch.empty(n, ci, h, h, device=dev)
    ops.conv3x3_igemm(dy.to(dev), wd, None, dx, plan=plan, ws=ws)
    e = (dx.cpu() - xr.grad).abs()
    print((n, ci, co, h), 'plan v%d s%d' % (plan & 255, plan >> 8), 'max err', e.max().item(), 'scale', xr.grad.abs().max().item(),
          'bad elems', (e > 1e-4).sum().item(), 'of', e.numel())
    if (e > 1e-4).any():
        idx = (e > 1e-4).nonzero()
        print('   first bad', idx[:5].tolist(), 'channels', sorted(set(idx[:, 1].tolist()))[:20], 'rows', sorted(set(idx[:, 2].tolist())), 'cols', sorted(set(idx[:,3].tolist())))
