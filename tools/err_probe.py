import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, oracle
from aide_amd import engine, utils as U
from test_gpu_models import build_pair, forced_relu_masks
dev = torch.device('cuda:0')
def run(kind, shape, wino, dbl=False):
    net, ref = build_pair(kind, False, dev)
    net.engine.config.use_winograd = bool(wino)
    g = torch.Generator().manual_seed(9)
    n, h, w = shape
    xs = [torch.randn(n, 3, h, w, generator=g) for _ in range(2 if kind == 'fuseunet' else 1)]
    t = (torch.rand(n, h, w, generator=g) > 0.8).long()
    wt = torch.tensor([1.0, 1.0])
    net.train(); ref.train()
    out = net(*[x.to(dev) for x in xs]); U.CEMDiceLoss(wt, wt, wt)(out, t.to(dev)).backward()
    plan = list(net.engine.plans.values())[0]
    if dbl:
        ref = ref.double(); xs = [x.double() for x in xs]
    with forced_relu_masks(net, ref, plan) as fm:
        if dbl:
            for k in fm.masks: fm.masks[k] = fm.masks[k].double()
        out_r = ref(*xs); oracle.CEMDiceLoss(wt.to(out_r.dtype), wt.to(out_r.dtype), wt.to(out_r.dtype))(out_r, t).backward()
    le = ((out.cpu().double() - out_r.double()).abs().max() / out_r.abs().max()).item()
    worst, wk = 0, ''
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        s = q.grad.abs().max().item()
        if k.endswith('.bias') and s < 1e-6: continue
        e = (p.grad.cpu().double() - q.grad.double()).abs().max().item() / s
        if e > worst: worst, wk = e, k
    print('%-9s %-14s wino=%d ref=%s  logits %.2e  worst grad %.2e (%s) flips %d' % (kind, shape, wino, 'f64' if dbl else 'f32', le, worst, wk, sum(fm.flips.values())))
for kind, shape in (('unet', (1, 160, 176)), ('fuseunet', (2, 64, 64)), ('fuseunet', (1, 128, 128))):
    for wino in (0, 1):
        run(kind, shape, wino); run(kind, shape, wino, dbl=True)
