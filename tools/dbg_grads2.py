import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from aide_amd.models_twomodalinputs import fuseunet
dev = torch.device('cuda:0')
size, nb = 32, 2
torch.manual_seed(2); ref = oracle.fuseunet(2)
torch.manual_seed(2); net = fuseunet(2).to(dev)
g = torch.Generator().manual_seed(1234)
xs = [torch.randn(nb, 3, size, size, generator=g) for _ in range(2)]
t = (torch.rand(nb, size, size, generator=g) > 0.7).long()
w = torch.tensor([1.0, 1.0])
caps = {}
def cap(name):
    def hook(m, i, o):
        o.retain_grad(); caps[name] = o
    return hook
for k in range(1, 5):
    ub = getattr(ref, 'up_block%d' % k)
    ub.bilinear_up.register_forward_hook(cap('up%d' % k))
    ub.block.register_forward_hook(cap('up%d_out' % k))
    ub.bilinear_up[0].register_forward_hook(cap('upsampled%d' % k))
out_r = ref(*xs); out_r.retain_grad()
oracle.CEMDiceLoss(w, w, w)(out_r, t).backward()
out = net(*[x.to(dev) for x in xs])
out.backward(out_r.grad.to(dev))
plan = list(net.engine.plans.values())[0]
gr = net.engine.graph
roots = {t.name: t for t in gr.roots}
def cmp(name, ours, ref_t):
    e = (ours.cpu() - ref_t).abs()
    print('%-16s act/grad err %.2e scale %.2e  bad %d/%d' % (name, e.max().item(), ref_t.abs().max().item(), (e > 1e-3 * ref_t.abs().max()).sum().item(), e.numel()))
    return e
for k in (4, 3, 2, 1):
    cat = roots['cat_s%d' % (5 - k)]
    prev = cat.C // 2
    cmp('fwd up%d' % k, plan.act[id(cat)][:, :prev], caps['up%d' % k].detach())
    e = cmp('d up%d' % k, plan.grad[id(cat)][:, :prev], caps['up%d' % k].grad)
    if e.max() > 1e-6:
        idx = (e > 1e-3 * caps['up%d' % k].grad.abs().max()).nonzero()
        print('    bad n', sorted(set(idx[:,0].tolist())), 'ch', len(set(idx[:,1].tolist())), 'rows', sorted(set(idx[:,2].tolist())), 'cols', sorted(set(idx[:,3].tolist())))
    cmp('d up%d_out' % k, plan.grad[id(roots['up%d_out' % k])], caps['up%d_out' % k].grad)
    cmp('d upsampled%d' % k, plan.grad[id(roots['upsampled%d' % k])], caps['upsampled%d' % k].grad)
print('--- relu mask flips (ours>0 vs ref>0) ---')
for k in (4, 3, 2, 1):
    cat = roots['cat_s%d' % (5 - k)]
    prev = cat.C // 2
    a = plan.act[id(cat)][:, :prev].cpu(); r = caps['up%d' % k].detach()
    flips = ((a > 0) != (r > 0)).nonzero()
    print('up%d flips' % k, flips.tolist()[:5], [ (a[tuple(f)].item(), r[tuple(f)].item()) for f in flips[:5]])
    if len(flips):
        f = tuple(flips[0]); print('    dA at flip', caps['up%d' % k].grad[f].item())
