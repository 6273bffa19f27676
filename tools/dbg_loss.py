import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from aide_amd import utils as U
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
z = torch.randn(2, 2, 32, 32, generator=g)
t = (torch.rand(2, 32, 32, generator=g) > 0.7).long()
w = torch.tensor([1.0, 1.0])
for name in ('CEMDiceLoss', 'CEMDiceLossImage', 'CrossEntropyLoss2d', 'MulticlassDiceLoss'):
    kw = dict(cediceweight=w, ceclassweight=w, diceclassweight=w) if name.startswith('CEM') else {}
    zr = z.clone().requires_grad_(True)
    lr = getattr(oracle, name)(**kw)(zr, t); lr.sum().backward()
    zd = z.to(dev).requires_grad_(True)
    l = getattr(U, name)(**kw)(zd, t.to(dev)); l.sum().backward()
    print(name, l.detach().cpu(), lr.detach(), 'grad rel err', ((zd.grad.cpu() - zr.grad).abs().max() / zr.grad.abs().max()).item())
