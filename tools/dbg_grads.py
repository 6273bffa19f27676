import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from aide_amd.models_twomodalinputs import fuseunet
from aide_amd import utils as U
dev = torch.device('cuda:0')
size, nb = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(2); ref = oracle.fuseunet(2)
torch.manual_seed(2); net = fuseunet(2).to(dev)
g = torch.Generator().manual_seed(1234)
xs = [torch.randn(nb, 3, size, size, generator=g) for _ in range(2)]
t = (torch.rand(nb, size, size, generator=g) > 0.7).long()
w = torch.tensor([1.0, 1.0])
out_r = ref(*xs); out_r.retain_grad()
oracle.CEMDiceLoss(w, w, w)(out_r, t).backward()
out = net(*[x.to(dev) for x in xs])
# feed the REFERENCE dlogits to isolate the network backward
out.backward(out_r.grad.to(dev))
names = [k for k, _ in net.named_parameters()]
for (k, p), (_, q) in reversed(list(zip(net.named_parameters(), ref.named_parameters()))):
    e = (p.grad.cpu().double() - q.grad.double()).abs().max().item()
    s = q.grad.double().abs().max().item()
    print('%-50s err %.2e  scale %.2e  rel %.2e' % (k, e, s, e / (s + 1e-30)))
