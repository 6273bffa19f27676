import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aide_amd.models_twomodalinputs import fuseunet
from aide_amd import utils as U
from aide_amd.optim import Adam
from aide_amd.synthetic import chaos_batch
dev = torch.device('cuda:0')
torch.manual_seed(2)
net = fuseunet(2).to(dev); net.train()
w = torch.tensor([1.0, 1.0])
crit = U.CEMDiceLoss(w, w, w); opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
xin, xout, t = [v.to(dev) for v in chaos_batch(4, 256, 1234)]
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
acc = [0.0] * 4
for it in range(25):
    e0 = ev(); opt.zero_grad(); out = net(xin, xout); e1 = ev(); loss = crit(out, t); e2 = ev(); loss.backward(); e3 = ev(); opt.step(); e4 = ev()
    torch.cuda.synchronize()
    if it >= 5:
        for k, (a, b) in enumerate(((e0, e1), (e1, e2), (e2, e3), (e3, e4))): acc[k] += a.elapsed_time(b) / 20
print('forward %.3f ms  loss %.3f  backward %.3f  adam %.3f  total %.3f' % (*acc, sum(acc)))
