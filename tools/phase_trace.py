"""Unprofiled per-op timeline of one training step (main stream): a HIP event before every op of the forward and the
backward chain, the loss and the optimizer.  usage (GPU box): python tools/phase_trace.py [workload] [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from aide_amd import utils as U
from aide_amd.optim import Adam
from aide_amd.synthetic import chaos_batch

wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
model, batch, size, _ = B.WORKLOADS[wl]
dev = torch.device('cuda:0')
net = B.build(model, dev); net.train()
net.engine.precision = B.WORKLOAD_PRECISION.get(wl, 'fp32')
w = torch.tensor([1.0, 1.0]); crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
xin, xout, t = chaos_batch(batch, size, seed=1234, single_modal=(model != 'fuseunet'))
xin, t = xin.to(dev), t.to(dev); xout = xout.to(dev) if xout is not None else None
def step():
    opt.zero_grad(); out = net(xin, xout) if xout is not None else net(xin)
    loss = crit(out, t); loss.backward(); opt.step()
for _ in range(4): step()
marks = []
def mark(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
def trace(d, st):
    name = st['kind']
    if st['kind'] in ('conv', 'convT'):
        name += ' %d->%d L%d' % (st['src'].C, st['dst'].C, st['dst'].level)
    mark(d + ' ' + name)
res = []
for i in range(steps):
    plan = [p for p in net.engine.plans.values() if p.training][0]
    plan.trace = trace if i == steps - 1 else None
    torch.cuda.synchronize()
    if i == steps - 1: mark('start')
    opt.zero_grad(); out = net(xin, xout) if xout is not None else net(xin)
    if i == steps - 1: mark('loss')
    loss = crit(out, t); loss.backward()
    if i == steps - 1: mark('adam')
    opt.step()
    if i == steps - 1: mark('end')
torch.cuda.synchronize()
t0 = marks[0][1]
prev = 0.0
for (tag, e), (_, e2) in zip(marks, marks[1:] + [marks[-1]]):
    print('%9.1f  +%7.1f  %s' % (t0.elapsed_time(e) * 1e3, e.elapsed_time(e2) * 1e3, tag))
