"""Host (Python) time of one training step: the wall time the host needs to ENQUEUE a step, measured over a run of steps
with no synchronisation inside (the GPU falls behind; nothing waits for it), plus a cProfile breakdown of the same loop.
usage (GPU box): python tools/host_profile.py [workload] [steps] [--profile]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from aide_amd import utils as U
from aide_amd.optim import Adam
from aide_amd.synthetic import chaos_batch

args = [a for a in sys.argv[1:] if not a.startswith('--')]
wl = args[0] if args else 'c2'
steps = int(args[1]) if len(args) > 1 else 8
model, batch, size, _ = B.WORKLOADS[wl]
dev = torch.device('cuda:0')
net = B.build(model, dev); net.train()
net.engine.precision = B.WORKLOAD_PRECISION.get(wl, 'fp32')
w = torch.tensor([1.0, 1.0]); crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
xin, xout, t = chaos_batch(batch, size, seed=1234, single_modal=(not model.startswith('fuseunet')))
xin, t = xin.to(dev), t.to(dev); xout = xout.to(dev) if xout is not None else None


def step():
    opt.zero_grad()
    out = net(xin, xout) if xout is not None else net(xin)
    loss = crit(out, t)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()

# where the host time of a step goes inside the engine (the backward runs on autograd's device thread, which cProfile does
# not see): wall time of the wrapped functions, accumulated over the measured steps
import collections
from aide_amd import engine as E, tape as T
acc = collections.defaultdict(float)
def wrap(owner, name, tag):
    fn = getattr(owner, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[tag] += time.perf_counter() - t
    setattr(owner, name, w)
wrap(T.Tape, 'replay', 'tape.replay (forward + backward)')
wrap(E.Plan, '_pack_filters', 'plan._pack_filters')
wrap(E.Plan, 'backward', 'plan.backward')
wrap(E.Plan, 'forward', 'plan.forward')
wrap(E.Plan, '_fingerprint', 'plan._fingerprint')
wrap(E.Engine, '_refresh_params', 'engine._refresh_params')
bw = E._NetFunction.backward
def bwd(ctx, g):
    t = time.perf_counter()
    try:
        return bw(ctx, g)
    finally:
        acc['_NetFunction.backward'] += time.perf_counter() - t
E._NetFunction.backward = staticmethod(bwd)
phases = [0.0] * 5
t_all = time.perf_counter()
for _ in range(steps):
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); out = net(xin, xout) if xout is not None else net(xin)
    t2 = time.perf_counter(); loss = crit(out, t)
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
        phases[k] += (b - a) / steps
host = (time.perf_counter() - t_all) / steps
torch.cuda.synchronize()
gpu = (time.perf_counter() - t_all) / steps
print('host enqueue time per step: %.3f ms (zero_grad %.3f, forward %.3f, loss %.3f, backward %.3f, optimizer %.3f); '
      'wall per step incl. GPU drain: %.3f ms' % (host * 1e3, *[p * 1e3 for p in phases], gpu * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-40s %.3f ms / step' % (k, v / steps * 1e3))
plan = [p for p in net.engine.plans.values() if p.training][0]
for nm, tp in (('forward', plan._tape_f), ('backward', plan._tape_b)):
    if tp is not None:
        c = collections.Counter(('py' if e[0] is None else e[3]) for e in tp.calls)
        print('  %s tape: %d entries: %s' % (nm, len(tp.calls), ', '.join('%s x%d' % kv for kv in c.most_common())))
if '--profile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
