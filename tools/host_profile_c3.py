import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench as B
from aide_amd.optim import Adam
from aide_amd.synthetic import chaos_batch
from aide_amd.utils import CoTeachingProposedLoss
from aide_amd.train_files.trainchaos_proposed_30cases1labeled import coteach_step
dev = torch.device('cuda:0')
n1, n2 = B.build('fuseunet', dev), B.build('fuseunet', dev); n1.train(); n2.train()
o1, o2 = Adam(n1.parameters(), lr=1e-4, amsgrad=True), Adam(n2.parameters(), lr=1e-4, amsgrad=True)
op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
xin, xout, t = [v.to(dev) for v in chaos_batch(4, 256, seed=1234)]
augs = [((xin * (1 + 0.05 * k)), (xout * (1 - 0.05 * k))) for k in range(4)]
augset = {'augno': [4] * 4}
for k in range(4):
    augset['hflip%d' % (k + 1)] = [(k + b) % 2 for b in range(4)]
    augset['degree%d' % (k + 1)] = [15.0 * (k + 1) - 7.0 * b for b in range(4)]
step = lambda: coteach_step(n1, n2, o1, o2, op, xin, xout, augs, t, t, 0.25, augset=augset, pipeline=True)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8): step()
h = (time.perf_counter() - t0) / 8
torch.cuda.synchronize()
w = (time.perf_counter() - t0) / 8
print('c3 host enqueue %.2f ms, wall %.2f ms per step' % (h * 1e3, w * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(4): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
