"""Same-process interleaved A/B of engine.config switches on the unprofiled co-teaching step (C3: two FuseUNets, 256x256, bs 4).
python tools/r6_ab_config_c3.py <steps> <rounds> <switch>=<0|1> [...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd.models_twomodalinputs import fuseunet                     # noqa: E402
from aide_amd.optim import Adam                                         # noqa: E402
from aide_amd.synthetic import chaos_batch                              # noqa: E402
from aide_amd.utils import CoTeachingProposedLoss                       # noqa: E402
from aide_amd.train_files.trainchaos_proposed_30cases1labeled import coteach_step, join_networks    # noqa: E402


def main():
    steps, rounds = int(sys.argv[1]), int(sys.argv[2])
    variants = [('defaults', None)] + [(a, a.split('=')) for a in sys.argv[3:]]
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    n1, n2 = fuseunet(2).to(dev), fuseunet(2).to(dev)
    n1.train(); n2.train()
    o1, o2 = Adam(n1.parameters(), lr=1e-4, amsgrad=True), Adam(n2.parameters(), lr=1e-4, amsgrad=True)
    op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
    xin, xout, t = chaos_batch(4, 256, seed=1234)
    xin, xout, t = xin.to(dev), xout.to(dev), t.to(dev)
    augs = [((xin * (1 + 0.05 * k)), (xout * (1 - 0.05 * k))) for k in range(4)]
    augset = {'augno': [4] * 4}
    for k in range(4):
        augset['hflip%d' % (k + 1)] = [(k + b) % 2 for b in range(4)]
        augset['degree%d' % (k + 1)] = [15.0 * (k + 1) - 7.0 * b for b in range(4)]

    def step():
        coteach_step(n1, n2, o1, o2, op, xin, xout, augs, t, t, 0.25, augset=augset, pipeline=True)

    def timed():
        for _ in range(4):
            step()
        join_networks(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        join_networks(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    res = {n: [] for n, _ in variants}
    for r in range(rounds):
        for name, kv in variants:
            for net in (n1, n2):
                for k in net.engine.config.DEFAULTS:
                    setattr(net.engine.config, k, net.engine.config.DEFAULTS[k])
                if kv is not None:
                    setattr(net.engine.config, kv[0], bool(int(kv[1])))
            res[name].append(timed())
    base = sum(res['defaults']) / rounds
    for name, _ in variants:
        ms = sum(res[name]) / rounds
        print('%-28s %s ms  %.1f images/s  %+.2f %%' % (name, ' / '.join('%.3f' % v for v in res[name]), 4e3 / ms, (base / ms - 1) * 100))


if __name__ == '__main__':
    main()
