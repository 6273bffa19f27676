"""Repeat tests/test_gpu_steps.py::test_coteaching_two_streams_is_bit_identical's computation and report WHERE a mismatch appears.
python tools/r6_flaky_c3.py [repeats] [switch=0 ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aide_amd.models_twomodalinputs import fuseunet                     # noqa: E402
from aide_amd.optim import Adam                                         # noqa: E402
from aide_amd.utils import CoTeachingProposedLoss                       # noqa: E402
from aide_amd.train_files import trainchaos_proposed_30cases1labeled as M    # noqa: E402


def run(two, sw, dev, fx):
    T = lambda k: torch.from_numpy(fx[k]).to(dev)
    augs = [(T('aug%d_in' % i), T('aug%d_out' % i)) for i in range(4)]
    n = T('xin').shape[0]
    augset = {'augno': [4] * n}
    for k in range(4):
        augset['hflip%d' % (k + 1)] = [(k + b) % 2 for b in range(n)]
        augset['degree%d' % (k + 1)] = [15.0 * (k + 1) - 7.0 * b for b in range(n)]
    M.TWO_NET_STREAMS[0] = bool(two)
    torch.manual_seed(2)
    n1, n2 = fuseunet(2).to(dev), fuseunet(2).to(dev)
    for net in (n1, n2):
        net.engine.config.shared_packs = bool(two)
        for k, v in sw.items():
            setattr(net.engine.config, k, v)
    n1.train(); n2.train()
    o1, o2 = Adam(n1.parameters(), lr=1e-3, amsgrad=True), Adam(n2.parameters(), lr=1e-3, amsgrad=True)
    op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
    trace = []
    for _ in range(3):
        r = M.coteach_step(n1, n2, o1, o2, op, T('xin'), T('xout'), augs, T('t1'), T('t2'), 0.25, augset=augset,
                           pipeline=(two == 'pipelined'))
        trace.append((r['loss1'].clone(), r['loss2'].clone(), r['outputs1'].clone(), r['outputs2'].clone(), r['pl1'].clone(), r['pl2'].clone()))
    M.join_networks()
    torch.cuda.synchronize()
    return trace, [p.detach().clone() for p in list(n1.parameters()) + list(n2.parameters())], [nm for nm, _ in list(n1.named_parameters()) + list(n2.named_parameters())]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    sw = {a.split('=')[0]: bool(int(a.split('=')[1])) for a in sys.argv[2:]}
    dev = torch.device('cuda:0')
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'g4_proposed.npz'))
    ref = run(False, sw, dev, fx)
    bad = 0
    for it in range(reps):
        for two in (False, True, 'pipelined'):
            got = run(two, sw, dev, fx)
            for s, (a, b) in enumerate(zip(ref[0], got[0])):
                for name, x, y in zip(('loss1', 'loss2', 'out1', 'out2', 'pl1', 'pl2'), a, b):
                    if not torch.equal(x, y):
                        bad += 1
                        d = (x.float() - y.float()).abs()
                        print('rep %d variant %s step %d %s differs: max %.3e, nan %d, count %d' % (
                            it, two, s, name, float(d.max()), int(torch.isnan(y).sum()), int((x != y).sum())), flush=True)
                        break
                else:
                    continue
                break
            else:
                pd = [(nm, float((p - q).abs().max())) for nm, p, q in zip(ref[2], ref[1], got[1]) if not torch.equal(p, q)]
                if pd:
                    bad += 1
                    print('rep %d variant %s: traces equal, %d parameters differ, first %s' % (it, two, len(pd), pd[:3]), flush=True)
    print('done: %d mismatching runs of %d' % (bad, 3 * reps))


if __name__ == '__main__':
    main()
