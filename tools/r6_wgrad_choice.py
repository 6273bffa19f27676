"""the weight-gradient algorithm per layer against today's main queue.  Same-process, interleaved A/B of the
unprofiled step (c2: FuseUNet 256x256 bs 4; c4: UNet 320x320 bs 4): baseline (F(4x4) on half of the chip wherever supported)
against the F(2x2) weight gradient (whole chip) on each decoder layer where it wins alone (profiles/r05_layers_c2_fp32.txt),
individually and together.  python tools/r6_wgrad_choice.py [c2|c4] [steps] [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd import engine as E, utils as U           # noqa: E402
from aide_amd.optim import Adam                        # noqa: E402
from aide_amd.synthetic import chaos_batch             # noqa: E402


def make(workload, dev):
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNet
    torch.manual_seed(2)
    two = workload == 'c2'
    net = (fuseunet(2) if two else UNet(2)).to(dev)
    net.train()
    size = 256 if two else 320
    xin, xout, t = chaos_batch(4, size, seed=1234, single_modal=not two)
    args = (xin.to(dev), xout.to(dev)) if two else (xin.to(dev),)
    w = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    t = t.to(dev)

    def step():
        opt.zero_grad()
        loss = crit(net(*args), t)
        loss.backward()
        opt.step()
    return net, step


def timed(step, steps):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device('cuda:0')
    net, step = make(workload, dev)
    step()
    plan = list(net.engine.plans.values())[-1]
    layers = []
    for st in plan.steps:
        if st['kind'] == 'conv' and st.get('wino_w') == 4:
            n, co, hh, ww = st['z'].shape
            ci = st['src'].C
            if st['flops'] >= 15e9 and (co, ci, hh, ww) not in layers:
                layers.append((co, ci, hh, ww))
    variants = [('baseline F(4x4)', {})] + [('%d->%d @%d F(2x2)' % (ci, co, hh), {(co, ci, hh, ww): 2}) for co, ci, hh, ww in layers]
    variants.append(('all of them F(2x2)', {k: 2 for k in layers}))
    res = {name: [] for name, _ in variants}
    for r in range(rounds):
        for name, ov in variants:
            E.WGRAD_OVERRIDE.clear()
            E.WGRAD_OVERRIDE.update(ov)
            net.engine._config_changed()               # drop plans / tapes / packs: rebuilt under the override
            res[name].append(timed(step, steps))
    base = sum(res[variants[0][0]]) / rounds
    print('| layer(s) on the F(2x2) weight gradient | ms / step (%d rounds x %d steps) | images/s | vs baseline |' % (rounds, steps))
    print('|---|---|---|---|')
    for name, _ in variants:
        ms = sum(res[name]) / rounds
        print('| %s | %s | %.1f | %+.2f %% |' % (name, ' / '.join('%.3f' % v for v in res[name]), 4e3 / ms, (base / ms - 1) * 100))


if __name__ == '__main__':
    main()
