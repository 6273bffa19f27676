"""Same-process interleaved A/B of engine.config switches on the unprofiled step.
python tools/r6_ab_config.py <c2|c4> <steps> <rounds> <switch>=<0|1> [...]   (each switch is one variant against the defaults)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                       # noqa: E402
from r6_wgrad_choice import make, timed            # noqa: E402


def main():
    workload, steps, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    variants = [('defaults', None)] + [(a, a.split('=')) for a in sys.argv[4:]]
    net, step = make(workload, torch.device('cuda:0'))
    res = {n: [] for n, _ in variants}
    for r in range(rounds):
        for name, kv in variants:
            for k in net.engine.config.DEFAULTS:
                setattr(net.engine.config, k, net.engine.config.DEFAULTS[k])
            if kv is not None:
                setattr(net.engine.config, kv[0], bool(int(kv[1])))
            res[name].append(timed(step, steps))
    base = sum(res['defaults']) / rounds
    for name, _ in variants:
        ms = sum(res[name]) / rounds
        print('%-28s %s ms  %.1f images/s  %+.2f %%' % (name, ' / '.join('%.3f' % v for v in res[name]), 4e3 / ms, (base / ms - 1) * 100))


if __name__ == '__main__':
    main()
