"""Filter re-layout (the pack launches of one FuseUNet step) timed alone.  python tools/bench_pack.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd import engine as E                         # noqa: E402
from aide_amd.models_twomodalinputs import fuseunet      # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda:0')
    net = fuseunet(2).to(dev).train()
    x = torch.randn(4, 3, 256, 256, device=dev)
    net(x, x)
    plan = list(net.engine.plans.values())[0]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        E.PARAM_EPOCH[0] += 1
        e0.record()
        plan._pack_filters()
        if plan.side_fwd is not None:
            torch.cuda.current_stream().wait_stream(plan.side_fwd)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print('filter re-layout of one step: %.3f ms' % (tot / reps))


if __name__ == '__main__':
    main()
