"""VERDICT r5 item 6, the cheap half: what is there to gain from co-scheduling the two networks' stacked augmentation passes?
Times ONLY those passes (4 x bs 4 stacked, train-mode BatchNorm, lazy BatchNorm, as in the C3 step): (a) network 1 and network 2
on two streams (the step's schedule), (b) one after the other on one stream (every launch has the chip to itself), (c) one
network alone x 2.  If (a) ~ (b) the chip is full either way and one grouped launch per layer cannot beat them.
python tools/r6_c3_stacked.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd.models_twomodalinputs import fuseunet     # noqa: E402
from aide_amd.synthetic import chaos_batch              # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    n1, n2 = fuseunet(2).to(dev), fuseunet(2).to(dev)
    n1.train(); n2.train()
    xin, xout, _ = chaos_batch(4, 256, seed=1)
    augs = [((xin * (1 + 0.05 * k)).to(dev), (xout * (1 - 0.05 * k)).to(dev)) for k in range(4)]
    s2 = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream()

    def two_streams():
        s2.wait_stream(cur)
        n1.forward_groups(augs)
        with torch.cuda.stream(s2):
            n2.forward_groups(augs)
        cur.wait_stream(s2)

    def one_stream():
        n1.forward_groups(augs)
        n2.forward_groups(augs)

    def alone():
        n1.forward_groups(augs)

    def t(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    for r in range(3):
        print('round %d: two streams %.3f ms | one stream %.3f ms | one network alone %.3f ms (x2 = %.3f)'
              % (r, t(two_streams), t(one_stream), t(alone), 2 * t(alone)), flush=True)


if __name__ == '__main__':
    main()
