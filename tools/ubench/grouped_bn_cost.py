"""What the per-group BatchNorm launches of the stacked augmentation forward cost (GPU box):
time of net.forward_groups(4 groups of 4) against the same 16 images as ONE BatchNorm batch (different statistics, same
convolution / pooling / up-sampling work) -- the difference is what one launch per layer for all groups could recover."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from aide_amd.synthetic import chaos_batch
dev = torch.device('cuda:0')
net = B.build('fuseunet', dev); net.train()
xin, xout, _ = chaos_batch(4, 256, seed=1)
xin, xout = xin.to(dev), xout.to(dev)
augs = [(xin * (1 + 0.05 * k), xout * (1 - 0.05 * k)) for k in range(4)]
x16a, x16b = torch.cat([a for a, _ in augs]), torch.cat([b for _, b in augs])

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def single():
    with torch.no_grad():
        return net(x16a, x16b)
print('grouped (4 x 4 images, BatchNorm per group): %.3f ms' % timeit(lambda: net.forward_groups(augs)))
print('one batch of 16 (one BatchNorm group)       : %.3f ms' % timeit(single))
