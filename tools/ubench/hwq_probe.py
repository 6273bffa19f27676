"""Which HIP streams share a hardware queue?  (The runtime maps a process' streams onto GPU_MAX_HW_QUEUES queues; a queue
runs its packets in order.)  For every pair (a, b): a long spin kernel on a, a tiny kernel on b, and the time until b's kernel
is done -- short = different queues.  usage (GPU box): python tools/ubench/hwq_probe.py [n_streams] [rccl]"""
import os, sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
if len(sys.argv) > 2 and sys.argv[2] == 'rccl':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29551')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    t = torch.ones(1 << 20, device=dev); dist.all_reduce(t); torch.cuda.synchronize()      # RCCL's stream exists and has been used
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n)]
names = ['main'] + ['s%d' % i for i in range(n)]
x = [torch.zeros(64, device=dev) for _ in streams]
for s, xi in zip(streams, x):          # first use in creation order
    with torch.cuda.stream(s):
        xi.add_(1)
torch.cuda.synchronize()
SPIN = 4_000_000                       # ~2 ms
def independent(a, b):
    torch.cuda.synchronize()
    e = torch.cuda.Event()
    with torch.cuda.stream(streams[a]):
        torch.cuda._sleep(SPIN)
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[b]):
        x[b].add_(1)
        e.record()
    e.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt < 1e-3
classes = []
for i in range(len(streams)):
    for c in classes:
        if not independent(c[0], i) or not independent(i, c[0]):
            c.append(i); break
    else:
        classes.append([i])
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES', '(default)'), '| streams that share a hardware queue:')
for c in classes:
    print('  ', ' '.join(names[i] for i in c))
