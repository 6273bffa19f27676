"""Per-case inference throughput (trainchaos_comparison_1case.py:233-273 -- eval-mode forward + label map of
every slice of a case).  usage (GPU box): python tools/bench_infer.py [model] [slices] [size] [batch] [reps]
Prints slices/s of aide_amd.inference.predict_labels on a synthetic CHAOS-shaped case resident in HBM, and the share of
the BatchNorm / pooling / up-sampling passes when run under rocprofv3 --kernel-trace --stats."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from aide_amd.inference import predict_labels
from aide_amd.synthetic import chaos_batch

model = sys.argv[1] if len(sys.argv) > 1 else 'fuseunet'
slices = int(sys.argv[2]) if len(sys.argv) > 2 else 32
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 16
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device('cuda:0')
net = B.build(model, dev)
# a few training steps first: running statistics that are not the initial (0, 1)
net.train()
xin, xout, _ = chaos_batch(4, size, seed=7, single_modal=(model != 'fuseunet'))
with torch.no_grad():
    for _ in range(2):
        net(xin.to(dev), xout.to(dev)) if xout is not None else net(xin.to(dev))
net.eval()
xin, xout, _ = chaos_batch(slices, size, seed=1234, single_modal=(model != 'fuseunet'))
ins = [xin.to(dev)] + ([xout.to(dev)] if xout is not None else [])
for _ in range(3):
    lab = predict_labels(net, *ins, batch_size=batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    lab = predict_labels(net, *ins, batch_size=batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps(dict(metric='inference slices/sec %s %dx%d' % (model, size, size), value=round(slices / dt, 1),
                      ms_per_case=round(dt * 1e3, 3), slices=slices, batch=batch, foreground=float(lab.float().mean().item()),
                      switches={k: v for k, v in os.environ.items() if k.startswith('AIDE_')})))
