"""What the data-parallel plumbing costs a rank's own step: the C2 step on ONE GPU with the bucketed gradient all-reduce
installed on a single-rank RCCL group (the collectives are local, but the bucket hooks, the communication streams and RCCL's
kernels are all there -- and so is the mapping of six streams onto the runtime's hardware queues).
usage (GPU box): python tools/bench_comm1.py [none|rccl] [workload] [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench as B
from aide_amd import utils as U
from aide_amd.optim import Adam
from aide_amd.synthetic import chaos_batch
from aide_amd.distributed import GradAllReduce, broadcast_module

mode = sys.argv[1] if len(sys.argv) > 1 else 'rccl'
wl = sys.argv[2] if len(sys.argv) > 2 else 'c2'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
model, batch, size, _ = B.WORKLOADS[wl]
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
red = None
if mode == 'rccl':
    if os.environ.get('AIDE_RESERVE_QUEUE', '1') != '0':          # what distributed.init_from_env does (A-B: 0)
        from aide_amd import streams
        streams.reserve_queue(dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
net = B.build(model, dev); net.train()
net.engine.precision = B.WORKLOAD_PRECISION.get(wl, 'fp32')
if mode == 'rccl':
    broadcast_module(net)
    red = GradAllReduce(net, force=True)
    red.time_exposed = True
w = torch.tensor([1.0, 1.0]); crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
xin, xout, t = chaos_batch(batch, size, seed=1234, single_modal=(model != 'fuseunet'))
xin, t = xin.to(dev), t.to(dev); xout = xout.to(dev) if xout is not None else None
def step():
    opt.zero_grad(); out = net(xin, xout) if xout is not None else net(xin)
    loss = crit(out, t); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(json.dumps(dict(mode=mode, workload=wl, 
                      images_per_s=round(batch / dt, 2), ms_per_step=round(dt * 1e3, 3),
                      exposed_ms=round(red.exposed_ms(), 4) if red else None,
                      buckets=red.describe()['buckets'] if red else None,
                      hw_queues=red.describe()['hw_queues'] if red else None)))
if mode == 'rccl':
    dist.destroy_process_group()
