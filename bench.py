#!/usr/bin/env python
"""Headline benchmark: training images/sec of the FuseUNet hot path on N MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one optimizer step of the comparison loop (train_files/trainchaos_comparison_1case.py:
195-199): zero_grad, FuseUNet forward, CEMDiceLoss, backward, Adam(amsgrad) [+ gradient all-reduce],
on a synthetic CHAOS-shaped batch (bs 4/GPU, 2 x 3 x 256 x 256, fp32) already resident in HBM.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md chip table
BF16_MFMA_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak, same table
WORKLOAD_PRECISION = {'c5': 'bf16'}
WORKLOADS = {
    # name: (model, per-GPU batch, image size, fwd+bwd algorithmic GFLOP / image)
    'c2': ('fuseunet', 4, 256, 348.40),
    # AIDE proposed co-teaching step: two FuseUNets, per net 4 augmented forwards + 1 forward + 1 backward
    'c3': ('coteach', 4, 256, 1626.5),
    'c4': ('UNet', 4, 320, 612.32),
    'c2-512': ('fuseunet', 4, 512, 1393.58),
    # BASELINE config 5: FuseUNet bf16 MFMA path, 512x512 2-modal, bs=8/GPU (bf16 conv operands, fp32 accumulate,
    # fp32 master weights / BatchNorm statistics / loss / Adam)
    'c5': ('fuseunet', 8, 512, 1393.58),
    'tiny': ('fuseunet', 2, 64, 348.40 / 16),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)     # 50 x 7.3 ms: the one instrumented (single-stream) step costs 0.7 % of the line instead of 1.7 %
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--batch_size', type=int, default=None, help='per-GPU batch (default: workload)')
    ap.add_argument('--precision', default=None, choices=('fp32', 'bf16'),
                    help='conv arithmetic (default: fp32, bf16 for workload c5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=None,
                    help='oracle steps timed for cpu_baseline (default: 5 for c2 = about 10 s of CPU work, 2 for the larger workloads)')
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='do not bracket the MFMA conv launches with HIP events (roofline -> null)')
    ap.add_argument('--event-steps', type=int, default=1,
                    help='how many of the timed steps (the last ones) carry per-kernel HIP events; every '
                         'event pair costs ~30 us of queue bubbles, so instrumenting all steps would '
                         'distort `value` by >20 %%')
    ap.add_argument('--cpu-threads', type=int, default=32)
    return ap.parse_args()


def build(model_name, device):
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNet
    torch.manual_seed(2)                                  # reference default --torch_seed 2
    net = fuseunet(2) if model_name == 'fuseunet' else UNet(2)
    return net.to(device)


def cpu_baseline(model_name, batch, size, steps, max_threads):
    """The oracle (plain aten, CPU) timed on the host cores on the SAME workload, bounded sample.
    oneDNN stops scaling (and regresses badly) long before 256 threads on this shape, so the thread
    count is capped; `cores` reports the threads actually used."""
    import oracle
    from aide_amd.synthetic import chaos_batch
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(2)
    net = oracle.fuseunet(2) if model_name == 'fuseunet' else oracle.UNet(2)
    net.train()
    w = torch.tensor([1.0, 1.0])
    crit = oracle.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, amsgrad=True)
    xin, xout, t = chaos_batch(batch, size, seed=1234, single_modal=(model_name != 'fuseunet'))
    oracle.comparison_step(net, crit, opt, xin, xout, t)          # warm-up
    t0 = time.time()
    for _ in range(steps):
        oracle.comparison_step(net, crit, opt, xin, xout, t)
    dt = (time.time() - t0) / steps
    return dict(value=batch / dt, unit='images/sec', cores=torch.get_num_threads(), kind='port',
                sample='%d optimizer steps of the same %s bs=%d %dx%d fp32 step (oracle = aten CPU '
                       'restatement, bit-equal to the reference; 1 warm-up)' % (steps, model_name, batch, size, size),
                sec_per_step=dt)


def main_coteach(args, rank, world, device, batch, size, gflop_img):
    """BASELINE config 3 (not the headline): the AIDE proposed step; N>1 = data-parallel replicas with per-replica
    BatchNorm statistics and small-loss selection, both networks' gradients mean-all-reduced."""
    from aide_amd.optim import Adam
    from aide_amd.distributed import GradAllReduce, broadcast_module
    from aide_amd.synthetic import chaos_batch
    from aide_amd.utils import CoTeachingProposedLoss
    from aide_amd.train_files.trainchaos_proposed_30cases1labeled import coteach_step
    n1, n2 = build('fuseunet', device), build('fuseunet', device)
    n1.train(); n2.train()
    precision = args.precision or 'fp32'
    n1.engine.precision = n2.engine.precision = precision
    peak = BF16_MFMA_PEAK_TFLOPS if precision == 'bf16' else FP32_MFMA_PEAK_TFLOPS
    if world > 1:
        broadcast_module(n1); broadcast_module(n2)
    reducers = [GradAllReduce(n1), GradAllReduce(n2)] if world > 1 else None      # installed on the engines
    o1, o2 = Adam(n1.parameters(), lr=1e-4, amsgrad=True), Adam(n2.parameters(), lr=1e-4, amsgrad=True)
    op = CoTeachingProposedLoss(cediceweight=[1.0, 1.0], ceclassweight=[1.0, 1.0], segcor_weight=[1.0, 10.0])
    xin, xout, t = chaos_batch(batch, size, seed=1234 + rank)
    xin, xout, t = xin.to(device), xout.to(device), t.to(device)
    augs = [((xin * (1 + 0.05 * k)), (xout * (1 - 0.05 * k))) for k in range(4)]
    augset = {'augno': [4] * batch}
    for k in range(4):
        augset['hflip%d' % (k + 1)] = [(k + b) % 2 for b in range(batch)]
        augset['degree%d' % (k + 1)] = [15.0 * (k + 1) - 7.0 * b for b in range(batch)]

    def step():
        return coteach_step(n1, n2, o1, o2, op, xin, xout, augs, t, t, 0.25, augset=augset)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    if rank != 0:
        return
    value = batch * world * args.steps / el
    print(json.dumps(dict(metric='training images/sec AIDE co-teaching (2x FuseUNet) %dx%dx2 bs=%d/GPU' % (size, size, batch),
                          value=round(value, 2), unit='images/sec', n_gpus=world, steps=args.steps, warmup=args.warmup,
                          ms_per_step=round(el / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                          vs_baseline=None, dtype='f32' if precision == 'fp32' else 'bf16', data='synthetic',
                          config=dict(workload='c3 two fuseunet co-teaching step (4 aug fwd + fwd + bwd + Adam per net, '
                                               'on-device reverseaug, fused selection), %dx%d, bs=%d, %s' % (size, size, batch, precision),
                                      alg_gflop_per_image=gflop_img),
                          step_tflops=round(value * gflop_img / 1e3, 2),
                          step_mfma_frac=round(value * gflop_img / 1e3 / peak, 4),
                          final_loss=[round(float(r['loss1']), 6), round(float(r['loss2']), 6)],
                          roofline=None, cpu_baseline=None)))


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product path has no CPU fallback)')
    from aide_amd.distributed import init_from_env
    rank, world, device = init_from_env()      # one process per GPU; RCCL group when WORLD_SIZE > 1
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    from aide_amd import utils as U
    from aide_amd.optim import Adam
    from aide_amd.synthetic import chaos_batch
    from aide_amd.profiling import KernelTimer
    from aide_amd.distributed import GradAllReduce, broadcast_module

    model_name, batch, size, gflop_img = WORKLOADS[args.workload]
    if args.batch_size:
        batch = args.batch_size
    if model_name == 'coteach':
        main_coteach(args, rank, world, device, batch, size, gflop_img)
        if world > 1:
            dist.destroy_process_group()
        return
    net = build(model_name, device)
    net.train()
    precision = args.precision or WORKLOAD_PRECISION.get(args.workload, 'fp32')
    net.engine.precision = precision
    peak = BF16_MFMA_PEAK_TFLOPS if precision == 'bf16' else FP32_MFMA_PEAK_TFLOPS
    if world > 1:
        broadcast_module(net)
    reducer = GradAllReduce(net) if world > 1 else None
    w = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    xin, xout, tgt = chaos_batch(batch, size, seed=1234 + rank, single_modal=(model_name != 'fuseunet'))
    xin, tgt = xin.to(device), tgt.to(device)
    xout = xout.to(device) if xout is not None else None

    def step():
        opt.zero_grad()
        out = net(xin, xout) if xout is not None else net(xin)
        loss = crit(out, tgt)
        loss.backward()
        opt.step()
        return loss

    for w_i in range(args.warmup):
        if w_i == args.warmup - 1 and not args.no_kernel_events:
            net.engine.profiler = KernelTimer(reserve=256)      # the instrumented code path warms up outside the timed region
        step()
        net.engine.profiler = None
    ev_steps = min(args.event_steps, args.steps)
    timer = None if args.no_kernel_events else KernelTimer(reserve=256 * max(ev_steps, 1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timer is not None and i == args.steps - ev_steps:
            net.engine.profiler = timer           # HIP-event pairs around the MFMA conv launches
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    net.engine.profiler = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = batch * world * args.steps / elapsed
        roof = None
        kernels = {}
        if timer is not None:
            agg = timer.summary()
            for k, a in agg.items():
                kernels[k] = dict(launches=a['launches'], avg_ms=round(a['avg_ms'], 5),
                                  total_ms_per_step=round(a['ms'] / ev_steps, 4),
                                  tflops=round(a['tflops'], 2), executed_tflops=round(a['executed_tflops'], 2))
            if agg:
                dom = max(agg.items(), key=lambda kv: kv[1]['ms'])
                a = dom[1]
                traffic = None
                tpath = os.path.join(ROOT, 'profiles', {'c2': 'r01_traffic.json', 'c5': 'r01_traffic_c5.json'}.get(
                    args.workload, 'none'))
                if os.path.exists(tpath):
                    # PMC counters cannot be read from inside this process: the per-launch HBM bytes of
                    # this kernel family come from the committed rocprofv3 --pmc passes (profiles/)
                    traffic = json.load(open(tpath)).get(dom[0], {}).get('hbm_bytes_per_launch')
                roof = dict(bound='mfma', kernel=dom[0], achieved=round(a['tflops'], 2),
                            peak=peak, unit='TFLOP/s',
                            frac=round(a['tflops'] / peak, 4), traffic=traffic,
                            # Winograd launches execute 16/36 of the algorithmic multiplies: `achieved`
                            # (algorithmic, as the contract defines it) may exceed the MFMA peak,
                            # `executed` is what the MFMA pipe really sustains
                            executed=round(a['executed_tflops'], 2),
                            executed_frac=round(a['executed_tflops'] / peak, 4),
                            launches_per_step=a['launches'] // ev_steps,
                            avg_launch_ms=round(a['avg_ms'], 5),
                            alg_gflop_per_launch=round(a['flops'] / a['launches'] / 1e9, 3))
                # Kernels that live on the side stream are launched with fewer workgroups than CUs on purpose (the F(4x4)
                # weight gradient: 128, the bf16 one: 192 -- DESIGN.md 4.6): alone in the instrumented step they run on
                # that share of the chip, so `frac` (against the whole chip's peak, as the contract defines it) is shown
                # next to the fraction of the share they occupy; `next` is the second-largest MFMA kernel family.
                share = {'conv3x3_wgrad4_kernel': 128.0 / 256.0, 'conv3x3_wgrad_bf16_kernel': 192.0 / 256.0}.get(dom[0])
                if share is not None:
                    roof['chip_share'] = share
                    roof['frac_of_share'] = round(a['tflops'] / peak / share, 4)
                    roof['executed_frac_of_share'] = round(a['executed_tflops'] / peak / share, 4)
                rest = sorted(((k, v) for k, v in agg.items() if k != dom[0]), key=lambda kv: -kv[1]['ms'])
                if rest:
                    k2, a2 = rest[0]
                    roof['next'] = dict(kernel=k2, achieved=round(a2['tflops'], 2), frac=round(a2['tflops'] / peak, 4),
                                        executed_frac=round(a2['executed_tflops'] / peak, 4),
                                        launches_per_step=a2['launches'] // ev_steps, avg_launch_ms=round(a2['avg_ms'], 5))
        cpu = None
        cpu_steps = args.cpu_steps if args.cpu_steps is not None else (5 if args.workload in ('c2', 'tiny') else 2)
        if world == 1 and not args.no_cpu_baseline and cpu_steps > 0:
            cpu = cpu_baseline(model_name, batch, size, cpu_steps, args.cpu_threads)
        line = dict(metric='training images/sec %s bs=%d/GPU' % (
                        ('FuseUNet %dx%dx2' if model_name == 'fuseunet' else 'UNet %dx%d') % (size, size), batch),
                    value=round(value, 2), unit='images/sec', n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype='f32' if precision == 'fp32' else 'bf16',
                    data='synthetic',
                    config=dict(workload='%s %s fwd+loss+bwd+Adam(amsgrad), %dx%d %s, bs=%d/GPU, %s'
                                         % (args.workload, model_name, size, size,
                                            '2-modal' if model_name == 'fuseunet' else '1-modal', batch,
                                            'fp32' if precision == 'fp32' else
                                            'bf16 conv operands / fp32 accumulate, fp32 BN+loss+Adam'),
                                global_batch=batch * world, parallelism='dp%d' % world,
                                alg_gflop_per_image=gflop_img),
                    step_tflops=round(value * gflop_img / 1e3, 2),
                    step_mfma_frac=round(value * gflop_img / 1e3 / world / peak, 4),
                    final_loss=round(final_loss, 6), roofline=roof, kernels=kernels, cpu_baseline=cpu)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
