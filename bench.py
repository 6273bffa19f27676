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
    # learned_bilinear=True variant (ConvTranspose2d(2,2) up path, netblocks.py:12): 245.32 GF / image
    'c2-learned': ('fuseunet_learned', 4, 256, 245.32),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--batch_size', type=int, default=None, help='per-GPU batch (default: workload)')
    ap.add_argument('--precision', default=None, choices=('fp32', 'bf16'),
                    help='conv arithmetic (default: fp32, bf16 for workload c5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=None,
                    help='oracle steps timed for cpu_baseline (default: 5 for c2 = about 10 s of CPU work, 2 for the larger workloads)')
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='do not time the MFMA conv dispatches with HIP events (roofline -> null)')
    ap.add_argument('--event-steps', type=int, default=None,
                    help='how many of the timed steps (the last ones) carry dispatch start/stop events: a launch with '
                         'events opens a ~7 us gap in its queue (the kernel durations themselves are unchanged), an '
                         'instrumented C2 step runs ~6 %% longer: 3 of 50 steps cost the line 0.2-0.3 %%, the 8 of '
                         'rounds 2-4 cost 0.7-1.0 %% (same-box A/B, HISTORY §11.7).  Default: 3, but at most one timed step in ten')
    ap.add_argument('--traffic', default='live', choices=('live', 'none'),
                    help="roofline.traffic: 'live' = two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, separately, "
                         "--kernel-trace only) of this workload run as sub-processes after the timed region (rank 0, N=1)")
    ap.add_argument('--traffic-timeout', type=int, default=150)
    ap.add_argument('--cpu-threads', type=int, default=16,
                    help='torch threads of the cpu_baseline leg: the fastest count of the sweep over 8 .. 256 threads on the '
                         '256-thread host of the GPU box (tools/cpu_thread_sweep.py, profiles/r04_cpu_threads_c2.txt: 16 -> 2.46 '
                         'images/s, 32 -> 2.02, 64 -> 1.10, 128 -> 0.63, 256 -> 125 s per step)')
    ap.add_argument('--allow-probes', action='store_true',
                    help='run although AIDE_HIP_LIB selects another library build; the line is then marked INVALID (A-B tooling only)')
    return ap.parse_args()


KNOWN_SWITCHES = ('AIDE_HIP_LIB', 'AIDE_DIST_BACKEND', 'AIDE_PICK_STREAMS', 'AIDE_REPLAY', 'AIDE_DIRECT_GRADS', 'AIDE_ABI_COVERAGE')


def switches():
    """every AIDE_* variable in this process' environment (echoed in the line) and the ones that make the run something
    other than the shipped product: AIDE_HIP_LIB selects another library build (tools/probes/mk_probe.py: ablation builds
    skip work or compute wrong results).  The package reads exactly KNOWN_SWITCHES; anything else is a typo or a switch of
    an older tree that would silently do nothing."""
    act = {k: v for k, v in sorted(os.environ.items()) if k.startswith('AIDE_')}
    unknown = sorted(k for k in act if k not in KNOWN_SWITCHES)
    probes = sorted(k for k in act if k == 'AIDE_HIP_LIB' and act[k])
    return act, probes, unknown


def probe_guard(args):
    act, probes, unknown = switches()
    if unknown:
        raise SystemExit('bench.py: unknown switch(es) %s -- this tree reads only %s' % (', '.join(unknown), ', '.join(KNOWN_SWITCHES)))
    if probes and not args.allow_probes:
        raise SystemExit('bench.py: refusing to run with %s set (a non-default library build: ablation builds skip work or '
                         'compute wrong results, the line would not be a measurement; --allow-probes marks it INVALID instead)'
                         % ', '.join(probes))
    return act, probes


def event_steps(args):
    """how many of the timed steps carry dispatch events: --event-steps, but (unless given explicitly) at most one step in
    ten -- an instrumented step runs ~6 % longer, and a short run should not pay for its own measurement"""
    if args.event_steps is not None:
        return max(1, min(args.event_steps, args.steps))
    return max(1, min(3, args.steps // 10))


def timed_steps(step, args, world, device, timer, ev_steps):
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides.  -> (max over ranks of the
    elapsed seconds, this rank's elapsed seconds, [per-rank seconds], last step's return value)"""
    for _ in range(args.warmup):
        step()
    if timer is not None:
        timer.start()                             # creates the events: outside the timed region
        timer.stop()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        if timer is not None and i == args.steps - ev_steps:
            timer.arm()                           # the last ev_steps timed steps: their MFMA conv launches carry events
        out = step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0            # this rank's own K steps (before it waits for the others)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if timer is not None:
        timer.stop()
    per_rank = [t_local]
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([t_local], device=device, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(v.item()) for v in allr]
    return elapsed, t_local, per_rank, out


def replica_check(nets, reducers, world, device):
    """N > 1, after the timed region (collectives on every rank): do all ranks hold the same bucket plan, and are the
    replicas still bit-identical after the optimizer steps?  (exact: the parameters' bit patterns summed as integers)"""
    if world == 1:
        return {}
    plans = [tuple((s, e, tuple(i)) for s, e, i in r.sched.buckets) for r in reducers]
    gathered = [None] * world
    dist.all_gather_object(gathered, plans)
    sums = []
    for net in nets:
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).view(torch.int32)
        sums.append(flat.to(torch.int64).sum())
    mine = torch.stack(sums).to(device)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return dict(bucket_plan_identical=all(g == gathered[0] for g in gathered),
                replicas_identical=all(bool(torch.equal(a, allr[0])) for a in allr))


def comm_block(world, reducers, per_rank, steps, check=None):
    """N > 1: what was exchanged and what of it the step had to wait for."""
    from aide_amd.distributed import comm_environment, first_contact
    if world == 1:
        return dict(backend=None, ranks=1, env=comm_environment())
    desc = [r.describe() for r in reducers]
    exposed = [r.exposed_ms() for r in reducers]
    try:
        ver = '.'.join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == 'nccl' else None
    except Exception:
        ver = None
    ms = [t / steps * 1e3 for t in per_rank]
    return dict(backend=dist.get_backend(), ranks=world, rccl_version=ver,
                buckets=sum(d['buckets'] for d in desc), bytes_per_step=sum(d['bytes_per_step'] for d in desc),
                env=comm_environment(), **(check or {}),
                # link topology, RCCL's channel count, stream fallback (multi-GPU: first contact with real peers)
                first_contact=first_contact() if dist.get_rank() == 0 else None,
                hw_queues=next((d['hw_queues'] for d in desc if d.get('hw_queues')), None),
                exposed_ms=(round(sum(e for e in exposed if e is not None), 4)
                            if any(e is not None for e in exposed) else None),
                exposed_note='mean over the last 8 timed steps of the event-timed wait of the compute stream for the '
                             'all-reduces still in flight at the end of a backward pass (rank 0; summed over the models)',
                rank_ms_per_step=dict(min=round(min(ms), 3), max=round(max(ms), 3)))


def roofline_block(timer, ev_steps, steps, peak, args, precision, batch, world):
    """`roofline` + per-family `kernels` from the dispatch timer."""
    agg = timer.summary()
    kernels = {}
    for k, a in agg.items():
        kernels[k] = dict(launches_per_step=round(a['launches'] / ev_steps, 2), avg_us=round(a['avg_ms'] * 1e3, 2),
                          max_us=round(a['max_ms'] * 1e3, 2), total_ms_per_step=round(a['ms'] / ev_steps, 4),
                          tflops=round(a['tflops'], 2), executed_tflops=round(a['executed_tflops'], 2))
    if not agg:
        return None, kernels
    # dominant kernel = the MFMA family with the largest summed dispatch time over the timed steps
    dom = max(agg.items(), key=lambda kv: kv[1]['ms'])
    a = dom[1]
    traffic, tnote = None, 'not measured (--traffic none)'
    if args.traffic == 'live' and world == 1:
        traffic, tnote = measure_traffic(args, dom[0], precision, batch)
    elif world > 1:
        tnote = 'measured at N=1 only'
    roof = dict(bound='mfma', kernel=dom[0], achieved=round(a['tflops'], 2),
                peak=peak, unit='TFLOP/s',
                # `achieved` / `frac` count ALGORITHMIC (direct-convolution) flop per launch, as the contract defines it;
                # a Winograd F(4x4) launch executes 36/144 of those multiplies (F(2x2): 16/36), so `frac` may exceed 1.
                # `executed` / `executed_frac` is what the MFMA pipe really sustains: the hardware-utilisation number.
                frac=round(a['tflops'] / peak, 4),
                executed_frac=round(a['executed_tflops'] / peak, 4),
                executed=round(a['executed_tflops'], 2),
                traffic=traffic, traffic_source=tnote,
                launches_per_step=round(a['launches'] / ev_steps, 2),
                avg_launch_us=round(a['avg_ms'] * 1e3, 2),
                alg_gflop_per_launch=round(a['flops'] / a['launches'] / 1e9, 3),
                timing='hipExtLaunchKernelGGL start/stop events on the launch stream, every launch of the last %d of '
                       'the %d timed steps, multi-stream schedule (dispatch begin..end, as rocprofv3 --kernel-trace)'
                       % (ev_steps, steps),
                dropped_launches=timer.dropped)
    rest = sorted(((k, v) for k, v in agg.items() if k != dom[0]), key=lambda kv: -kv[1]['ms'])
    if rest:
        k2, a2 = rest[0]
        roof['next'] = dict(kernel=k2, achieved=round(a2['tflops'], 2), frac=round(a2['tflops'] / peak, 4),
                            executed_frac=round(a2['executed_tflops'] / peak, 4),
                            launches_per_step=round(a2['launches'] / ev_steps, 2),
                            avg_launch_us=round(a2['avg_ms'] * 1e3, 2))
    # all MFMA conv dispatches of a step together: executed multiplies / summed dispatch time
    tot_ms = sum(v['ms'] for v in agg.values())
    roof['all_mfma_kernels'] = dict(
        sum_dispatch_ms_per_step=round(tot_ms / ev_steps, 3),
        executed_tflops=round(sum(v['executed'] for v in agg.values()) / (tot_ms * 1e-3) / 1e12, 2),
        executed_frac=round(sum(v['executed'] for v in agg.values()) / (tot_ms * 1e-3) / 1e12 / peak, 4))
    return roof, kernels


QUEUE_STEPS = 3


def queue_blocks(step, args, device, peak, ms_step, world):
    """`roofline.critical` + `streaming`, measured AFTER the timed region on QUEUE_STEPS extra steps in
    which EVERY kernel family of the C ABI carries its dispatch's start / stop events (aide_ktimer_dump: begin, end, launch
    stream of every launch): which launch stream bounds the step, what it is busy with, and what the HBM-bound families
    (BatchNorm, pooling, up-sampling) stream against 8 TB/s.  Every rank runs the steps (collectives), rank 0 reports."""
    if args.no_kernel_events:
        return None, None
    from aide_amd.profiling import DispatchTimer, KT_ALL, queue_report
    from aide_amd import engine as _eng
    t = DispatchTimer(capacity=1600 * QUEUE_STEPS, families=KT_ALL)
    t.start()
    t.stop()
    step()                                        # (a step between the timed region and the window: nothing of it is recorded)
    torch.cuda.synchronize()
    t.arm()
    for _ in range(QUEUE_STEPS):
        step()
    torch.cuda.synchronize()
    t.stop()
    rows = t.timeline()
    names = {torch.cuda.current_stream().cuda_stream: 'main'}
    pref = _eng._preferred(device)
    for k, label in (('side', 'weight-gradient'), ('lane', 'lane')):
        if pref.get(k) is not None:
            names.setdefault(pref[k].cuda_stream, label)
    critical, streaming = queue_report(rows, QUEUE_STEPS, names, peak)
    if critical is not None:
        # the instrumented steps run longer (every event pair opens a few us behind its dispatch): the gaps of the critical
        # stream are priced against the UNINSTRUMENTED step of the timed region
        critical['gaps_ms_per_step'] = round(ms_step - critical['busy_ms_per_step'], 4)
        critical['ms_per_step'] = round(ms_step, 4)
        critical['timing'] = ('dispatch begin / end events of every launch of %d extra steps after the timed region, all '
                              'kernel families armed; busy = summed dispatch time per launch stream' % QUEUE_STEPS)
        critical['dropped_launches'] = t.dropped
    return critical, streaming


def build(model_name, device):
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNet
    torch.manual_seed(2)                                  # reference default --torch_seed 2
    if model_name == 'fuseunet_learned':
        net = fuseunet(2, learned_bilinear=True)
    else:
        net = fuseunet(2) if model_name == 'fuseunet' else UNet(2)
    return net.to(device)


def cpu_baseline(model_name, batch, size, steps, max_threads):
    """The oracle (plain aten, CPU) timed on the host cores on the SAME workload, bounded sample.
    oneDNN stops scaling (and regresses badly) long before 256 threads on this shape -- the sweep is in
    profiles/r04_cpu_threads_c2.txt -- so the thread count is the fastest one measured; `cores` reports the
    threads actually used, `host_cores` what the box has."""
    import oracle
    from aide_amd.synthetic import chaos_batch
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(2)
    net = (oracle.fuseunet(2, learned_bilinear=True) if model_name == 'fuseunet_learned' else
           oracle.fuseunet(2) if model_name == 'fuseunet' else oracle.UNet(2))
    net.train()
    w = torch.tensor([1.0, 1.0])
    crit = oracle.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, amsgrad=True)
    xin, xout, t = chaos_batch(batch, size, seed=1234, single_modal=(not model_name.startswith('fuseunet')))
    oracle.comparison_step(net, crit, opt, xin, xout, t)          # warm-up
    t0 = time.time()
    for _ in range(steps):
        oracle.comparison_step(net, crit, opt, xin, xout, t)
    dt = (time.time() - t0) / steps
    return dict(value=batch / dt, unit='images/sec', cores=torch.get_num_threads(), host_cores=os.cpu_count(),
                kind='port',
                sample='%d optimizer steps of the same %s bs=%d %dx%d fp32 step (oracle = aten CPU '
                       'restatement, bit-equal to the reference; 1 warm-up)' % (steps, model_name, batch, size, size),
                sec_per_step=dt)


def measure_traffic(args, kernel, precision, batch):
    """HBM bytes per launch of `kernel` from the PMC counters, collected as MI355X_MICROARCH.md's HBM / rocprofv3 section
    prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (with --kernel-trace only), raw unit KiB,
    FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes; calibrated on this stack
    for 16-byte-per-lane reads in 64 / 128 / 256 / 1024-byte segments -- the row pieces the conv kernels read -- by
    tools/fetch_calib.sh: FETCH_SIZE = 0.500 x bytes, WRITE_SIZE = 1.00-1.03 x bytes, profiles/r03_fetch_calib.txt).  Each pass profiles a
    short run (1 warm-up + 2 steps) of this same workload in a sub-process.  -> (bytes per launch | None, note)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(('ROCPROFILER_', 'ROCP_')) for k in os.environ):
        return None, 'skipped: this process already runs under a rocprofiler tool'
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return None, 'rocprofv3 not found'
    per = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='aide_pmc_', dir='/tmp')
        cmd = [rocprof, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
               sys.executable, os.path.abspath(__file__), '--workload', args.workload, '--precision', precision,
               '--batch_size', str(batch), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-kernel-events',
               '--traffic', 'none']
        env = dict(os.environ, TMPDIR='/tmp')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, timeout=args.traffic_timeout, capture_output=True, text=True)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, '%s pass timed out' % ctr
        files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        if r.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, '%s pass failed (rc %d)' % (ctr, r.returncode)
        n, kb = 0, 0.0
        for f in files:
            for row in csv.DictReader(open(f)):
                if row.get('Counter_Name') == ctr and kernel in row.get('Kernel_Name', ''):
                    n += 1
                    kb += float(row['Counter_Value'])
        shutil.rmtree(d, ignore_errors=True)
        if n == 0:
            return None, 'kernel %s not found in the %s pass' % (kernel, ctr)
        per[ctr] = (kb * 1024.0 / n, n)          # the counter unit is KiB (tools/fetch_calib.sh: WRITE_SIZE = 0.977 x bytes / 1000)
    fetch, write = 2.0 * per['FETCH_SIZE'][0], per['WRITE_SIZE'][0]
    return int(fetch + write), ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over %d / %d launches of a 3-step run of '
                                'this workload: fetch %.1f MB (raw x2, gfx950 correction) + write %.1f MB per launch'
                                % (per['FETCH_SIZE'][1], per['WRITE_SIZE'][1], fetch / 1e6, write / 1e6))


def cpu_baseline_coteach(batch, size, steps, max_threads, augset):
    """The oracle's proposed step (oracle/steps.py::proposed_step = trainchaos_proposed_30cases1labeled.py:260-325, with
    its PIL reverse augmentation) timed on the host cores on the SAME synthetic batch: 1 warm-up + `steps` steps."""
    import oracle
    from oracle.steps import reverseaug as pil_reverseaug
    from aide_amd.synthetic import chaos_batch
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(2)
    n1 = oracle.fuseunet(2)
    torch.manual_seed(2)
    n2 = oracle.fuseunet(2)
    n1.train(); n2.train()
    w = torch.tensor([1.0, 1.0])
    crit = oracle.CEMDiceLossImage(cediceweight=w, ceclassweight=w, diceclassweight=w)
    corr = oracle.MulticlassMSELoss(reduction='none')
    o1 = torch.optim.Adam(n1.parameters(), lr=1e-4, amsgrad=True)
    o2 = torch.optim.Adam(n2.parameters(), lr=1e-4, amsgrad=True)
    xin, xout, t = chaos_batch(batch, size, seed=1234)
    augs = [((xin * (1 + 0.05 * k)), (xout * (1 - 0.05 * k))) for k in range(4)]
    rev = lambda lst: pil_reverseaug(augset, lst, 2)

    def step():
        return oracle.proposed_step(n1, n2, crit, corr, o1, o2, xin, xout, augs, t, t, 0.25, reverse=rev)
    step()                                                        # warm-up
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = (time.time() - t0) / steps
    return dict(value=batch / dt, unit='images/sec', cores=torch.get_num_threads(), host_cores=os.cpu_count(),
                kind='port',
                sample='%d co-teaching step(s) of the same two-fuseunet bs=%d %dx%d fp32 workload (oracle = aten CPU + PIL '
                       'restatement of the proposed inner loop, pinned bit-equal to the reference; 1 warm-up)'
                       % (steps, batch, size, size),
                sec_per_step=dt)


PIPELINE_C3 = [True]       # coteach_step(pipeline=True): the train script's own setting (train_files/trainchaos_proposed_30cases1labeled.py Train)


def main_coteach(args, rank, world, device, batch, size, gflop_img, act, probes):
    """BASELINE config 3 (not the headline): the AIDE proposed step; N>1 = data-parallel replicas with per-replica
    BatchNorm statistics and small-loss selection, both networks' gradients mean-all-reduced."""
    from aide_amd.optim import Adam
    from aide_amd.distributed import GradAllReduce, broadcast_module
    from aide_amd.profiling import DispatchTimer
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
    reducers = [GradAllReduce(n1), GradAllReduce(n2)] if world > 1 else []      # installed on the engines
    for r in reducers:
        r.time_exposed = True
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
        return coteach_step(n1, n2, o1, o2, op, xin, xout, augs, t, t, 0.25, augset=augset, pipeline=PIPELINE_C3[0])
    ev_steps = event_steps(args)
    timer = None if args.no_kernel_events else DispatchTimer(capacity=1200 * ev_steps)
    el, _, per_rank, r = timed_steps(step, args, world, device, timer, ev_steps)
    final = [round(float(r['loss1']), 6), round(float(r['loss2']), 6)]
    check = replica_check([n1, n2], reducers, world, device)
    if timer is not None:
        timer.summary()                           # (read before queue_blocks re-arms the library's one timer)
    critical, streaming = queue_blocks(step, args, device, peak, el / args.steps * 1e3, world)
    if rank != 0:
        return
    value = batch * world * args.steps / el
    roof, kernels = (None, {}) if timer is None else roofline_block(timer, ev_steps, args.steps, peak, args, precision,
                                                                    batch, world)
    if roof is not None:
        roof['critical'] = critical
    cpu = None
    cpu_steps = args.cpu_steps if args.cpu_steps is not None else 1
    if world == 1 and not args.no_cpu_baseline and cpu_steps > 0:
        cpu = cpu_baseline_coteach(batch, size, cpu_steps, args.cpu_threads, augset)
    line = dict(metric='training images/sec AIDE co-teaching (2x FuseUNet) %dx%dx2 bs=%d/GPU' % (size, size, batch),
                value=round(value, 2), unit='images/sec', n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=round(el / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                vs_baseline=None, dtype='f32' if precision == 'fp32' else 'bf16', data='synthetic',
                config=dict(workload='c3 two fuseunet co-teaching step (4 aug fwd + fwd + bwd + Adam per net, '
                                     'on-device reverseaug, fused selection%s), %dx%d, bs=%d/GPU, %s'
                                     % (', network 2 pipelined across the step boundary' if PIPELINE_C3[0] else '',
                                        size, size, batch, precision),
                            global_batch=batch * world, parallelism='dp%d' % world, alg_gflop_per_image=gflop_img),
                step_tflops=round(value * gflop_img / 1e3, 2),
                step_algorithmic_frac=round(value * gflop_img / 1e3 / world / peak, 4),
                mfma_executed_frac=None if roof is None else roof['all_mfma_kernels']['executed_frac'],
                final_loss=final, comm=comm_block(world, reducers, per_rank, args.steps, check), switches=act,
                roofline=roof, streaming=streaming, kernels=kernels, cpu_baseline=cpu)
    if probes:
        line['INVALID'] = 'non-default library build: %s' % ', '.join(probes)
    print(json.dumps(line))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec this command as N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 (what the driver does itself for N > 1).  The rank-0 child prints the JSON line."""
    import socket
    import subprocess
    backend = os.environ.get('AIDE_DIST_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and ndev < args.gpus:
        raise SystemExit('bench.py: --gpus %d but only %d HIP device(s) visible (RCCL needs one device per rank; '
                         'AIDE_DIST_BACKEND=gloo is the dry-run backend for boxes with fewer GPUs)' % (args.gpus, ndev))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    act, probes = probe_guard(args)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product path has no CPU fallback)')
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    from aide_amd.distributed import init_from_env
    rank, world, device = init_from_env(rccl_log=True)      # one process per GPU; RCCL group when WORLD_SIZE > 1
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from aide_amd import utils as U
    from aide_amd.optim import Adam
    from aide_amd.synthetic import chaos_batch
    from aide_amd.profiling import DispatchTimer
    from aide_amd.distributed import GradAllReduce, broadcast_module

    model_name, batch, size, gflop_img = WORKLOADS[args.workload]
    if args.batch_size:
        batch = args.batch_size
    if model_name == 'coteach':
        main_coteach(args, rank, world, device, batch, size, gflop_img, act, probes)
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
    reducers = [GradAllReduce(net)] if world > 1 else []
    for r in reducers:
        r.time_exposed = True
    w = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    xin, xout, tgt = chaos_batch(batch, size, seed=1234 + rank, single_modal=(not model_name.startswith('fuseunet')))
    xin, tgt = xin.to(device), tgt.to(device)
    xout = xout.to(device) if xout is not None else None

    def step():
        opt.zero_grad()
        out = net(xin, xout) if xout is not None else net(xin)
        loss = crit(out, tgt)
        loss.backward()
        opt.step()
        return loss

    # every launch of every MFMA conv kernel in the timed steps carries a start / stop event pair holding the dispatch's
    # own begin / end timestamps (C ABI kernel timer): the steps keep their multi-stream schedule, nothing is serialised
    ev_steps = event_steps(args)
    timer = None if args.no_kernel_events else DispatchTimer(capacity=200 * ev_steps)
    elapsed, _, per_rank, loss = timed_steps(step, args, world, device, timer, ev_steps)
    final_loss = float(loss.item())
    check = replica_check([net], reducers, world, device)
    if timer is not None:
        timer.summary()                           # (read before queue_blocks re-arms the library's one timer)
    critical, streaming = queue_blocks(step, args, device, peak, elapsed / args.steps * 1e3, world)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = batch * world * args.steps / elapsed
        roof, kernels = (None, {}) if timer is None else roofline_block(timer, ev_steps, args.steps, peak, args,
                                                                        precision, batch, world)
        if roof is not None:
            roof['critical'] = critical
        cpu = None
        cpu_steps = args.cpu_steps if args.cpu_steps is not None else (5 if args.workload in ('c2', 'tiny') else 2)
        if world == 1 and not args.no_cpu_baseline and cpu_steps > 0:
            cpu = cpu_baseline(model_name, batch, size, cpu_steps, args.cpu_threads)
        line = dict(metric='training images/sec %s bs=%d/GPU' % (
                        ('FuseUNet %dx%dx2' if model_name.startswith('fuseunet') else 'UNet %dx%d') % (size, size), batch),
                    value=round(value, 2), unit='images/sec', n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype='f32' if precision == 'fp32' else 'bf16',
                    data='synthetic',
                    config=dict(workload='%s %s fwd+loss+bwd+Adam(amsgrad), %dx%d %s, bs=%d/GPU, %s'
                                         % (args.workload, model_name, size, size,
                                            '2-modal' if model_name.startswith('fuseunet') else '1-modal', batch,
                                            'fp32' if precision == 'fp32' else
                                            'bf16 conv operands / fp32 accumulate, fp32 BN+loss+Adam'),
                                global_batch=batch * world, parallelism='dp%d' % world,
                                alg_gflop_per_image=gflop_img),
                    # algorithmic flop of the whole step / time / peak (> 1 is possible: Winograd executes 1/4 of them)
                    step_tflops=round(value * gflop_img / 1e3, 2),
                    step_algorithmic_frac=round(value * gflop_img / 1e3 / world / peak, 4),
                    # executed multiplies of ALL MFMA conv dispatches / their summed dispatch time / peak: utilisation
                    mfma_executed_frac=None if roof is None else roof['all_mfma_kernels']['executed_frac'],
                    final_loss=round(final_loss, 6),
                    comm=comm_block(world, reducers, per_rank, args.steps, check), switches=act,
                    roofline=roof, streaming=streaming, kernels=kernels, cpu_baseline=cpu)
        if probes:
            line['INVALID'] = 'non-default library build: %s' % ', '.join(probes)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
