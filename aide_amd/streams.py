"""Which HIP streams really run side by side?

The HIP runtime maps the streams of a process onto GPU_MAX_HW_QUEUES (4) hardware queues and a queue runs its packets in
order, so two "concurrent" streams that land in one queue serialise -- and which streams share is decided by creation order
inside the runtime (tools/ubench/hwq_probe.py: with an RCCL communicator in the process the FIRST stream created afterwards
shares the default stream's queue).  A single process gets away with it (main, weight gradients, second forward lane and
filter packs take one queue each); a data-parallel rank does not: RCCL's own stream takes a queue, and whether the weight-
gradient stream, the lane or RCCL then share the main stream's queue is luck (measured on one MI355X, single-rank group:
608 -> 567 images/s with RCCL's kernels in the main stream's queue, 466 with the weight-gradient stream there).

So a data-parallel rank measures instead of hoping: `pick()` classifies a handful of candidate streams by hardware queue
with a spin kernel (a long spin on one stream, a tiny kernel on the other: does it finish first?), finds the queue RCCL's
stream lives in the same way (a tiny all-reduce beside a spin), and hands the engine a weight-gradient stream that shares
neither the main stream's queue nor RCCL's, and a lane stream away from both compute streams.  Costs ~50 ms once per process."""
import torch

_PLACEHOLDERS = []


def reserve_queue(device):
    """Call BEFORE the process group's first collective.  A stream that has been created and used before RCCL's own stream
    exists makes the runtime put RCCL's stream into a hardware queue of its own; without it RCCL's stream shares the
    DEFAULT stream's queue (tools/bench_comm1.py, single-rank group on one MI355X: 571 vs 599 images/s per rank, 613 without
    any data-parallel plumbing -- every all-reduce kernel and every event wait in front of it sat in the main stream's
    queue).  pick() verifies the outcome and reports it in the bench line."""
    s = torch.cuda.Stream(device=device)
    with torch.cuda.stream(s):
        torch.zeros(8, device=device).add_(1)
    s.synchronize()
    _PLACEHOLDERS.append(s)              # kept for the life of the process
    return s


PREFERRED = {}          # device index -> dict(side=Stream, lane=Stream): what engine.Plan uses instead of fresh streams
_SPIN = 3_000_000       # torch.cuda._sleep cycles: ~1.5 ms
_BESIDE = 0.25          # "beside" = the probe was done at least this fraction of the spin's duration before the spin ended


def _beside(spin_stream, probe, spin=_SPIN):
    """probe(): enqueue something tiny, return the torch.cuda.Event(enable_timing=True) recorded right behind it.  True when
    it completed while `spin_stream` was still spinning (different hardware queues).  Decided on DEVICE timestamps (the
    probe's completion against the spin's begin / end events), not on host wall-clock: host jitter, or a collective that
    waits for a late peer, cannot turn "ran beside" into "waited" -- only a probe that really finished before the spin did
    counts as beside."""
    torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(spin_stream):
        b.record()
        torch.cuda._sleep(spin)
        e.record()
    done = probe()
    torch.cuda.synchronize()
    return done.elapsed_time(e) > _BESIDE * b.elapsed_time(e)


def classify(streams, scratch):
    """-> list of classes (lists of indices into streams) that share a hardware queue"""
    def tiny(i):
        def run():
            e = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(streams[i]):
                scratch[i].add_(1)
                e.record()
            return e
        return run
    classes = []
    for i in range(len(streams)):
        for c in classes:
            if not _beside(streams[c[0]], tiny(i)):
                c.append(i)
                break
        else:
            classes.append([i])
    return classes


def pick(device, process_group=None, candidates=8, log=None):
    """Choose the weight-gradient and lane streams of a data-parallel rank (see the module docstring) and register them in
    PREFERRED[device.index].  -> dict(side, lane, classes, rccl_class) for the record."""
    import torch.distributed as dist
    if not hasattr(torch.cuda, '_sleep'):                 # (the spin kernel of the probe)
        return None
    main = torch.cuda.current_stream(device)
    streams = [main] + [torch.cuda.Stream(device=device) for _ in range(candidates)]
    scratch = [torch.zeros(64, device=device) for _ in streams]
    for s, x in zip(streams, scratch):                     # first use, in creation order
        with torch.cuda.stream(s):
            x.add_(1)
    torch.cuda.synchronize()
    classes = classify(streams, scratch)
    main_cls = next(k for k, c in enumerate(classes) if 0 in c)
    rccl_cls = None
    if dist.is_initialized() and dist.get_backend(process_group) == 'nccl':
        tiny = torch.ones(256, device=device)
        dist.all_reduce(tiny, group=process_group)          # communicator and RCCL's stream exist and have run
        torch.cuda.synchronize()
        def collective(issue_stream):
            def run():
                e = torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(issue_stream):
                    dist.all_reduce(tiny, group=process_group, async_op=True).wait()
                    e.record()
                return e
            return run
        # EVERY rank issues the same number of collectives here (a fixed number of rounds, no early exit): ranks whose
        # streams were handed out differently must not fall out of step.  A collective completes only once every rank has
        # issued it, so each round starts from a barrier (the skew between ranks is then far below the spin, which is twice
        # as long as the classification's), every class is probed REPS times and flagged by majority.  With real peers a
        # round is slow on all ranks when RCCL's stream is blocked on any of them -- the class is then avoided everywhere,
        # which is the safe side.
        REPS, ROUNDS = 3, 4
        votes = [0] * len(classes)
        for rep in range(REPS):
            for r in range(ROUNDS):
                k = r % len(classes)
                other = (k + 1) % len(classes)                            # issue from a queue that is not the one spinning
                torch.cuda.synchronize()
                dist.barrier(group=process_group)
                ok = _beside(streams[classes[k][0]], collective(streams[classes[other][0]]), spin=2 * _SPIN)
                if not ok and r < len(classes):
                    votes[k] += 1
        rccl_cls = sorted(k for k, v in enumerate(votes) if 2 * v > REPS)
    rccl_set = set(rccl_cls or ())
    free = [k for k in range(len(classes)) if k != main_cls and k not in rccl_set]
    out = dict(classes=[[('main' if i == 0 else 's%d' % (i - 1)) for i in c] for c in classes],
               main_class=main_cls, rccl_class=rccl_cls, side=None, lane=None)
    if not free:
        import warnings
        warnings.warn('aide_amd.streams.pick: no hardware queue is free of both the main stream and RCCL (classes %s, main %d, '
                      'rccl %s): the engine falls back to streams as the runtime hands them out' % (out['classes'], main_cls, rccl_cls))
    if free:
        side = streams[classes[free[0]][0]]
        shared = [k for k in rccl_set if k != main_cls]
        lane_cls = free[1] if len(free) > 1 else (shared[0] if shared else free[0])
        lane_members = [i for i in classes[lane_cls] if streams[i] is not side]
        lane = streams[lane_members[0]] if lane_members else None
        # a third stream for callers that run a second model beside the first (the co-teaching step's network 2): no queue is
        # left for it, so it shares the lane's -- the least harmful partner (RCCL's queue would put all-reduces in front of
        # it, the weight-gradient queue would serialise its backward pass)
        aux = streams[lane_members[1]] if len(lane_members) > 1 else lane
        PREFERRED[device.index if device.index is not None else torch.cuda.current_device()] = dict(side=side, lane=lane, aux=aux)
        out['side'], out['lane'] = free[0], lane_cls
    if log is not None:
        log(out)
    return out
