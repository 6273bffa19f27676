"""Data-parallel replicas over RCCL/xGMI: one process per GPU, per-replica BatchNorm statistics and
per-replica small-loss selection, gradient MEAN all-reduce of the flat gradient arena.

The whole-network backward runs as one launch sequence, so gradient buckets are carved out of the
flat arena (contiguous parameter ranges, ~25 MB) and each bucket's all-reduce is issued on a side
stream as soon as the last kernel writing into it has been enqueued — it overlaps the remaining
dgrad/wgrad kernels.  With world_size == 1 nothing is installed.

The reference only has single-process nn.DataParallel (trainchaos_comparison_1case.py:131-134); this
is the multi-process replacement BASELINE.json asks for.
"""
import os

import torch
import torch.distributed as dist

# Stream budget.  The HIP runtime maps the streams of a process onto GPU_MAX_HW_QUEUES (4) hardware queues, a fifth stream
# shares the queue of an earlier one -- and a queue runs its packets in order.  Measured on one MI355X with the bucketed
# all-reduce installed on a single-rank RCCL group (tools/bench_comm1.py: the collectives are local, the plumbing is all
# there): with a bucket stream of its own AND RCCL's stream next to main / weight gradients / second forward lane / filter
# packs, RCCL's kernels landed in the MAIN stream's queue (rocprofv3 --kernel-trace: same Queue_Id), and the event wait in
# front of a bucket's all-reduce -- for the weight-gradient stream, which runs ~0.4 ms behind the main stream -- stalled the
# main stream with it: C2 608 -> 567 images/s per rank before a single byte crosses xGMI, and with real peers the whole
# all-reduce would sit in that queue.  More queues are no way out (GPU_MAX_HW_QUEUES=8 with RCCL: 363 images/s), nor are
# high-priority streams for RCCL and the buckets (452).  So the process keeps to FOUR streams: main, second forward lane,
# weight gradients (which also carries the filter re-layout at the start of the forward pass) and RCCL's own; a bucket's
# collective is issued FROM the weight-gradient stream (its tail is a superset of what the bucket waits for; it only has to
# be ordered behind the main stream, which runs ahead of it).
# AIDE_PICK_STREAMS=0: take the streams as the runtime hands them out instead of measuring (aide_amd/streams.py)
PICK_STREAMS = [os.environ.get('AIDE_PICK_STREAMS', '1') != '0']
# RCCL / HIP-runtime tuning is NOT wrapped: RCCL reads its own environment (NCCL_MAX_NCHANNELS, NCCL_MIN_NCHANNELS,
# NCCL_ALGO, NCCL_PROTO, RCCL_*), the runtime GPU_MAX_HW_QUEUES -- set them in the launcher's environment;
# comm_environment() records what was set so that a bench line says under which settings it ran.
_COMM_ENV_PREFIXES = ('NCCL_', 'RCCL_', 'GPU_MAX_HW_QUEUES', 'HSA_ENABLE_IPC_MODE_LEGACY', 'HIP_VISIBLE_DEVICES',
                      'ROCR_VISIBLE_DEVICES')


def comm_environment():
    """-> {name: value} of every RCCL / HIP-runtime variable set in this process that shapes the exchange"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(_COMM_ENV_PREFIXES)}


_RCCL_LOG = [None]          # file RCCL's INFO log of this rank goes to (set by init_from_env when nobody else configured NCCL_DEBUG)


def first_contact(device=None, topo=True):
    """What a first run on real peers should leave on record (bench.py `comm.first_contact`, rank 0, N > 1): the link topology as
    `rocm-smi --showtopo` prints it, the channel count RCCL chose for this communicator (parsed from its own INFO log: the
    workgroups its kernels take from the CUs the 144 KB-LDS convolution kernels want), and whether streams.pick() found a
    hardware queue free of both the main stream and RCCL -- if not, the documented fallback is AIDE_PICK_STREAMS=0 (streams as the
    runtime hands them out) and the record says so.  Everything here is best effort: a missing tool or log yields None."""
    import re
    import subprocess
    out = dict(topology=None, rccl_channels=None, rccl_log=None, streams_fallback=None)
    if topo:
        try:
            r = subprocess.run(['rocm-smi', '--showtopo'], capture_output=True, text=True, timeout=30)
            lines = [ln.rstrip() for ln in r.stdout.splitlines() if ln.strip() and not set(ln.strip()) <= set('=-')]
            out['topology'] = lines[:120]
        except Exception as e:                      # noqa: BLE001  (tool missing / timeout: recorded, not fatal)
            out['topology'] = 'unavailable: %s' % (e,)
    path = _RCCL_LOG[0]
    if path and os.path.exists(path):
        try:
            txt = open(path, errors='replace').read()
            m = re.findall(r'(\d+) coll channels', txt)
            ch = re.findall(r'Channel (\d+)/(\d+)', txt)
            out['rccl_channels'] = int(m[-1]) if m else (int(ch[-1][1]) if ch else None)
            out['rccl_log'] = [ln for ln in txt.splitlines() if 'channels' in ln or 'Init COMPLETE' in ln or 'Using network' in ln][-6:]
        except Exception as e:                      # noqa: BLE001
            out['rccl_log'] = 'unreadable: %s' % (e,)
        try:                                        # parsed: the per-pid temp file is not left behind
            os.remove(path)
            _RCCL_LOG[0] = None
        except OSError:
            pass
    if not PICK_STREAMS[0]:
        out['streams_fallback'] = 'AIDE_PICK_STREAMS=0: streams as the runtime hands them out (no hardware-queue measurement)'
    else:
        from . import streams
        idx = device.index if device is not None and device.index is not None else torch.cuda.current_device()
        if idx not in streams.PREFERRED:
            out['streams_fallback'] = ('streams.pick() found no hardware queue free of the main stream and RCCL (or did not run): the '
                                       'engine uses streams as the runtime hands them out; AIDE_PICK_STREAMS=0 makes that explicit')
    return out


def make_buckets(offsets, numels, bucket_elems):
    """Greedy contiguous buckets over the flat arena, walked in arena (offset) order -- the engine lays the arena out
    in backward-completion order, so bucket k is complete before bucket k+1. -> list of (start, end, [param indices])."""
    buckets, cur, start = [], [], None
    for i in sorted(range(len(offsets)), key=lambda k: offsets[k]):
        o, n = offsets[i], numels[i]
        if start is None:
            start = o
        cur.append(i)
        end = o + n
        if end - start >= bucket_elems:
            buckets.append((start, end, cur))
            cur, start = [], None
    if cur:
        buckets.append((start, offsets[cur[-1]] + numels[cur[-1]], cur))
    return buckets


class BucketScheduler(object):
    """Tracks which parameters' gradients are complete during a backward pass and yields buckets
    that have become fully ready (pure host logic; unit-tested on CPU)."""

    def __init__(self, buckets, nparams):
        self.buckets = buckets
        self.owner = [None] * nparams
        for b, (_, _, idxs) in enumerate(buckets):
            for i in idxs:
                self.owner[i] = b
        self.reset()

    def reset(self):
        self.pending = [len(idxs) for _, _, idxs in self.buckets]
        self.done = set()

    def mark(self, param_indices):
        ready = []
        for i in param_indices:
            if i in self.done:
                continue
            self.done.add(i)
            b = self.owner[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                ready.append(b)
        return ready


def init_from_env(device_ids=None, rccl_log=False):
    """One process per GPU under `python -m torch.distributed.run`: reads RANK / LOCAL_RANK / WORLD_SIZE, selects this
    rank's device (device_ids[local_rank] if given -- the train scripts' --gpu_order -- else local_rank) and, for
    WORLD_SIZE > 1, joins the RCCL process group.  -> (rank, world, device).  AIDE_DIST_BACKEND=gloo is a dry-run backend
    for boxes with fewer GPUs than ranks (ranks wrap around the visible devices; RCCL refuses two ranks per device).
    A --gpu_order index the box does not have (the reference's defaults name GPU 1; there CUDA_VISIBLE_DEVICES would leave no
    device and the script would silently train on the CPU -- a path this package does not have) wraps around the visible
    devices with a warning.  rccl_log=True (bench.py): rank 0 routes RCCL's INIT log to a temp file for first_contact(),
    which removes it."""
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    backend = os.environ.get('AIDE_DIST_BACKEND', 'nccl')
    if not torch.cuda.is_available():
        raise RuntimeError('aide_amd needs a HIP device (the product path has no CPU fallback)')
    if device_ids is not None:
        index = device_ids[local_rank % len(device_ids)]
    else:
        index = local_rank
    if backend != 'nccl':
        index %= torch.cuda.device_count()
    elif index >= torch.cuda.device_count():
        import warnings
        warnings.warn('aide_amd: device %d requested (--gpu_order) but %d visible: using device %d'
                      % (index, torch.cuda.device_count(), index % torch.cuda.device_count()))
        index %= torch.cuda.device_count()
    torch.cuda.set_device(index)
    device = torch.device('cuda', index)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this host driver
        if backend == 'nccl':
            if PICK_STREAMS[0]:
                from . import streams
                streams.reserve_queue(device)            # RCCL's stream then gets a hardware queue of its own (streams.py)
            if rccl_log and rank == 0 and 'NCCL_DEBUG' not in os.environ and 'NCCL_DEBUG_FILE' not in os.environ:
                # rank 0 keeps RCCL's own account of the communicator (channel count, transport) for first_contact()
                import tempfile
                _RCCL_LOG[0] = os.path.join(tempfile.gettempdir(), 'aide_rccl_rank0_%d.log' % os.getpid())
                os.environ['NCCL_DEBUG'] = 'INFO'
                os.environ['NCCL_DEBUG_SUBSYS'] = 'INIT'
                os.environ['NCCL_DEBUG_FILE'] = _RCCL_LOG[0]
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device


def attach(module):
    """Data-parallel replica set-up of one aide_amd model: rank 0's parameters / buffers everywhere, then the bucketed
    gradient mean all-reduce on its engine.  No-op (returns None) for a single process."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    broadcast_module(module)
    return GradAllReduce(module)


def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
    # writes through .data bump neither tensor._version nor the optimizer's epoch: the engine's packed / Winograd-
    # transformed filter copies are keyed on those, so tell it explicitly (any other `.data` edit of weights needs the same)
    from . import engine
    engine.PARAM_EPOCH[0] += 1
    engine.STATS_EPOCH[0] += 1           # ... and the folded eval-mode BatchNorm coefficients on the running statistics


class GradAllReduce(object):
    """Installs the bucketed, overlapped gradient mean all-reduce on an aide_amd model's engine."""

    def __init__(self, model, bucket_mb=25.0, process_group=None, force=False):
        self.engine = model.engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * 1e6 / 4)
        self.sched = None
        self.flat = None
        self.works = []
        self.time_exposed = False        # bench.py: event-time the main stream's wait for the communication stream
        self._exposed = []
        self.stream_plan = None
        self._gen = 0                    # bumped whenever the bucket plan is rebuilt: stamps the per-op index caches
        if self.world > 1 or force:          # force: exercise the RCCL path on a single-rank group (tests)
            p0 = next(iter(model.parameters()), None) if hasattr(model, 'parameters') else None
            dev = p0.device if p0 is not None else None
            if PICK_STREAMS[0] and dev is not None and dev.type == 'cuda' and not getattr(self.engine, 'plans', None):
                # before the engine builds its first plan: streams that really run beside each other and beside RCCL's
                from . import streams
                if dev.index not in streams.PREFERRED:
                    with torch.cuda.device(dev):
                        self.stream_plan = streams.pick(dev, process_group)
            self.engine.after_backward_op = self._after_op
            self.engine.grad_hook = self._finish
            self.engine.before_backward = self._begin

    def _ensure(self):
        eng = self.engine
        if self.sched is None or self._params is not eng.params:
            numels = [p.numel() for p in eng.params]
            self.sched = BucketScheduler(make_buckets(eng.offsets, numels, self.bucket_elems), len(numels))
            self._params = eng.params
            self._pidx = {id(p): i for i, p in enumerate(eng.params)}
            self._gen += 1

    def _begin(self, flat):
        self._ensure()
        self.sched.reset()
        self.flat = flat
        self.works = []

    def _launch(self, b):
        start, end, _ = self.sched.buckets[b]
        view = self.flat[start:end]
        avg = dist.get_backend(self.pg) == 'nccl'          # RCCL averages in the collective itself: no extra pass
        if view.is_cuda:
            side = getattr(self.engine, 'side_stream', None)
            if side is None:                               # no weight-gradient stream (single-stream schedule): in line
                self._reduce(view, avg)
                return
            # everything that writes this bucket has been enqueued, on the main stream and on the weight-gradient stream.
            # The collective is issued from the weight-gradient stream itself: RCCL's stream then waits for that stream's
            # tail -- a superset of the bucket's weight gradients -- and the stream only has to be ordered behind the main
            # stream, which runs ahead of it anyway.
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                self._reduce(view, avg)
        else:
            self._reduce(view, avg)

    def _reduce(self, view, avg):
        if avg:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
        else:
            # gloo (CPU unit tests; AIDE_DIST_BACKEND=gloo dry runs of the N>1 bench on one GPU) has no AVG
            view.div_(self.world)
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _after_op(self, st):
        # The parameter indices of an op are cached ON its step dict (this runs as a tape callback in every backward pass),
        # stamped with this reducer and its bucket-plan generation: a step dict dies with its plan (MAX_PLANS eviction, a
        # precision switch), so a new plan's step can never inherit another op's entry -- a cache keyed on id(step) could,
        # CPython reuses the ids of freed dicts.
        tag = st.get('_ddp_pidx')
        if tag is not None and tag[0] is self and tag[1] == self._gen:
            idxs = tag[2]
        else:
            idxs = []
            for key in ('conv', 'bn', 'mod'):
                m = st.get(key)
                if m is not None:
                    idxs += [self._pidx[id(p)] for p in m.parameters()]
            st['_ddp_pidx'] = (self, self._gen, idxs)
        for b in self.sched.mark(idxs):
            self._launch(b)

    def _finish(self, flat):
        # anything not yet launched (should be nothing), then make the compute stream wait for comm
        for b, pend in enumerate(self.sched.pending):
            if pend > 0:
                self.sched.pending[b] = 0
                self._launch(b)
        timed = self.time_exposed and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self.works:
            w.wait()
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
        # the Work objects hold their bucket views (and `flat` the arena): dropped here, not at the next _begin -- the engine's
        # "is a gradient of the last pass still held by somebody?" check (engine._views_held_elsewhere: storage use count of
        # the arena) runs BEFORE the next pass' before_backward and would see them as a caller's aliases, i.e. take a fresh
        # 107 MB arena every step
        self.works = []
        self.flat = None

    # ---- what bench.py reports about the exchange
    def describe(self):
        """-> dict(buckets, bytes_per_step): the bucket plan of this module's gradient arena"""
        self._ensure()
        sp = self.stream_plan
        return dict(buckets=len(self.sched.buckets),
                    bytes_per_step=int(sum(4 * (end - start) for start, end, _ in self.sched.buckets)),
                    hw_queues=None if sp is None else dict(classes=sp['classes'], main=sp['main_class'], rccl=sp['rccl_class'],
                                                           side=sp['side'], lane=sp['lane']))

    def exposed_ms(self, last=8):
        """mean time (ms) the compute stream stalled at the end of a backward pass waiting for the all-reduces still in
        flight, over the last `last` timed steps (event pair around the wait; needs time_exposed = True and an idle device)"""
        pairs = self._exposed[-last:]
        if not pairs:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
