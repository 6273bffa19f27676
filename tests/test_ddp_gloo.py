"""CPU, world_size 2, gloo: the bucketed gradient mean all-reduce that bench.py --gpus N installs on
the engine (aide_amd/distributed.py). Replicas shard the data (per-replica BatchNorm statistics and
small-loss selection); the only exchange step is the gradient all-reduce, checked here."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FakeEngine(object):
    def __init__(self, modules):
        self.params = [p for m in modules for p in m.parameters()]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.flat_numel = off
        self.after_backward_op = self.grad_hook = self.before_backward = None


class _FakeModel(object):
    def __init__(self, modules):
        self.engine = _FakeEngine(modules)


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from aide_amd.distributed import GradAllReduce, broadcast_module
    torch.manual_seed(100 + rank)                     # replicas start different ...
    mods = [nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 16, 3), nn.BatchNorm2d(16), nn.Conv2d(16, 2, 1)]
    for m in mods:
        broadcast_module(m)                           # ... and are made identical to rank 0
    model = _FakeModel(mods)
    red = GradAllReduce(model, bucket_mb=0.002)       # tiny buckets -> several of them
    eng = model.engine
    assert eng.after_backward_op is not None and eng.grad_hook is not None
    g = torch.Generator().manual_seed(7 + rank)
    flat = torch.randn(eng.flat_numel, generator=g)
    mine = flat.clone()
    eng.before_backward(flat)
    # backward visits the ops in reverse order (head first)
    eng.after_backward_op(dict(kind='head', conv=mods[4]))
    eng.after_backward_op(dict(kind='conv', conv=mods[2], bn=mods[3]))
    eng.after_backward_op(dict(kind='pool'))
    eng.after_backward_op(dict(kind='conv', conv=mods[0], bn=mods[1]))
    eng.grad_hook(flat)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered) / world
    # only the parameter slots are defined (padding between 16-byte aligned slots is don't-care)
    ok = all(torch.allclose(flat[o:o + p.numel()], expect[o:o + p.numel()], atol=1e-6)
             for o, p in zip(eng.offsets, eng.params))
    psum = torch.stack([p.detach().sum() for p in eng.params]).sum()
    sums = [torch.empty_like(psum) for _ in range(world)]
    dist.all_gather(sums, psum)
    q.put((rank, bool(ok), len(red.sched.buckets), bool(torch.equal(sums[0], sums[1]))))
    dist.destroy_process_group()


def test_bucketed_grad_mean_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, nb, same in res:
        assert ok, 'rank %d: reduced gradients are not the mean over ranks' % rank
        assert nb >= 2 and same


def test_op_index_cache_lives_on_the_step_dict():
    """The per-op parameter indices are cached on the op's own step dict, stamped with the reducer and its bucket-plan
    generation.  (A cache keyed on id(step dict) goes stale when a plan is evicted and rebuilt: CPython hands the id of
    a freed dict to a new one, and a bucket would then be reduced before its gradients are written.)"""
    from aide_amd.distributed import GradAllReduce
    torch.manual_seed(0)
    mods = [nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 16, 3), nn.BatchNorm2d(16), nn.Conv2d(16, 2, 1)]
    model = _FakeModel(mods)
    eng = model.engine
    red = GradAllReduce(model, bucket_mb=0.002, force=True)
    launched = []
    red._launch = launched.append                      # no process group here: record which buckets would go out
    flat = torch.zeros(eng.flat_numel)
    idx = {id(p): i for i, p in enumerate(eng.params)}

    def want(*ms):
        return sorted(idx[id(p)] for m in ms for p in m.parameters())

    # "plan 1": op A = the first conv block
    red._begin(flat)
    st = dict(kind='conv', conv=mods[0], bn=mods[1])
    red._after_op(st)
    assert sorted(red.sched.done) == want(mods[0], mods[1])
    stale_id = id(st)
    del st
    # "plan 2" after an eviction: fresh step dicts, one of which gets the freed dict's id -- but describes another op
    red._begin(flat)
    fresh = None
    keep = []
    for _ in range(64):
        d = dict(kind='conv', conv=mods[2], bn=mods[3])
        if id(d) == stale_id:
            fresh = d
            break
        keep.append(d)
    fresh = fresh if fresh is not None else keep[0]    # (no id reuse on this interpreter: the check below still holds)
    red._after_op(fresh)
    assert sorted(red.sched.done) == want(mods[2], mods[3])
    # the same dict seen again: served from its own tag
    assert fresh['_ddp_pidx'][0] is red and sorted(fresh['_ddp_pidx'][2]) == want(mods[2], mods[3])
    # a rebuilt parameter list (engine._refresh_params) bumps the generation: tags of the old one are ignored
    gen = red._gen
    eng.params = list(eng.params)
    red._begin(flat)
    assert red._gen == gen + 1
    fresh['_ddp_pidx'] = (red, gen, [0])               # a stale tag of the previous generation
    red._after_op(fresh)
    assert sorted(red.sched.done) == want(mods[2], mods[3])
