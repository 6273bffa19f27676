"""Two data-parallel ranks (-m gpu): the real engine, streams, events and bucket schedule of aide_amd.distributed with a
real exchange between two processes.

backend = gloo: both ranks on ONE MI355X (RCCL refuses two ranks on one device, so the exchange itself goes over gloo,
AIDE_DIST_BACKEND=gloo, the dry-run backend); everything on the GPU side is the product path.
backend = nccl: one rank per device over real RCCL (ReduceOp.AVG on the communication stream, `device_id=` init, the
tape-replayed bucket hooks); switches itself on wherever >= 2 HIP devices are visible, skipped (with the reason) on a
1-GPU box.

Checked: the reduced gradient arena equals the mean of the two ranks' local gradients (bit-exact against an all-gather of
the local arenas reduced in rank order ... up to the (a+b)/2 vs a/2+b/2 rounding: 1e-6), and both ranks hold identical
parameters after the fused Adam step."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, kind, out, backend='gloo'):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), AIDE_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from aide_amd import utils as U
    from aide_amd.distributed import init_from_env, attach
    from aide_amd.optim import Adam
    from aide_amd.synthetic import chaos_batch
    from aide_amd.train_files.trainchaos_comparison_1case import build_model
    r, w, dev = init_from_env()
    assert (r, w) == (rank, world)
    assert dist.get_backend() == backend
    if backend == 'nccl':
        assert dev.index == rank            # one device per rank
    torch.manual_seed(2 + rank)                    # different initial weights per rank: attach() must broadcast rank 0's
    net = build_model(kind, 2).to(dev)
    net.train()
    single = not kind.startswith('fuseunet')
    xin, xout, t = chaos_batch(2, 64, seed=100 + rank, single_modal=single)
    args = (xin.to(dev),) if single else (xin.to(dev), xout.to(dev))
    one = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(cediceweight=one, ceclassweight=one, diceclassweight=one)
    reducer = attach(net)
    assert reducer is not None
    ref0 = [torch.empty_like(p) for p in net.parameters()]
    for p, q in zip(net.parameters(), ref0):
        q.copy_(p.detach())
        dist.broadcast(q, 0)
        assert torch.equal(q, p.detach()), 'parameters were not broadcast from rank 0'
    # local gradients (reducer hooks off), then the reduced ones
    eng = net.engine
    hooks = (eng.after_backward_op, eng.grad_hook, eng.before_backward)
    eng.after_backward_op = eng.grad_hook = eng.before_backward = None
    crit(net(*args), t.to(dev)).backward()
    local = [p.grad.detach().clone() for p in net.parameters()]
    net.zero_grad()
    eng.after_backward_op, eng.grad_hook, eng.before_backward = hooks
    # three steps through the reducer: the first records the backward launch tape (bucket hooks are tape entries), the
    # later ones replay it -- the replayed exchange must produce the same mean
    arenas = []
    for _ in range(3):
        net.zero_grad()
        crit(net(*args), t.to(dev)).backward()
        arenas.append(eng._arena.data_ptr())
    # the reducer's bucket views / Work objects must not look like a caller's aliases of the gradient arena: the engine would
    # take (and zero-fill) a fresh arena every step and Adam would lose its cached pointer table
    assert len(set(arenas)) == 1, 'the gradient arena moved between data-parallel steps: %s' % (arenas,)
    worst = 0.0
    for p, g in zip(net.parameters(), local):
        parts = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        mean = sum(parts) / world
        scale = mean.abs().max().item() + 1e-20
        worst = max(worst, (p.grad - mean).abs().max().item() / scale)
    assert worst < 1e-5, 'reduced gradient differs from the mean of the local ones: %.3e' % worst
    assert len(reducer.sched.buckets) >= 2 and not any(reducer.sched.pending)
    Adam(net.parameters(), lr=1e-4, amsgrad=True).step()
    for p in net.parameters():
        q = p.detach().clone()
        dist.broadcast(q, 0)
        assert torch.equal(q, p.detach()), 'replicas diverged after the optimizer step'
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        out.put(('ok', worst, len(reducer.sched.buckets)))
    dist.destroy_process_group()


@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
@pytest.mark.parametrize('kind', ['fuseunet', 'UNet'])
def test_two_ranks(dev, kind, backend):
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('RCCL needs one HIP device per rank: %d visible (runs on any box with >= 2)' % torch.cuda.device_count())
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, out, backend)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail('rank process hung')
        assert p.exitcode == 0
    status, worst, nb = out.get(timeout=5)
    assert status == 'ok' and nb >= 2
