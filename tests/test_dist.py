"""N > 1 path on CPU: world_size-2 gloo processes exercise the same helpers bench.py uses on RCCL
(process-group setup from the launcher environment, max-over-ranks timing, sum-over-ranks work,
per-rank scene seeds, DDP gradient averaging)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(rank))
    from doda_amd import dist as ddist
    import torch.distributed as dist
    w, r, lr = ddist.setup("gloo")
    assert (w, r, lr) == (world, rank, rank) and dist.get_backend() == "gloo"
    # bench bookkeeping: slowest rank defines the time, work adds up
    t, units = ddist.reduce_step_stats(1.0 + rank, [100.0 * (rank + 1), 7.0], torch.device("cpu"))
    assert t == float(world) and units == [100.0 * world * (world + 1) / 2, 7.0 * world]
    assert ddist.seed_for_rank(1000, rank) == 1000 + 100 * rank
    # DDP averages gradients of rank-local batches (scenes are independent units)
    torch.manual_seed(0)
    net = torch.nn.Linear(4, 3)
    ddp = ddist.wrap_ddp(net)
    x = torch.full((5, 4), float(rank + 1))
    ddp(x).sum().backward()
    expect = torch.full((3, 4), 5.0 * (1 + world) / 2)   # mean over ranks of 5*(rank+1)
    assert torch.allclose(net.weight.grad, expect)
    ddist.barrier()
    out.put((rank, t))
    dist.destroy_process_group()


def test_two_rank_gloo_bookkeeping_and_ddp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, 2.0), (1, 2.0)]


def test_single_process_is_a_noop():
    from doda_amd import dist as ddist
    saved = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        assert ddist.setup() == (1, 0, 0)
        t, units = ddist.reduce_step_stats(0.5, [3.0], torch.device("cpu"))
        assert t == 0.5 and units == [3.0]
        m = torch.nn.Linear(2, 2)
        assert ddist.wrap_ddp(m) is m
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def _grad_allreduce_worker(rank, world, port, q):
    import os
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from doda_amd import dist as ddist
    ddist.setup("gloo")
    torch.manual_seed(rank)                       # different initial weights per rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    red = ddist.GradAllReduce(net)                # broadcast from rank 0
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.manual_seed(100 + rank)
    x = torch.randn(4, 6)
    net(x).square().sum().backward()
    local = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    red.reduce()
    avg = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    q.put((rank, w0.tolist(), local.tolist(), avg.tolist()))
    ddist.barrier()


def test_grad_allreduce_world2_gloo():
    """GradAllReduce: same parameters on every rank after construction, gradients averaged."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_grad_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w_a, g_a, avg_a), (_, w_b, g_b, avg_b) = res
    assert w_a == w_b
    want = [(a + b) / 2 for a, b in zip(g_a, g_b)]
    assert max(abs(x - y) for x, y in zip(avg_a, want)) < 1e-6
    assert avg_a == avg_b


def _forced_single_rank_worker(port, q):
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      DODA_DIST_FORCE="1", DODA_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from doda_amd import dist as ddist
    assert ddist.setup() == (1, 0, 0) and dist.is_initialized() and dist.get_world_size() == 1
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    red = ddist.GradAllReduce(net, bucket_mb=1e-4)        # several buckets
    assert red.active and len(red.buckets) > 1
    net(torch.randn(4, 6)).square().sum().backward()
    net[2].bias.grad = None                                # a missing gradient travels as zeros and comes back as zeros
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    red.reduce()
    same = all(torch.equal(p.grad, g) if g is not None else bool((p.grad == 0).all())
               for p, g in zip(net.parameters(), local))
    q.put(same)
    dist.destroy_process_group()


def test_forced_collectives_on_a_group_of_one():
    """DODA_DIST_FORCE=1: a single launcher-started rank joins a process group and GradAllReduce runs its
    bucketed all-reduces anyway (how the RCCL path is exercised on a one-GPU box, tests/test_gpu_round3.py);
    the average over one rank is the gradient itself."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single_rank_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(60)
    assert p.exitcode == 0
