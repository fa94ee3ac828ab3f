"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (GradSync) and the uneven-input guard."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from slam_llm_amd.train import GradSync, all_ranks_have_data, setup_distributed
    r, lr, w = setup_distributed("cpu")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    n = 100_000
    flat = torch.full((n,), float(rank + 1))
    flat[: 10] = torch.arange(10, dtype=torch.float32) * (rank + 1)

    class FakeModel:
        grad_hooks = []

    gs = GradSync(flat, bucket_bytes=64 * 1024).attach(FakeModel)
    # backward produces the flat buffer prefix by prefix (last LLM layer first, projector tail last)
    for end in (10_000, 30_000, 30_001, 70_000):
        for hk in FakeModel.grad_hooks:
            hk(end)
    gs.finish()  # flushes the tail and averages
    expect = torch.full((n,), 1.5)
    expect[: 10] = torch.arange(10, dtype=torch.float32) * 1.5
    ok = torch.allclose(flat, expect)
    # second step re-uses the object
    flat.fill_(float(rank))
    for hk in FakeModel.grad_hooks:
        hk(n)
    gs.finish()
    ok = ok and torch.allclose(flat, torch.full((n,), 0.5))
    # gradient accumulation, k = 2 (reference: DDP all-reduces every backward, utils/train_utils.py:128-152; the result must be
    # avg(g1) + avg(g2) on every rank).  (a) train_step's protocol: disarmed on the non-stepping micro-step, one reduction of
    # the accumulated buffer on the last one.  (b) always armed (a caller that never disarms): every backward reduces, and
    # on_backward_begin retires the previous backward's collectives before the kernels accumulate into the buffer.
    g1 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    g2 = torch.full((n,), 10.0 * (rank + 1))
    want = torch.arange(n, dtype=torch.float32) * 1.5 + 15.0
    for always_armed in (False, True):
        flat.zero_()
        for micro, g in enumerate((g1, g2)):
            gs.arm(always_armed or micro == 1)
            gs.on_backward_begin()
            flat.add_(g)                       # the backward kernels accumulate in place
            for end in (40_000, n):
                gs.on_prefix(end)
        gs.finish()
        ok = ok and torch.allclose(flat, want)
    gs.arm(True)
    # uneven shards: rank 1 runs dry first -> everybody stops (reference: Join / monitored_barrier)
    flags = [all_ranks_have_data(step < (3 if rank == 0 else 2), torch.device("cpu")) for step in range(3)]
    q.put((rank, bool(ok), flags))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok, flags in res:
        assert ok, f"rank {rank}: averaged gradients wrong"
        assert flags == [True, True, False]


# ------------------------------------------------------------------------------------------------ Join policy at world 3
def _join_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from slam_llm_amd.train import GradSync, ranks_with_data, setup_distributed, train_step
    setup_distributed("cpu")
    n = 40_000
    plan = [9_000, 9_001, 25_000, n]            # a backward's prefix announcements (SlamHipModel.prefix_plan())

    class FakeStore:
        size = n
        pure_bf16 = False
        grad = torch.zeros(n)
        flat = torch.ones(n)

    class FakeModel:
        grad_hooks = []
        store = FakeStore

        @staticmethod
        def prefix_plan():
            return plan

        @staticmethod
        def attach_grad_views():
            pass

    class SGD:          # p -= lr * g on the flat buffers: enough to see whether the replicas stay identical
        steps = 0

        def step(self):
            FakeStore.flat.sub_(0.1 * FakeStore.grad)
            SGD.steps += 1

        def zero_grad(self):
            pass

    gs = GradSync(FakeModel, bucket_bytes=32 * 1024).attach(FakeModel)
    opt = SGD()
    n_batches = [3, 1, 2][rank]               # uneven shards: rank 1 runs dry after one batch, rank 2 after two
    it, history = 0, []
    while True:
        has = it < n_batches
        active = ranks_with_data(has, torch.device("cpu"))
        if active == 0:
            break
        if has:       # a "backward": this rank's gradient of iteration `it` is (rank + 1) * (it + 1) everywhere
            gs.arm(True)
            gs.on_backward_begin()
            FakeStore.grad.fill_(float((rank + 1) * (it + 1)))
            for end in plan:
                gs.on_prefix(end)
            gs.finish()
            opt.step()
        else:
            assert train_step(FakeModel, None, opt, None, gs) == (None, None)
        history.append((active, float(FakeStore.grad[0]), float(FakeStore.grad[-1])))
        it += 1
    q.put((rank, history, opt.steps, float(FakeStore.flat[0]), float(FakeStore.flat[-1])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradsync_join_policy_world3_gloo():
    """uneven shard counts (3 / 1 / 2 batches on ranks 0 / 1 / 2) under the Join policy: exhausted ranks shadow the collectives with zero
    gradients (GradSync.shadow_backward through train_step(batch=None)), so every rank runs three iterations, every iteration's buffer is
    the sum of the ACTIVE ranks' gradients over the WORLD size (DDP Join's divide_by_initial_world_size), and the replicas' parameters
    stay identical because every rank applies every averaged step."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_join_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # iteration 0: ranks 0, 1, 2 active with gradients 1, 2, 3 -> mean 2; iteration 1: ranks 0, 2 with 2, 6 -> 8 / 3; iteration 2: rank 0 with 3 -> 1
    want = [(3, 2.0), (2, 8.0 / 3.0), (1, 1.0)]
    for rank, history, steps, p0, p1 in res:
        assert steps == 3 and len(history) == 3, (rank, history)
        for (active, g0, g1), (wa, wg) in zip(history, want):
            assert active == wa and abs(g0 - wg) < 1e-6 and abs(g1 - wg) < 1e-6, (rank, history)
        assert abs(p0 - (1.0 - 0.1 * (2.0 + 8.0 / 3.0 + 1.0))) < 1e-6 and p0 == p1
    assert len({(p0, p1) for _, _, _, p0, p1 in res}) == 1      # identical replicas


# ------------------------------------------------------------------------------------------------ bench.py's rank logic at world 8
def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import time
    import bench
    from slam_llm_amd.train import GradSync, setup_distributed
    r, lr, w = setup_distributed("cpu")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    # a stub step whose duration depends on the rank: rank r sleeps (r + 1) x 4 ms, and "exposes" (r + 1) ms of communication
    calls = []

    def step():
        calls.append(time.perf_counter())
        time.sleep(0.004 * (rank + 1))
        return float(rank), 0.0
    steps, warmup = 5, 2
    res, elapsed, comm = bench.measure(step, steps, warmup, world, dist, lambda: None, dev, comm_ms_fn=lambda: float(rank + 1))
    # the flat-buffer exchange itself at world 8: mean over ranks, prefix buckets, projector tail
    n = 50_000
    flat = torch.full((n,), float(rank))

    class FakeModel:
        grad_hooks = []
    gs = GradSync(flat, bucket_bytes=32 * 1024).attach(FakeModel)
    for end in (8_000, 8_001, 33_000, n):
        for hk in FakeModel.grad_hooks:
            hk(end)
    gs.finish()
    mean_ok = bool(torch.allclose(flat, torch.full((n,), (world - 1) / 2.0)))
    q.put((rank, len(calls), res, elapsed, comm, bench.rank_seed(rank), mean_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bench_rank_logic_world8_gloo():
    """bench.py --gpus 8 without GPUs: eight gloo ranks drive bench.measure() with a stub step (rank r takes (r + 1) x 4 ms).  Checked:
    every rank runs exactly W + K steps; the elapsed time every rank ends up with is the SAME number = the slowest rank's (MAX over
    ranks, >= K x 32 ms); comm_exposed_ms is the MAX over ranks (8.0); the per-rank data seeds differ; the throughput formula is the
    whole-job aggregate (world x clips); and the flat-buffer mean all-reduce of GradSync is right at world 8."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    elapsed = {round(e, 9) for _, _, _, e, _, _, _ in res}
    assert len(elapsed) == 1, f"ranks disagree on the elapsed time: {elapsed}"
    e = res[0][3]
    assert e >= 5 * 0.004 * world * 0.98, e                       # the slowest rank's five steps
    assert e < 5 * 0.004 * world * 3 + 1.0, e                     # ... and not the SUM over ranks
    for rank, n_calls, last, _, comm, seed, mean_ok in res:
        assert n_calls == 7 and last == (float(rank), 0.0)
        assert comm == float(world)                               # MAX over ranks of the per-rank exposed time
        assert mean_ok
    assert len({seed for *_, seed, _ in res}) == world and res[3][5] == bench.rank_seed(3)
    value, ms = bench.throughput(world, 31, 30.0, 5, e)
    assert abs(value - world * 31 * 30.0 * 5 / e) < 1e-9 and abs(ms - e / 5 * 1e3) < 1e-9
