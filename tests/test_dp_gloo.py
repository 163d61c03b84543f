"""world_size-2 gloo test of the data-parallel runner (CPU, no GPU): sharding + the single all-gather."""
import os
import socket
import sys

import torch
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from trajectorycrafter_amd import dp
    r, w, _ = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    calls = []

    def run_one(i):
        calls.append(i)
        g = torch.Generator().manual_seed(100 + i)            # per-trajectory seed, independent of the rank
        return torch.randn(1, 3, 4, generator=g) + i

    out = dp.run_trajectories(run_one, n_items)
    q.put((rank, calls, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_shard_and_gather():
    world, n_items = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = torch.cat([torch.randn(1, 3, 4, generator=torch.Generator().manual_seed(100 + i)) + i for i in range(n_items)])
    for rank, calls, out in res:
        assert calls == list(range(rank, n_items, world))          # each rank ran only its own trajectories
        assert torch.equal(out, expect)                              # every rank holds all results, in trajectory order


def test_two_ranks_ragged_count():
    """3 trajectories on 2 ranks: rank 1 pads its block for the single all-gather, the padding is dropped, order is by trajectory."""
    world, n_items = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = torch.cat([torch.randn(1, 3, 4, generator=torch.Generator().manual_seed(100 + i)) + i for i in range(n_items)])
    for rank, calls, out in res:
        assert calls == list(range(rank, n_items, world))
        assert out.shape == (n_items, 3, 4) and torch.equal(out, expect)


def test_single_process_is_passthrough():
    sys.path.insert(0, ROOT)
    from trajectorycrafter_amd import dp
    out = dp.run_trajectories(lambda i: torch.full((1, 2), float(i)), 3)
    assert out[:, 0].tolist() == [0.0, 1.0, 2.0]
    assert dp.shard_indices(8, 3, 4) == [3, 7]


def test_guarded_returns_raises_and_times_out():
    import time
    from trajectorycrafter_amd import dp
    assert dp.guarded(lambda: 41 + 1, 5.0, "quick") == 42
    with pytest.raises(ZeroDivisionError):
        dp.guarded(lambda: 1 / 0, 5.0, "raises")
    t0 = time.time()
    with pytest.raises(dp.CollectiveHang, match="stuck call did not return within 1 s"):
        dp.guarded(lambda: time.sleep(30), 1.0, "stuck call")
    assert time.time() - t0 < 5
