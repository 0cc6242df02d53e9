"""CPU, world_size 2 over gloo: the N>1 plumbing bench.py uses (shard split,
barrier, max-over-ranks time, sum-over-ranks units)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from darknet_amd import dist as dkdist
    ctx = dkdist.DistCtx(backend="gloo")
    lo, hi = dkdist.shard_range(32, ctx.rank, ctx.world)
    ctx.barrier()
    # rank r "takes" (r+1) seconds for 16*10 images
    rate, tmax = dkdist.aggregate_throughput(ctx, 16 * 10, float(rank + 1))
    q.put((rank, lo, hi, rate, tmax))
    ctx.close()


def test_two_rank_gloo_aggregation():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 16), (16, 32)]
    for r in res:
        assert r[4] == 2.0 and abs(r[3] - 320 / 2.0) < 1e-9  # all images / slowest rank


def test_shard_range_rejects_uneven():
    sys.path.insert(0, ROOT)
    from darknet_amd import dist as dkdist
    assert dkdist.shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(AssertionError):
        dkdist.shard_range(10, 0, 4)
