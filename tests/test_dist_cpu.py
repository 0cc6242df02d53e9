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


def _seg_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from darknet_amd import dist as dkdist
    from darknet_amd.train_dist import bucket_segments
    ctx = dkdist.DistCtx(backend="gloo")
    convs = [0, 2, 3, 7, 8, 12, 13, 17, 20]
    sizes = [900, 40, 7000, 300, 12000, 64, 30000, 5, 2500]
    segs = bucket_segments(convs, sizes, 22, 4)
    n = sum(sizes)
    rng = np.random.default_rng(100 + rank)
    g = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
    whole = g.clone()
    ctx.dist.all_reduce(whole)
    works = [ctx.dist.all_reduce(g[off:off + cnt], async_op=True) for _, _, off, cnt in segs if cnt]
    for w in works:
        w.wait()
    q.put((rank, segs, bool(torch.equal(g, whole))))
    ctx.close()


def test_segmented_bucket_allreduce_two_ranks():
    """The overlapped trainer reduces the gradient bucket slice by slice (in backward order):
    the slices partition the bucket, follow the layers' backward order, and reducing them one by
    one gives exactly the whole-bucket all-reduce (gloo, world size 2)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_seg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = 900 + 40 + 7000 + 300 + 12000 + 64 + 30000 + 5 + 2500
    for rank, segs, same in res:
        assert same, "slice-wise all-reduce differs from the whole-bucket all-reduce"
        assert 1 <= len(segs) <= 4
        assert segs[0][0] == 22 and segs[-1][1] == 0
        for a, b in zip(segs, segs[1:]):
            assert a[1] == b[0] and a[2] == b[2] + b[3]      # contiguous in layers and in the bucket
        assert segs[-1][2] == 0 and sum(s[3] for s in segs) == total
