"""Multi-GPU plumbing for the batch-sharded forward path.

Inference shards naturally: images are independent, so the
global batch is split evenly over the ranks (one process per GPU, one replica of
the folded weights each) and there is NO data-path collective.  torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used only for the barrier that
brackets the timed region and for the max-over-ranks reduction of the elapsed
time.  The training path's gradient all-reduce lives in train_dist (later round).
"""
import os


def shard_range(global_batch, rank, world):
    """Contiguous, even split of image indices [0, global_batch) (the reference's
    GetPartialData slicing, src/data.cpp:890-901)."""
    assert global_batch % world == 0, "global batch must divide evenly over the ranks"
    per = global_batch // world
    return rank * per, (rank + 1) * per


class DistCtx:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if not dist.is_initialized():
                dist.init_process_group(backend=backend or "nccl", rank=self.rank,
                                        world_size=self.world)
            self.dist = dist

    def _tensor(self, v):
        import torch
        dev = "cuda" if (self.backend or "nccl") == "nccl" else "cpu"
        return torch.tensor([float(v)], dtype=torch.float64, device=dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, v):
        if self.dist is None:
            return float(v)
        t = self._tensor(v)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v):
        if self.dist is None:
            return float(v)
        t = self._tensor(v)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def aggregate_throughput(ctx, local_units, local_seconds):
    """Whole-job rate: units processed by ALL ranks / the slowest rank's time."""
    t = ctx.max(local_seconds)
    u = ctx.sum(local_units)
    return u / t, t
