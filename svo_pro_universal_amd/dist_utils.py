"""Multi-GPU plumbing for the bench and any multi-stream driver: the front end
shards by independent units (frame pairs / camera streams / seeds), one process
per GPU, no data-path collective.  torch.distributed is used
only for the barrier around the timed region and to combine per-rank counters.
Works with backend "nccl" (= RCCL over xGMI) on GPUs and "gloo" on CPUs."""
import os


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for this rank (strong-scaling split)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def problem_seed(rank, index):
    """Distinct deterministic scene seed per (rank, local index) for weak scaling."""
    return 1000003 * rank + index


def init(backend, rank, world, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def combine(dist, world, elapsed_s, units, device=None):
    """(max elapsed over ranks, sum of units over ranks): whole-job throughput is
    sum(units) / max(elapsed)."""
    if world == 1:
        return float(elapsed_s), int(units)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([int(units)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def ranks_in_collective(dist, world, device=None):
    """How many ranks the process group's all-reduce really spans: a sum of ones (1 without a group).  Every
    multi-rank benchmark line carries it, so that a scaling run shows that RCCL saw N ranks."""
    if world == 1 or dist is None:
        return 1
    import torch
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
