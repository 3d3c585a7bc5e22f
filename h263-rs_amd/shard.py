"""Stream sharding and throughput aggregation for multi-GPU runs (one process per GPU).

The path shards across independent streams only: stream `s` is pinned to one rank for
its lifetime, its frame store never leaves that GPU's HBM, and the only collectives are the barrier
around the timed region and the max-over-ranks reduction of the elapsed time.  Used by bench.py
(backend "nccl" = RCCL) and covered by tests/test_multirank.py with two gloo ranks on CPU.
"""
import time


def streams_of_rank(rank, world, streams_per_gpu, total_streams=None):
    """Global stream ids decoded by `rank`.

    Weak scaling (total_streams None): every rank owns `streams_per_gpu` streams, ids
    rank*streams_per_gpu ... ; with a fixed total, stream s goes to rank s mod world."""
    if total_streams is None:
        return list(range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu))
    return [s for s in range(total_streams) if s % world == rank]


def barrier(dist, sync_device=None):
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if sync_device is not None:
        sync_device()


def timed_region(dist, run, sync_device=None):
    """barrier + device sync | run() | device sync + barrier; returns the MAX elapsed seconds over ranks."""
    import torch
    if dist is not None and dist.is_initialized():
        # the closing barrier is inside the timed region: make sure it is a warm one (the first RCCL barriers of a
        # process set up communicators and cost about a millisecond)
        for _ in range(3):
            dist.barrier()
    barrier(dist, sync_device)
    t0 = time.perf_counter()
    run()
    if sync_device is not None:
        sync_device()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized():
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_pictures(dist, pictures_this_rank):
    """Whole-job picture count (sum over ranks)."""
    import torch
    if dist is None or not dist.is_initialized():
        return int(pictures_this_rank)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([pictures_this_rank], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def aggregate_rate(dist, units_this_rank, seconds_this_rank):
    """Whole-job rate of a leg every rank ran on its own streams at the same time: units summed over the ranks divided
    by the slowest rank's time (the end-to-end leg of bench.py: pictures per second of N host parsers + N GPUs)."""
    import torch
    if dist is None or not dist.is_initialized():
        return units_this_rank / seconds_this_rank if seconds_this_rank > 0 else 0.0, int(units_this_rank), seconds_this_rank
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    u = torch.tensor([units_this_rank], dtype=torch.int64, device=dev)
    t = torch.tensor([seconds_this_rank], dtype=torch.float64, device=dev)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    units, seconds = int(u.item()), float(t.item())
    return (units / seconds if seconds > 0 else 0.0), units, seconds


def local_world_size(world):
    """ranks that share this node (and its CPU quota): LOCAL_WORLD_SIZE as torch.distributed.run exports it, else the
    world size (bench.py runs on ONE node)"""
    import os
    try:
        n = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
    except ValueError:
        n = 0
    return max(1, n if n > 0 else world)


def parser_threads_for_rank(cpu_budget, world):
    """Host parser threads ONE rank may use: the CPUs this container may use (cgroup quota / affinity / physical cores:
    bench.physical_cores) divided by the ranks that share them -- never "one per hardware thread per rank" (8 ranks x 256
    threads on a 16-CPU quota only get throttled)."""
    return max(1, int(cpu_budget) // local_world_size(world))
