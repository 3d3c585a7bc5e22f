"""The N>1 path of bench.py on CPU: two gloo ranks exercise the stream sharding and the
barrier / max-over-ranks / sum aggregation of h263-rs_amd/shard.py (the GPU work itself is
covered by the -m gpu tests; no frame data ever crosses ranks)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.streams_of_rank(rank, world, 4)
    fixed = shard.streams_of_rank(rank, world, 0, total_streams=7)
    # rank 1 is deliberately slower: the reported time must be the max over ranks
    elapsed = shard.timed_region(dist, lambda: time.sleep(0.05 + 0.15 * rank))
    pictures = shard.aggregate_pictures(dist, len(mine) * 10)
    q.put((rank, mine, fixed, elapsed, pictures))
    dist.destroy_process_group()


def test_two_gloo_ranks_shard_streams_and_aggregate():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, f0, e0, p0), (r1, m1, f1, e1, p1) = res
    assert m0 == [0, 1, 2, 3] and m1 == [4, 5, 6, 7]               # weak scaling: disjoint, contiguous
    assert sorted(f0 + f1) == list(range(7)) and f0 == [0, 2, 4, 6]  # fixed total: s mod world
    assert e0 == pytest.approx(e1) and e0 >= 0.2                   # both ranks see the slower rank's time
    assert p0 == p1 == 80                                          # whole-job aggregate


def test_single_process_path_needs_no_process_group():
    assert shard.streams_of_rank(0, 1, 64) == list(range(64))
    e = shard.timed_region(None, lambda: time.sleep(0.01))
    assert 0.01 <= e < 1.0
    assert shard.aggregate_pictures(None, 12) == 12
