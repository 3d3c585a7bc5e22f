"""`python bench.py --gpus N` must itself produce N ranks (SURVEY 8e / BASELINE configs[4]: streams are independent,
state.rs:167,432-438, so the job shards one process per GPU).  Driven here with the CPU stand-in workload
(H263MI_BENCH_STUB: same launcher, rendezvous, barrier and aggregation code over gloo, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "3"], {"H263MI_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints the one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert out["pictures"] == 2 * 3 * 2 * 31 * 4             # both ranks' streams x steps x pictures per step


def test_fewer_devices_than_requested_is_an_error_not_a_smaller_job():
    # no stub; the devices are hidden from the child, so that 2 > visible devices on a GPU box as well
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "",
                                                               "ROCR_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert "device(s) visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "1"], {"H263MI_BENCH_STUB": "1", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE 2 != --gpus 1" in r.stderr


def test_gpus_8_stub_weak_and_strong():
    """BASELINE configs[4]: 512 streams over 8 GPUs.  The launcher, rendezvous, barrier and aggregation at the node's full
    rank count (CPU stand-in, gloo): weak scaling (64 streams per rank) and the strong form (--total-streams 512)."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"H263MI_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["pictures"] == 8 * 64 * 31 * 4
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--total-streams", "512"], {"H263MI_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["pictures"] == 512 * 31 * 4


def test_two_ranks_share_the_host_and_sum_their_end_to_end_legs():
    """VERDICT r4 item 6, without hardware: the paths a first 8-GPU run will take.  Two gloo ranks of the stand-in go
    through the SAME functions as the real main() (h263-rs_amd/shard.py): the strong form deals stream s to rank s mod N
    (SURVEY 8e), a rank's host parser threads are the container's CPU budget divided by the ranks that share it (not one
    per hardware thread per rank), and the end-to-end leg of the line is the pictures of all ranks over the slowest rank's
    time."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--total-streams", "10"], {"H263MI_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["pictures"] == 10 * 31 * 4
    assert out["streams_of_rank"] == [[0, 2, 4, 6, 8], [1, 3, 5, 7, 9]]
    assert out["local_world_size"] == 2
    assert out["parser_threads_per_rank"] == max(1, out["cpu_budget"] // 2)
    # rank 0: 100 pictures in 0.1 s, rank 1: 200 pictures in 0.2 s -> 300 pictures / 0.2 s
    assert out["e2e"]["pictures"] == 300 and abs(out["e2e"]["seconds"] - 0.2) < 1e-9
    assert abs(out["e2e"]["pictures_per_s"] - 1500.0) < 1e-6
    # weak form: contiguous blocks per rank
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--streams", "3"], {"H263MI_BENCH_STUB": "1"})
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["streams_of_rank"] == [[0, 1, 2], [3, 4, 5]]


def test_thread_budget_and_rate_helpers_single_process():
    import shard
    os.environ.pop("LOCAL_WORLD_SIZE", None)
    assert shard.parser_threads_for_rank(16, 1) == 16
    assert shard.parser_threads_for_rank(16, 8) == 2
    assert shard.parser_threads_for_rank(3, 8) == 1                       # never zero
    os.environ["LOCAL_WORLD_SIZE"] = "4"
    try:
        assert shard.parser_threads_for_rank(16, 8) == 4                  # two nodes of four ranks: the node's share
    finally:
        os.environ.pop("LOCAL_WORLD_SIZE")
    rate, units, seconds = shard.aggregate_rate(None, 640, 0.5)
    assert (rate, units, seconds) == (1280.0, 640, 0.5)
