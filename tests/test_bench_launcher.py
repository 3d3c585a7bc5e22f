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
