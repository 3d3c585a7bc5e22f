"""ThreadSanitizer over the host pipeline: the product's WorkerPool (spin-then-park, generation
word), the stream-affinity / stealing deal and the parse-into-pinned-staging path, built with g++ -fsanitize=thread against a
stub of the HIP runtime (tests/tsan: plain malloc for device and pinned memory, stub kernel launchers) and driven through the
C ABI by 1 .. 40 parser threads x random NULL streams x parked and spinning thread plans x pool teardown mid-spin, with two
batches and a mixed-size set running from three threads at once.  The clean run means something because the same driver with
ONE ordering deliberately broken (the task published with a relaxed store, worker_pool.cpp: H263MI_TSAN_BREAK_GENERATION_ORDER)
is caught.  The reference needs none of this: it is single-threaded safe Rust (`&mut self`, state.rs:138-141)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import recgen
import sorenson_enc as enc
from test_bitstream_e2e import make_codable

HERE = os.path.dirname(os.path.abspath(__file__))
TSAN = os.path.join(HERE, "tsan")
W, H = 176, 144


@pytest.fixture(scope="module")
def drivers():
    subprocess.check_call(["make", "-C", TSAN, "-s", "-j2", "all"])
    return os.path.join(TSAN, "tsan_driver"), os.path.join(TSAN, "tsan_driver_broken"), os.path.join(TSAN, "asan_driver")


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    """12 streams x 5 pictures (1 I + 4 P, shaped like real content: most macroblocks not coded) of QCIF Sorenson Spark, every
    stream with its own quantiser and deblocking flag (the driver renders with H263MI_STRENGTH_FROM_HEADER)"""
    d = tmp_path_factory.mktemp("tsan")
    streams, frames = 12, 5
    out = [struct.pack("<IIII", streams, frames, W, H)]
    for s in range(streams):
        q = 3 + 2 * s
        for f in range(frames):
            if f == 0:
                mbs, co = recgen.intra_picture(W, H, seed=1000 + s, max_level=30)
            else:
                mbs, co = recgen.inter_picture(W, H, seed=2000 + 10 * s + f, mv_range=10, p_4v=0.1, p_intra=0.05, p_coded=0.3,
                                               max_level=20)
            mbs = make_codable(mbs, q, 31 * s + f, 0 if f == 0 else 1)
            pic = enc.encode_picture(W, H, 0 if f == 0 else 1, q, mbs, co, temporal_reference=f, deblock_flag=s & 1)
            out.append(struct.pack("<I", len(pic)) + pic)
    path = d / "corpus.bin"
    path.write_bytes(b"".join(out))
    quota = d / "cpu.max"
    quota.write_text("200000 100000\n")          # a 2-CPU quota: calls with more threads than that take the parking plan
    return str(path), str(quota)


def run(binary, corpus, rounds, max_threads=40, quota=True, **env):
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0", H263MI_NUMA="0", **env)
    e["H263MI_CGROUP_CPU_MAX"] = corpus[1] if quota else os.devnull
    e.pop("LOCAL_WORLD_SIZE", None)
    return subprocess.run([binary, corpus[0], str(rounds), str(max_threads)], env=e, capture_output=True, text=True, timeout=900)


# spinning: the workers never park between calls (a 50 ms spin, no quota, at most 6 threads on this container's 8 CPUs) --
# every hand-over of a task goes through the generation word alone, never through the mutex
SPINNING = dict(max_threads=6, quota=False, H263MI_SPIN_US="50000")


@pytest.mark.parametrize("mode", ["direct", "packed", "spinning"])
def test_host_pipeline_is_clean_under_thread_sanitizer(drivers, corpus, mode):
    kw = {"direct": {}, "packed": {"H263MI_DIRECT_WORDS": "0"}, "spinning": SPINNING}[mode]
    r = run(drivers[0], corpus, 2, **kw)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, r.stderr[-2000:]
    assert "0 check failures" in r.stderr


def test_a_deliberately_broken_ordering_is_caught(drivers, corpus):
    """the generation word published with a relaxed store: a spinning worker reads the task (fn_, pending_) without a
    happens-before edge to the caller's writes -- ThreadSanitizer must say so"""
    r = run(drivers[1], corpus, 1, **SPINNING)
    assert "WARNING: ThreadSanitizer: data race" in r.stderr, r.stderr[-2000:]
    assert r.returncode == 66


@pytest.mark.parametrize("mode", ["direct", "packed"])
def test_host_pipeline_is_clean_under_address_sanitizer(drivers, corpus, mode):
    """the same driver and scenarios under AddressSanitizer + UBSan: parse-into-staging at per-stream pitches, sparse records,
    the 2-D copies out of the slot, mixed-set slot moves -- over malloc'd "device" and "pinned" memory, where one byte beyond an
    allocation is a report (the GPU tests run this code without a sanitizer; the parser alone has its own ASan fuzzer)"""
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", H263MI_NUMA="0", H263MI_CGROUP_CPU_MAX=corpus[1])
    if mode == "packed":
        e["H263MI_DIRECT_WORDS"] = "0"
    e.pop("LOCAL_WORLD_SIZE", None)
    r = subprocess.run([drivers[2], corpus[0], "2", "16"], env=e, capture_output=True, text=True, timeout=900)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "0 check failures" in r.stderr, r.stderr[-2000:]
