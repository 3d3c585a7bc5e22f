"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/h263mi.h declares, shares its record layout with the oracle, and -- with no GPU in the
container -- fails loudly instead of falling back to a CPU path.  No compute calls here."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

import h263mi
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not os.path.exists("/dev/kfd")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "h263mi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(h263mi_[a-z0-9_]+)\s*\(", src))
    names |= set(re.findall(r"extern const uint8_t (h263mi_[a-z0-9_]+)\[", src))
    return names


def test_library_exports_every_declared_symbol():
    L = h263mi.lib()
    declared = header_symbols()
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(L, name), "include/h263mi.h declares %s but libh263mi.so does not export it" % name
    assert declared == set(h263mi.EXPORTS), declared ^ set(h263mi.EXPORTS)
    assert L.h263mi_abi_version() == 7


def test_record_layout_matches_header_and_oracle():
    assert h263mi.MB_RECORD_DTYPE == orc.MB_RECORD_DTYPE
    assert h263mi.MB_RECORD_DTYPE.itemsize == 32
    assert C.sizeof(h263mi.PictureDesc) == 12
    offs = {n: h263mi.MB_RECORD_DTYPE.fields[n][1] for n in h263mi.MB_RECORD_DTYPE.names}
    assert offs == {"mb_type": 0, "quant": 1, "cbp": 2, "kill": 3, "mv": 4, "intradc": 20, "reserved": 26,
                    "coeff_index": 28}


def test_tables_and_messages():
    assert h263mi.quant_to_strength().tolist() == orc.quant_to_strength().tolist()
    L = h263mi.lib()
    assert b"uncoded iframe blocks" in L.h263mi_strerror(h263mi.ERR_UNCODED_IFRAME_BLOCKS)
    assert b"no CPU fallback" in L.h263mi_strerror(h263mi.ERR_NO_DEVICE)


def test_argument_validation_needs_no_device():
    # preconditions of deblock.rs:30,306 and bt601.rs:100-104 are checked before any device work
    buf = np.zeros(16, np.uint8)
    out = np.zeros(64, np.uint8)
    L = h263mi.lib()
    assert L.h263mi_deblock(buf.ctypes.data, 16, 5, 3, out.ctypes.data) == h263mi.ERR_INVALID_ARGUMENT   # 16 % 5
    assert L.h263mi_deblock(buf.ctypes.data, 16, 4, 0, out.ctypes.data) == h263mi.ERR_INVALID_ARGUMENT   # strength
    assert L.h263mi_deblock(buf.ctypes.data, 16, 4, 13, out.ctypes.data) == h263mi.ERR_INVALID_ARGUMENT
    assert L.h263mi_bt601_yuv420_to_rgba(None, 0, None, None, 0, 0, None) == h263mi.OK                   # empty picture
    assert L.h263mi_bt601_yuv420_to_rgba(buf.ctypes.data, 16, buf.ctypes.data, buf.ctypes.data, 3, 4,
                                         out.ctypes.data) == h263mi.ERR_INVALID_ARGUMENT                 # chroma size


def test_host_synth_matches_sim_generator():
    import simlib
    for kind in (h263mi.SYNTH_I_DENSE, h263mi.SYNTH_I_MIXED, h263mi.SYNTH_P):
        a_m, a_c = h263mi.synth_picture_host(kind, 64, 48, 5, 2)
        b_m, b_c = simlib.synth_picture(kind, 64, 48, 5, 2)
        assert a_m.tobytes() == b_m.tobytes() and a_c.tobytes() == b_c.tobytes()


@pytest.mark.skipif(not NO_GPU, reason="only meaningful where no GPU exists")
def test_no_gpu_fails_loudly_and_never_falls_back():
    assert h263mi.device_count() == 0
    with pytest.raises(h263mi.H263Error) as e:
        h263mi.H263State()
    assert e.value.code == h263mi.ERR_NO_DEVICE
    with pytest.raises(h263mi.H263Error) as e:
        h263mi.deblock(np.zeros(64, np.uint8), 8, 4)
    assert e.value.code == h263mi.ERR_NO_DEVICE
    with pytest.raises(h263mi.H263Error) as e:
        h263mi.yuv420_to_rgba(np.zeros(4, np.uint8), np.zeros(1, np.uint8), np.zeros(1, np.uint8), 2)
    assert e.value.code == h263mi.ERR_NO_DEVICE
    with pytest.raises(h263mi.H263Error) as e:
        h263mi.Batch(2, 64, 48)
    assert e.value.code == h263mi.ERR_NO_DEVICE


def test_kernel_isa_has_no_fused_multiply_add_and_no_compiler_selected_ashr_pk():
    """Bit-exactness guard on the generated gfx950 code: the reference IDCT rounds every product and
    every sum separately (idct.rs:52-65), so no v_fma/v_mac/v_mad_f32 may appear; and hipcc's
    v_ashr_pk_u8_i32 lowering is known to be wrong on gfx950 (tools/probes/probe_ashr_pk.hip)."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "h263-rs_amd", "csrc", "kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "-S", "--cuda-device-only", "-x", "hip", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    body = asm[asm.index("k_recon"):]
    for bad in ("v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32", "v_pk_fma_f32", "v_fma_mix"):
        assert bad not in asm, bad
    # v_ashr_pk_u8_i32 may only come from the hand-written asm of post_kernel.inl (which merges its
    # 16-bit halves with a byte permute), never from the compiler's own lowering
    in_asm = False
    for line in asm.splitlines():
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif "v_ashr_pk_" in line:
            assert in_asm, "compiler-selected v_ashr_pk: " + line
    assert "v_pk_mul_f32" in body or "v_mul_f32" in body


def test_oversized_pictures_are_rejected_before_any_allocation():
    """frame offsets are 32-bit on the device; a Sorenson custom format carries 16-bit width / height from
    an untrusted bitstream.  53600 x 53600 would wrap frame_bytes to 17 MB: every entry point refuses such a size up
    front (the check precedes the device check, so it is testable without a GPU)."""
    import ctypes as C
    L = h263mi.lib()
    out = C.c_void_p()
    cfg = h263mi.BackendCfg(0, 0, None)
    assert L.h263mi_batch_create(1, 53600, 53600, C.byref(cfg), C.byref(out)) == h263mi.ERR_PICTURE_FORMAT_INVALID
    assert L.h263mi_batch_create(1, 65535, 65535, C.byref(cfg), C.byref(out)) == h263mi.ERR_PICTURE_FORMAT_INVALID
    dummy = (C.c_uint8 * 16)()
    assert L.h263mi_deblock(dummy, 53600 * 53600, 53600, 5, dummy) == -103          # H263MI_ERR_OUT_OF_MEMORY
    assert L.h263mi_bt601_yuv420_to_rgba(dummy, 53600 * 53600, dummy, dummy, 26800 * 26800, 53600, dummy) == -103
    # a size that fits is not refused by the size check (it gets as far as the device check)
    assert L.h263mi_batch_create(1, 1920, 1080, C.byref(cfg), C.byref(out)) in (h263mi.OK, h263mi.ERR_NO_DEVICE)
    if out:
        L.h263mi_batch_destroy(out)


def test_committed_traffic_figure_belongs_to_the_kernel_sources_in_the_tree():
    """bench.py reports roofline.traffic from profiles/traffic_latest.json only when that file was measured on the
    kernel sources in the tree (a hash over csrc/*.inl, *.hip, *.h): editing a kernel without re-running
    tools/prof_final.sh would silently turn the figure into null on the driver's bench line."""
    import json
    import bench
    tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert tr["kernel_source_hash"] == bench.kernel_source_hash(), \
        "profiles/traffic_latest.json is from other kernel code: re-run tools/prof_final.sh and copy its traffic.json"
    k = tr["kernels"]["k_frame"]
    assert k["launches_sampled"] >= 60 and 0.9e9 < k["hbm_bytes_per_launch"] < 1.3e9


_PLAN_PROBE = r"""
import sys
sys.path.insert(0, %r)
import h263mi
print(";".join("%%d,%%d" %% h263mi.default_parser_threads(n) for n in (64, 5, 1, 1000)))
"""


def test_default_parser_threads_follow_the_cpu_time_quota(tmp_path):
    """h263mi_default_parser_threads (include/h263mi.h): without a quota, or under one that is no tighter than the CPUs the
    process may run on, the CPUs; under a binding quota of Q CPUs the fewest threads that give the rounds of Q + Q/2 threads;
    LOCAL_WORLD_SIZE divides both limits; H263MI_QUOTA_OVERSUBSCRIBE=0 gives Q."""
    import subprocess
    cpus = len(os.sched_getaffinity(0))
    if cpus < 4:
        pytest.skip("needs 4 CPUs in the affinity mask")

    def plan(quota_text, **env):
        f = tmp_path / "cpu.max"
        f.write_text(quota_text)
        e = dict(os.environ, H263MI_CGROUP_CPU_MAX=str(f))
        e.pop("LOCAL_WORLD_SIZE", None)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", _PLAN_PROBE % os.path.join(ROOT, "h263-rs_amd")], env=e, capture_output=True,
                             text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        return [tuple(int(v) for v in item.split(",")) for item in out.stdout.strip().split(";")]

    # no quota: the CPUs, never more threads than streams
    assert plan("max 100000") == [(min(cpus, 64), 0), (min(cpus, 5), 0), (1, 0), (min(cpus, 1000), 0)]
    # a quota that does not bind (more CPU time than CPUs): the same, the quota reported
    q = 2 * cpus
    assert plan("%d 100000" % (q * 100000)) == [(min(cpus, 64), q), (min(cpus, 5), q), (1, q), (min(cpus, 1000), q)]
    # a binding quota of 2 CPUs: up to 3 threads; 64 streams -> 22 rounds -> 3 threads; 5 streams -> 2 rounds -> 3; 1 stream -> 1
    assert plan("200000 100000") == [(3, 2), (3, 2), (1, 2), (3, 2)]
    assert plan("200000 100000", H263MI_QUOTA_OVERSUBSCRIBE="0") == [(2, 2), (2, 2), (1, 2), (2, 2)]
    # two ranks share the node: half the CPUs and half the quota each
    assert plan("400000 100000", LOCAL_WORLD_SIZE="2") == [(3, 2), (3, 2), (1, 2), (3, 2)]
