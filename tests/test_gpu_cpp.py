"""The reference's own yuv / deblock tests re-run through the C++ mirror of its API
(tests/cpp/reference_style_tests.cpp, h263-rs_amd/host/h263mi.hpp)."""
import os
import subprocess

import pytest

CPP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")


def _build():
    subprocess.check_call(["make", "-C", CPP, "-s"])
    return os.path.join(CPP, "reference_style_tests")


def test_cpp_mirror_builds_and_refuses_to_run_without_a_gpu():
    exe = _build()
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: covered by the gpu-marked test")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "no CPU fallback" in out.stdout


@pytest.mark.gpu
def test_reference_style_cpp_tests_pass_on_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "all reference-style tests passed" in out.stdout
