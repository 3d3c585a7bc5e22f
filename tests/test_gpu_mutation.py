"""Proof that the GPU parity suite DETECTS an FMA-contracted or re-associated IDCT.

Two mutants of the product library are built from the same sources (h263-rs_amd/Makefile `mutants`):
  libh263mi_fma.so       -ffp-contract=fast instead of off: the compiler fuses the IDCT's multiplies into its adds
  libh263mi_pairwise.so  -DH263MI_MUTATE_PAIRWISE: idct_1d sums its eight products as a balanced tree
On the committed block set (found with the FPU-free soft-float model, tools/find_sensitive_blocks.py) the real
library must reproduce the reference arithmetic (idct.rs:52-65, 171-196) pixel for pixel, and each mutant must NOT."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import h263mi
import mutation_probe

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "h263-rs_amd")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if h263mi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")


def expected_luma(blocks, key=None):
    """clamp(128 + residual) per fixture block; with key = a mutation name, the residuals that mutation produces on
    the pixels the fixture lists (the other pixels as in the reference arithmetic)"""
    w, h = mutation_probe.picture_size(len(blocks))
    y = np.full((h, w), 128, np.int32)
    for k, b in enumerate(blocks):
        res = np.array(b["residual"], np.int32)
        if key and key in b["detects"]:
            for yy, xx, v in b["detects"][key]:
                res[yy, xx] = v
        px, py = (k % mutation_probe.MB_COLS) * 16, (k // mutation_probe.MB_COLS) * 16
        y[py:py + 8, px:px + 8] += res
    return np.clip(y, 0, 255).astype(np.uint8)


def run_with(lib_path, tmp_path, name):
    out = str(tmp_path / (name + ".npy"))
    env = dict(os.environ, H263MI_LIB=lib_path)
    subprocess.run([sys.executable, os.path.join(HERE, "mutation_probe.py"), out], env=env, check=True, timeout=300)
    return np.load(out)


def test_the_product_library_reproduces_the_reference_arithmetic_on_the_sensitive_blocks():
    blocks = mutation_probe.fixture()
    got = mutation_probe.decode_luma()
    assert (got == expected_luma(blocks)).all()


@pytest.mark.parametrize("mutant", ["fma", "pairwise"])
def test_mutant_build_is_caught(mutant, tmp_path):
    lib = os.path.join(PKG, "mutants", "libh263mi_%s.so" % mutant)
    if not os.path.exists(lib):                                  # normally built by __graft_entry__.build()
        subprocess.check_call(["make", "-C", PKG, "-s", "mutants"])
    blocks = mutation_probe.fixture()
    want = expected_luma(blocks)
    got = run_with(lib, tmp_path, mutant)
    wrong_blocks = 0
    for k in range(len(blocks)):
        px, py = (k % mutation_probe.MB_COLS) * 16, (k // mutation_probe.MB_COLS) * 16
        wrong_blocks += bool((got[py:py + 8, px:px + 8] != want[py:py + 8, px:px + 8]).any())
    listed = sum(mutant in b["detects"] for b in blocks)
    # the comparison the parity tests make (output == oracle) fails on the mutant ...
    assert wrong_blocks >= 10, "the %s mutant went unnoticed (%d blocks differ)" % (mutant, wrong_blocks)
    # ... and it fails where the soft-float model of that mutation said it would
    predicted = expected_luma(blocks, mutant)
    agree = sum(bool((got[(k // 10) * 16:(k // 10) * 16 + 8, (k % 10) * 16:(k % 10) * 16 + 8] ==
                      predicted[(k // 10) * 16:(k // 10) * 16 + 8, (k % 10) * 16:(k % 10) * 16 + 8]).all())
                for k, b in enumerate(blocks) if mutant in b["detects"])
    print("%s mutant: %d blocks differ from the oracle; %d of the %d blocks the model lists match its prediction exactly"
          % (mutant, wrong_blocks, agree, listed))
    assert agree >= listed // 2
    # the same run through the product build (same process-isolated path) is clean
    clean = run_with(os.path.join(PKG, "libh263mi.so"), tmp_path, "product")
    assert (clean == want).all()


def test_dequantiser_mutant_is_caught_by_the_every_level_test():
    """libh263mi_dequant.so: the dequantiser computes 16 x the value with a saturating multiply-add and clears the four
    low bits, which is what turns a saturated 32767 into 16 * 2047 (recon_kernel.inl: dequant_pair_i16); the mutant
    leaves them set.  tests/test_gpu_round3.py::test_every_level_at_every_quantiser_dequantises_like_the_oracle must
    fail on it and pass on the product (both in fresh processes: the library is chosen at import)."""
    lib = os.path.join(PKG, "mutants", "libh263mi_dequant.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", PKG, "-s", "mutants"])
    # (the LEVELs within 9 bits: rounds with a wider LEVEL take the wrapping dequantiser, which this mutant does not touch)
    cmd = [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_round3.py"), "-x", "-q", "-k", "every_level and within_9_bits",
           "-p", "no:cacheprovider"]
    bad = subprocess.run(cmd, env=dict(os.environ, H263MI_LIB=lib), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "1 failed" in bad.stdout, bad.stdout[-800:]
    good = subprocess.run(cmd, env=dict(os.environ, H263MI_LIB=os.path.join(PKG, "libh263mi.so")), capture_output=True,
                          text=True, timeout=600)
    assert good.returncode == 0 and "1 passed" in good.stdout, good.stdout[-800:]


def test_wrap_mutant_is_caught_by_the_11_bit_level_test():
    """libh263mi_wrap.so: the dequantiser of rounds with LEVELs outside [-512, 511] uses a SATURATING multiply-add, i.e.
    clamps the mathematical product where a release build of the reference clamps the wrapped i16 one (rle.rs:130-133;
    recon_kernel.inl: dequant_pair_wrap) -- round 1-4's behaviour.  The 11-bit LEVEL sweep and the hand-derived known
    answers of tests/test_gpu_round5.py must fail on it and pass on the product (fresh processes: the library is chosen at
    import)."""
    lib = os.path.join(PKG, "mutants", "libh263mi_wrap.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", PKG, "-s", "mutants"])
    for key, n in (("hand_derived", 1), ("every_11_bit and dense", 1)):
        cmd = [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_round5.py"), "-x", "-q", "-k", key,
               "-p", "no:cacheprovider"]
        bad = subprocess.run(cmd, env=dict(os.environ, H263MI_LIB=lib), capture_output=True, text=True, timeout=600)
        assert bad.returncode != 0 and "1 failed" in bad.stdout, bad.stdout[-800:]
        good = subprocess.run(cmd, env=dict(os.environ, H263MI_LIB=os.path.join(PKG, "libh263mi.so")), capture_output=True,
                              text=True, timeout=600)
        assert good.returncode == 0 and "%d passed" % n in good.stdout, good.stdout[-800:]
