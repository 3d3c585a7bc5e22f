"""A slice of the parser fuzzer in the CPU suite (tests/parser/fuzz_parser.cpp, built with AddressSanitizer + UBSan).

The host parser (h263-rs_amd/host/bitstream.cpp) stands where the reference has safe Rust (h263/src/parser/reader.rs:37-441)
and writes records in place into pinned staging memory, so its evidence is a sanitizer run over mutated inputs: 10^4 of them
here (bit flips, smashes, truncations, splices, header-field sweeps of Sorenson and ITU-T flavoured seeds; every input through
the windowed and the field-by-field parser, the in-place writer with the smallest legal slot, reused parse buffers; reference
behaviours asserted on every input that still parses -- see the header of fuzz_parser.cpp), 10^7 and more with
`tests/parser/fuzz_parser_asan tests/golden/parser_fuzz_corpus.bin --inputs 10000000 --threads 8` (profiles/r05_*_fuzz_parser*.txt).
The corpus is data made by this repository's encoder: `python tools/gen_fuzz_corpus.py` regenerates it byte for byte."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CORPUS = os.path.join(HERE, "golden", "parser_fuzz_corpus.bin")
FUZZ_DIR = os.path.join(HERE, "parser")


def _fuzzer():
    subprocess.check_call(["make", "-C", FUZZ_DIR, "-s", "fuzz_parser_asan"])
    return os.path.join(FUZZ_DIR, "fuzz_parser_asan")


@pytest.mark.parametrize("seed", [1, 20261004])
def test_ten_thousand_mutated_inputs_under_asan_and_ubsan(seed):
    exe = _fuzzer()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, CORPUS, "--inputs", "10000", "--seed", str(seed), "--threads", "2", "--quiet"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-6000:]
    m = re.search(r"(\d+) inputs from (\d+) seeds .*: (\d+) still parsed, (\d+) layer checks, 0 findings", out.stdout)
    assert m, out.stdout
    inputs, seeds, parsed, layer = map(int, m.groups())
    assert inputs == 10000 and seeds >= 40
    assert parsed >= 800, "too few mutated inputs still parse: the invariants are hardly exercised"
    assert layer >= 40000
    # the mutations reach every kind of ending: end of picture, EOF inside a block, invalid codes of every layer, sizes
    # refused, unimplemented picture kinds
    codes = dict((int(a), int(b)) for a, b in re.findall(r"(-?\d+):(\d+)", out.stdout.split("return codes:")[1]))
    for rc in (0, -2, -3, -4, -5, -6, -7, -8, -12, -14, -16, -17):
        assert codes.get(rc, 0) > 0, (rc, codes)


def test_the_committed_corpus_is_what_the_generator_makes(tmp_path):
    """the corpus is data of this repository's own encoder: regenerating it gives the committed bytes"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_fuzz_corpus", os.path.join(ROOT, "tools", "gen_fuzz_corpus.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    import struct
    blob = struct.pack("<I", 0)
    seeds = gen.seeds()
    blob = struct.pack("<I", len(seeds)) + b"".join(struct.pack("<II", o, len(d)) + d for o, d in seeds)
    assert blob == open(CORPUS, "rb").read()
