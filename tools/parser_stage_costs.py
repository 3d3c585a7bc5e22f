#!/usr/bin/env python3
"""Where the host parser's time goes, by stage ("a profile that shows which parser stage cannot go faster").

No sampling profiler in this image, so the stages are priced by design: 1080p Sorenson Spark pictures whose CONTENT varies --
share of macroblocks not coded (COD = 1), coded inter macroblocks with one or four vectors, intra macroblocks, coded blocks,
events per block -- are parsed by tools/parse_rate.cpp (one thread, best of 60), and the parse times are regressed on the
counts of what each picture holds:

    time = c0 + a * uncoded MBs + b * inter MBs (1 vector) + b4 * inter MBs (4 vectors) + i * intra MBs
              + k * coded blocks + e * events

The coefficients are the stages' prices per unit on this machine (ns, and cycles at the clock given with --ghz); the
table under them says what share of a realistic P picture, of the dense test P picture and of a realistic key frame each
stage is.  usage: python tools/parser_stage_costs.py [--ghz 2.1] [--out file]"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "h263-rs_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import recgen  # noqa: E402
import sorenson_enc as enc  # noqa: E402
from test_bitstream_e2e import make_codable  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ghz", type=float, default=0.0, help="clock of the core the parser runs on (0: read /proc/cpuinfo's current MHz)")
ap.add_argument("--out", default=None)
args = ap.parse_args()
W, H, Q = 1920, 1080, 10


def counts(mbs, co, intra_picture):
    mv_any = mbs["mv"].reshape(len(mbs), -1).any(axis=1)
    intra = (mbs["mb_type"] == 3) | (mbs["mb_type"] == 4)
    four = (mbs["mb_type"] == 2) | (mbs["mb_type"] == 5)
    uncoded = (~intra) & (~four) & (mbs["cbp"] == 0) & (~mv_any) if not intra_picture else np.zeros(len(mbs), bool)
    blocks = int(sum(bin(int(c)).count("1") for c in mbs["cbp"]))
    events = int(np.count_nonzero(co))
    return dict(uncoded=int(uncoded.sum()), inter1=int(((~intra) & (~four) & (~uncoded)).sum()), inter4=int(four.sum()),
                intra=int(intra.sum()), blocks=blocks, events=events)


def pictures():
    out = []
    for k, (ps, pc) in enumerate(((0.9, 0.05), (0.75, 0.15), (0.6, 0.15), (0.6, 0.4), (0.3, 0.15), (0.0, 0.3))):
        out.append(("realistic P, %2.0f %% skipped, %2.0f %% of blocks coded" % (100 * ps, 100 * pc), False,
                    recgen.realistic_inter_picture(W, H, 7000 + k, p_skip=ps, p_coded=pc)))
    for k, (pc, p4, pi) in enumerate(((0.1, 0.0, 0.0), (0.25, 0.1, 0.0), (0.25, 0.5, 0.0), (0.5, 0.1, 0.1), (0.8, 0.0, 0.0))):
        out.append(("random-vector P, %2.0f %% coded, %2.0f %% 4MV, %2.0f %% intra" % (100 * pc, 100 * p4, 100 * pi), False,
                    recgen.inter_picture(W, H, 8000 + k, mv_range=32, p_coded=pc, p_4v=p4, p_intra=pi, quant=Q)))
    for k, pc in enumerate((0.2, 0.7, 1.0)):
        out.append(("realistic key frame, %3.0f %% of blocks coded" % (100 * pc), True,
                    recgen.realistic_intra_picture(W, H, 300 + k, p_coded=pc)))
    import h263mi
    out.append(("bench workload P (every macroblock coded, random vectors)", False, h263mi.synth_picture_host(h263mi.SYNTH_P, W, H, 200, 1)))
    out.append(("bench workload key frame (mixed classes, 20 events per block)", True,
                h263mi.synth_picture_host(h263mi.SYNTH_I_MIXED, W, H, 200, 0)))
    return out


with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "parse_rate")
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-I" + os.path.join(R, "include"), "-o", exe,
                           os.path.join(R, "tools", "parse_rate.cpp"), os.path.join(R, "h263-rs_amd", "host", "bitstream.cpp")])
    rows = []
    for k, (name, is_i, (mbs, co)) in enumerate(pictures()):
        mbs = make_codable(mbs, Q, k, 0 if is_i else 1)
        data = enc.encode_picture(W, H, 0 if is_i else 1, Q, mbs, co, temporal_reference=k)
        f = os.path.join(td, "p%d.bin" % k)
        open(f, "wb").write(data)
        best = 1e9
        for _ in range(3):                                       # the machine is shared: the minimum of three runs
            txt = subprocess.check_output([exe, f], text=True)
            best = min(best, float(re.search(r"([0-9.]+) ms per parse", txt).group(1)))
        c = counts(mbs, co, is_i)
        c.update(name=name, ms=best, bytes=len(data))
        rows.append(c)

ghz = args.ghz
if not ghz:
    mhz = [float(m) for m in re.findall(r"cpu MHz\s*:\s*([0-9.]+)", open("/proc/cpuinfo").read())]
    ghz = max(mhz) / 1e3 if mhz else 1.0
keys = ["uncoded", "inter1", "inter4", "intra", "blocks", "events"]
A = np.array([[1.0] + [r[k] for k in keys] for r in rows])
y = np.array([r["ms"] * 1e6 for r in rows])                       # ns
# non-negative least squares by active-set elimination (scipy is there, but this is six columns)
from scipy.optimize import nnls  # noqa: E402
coef, _ = nnls(A, y)
fit = A @ coef
lines = []
model = "unknown"
for ln in open("/proc/cpuinfo"):
    if ln.startswith("model name"):
        model = ln.split(":", 1)[1].strip()
        break
lines.append("# host parser, price per stage (tools/parser_stage_costs.py): %s, %.2f GHz, one thread, 1080p pictures" % (model, ghz))
lines.append("# %-62s %9s %8s %8s %7s %7s %8s %9s | %9s %9s" % ("picture", "uncoded", "inter 1V", "inter 4V", "intra", "blocks",
                                                                 "events", "bytes", "ms parse", "ms model"))
for r, m in zip(rows, fit):
    lines.append("  %-62s %9d %8d %8d %7d %7d %8d %9d | %9.3f %9.3f" % (r["name"], r["uncoded"], r["inter1"], r["inter4"], r["intra"],
                                                                        r["blocks"], r["events"], r["bytes"], r["ms"], m / 1e6))
lines.append("")
names = dict(uncoded="a macroblock that is not coded (COD = 1; taken in runs out of the header window)",
             inter1="an inter macroblock's header: COD, MCBPC, CBPY, (DQUANT), one vector pair, its prediction",
             inter4="an inter macroblock with four vectors (three more pairs and predictions)",
             intra="an intra macroblock's header (MCBPC, CBPY) and its six INTRADC codes",
             blocks="a coded block: entering and leaving the TCOEF loop, the block's offset word",
             events="a TCOEF event: one table lookup, run/level/last or ESCAPE, de-zigzag, the event word")
lines.append("price per unit (non-negative least squares over the %d pictures; constant %.1f us per picture):" % (len(rows), coef[0] / 1e3))
for k, c in zip(keys, coef[1:]):
    lines.append("  %-8s %6.1f ns = %5.0f cycles   %s" % (k, c, c * ghz, names[k]))
lines.append("")
lines.append("share of the parse time by stage:")
for want in ("realistic P, 60 % skipped, 15 %", "bench workload P", "realistic key frame,  70 %"):
    r = next(r for r in rows if r["name"].startswith(want))
    parts = [coef[0]] + [coef[1 + i] * r[k] for i, k in enumerate(keys)]
    tot = sum(parts)
    lines.append("  %-58s %s" % (r["name"][:58], "  ".join("%s %4.1f %%" % (k, 100 * p / tot) for k, p in zip(["fixed"] + keys, parts))))
txt = "\n".join(lines)
print(txt)
if args.out:
    open(args.out, "w").write(txt + "\n")
