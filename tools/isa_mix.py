#!/usr/bin/env python3
"""Static per-phase instruction mix of a gfx950 kernel, from the compiler's assembly.

    python tools/isa_mix.py [k_recon|k_post|k_frame] [-D MACRO ...] > profiles/rNN_isa_mix_<kernel>.txt

Builds h263-rs_amd/csrc/kernels.hip for the device only with -DH263MI_ISA_MARKERS (comments at the phase
boundaries, no instruction emitted) and counts, per phase, the instructions between two markers in program order:

  valu_fast  VALU opcodes that issue in ~2.4 cycles per wave64 on MI355X (profiles/r01_valu_rate.txt):
             v_add/sub/and/or/xor_u32|b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, v_mul/add/sub/fma_f32
  valu_slow  every other VALU opcode (~4.2-4.6 cycles: v_perm, v_med3, v_bfe, v_cndmask, v_cmp, cvt, 24-bit
             multiplies, all v_pk_*, SDWA/DPP forms, v_lshlrev, v_lerp_u8, v_alignbyte ...)
  salu / smem / lds / vmem_rd / vmem_wr / wait (s_waitcnt, s_nop) / branch

Code that exists in two instantiations carries the instantiation in the marker: interior_ / edge_ (tiles of k_post
without / with bounds handling), mc_ / intra_ (reconstruction waves with / without a prediction to fetch); a phase that
the compiler unrolled (the four strips of a post tile) is summed over its copies, `copies` says how many.

Basic blocks that hold the picture-border tap gather of k_recon (recognised by v_lshrrev_b64, which occurs nowhere
else) are reported separately as `border`: one wave in 8-12 executes them on the bench workload.  The table is
static: a loop body counts once and both sides of a scalar branch count -- see the notes printed below the table.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32", "v_not_b32"}
MANGLED = {"k_recon": "_ZN6h263mi7k_reconENS_9ReconArgsE", "k_post": "_ZN6h263mi6k_postENS_8PostArgsE",
           "k_frame": "_ZN6h263mi7k_frameENS_9ReconArgsENS_8PostArgsENS_9FrameGeomE"}
CLASSES = ["valu_fast", "valu_slow", "salu", "smem", "lds", "vmem_rd", "vmem_wr", "wait", "branch"]


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_"):
        if op.endswith("_sdwa") or op.endswith("_dpp"):
            return "valu_slow"
        return "valu_fast" if base in FAST else "valu_slow"
    if op in ("s_waitcnt", "s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vmem_rd"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or "atomic" in op:
        return "vmem_wr"
    return "salu"


def main():
    argv = sys.argv[1:]
    kernel = "k_recon"
    defines = []
    while argv:
        a = argv.pop(0)
        if a == "-D":
            defines.append(argv.pop(0))
        elif a.startswith("-D"):
            defines.append(a[2:])
        else:
            kernel = a
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
               "-fno-strict-aliasing", "-Wno-unused-function", "-DH263MI_ISA_MARKERS"] + ["-D" + d for d in defines] + \
              ["-x", "hip", "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "h263-rs_amd", "csrc", "kernels.hip")]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    name = MANGLED[kernel]
    # (up to the end of the function: a kernel with several exits has several s_endpgm)
    body = re.split(r"\n%s:[^\n]*\n" % re.escape(name), text, maxsplit=1)[1].split(".Lfunc_end", 1)[0].splitlines()
    meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), text, re.S).group(1)

    # split into basic blocks, remember the phase each instruction belongs to
    phase = "prologue"
    blocks = []          # (phase at first instruction, [ops])
    cur = []
    for line in body:
        line = line.strip()
        m = re.match(r"; ISA_MARK (\w+)", line)
        if m:
            if cur:
                blocks.append((phase, cur))
                cur = []
            phase = "after " + m.group(1)
            continue
        if not line or line.startswith(";") or line.startswith("."):
            if re.match(r"\.LBB\d+_\d+:", line) and cur:
                blocks.append((phase, cur))
                cur = []
            continue
        op = line.split()[0]
        cur.append(op)
        if op.startswith("s_cbranch") or op == "s_branch":
            blocks.append((phase, cur))
            cur = []
    if cur:
        blocks.append((phase, cur))

    table = collections.OrderedDict()
    opcount = collections.Counter()
    copies = collections.Counter()
    last = None
    for ph, ops in blocks:
        if ph != last:
            copies[ph] += 1
            last = ph
        border = sum(o.startswith("v_lshrrev_b64") for o in ops) >= 4
        row = table.setdefault(ph, {"main": collections.Counter(), "border": collections.Counter()})
        for o in ops:
            row["border" if border else "main"][classify(o)] += 1
            if not border:
                opcount[o] += 1
    print("# %s: static instruction mix per phase (gfx950, hipcc -O3 -ffp-contract=off%s)" %
          (kernel, "".join(" -D" + d for d in defines)))
    for k in ("sgpr_count", "vgpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
        m = re.search(r"\.%s:\s+(\d+)" % k, meta)
        if m:
            print("# %s = %s" % (k, m.group(1)))
    hdr = "%-30s" % "phase" + "".join("%10s" % c for c in CLASSES) + "%10s" % "border*" + "%8s" % "copies"
    print(hdr)
    tot = collections.Counter()
    tot_border = 0
    for ph, row in table.items():
        nb = sum(row["border"].values())
        print("%-30s" % ph + "".join("%10d" % row["main"][c] for c in CLASSES) + "%10d" % nb + "%8d" % copies[ph])
        tot.update(row["main"])
        tot_border += nb
    print("%-30s" % "total" + "".join("%10d" % tot[c] for c in CLASSES) + "%10d" % tot_border)
    print("# border* = all instructions of the basic blocks that gather picture-edge taps (executed by the waves that\n"
          "#           touch the left / right picture edge only); not included in the other columns")
    print("# most frequent opcodes outside the border blocks:")
    for o, n in opcount.most_common(40):
        print("#   %-28s %4d  %s" % (o, n, classify(o)))


if __name__ == "__main__":
    main()
