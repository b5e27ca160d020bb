"""Code-placement tuning of the rollout kernels' substep loops ("straddles").

A lone wavefront per SIMD fetches its code in 32-byte pieces; an 8-byte instruction that straddles such a boundary costs
~0.8 issue slots more than one that does not (measured: builds whose loops differ ONLY in their start address differ by
up to 1.3 % in kernel time, in the order of their straddle counts).  Where the unrolled substep loop starts is an
accident of the code in front of it — so for the instantiations the built-in models run, this script compiles the
kernel once (pads off), counts the straddling 8-byte instructions of the loop for each of the eight possible 4-byte
shifts, and writes the shift that minimises them to csrc/mbd_phase_gen.inc as a number of `s_nop 0` (4 bytes each, run
once per control step) the kernel puts in front of the loop.  Deterministic for a given compiler; build() runs it when
the kernel sources change."""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "model-based-diffusion_amd", "csrc")
OUT = os.path.join(CSRC, "mbd_phase_gen.inc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-DMBD_PHASE_TUNING", "-S", "--cuda-device-only"]
# (kind, template arguments, key of the generated table)
TARGETS = [
    ("3d", "16,true,false,3,1,1,-4,-6,0,false,true,3,false,false,0,7", (1, 0)),  # humanoidrun
    ("3d", "16,true,false,3,1,1,-4,-6,0,false,true,3,false,false,3,5", (1, 3)),  # humanoidtrack
    ("3d", "16,true,false,3,5,1,-4,-6,0,false,true,3,false,false,4,7", (5, 4)),  # humanoidstandup
    ("3d", "16,true,false,3,1,1,-4,-6,0,false,true", (1, -1)),                    # humanoid-shaped, other rewards
    ("planar", "4,2,1,0,0,1,20", (4, 2, 1, 0, 0, 1, 20)),      # hopper
    ("planar", "8,2,1,-3,0,1,20", (8, 2, 1, -3, 0, 1, 20)),    # walker2d
    ("planar", "8,2,1,-3,1,2,16", (8, 2, 1, -3, 1, 2, 16)),    # halfcheetah
    ("planar", "4,0,1,0,2,5,4", (4, 0, 1, 0, 2, 5, 4)),        # cartpole
]


def inputs():
    return [os.path.join(CSRC, f) for f in ("mbd_kernels.h", "mbd_planar.h", "mbd_math.h")] + [os.path.abspath(__file__)]


def substep_loop(asm_lines):
    """(label, instruction count) of the unrolled substep loop: the longest backward-branch region without a memory
    access that does not contain another such region of comparable size (tools/count_flops.py)."""
    lab = {}
    for k, l in enumerate(asm_lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = k
    loops = []
    for k, l in enumerate(asm_lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            seg = asm_lines[lab[m.group(1)]:k + 1]
            n = sum(1 for x in seg if x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";")))
            mem = any(("global_" in x or "scratch_" in x or "flat_" in x) for x in seg)
            loops.append((m.group(1), lab[m.group(1)], k, n, mem))
    cands = sorted([t for t in loops if not t[4]], key=lambda t: -t[3])
    bt = next(t for t in cands
              if not any(u is not t and u[1] >= t[1] and u[2] <= t[2] and u[3] >= 0.45 * t[3] for u in cands))
    return bt[0], bt[3]


def analyse(kind, targs, tuned=False):
    """tuned=True: compile WITH the generated pads (what the library gets): `now` is then the count to expect."""
    hdr, kern = ("mbd_planar.h", "rollout_planar_kernel") if kind == "planar" else ("mbd_kernels.h", "rollout_kernel")
    with tempfile.TemporaryDirectory() as td:
        src, asm, obj = (os.path.join(td, n) for n in ("k.hip", "k.s", "k.o"))
        with open(src, "w") as f:
            f.write(f'#include "{CSRC}/{hdr}"\ntemplate __global__ void mbd::{kern}<{targs}>(mbd::RolloutParams);\n')
        flags = [f for f in FLAGS if not (tuned and f == "-DMBD_PHASE_TUNING")]
        if kind == "planar":  # (the flags of its translation unit, mbd_planar.hip: __graft_entry__.TUS)
            flags = flags + ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
        else:  # (the built-in humanoids' instantiations: mbd_hot3d.hip)
            flags = flags + ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, src, "-o", asm], check=True, capture_output=True)
        lines = open(asm).read().split("\n")
        start = [i for i, l in enumerate(lines) if re.match(rf"^_ZN3mbd\d+{kern}.*:", l)][0]
        end = [i for i, l in enumerate(lines) if i > start and ".Lfunc_end" in l][0]
        label, n = substep_loop(lines[start:end])
        subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-Wa,-L",
                        "-c", asm, "-o", obj], check=True, capture_output=True)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", obj], capture_output=True, text=True).stdout.split("\n")
    begin, ins, in_kernel = None, [], False
    for x in dis:
        m = re.match(r"^([0-9a-f]+) <(.+)>:", x)
        if m:
            if kern in m.group(2):
                in_kernel = True
            if in_kernel and m.group(2) == label and begin is None:
                begin = int(m.group(1), 16)
        m = re.search(r"//\s+([0-9A-F]+):\s+((?:[0-9A-F]{8}\s*)+)", x)
        if m and begin is not None and len(ins) < n:
            ins.append((int(m.group(1), 16), 4 * len(m.group(2).split())))
    counts = {s: sum(1 for a, nb in ins if nb == 8 and (a + s) % 32 == 28) for s in range(0, 32, 4)}
    best = min(counts, key=lambda s: (counts[s], s))
    return dict(label=label, instructions=n, straddles=counts, shift=best, pad=best // 4, now=counts[0])


def generate(path=OUT, verbose=True, check=True):
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(TARGETS)) as ex:
        res = list(ex.map(lambda t: analyse(t[0], t[1]), TARGETS))
    _write(path, res, verbose)
    if check:  # a pad that does not move the loop as predicted (the compiler placed it elsewhere) is dropped
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(TARGETS)) as ex:
            got = list(ex.map(lambda t: analyse(t[0], t[1], tuned=True), TARGETS))
        bad = [i for i, (r, g) in enumerate(zip(res, got))
               if g["now"] != r["straddles"][r["shift"]] or g["instructions"] != r["instructions"]]
        for i in bad:
            if verbose:
                print(f"{TARGETS[i][1]}: predicted {res[i]['straddles'][res[i]['shift']]}, built {got[i]['now']}: pad dropped",
                      file=sys.stderr)
            res[i].update(shift=0, pad=0)
        if bad:
            _write(path, res, False)
    return res


def _write(path, res, verbose):
    rows3, rowsp = [], []
    for (kind, targs, key), r in zip(TARGETS, res):
        if verbose:
            print(f"{kind:6s} <{targs}>: loop of {r['instructions']} instructions, straddles by shift {r['straddles']} "
                  f"-> {r['pad']} s_nop ({r['now']} -> {r['straddles'][r['shift']]})", file=sys.stderr)
        (rows3 if kind == "3d" else rowsp).append((key, r))
    with open(path + ".tmp", "w") as f:
        f.write("// generated by tools/tune_phase.py — do not edit.  s_nop 0 (4 bytes each) in front of the unrolled substep loop\n"
                "// of the instantiations the built-in models run: the shift that leaves the fewest 8-byte instructions\n"
                "// straddling a 32-byte fetch boundary (straddles per loop iteration before -> after in the comments).\n")
        f.write("constexpr int mbd_pad_3d(int maxcol, int rk) {\n  return ")
        for (mc, rk), r in rows3:
            f.write(f"(maxcol == {mc} && rk == {rk}) ? {r['pad']} /* {r['now']} -> {r['straddles'][r['shift']]} */\n       : ")
        f.write("0;\n}\n")
        f.write("constexpr int mbd_pad_planar(int lps, int maxcol, int d0, int d1, int fl, int rk, int nfr) {\n  return ")
        for (lps, mc, d0, d1, fl, rk, nfr), r in rowsp:
            f.write(f"(lps == {lps} && maxcol == {mc} && d0 == {d0} && d1 == {d1} && fl == {fl} && rk == {rk} && nfr == {nfr}) ? {r['pad']} "
                    f"/* {r['now']} -> {r['straddles'][r['shift']]} */\n       : ")
        f.write("0;\n}\n")
    os.replace(path + ".tmp", path)


def verify():
    """The library's kernels (pads on) must show the straddle counts the tuning predicted."""
    res = generate(verbose=False, check=False)
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(TARGETS)) as ex:
        got = list(ex.map(lambda t: analyse(t[0], t[1], tuned=True), TARGETS))
    ok = True
    for (kind, targs, _), r, g in zip(TARGETS, res, got):
        want = r["straddles"][r["shift"]]
        print(f"{kind:6s} <{targs}>: predicted {want}, built {g['now']}, loop {g['instructions']} instructions")
        ok = ok and want == g["now"] and g["instructions"] == r["instructions"]
    return ok


if __name__ == "__main__":
    if "--verify" in sys.argv:
        sys.exit(0 if verify() else 1)
    generate()
