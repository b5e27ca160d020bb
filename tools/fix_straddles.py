"""Assembly post-pass for the rollout kernels: no 8-byte instruction left straddling a 32-byte fetch boundary.

A lone wavefront per SIMD fetches its code in 32-byte pieces, and an 8-byte instruction that straddles such a boundary
costs it ~0.8 issue slots more than one that does not (builds that differ ONLY in where the substep loop
starts differ in kernel time in the order of their straddle counts — ~57 straddles per humanoid substep, 4 % of it).
Every VOP1 / VOP2 / VOPC instruction has an 8-byte VOP3 encoding of identical semantics (`_e32` -> `_e64`): promoting
one 4-byte instruction in front of a would-be straddler moves the straddler to the next fetch piece for 4 bytes of
code.  This pass walks each rollout kernel of the compiler's assembly, and wherever an 8-byte instruction would start
at offset 28 (mod 32) re-encodes the nearest preceding promotable instruction of the same fetch piece.  Nothing is
reordered, added or removed: the instruction stream — and every hazard the compiler resolved — is unchanged.

    fix(asm_text) -> (fixed_text, stats)       used by __graft_entry__.build(); falls back to the plain build on any doubt
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
STRADDLE, PROMOTE = 0.8, 0.125  # issue slots: a straddling 8-byte instruction; 4 more bytes of code (1 slot per 32)
TARGET_FUNCS = ("3mbd14rollout_kernel", "3mbd21rollout_planar_kernel", "3mbd18rollout_pk2_kernel")  # (mangled: not car2d_rollout_kernel)


def _is_instr(line):
    t = line.strip()
    return line.startswith("\t") and t and not t.startswith((".", ";", "//"))


def _assemble_sizes(asm_text, td):
    """{function: [size of each instruction in order]} from assembling the text and disassembling the object."""
    s, o = os.path.join(td, "a.s"), os.path.join(td, "a.o")
    with open(s, "w") as f:
        f.write(asm_text)
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o],
                   check=True, capture_output=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", o], capture_output=True, text=True, check=True).stdout
    sizes, cur = {}, None
    for x in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", x)
        if m:
            cur = m.group(1)
            sizes.setdefault(cur, [])
            continue
        m = re.search(r"//\s+[0-9A-F]+:\s+((?:[0-9A-F]{8}\s*)+)", x)
        if m and cur is not None:
            sizes[cur].append((4 * len(m.group(1).split()), x.strip().split()[0]))
    return sizes


def _base(mnemonic):
    return re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mnemonic)


def _functions(lines):
    """[(name, first line, last line)] of the global functions of the assembly."""
    out, cur = [], None
    for k, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and cur is None:
            cur = (m.group(1), k)
        if cur is not None and l.startswith(".Lfunc_end"):
            out.append((cur[0], cur[1], k))
            cur = None
    return out


def fix(asm_text):
    lines = asm_text.split("\n")
    stats = {}
    with tempfile.TemporaryDirectory() as td:
        sizes = _assemble_sizes(asm_text, td)
        for name, a, b in _functions(lines):
            if not any(t in name for t in TARGET_FUNCS):
                continue
            idx = [k for k in range(a, b) if _is_instr(lines[k])]
            sz = sizes.get(name)
            # (the object may carry padding behind the function's last instruction; everything up to there must match
            # the assembly instruction by instruction)
            if sz is None or len(sz) < len(idx):
                raise RuntimeError(f"{name}: {len(idx)} instructions in the assembly, {0 if sz is None else len(sz)} in the object")
            for i, k in enumerate(idx):
                if _base(lines[k].strip().split()[0]) != _base(sz[i][1]):
                    raise RuntimeError(f"{name}: instruction {i}: '{lines[k].strip()}' in the assembly, '{sz[i][1]}' in the object")
            sz = [x[0] for x in sz[:len(idx)]]
            promotable = [bool(re.match(r"^v_\w+_e32\b", lines[k].strip())) and sz[i] == 4 for i, k in enumerate(idx)]
            before, o = 0, 0
            for s_ in sz:
                before += s_ == 8 and o % 32 == 28
                o += s_
            # how often an instruction runs: 30^(loop depth), loops = regions closed by a backward branch to a label
            lab, depth = {}, [0] * len(idx)
            pos = {k: i for i, k in enumerate(idx)}
            nxt = len(idx)
            first_instr_at = [0] * (b - a + 1)
            for k in range(b, a - 1, -1):  # line -> index of the first instruction at or after it
                if k in pos:
                    nxt = pos[k]
                first_instr_at[k - a] = nxt
            for k in range(a, b):
                m = re.match(r"^(\.LBB\d+_\d+):", lines[k])
                if m:
                    lab[m.group(1)] = first_instr_at[k - a]
            for i, k in enumerate(idx):
                m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[k])
                if m and m.group(1) in lab and lab[m.group(1)] <= i:
                    for t in range(lab[m.group(1)], i + 1):
                        depth[t] += 1
            weight = [30.0 ** min(d, 3) for d in depth]
            # Dynamic programme over (instruction, offset mod 32): which promotable instructions to re-encode so that
            # the weighted cost — STRADDLE per straddling 8-byte instruction, PROMOTE per 4 bytes of added code — is least
            INF = float("inf")
            n = len(idx)
            cost = [[INF] * 8 for _ in range(n + 1)]
            back = [[None] * 8 for _ in range(n + 1)]
            cost[0][0] = 0.0
            for i in range(n):
                w = weight[i]
                for ph in range(8):
                    c = cost[i][ph]
                    if c == INF:
                        continue
                    for prom in ((False, True) if promotable[i] else (False,)):
                        size = 8 if prom else sz[i]
                        cc = c + (PROMOTE * w if prom else 0.0) + (STRADDLE * w if (size == 8 and ph == 7) else 0.0)
                        np_ = (ph + size // 4) % 8
                        if cc < cost[i + 1][np_]:
                            cost[i + 1][np_] = cc
                            back[i + 1][np_] = (ph, prom)
            ph = min(range(8), key=lambda q: cost[n][q])
            promoted = 0
            for i in range(n, 0, -1):
                pph, prom = back[i][ph]
                if prom:
                    k = idx[i - 1]
                    lines[k] = re.sub(r"^(\s*v_\w+)_e32\b", r"\1_e64", lines[k], count=1)
                    sz[i - 1] = 8
                    promoted += 1
                ph = pph
            left, o = 0, 0
            for s_ in sz:
                left += s_ == 8 and o % 32 == 28
                o += s_
            stats[name] = dict(instructions=len(idx), straddles_before=before, promoted=promoted, left=left)
        fixed = "\n".join(lines)
        # the promoted encodings must assemble to exactly the predicted sizes, and no straddle may remain unaccounted for
        sizes2 = _assemble_sizes(fixed, td)
        for name, st in stats.items():
            o, n = 0, 0
            for s_, _ in sizes2[name][:st["instructions"]]:
                n += s_ == 8 and o % 32 == 28
                o += s_
            if len(sizes2[name]) < st["instructions"] or n != st["left"]:
                raise RuntimeError(f"{name}: after the pass {n} straddles (expected {st['left']}), {len(sizes2[name])} instructions")
            st["straddles_after"] = n
    return fixed, stats


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    out, st = fix(text)
    open(sys.argv[2], "w").write(out)
    tot = lambda k: sum(v[k] for v in st.values())
    print(f"{len(st)} kernels: {tot('straddles_before')} straddling 8-byte instructions -> {tot('straddles_after')}, "
          f"{tot('promoted')} instructions re-encoded e32 -> e64")
