"""Static FP32 operation count of the rollout kernel's substep loop, from the gfx950 ISA hipcc emits.

Counts per LANE per physics substep: v_fma/v_fmac/v_fmaak/v_fmamk = 2 flops, v_pk_fma = 4, v_pk_mul/add = 2,
v_mul/v_add/v_sub = 1, v_rcp/v_sqrt/v_div_* = 1 each (the exact-division and exact-sqrt expansions are
counted by their constituent instructions).  Used for the VALU view of the roofline in DESIGN.md/bench.py:
flops per launch = count * 64 lanes * waves * H * n_frames (all lanes, padding lanes included).
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# env -> template arguments of the instantiation launch_rollout() picks for it (csrc/mbd_env.hip)
INSTANCES = {
    "humanoidrun": "16,true,false,3,1,1,-4,-6,0,false,true,3,false,false,0,7",
    "humanoidtrack": "16,true,false,3,1,1,-4,-6,0,false,true,3,false,false,3,5",
    "humanoidstandup": "16,true,false,3,5,1,-4,-6,0,false,true,3,false,false,4,7",
    "humanoidstandup_help": "16,true,false,3,5,1,-4,-6,0,false,true,3,false,false,4,7,true,true",
    "ant": "16,true,false,4,2,1,-2,-4,-6,false,false,3,false,false,6,10,false,true",
    "halfcheetah": "8,true,true,4,2,1,-3,0,0,false,false,2,true",
    "walker2d": "8,false,true,4,2,1,-3,0,0,true,false,2,true,true",
    "hopper": "4,false,true,4,2,1,0,0,0,true,false,2,true,true",
    "cartpole": "4,true,true,4,2,1,0,0,0,false,false,2,true",
    # the planar restatement (mbd_planar.h: rollout_planar_kernel<LPS, MAXCOL, D0, D1>) the planar models actually run
    # (the last argument: n_frames as a compile-time constant — the loop runs twice over n_frames / 2 substeps in line)
    "hopper_planar": "planar:4,2,1,0,0,1,20",
    "halfcheetah_planar": "planar:8,2,1,-3,1,2,16",
    "walker2d_planar": "planar:8,2,1,-3,0,1,20",
    "cartpole_planar": "planar:4,0,1,0,2,5,4",
    # the EARLY-OUT instantiations (round 6: fewer candidates per wavefront, a wave-uniform branch around the contact code): the
    # loop's fall-through path is the substep WITHOUT a contact; `contact_path_instructions` = that path with the out-of-line
    # contact block in place of what it skips.  Executed counts (PMC) lie between the two: profiles/r06_cpw_pmc.txt
    "hopper_planar_eo": "planar:4,2,1,0,0,1,20,false,true",
    "walker2d_planar_eo": "planar:8,2,1,-3,0,1,20,false,true",
    "halfcheetah_planar_eo": "planar:8,2,1,-3,1,2,0,false,true",
    # two candidates per lane (mbd_pk2.h: rollout_pk2_kernel<MAXCOL, RK, NFR>): the counts are per candidate PAIR
    "humanoidrun_pk2": "pk2:1,0,7",
    "humanoidtrack_pk2": "pk2:1,3,5",
    "humanoidstandup_pk2": "pk2:5,4,7",
    "ant_pk2": "pk2:2,6,10,1,1",
}


def count(targs):
    csrc = os.path.join(ROOT, "model-based-diffusion_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "k.hip")
        kern, hdr = "rollout_kernel", "mbd_kernels.h"
        # (the flags of the translation unit an instantiation is built in, __graft_entry__.TUS: the DPP instantiations of
        # the humanoids and ant are mbd_hot3d.hip's)
        hot = targs.startswith(("16,true,false,3,1,1,-4,-6", "16,true,false,3,5,1,-4,-6", "16,true,false,4,2,1,-2,-4,-6,false,false"))
        extra = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if hot else []
        if targs.startswith("planar:"):
            kern, hdr, targs = "rollout_planar_kernel", "mbd_planar.h", targs[len("planar:"):]
            extra = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]  # (the flags of its translation unit: build())
        if targs.startswith("pk2:"):
            kern, hdr, targs = "rollout_pk2_kernel", "mbd_pk2.h", targs[len("pk2:"):]
            extra = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]  # (the flags of its translation unit: build())
        with open(src, "w") as f:
            f.write(f'#include "{csrc}/{hdr}"\ntemplate __global__ void mbd::{kern}<{targs}>(mbd::RolloutParams);\n')
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", *extra, *os.environ.get("MBD_COUNT_DEFS", "").split(),
                        "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        body = open(out).read().split("\n")
    start = [i for i, l in enumerate(body) if re.match(r"^_ZN3mbd\d+rollout_(planar_|pk2_)?kernel.*:", l)][0]
    end = [i for i, l in enumerate(body) if i > start and ".Lfunc_end" in l][0]
    meta = "\n".join(body[end:])
    body = body[start:end]
    lab = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = k
    loops = []  # (first line, last line, instructions) of every backward branch
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            ins = [x.strip().split()[0] for x in body[lab[m.group(1)]:k + 1]
                   if x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";"))]
            loops.append((lab[m.group(1)], k, ins))
    # the longest backward-branch region is the control-step loop (over H); the substep loop (over n_frames) is the
    # longest region strictly inside it (the other backward branches in there are out-of-line slow paths — the exact
    # square root of the quaternion renormalisation — jumping back into the loop, and the small gather loops)
    outer = max(loops, key=lambda t: len(t[2]))
    inner = [t for t in loops if t[0] > outer[0] and t[1] < outer[1] and len(t[2]) <= len(outer[2]) - 40]
    # the substep loop touches no global memory (actions are fetched per CONTROL step, rewards stored per control step):
    # the longest backward-branch region without a global / flat / scratch access
    nomem = [t for t in loops if not any(x.startswith(("global_", "flat_", "scratch_", "buffer_")) for x in t[2])]
    # ... that does not itself contain another such region of comparable size (the remainder loop of the unrolled
    # substeps can close a region around the unrolled loop)
    cands = sorted(nomem or inner, key=lambda t: -len(t[2]))
    bt = next(t for t in cands
              if not any(u is not t and u[0] >= t[0] and u[1] <= t[1] and len(u[2]) >= 0.45 * len(t[2]) for u in cands))
    best, loop_text = bt[2], body[bt[0]:bt[1] + 1]
    c = collections.Counter(best)
    flops = 0
    for k, v in c.items():
        if k.startswith("v_pk_fma"):
            flops += 4 * v
        elif k.startswith(("v_pk_mul_f32", "v_pk_add_f32")):
            flops += 2 * v
        elif k.startswith(("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32")):  # (incl. v_fmac_f32_dpp)
            flops += 2 * v
        elif k.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_rcp_f32", "v_sqrt_f32", "v_div_")):
            flops += v
    vg = re.search(r"; NumVgprs:\s+(\d+)", meta)
    sc = re.search(r"; ScratchSize:\s+(\d+)", meta)
    # issue slots of a lone wavefront per SIMD (tools/probes/probe_issue.hip): one per instruction, one more for a VALU
    # compare and for a transcendental, N more for an "s_nop N" (a wait state is a whole 4-cycle slot)
    nops = [int(x.split()[1]) for x in loop_text if x.strip().startswith("s_nop")]
    slots = len(best) + sum(v for k, v in c.items() if k.startswith("v_cmp")) + sum(nops) + sum(
        v for k, v in c.items() if k.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_")))
    unroll = 4 if kern == "rollout_planar_kernel" else 2  # (substeps per iteration of the substep loop)
    if kern == "rollout_planar_kernel" and len(targs.split(",")) >= 7 and int(targs.split(",")[6]) > 0:
        unroll = int(targs.split(",")[6]) // 2
    if kern == "rollout_kernel" and len(targs.split(",")) >= 16 and targs.split(",")[15].strip().isdigit() and int(targs.split(",")[15]) > 0:
        unroll = int(targs.split(",")[15]) // 2
    if kern == "rollout_pk2_kernel":
        unroll = int(targs.split(",")[2]) // 2 if int(targs.split(",")[2]) > 1 else 2
    if unroll > 1:
        c = collections.Counter({k: v / unroll for k, v in c.items()})
        flops, slots, nops = flops / unroll, slots / unroll, nops
        best = best[:len(best) // unroll]
    res = {"template_args": targs, "instructions_per_substep": len(best), "issue_slots_estimate": slots,
           "valu_per_substep": sum(v for k, v in c.items() if k.startswith("v_")),
           "lds_instr_per_substep": sum(v for k, v in c.items() if k.startswith("ds_")),
           "dpp_per_substep": sum(v for k, v in c.items() if "dpp" in k),
           "agpr_moves_per_substep": sum(v for k, v in c.items() if k.startswith("v_accvgpr")),
           "s_nop_per_substep": c.get("s_nop", 0), "s_waitcnt_per_substep": c.get("s_waitcnt", 0),
           "fp32_flops_per_lane_substep": flops, "vgpr": int(vg.group(1)) if vg else None,
           "scratch_bytes": int(sc.group(1)) if sc else None}
    # early-out instantiations: forward branches out of the loop into blocks placed behind it that jump back in
    is_instr = lambda x: x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";"))
    cold = []
    for k in range(bt[0], bt[1] + 1):
        m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", body[k])
        if not m or m.group(1) not in lab or lab[m.group(1)] <= bt[1]:
            continue
        t0 = lab[m.group(1)]
        t1 = next((j for j in range(t0, len(body)) if re.search(r"s_branch\s+(\.LBB\d+_\d+)", body[j])), None)
        if t1 is None:
            continue
        back = re.search(r"s_branch\s+(\.LBB\d+_\d+)", body[t1]).group(1)
        if back not in lab or not (bt[0] <= lab[back] <= bt[1]) or lab[back] <= k:
            continue
        n_cold = sum(1 for x in body[t0:t1 + 1] if is_instr(x))
        n_skip = sum(1 for x in body[k + 1:lab[back]] if is_instr(x))
        if n_cold >= 40:  # (the contact blocks; the renormalisation's exact side and the like are a handful of instructions)
            cold.append((n_cold, n_skip))
    if cold:
        res["contact_block_instructions"] = sum(c[0] for c in cold) / len(cold)
        res["contact_path_instructions_per_substep"] = len(best) + sum(c[0] - c[1] for c in cold) / len(cold)
        res["no_contact_path_instructions_per_substep"] = len(best)
    if os.environ.get("MBD_COUNT_DUMP"):  # the substep loop's ISA, for reading
        with open(os.environ["MBD_COUNT_DUMP"], "w") as f:
            f.write("\n".join(loop_text))
    return res, c


def main():
    tag = next((a for a in sys.argv[1:] if a.startswith("r") and a[1:].isdigit()), "r03")
    envs = [a for a in sys.argv[1:] if a in INSTANCES] or list(INSTANCES)
    out = {}
    for env in envs:
        res, c = count(INSTANCES[env])
        out[env] = res
        print(env, json.dumps(res))
        if "--hist" in sys.argv:
            for k, v in c.most_common():
                print(f"{v:7.2f} {k}")
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "profiles", f"{tag}_static_flops.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
