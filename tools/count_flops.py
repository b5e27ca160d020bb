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
KERNEL = "_ZN3mbd14rollout_kernelILi16ELb1ELb0ELi3ELi1ELi1ELin4ELin6ELi0EEEvNS_13RolloutParamsE"


def main():
    csrc = os.path.join(ROOT, "model-based-diffusion_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "k.hip")
        with open(src, "w") as f:
            f.write(f'#include "{csrc}/mbd_kernels.h"\ntemplate __global__ void mbd::rollout_kernel<16,true,false,3,1,1,-4,-6>(mbd::RolloutParams);\n')
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-S", "--cuda-device-only", src,
                        "-o", out], check=True, capture_output=True)
        body = open(out).read().split("\n")
    start = [i for i, l in enumerate(body) if l.startswith(KERNEL + ":")][0]
    end = [i for i, l in enumerate(body) if i > start and ".Lfunc_end" in l][0]
    body = body[start:end]
    lab = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            ins = [x.strip().split()[0] for x in body[lab[m.group(1)]:k + 1]
                   if x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";"))]
            loops.append((sum(1 for i in ins if i.startswith("ds_")), sum(1 for i in ins if "dpp" in i), ins))
    # the substep loop = the smallest loop holding one substep's DPP row shifts (57 of them: the parent<->child
    # traffic); the control-step loop around it holds them too, but is longer.  (The compiler may rotate a few of
    # the 13 prefetching ds_bpermute of a substep into the loop's entry block, so they are not a reliable marker.)
    best = min((ins for n, d, ins in loops if d >= 50), key=len)
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
    res = {"instructions_per_substep": len(best), "valu_per_substep": sum(v for k, v in c.items() if k.startswith("v_")),
           "lds_instr_per_substep": sum(v for k, v in c.items() if k.startswith("ds_")), "fp32_flops_per_lane_substep": flops}
    print(json.dumps(res))
    if "--hist" in sys.argv:
        for k, v in c.most_common():
            print(f"{v:5d} {k}")
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "profiles", "r01_static_flops.json"), "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
