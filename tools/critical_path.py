"""The latency floor of a substep, measured instead of asserted: the longest chain of DEPENDENT
operations in one physics substep of one candidate, from the op counter compiled into the CPU restatement
(oracle/count_ops.cc: every value carries the depth of the chain that produced it).  Protocol as tools/count_ops.py: reset
pose, 12 control steps with every action at 0.3 (contacts active), then 8 substeps with depth tracking; the growth of
the deepest state per substep is the recurrence's critical path — what one substep costs on hardware with unlimited lanes.

    python tools/critical_path.py r05        -> profiles/r05_critical_path.json

Beside it: the instructions per substep the kernels issue today (profiles/<tag>_static_flops.json, when present) and the
ratio — the parallelism a wider layout (more lanes per candidate) could still harvest, at best."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    sys.path.insert(0, p)
from mbd_hip.envs import specs  # noqa: E402
from mbd_hip.model import Model  # noqa: E402
from oracle import oracle as orc_mod  # noqa: E402

KERNEL_OF = {"humanoidrun": "humanoidrun", "humanoidtrack": "humanoidtrack", "humanoidstandup": "humanoidstandup_help", "ant": "ant",
             "hopper": "hopper_planar", "halfcheetah": "halfcheetah_planar", "walker2d": "walker2d_planar", "cartpole": "cartpole_planar"}


def measure(name, orc, n_sub=8):
    with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{name}.json")) as f:
        m = Model.from_json(f.read())
    ms = m.to_struct()
    s = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
    a = np.full(m.act_size(), 0.3, np.float32)
    for _ in range(12):
        s, _ = orc.env_step(ms, s, a)
    d = orc_mod.depth_substeps(ms, s, a, n_sub).astype(np.int64)
    growth = np.diff(d[:, -1])
    per_link = (d[-1, :-1] - d[-2, :-1]).tolist()
    counts, _ = orc_mod.count_substep(ms, s, a)
    ops = counts["add"] + counts["mul"] + counts["fma"] + counts["div"] + counts["sqrt"] + counts["cmp"]
    return {"links": m.n_links, "first_substep_depth": int(d[0, -1]), "depth_per_substep": int(growth[-1]),
            "depth_per_substep_by_link": per_link, "growth_all_substeps": growth.tolist(),
            "ops_per_substep_all_links": int(ops), "ops_per_link": ops / m.n_links,
            "ops_per_link_over_depth": ops / m.n_links / float(growth[-1])}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    orc_mod.build()
    orc = orc_mod.Oracle("f32")
    static = {}
    for t in (tag, "r04"):
        p = os.path.join(ROOT, "profiles", f"{t}_static_flops.json")
        if os.path.exists(p):
            static = json.load(open(p))
            break
    out = {"unit": "dependent issue slots (add/mul/fma/min/max/clip/copysign 1, division 7, square root 5; negation and |x| 0)",
           "caveat": "lower bound: a select's dependence on its condition and the link-to-link exchanges (one DPP slot each) are not seen"}
    for name in specs.SPECS:
        r = measure(name, orc)
        k = static.get(KERNEL_OF.get(name, ""), {})
        if k:
            r["kernel_instructions_per_substep"] = k["instructions_per_substep"]
            r["kernel_instructions_over_depth"] = k["instructions_per_substep"] / r["depth_per_substep"]
        out[name] = r
        print(name, json.dumps(r))
    with open(os.path.join(ROOT, "profiles", f"{tag}_critical_path.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
