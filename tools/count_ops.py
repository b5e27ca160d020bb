"""F_sub(model): flops of ONE physics substep per built-in model from the op counter compiled into the CPU
restatement (oracle/count_ops.cc).  Protocol: reset pose, 12 control steps with every action at
0.3 (the feet are on the ground, contacts active), then one counted substep.  Writes profiles/<round>_op_counts.json,
which bench.py's `valu` block reads when the live count of its cpu_baseline leg is not available."""
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


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    orc_mod.build()
    orc = orc_mod.Oracle("f32")
    out = {}
    for name in specs.SPECS:
        with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{name}.json")) as f:
            m = Model.from_json(f.read())
        ms = m.to_struct()
        s = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
        a = np.full(m.act_size(), 0.3, np.float32)
        for _ in range(12):
            s, _ = orc.env_step(ms, s, a)
        d, _ = orc_mod.count_substep(ms, s, a)
        d["links"] = m.n_links
        d["flops_per_link"] = d["flops"] / m.n_links
        out[name] = d
        print(name, d)
    with open(os.path.join(ROOT, "profiles", f"{tag}_op_counts.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
