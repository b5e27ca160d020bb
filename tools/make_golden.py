"""Self-goldens: outputs of this repo's CPU oracle on fixed seeds, committed under tests/golden/ so that
any change of the numerical contract (oracle/spec_math.h, physics stages) is a visible diff."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    sys.path.insert(0, p)
from mbd_hip.model import Model  # noqa: E402
from oracle import oracle as orc_mod, planner as op  # noqa: E402

CASES = [("car2d", 0, 32, 30, 50, 0.1, 1, 6), ("humanoidrun", 0, 16, 12, 20, 0.1, 1, 3),
         ("humanoidrun", 1, 16, 12, 20, 0.1, 0, 3), ("hopper", 0, 16, 12, 20, 0.1, 1, 3),
         ("halfcheetah", 0, 16, 12, 20, 0.4, 1, 3), ("humanoidtrack", 0, 16, 12, 20, 0.1, 1, 3),
         ("walker2d", 0, 16, 12, 20, 0.1, 1, 3), ("humanoidstandup", 0, 16, 12, 20, 0.1, 1, 3),
         ("cartpole", 0, 16, 12, 20, 0.1, 1, 3), ("ant", 0, 16, 12, 20, 0.1, 1, 3)]


def main():
    orc_mod.build()
    orc = orc_mod.Oracle("f32")
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, seed, N, H, Nd, temp, impl, steps in CASES:
        if name == "car2d":
            xref = np.load(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", "car2d_xref.npy"))
            env = op.OracleEnv(orc, "car2d", xref=xref)
        else:
            with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{name}.json")) as f:
                m = Model.from_json(f.read())
            env = op.OracleEnv(orc, name, m.to_struct(), init_q=m.init_q)
        r = op.run_diffusion(orc, env, seed, N, H, Nd, temp, impl=impl, max_steps=steps)
        np.savez_compressed(os.path.join(out, f"self_{name}_s{seed}_i{impl}.npz"), env=name, seed=seed, N=N, H=H,
                            Nd=Nd, temp=temp, impl=impl, steps=steps, state_init=r["state_init"],
                            mu_0ts=r["mu_0ts"], rew_means=r["rew_means"])
        print(name, seed, impl, r["rew_means"])


if __name__ == "__main__":
    main()
