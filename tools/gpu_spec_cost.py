"""What a specification switch costs: steps/s of a plan whose model carries the flag word, for every single flag.
A model whose flags differ from the library's tuned word (MBD_TUNED_SPEC: MBD_DEFAULT_SPEC = contact_avg in the shipped build) runs the general SPEC
instantiations (16-lane groups, shuffle exchange, every switch read at run time); one that matches runs the tuned kernels.
usage (GPU box): python tools/gpu_spec_cost.py [--lib PATH]   (the table goes to stdout)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
if "--lib" in sys.argv:
    os.environ["MBD_HIP_LIB"] = sys.argv[sys.argv.index("--lib") + 1]

import numpy as np  # noqa: E402


def main():
    from conftest import load_model
    from mbd_hip import _capi
    from mbd_hip.envs.base import RigidBodyEnv
    from mbd_hip.planners.mbd_planner import Args, Plan
    cases = [("hopper", 512, 0.1), ("halfcheetah", 1024, 0.4), ("walker2d", 1024, 0.1), ("ant", 1024, 0.1),
             ("humanoidstandup", 1024, 0.1), ("humanoidrun", 1024, 0.1)]
    names = {4: "contact_avg (4: the default)", 0: "summed (0)", 8: "gauss_seidel (8)", 16: "friction_vel_bound (16)", 32: "restitution_min (32)",
             64: "euler_extrinsic (64)", 128: "gyroscopic (128)"}
    print(f"library: {os.environ.get('MBD_HIP_LIB', 'lib/libmbd_hip.so')}")
    print("| env, N | " + " | ".join(names[b] for b in names) + " |")
    print("|---|" + "---:|" * len(names))
    for env_name, N, temp in cases:
        row = []
        for bits in names:
            m = load_model(env_name)
            planar = (int(m.fields["flags"]) & 2) != 0
            if bits in (64, 128) and planar:
                row.append("–")  # (no such switch in the plane)
                continue
            env = RigidBodyEnv(env_name, model=m.with_spec(bits))
            st = env.reset(_capi.prng_key(1))
            best = 0.0
            for rep in range(3):
                p = Plan(env, Args(env_name=env_name, Nsample=N, Hsample=50, Ndiffuse=41, temp_sample=temp,
                                   disable_recommended_params=True, not_render=True))
                p.set_state0(st)
                _, _, _, secs = p.run(_capi.prng_key(3))
                p.close()
                best = max(best, 40.0 / secs)
            row.append(f"{best:.0f}")
        print(f"| {env_name}, {N} | " + " | ".join(row) + " |", flush=True)


if __name__ == "__main__":
    main()
