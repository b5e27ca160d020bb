"""What the DATA-level guess "only the feet collide" does (round-5 verdict item 6 / weak 7): for the seed-0 plans of
hopper, walker2d and halfcheetah on the CPU checker — candidates drawn around the plan's own Ybar at five diffusion stages —
the share of candidates whose TORSO origin goes below z = 0 at some control step, with the shipped models (feet only) and with
`collide_all_capsules` (every capsule end a sphere collider).  CPU only (oracle/): python tools/model_guess_report.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    sys.path.insert(0, p)


def compile_env(name, collide_all):
    from mbd_hip import mjcf
    from mbd_hip.envs import specs
    spec = specs.SPECS[name]
    path = os.path.join(ROOT, "model-based-diffusion_amd", "assets", spec["xml"])
    return mjcf.load(path, env_name=name, n_frames=spec["n_frames"], track_names=("torso",), reset_noise=spec["reset_noise"],
                     reward_params=spec.get("reward_params", ()), gear_override=spec.get("gear_override", ()),
                     collide_all_capsules=collide_all, warn_unstable=False)


def report(orc, name, N=256, H=50, temp=0.1, stages=(99, 75, 50, 25, 1), collide_all=False):
    from oracle import planner as op
    m = compile_env(name, collide_all)
    env = op.OracleEnv(orc, name, m.to_struct(), init_q=m.init_q)
    res = op.run_diffusion(orc, env, 0, N, H, 100, temp, impl=1)
    _, _, sigmas = orc.schedule(1e-4, 1e-2, 100)
    ms, st = m.to_struct(), res["state_init"]
    out = []
    for i in stages:
        Ybar = res["mu_0ts"][99 - i - 1] if i < 99 else np.zeros_like(res["mu_0ts"][0])
        eps = orc.normal(orc.prng_key(1000 + i), (N, H, m.act_size()))
        Y = np.clip(eps * np.float32(sigmas[i]) + Ybar, -1, 1).astype(np.float32)
        _, xpos = orc.rollout(ms, st, Y, want_xpos=True)
        z = xpos[:, :, 0, 2]
        out.append((i, float((z.min(axis=1) < 0.0).mean()), float(z.min())))
    return int(m.fields["n_col"]), bool(int(m.fields["flags"]) & 2), float(res["rew_final"]), out


def main():
    from oracle import oracle as orc_mod
    orc_mod.build()
    orc = orc_mod.Oracle("f32_omp")
    print("| env | colliders | sphere colliders (planar kernels?) | rew_final (seed 0, N=256) | share of candidates whose torso origin dips below z = 0, at i = 99 / 75 / 50 / 25 / 1 | lowest torso z |")
    print("|---|---|---:|---:|---|---:|")
    for name, temp in (("hopper", 0.1), ("walker2d", 0.1), ("halfcheetah", 0.4)):
        for ca in (False, True):
            ncol, planar, rf, rows = report(orc, name, temp=temp, collide_all=ca)
            print(f"| {name} | {'every capsule' if ca else 'feet only (shipped)'} | {ncol} ({'yes' if planar else 'no: 3-D'}) | {rf:.3f} | "
                  + " / ".join(f"{s:.2f}" for _, s, _ in rows) + f" | {min(z for _, _, z in rows):.2f} |", flush=True)


if __name__ == "__main__":
    main()
