"""What the data-level switch `collide_all_capsules` costs: steps/s of 40-step plans at the BASELINE sizes with the
shipped models (feet only) and with every capsule colliding.  usage (GPU box): python tools/gpu_collide_all_cost.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    sys.path.insert(0, p)


def main():
    from mbd_hip import _capi, mjcf
    from mbd_hip.envs import specs
    from mbd_hip.envs.base import RigidBodyEnv
    from mbd_hip.planners.mbd_planner import Args, Plan
    print("| env, N | feet only (shipped) | every capsule | colliders, per link max, planar kernels? |")
    print("|---|---:|---:|---|")
    for name, N, temp in (("hopper", 512, 0.1), ("walker2d", 1024, 0.1), ("halfcheetah", 1024, 0.4)):
        row, note = [], ""
        for ca in (False, True):
            spec = specs.SPECS[name]
            m = mjcf.load(os.path.join(ROOT, "model-based-diffusion_amd", "assets", spec["xml"]), env_name=name, n_frames=spec["n_frames"],
                          reset_noise=spec["reset_noise"], reward_params=spec.get("reward_params", ()),
                          gear_override=spec.get("gear_override", ()), collide_all_capsules=ca, warn_unstable=False)
            env = RigidBodyEnv(name, model=m)
            st = env.reset(_capi.prng_key(1))
            best = 0.0
            for rep in range(3):
                p = Plan(env, Args(env_name=name, Nsample=N, Hsample=50, Ndiffuse=41, temp_sample=temp, disable_recommended_params=True, not_render=True))
                p.set_state0(st)
                best = max(best, 40.0 / p.run(_capi.prng_key(3))[3])
                p.close()
            row.append(f"{best:.0f}")
            if ca:
                cl = list(m.fields["col_link"][:int(m.fields["n_col"])])
                note = f"{int(m.fields['n_col'])}, {max(cl.count(l) for l in set(cl))}, {'yes' if int(m.fields['flags']) & 2 else 'no: general 3-D'}"
        print(f"| {name}, {N} | {row[0]} | {row[1]} | {note} |", flush=True)


if __name__ == "__main__":
    main()
