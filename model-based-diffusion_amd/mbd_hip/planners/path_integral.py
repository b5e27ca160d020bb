"""Drop-in for mbd/planners/path_integral.py: MPPI / CMA-ES / CEM baselines on the same HIP rollout kernel.
Same ``Args`` fields and defaults (:17-30), same recommended overrides (:68-92), same RNG chain
(:57,99,144) and update rules (:33-52)."""
from __future__ import annotations

from dataclasses import dataclass


from .. import _capi
from ..envs import get_env
from ..envs.base import prng_impl
from .mbd_planner import Plan


@dataclass
class Args:
    # exp
    seed: int = 0
    disable_recommended_params: bool = False
    update_method: str = "mppi"  # mppi, cma-es, cem
    # env
    env_name: str = "ant"
    # diffusion
    Nsample: int = 2048  # number of samples
    Hsample: int = 50  # horizon
    Nrefine: int = 100  # number of repeat steps
    temp_sample: float = 0.1  # temperature for sampling


UPDATE_METHODS = {"mppi": 1, "cma-es": 2, "cem": 3}
TEMP_RECOMMEND = {"ant": 0.1, "halfcheetah": 0.4, "hopper": 0.1, "humanoidstandup": 0.1, "humanoidrun": 0.1,
                  "walker2d": 0.1, "pushT": 0.2}
NREFINE_RECOMMEND = {"pushT": 200, "humanoidrun": 300}
NSAMPLE_RECOMMEND = {"humanoidrun": 8192}
HSAMPLE_RECOMMEND = {"pushT": 40}


def run_path_integral(args: Args, device: int = 0, return_details: bool = False):
    rng = _capi.prng_key(args.seed)  # :57
    method = UPDATE_METHODS[args.update_method]  # KeyError on an unknown method, like the reference's dict (:59-63)
    if not args.disable_recommended_params:  # :86-92
        args.temp_sample = TEMP_RECOMMEND.get(args.env_name, args.temp_sample)
        args.Nrefine = NREFINE_RECOMMEND.get(args.env_name, args.Nrefine)
        args.Nsample = NSAMPLE_RECOMMEND.get(args.env_name, args.Nsample)
        args.Hsample = HSAMPLE_RECOMMEND.get(args.env_name, args.Hsample)
        print(f"override temp_sample to {args.temp_sample}")
    env = get_env(args.env_name, device=device)  # :93
    impl = prng_impl()
    rng, rng_reset = _capi.prng_split(rng, 2, impl)  # :99
    state_init = env.reset(rng_reset)
    rng_exp, rng = _capi.prng_split(rng, 2, impl)  # :144
    plan = Plan(env, args, update_method=method)
    plan.set_state0(state_init)
    mu, rew_means, rew_final, secs = plan.run(rng_exp)  # update() :130-142 and eval_us(...).mean() :146
    sigma = plan.get_sigma()
    plan.close()
    if return_details:
        return rew_final, dict(mu_0ts=mu, rew_means=rew_means, loop_seconds=secs, sigma_final=sigma,
                               state_init=state_init)
    return rew_final


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    for f in Args.__dataclass_fields__.values():
        if isinstance(f.default, bool):
            p.add_argument(f"--{f.name}", action="store_true")
        else:
            p.add_argument(f"--{f.name}", type=type(f.default), default=f.default)
    rew = run_path_integral(Args(**vars(p.parse_args())))
    print(f"rew: {rew:.2e}")  # :153
