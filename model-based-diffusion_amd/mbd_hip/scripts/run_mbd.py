"""Drop-in for mbd/scripts/run_mbd.py (:17-64): the 8-seed sweep and the 8-temperature sweep.

The reference runs the plans one after another and times each run end to end (`time()` around
`run_diffusion`, :21,34).  Here the independent plans of a sweep run TOGETHER: plans of one env with the same
sizes and schedule — both of the reference's sweeps — go through `mbd_sweep_*` in lockstep, one rollout launch over
all their candidates and one score launch per diffusion step (at N=1024 a plan's rollout is 256 wavefronts, a quarter
of the chip's SIMDs; eight of them fill it); anything else (mixed sizes) is enqueued concurrently, one `mbd_plan`
and one HIP stream per plan, stepped round-robin from the host.  Results are bit-identical to running the plans
sequentially either way (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, replace

import numpy as np

from .. import _capi
from ..envs import get_env
from ..envs.base import prng_impl
from ..planners import mbd_planner
from ..planners.mbd_planner import Plan, apply_recommended


@dataclass
class Args:
    algo: str = "mbd"  # path_integral, mbd
    update_method: str = "mppi"
    mode: str = "seed"  # "seed" | "temp"
    env_name: str = "ant"


def _resolved(a):
    """a copy of the plan's Args with the recommended overrides applied (mbd_planner.py:64-69), quietly"""
    import contextlib
    import io
    a = replace(a)
    with contextlib.redirect_stdout(io.StringIO()):
        apply_recommended(a)
    return a


def _batchable(plan_args):
    """One env, one set of sizes / schedule, <= 32 plans of <= 12288 candidates: what mbd_sweep_* takes."""
    from dataclasses import asdict
    ds = []
    for a in plan_args:
        d = asdict(_resolved(a))
        for k in ("seed", "temp_sample", "not_render"):
            d.pop(k, None)
        ds.append(d)
    return (1 < len(plan_args) <= 32 and all(d == ds[0] for d in ds) and ds[0]["Nsample"] * 4 <= 48 * 1024
            and ds[0]["env_name"] not in ("car2d", "pushT"))


def run_sweep(plan_args, device: int = 0):
    """The plans as ONE mbd_sweep (lockstep, one rollout launch per diffusion step).  Same returns as run_concurrent."""
    from ..planners.mbd_planner import Sweep
    impl = prng_impl()
    a0 = replace(plan_args[0])
    apply_recommended(a0)
    env = get_env(a0.env_name, device=device)
    sweep = Sweep(env, a0, len(plan_args), temps=[_resolved(a).temp_sample for a in plan_args])
    keys = []
    for k, a in enumerate(plan_args):
        rng = _capi.prng_key(a.seed)  # mbd_planner.py:40
        rng, rng_reset = _capi.prng_split(rng, 2, impl)  # :79
        sweep.set_state0(k, env.reset(rng_reset))
        rng_exp, _ = _capi.prng_split(rng, 2, impl)  # :150
        keys.append(rng_exp)
    mu, _, rews, secs = sweep.run(np.array(keys, np.uint32))  # (secs: the lockstep loop, like run_concurrent's)
    sweep.close()
    return [float(r) for r in rews], [m for m in mu], secs


def run_concurrent(plan_args, device: int = 0, batched: bool | None = None):
    """Run several independent MBD plans (a list of mbd_planner.Args) together on one GPU: as one sweep (lockstep, one
    launch per step) when they share env, sizes and schedule (batched=None decides; True / False force), as concurrent
    plans on separate streams otherwise.  Returns (rew_final list, mu_0ts list, wall seconds of the whole batch)."""
    import torch
    if batched is None:
        batched = _batchable(plan_args)
    if batched:
        return run_sweep(plan_args, device)
    dev = torch.device("cuda", device)
    impl = prng_impl()
    jobs = []
    for a in plan_args:
        a = replace(a)
        rng = _capi.prng_key(a.seed)  # mbd_planner.py:40
        apply_recommended(a)
        env = get_env(a.env_name, device=device)
        rng, rng_reset = _capi.prng_split(rng, 2, impl)  # :79
        state_init = env.reset(rng_reset)
        rng_exp, _ = _capi.prng_split(rng, 2, impl)  # :150
        plan = Plan(env, a, shares_device=len(plan_args) > 1)
        plan.set_state0(state_init)
        HNu = a.Hsample * env.action_size
        jobs.append(dict(args=a, env=env, plan=plan, stream=torch.cuda.Stream(dev),
                         key=(C.c_uint32 * 2)(int(rng_exp[0]), int(rng_exp[1])),
                         Ybar=torch.zeros(HNu, dtype=torch.float32, device=dev),
                         rew=torch.zeros(1, dtype=torch.float32, device=dev),
                         mu=torch.zeros((a.Ndiffuse - 1, HNu), dtype=torch.float32, device=dev)))
    torch.cuda.synchronize(dev)
    t0 = time.time()
    nd_max = max(j["args"].Ndiffuse for j in jobs)
    for step in range(nd_max - 1):  # round-robin: one diffusion step of every plan per pass
        for j in jobs:
            a = j["args"]
            i = a.Ndiffuse - 1 - step
            if i < 1:
                continue
            s = j["stream"]
            _capi.check(j["plan"].lib.mbd_plan_reverse_once(j["plan"].h, i, j["key"], j["Ybar"].data_ptr(),
                                                            j["rew"].data_ptr(), s.cuda_stream))
            with torch.cuda.stream(s):
                j["mu"][step].copy_(j["Ybar"], non_blocking=True)
    torch.cuda.synchronize(dev)
    secs = time.time() - t0
    rews, mus = [], []
    for j in jobs:
        a = j["args"]
        mu = j["mu"].cpu().numpy().reshape(a.Ndiffuse - 1, a.Hsample, -1)
        rews.append(j["plan"].eval(mu[-1]))  # mbd_planner.py:179-180
        mus.append(mu)
        j["plan"].close()
    return rews, mus, secs


def replica_bounds(P: int, world: int, rank: int):
    """Plans [begin, begin + count) of a sweep of P independent plans owned by `rank`: contiguous, as equal as possible
    (the first P % world ranks take one more)."""
    if P < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError(f"replica_bounds({P}, {world}, {rank})")
    base, extra = divmod(P, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def run_replicated(plan_args, device: int | None = None, group=None):
    """The multi-GPU form of a sweep (run_mbd.py:17-64): its plans are independent, so with ``torch.distributed``
    initialised they are REPLICAS over the ranks — rank r runs plans replica_bounds(P, world, r) on its GPU through
    run_concurrent (one mbd_sweep per rank), nothing is exchanged while they run, and the results are gathered ONCE at the
    end (all_gather_object of the final rewards and mu_0ts).  Every rank returns the same (rew_final list in plan order,
    mu_0ts list, max over ranks of the batch seconds); bit-identical to run_concurrent(plan_args) on one GPU."""
    import os
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return run_concurrent(plan_args, 0 if device is None else device)
    rank = dist.get_rank(group)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    begin, count = replica_bounds(len(plan_args), world, rank)
    own = list(plan_args[begin:begin + count])
    rews, mus, secs = run_concurrent(own, device) if own else ([], [], 0.0)
    box = [None] * world
    dist.all_gather_object(box, ([float(r) for r in rews], [np.asarray(m) for m in mus], float(secs)), group=group)
    return ([r for part in box for r in part[0]], [m for part in box for m in part[1]], max(part[2] for part in box))


def _run_path_integral_seq(arg_list, device):
    """The path-integral baselines (run_mbd.py:22-26,46-50) one plan after another, each timed end to end like the
    reference does (`time()` around the call, :21,34) — what plans that do not batch fall back to."""
    from ..planners import path_integral
    rews, times = [], []
    for a in arg_list:
        t0 = time.time()
        rews.append(path_integral.run_path_integral(a, device=device))
        times.append(time.time() - t0)
    return np.array(rews), np.array(times)


def run_path_integral_sweep(arg_list, device: int = 0, return_details: bool = False):
    """The path-integral plans of a sweep (a list of path_integral.Args: one env, one update_method, the same sizes; seeds
    and temperatures may differ) as ONE mbd_sweep: per refinement step one sampling launch, one rollout launch over all
    the plans' candidates and the update rule's kernels with one row per plan.  Bit-identical to running
    path_integral.run_path_integral on each.  Returns (rew_final array, seconds of the lockstep loop[, details])."""
    import contextlib
    import io
    from ..planners import path_integral
    from ..planners.mbd_planner import Sweep
    if len(arg_list) == 0:
        return (np.zeros(0, np.float32), 0.0, {}) if return_details else (np.zeros(0, np.float32), 0.0)
    impl = prng_impl()
    resolved = []
    for a in arg_list:
        a = replace(a)
        if not a.disable_recommended_params:  # path_integral.py:86-92
            a.temp_sample = path_integral.TEMP_RECOMMEND.get(a.env_name, a.temp_sample)
            a.Nrefine = path_integral.NREFINE_RECOMMEND.get(a.env_name, a.Nrefine)
            a.Nsample = path_integral.NSAMPLE_RECOMMEND.get(a.env_name, a.Nsample)
            a.Hsample = path_integral.HSAMPLE_RECOMMEND.get(a.env_name, a.Hsample)
        resolved.append(a)
    a0 = resolved[0]
    same = all((a.env_name, a.update_method, a.Nsample, a.Hsample, a.Nrefine) ==
               (a0.env_name, a0.update_method, a0.Nsample, a0.Hsample, a0.Nrefine) for a in resolved)
    if not same or len(resolved) > 32 or a0.Nsample * 4 > 48 * 1024 or a0.env_name in ("car2d", "pushT"):
        rews, times = _run_path_integral_seq(arg_list, device)
        return (rews, float(times.sum()), None) if return_details else (rews, float(times.sum()))
    env = get_env(a0.env_name, device=device)
    sweep = Sweep(env, a0, len(resolved), temps=[a.temp_sample for a in resolved],
                  update_method=path_integral.UPDATE_METHODS[a0.update_method])
    keys = []
    for k, a in enumerate(resolved):
        rng = _capi.prng_key(a.seed)  # path_integral.py:57
        rng, rng_reset = _capi.prng_split(rng, 2, impl)  # :99
        sweep.set_state0(k, env.reset(rng_reset))
        rng_exp, _ = _capi.prng_split(rng, 2, impl)  # :144
        keys.append(rng_exp)
    mu, rew_means, rews, secs = sweep.run(np.array(keys, np.uint32))
    sigmas = sweep.get_sigmas()
    sweep.close()
    if return_details:
        return np.array(rews), secs, dict(mu_0ts=mu, rew_means=rew_means, sigma_final=sigmas)
    return np.array(rews), secs


def _local_device(device):
    """The GPU a sweep of this process runs on: the argument, else LOCAL_RANK under torch.distributed.run, else 0."""
    import os
    return int(os.environ.get("LOCAL_RANK", "0")) if device is None else device


def _time_line(secs, n_plans):
    """The reference prints `time: mean \\pm std` over its eight sequential runs (run_mbd.py:39); a lockstep batch has one
    wall time, so every plan's share is the same and the spread is 0 — printed in the reference's format, the batch beside it."""
    per = secs / max(n_plans, 1)
    return f"time: {per:.2f} \\pm {0.0:.2f} (per plan; {secs:.2f} s for the lockstep batch of {n_plans})"


def run_multiple_seed(args: Args, device: int | None = None, **plan_kw):
    """run_mbd.py:17-39: seeds 0..7, mean +- std of the final reward and the time."""
    if args.algo == "path_integral":  # :22-26
        from ..planners import path_integral
        plans = [path_integral.Args(seed=seed, env_name=args.env_name, update_method=args.update_method, **plan_kw)
                 for seed in range(8)]
        # (one sweep on this process's GPU: one rollout launch per refinement step)
        rews, secs = run_path_integral_sweep(plans, _local_device(device))
        print(f"rew: {rews.mean():.2f} \\pm {rews.std():.2f}")
        print(_time_line(secs, len(plans)))
        return rews, float(secs)
    if args.algo != "mbd":
        raise NotImplementedError  # :32-33
    plans = [mbd_planner.Args(seed=seed, env_name=args.env_name, not_render=True, **plan_kw) for seed in range(8)]
    rews, _, secs = run_replicated(plans, device)  # (one GPU: run_concurrent; several: the plans as replicas over ranks)
    rews = np.array(rews)
    print(f"rew: {rews.mean():.2f} \\pm {rews.std():.2f}")
    print(_time_line(secs, len(plans)))
    return rews, secs


def run_multiple_temp(args: Args, device: int | None = None, **plan_kw):
    """run_mbd.py:42-64: temperature sweep at seed 0 (mbd: recommended params disabled; path_integral: the
    reference neither disables them nor forwards update_method, :46-50 — replicated, not "fixed")."""
    temps = np.array([0.01, 0.03, 0.06, 0.1, 0.2, 0.4, 0.6, 0.8])
    if args.algo == "path_integral":
        from ..planners import path_integral
        rews, _ = run_path_integral_sweep(
            [path_integral.Args(seed=0, env_name=args.env_name, temp_sample=float(t), **plan_kw) for t in temps],
            _local_device(device))
    elif args.algo == "mbd":
        plans = [mbd_planner.Args(seed=0, env_name=args.env_name, temp_sample=float(t), not_render=True,
                                  disable_recommended_params=True, **plan_kw) for t in temps]
        rews, _, _ = run_replicated(plans, device)
        rews = np.array(rews)
    else:
        raise NotImplementedError
    best_temp = temps[np.argmax(rews)]
    print(f"rews: {rews}")
    print(f"best_temp: {best_temp:.2f}")
    return rews, best_temp


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    for f in Args.__dataclass_fields__.values():
        p.add_argument(f"--{f.name}", type=str, default=f.default)
    a = Args(**vars(p.parse_args()))
    if a.mode == "seed":
        run_multiple_seed(a)
    elif a.mode == "temp":
        run_multiple_temp(a)
    else:
        raise NotImplementedError
