"""Drop-in for mbd/planners/mbd_planner.py: same ``Args`` fields and defaults (:18-35), same
recommended-parameter overrides (:45-68), same RNG chain (:40,79,150,103), same return value (:182).

The reverse loop (:138-148) runs through libmbd_hip.so.  With ``torch.distributed`` initialised the N
candidates are sharded over ranks (one process per GPU): per diffusion step each rank rolls out its
shard, ONE all-gather (RCCL over xGMI) exchanges the N mean rewards, and every rank finishes the
step redundantly from identical inputs — results are bit-identical for every world size.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass

import numpy as np

from .. import _capi
from ..envs import get_env
from ..envs.base import prng_impl


@dataclass
class Args:
    # exp
    seed: int = 0
    disable_recommended_params: bool = False
    not_render: bool = False
    # env
    env_name: str = "ant"  # in scope here: "car2d", "hopper", "halfcheetah", "humanoidrun", "humanoidtrack"
    # diffusion
    Nsample: int = 2048  # number of samples
    Hsample: int = 50  # horizon
    Ndiffuse: int = 100  # number of diffusion steps
    temp_sample: float = 0.1  # temperature for sampling
    beta0: float = 1e-4  # initial beta
    betaT: float = 1e-2  # final beta
    enable_demo: bool = False


# mbd_planner.py:45-63
TEMP_RECOMMEND = {"ant": 0.1, "halfcheetah": 0.4, "hopper": 0.1, "humanoidstandup": 0.1, "humanoidrun": 0.1,
                  "walker2d": 0.1, "pushT": 0.2}
NDIFFUSE_RECOMMEND = {"pushT": 200, "humanoidrun": 300}
NSAMPLE_RECOMMEND = {"humanoidrun": 8192}
HSAMPLE_RECOMMEND = {"pushT": 40}


def apply_recommended(args: Args) -> None:
    if not args.disable_recommended_params:  # mbd_planner.py:64-69
        args.temp_sample = TEMP_RECOMMEND.get(args.env_name, args.temp_sample)
        args.Ndiffuse = NDIFFUSE_RECOMMEND.get(args.env_name, args.Ndiffuse)
        args.Nsample = NSAMPLE_RECOMMEND.get(args.env_name, args.Nsample)
        args.Hsample = HSAMPLE_RECOMMEND.get(args.env_name, args.Hsample)
        print(f"override temp_sample to {args.temp_sample}")


class Sweep:
    """Owner of an ``mbd_sweep`` handle: several plans of one env (same sizes and schedule; seeds, start states and
    temperatures may differ) advanced in lockstep — ONE rollout launch over all their candidates and ONE score launch per
    diffusion step (mbd/scripts/run_mbd.py:17-64).  Every plan's result is bit-identical to ``Plan.run`` on its own.
    ``update_method``: 0 MBD plans; 1 / 2 / 3 the path-integral baselines mppi / cma-es / cem (``args`` is then a
    path_integral.Args: Nrefine plays Ndiffuse)."""

    def __init__(self, env, args, n_plans: int, temps=None, literal_score: bool = True, update_method: int = 0):
        self.lib = _capi.load()
        self.env = env
        cfg = _capi.PlanConfig()
        cfg.Nsample, cfg.Hsample = args.Nsample, args.Hsample
        cfg.Ndiffuse = getattr(args, "Ndiffuse", None) or args.Nrefine  # path_integral.Args calls it Nrefine
        cfg.temp_sample = args.temp_sample
        cfg.beta0, cfg.betaT = getattr(args, "beta0", 1e-4), getattr(args, "betaT", 1e-2)
        cfg.enable_demo = int(getattr(args, "enable_demo", False))
        cfg.update_method = int(update_method)
        cfg.prng_impl = prng_impl()
        cfg.shard_begin, cfg.shard_count = 0, args.Nsample
        cfg.literal_score = int(literal_score)
        self.cfg, self.P = cfg, int(n_plans)
        t = None if temps is None else np.ascontiguousarray(temps, np.float32).reshape(self.P)
        h = C.c_void_p()
        _capi.check(self.lib.mbd_sweep_create(env.handle, C.byref(cfg), self.P, None if t is None else _capi.np_ptr(t),
                                              C.byref(h)))
        self.h = h
        self.Nd, self.H, self.Nu = cfg.Ndiffuse, args.Hsample, env.action_size

    def set_state0(self, k: int, state):
        st = np.ascontiguousarray(state.pipeline_state, np.float32).reshape(-1)
        _capi.check(self.lib.mbd_sweep_set_state0(self.h, int(k), _capi.np_ptr(st)))

    def run(self, keys, outputs: bool = True):
        """keys [P, 2] = rng_exp of every plan.  Returns (mu_0ts [P, Nd-1, H, Nu], rew_means [P, Nd-1], rew_final [P],
        seconds of the lockstep loop).  ``outputs=False``: the lockstep loop only — no host copies, no final evaluation
        (timing runs); the three arrays are then None."""
        k = np.ascontiguousarray(keys, np.uint32).reshape(self.P, 2)
        secs = C.c_double()
        if not outputs:
            _capi.check(self.lib.mbd_sweep_run(self.h, _capi.np_ptr(k), None, None, None, C.byref(secs)))
            return None, None, None, secs.value
        mu = np.zeros((self.P, self.Nd - 1, self.H, self.Nu), np.float32)
        rm = np.zeros((self.P, self.Nd - 1), np.float32)
        rf = np.zeros(self.P, np.float32)
        _capi.check(self.lib.mbd_sweep_run(self.h, _capi.np_ptr(k), _capi.np_ptr(mu), _capi.np_ptr(rm), _capi.np_ptr(rf),
                                           C.byref(secs)))
        return mu, rm, rf, secs.value

    def get_sigmas(self):
        """path-integral sweeps: every plan's carried sigma after the last run (path_integral.py:113,131)."""
        out = np.zeros(self.P, np.float32)
        _capi.check(self.lib.mbd_sweep_get_sigmas(self.h, _capi.np_ptr(out)))
        return out

    def kernel_time(self, enable=True):
        ms, n = C.c_float(), C.c_int()
        _capi.check(self.lib.mbd_sweep_kernel_time(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.h is not None:
            self.lib.mbd_sweep_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Thin owner of an ``mbd_plan`` handle."""

    def __init__(self, env, args, shard_begin: int = 0, shard_count: int = None, literal_score: bool = True,
                 update_method: int = 0, shares_device: bool = False):
        self.lib = _capi.load()
        self.env = env
        cfg = _capi.PlanConfig()
        cfg.Nsample, cfg.Hsample = args.Nsample, args.Hsample
        cfg.Ndiffuse = getattr(args, "Ndiffuse", None) or args.Nrefine  # path_integral.Args calls it Nrefine
        cfg.temp_sample = args.temp_sample
        cfg.beta0, cfg.betaT = getattr(args, "beta0", 1e-4), getattr(args, "betaT", 1e-2)
        cfg.enable_demo = int(getattr(args, "enable_demo", False))
        cfg.update_method = update_method
        cfg.prng_impl = prng_impl()
        cfg.shard_begin = shard_begin
        cfg.shard_count = args.Nsample if shard_count is None else shard_count
        cfg.literal_score = int(literal_score)
        cfg.shares_device = int(shares_device)  # other plans run on this GPU at the same time (concurrent sweeps)
        self.cfg = cfg
        h = C.c_void_p()
        _capi.check(self.lib.mbd_plan_create(env.handle, C.byref(cfg), C.byref(h)))
        self.h = h
        self.Nd, self.H, self.Nu = cfg.Ndiffuse, args.Hsample, env.action_size

    def schedule(self):
        a, ab, s = (np.zeros(self.Nd, np.float32) for _ in range(3))
        _capi.check(self.lib.mbd_plan_schedule(self.h, _capi.np_ptr(a), _capi.np_ptr(ab), _capi.np_ptr(s)))
        return a, ab, s

    def set_state0(self, state):
        st = np.ascontiguousarray(state.pipeline_state, np.float32).reshape(-1)
        _capi.check(self.lib.mbd_plan_set_state0(self.h, _capi.np_ptr(st)))

    def enable_timing(self, on=True):
        _capi.check(self.lib.mbd_plan_enable_timing(self.h, int(on)))

    def kernel_time(self, reset=True):
        ms, n = C.c_float(), C.c_int()
        _capi.check(self.lib.mbd_plan_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def run(self, key):
        """Whole reverse loop on one GPU. Returns (mu_0ts [Nd-1,H,Nu], rew_means [Nd-1], rew_final, secs)."""
        mu = np.zeros((self.Nd - 1, self.H, self.Nu), np.float32)
        rm = np.zeros(self.Nd - 1, np.float32)
        rf, secs = C.c_float(), C.c_double()
        _capi.check(self.lib.mbd_plan_run(self.h, _capi.key_array(key), _capi.np_ptr(mu), _capi.np_ptr(rm),
                                          C.byref(rf), C.byref(secs)))
        return mu, rm, rf.value, secs.value

    def get_sigma(self) -> float:
        v = C.c_float()
        _capi.check(self.lib.mbd_plan_get_sigma(self.h, C.byref(v)))
        return v.value

    def set_sigma(self, v: float):
        _capi.check(self.lib.mbd_plan_set_sigma(self.h, float(v)))

    def eval(self, Y) -> float:
        Y = np.ascontiguousarray(Y, np.float32)
        rf = C.c_float()
        _capi.check(self.lib.mbd_plan_eval(self.h, _capi.np_ptr(Y), C.byref(rf)))
        return rf.value

    def peek(self, want_weights=True):
        N, sh = self.cfg.Nsample, self.cfg.shard_count
        Y0s = np.zeros((N, self.H, self.Nu), np.float32)
        rewss = np.zeros((sh, self.H), np.float32)
        w = np.zeros(N, np.float32)
        _capi.check(self.lib.mbd_plan_peek(self.h, _capi.np_ptr(Y0s), _capi.np_ptr(rewss), _capi.np_ptr(w)))
        return Y0s, rewss, w

    def close(self):
        if self.h is not None:
            self.lib.mbd_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostProgress:
    """The per-step mean rewards the reference shows on its progress bar (mbd_planner.py:147), delivered WITHOUT a
    stream synchronisation: one slot per diffusion step in pinned, device-visible host memory, pre-filled with NaN;
    ``mbd_plan_score_update`` is handed the slot's address as its ``d_rew_mean`` and the score kernel's own store
    lands there.  ``wait(k)`` spins on the slot (falling back to a device synchronisation after ``timeout_s``, e.g.
    when the mean itself is NaN); the device meanwhile runs on into the weighted mean and the next step."""

    def __init__(self, n: int, device):
        import torch
        self.t = torch.full((max(n, 1),), float("nan"), dtype=torch.float32).pin_memory()
        self.v = self.t.numpy()
        self.device = device
        self.gave_up = False  # a slot stayed NaN past the timeout (a diverged plan's mean IS NaN): synchronise from then on

    def ptr(self, k: int) -> int:
        return self.t.data_ptr() + 4 * k

    def reset(self, k: int) -> None:
        self.v[k] = np.nan

    def wait(self, k: int, timeout_s: float = 2.0) -> float:
        import torch
        v, t0, spins = self.v, None, 0
        if self.gave_up:  # (a genuinely NaN mean looks like "not written yet": do not spin the timeout again every step)
            torch.cuda.synchronize(self.device)
            return float(v[k])
        while v[k] != v[k]:  # NaN: not written yet
            spins += 1
            if spins & 0xfff == 0:
                t0 = t0 or time.perf_counter()
                if time.perf_counter() - t0 > timeout_s:
                    torch.cuda.synchronize(self.device)
                    self.gave_up = True
                    break
        return float(v[k])


def shard_bounds(N: int, world: int, rank: int):
    """Candidates [begin, begin+count) owned by `rank`: contiguous, equal shards (N % world == 0)."""
    if N % world:
        raise ValueError(f"Nsample={N} must be divisible by the world size {world}")
    sh = N // world
    return rank * sh, sh


class P2PExchange:
    """Owner of an ``mbd_exchange`` handle: the step's all-gather as direct peer writes into every rank's receive
    window (include/mbd_hip.h "in-library exchange") instead of a collective-library call.  The IPC handles of the
    windows travel once, at construction, through the process group (any backend)."""

    def __init__(self, device: int, rows: int, shard: int, group=None):
        """Collective over ``group``: every rank must construct it.  A failure on ANY rank (no IPC, no peer access) is
        agreed on before anybody returns — all ranks raise, nobody is left waiting in a collective."""
        import torch
        import torch.distributed as dist
        self.lib = _capi.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.rows, self.shard = rows, shard
        self.h = None
        why = None
        mine = (C.c_ubyte * 64)()
        try:
            h = C.c_void_p()
            _capi.check(self.lib.mbd_exchange_create(device, self.rank, self.world, rows, shard, C.byref(h)))
            self.h = h
            _capi.check(self.lib.mbd_exchange_local_handle(self.h, mine))
        except Exception as e:  # noqa: BLE001
            why = f"rank {self.rank}: {e}"
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(mine), group=group)
        if why is None:
            try:
                blob = (C.c_ubyte * (64 * self.world)).from_buffer_copy(b"".join(handles))
                _capi.check(self.lib.mbd_exchange_connect(self.h, blob))
            except Exception as e:  # noqa: BLE001
                why = f"rank {self.rank}: {e}"
        on_gpu = dist.get_backend(group) == "nccl"
        ok = torch.tensor([0 if why else 1], dtype=torch.int32, device=torch.device("cuda", device) if on_gpu else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)  # (also the barrier: every window is mapped everywhere)
        if int(ok.item()) == 0:
            self.close()
            raise _capi.MbdError(_capi.MBD_ERR_STATE, why or "the in-library exchange could not be set up on another rank")

    def all_gather(self, local, stream: int) -> int:
        """local: CUDA tensor [rows, shard].  Returns the device address of the gathered [rows, world * shard] values
        (valid until the next call); asynchronous on ``stream``."""
        out = C.c_void_p()
        _capi.check(self.lib.mbd_exchange_all_gather(self.h, local.data_ptr(), C.byref(out), stream))
        return out.value

    def status(self):
        _capi.check(self.lib.mbd_exchange_status(self.h))

    def close(self):
        if self.h is not None:
            self.lib.mbd_exchange_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exchange_rewards(local, world: int, group=None):
    """The ONE exchange step of a diffusion step: every rank contributes the per-candidate values of its
    shard (``local`` [rows, shard]) and receives all N of them, rank-major = candidate order, as
    [rows, N].  One all-gather (RCCL over xGMI on GPUs; gloo in the CPU tests and in the two-ranks-on-one-GPU
    dry runs, where device tensors are staged through the host)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    rows, sh = local.shape
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty((world * rows, sh), dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.detach().cpu().contiguous(), group=group)
        gathered = host.to(local.device)
    else:
        gathered = torch.empty((world * rows, sh), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)  # concatenation along dim 0
    return gathered.view(world, rows, sh).permute(1, 0, 2).reshape(rows, world * sh).contiguous()


def reverse_distributed(plan: Plan, key, device, group=None, sync_every_step: bool = False, progress=None,
                        phase_times: dict = None, collective: str = None):
    """reverse() (mbd_planner.py:138-148) with the candidates sharded over the ranks of ``group``.
    One all-gather of the per-candidate mean rewards per diffusion step (plus the demo log-densities
    when enabled, packed in the same buffer). Returns (mu_0ts, rew_means) as CUDA tensors.

    ``sync_every_step`` / ``progress``: the reference formats the step's mean reward for its progress bar every
    step (:147), which is a device->host read per step; ``progress(i, rew)`` receives that value.
    ``phase_times``: a dict that receives HIP-event milliseconds per step of phase 1 (sample + rollout), the
    exchange and phase 2 (score + weighted mean) — each phase is then fenced, for measurement only.
    ``collective``: "torch" (all_gather_into_tensor: RCCL over xGMI, or gloo) or "p2p" (the in-library exchange:
    peer writes into every rank's window, no collective library on the step's path); default: $MBD_COLLECTIVE or
    "torch".  Same values either way."""
    import torch
    import torch.distributed as dist

    N, sh = plan.cfg.Nsample, plan.cfg.shard_count
    # the exchange follows the PLAN's shard layout, not whatever process group happens to be initialised: an unsharded
    # plan (force_single under torchrun) must not all-gather — it would score rank 0's rewards on every rank
    world = N // sh
    if sh * world != N:
        raise ValueError(f"shard_count={sh} does not divide Nsample={N}")
    if world > 1:
        group_world = dist.get_world_size(group) if dist.is_initialized() else 1
        if group_world != world:
            raise ValueError(f"the plan is sharded over {world} ranks but the process group has {group_world}")
    demo = bool(plan.cfg.enable_demo)
    rows = 2 if demo else 1
    HNu = plan.H * plan.Nu
    dev = torch.device("cuda", device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    mu = torch.zeros((plan.Nd - 1, HNu), dtype=torch.float32, device=dev)
    rew_means = torch.zeros(plan.Nd - 1, dtype=torch.float32, device=dev)
    Ybar = torch.zeros(HNu, dtype=torch.float32, device=dev)  # YN = zeros (mbd_planner.py:95)
    local = torch.zeros((rows, sh), dtype=torch.float32, device=dev)
    rng = np.asarray(key, np.uint32)
    impl = plan.cfg.prng_impl
    lib = plan.lib
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if phase_times is not None else None
    acc = [0.0, 0.0, 0.0]
    collective = collective or os.environ.get("MBD_COLLECTIVE", "torch")
    p2p = None
    if world > 1 and collective == "p2p":
        # the constructor agrees on a failure across the ranks (all raise, or none): a runtime without fine-grained device
        # memory or peer mappings (mbd_exchange_create: MBD_ERR_UNSUPPORTED) falls back — on EVERY rank — to the collective
        # library's all-gather, as include/mbd_hip.h tells the caller to; same values either way
        try:
            p2p = P2PExchange(device, rows, sh, group)
        except _capi.MbdError as e:
            import warnings
            warnings.warn(f"in-library exchange unavailable ({e}): falling back to torch.distributed all-gather")
            p2p = None
    host = HostProgress(plan.Nd - 1, dev) if (sync_every_step or progress is not None) else None
    # Host work that does not depend on the GPU — the key chain (rng, Y0s_rng = split(rng), mbd_planner.py:103) and the
    # declaration of the FOLLOWING step's key (its normals are generated beside this step's rollout) — is done while
    # the device runs the previous step, before the host waits for that step's mean reward: the wait is followed by
    # the rollout launch and nothing else.
    keys = _capi.prng_split(rng, 2, impl)
    rng, ks = keys[0], _capi.key_array(keys[1])
    keys = _capi.prng_split(rng, 2, impl)
    if plan.Nd - 1 > 1:
        _capi.check(lib.mbd_plan_prefetch_noise(plan.h, _capi.key_array(keys[1]), stream))
    p_loc0, p_loc1 = local[0].data_ptr(), (local[1].data_ptr() if demo else None)
    for i in range(plan.Nd - 1, 0, -1):
        if ev:
            ev[0].record()
        _capi.check(lib.mbd_plan_sample_rollout(plan.h, i, ks, Ybar.data_ptr(), p_loc0, p_loc1, stream))
        if ev:
            ev[1].record()
        if p2p is not None:
            base = p2p.all_gather(local, stream)
            p_all0, p_all1 = base, (base + 4 * N if demo else None)
        else:
            allv = exchange_rewards(local, world, group)
            p_all0, p_all1 = allv[0].data_ptr(), (allv[1].data_ptr() if demo else None)
        if ev:
            ev[2].record()
        out = mu[plan.Nd - 1 - i]
        k = plan.Nd - 1 - i
        _capi.check(lib.mbd_plan_score_update(plan.h, i, ks, Ybar.data_ptr(), p_all0, p_all1, out.data_ptr(),
                                              host.ptr(k) if host else rew_means[k:].data_ptr(), stream))
        Ybar = out
        if ev:
            ev[3].record()
            ev[3].synchronize()
            for j in range(3):
                acc[j] += ev[j].elapsed_time(ev[j + 1])
        if i > 1:  # the next step's keys, and the declaration of the one after it
            rng, ks = keys[0], _capi.key_array(keys[1])
            keys = _capi.prng_split(rng, 2, impl)
            if i > 2:
                _capi.check(lib.mbd_plan_prefetch_noise(plan.h, _capi.key_array(keys[1]), stream))
        if host is not None:  # the reference formats the reward every step (:147): one host read per step
            r = host.wait(k)
            if progress is not None:
                progress(i, r)
    if phase_times is not None:
        n = max(plan.Nd - 1, 1)
        phase_times.update(phase1_ms=acc[0] / n, exchange_ms=acc[1] / n, phase2_ms=acc[2] / n, steps=plan.Nd - 1)
    if host is not None:
        torch.cuda.synchronize(dev)
        rew_means.copy_(host.t[: plan.Nd - 1])
    if p2p is not None:
        p2p.status()  # (raises when a wait ran into its time limit: a peer never arrived)
        p2p.close()
    return mu.view(plan.Nd - 1, plan.H, plan.Nu), rew_means


def run_diffusion(args: Args, device: int = None, return_details: bool = False, progress=None,
                  force_single: bool = False, measure_phases: bool = False, collective: str = None):
    """mbd_planner.py:38-182. Returns rew_final (float); ``return_details`` adds a dict with mu_0ts,
    per-step mean rewards and the reverse-loop wall time.  ``progress(i, rew)`` is called after every diffusion
    step with the step's mean reward, like the reference's progress bar (:147) — one device->host read per step;
    without it the loop runs asynchronously and the means are read once at the end.  ``force_single`` ignores an
    initialised process group (every rank then runs the whole plan)."""
    import torch
    import torch.distributed as dist

    distributed = (not force_single) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if distributed else 0
    rng = _capi.prng_key(args.seed)  # :40
    apply_recommended(args)
    env = get_env(args.env_name, device=device)  # :70
    impl = prng_impl()
    rng, rng_reset = _capi.prng_split(rng, 2, impl)  # :79  NOTE: rng_reset should never be changed.
    state_init = env.reset(rng_reset)  # :80
    rng_exp, rng = _capi.prng_split(rng, 2, impl)  # :150

    if distributed:
        begin, sh = shard_bounds(args.Nsample, dist.get_world_size(), dist.get_rank())
        plan = Plan(env, args, shard_begin=begin, shard_count=sh)
    else:
        plan = Plan(env, args)
    plan.set_state0(state_init)
    _, _, sigmas = plan.schedule()
    print(f"init sigma = {sigmas[-1]:.2e}")  # :93

    phases = {} if measure_phases else None
    if distributed or progress is not None or measure_phases:
        torch.cuda.set_device(device)
        torch.cuda.synchronize(device)
        t0 = time.time()
        mu_t, rm_t = reverse_distributed(plan, rng_exp, device, progress=progress, phase_times=phases,
                                         collective=collective)
        torch.cuda.synchronize(device)
        secs = time.time() - t0
        mu, rew_means = mu_t.cpu().numpy(), rm_t.cpu().numpy()
        rew_final = plan.eval(mu[-1])  # :179-180
    else:
        mu, rew_means, rew_final, secs = plan.run(rng_exp)

    if not args.not_render and (not distributed or dist.get_rank() == 0):  # :152-156 (mu_0ts.npy only)
        path = os.path.join(os.getcwd(), "results", args.env_name)
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "mu_0ts.npy"), mu)
        # stand-in for rollout.html / rollout.png (:157-178): the replayed final plan as arrays
        from ..utils import rollout_states
        np.savez_compressed(os.path.join(path, "rollout_states.npz"), **rollout_states(env, state_init, mu[-1]))
    plan.close()
    if return_details:
        return rew_final, dict(mu_0ts=mu, rew_means=rew_means, loop_seconds=secs, state_init=state_init,
                               steps_per_sec=(args.Ndiffuse - 1) / secs, sharded=bool(distributed),
                               world=dist.get_world_size() if distributed else 1, phase_ms=phases)
    return rew_final


if __name__ == "__main__":
    import argparse

    p = argparse.ArgumentParser()
    for f in Args.__dataclass_fields__.values():
        if f.type in ("bool", bool):
            p.add_argument(f"--{f.name}", action="store_true")
        else:
            p.add_argument(f"--{f.name}", type=type(f.default), default=f.default)
    ns = p.parse_args()
    rew_final = run_diffusion(Args(**vars(ns)), progress=lambda i, rew: print(f"\rDiffusing i={i:4d} rew {rew:.2e}",
                                                                                 end="", flush=True))  # :141-147
    print(f"\nfinal reward = {rew_final:.2e}")  # :187
