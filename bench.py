#!/usr/bin/env python3
"""bench.py — diffusion-steps/sec of the reverse-diffusion hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one reverse-diffusion step (mbd_planner.py:97-135): sample N candidates, roll them out
H=50 control steps through the positional rigid-body simulator, score, softmax, weighted mean.
Workload at one GPU = the configuration the metric is quoted on: humanoidrun, N=1024, H=50,
Ndiffuse=100, temp 0.1, seed 0, disable_recommended_params.  With G GPUs the
candidates are sharded: 1024 per GPU (weak scaling, N_total = 1024*G), one all-gather of the N mean
rewards per step.  `value` is the whole-job rate in units of 1024-candidate diffusion steps per second
(= plain diffusion-steps/sec at G=1); the un-normalised loop rate is reported as `steps_per_sec`.
Inputs are resident in HBM when the timed region starts; data is synthetic (seeded PRNG).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_PER_GPU = int(os.environ.get("MBD_BENCH_N", "1024"))  # 1024 = the metric config; other values: experiments only
H, ND, TEMP, ENV = 50, 100, 0.1, "humanoidrun"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def b_alg_bytes(N, Hh, Nu, demo=False):
    """ALGORITHMIC bytes of one diffusion step: write+read Y0s, write+read rews,
    read Ybar_i, write Ybar_{i-1}."""
    return 4 * (2 * N * Hh * Nu + 2 * N + 2 * Hh * Nu) + (8 * N if demo else 0)


def pmc_traffic():
    """HBM bytes per rollout-kernel launch from the newest committed PMC pass (profiles/rNN_pmc.json,
    FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950); None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    for k, v in d.items():
        if "rollout_kernel<16" in k and "hbm_bytes_per_launch_corrected" in v:
            return v["hbm_bytes_per_launch_corrected"], os.path.basename(files[-1])
    return None, None


def valu_view(kern_ms, n_waves, substeps):
    """FP32 VALU view of the same kernel (the limit that actually binds): static flops per lane per
    substep from tools/count_flops.py (profiles/rNN_static_flops.json) x lanes x substeps / duration,
    against the 157.3 TFLOP/s vector-FP32 peak of MI355X_MICROARCH.md."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_static_flops.json")))
    if not files or kern_ms <= 0:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    flops = d["fp32_flops_per_lane_substep"] * 64.0 * n_waves * substeps
    tf = flops / (kern_ms * 1e-3) / 1e12
    return {"bound": "fp32-valu-issue", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
            "flops_per_launch": flops, "valu_instr_per_wave_substep": d["valu_per_substep"],
            "waves": n_waves, "simds_occupied_frac": min(1.0, n_waves / 1024.0),
            "source": os.path.basename(files[-1])}


def cpu_baseline(seconds_budget=15.0):
    """The oracle (a port, NOT the JAX reference — jax/brax are absent) timed on the host cores on a
    bounded sample of the same workload: consecutive reverse-diffusion steps at N=1024, H=50."""
    import numpy as np
    from oracle import oracle as orc_mod
    from oracle import planner as op
    from mbd_hip.model import Model
    orc_mod.build()
    orc = orc_mod.Oracle("f32_omp")
    with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{ENV}.json")) as f:
        m = Model.from_json(f.read())
    env = op.OracleEnv(orc, ENV, m.to_struct(), init_q=m.init_q)
    key = orc.prng_key(0)
    rng, rng_reset = orc.split(key, 2, 1)
    state0 = env.reset(rng_reset, 1)
    sched = orc.schedule(1e-4, 1e-2, ND)
    Ybar = np.zeros((H, env.Nu), np.float32)
    r = orc.split(rng, 2, 1)[0]
    steps, t0 = 0, time.time()
    i = ND - 1
    while True:
        r, Ybar, _, _ = op.reverse_once(orc, env, state0, i, r, Ybar, sched, N_PER_GPU, H, TEMP, 1)
        steps += 1
        i -= 1
        if time.time() - t0 > seconds_budget or i < 1:  # 10-15 s of host work, at most one whole plan (99 steps)
            break
    dt = time.time() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": steps / dt, "unit": "diffusion-steps/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} consecutive reverse-diffusion steps of {ENV} N={N_PER_GPU} H={H} "
                      f"(CPU oracle, OpenMP over candidates, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-final-reward", action="store_true")
    args = ap.parse_args()

    # before the HIP runtime initialises: the host driver only supports dmabuf IPC (RCCL peer mappings)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    __graft_entry__.build()  # serialised across ranks by a file lock; a no-op when the library is fresh
    # MBD_FORCE_DIST=1: take the multi-rank code path (RCCL init, all-gather, barrier, max-reduce) with ONE rank —
    # what the single-GPU test box can exercise of it (tests/test_gpu_parity.py)
    distributed = world > 1 or os.environ.get("MBD_FORCE_DIST") == "1"
    backend = os.environ.get("MBD_DIST_BACKEND", "nccl")  # "gloo": 2-rank dry runs on a single-GPU box
    n_dev = torch.cuda.device_count()
    if distributed:
        if backend == "nccl":
            assert n_dev >= world, f"{world} ranks need {world} GPUs, found {n_dev}"
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(n_dev, 1)
            dist.init_process_group(backend)
        dist.barrier()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run"

    from mbd_hip import _capi
    from mbd_hip.envs import get_env
    from mbd_hip.planners.mbd_planner import Args, Plan, run_diffusion
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    N_total = N_PER_GPU * world
    pargs = Args(seed=0, env_name=ENV, Nsample=N_total, Hsample=H, Ndiffuse=ND, temp_sample=TEMP,
                 disable_recommended_params=True, not_render=True)
    env = get_env(ENV, device=local_rank)
    key = _capi.prng_key(pargs.seed)
    rng, rng_reset = _capi.prng_split(key, 2)
    state_init = env.reset(rng_reset)
    rng_exp, _ = _capi.prng_split(rng, 2)
    plan = Plan(env, pargs, shard_begin=rank * N_PER_GPU, shard_count=N_PER_GPU)
    plan.set_state0(state_init)
    Nu, HNu = env.action_size, H * env.action_size
    lib, stream = plan.lib, torch.cuda.current_stream(dev).cuda_stream

    Ybar = torch.zeros(HNu, dtype=torch.float32, device=dev)
    Ynext = torch.zeros(HNu, dtype=torch.float32, device=dev)
    local = torch.zeros(N_PER_GPU, dtype=torch.float32, device=dev)
    allv = torch.zeros(N_total, dtype=torch.float32, device=dev)
    rew_mean = torch.zeros(1, dtype=torch.float32, device=dev)
    state = {"rng": np.asarray(rng_exp, np.uint32), "i": ND - 1}

    def step():
        nonlocal Ybar, Ynext
        if state["i"] < 1:  # start the next plan: YN = zeros, fresh schedule position
            state["i"] = ND - 1
            Ybar.zero_()
        keys = _capi.prng_split(state["rng"], 2)
        state["rng"], ks = keys[0], _capi.key_array(keys[1])
        i = state["i"]
        _capi.check(lib.mbd_plan_sample_rollout(plan.h, i, ks, Ybar.data_ptr(), local.data_ptr(), None, stream))
        if distributed and backend == "nccl":
            dist.all_gather_into_tensor(allv, local)  # the ONE collective of a diffusion step (RCCL/xGMI)
            src = allv
        elif distributed:  # gloo dry run: stage through the host
            host = torch.empty(N_total, dtype=torch.float32)
            dist.all_gather_into_tensor(host, local.cpu())
            allv.copy_(host)
            src = allv
        else:
            src = local
        _capi.check(lib.mbd_plan_score_update(plan.h, i, ks, Ybar.data_ptr(), src.data_ptr(), None,
                                              Ynext.data_ptr(), rew_mean.data_ptr(), stream))
        Ybar, Ynext = Ynext, Ybar
        state["i"] = i - 1

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    plan.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    plan.enable_timing(False)
    kern_ms, kern_n = plan.kernel_time()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    final = None
    if rank == 0 and not args.no_final_reward and not distributed:
        # the metric's second half: final reward of complete plans, seeds 0..7 as mbd/scripts/run_mbd.py:20
        rews = []
        for seed in range(8):
            a = Args(seed=seed, env_name=ENV, Nsample=N_PER_GPU, Hsample=H, Ndiffuse=ND, temp_sample=TEMP,
                     disable_recommended_params=True, not_render=True)
            with contextlib.redirect_stdout(sys.stderr):  # the reference prints "init sigma = ..." (:92); stdout
                rews.append(float(run_diffusion(a, device=local_rank)))  # carries the one JSON line only
        final = {"seeds": list(range(8)), "rew_final": rews, "mean": float(np.mean(rews)),
                 "std": float(np.std(rews))}
    plan.close()

    if rank == 0:
        steps_per_sec = args.steps / elapsed
        value = steps_per_sec * (N_total / N_PER_GPU)
        balg = b_alg_bytes(N_PER_GPU, H, Nu)
        achieved = (balg / 1e9) / (kern_ms / 1e3) if kern_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic()
        out = {
            "metric": "diffusion-steps/sec (N=1024, H=50) + final reward, humanoidrun, 1/2/4/8 GPU",
            "value": value, "unit": "diffusion-steps/sec (1024-candidate steps, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "steps_per_sec": steps_per_sec,
            "config": {"workload": f"{ENV} N={N_PER_GPU}/GPU (N_total={N_total}) H={H} Ndiffuse={ND} "
                                   f"temp={TEMP} seed=0 disable_recommended_params",
                       "N_total": N_total, "N_per_gpu": N_PER_GPU, "H": H, "Nu": Nu,
                       "collective": "all_gather(rews) per step" if distributed else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "rollout_kernel<16,iso,noslide,3,1,dpp(1,-4,-6)>", "kernel_avg_ms": kern_ms,
                         "kernel_launches": kern_n, "algorithmic_bytes_per_launch": balg,
                         "note": "state stays in VGPRs for all H*n_frames substeps: the kernel is bound by "
                                 "dependent FP32 VALU issue, not HBM (DESIGN.md §Roofline)"},
            "valu": valu_view(kern_ms, N_PER_GPU // 4, H * 7),
            "final_reward": final,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
