#!/usr/bin/env python3
"""bench.py — diffusion-steps/sec of the reverse-diffusion hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--scaling strong|weak]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one reverse-diffusion step (mbd_planner.py:97-135): sample N candidates, roll them out
H control steps through the positional rigid-body simulator, score, softmax, weighted mean — PLUS the
host read of the step's mean reward that the reference's progress bar forces every step
(mbd_planner.py:147; SURVEY.md §8(d) defines the metric with that read inside): the host does not dispatch
step k+1 before it holds step k's mean reward.  `value` is measured with the per-step read; `value_async`
(same K steps, reads dropped) is reported beside it.

--config (default `metric`, the configuration BASELINE.json's metric is quoted on):
   metric            humanoidrun  N=1024 H=50 temp 0.1
   hopper512         hopper       N=512  H=50 temp 0.1   (BASELINE config 2)
   halfcheetah1024   halfcheetah  N=1024 H=50 temp 0.4   (config 3)
   humanoidrun4096   humanoidrun  N=4096 H=50 temp 0.1   (config 4)
   humanoidtrack2048demo  humanoidtrack N=2048 H=50 temp 0.1 enable_demo (config 5)
   car2d             car2d        N=128  H=30 Ndiffuse=50 (config 1)
   humanoidrun8192   humanoidrun  N=8192 H=50 temp 0.1   (the reference's own default N for humanoidrun,
                                                          mbd_planner.py:54-60; the two-candidates-per-lane kernel)
   sweep8            the reference's 8-seed sweep (mbd/scripts/run_mbd.py:17-39) at the metric's sizes: 8 plans x
                     N=1024 in lockstep (mbd_sweep_run); a "step" is one diffusion step of all 8 plans and `value`
                     counts plan-steps/sec.  With G > 1 GPUs the plans are REPLICAS over the ranks (seeds r*8/G ...
                     per rank, no communication inside a run, rewards gathered once at the end): strong = the
                     reference's 8 plans over G GPUs, weak = 8 plans per GPU
Ndiffuse=100, seed 0, disable_recommended_params everywhere.

--scaling with G > 1 GPUs (candidates are sharded over ranks, ONE all-gather of the N mean rewards per step):
   strong (default)  N_total = the config's N, N/G candidates per GPU — the literal metric ("N=1024 at 1/2/4/8
                     GPUs").  A rollout is latency-bound (time ~ instructions per wavefront, not wavefronts), so
                     this curve is expected to be flat; it is reported as measured.
   weak              N candidates PER GPU, N_total = N*G; `value` is then normalised to N-candidate steps.
Both are measured in every multi-GPU run; the one not selected is reported under "other_scaling".
Inputs are resident in HBM when the timed region starts; data is synthetic (seeded PRNG).

The timed region is EXACTLY K steps between two fences (barrier + device synchronisation); it is repeated --repeats
times (default 5) inside the one command and `value` is the MEDIAN block (`value_min` / `value_max` beside it): one
block of 20 steps is 11 ms, and box-to-box noise is larger than any round's gain.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TF = 157.3   # MI355X_MICROARCH.md: vector FP32 peak

CONFIGS = {
    # name: env, N, H, Ndiffuse, temp, demo, lanes per candidate (rollout kernel), kernel label
    "metric": dict(env="humanoidrun", N=1024, H=50, Nd=100, temp=0.1, demo=False, lps=16,
                   kernel="rollout_kernel<16,iso,noslide,3,1,dpp(1,-4,-6)>"),
    # (round 6: ONE candidate per wavefront — 512 wavefronts — and a wave-uniform early-out around the contact code)
    "hopper512": dict(env="hopper", N=512, H=50, Nd=100, temp=0.1, demo=False, lps=4, cpw=1, static="hopper_planar_eo",
                      kernel="rollout_planar_kernel<4,2 colliders,dpp(1),early-out> (one candidate per wavefront)"),
    "halfcheetah1024": dict(env="halfcheetah", N=1024, H=50, Nd=100, temp=0.4, demo=False, lps=8,
                            static="halfcheetah_planar", kernel="rollout_planar_kernel<8,2 colliders,dpp(1,-3)>"),
    "humanoidrun4096": dict(env="humanoidrun", N=4096, H=50, Nd=100, temp=0.1, demo=False, lps=16,
                            kernel="rollout_kernel<16,iso,noslide,3,1,dpp(1,-4,-6)>"),
    "humanoidtrack2048demo": dict(env="humanoidtrack", N=2048, H=50, Nd=100, temp=0.1, demo=True, lps=16,
                                  kernel="rollout_kernel<16,iso,noslide,3,1,dpp(1,-4,-6)>"),
    "car2d": dict(env="car2d", N=128, H=30, Nd=50, temp=0.1, demo=False, lps=1, kernel="car2d_rollout_kernel"),
    "humanoidrun8192": dict(env="humanoidrun", N=8192, H=50, Nd=100, temp=0.1, demo=False, lps=16, cpw=8,
                            static="humanoidrun_pk2", kernel="rollout_pk2_kernel<1,humanoidrun,7> (two candidates per lane)"),
    "sweep8": dict(env="humanoidrun", N=1024, H=50, Nd=100, temp=0.1, demo=False, lps=16, cpw=8, plans=8,
                   static="humanoidrun_pk2", kernel="rollout_pk2_kernel<1,humanoidrun,7> over 8 plans x 1024 candidates"),
}


def b_alg_bytes(N, Hh, Nu, demo=False):
    """ALGORITHMIC bytes of one diffusion step: write+read Y0s, write+read rews,
    read Ybar_i, write Ybar_{i-1} (+ the demo log-densities)."""
    return 4 * (2 * N * Hh * Nu + 2 * N + 2 * Hh * Nu) + (8 * N if demo else 0)


def committed(name):
    """A committed evidence file of the newest round under profiles/ (None when absent)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{name}")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def pmc_traffic(config):
    """HBM bytes per rollout-kernel launch from the newest committed PMC pass (profiles/rNN_pmc.json, one entry per
    bench config; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950); None when absent."""
    d, src = committed("pmc.json")
    if not d:
        return None, None
    ent = d.get(config) or (d if config == "metric" else {})
    for k, v in ent.items():
        if isinstance(v, dict) and "rollout_" in k and "_kernel" in k and "car2d" not in k and "hbm_bytes_per_launch_corrected" in v:
            return v["hbm_bytes_per_launch_corrected"], src
    return None, None


def op_counts(env_name):
    """F_sub(model): flops of one substep from the op counter compiled into the CPU restatement
    (oracle/count_ops.cc), as committed by tools/count_ops.py under profiles/."""
    d, src = committed("op_counts.json")
    if d and env_name in d:
        return d[env_name], src
    return None, None


def valu_view(cfg, n_local, kern_ms, n_frames, fsub, fsub_src):
    """FP32 VALU view of the rollout kernel (the limit that binds): ALGORITHMIC flops = candidates x H x n_frames x
    F_sub (op counter, real links and contacts only) against the 157.3 TFLOP/s vector peak; `issued` = the same
    with the flops the emitted ISA executes per lane (tools/count_flops.py: padding lanes and masked slots
    included)."""
    if kern_ms <= 0 or cfg["lps"] <= 1:
        return None
    cpw = cfg.get("cpw", 64 // cfg["lps"])  # candidates per wavefront (8: two per lane)
    waves = (n_local + cpw - 1) // cpw
    substeps = cfg["H"] * n_frames
    out = {"bound": "fp32-valu-issue", "peak": VALU_PEAK_TF, "unit": "TFLOP/s", "waves": waves,
           "simds_occupied_frac": min(1.0, waves / 1024.0)}
    if fsub:
        flops = float(fsub["flops"]) * n_local * substeps
        tf = flops / (kern_ms * 1e-3) / 1e12
        out.update(achieved=tf, frac=tf / VALU_PEAK_TF, algorithmic_frac=tf / VALU_PEAK_TF, flops_per_launch=flops,
                   flops_per_substep=fsub["flops"], op_counts=fsub, source=fsub_src)
    st, src = committed("static_flops.json")
    ent = (st or {}).get(cfg.get("static", cfg["env"]))
    if ent:
        fl = ent["fp32_flops_per_lane_substep"] * 64.0 * waves * substeps
        out["issued"] = {"flops_per_launch": fl, "achieved": fl / (kern_ms * 1e-3) / 1e12,
                         "frac": fl / (kern_ms * 1e-3) / 1e12 / VALU_PEAK_TF,
                         "valu_instr_per_wave_substep": ent["valu_per_substep"],
                         "instr_per_wave_substep": ent["instructions_per_substep"], "source": src}
        if "contact_path_instructions_per_substep" in ent:
            # an early-out instantiation: the static count above is the substep WITHOUT a contact (the loop's fall-through
            # path); one with a contact runs the out-of-line block instead.  What was EXECUTED comes from the PMC pass of the
            # same config (SQ_INSTS_VALU / rollout wavefronts / substeps; the launch's noise workgroups add < 0.5 %)
            eo = {"no_contact_path_instr_per_substep": ent["no_contact_path_instructions_per_substep"],
                  "contact_path_instr_per_substep": ent["contact_path_instructions_per_substep"]}
            pm, pm_src = committed("pmc.json")
            for k, v in ((pm or {}).get(cfg.get("name", ""), {}) or {}).items():
                if isinstance(v, dict) and "rollout_" in k and "SQ_INSTS_VALU" in v:
                    ex = v["SQ_INSTS_VALU"]["avg_per_launch"] / waves / substeps
                    eo["executed_valu_per_wave_substep"] = ex
                    lo, hi = ent["valu_per_substep"], ent["valu_per_substep"] + (eo["contact_path_instr_per_substep"] - eo["no_contact_path_instr_per_substep"])
                    eo["substeps_without_contact_frac"] = max(0.0, min(1.0, (hi - ex) / max(hi - lo, 1e-9)))
                    eo["source"] = pm_src
            out["issued"]["early_out"] = eo
    # the latency floor, measured (tools/critical_path.py: the op counter tracks the depth of every value's chain): the
    # longest chain of DEPENDENT operations of one substep of one candidate.  x 4 clocks per dependent issue x the
    # launch's substeps = the time of this rollout on a machine with unlimited lanes per candidate; the kernel issues
    # `instr_per_wave_substep` instructions where `critical_path_ops` are on the chain — the rest is parallelism that
    # lives in vector components and (parent, child) pairs, harvested two-wide by v_pk_* (docs/experiments.md §5:
    # the component-per-lane probe measured 1.09x, the two-wavefront pipeline 1.10x at best)
    cp, cp_src = committed("critical_path.json")
    cpe = (cp or {}).get(cfg["env"])
    if cpe:
        out["critical_path_ops"] = cpe["depth_per_substep"]
        out["critical_path"] = {"ops_per_substep": cpe["depth_per_substep"], "unit": (cp or {}).get("unit"),
                                "floor_ms_at_4clk_2p4GHz": cpe["depth_per_substep"] * 4.0 * substeps / 2.4e9 * 1e3,
                                "kernel_over_floor": kern_ms / (cpe["depth_per_substep"] * 4.0 * substeps / 2.4e9 * 1e3),
                                "ops_per_link_over_depth": cpe["ops_per_link_over_depth"], "source": cp_src}
    return out


def reference_baseline(cfg, seconds_budget=25.0):
    """the REAL reference timed on this box's host cores — tried first, every run.  Needs jax
    and brax importable and a checkout of the reference ($MBD_REFERENCE_PATH; there is none on the GPU box of this
    pool, and jax / brax are not in the image: the import fails and the caller falls back to the port).  What is
    timed: mbd.planners.mbd_planner.run_diffusion (mbd_planner.py:38-182) on JAX's CPU backend at the config's sizes,
    twice, with Ndiffuse = 2 and Ndiffuse = 2 + K — the difference is K reverse-diffusion steps without the tracing /
    compilation both runs pay.  Returns (record, None) or (None, reason)."""
    ref = os.environ.get("MBD_REFERENCE_PATH", "/root/reference")
    try:
        os.environ.setdefault("JAX_PLATFORMS", "cpu")
        import jax
        import brax
        if not os.path.isdir(os.path.join(ref, "mbd")):
            raise ImportError(f"no reference checkout at {ref}")
        if ref not in sys.path:
            sys.path.insert(0, ref)
        from mbd.planners import mbd_planner as ref_planner
    except Exception as e:  # noqa: BLE001 — ImportError, or whatever a half-installed jax raises
        return None, f"{type(e).__name__}: {e}"

    last = {}

    def run(nd):
        a = ref_planner.Args(seed=0, disable_recommended_params=True, not_render=True, env_name=cfg["env"],
                             Nsample=cfg["N"], Hsample=cfg["H"], Ndiffuse=nd, temp_sample=cfg["temp"],
                             enable_demo=cfg["demo"])
        t0 = time.time()
        with contextlib.redirect_stdout(sys.stderr):
            rf = ref_planner.run_diffusion(a)
        last["ndiffuse"], last["rew_final"] = nd, rf
        return time.time() - t0

    try:
        t_short = run(2)                      # 1 step + tracing / compilation / reset / final rollout
        K = 2
        t_long = run(2 + K)
        per = max((t_long - t_short) / K, 1e-9)
        while t_long < seconds_budget / 2 and K < 64:  # a longer sample while it stays inside the budget
            K *= 2
            t_long = run(2 + K)
            per = max((t_long - t_short) / K, 1e-9)
    except Exception as e:  # noqa: BLE001
        return None, f"reference run failed: {type(e).__name__}: {e}"
    versions = {"jax": getattr(jax, "__version__", "?"), "brax": getattr(brax, "__version__", "?")}
    rec = {"value": 1.0 / per, "unit": "diffusion-steps/sec", "cores": os.cpu_count() or 1, "kind": "jax-reference",
           "versions": versions,
           "sample": f"the reference's run_diffusion on JAX's CPU backend, {cfg['env']} N={cfg['N']} H={cfg['H']}: "
                     f"(Ndiffuse={2 + K}: {t_long:.1f} s) - (Ndiffuse=2: {t_short:.1f} s) over {K} steps"}
    # FIRST CONTACT with a real Brax, automatic (round 6): the box that can import jax + brax also dumps the reference's
    # records for this line's env and holds the checker to them — no human in the loop
    rec["parity_jax"] = parity_vs_jax(ref, cfg, versions, last)
    return rec, None


def parity_vs_jax(ref, cfg, versions, last_run):
    """The `parity_jax` object of the line: what tools/dump_golden.py records of the REAL reference (jax + brax importable,
    $MBD_REFERENCE_PATH) for this line's env — three reverse_once steps at N = min(N, 64), one physics substep stage by stage
    from the reset pose and from a settled pose, the bounce — against this repo's checker (tools/compare_golden.py):
      first_mismatch_stage   the first stage of Brax's substep the DEFAULT specification misses by more than 1e-5 (null: none),
      fitted_flags / fitted  the word of DESIGN.md §9's switches that fits best (--search) and its names: a word other than this
                             build's mbd_tuned_spec() (4 = contact_avg as shipped) is answered by the variant library of that word (libmbd_hip_sum.so = 0),
      max_rel                teacher-forced: the checker's rewards on the reference's candidates of those steps vs Brax's
                             (the north star's 1e-5), per quantity in max_rel_rewards / max_rel_ybar,
      rew_final_ref          what the reference's last timed run_diffusion returned (seed 0, its Ndiffuse beside it; main() adds
                             this library's rew_final of the same arguments).
    Outside the timed region; the GPU side equals the checker bit for bit (`parity`), so checker-vs-Brax is the open half.
    Every failure is reported inside the object — it never costs the line its value."""
    import tempfile
    out = {"versions": versions, "env": cfg["env"], "tolerance": 1e-5}
    try:
        out["rew_final_ref"], out["rew_final_ndiffuse"] = float(last_run.get("rew_final")), int(last_run.get("ndiffuse"))
    except Exception:  # noqa: BLE001
        pass
    try:
        import numpy as np
        tools = os.path.join(ROOT, "tools")
        if tools not in sys.path:
            sys.path.insert(0, tools)
        import compare_golden
        import dump_golden
        td = tempfile.mkdtemp(prefix="mbd_parity_jax_")
        with contextlib.redirect_stdout(sys.stderr):
            path = dump_golden.dump(ref, cfg["env"], min(int(cfg["N"]), 64), int(cfg["H"]), 3, out_dir=td)
        out["golden"] = os.path.basename(path)
        g = np.load(path)
        out["records"] = {"reverse_once_steps": len([k for k in g.files if k.startswith("rewss_")]),
                          "substep_stages": bool("stage_1_acceleration_x_pos" in g.files), "settled_substep": bool("contact_substep_in_x_pos" in g.files),
                          "bounce": bool("bounce_vz" in g.files)}
        if "substep_in_x_pos" in g.files:
            with contextlib.redirect_stdout(sys.stderr):
                lines, first = compare_golden.compare(path, 1e-5)
                rows = compare_golden.search(path, 1e-5)
            out["first_mismatch_stage"] = None if first is None else first[0]
            if first is not None:
                out["first_mismatch"] = {"stage": first[0], "link": int(first[1]), "quantity": first[2], "err": float(first[3])}
            best = rows[0]
            out["fitted_flags"], out["fitted"] = int(best[0]), list(best[1]) or ["none"]
            out["fitted_first_mismatch_stage"] = None if best[2] is None else best[2][0]
            out["report"] = lines[-12:]
        # record (A), teacher-forced through the checker (the cpu_baseline leg may use it)
        from mbd_hip.model import Model
        from oracle import oracle as orc_mod
        orc_mod.build()
        orc = orc_mod.Oracle("f32")
        with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{cfg['env']}.json")) as f:
            m = Model.from_json(f.read())   # (as compiled: the default word)
        if "fitted_flags" in out and out.get("fitted_first_mismatch_stage", 1) is None:
            m = m.with_spec(out["fitted_flags"])
        ms = m.to_struct()
        st = orc.forward(ms, np.asarray(g["q0"], np.float32), np.asarray(g["qd0"], np.float32))
        er, ey = 0.0, 0.0
        for k in range(out["records"]["reverse_once_steps"]):
            rew = orc.rollout(ms, st, np.asarray(g[f"Y0s_{k}"], np.float32))
            ref_r = np.asarray(g[f"rewss_{k}"], np.float32)
            er = max(er, float((np.abs(rew - ref_r) / np.maximum(np.abs(ref_r), 1.0)).max()))
            w = np.asarray(g[f"weights_{k}"], np.float64)
            yb = np.einsum("n,nij->ij", w, np.asarray(g[f"Y0s_{k}"], np.float64))
            ref_y = np.asarray(g[f"Ybar_{k}"], np.float64)
            ey = max(ey, float((np.abs(yb - ref_y) / np.maximum(np.abs(ref_y), 1.0)).max()))
        out["max_rel_rewards"], out["max_rel_ybar"], out["max_rel"] = er, ey, max(er, ey)
        out["within_tolerance"] = bool(max(er, ey) <= 1e-5)
        out["flags_of_the_teacher_forced_model"] = int(m.fields["flags"]) & 252
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def cpu_baseline(cfg, seconds_budget=15.0):
    """The CPU side of the comparison, on the GPU box's host cores, rank 0 at N=1: the real reference when it can be
    imported (reference_baseline: kind "jax-reference"), otherwise the oracle (kind "port": a restatement, NOT the JAX
    reference) on a bounded sample of the same workload — consecutive reverse-diffusion steps of the config.  Also
    returns the op counter's F_sub for the config's model (the oracle may only be touched from this leg)."""
    ref_rec, ref_why = reference_baseline(cfg)
    port, fsub = port_baseline(cfg, seconds_budget)
    if ref_rec is not None:
        ref_rec["_trajectory"] = port.pop("_trajectory")
        ref_rec["port"] = {k: port[k] for k in ("value", "cores", "physical_cores", "host", "phase_ms", "sample")}
        return ref_rec, fsub
    port["reference_attempt"] = ref_why
    return port, fsub


def usable_cpus():
    """CPUs this process can actually keep busy: the affinity mask, capped by the cgroup's CPU quota (cpu.max =
    "quota period": a container with 256 hardware threads visible and a quota of 16 runs 256 spinning OpenMP threads
    into the throttle — measured on the bench box: one rollout 27 ms with 16 threads, 100-900 ms with 256).  Returns
    (threads to use, a dict describing what was found)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    info = {"hardware_threads": os.cpu_count() or 1, "affinity": n, "cgroup_quota_cpus": None}
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            info["cgroup_quota_cpus"] = float(q) / float(per)
    except Exception:  # noqa: BLE001 — cgroup v1 / no cgroup
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                info["cgroup_quota_cpus"] = q / per
        except Exception:  # noqa: BLE001
            pass
    if info["cgroup_quota_cpus"]:
        n = max(1, min(n, int(math.floor(info["cgroup_quota_cpus"] + 1e-9))))
    if os.environ.get("OMP_NUM_THREADS"):
        n = int(os.environ["OMP_NUM_THREADS"])
    return n, info


def port_baseline(cfg, seconds_budget=15.0):
    """The oracle (a port, NOT the JAX reference) timed on the host cores on a
    bounded sample of the same workload: consecutive reverse-diffusion steps of the config.  Also returns the
    op counter's F_sub for the config's model (the oracle may only be touched from this leg)."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as orc_mod
    from oracle import planner as op
    from mbd_hip.model import Model
    orc_mod.build()
    orc = orc_mod.Oracle("f32_omp")
    threads, cpu_info = usable_cpus()
    orc.lib.orc_set_threads.restype = C.c_int
    threads = int(orc.lib.orc_set_threads(C.c_int(threads)))
    name, N, H, Nd, temp = cfg["env"], cfg["N"], cfg["H"], cfg["Nd"], cfg["temp"]
    compiled = os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled")
    fsub = None
    if name == "car2d":
        env = op.OracleEnv(orc, "car2d", xref=np.load(os.path.join(compiled, "car2d_xref.npy")))
    else:
        with open(os.path.join(compiled, f"{name}.json")) as f:
            m = Model.from_json(f.read())
        xref = np.load(os.path.join(compiled, "jog_xref.npy")) if name == "humanoidtrack" else None
        env = op.OracleEnv(orc, name, m.to_struct(), xref=xref, rew_xref=1.0 if xref is not None else 0.0,
                           init_q=m.init_q)
        ms = m.to_struct()
        s = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
        a = np.full(m.act_size(), 0.3, np.float32)
        for _ in range(12):  # the protocol of tools/count_ops.py: a state with the feet on the ground
            s, _ = orc.env_step(ms, s, a)
        fsub, _ = orc_mod.count_substep(ms, s, a)
    key = orc.prng_key(0)
    rng, rng_reset = orc.split(key, 2, 1)
    state0 = env.reset(rng_reset, 1)
    sched = orc.schedule(1e-4, 1e-2, Nd)
    Ybar = np.zeros((H, env.Nu), np.float32)
    r = orc.split(rng, 2, 1)[0]
    # one untimed step first: the OpenMP runtime starts its threads, the pages of the buffers get touched
    op.reverse_once(orc, env, state0, Nd - 1, r, Ybar, sched, N, H, temp, 1, enable_demo=cfg["demo"])
    steps, t0 = 0, time.time()
    i = Nd - 1
    phases = {}
    traj = {"seed": 0, "state_init": np.asarray(state0, np.float32), "mu_0ts": [], "rew_means": [], "rew_final": None}
    while True:
        r, Ybar, rm, _ = op.reverse_once(orc, env, state0, i, r, Ybar, sched, N, H, temp, 1, enable_demo=cfg["demo"],
                                         timers=phases)
        traj["mu_0ts"].append(Ybar)      # (the plan of seed 0 from its first step on: what `parity` compares)
        traj["rew_means"].append(rm)
        steps += 1
        i -= 1
        if time.time() - t0 > seconds_budget or i < 1:  # 10-15 s of host work, at most one whole plan
            break
    dt = time.time() - t0
    if i < 1:  # the whole plan ran: its final evaluation too (mbd_planner.py:179-180), outside the timed sample
        traj["rew_final"] = float(op.mean_h(orc, np.ascontiguousarray(env.rollout(state0, Ybar[None])))[0])
    try:  # physical cores beside the hardware threads OpenMP uses (SMT siblings share the FP units)
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:  # noqa: BLE001
        physical = None
    phase_ms = {k: 1e3 * v / steps for k, v in phases.items()}
    # every phase runs on all threads (oracle/mbd_oracle_core.c: the sampler over candidates, the weighted mean over its
    # columns; oracle/mbd_oracle_physics.c: the rollout over candidates); what stays serial inside a step is the [N]
    # standardise / softmax arithmetic (canonical reduction order), mean_H and the Python glue: the share of the step
    # outside the three timed library phases plus the score phase's serial part is reported as `serial_share_upper_bound`
    serial = max(0.0, dt - sum(phases.values())) + phases.get("score", 0.0)
    return {"value": steps / dt, "unit": "diffusion-steps/sec", "cores": threads, "physical_cores": physical,
            "host": cpu_info, "kind": "port", "phase_ms": phase_ms, "serial_share_upper_bound": serial / dt,
            "sample": f"{steps} consecutive reverse-diffusion steps of {name} N={N} H={H} "
                      f"(CPU oracle, OpenMP over candidates in sampler, rollout and weighted mean, {threads} threads = the CPUs "
                      f"this container may use (affinity, cgroup quota), {dt:.1f} s)", "_trajectory": traj}, fsub


def whole_run_parity(traj, gpu_details, gpu_rew_final):
    """The `parity` object of the line: the plan of seed 0 as the GPU ran it (run_diffusion through the C
    ABI) against the consecutive steps of the SAME plan the cpu_baseline leg's checker just ran — mu_0ts (Ybar after every
    step), the per-step mean rewards and, when the checker ran the whole plan, rew_final (mbd_planner.py:138-151,179-180).
    Not teacher-forced: step k sees what step k-1 left, on both sides.  Outside the timed region."""
    import numpy as np
    k = len(traj["mu_0ts"])
    cm, cr = np.stack(traj["mu_0ts"]).astype(np.float32), np.asarray(traj["rew_means"], np.float32)
    gm, gr = np.asarray(gpu_details["mu_0ts"], np.float32)[:k].reshape(cm.shape), np.asarray(gpu_details["rew_means"], np.float32)[:k]
    same_init = bool(np.array_equal(np.asarray(gpu_details["state_init"].pipeline_state, np.float32).reshape(-1),
                                    traj["state_init"].reshape(-1)))
    eq = same_init and bool(np.array_equal(cm, gm)) and bool(np.array_equal(cr, gr))
    den = np.maximum(np.abs(cm), 1e-6)
    out = {"against": "cpu checker (oracle/, the separately written C restatement; NOT the JAX reference)", "plan": "seed 0 of the "
           "line's config, free-running (every step from the previous step's own result)", "steps": int(k),
           "quantities": "state_init, mu_0ts[:steps], rew_means[:steps]" + (", rew_final" if traj["rew_final"] is not None else ""),
           "max_rel_mu": float((np.abs(cm - gm) / den).max()), "max_abs_rew_mean": float(np.abs(cr - gr).max())}
    if traj["rew_final"] is not None:
        out["rew_final_cpu"], out["rew_final_gpu"] = float(np.float32(traj["rew_final"])), float(np.float32(gpu_rew_final))
        eq = eq and np.float32(traj["rew_final"]) == np.float32(gpu_rew_final)
    out["bit_equal"] = bool(eq)
    out["max_rel"] = max(out["max_rel_mu"], out["max_abs_rew_mean"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="metric")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed blocks of exactly --steps steps each (fenced on both sides); `value` is the median block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-final-reward", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="default config on one GPU: skip the short extra measurements (the reference's default humanoidrun "
                         "size N=8192, its 8-seed sweep) that ride in the line under `extras`")
    ap.add_argument("--collective", choices=("torch", "p2p", "both"), default="both",
                    help="N > 1: the step's exchange — torch (all_gather_into_tensor: RCCL over xGMI), p2p (the in-library "
                         "windows, mbd_exchange_*), both (torch is `value`, p2p is measured after it and reported beside)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config], name=args.config)
    ENV, N_CFG, H, ND, TEMP, DEMO = cfg["env"], cfg["N"], cfg["H"], cfg["Nd"], cfg["temp"], cfg["demo"]
    if os.environ.get("MBD_BENCH_N"):  # experiments only
        N_CFG = int(os.environ["MBD_BENCH_N"])

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
    # the ranks the COLLECTIVE LIBRARY actually joined: a sum of ones through the process group itself (RCCL over xGMI with
    # the nccl backend), and the distinct devices behind them — the line fails loudly when either is not --gpus
    rccl_world = None
    if distributed:
        ones = torch.ones(1, dtype=torch.int32, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl_world = int(ones.item())
        if rccl_world != world or dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus}, WORLD_SIZE {world}, but the process group reduced over {rccl_world} "
                             f"rank(s) (backend {backend})")
        if backend == "nccl":
            devs = [None] * world
            # (host name from the socket layer — $HOSTNAME is often not exported to non-interactive processes, and two nodes'
            # ranks with the same local_rank would then collide — plus the device's own identity: two ranks that were handed
            # the same physical GPU under different local ranks share a PCI bus id / uuid)
            import socket
            ident = str(local_rank)
            try:
                pr = torch.cuda.get_device_properties(local_rank)
                ident = str(getattr(pr, "uuid", None) or getattr(pr, "pci_bus_id", None) or local_rank)
            except Exception:  # noqa: BLE001
                pass
            dist.all_gather_object(devs, (socket.gethostname(), ident))
            if len(set(devs)) != world:
                raise SystemExit(f"bench.py: {world} ranks share {len(set(devs))} device(s): {devs}")

    from mbd_hip import _capi
    from mbd_hip.envs import get_env
    from mbd_hip.planners.mbd_planner import Args, HostProgress, Plan, run_diffusion
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    env = get_env(ENV, device=local_rank)
    Nu, HNu = env.action_size, H * env.action_size
    n_frames = 1 if ENV == "car2d" else int(env.sys.fields["n_frames"])
    stream = torch.cuda.current_stream(dev).cuda_stream
    rows = 2 if DEMO else 1

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not distributed:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sweep_keys_states(seeds):
        keys, states = [], []
        for sd in seeds:
            rng, rng_reset = _capi.prng_split(_capi.prng_key(int(sd)), 2)
            states.append(env.reset(rng_reset))
            keys.append(_capi.prng_split(rng, 2)[0])  # rng_exp (:150)
        return np.array(keys, np.uint32), states

    def measure_sweep(seeds, steps=None, warmup=None, repeats=None, local=False):
        """config sweep8: blocks of K lockstep diffusion steps of the plans with these seeds (mbd_sweep_run with
        Ndiffuse = K + 1, no host outputs: the timed region is the K steps and nothing else).  No per-step host read: the
        reference's sweep prints nothing per step either (not_render runs of run_diffusion keep their progress bar, but
        the sweep's measure is the time of whole runs).  Returns (seconds of every block — max over ranks —, average
        rollout-launch ms, launches, this rank's own seconds per block).  local=True: no fence between the ranks and no
        reduction — this rank's own clock only (a rank that fails cannot leave the others waiting)."""
        from mbd_hip.planners.mbd_planner import Sweep
        P = len(seeds)
        steps, warmup = steps or args.steps, (args.warmup if warmup is None else warmup)
        repeats = repeats or args.repeats
        # ONE sweep object (its buffers are touched by the warm-up, not by the timed runs): warm-up runs of all its steps
        # (K >= W untimed steps), R timed runs of the same K steps, and one more run with HIP events around the rollout
        # launches for the kernel time (two event records cost a lockstep step ~7 us, 0.7 % — the value runs have none)
        nd = steps + 1  # (a timed run is exactly K steps)
        a = Args(seed=0, env_name=ENV, Nsample=N_CFG, Hsample=H, Ndiffuse=max(nd, 2), temp_sample=TEMP,
                 disable_recommended_params=True, not_render=True)
        sw = Sweep(env, a, P)
        try:
            keys, states = sweep_keys_states(seeds)
            for k, st0 in enumerate(states):
                sw.set_state0(k, st0)
            done = 0
            while done < max(warmup, 1):  # warm-up: whole runs until at least W steps have been taken
                sw.run(keys, outputs=False)
                done += steps
            blocks, own = [], []
            for _ in range(repeats):
                if local:
                    torch.cuda.synchronize(dev)
                else:
                    fence()
                t0 = time.perf_counter()
                sw.run(keys, outputs=False)
                torch.cuda.synchronize(dev)
                own.append(time.perf_counter() - t0)
                if local:
                    blocks.append(own[-1])
                    continue
                fence()
                blocks.append(max_over_ranks(time.perf_counter() - t0))
            sw.kernel_time(enable=True)
            sw.run(keys, outputs=False)
            torch.cuda.synchronize(dev)
            kern_ms, kern_n = sw.kernel_time(enable=False)
        finally:
            sw.close()
        return blocks, kern_ms, kern_n, own

    def sweep_plan_args(seeds):
        return [Args(seed=int(sd), env_name=ENV, Nsample=N_CFG, Hsample=H, Ndiffuse=ND, temp_sample=TEMP,
                     disable_recommended_params=True, not_render=True) for sd in seeds]

    def measure(N_total, N_local, collective="torch"):
        """K synced + K async steps of a plan with N_total candidates of which this rank owns N_local."""
        pargs = Args(seed=0, env_name=ENV, Nsample=N_total, Hsample=H, Ndiffuse=ND, temp_sample=TEMP,
                     enable_demo=DEMO, disable_recommended_params=True, not_render=True)
        key = _capi.prng_key(pargs.seed)
        rng, rng_reset = _capi.prng_split(key, 2)
        state_init = env.reset(rng_reset)
        rng_exp, _ = _capi.prng_split(rng, 2)
        plan = Plan(env, pargs, shard_begin=rank * N_local, shard_count=N_local)
        plan.set_state0(state_init)
        lib = plan.lib
        Ybar = torch.zeros(HNu, dtype=torch.float32, device=dev)
        Ynext = torch.zeros(HNu, dtype=torch.float32, device=dev)
        local = torch.zeros((rows, N_local), dtype=torch.float32, device=dev)
        allv = torch.zeros((rows, N_total), dtype=torch.float32, device=dev)
        gath = torch.zeros((max(world, 1) * rows, N_local), dtype=torch.float32, device=dev)
        rew_mean = torch.zeros(1, dtype=torch.float32, device=dev)
        # the per-step host read (mbd_planner.py:147) without a stream synchronisation: the score kernel stores the
        # step's mean reward into a pinned host slot and the host spins on it (mbd_hip HostProgress)
        host = HostProgress(1, dev)
        st = {"rng": np.asarray(rng_exp, np.uint32), "i": ND - 1, "Ybar": Ybar, "Ynext": Ynext}
        p2p = None
        if distributed and collective == "p2p":
            from mbd_hip.planners.mbd_planner import P2PExchange
            p2p = P2PExchange(local_rank, rows, N_local)

        p_loc0, p_loc1 = local[0].data_ptr(), (local[1].data_ptr() if DEMO else None)
        p_rewmean = rew_mean.data_ptr()

        def prepare():
            """Everything the next step needs that does not depend on the GPU — the key chain, the declaration of the
            step after it (mbd_plan_prefetch_noise) — done while the device runs the current step, BEFORE the host
            waits for its mean reward: the wait is then followed by the rollout launch and nothing else."""
            if st["i"] < 1:  # start the next plan: YN = zeros, fresh schedule position (enqueued behind the last update)
                st["i"] = ND - 1
                st["Ybar"].zero_()
            keys = st.get("keys")
            if keys is None:
                keys = _capi.prng_split(st["rng"], 2)
            st["rng"], st["ks"] = keys[0], _capi.key_array(keys[1])
            # the step after the next: its noise depends on its key only — declared now, generated beside the next
            # rollout (a hint; same results)
            st["keys"] = _capi.prng_split(st["rng"], 2)
            _capi.check(lib.mbd_plan_prefetch_noise(plan.h, _capi.key_array(st["keys"][1]), stream))
            st["pY"], st["pYn"] = st["Ybar"].data_ptr(), st["Ynext"].data_ptr()

        def step(read_back):
            i, ks, pY, pYn = st["i"], st["ks"], st["pY"], st["pYn"]
            _capi.check(lib.mbd_plan_sample_rollout(plan.h, i, ks, pY, p_loc0, p_loc1, stream))
            if p2p is not None:  # the in-library exchange: peer stores into every rank's window + flags
                base = p2p.all_gather(local, stream)
                p_s0, p_s1 = base, (base + 4 * N_total if DEMO else None)
            elif distributed and backend == "nccl":
                dist.all_gather_into_tensor(gath, local)  # the ONE collective of a diffusion step (RCCL/xGMI)
                src = gath.view(world, rows, N_local).permute(1, 0, 2).reshape(rows, N_total) if rows > 1 else \
                    gath.view(1, N_total)
                src = src.contiguous() if rows > 1 else src
                p_s0, p_s1 = src[0].data_ptr(), (src[1].data_ptr() if DEMO else None)
            elif distributed:  # gloo dry run: stage through the host
                staged = torch.empty((world * rows, N_local), dtype=torch.float32)
                dist.all_gather_into_tensor(staged, local.cpu())
                allv.copy_(staged.view(world, rows, N_local).permute(1, 0, 2).reshape(rows, N_total))
                p_s0, p_s1 = allv[0].data_ptr(), (allv[1].data_ptr() if DEMO else None)
            else:
                p_s0, p_s1 = p_loc0, p_loc1
            if read_back:
                host.reset(0)
            _capi.check(lib.mbd_plan_score_update(plan.h, i, ks, pY, p_s0, p_s1, pYn,
                                                  host.ptr(0) if read_back else p_rewmean, stream))
            st["Ybar"], st["Ynext"] = st["Ynext"], st["Ybar"]
            st["i"] = i - 1
            prepare()
            if read_back:  # pbar.set_postfix({"rew": f"{rew:.2e}"}) (mbd_planner.py:147): the host has the step's
                return host.wait(0)  # mean reward in hand before it dispatches the next step
            return None

        prepare()

        def timed(read_back):
            for _ in range(args.warmup):
                step(read_back)
            fence()
            # The rollout kernel's duration comes from HIP events around every launch on the launch stream.  Two event
            # records per step cost the stream ~7.5 us (profiles/r02_timeline.md: ~5.5 us of idle queue per record),
            # 1.2 % of a step — so they bracket the launches of the K steps of the value_async leg, and the headline
            # leg (value: the K steps with the per-step host read) runs uninstrumented.  MBD_BENCH_EVENTS=all|none
            # moves them (experiments).
            ev_mode = os.environ.get("MBD_BENCH_EVENTS", "async")
            plan.enable_timing(ev_mode == "all" or (ev_mode == "async" and not read_back))
            blocks = []
            for _ in range(args.repeats):  # R blocks of exactly K steps, each fenced on both sides
                fence()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step(read_back)
                fence()
                blocks.append(max_over_ranks(time.perf_counter() - t0))
            plan.enable_timing(False)
            kern_ms, kern_n = plan.kernel_time()
            return blocks, kern_ms, kern_n

        sync = timed(True)
        asyn = timed(False)
        if p2p is not None:
            p2p.status()
            p2p.close()
        plan.close()
        return sync, asyn

    def med(xs):
        return float(np.median(np.asarray(xs, np.float64)))

    strong_local = max(1, N_CFG // world)
    runs = {"strong": (strong_local * world, strong_local)}
    if world > 1:
        runs["weak"] = (N_CFG * world, N_CFG)
    is_sweep = "plans" in cfg
    sweep_plans = {}
    if is_sweep:
        # the reference's only timing protocol (scripts/run_mbd.py:17-39): independent plans -> REPLICAS over the ranks,
        # no communication inside a run.  strong: the sweep's P plans over the G GPUs (seeds r*P/G ...: at G = P one plan
        # per GPU, the one-candidate kernel); weak: P plans per GPU, seeds 0..G*P-1
        P = cfg["plans"]
        if P % world:
            raise SystemExit(f"sweep8: {P} plans do not divide over {world} ranks")
        sweep_plans["strong"] = list(range(rank * (P // world), (rank + 1) * (P // world)))
        if world > 1:
            sweep_plans["weak"] = list(range(rank * P, (rank + 1) * P))
        res = {}
        for k, seeds in sweep_plans.items():
            blocks, kms, kn, own = measure_sweep(seeds)
            res[k] = (blocks, blocks, kms, kn, own)
    else:
        main_coll = "p2p" if args.collective == "p2p" else "torch"
        res = {}
        for k, v in runs.items():
            if k == args.scaling or world > 1:
                (bs, _, _), (ba, kms, kn) = sy, asy = measure(*v, collective=main_coll)
                if kn == 0:  # MBD_BENCH_EVENTS=all / none
                    kms, kn = sy[1], sy[2]
                res[k] = (bs, ba, kms, kn, None)
    if args.scaling not in res:  # one GPU, --scaling weak: the same measurement
        res[args.scaling] = res["strong"]
    # the other collective, after the headline measurement and guarded: a failure here (a peer that cannot map a
    # window, a wait that runs into its limit) is reported, it never costs the line its `value`
    other_coll = None
    if distributed and world > 1 and args.collective == "both" and not is_sweep:
        # a canary first: one short-lived process per rank maps the peers' windows and runs a few exchanges with known
        # values (mbd_hip.planners.exchange_canary).  Peer stores into mapped memory are what ends in an uncatchable GPU
        # fault when a node's peer access is not what the code assumes — that must cost a canary, not this process and
        # its RCCL measurement.  Every rank learns the worst exit status before anybody goes on.
        import shutil
        import subprocess
        import tempfile
        # a FRESH rendezvous directory per run (rank 0 makes it, everybody learns its name): files of a crashed earlier
        # run — stale IPC handles, stale barrier files — can never be read
        box = [tempfile.mkdtemp(prefix="mbd_canary_") if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rdv = box[0]
        try:
            try:
                cp = subprocess.run([sys.executable, "-m", "mbd_hip.planners.exchange_canary", str(rank), str(world),
                                     str(local_rank), rdv], cwd=os.path.join(ROOT, "model-based-diffusion_amd"),
                                    capture_output=True, text=True, timeout=240)
                canary_rc, canary_err = cp.returncode, cp.stderr[-300:]
            except Exception as e:  # noqa: BLE001
                canary_rc, canary_err = 99, f"{type(e).__name__}: {e}"
            worst = torch.tensor([abs(canary_rc)], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        finally:
            if rank == 0:
                shutil.rmtree(rdv, ignore_errors=True)
        if int(worst.item()) != 0:
            other_coll = {"collective": "p2p", "error": f"canary failed (worst exit status {int(worst.item())}; rank {rank}: "
                                                            f"{canary_rc} {canary_err.strip()[-200:]})"}
    if distributed and world > 1 and args.collective == "both" and not is_sweep and other_coll is None:
        try:
            (ocs, _, _), (oca, okms, _) = measure(*runs[args.scaling], collective="p2p")
            other_coll = {"collective": "p2p (mbd_exchange_*: hipIpc-mapped windows, peer stores + epoch flags)",
                          "elapsed": med(ocs), "elapsed_async": med(oca), "kernel_avg_ms": okms}
        except Exception as e:  # noqa: BLE001
            other_coll = {"collective": "p2p", "error": f"{type(e).__name__}: {e}"}
        ok = torch.tensor([0 if "error" in other_coll else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and "error" not in other_coll:
            other_coll = {"collective": "p2p", "error": "failed on another rank"}

    # where a sharded step's time goes: HIP events around phase 1 (sample + rollout), the exchange and phase 2 (score +
    # weighted mean) of a whole plan, each phase fenced (measurement only; run_diffusion(measure_phases=True))
    phase_ms = None
    if distributed and not is_sweep:
        a = Args(seed=0, env_name=ENV, Nsample=runs[args.scaling][0], Hsample=H, Ndiffuse=min(ND, 30), temp_sample=TEMP,
                 enable_demo=DEMO, disable_recommended_params=True, not_render=True)
        with contextlib.redirect_stdout(sys.stderr):
            _, det = run_diffusion(a, device=local_rank, return_details=True, measure_phases=True,
                                   collective="torch" if args.collective != "p2p" else "p2p")
        phase_ms = det["phase_ms"]

    final = None
    if not args.no_final_reward and is_sweep:
        # the sweep's second half: rew_final of the COMPLETE plans (Ndiffuse of the config), every rank its own seeds in one
        # sweep, gathered once at the end; rank 0 then runs all of them as ONE sweep on its GPU — plans are independent,
        # the two must agree bit for bit
        # (the product's multi-GPU sweep: mbd_hip.scripts.run_mbd.run_replicated)
        from mbd_hip.scripts.run_mbd import run_concurrent, run_replicated
        P = cfg["plans"]
        rews, _, _ = run_replicated(sweep_plan_args(range(P)), device=local_rank)
        final = {"seeds": list(range(P)), "rew_final": rews, "mean": float(np.mean(rews)), "std": float(np.std(rews)),
                 "N": N_CFG}
        if distributed and rank == 0:
            single, _, _ = run_concurrent(sweep_plan_args(range(P)), local_rank)
            final["replicated_over"] = world
            final["equals_one_gpu_bitwise"] = bool(np.array_equal(np.float32(rews), np.float32(single)))
            final["rew_final_one_gpu"] = single
    elif not args.no_final_reward and (rank == 0 or distributed):
        # the metric's second half: final reward of complete plans, seeds 0..7 as mbd/scripts/run_mbd.py:20.  With
        # G > 1 ranks every plan runs SHARDED over the ranks (the product's multi-GPU path: run_diffusion ->
        # reverse_distributed) and, on rank 0, once more on one GPU: the two must agree bit for bit (variant B: every
        # rank re-derives the softmax and the weighted mean over all N from identical inputs)
        rews, single = [], []
        n_final = N_CFG if not distributed else runs[args.scaling][0]
        for seed in range(8):
            a = Args(seed=seed, env_name=ENV, Nsample=n_final, Hsample=H, Ndiffuse=ND, temp_sample=TEMP,
                     enable_demo=DEMO, disable_recommended_params=True, not_render=True)
            with contextlib.redirect_stdout(sys.stderr):  # the reference prints "init sigma = ..." (:92); stdout
                rews.append(float(run_diffusion(Args(**vars(a)), device=local_rank)))  # carries the one JSON line only
                if distributed and rank == 0:
                    single.append(float(run_diffusion(Args(**vars(a)), device=local_rank, force_single=True)))
        final = {"seeds": list(range(8)), "rew_final": rews, "mean": float(np.mean(rews)),
                 "std": float(np.std(rews)), "N": n_final}
        if distributed and rank == 0:
            final["sharded_over"] = world
            final["equals_one_gpu_bitwise"] = bool(np.array_equal(np.float32(rews), np.float32(single)))
            final["rew_final_one_gpu"] = single

    # Short extra measurements that ride in the default line (one GPU, config `metric`): the sizes above the metric's —
    # the reference's own default humanoidrun plan (N = 8192, mbd_planner.py:54-60) and its 8-seed sweep
    # (scripts/run_mbd.py:17-39) — so that the driver's record of this command holds them too.  ~3 s.  Each extra is
    # guarded on its own: a failure is reported as {"error": ...} under its name and never costs the line its value.
    extras = None
    if rank == 0 and not distributed and args.config == "metric" and not args.no_extras and not os.environ.get("MBD_BENCH_N"):
        extras = {}

        def extra_big():
            a = Args(seed=0, env_name=ENV, Nsample=8192, Hsample=H, Ndiffuse=41, temp_sample=TEMP,
                     disable_recommended_params=True, not_render=True)
            rng0, rng_reset = _capi.prng_split(_capi.prng_key(0), 2)
            big = Plan(env, a)
            try:
                big.set_state0(env.reset(rng_reset))
                big.run(_capi.prng_split(rng0, 2)[0])            # warm-up
                secs = [big.run(_capi.prng_split(rng0, 2)[0])[3] for _ in range(3)]
                big.enable_timing(True)
                big.run(_capi.prng_split(rng0, 2)[0])
                big.enable_timing(False)
                kms, _ = big.kernel_time()
            finally:
                big.close()
            return {"workload": "humanoidrun N=8192 H=50, 40 steps of mbd_plan_run (no per-step host read), median of 3 runs",
                    "steps_per_sec": 40 / med(secs), "kernel_avg_ms": kms, "_candidates": 8192,
                    "kernel": CONFIGS["humanoidrun8192"]["kernel"]}

        def extra_sweep():
            cfg_sw = CONFIGS["sweep8"]
            blocks, sw_kms, _, _ = measure_sweep(list(range(cfg_sw["plans"])), steps=40, warmup=5, repeats=3)
            return {"workload": "8 plans x humanoidrun N=1024 H=50 in lockstep, 40 steps of mbd_sweep_run, median of 3 runs",
                    "plan_steps_per_sec": cfg_sw["plans"] * 40 / med(blocks), "kernel_avg_ms": sw_kms,
                    "_candidates": 8192, "kernel": cfg_sw["kernel"]}

        for name, fn in (("humanoidrun8192", extra_big), ("sweep8", extra_sweep)):
            try:
                extras[name] = fn()
            except Exception as e:  # noqa: BLE001 — extras never cost the line its value
                extras[name] = {"error": f"{type(e).__name__}: {e}"}

    # With G > 1 ranks the metric's own curve is flat by construction (a rollout is latency-bound: N / G candidates take as
    # long as N).  The workload of the reference that DOES scale — its 8-seed sweep, plans as replicas — rides in the
    # same line: every rank runs 8 plans of N = 1024 (its own seeds) as one sweep on its GPU, no collective inside and no
    # fence between the ranks; ONE gather of the per-rank rates at the end, reached by every rank whatever happened.
    if distributed and world > 1 and args.config == "metric" and not args.no_extras and not os.environ.get("MBD_BENCH_N"):
        try:
            P_x = CONFIGS["sweep8"]["plans"]
            _, x_kms, _, x_own = measure_sweep(list(range(rank * P_x, (rank + 1) * P_x)), steps=40, warmup=5, repeats=3,
                                               local=True)
            mine = {"plan_steps_per_sec": P_x * 40 / med(x_own), "kernel_avg_ms": x_kms}
        except Exception as e:  # noqa: BLE001 — extras never cost the line its value
            mine = {"error": f"{type(e).__name__}: {e}"}
        box = [None] * world
        dist.all_gather_object(box, mine)
        if rank == 0:
            good = [b for b in box if isinstance(b, dict) and "plan_steps_per_sec" in b]
            extras = {"sweep8_replicas": {
                "workload": f"8 plans x humanoidrun N=1024 H=50 per GPU as one sweep each ({world} GPUs, {8 * world} plans), 40 "
                            "steps of mbd_sweep_run, median of 3 runs per rank; SUM of the ranks' own rates (no fence, no "
                            "collective: the plans are independent replicas) — `--config sweep8 --gpus G` is the fenced form",
                "plan_steps_per_sec": sum(b["plan_steps_per_sec"] for b in good), "ranks_ok": len(good), "per_rank": box}}

    per_rank = None
    if is_sweep and distributed:  # every rank's own rate (its plans, its clock), for the line
        own = res[args.scaling][4]
        mine = len(sweep_plans[args.scaling]) * args.steps / med(own)
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box

    if rank == 0:
        def rate(mode, which):
            """(median, min, max) rate and the median ms per step of the leg's R blocks"""
            bs, ba = res[mode][0], res[mode][1]
            blocks = bs if which == "sync" else ba
            if is_sweep:
                units = len(sweep_plans.get(mode, sweep_plans["strong"])) * world * args.steps  # plan-steps per block
            else:
                N_total, _ = runs.get(mode, runs["strong"])
                units = args.steps * (N_total / N_CFG if mode == "weak" else 1.0)
            return units / med(blocks), units / max(blocks), units / min(blocks), 1e3 * med(blocks) / args.steps

        mode = args.scaling
        kern_ms, kern_n = res[mode][2], res[mode][3]
        N_total, N_local = runs.get(mode, runs["strong"])
        value, vmin, vmax, ms_sync = rate(mode, "sync")
        value_async, amin, amax, ms_async = rate(mode, "async")
        balg = b_alg_bytes(N_local, H, Nu, DEMO)
        achieved = (balg / 1e9) / (kern_ms / 1e3) if kern_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(args.config)
        out = {
            "metric": "diffusion-steps/sec (N=1024, H=50) + final reward, humanoidrun, 1/2/4/8 GPU",
            "value": value, "unit": "diffusion-steps/sec (whole job, per-step host read of the mean reward included)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_sync, "higher_is_better": True, "scaling": mode,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": args.repeats, "value_min": vmin, "value_max": vmax,
            "value_note": f"median of {args.repeats} timed blocks of exactly {args.steps} steps, each fenced (barrier + "
                          "device synchronisation) on both sides, max over ranks per block",
            "value_async": value_async, "value_async_min": amin, "value_async_max": amax, "ms_per_step_async": ms_async,
            "config": {"workload": f"{args.config}: {ENV} N_total={N_total} ({N_local}/GPU) H={H} Ndiffuse={ND} "
                                   f"temp={TEMP}{' enable_demo' if DEMO else ''} seed=0 disable_recommended_params",
                       "name": args.config, "N_total": N_total, "N_per_gpu": N_local, "H": H, "Nu": Nu,
                       "n_frames": n_frames,
                       "collective": (("all-gather of N/G rewards per step (variant B; == all-reduce of the zero-padded vector) through " +
                                       ("torch.distributed (RCCL over xGMI)" if args.collective != "p2p" else
                                        "the in-library windows (mbd_exchange_*)") +
                                       " (variant B: every rank re-derives the softmax and the weighted mean over all N "
                                       "from identical inputs)") if distributed else "none")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": cfg["kernel"], "kernel_avg_ms": kern_ms,
                         "kernel_launches": kern_n, "kernel_timing": "HIP events around every rollout launch of the "
                         "value_async leg (same R x K steps; the value leg runs without them: two event records cost a "
                         "step ~7.5 us)", "algorithmic_bytes_per_launch": balg,
                         "note": "state stays in VGPRs for all H*n_frames substeps: the kernel is bound by "
                                 "dependent FP32 VALU issue, not HBM (DESIGN.md §Roofline)"},
            "final_reward": final,
        }
        n_valu = N_local
        if is_sweep:  # plan-steps: K lockstep steps of the plans
            P = cfg["plans"]
            p_local = len(sweep_plans[mode])
            out["unit"] = (f"plan-steps/sec: diffusion steps of the plans of a seed sweep ({p_local} per GPU in lockstep, "
                           "mbd_sweep_run; replicas over the GPUs, no communication inside a run)")
            out["config"]["workload"] = (f"sweep8: {p_local * world} plans x {ENV} N={N_CFG} H={H} temp={TEMP}, seeds "
                                         f"0..{p_local * world - 1}, {p_local} per GPU: one rollout launch over "
                                         f"{p_local * N_CFG} candidates + one score launch per step and GPU")
            out["config"].update(plans=p_local * world, plans_per_gpu=p_local, N_total=p_local * world * N_CFG,
                                 N_per_gpu=p_local * N_CFG,
                                 collective="none inside a run (independent plans); rew_final gathered once at the end")
            if p_local * N_CFG <= 4096:  # (the launch selection of csrc: two candidates per lane above 4096 candidates)
                out["roofline"]["kernel"] = CONFIGS["metric"]["kernel"] + f" over {p_local} plan(s) x {N_CFG} candidates"
            bal = p_local * b_alg_bytes(N_CFG, H, Nu, DEMO)
            out["roofline"].update(achieved=(bal / 1e9) / (kern_ms / 1e3) if kern_ms > 0 else 0.0,
                                   algorithmic_bytes_per_launch=bal)
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
            out["roofline"]["traffic"] = None
            out["roofline"]["kernel_timing"] = ("HIP events around every rollout launch of one MORE run of the same K "
                                                "lockstep steps (the value runs carry no events)")
            if per_rank is not None:
                out["per_rank_plan_steps_per_sec"] = per_rank
            n_valu = p_local * N_CFG
        if distributed:
            out["rccl_world"] = rccl_world
            out["dist_backend"] = backend + (" (RCCL over xGMI)" if backend == "nccl" else " (dry run: ranks may share a device)")
        # what a reader of the N = 1, 2, 4, 8 lines should expect BEFORE computing an efficiency from them (the driver does
        # that itself from `value`): the literal metric shards ONE plan's N candidates, and a rollout is latency-bound —
        # N / G candidates take as long as N while a launch leaves SIMDs idle (N <= 4096 on this chip)
        out["scaling_expectation"] = {
            "strong": "FLAT by design: phase 1 (sample + rollout) of N/G candidates costs what N do (one wavefront per SIMD "
                      "either way: ~0.55 ms at any N/G <= 1024 for humanoidrun), plus one all-gather of N/G rewards per rank and "
                      "step (~10-30 us): value(G) ~ value(1) x 0.93-0.97, efficiency ~ 1/G",
            "weak": "N per GPU, G x N in total: ~0.95 of linear in candidates per second (the same phase 1 per rank, the score over "
                    "G x N candidates recomputed on every rank: +8 us per 1024 candidates)",
            "sweep8_replicas": "LINEAR: independent plans as replicas over the ranks (the reference's own sweep protocol, "
                               "scripts/run_mbd.py:17-39), no collective inside a run — the workload of the reference that fills "
                               "more than one GPU; `--config sweep8 --gpus G` is its fenced form, extras.sweep8_replicas the "
                               "rider of the metric line",
            "measured_on_hardware": "no run on more than one DEVICE exists yet (rounds 1-5: the driver's SCALE was skipped, no "
                                    "8-GPU node); ranks > 1 have only run as processes sharing one GPU",
        }
        if phase_ms is not None:
            out["phase_ms"] = phase_ms
        if other_coll is not None:
            if "elapsed" in other_coll:
                nt = runs[mode][0]
                fac = (nt / N_CFG if mode == "weak" else 1.0)
                other_coll["value"] = args.steps / other_coll.pop("elapsed") * fac
                other_coll["value_async"] = args.steps / other_coll.pop("elapsed_async") * fac
            out["other_collective"] = other_coll
        if world > 1:
            other = "weak" if mode == "strong" else "strong"
            ov, _, _, oms = rate(other, "sync")
            oa, _, _, _ = rate(other, "async")
            out["other_scaling"] = {"scaling": other, "value": ov, "ms_per_step": oms, "value_async": oa}
            if is_sweep:
                out["other_scaling"].update(plans=len(sweep_plans[other]) * world, plans_per_gpu=len(sweep_plans[other]))
            else:
                out["other_scaling"].update(N_total=runs[other][0], N_per_gpu=runs[other][1])
        fsub, fsub_src = op_counts(ENV)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], live = cpu_baseline(cfg)
            if live:
                fsub, fsub_src = live, "oracle/count_ops.cc (this run)"
            pj = out["cpu_baseline"].pop("parity_jax", None) if isinstance(out["cpu_baseline"], dict) else None
            if pj is not None:
                if "rew_final_ndiffuse" in pj and not is_sweep:  # this library's rew_final of the reference's last timed arguments
                    try:
                        a = Args(seed=0, env_name=ENV, Nsample=N_CFG, Hsample=H, Ndiffuse=pj["rew_final_ndiffuse"], temp_sample=TEMP,
                                 enable_demo=DEMO, disable_recommended_params=True, not_render=True)
                        with contextlib.redirect_stdout(sys.stderr):
                            pj["rew_final_ours"] = float(run_diffusion(a, device=local_rank, force_single=True))
                    except Exception as e:  # noqa: BLE001
                        pj["rew_final_ours_error"] = f"{type(e).__name__}: {e}"
                out["parity_jax"] = pj
            traj = out["cpu_baseline"].pop("_trajectory", None)
            if traj is not None and not is_sweep and not os.environ.get("MBD_BENCH_N"):
                try:  # (guarded like the extras: a failure here is reported, it never costs the line its value)
                    a = Args(seed=0, env_name=ENV, Nsample=N_CFG, Hsample=H, Ndiffuse=ND, temp_sample=TEMP, enable_demo=DEMO,
                             disable_recommended_params=True, not_render=True)
                    with contextlib.redirect_stdout(sys.stderr):
                        g_rf, g_det = run_diffusion(a, device=local_rank, return_details=True, force_single=True)
                    out["parity"] = whole_run_parity(traj, g_det, g_rf)
                except Exception as e:  # noqa: BLE001
                    out["parity"] = {"error": f"{type(e).__name__}: {e}"}
        if is_sweep and n_valu <= 4096:  # (the one-candidate kernel's geometry and static counts)
            cfg = dict(cfg, cpw=4, static="humanoidrun")
        out["valu"] = valu_view(cfg, n_valu, kern_ms, n_frames, fsub, fsub_src)
        if extras is not None:
            for e in extras.values():  # (the VALU view of the extra workloads needs the op count of this line)
                if not isinstance(e, dict):
                    continue
                cands = e.pop("_candidates", None)
                if fsub and cands and e.get("kernel_avg_ms"):
                    e["valu_algorithmic_frac"] = (float(fsub["flops"]) * cands * H * n_frames /
                                                  (e["kernel_avg_ms"] * 1e-3) / 1e12 / VALU_PEAK_TF)
            out["extras"] = extras
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
