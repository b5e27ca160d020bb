"""bench.py's CPU-baseline leg tries the REAL reference first (jax + brax + a checkout of the
reference) and falls back to the port.  Neither jax nor brax exists in this image, so the reference branch is
exercised with stub modules that have the reference's call surface (mbd.planners.mbd_planner.Args / run_diffusion)."""
import importlib
import os
import sys
import textwrap

import pytest

from conftest import ROOT


@pytest.fixture()
def bench_mod():
    sys.path.insert(0, ROOT)
    try:
        yield importlib.import_module("bench")
    finally:
        sys.path.remove(ROOT)


def test_falls_back_to_the_port_when_the_reference_cannot_be_imported(bench_mod, monkeypatch):
    monkeypatch.setenv("MBD_REFERENCE_PATH", "/nonexistent")
    rec, why = bench_mod.reference_baseline(bench_mod.CONFIGS["car2d"])
    assert rec is None and ("ModuleNotFoundError" in why or "ImportError" in why)
    port, _ = bench_mod.cpu_baseline(bench_mod.CONFIGS["car2d"], seconds_budget=0.5)
    assert port["kind"] == "port" and port["value"] > 0 and "reference_attempt" in port


def test_reference_branch_with_stub_modules(bench_mod, tmp_path, monkeypatch):
    stubs = tmp_path / "stubs"
    ref = tmp_path / "ref"
    for d in (stubs / "jax", stubs / "brax", ref / "mbd" / "planners"):
        d.mkdir(parents=True)
    (stubs / "jax" / "__init__.py").write_text("__version__ = '0.0-stub'\n")
    (stubs / "brax" / "__init__.py").write_text("__version__ = '0.0-stub'\n")
    (ref / "mbd" / "__init__.py").write_text("")
    (ref / "mbd" / "planners" / "__init__.py").write_text("")
    (ref / "mbd" / "planners" / "mbd_planner.py").write_text(textwrap.dedent('''
        import time
        from dataclasses import dataclass
        CALLS = []
        @dataclass
        class Args:
            seed: int = 0
            disable_recommended_params: bool = False
            not_render: bool = False
            env_name: str = "ant"
            Nsample: int = 2048
            Hsample: int = 50
            Ndiffuse: int = 100
            temp_sample: float = 0.1
            beta0: float = 1e-4
            betaT: float = 1e-2
            enable_demo: bool = False
        def run_diffusion(args):
            CALLS.append(args)
            time.sleep(0.05 + 0.01 * (args.Ndiffuse - 1))   # "compilation" + 10 ms per diffusion step
            print("init sigma = stub")                        # (the reference prints to stdout: must not reach ours)
            return 1.0
    '''))
    monkeypatch.syspath_prepend(str(stubs))
    monkeypatch.setenv("MBD_REFERENCE_PATH", str(ref))
    for m in ("jax", "brax", "mbd", "mbd.planners", "mbd.planners.mbd_planner"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    try:
        rec, why = bench_mod.reference_baseline(bench_mod.CONFIGS["metric"], seconds_budget=0.6)
        assert why is None and rec["kind"] == "jax-reference" and rec["versions"] == {"jax": "0.0-stub", "brax": "0.0-stub"}
        assert 50 < rec["value"] < 200, rec  # 10 ms per step -> ~100 steps/s, the fixed 50 ms cancelled out
        calls = sys.modules["mbd.planners.mbd_planner"].CALLS
        assert all(c.env_name == "humanoidrun" and c.Nsample == 1024 and c.Hsample == 50 and c.disable_recommended_params
                   and c.not_render for c in calls)
        assert calls[0].Ndiffuse == 2 and calls[-1].Ndiffuse > 2
        # round 6: the same branch also dumps the reference's records and holds the checker to them (`parity_jax`).  The stub
        # reference has no envs to dump: the failure is reported INSIDE the object, with what the timed run returned
        pj = rec["parity_jax"]
        assert pj["versions"] == {"jax": "0.0-stub", "brax": "0.0-stub"} and pj["rew_final_ref"] == 1.0
        assert pj["rew_final_ndiffuse"] == calls[-1].Ndiffuse and "error" in pj and pj["env"] == "humanoidrun"
    finally:
        for m in ("jax", "brax", "mbd", "mbd.planners", "mbd.planners.mbd_planner"):
            sys.modules.pop(m, None)
        if str(ref) in sys.path:
            sys.path.remove(str(ref))


def test_whole_run_parity_record(bench_mod):
    """bench.py's `parity` object: bit equality of the checker's consecutive steps with the GPU's run of the same plan, and
    the size of a difference when there is one."""
    import types

    import numpy as np
    g = np.random.default_rng(0)
    mu = g.normal(size=(5, 4, 3)).astype(np.float32)
    rm = g.normal(size=5).astype(np.float32)
    st = g.normal(size=(2, 13)).astype(np.float32)
    traj = {"seed": 0, "state_init": st, "mu_0ts": list(mu[:3]), "rew_means": list(rm[:3]), "rew_final": None}
    det = {"mu_0ts": mu, "rew_means": rm, "state_init": types.SimpleNamespace(pipeline_state=st.copy())}
    rec = bench_mod.whole_run_parity(traj, det, 0.5)
    assert rec["bit_equal"] is True and rec["steps"] == 3 and rec["max_rel"] == 0.0 and "rew_final_cpu" not in rec
    traj["mu_0ts"], traj["rew_means"], traj["rew_final"] = list(mu), list(rm), 0.5
    assert bench_mod.whole_run_parity(traj, det, 0.5)["bit_equal"] is True
    assert bench_mod.whole_run_parity(traj, det, 0.25)["bit_equal"] is False
    mu2 = mu.copy()
    mu2[4, 0, 0] *= np.float32(1.0 + 2e-5)
    rec = bench_mod.whole_run_parity(traj, dict(det, mu_0ts=mu2), 0.5)
    assert rec["bit_equal"] is False and 1e-5 < rec["max_rel"] < 1e-4


def test_port_baseline_hands_its_trajectory_to_the_parity_leg(bench_mod):
    port, _ = bench_mod.port_baseline(bench_mod.CONFIGS["car2d"], seconds_budget=30.0)
    tr = port["_trajectory"]
    assert len(tr["mu_0ts"]) == 49 and tr["rew_final"] is not None and tr["state_init"].shape[-1] == 3


def test_parity_jax_object_from_a_dump(bench_mod, orc, tmp_path, monkeypatch):
    """bench.py::parity_vs_jax on a file in tools/dump_golden.py's schema (written here by THIS repo's checker under a planted
    switch, like tests/test_golden.py's: plumbing, not a golden): the stage the default specification misses first, the word of
    switches that fits, the teacher-forced relative error — what a box with jax + brax will put into the line."""
    import sys

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_golden
    from conftest import load_model
    from test_golden import _synthetic_stage_file
    planted = 12   # ("Brax" = the checker with stage (6) Gauss-Seidel on top of the default word, contact_avg)

    def fake_dump(ref, env_name, N, H, steps, out_dir=".", records="ABC"):
        path = os.path.join(out_dir, f"golden_{env_name}_N{N}_H{H}.npz")
        _synthetic_stage_file(orc, path, env_name, flags=planted, steps=20, action=0.0)
        g = dict(np.load(path))
        m = load_model(env_name).with_spec(planted)
        q0, qd0 = m.init_q, np.zeros(m.qd_size(), np.float32)
        st = orc.forward(m.to_struct(), q0, qd0)
        Y = np.clip(np.random.default_rng(0).normal(size=(N, H, m.act_size())) * 0.5, -1, 1).astype(np.float32)
        w = np.full(N, 1.0 / N, np.float32)
        g.update(q0=q0, qd0=qd0, Y0s_0=Y, rewss_0=orc.rollout(m.to_struct(), st, Y), weights_0=w,
                 Ybar_0=np.einsum("n,nij->ij", w.astype(np.float64), Y.astype(np.float64)).astype(np.float32))
        np.savez(path, **g)
        return path
    monkeypatch.setattr(dump_golden, "dump", fake_dump)
    pj = bench_mod.parity_vs_jax("/nonexistent", dict(bench_mod.CONFIGS["hopper512"], N=8, H=6), {"jax": "x", "brax": "y"},
                                 {"rew_final": 2.5, "ndiffuse": 6})
    assert "error" not in pj, pj
    assert pj["rew_final_ref"] == 2.5 and pj["rew_final_ndiffuse"] == 6 and pj["golden"] == "golden_hopper_N8_H6.npz"
    assert pj["first_mismatch_stage"] == "6_contact_velocity" or pj["first_mismatch_stage"] == "contact:6_contact_velocity"
    assert pj["fitted_flags"] == planted and pj["fitted"] == ["contact_avg", "contact6_gauss_seidel"] and pj["fitted_first_mismatch_stage"] is None
    # teacher-forced under the fitted word: the checker reproduces the "reference" exactly
    assert pj["flags_of_the_teacher_forced_model"] == planted and pj["max_rel"] < 1e-6 and pj["within_tolerance"] is True
    assert pj["records"]["reverse_once_steps"] == 1 and pj["records"]["substep_stages"] and pj["records"]["settled_substep"]
