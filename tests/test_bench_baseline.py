"""bench.py's CPU-baseline leg tries the REAL reference first (jax + brax + a checkout of the
reference) and falls back to the port.  Neither jax nor brax exists in this image, so the reference branch is
exercised with stub modules that have the reference's call surface (mbd.planners.mbd_planner.Args / run_diffusion)."""
import importlib
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
