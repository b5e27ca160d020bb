"""Golden vectors.

(1) tests/golden/self_*.npz — produced by tools/make_golden.py from THIS repo's oracle: they pin the
    oracle (and through the GPU tests the kernels) against accidental changes of the numerical contract.
(2) tests/golden/golden_*.npz — produced by tools/dump_golden.py under a real jax+brax install. None can
    be generated in the build container (jax/brax absent, no network): while absent the reference parity of
    the Brax-backed envs is UNPINNED and this test says so instead of passing silently."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT, load_model

GOLD = os.path.join(ROOT, "tests", "golden")


def test_self_goldens_pin_the_numerical_contract(orc):
    files = sorted(glob.glob(os.path.join(GOLD, "self_*.npz")))
    assert files, "run tools/make_golden.py"
    from oracle import planner as op
    for f in files:
        g = np.load(f)
        name = str(g["env"])
        if name == "car2d":
            xref = np.load(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", "car2d_xref.npy"))
            env = op.OracleEnv(orc, "car2d", xref=xref)
        else:
            m = load_model(name)
            env = op.OracleEnv(orc, name, m.to_struct(), init_q=m.init_q)
        res = op.run_diffusion(orc, env, int(g["seed"]), int(g["N"]), int(g["H"]), int(g["Nd"]), float(g["temp"]),
                               impl=int(g["impl"]), max_steps=int(g["steps"]))
        assert np.array_equal(res["state_init"], g["state_init"]), f
        assert np.array_equal(res["mu_0ts"], g["mu_0ts"]), f
        assert np.array_equal(res["rew_means"], g["rew_means"]), f


def test_reference_goldens_or_report_unpinned(orc):
    """Consumes tests/golden/golden_*.npz (tools/dump_golden.py under a real jax + brax install): the compiled
    system, ONE substep stage by stage (tools/compare_golden.py names the first stage and link beyond 1e-5) and the
    teacher-forced first-step rewards.  While no such file exists the reference parity of the Brax-backed envs is
    UNPINNED and this test says so instead of passing silently."""
    files = sorted(glob.glob(os.path.join(GOLD, "golden_*.npz")))
    if not files:
        pytest.xfail("parity unpinned: no jax+brax golden vectors (tools/dump_golden.py, see tests/golden/README.md) "
                     "are available — the physics is NOT held to the reference yet")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import compare_golden
    for f in files:
        g = np.load(f)
        name = os.path.basename(f).split("_")[1]
        if "substep_in_x_pos" in g:
            lines, first = compare_golden.compare(f, 1e-5)
            assert first is None, "\n".join(lines)
        m = load_model(name)
        ms = m.to_struct()
        st = orc.forward(ms, g["q0"].astype(np.float32), g["qd0"].astype(np.float32))
        rew = orc.rollout(ms, st, g["Y0s_0"].astype(np.float32))
        assert np.allclose(rew[:, 0], g["rewss_0"][:, 0], rtol=1e-5, atol=1e-6), "first-step rewards differ from Brax"


def _synthetic_stage_file(orc, path, name, corrupt=None, flags=0, steps=10, action=0.3):
    """A file in tools/dump_golden.py's schema whose records come from THIS repo's oracle (tmp only, never
    committed): exercises tools/compare_golden.py's plumbing — it is not a golden of the reference."""
    import ctypes as C
    m = load_model(name).with_spec(flags)  # (flags: the specification switches the "reference" of this file decides by)
    ms = m.to_struct()
    L = m.n_links
    s = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
    a = np.full(m.act_size(), action, np.float32)
    for _ in range(steps):
        s, _ = orc.env_step(ms, s, a)
    f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    orc.lib.orc_substep_stages.argtypes = [C.c_void_p, f32, f32, f32, f32]
    out, stages = np.zeros((L, 13), np.float32), np.zeros((6, L, 13), np.float32)
    orc.lib.orc_substep_stages(C.addressof(ms), s.reshape(-1), a, out.reshape(-1), stages.reshape(-1))
    rec = dict(substep_action=a, stage_composition_matches_pipeline_step=np.asarray(True),
               sys_link_mass=1.0 / np.asarray(m.fields["inv_mass"], np.float64),
               sys_actuator_gear=np.asarray(m.fields["act_gear"], np.float64))

    def put(prefix, st):
        rec[f"{prefix}_x_pos"], rec[f"{prefix}_x_rot"] = st[:, 0:3].copy(), st[:, 3:7].copy()
        rec[f"{prefix}_xd_vel"], rec[f"{prefix}_xd_ang"] = st[:, 7:10].copy(), st[:, 10:13].copy()
    put("substep_in", s)
    put("substep_out", out)
    import compare_golden
    for k, stn in enumerate(compare_golden.STAGES):
        put(f"stage_{stn}", stages[k])
    rec["stage_1_acceleration_xdd_vel"] = stages[0][:, 7:10] + np.asarray(m.fields["gravity"], np.float32)
    rec["stage_1_acceleration_xdd_ang"] = stages[0][:, 10:13].copy()
    # the second record of the schema: the substep from a settled state, keys prefixed "contact_"
    for k in [k for k in rec if not k.startswith("sys_")]:
        rec["contact_" + k] = np.array(rec[k], copy=True)
    if corrupt:
        stage, link, key, delta = corrupt
        rec[f"stage_{stage}_{key}"][link, 0] += delta
    np.savez(path, **rec)


@pytest.mark.parametrize("name", ["humanoidrun", "hopper"])
def test_compare_golden_localises_a_mismatch(orc, tmp_path, name):
    """tools/compare_golden.py on files in the dump schema: all stages agree when the records are the oracle's own,
    and a deviation planted in ONE stage of ONE link is reported as exactly that stage and link."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import compare_golden
    ok = str(tmp_path / f"golden_{name}_N1_H1.npz")
    _synthetic_stage_file(orc, ok, name)
    lines, first = compare_golden.compare(ok, 1e-5)
    assert first is None, "\n".join(lines)
    bad = str(tmp_path / "bad" / f"golden_{name}_N1_H1.npz")
    os.makedirs(os.path.dirname(bad))
    _synthetic_stage_file(orc, bad, name, corrupt=("3_joint_position", 2, "x_pos", 3e-4))
    lines, first = compare_golden.compare(bad, 1e-5)
    assert first is not None and first[0] == "3_joint_position" and first[1] == 2 and first[2] == "pos", lines
    assert any("FIRST MISMATCH" in l and "3_joint_position" in l for l in lines)
    # ... and one planted in the settled-state record only (stage 6: the friction bound's stage) is named as such
    bad2 = str(tmp_path / "bad2" / f"golden_{name}_N1_H1.npz")
    os.makedirs(os.path.dirname(bad2))
    _synthetic_stage_file(orc, bad2, name, corrupt=("6_contact_velocity", 1, "xd_vel", 2e-3))
    g = dict(np.load(bad2))
    g["stage_6_contact_velocity_xd_vel"], g["contact_stage_6_contact_velocity_xd_vel"] = \
        g["contact_stage_6_contact_velocity_xd_vel"], g["stage_6_contact_velocity_xd_vel"]
    np.savez(bad2, **g)
    lines, first = compare_golden.compare(bad2, 1e-5)
    assert first is not None and first[0] == "contact:6_contact_velocity" and first[1] == 1, lines


@pytest.mark.parametrize("name,planted,steps,action", [
    ("humanoidstandup", 8, 80, 0.0), ("humanoidstandup", 4, 80, 0.0), ("humanoidstandup", 4 | 8, 80, 0.0),  # lying on its back
    ("humanoidstandup", 16, 10, 0.3), ("humanoidrun", 64, 10, 0.3), ("humanoidrun", 64 | 16, 30, 0.3),
    ("hopper", 8, 20, 0.0), ("hopper", 8 | 4, 20, 0.0), ("humanoidrun", 0, 10, 0.3)])
def test_compare_golden_search_finds_the_planted_switches(orc, tmp_path, name, planted, steps, action):
    """tools/compare_golden.py --search (round-3 verdict item 3): a golden whose "reference" decides some of DESIGN.md §9's
    code-level guesses the other way — planted here by producing the file with the checker under the default word with the
    `planted` switches FLIPPED (word = DEFAULT_SPEC ^ planted: since round 6 the default has contact_avg set, so planting 4
    means a "reference" that sums) — is replayed under all 64 words; the best-ranked one must reproduce the file (every stage
    within tolerance), must flip nothing that was not planted, and the default word must NOT fit (unless nothing was planted:
    then the default wins)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import compare_golden
    from mbd_hip.model import DEFAULT_SPEC
    word = DEFAULT_SPEC ^ planted
    f = str(tmp_path / f"golden_{name}_N1_H1.npz")
    _synthetic_stage_file(orc, f, name, flags=word, steps=steps, action=action)
    rows = compare_golden.search(f, 1e-6)
    best = rows[0]
    assert best[2] is None, rows[:4]
    by_flags = {r[0]: r for r in rows}
    assert by_flags[word][2] is None                         # the planted word fits ...
    flipped = best[0] ^ DEFAULT_SPEC
    if planted:
        assert by_flags[DEFAULT_SPEC][2] is not None, "the planted switches did not act on this state"
        assert flipped & planted == flipped, rows[:4]        # ... and the winner flips nothing that was not planted
        # every planted switch the winner leaves alone must be one that does not act here (the file fits without flipping it)
        for r in rows:
            if r[2] is None:
                assert (r[0] ^ DEFAULT_SPEC) & flipped == flipped, (best, r)   # all fitting words flip the winner's switches
    else:
        assert best[0] == DEFAULT_SPEC
