"""tools/dump_golden.py is the one door to "parity green" (SURVEY.md §8(c)): it needs a real jax + brax, which no box of this
project has — so until round 6 it had never executed anywhere.  Here its record (A) — the reverse_once inputs / outputs —
runs under tools/make_ref_golden.py's numpy stand-in for jax with Brax's PipelineEnv served by this repo's checker: the
reference's own envs, rollout_us and get_env are imported and executed by dump(), and the file it writes is read back by the
consumer's code.  NOT a golden of the reference (the physics underneath is this repo's): a test that the hatch opens."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_model

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mbd")), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("env_name,N,H", [("hopper", 6, 5), ("humanoidrun", 4, 3)])
def test_dump_golden_record_A_runs_under_the_numpy_stand_in(orc, tmp_path, env_name, N, H):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_golden
    import make_ref_golden as mrg
    saved_modules, saved_path, saved_bc = dict(sys.modules), list(sys.path), sys.dont_write_bytecode
    try:
        mrg.install(orc, 1)
        mrg.BRAX["orc"] = orc
        base = sys.modules["brax.envs.base"]
        base.PipelineEnv, base.State = mrg.OrcPipelineEnv, mrg.BraxEnvState
        for k in [k for k in sys.modules if k == "mbd" or k.startswith("mbd.")]:
            del sys.modules[k]
        path = dump_golden.dump(REF, env_name, N, H, 2, out_dir=str(tmp_path), records="A")
        state_init = np.array(mrg.BRAX["last_init"], np.float32)
    finally:
        for k in [k for k in sys.modules if k not in saved_modules]:
            del sys.modules[k]
        sys.modules.update(saved_modules)
        sys.path[:] = saved_path
        sys.dont_write_bytecode = saved_bc
    assert os.path.basename(path) == f"golden_{env_name}_N{N}_H{H}.npz"
    g = np.load(path)
    m = load_model(env_name)
    Nu = m.act_size()
    # the schema tests/test_golden.py and tools/compare_golden.py read
    for k in ("jax_version", "brax_version", "threefry_partitionable", "alphas_bar", "sigmas", "q0", "qd0", "x0_pos", "x0_rot"):
        assert k in g, k
    assert g["alphas_bar"].shape == (100,) and abs(float(g["alphas_bar"][-1]) - 0.6024805) < 1e-6   # (SURVEY §8 A0's KAT)
    for k in range(2):
        assert g[f"key_{k}"].shape == (2,) and g[f"eps_{k}"].shape == (N, H, Nu) and g[f"Y0s_{k}"].shape == (N, H, Nu)
        assert g[f"rewss_{k}"].shape == (N, H) and g[f"weights_{k}"].shape == (N,) and g[f"Ybar_{k}"].shape == (H, Nu)
        assert g[f"xpos_{k}"].shape[:2] == (N, H) and g[f"xpos_{k}"].shape[-1] == 3
        assert abs(float(g[f"weights_{k}"].sum()) - 1.0) < 1e-5
        # what the consumer does with it: the checker, teacher-forced with the file's candidates, reproduces the file's rewards
        # (here they ARE the checker's — through the reference's wrapper code — so this closes the loop, it proves no physics)
        rew = orc.rollout(m.to_struct(), state_init, np.asarray(g[f"Y0s_{k}"], np.float32))
        assert np.allclose(rew, g[f"rewss_{k}"], rtol=1e-5, atol=1e-6), k
    assert not any(k.startswith("stage_") for k in g.files)   # (records (B), (C) need Brax's internals: not under the stand-in)
