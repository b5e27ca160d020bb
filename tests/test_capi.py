"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/mbd_hip.h declares,
agrees with the oracle on the struct layout and the host-side PRNG, and fails LOUDLY without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_model


def _declared():
    text = open(os.path.join(ROOT, "include", "mbd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mbd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mbd_hip.h but not exported"
    from mbd_hip import _capi
    assert sorted(_capi.EXPORTS) == names


def test_struct_layout_agrees_between_ctypes_header_and_oracle(lib, orc):
    from mbd_hip.model import MbdModel
    assert C.sizeof(MbdModel) == orc.lib.orc_model_bytes()
    ms = load_model("humanoidrun").to_struct()
    # the oracle reads the same bytes: forward kinematics sees 11 links at the expected places
    st = orc.forward(ms, np.asarray(ms.init_q[:24], np.float32), np.zeros(23, np.float32))
    assert st.shape == (11, 13) and abs(orc.link_positions(ms, st)[0, 2] - 1.4) < 1e-6


def test_host_prng_matches_oracle(lib, orc):
    from mbd_hip import _capi
    for seed in (0, 1, 12345, (3 << 32) | 9):
        k = _capi.prng_key(seed)
        assert np.array_equal(k, orc.prng_key(seed))
        for impl in (0, 1):
            for num in (2, 3, 5):
                assert np.array_equal(_capi.prng_split(k, num, impl), orc.split(k, num, impl))


def test_host_prng_published_jax_key_splits(lib):
    """mbd_prng_key / mbd_prng_split against key values printed in JAX's own documentation (see
    tests/test_oracle_prng.py for the sources): legacy layout PRNGKey(0), partitionable layout key(42)."""
    from mbd_hip import _capi
    assert _capi.prng_split(_capi.prng_key(0), 2, 0).tolist() == [[4146024105, 967050713],
                                                                  [2718843009, 1272950319]]
    assert _capi.prng_split(_capi.prng_key(42), 2, 1)[0].tolist() == [1832780943, 270669613]


def test_args_mirror_the_reference_dataclass():
    from mbd_hip.planners.mbd_planner import Args, apply_recommended
    a = Args()
    assert (a.seed, a.env_name, a.Nsample, a.Hsample, a.Ndiffuse) == (0, "ant", 2048, 50, 100)
    assert (a.temp_sample, a.beta0, a.betaT, a.enable_demo, a.not_render) == (0.1, 1e-4, 1e-2, False, False)
    a = Args(env_name="humanoidrun")
    apply_recommended(a)   # mbd_planner.py:54-68: silently N=8192, Ndiffuse=300 unless disabled
    assert (a.Nsample, a.Ndiffuse, a.temp_sample) == (8192, 300, 0.1)
    b = Args(env_name="halfcheetah", temp_sample=0.3)
    apply_recommended(b)
    assert b.temp_sample == 0.4
    c = Args(env_name="humanoidrun", disable_recommended_params=True, Nsample=1024)
    apply_recommended(c)
    assert c.Nsample == 1024


def test_get_env_errors_like_the_reference(lib):
    from mbd_hip import _capi
    from mbd_hip.envs import get_env
    with pytest.raises(ValueError, match="Unknown environment"):
        get_env("no_such_env")
    with pytest.raises(ValueError):
        get_env("pushT")  # in the reference registry (generalized backend), outside the hot-path scope
    if _capi.device_count() == 0:
        # no GPU here: the product must fail loudly, never fall back to a CPU path
        with pytest.raises(_capi.MbdError) as e:
            get_env("humanoidrun")
        assert e.value.code == _capi.MBD_ERR_NO_DEVICE
        with pytest.raises(_capi.MbdError):
            get_env("car2d")


def test_builtin_models_are_the_compiled_assets(lib):
    """mbd_env_create(name) needs no Python: the library embeds every compiled model (tools/gen_models_inc.py).
    Host-only accessors: the names it knows and, bit for bit, the models it would create."""
    from mbd_hip.envs import specs
    from mbd_hip.model import MbdModel
    names, k = [], 0
    while True:
        n = lib.mbd_env_name(k)
        if n is None:
            break
        names.append(n.decode())
        k += 1
    assert names[0] == "car2d" and sorted(names[1:]) == sorted(specs.SPECS)
    for name in specs.SPECS:
        st = MbdModel()
        assert lib.mbd_builtin_model(name.encode(), C.byref(st)) == 0
        assert bytes(st) == bytes(load_model(name).to_struct()), name
    st = MbdModel()
    assert lib.mbd_builtin_model(b"pushT", C.byref(st)) == -2


def test_create_by_name_errors(lib):
    """Unknown / out-of-scope names are rejected before a device is looked for (-> ValueError in the shim, like
    mbd/envs/__init__.py:33); known names fail LOUDLY without a device."""
    from mbd_hip import _capi
    lib.mbd_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert lib.mbd_env_create(b"no_such_env", 0, C.byref(h)) == _capi.MBD_ERR_UNSUPPORTED
    assert b"Unknown environment: no_such_env" in lib.mbd_last_error()
    assert lib.mbd_env_create(b"pushT", 0, C.byref(h)) == _capi.MBD_ERR_UNSUPPORTED
    assert lib.mbd_env_create(None, 0, C.byref(h)) == _capi.MBD_ERR_INVALID
    if _capi.device_count() == 0:
        for name in (b"humanoidrun", b"car2d", b"hopper"):
            assert lib.mbd_env_create(name, 0, C.byref(h)) == _capi.MBD_ERR_NO_DEVICE


def test_model_from_struct_round_trip():
    from mbd_hip.model import Model
    for name in ("humanoidtrack", "hopper", "ant"):
        m = load_model(name)
        m2 = Model.from_struct(m.to_struct(), m.link_names, m.actuator_names, name)
        assert bytes(m2.to_struct()) == bytes(m.to_struct())
        assert np.array_equal(m2.init_q, m.init_q) and len(m2.fields["track_link"]) == m.fields["n_track"]


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "model-based-diffusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("checker", ""), f"{f} mentions the oracle"


def test_dpp_lane_layouts_of_the_builtin_models(lib):
    """Host logic of the DPP exchange (csrc/mbd_env.hip::find_dpp_layout): every built-in tree gets a layout in
    which each link sits on its own lane of the LPS-lane group and every s-th child (in link order) sits exactly
    shift[s] lanes below its parent — what the kernels' row shifts assume."""
    lib.mbd_debug_dpp_layout.restype = C.c_int
    want_family = {"humanoidrun": 0, "humanoidtrack": 0, "humanoidstandup": 0, "walker2d": 1, "halfcheetah": 1,
                   "hopper": 2, "cartpole": 2, "ant": 3}
    for name, fam in want_family.items():
        m = load_model(name)
        ms = m.to_struct()
        tab = (C.c_byte * 32)()
        shifts = (C.c_int * 4)()
        assert lib.mbd_debug_dpp_layout(C.byref(ms), tab, shifts) == fam, name
        L, parent = m.fields["n_links"], [int(x) for x in m.fields["parent"]]
        lps = 4 if L <= 4 else (8 if L <= 8 else 16)
        lane = [tab[16 + l] for l in range(L)]
        assert sorted(set(lane)) == sorted(lane) and min(lane) >= 0 and max(lane) < lps, name
        assert all(tab[lane[l]] == l for l in range(L))
        assert sum(1 for i in range(16) if tab[i] >= 0) == L
        for l in range(1, L):
            slot = sum(1 for c in range(l) if parent[c] == parent[l])
            assert shifts[slot] != 0 and lane[parent[l]] == lane[l] + shifts[slot], (name, l)


def test_c_abi_argument_errors_without_a_device(lib):
    """Every export returns an int status and leaves a message in mbd_last_error(); NULL / out-of-range arguments
    are rejected before anything touches a device (so this runs on a CPU-only box), destroying NULL is fine."""
    from mbd_hip import _capi
    lib.mbd_last_error.restype = C.c_char_p
    INVALID = _capi.MBD_ERR_INVALID
    assert lib.mbd_prng_split(None, 2, 1, None) == INVALID and b"prng_split" in lib.mbd_last_error()
    assert lib.mbd_env_info(None, None, None, None, None, None, None) == INVALID
    assert lib.mbd_device_count(None) == INVALID
    assert lib.mbd_plan_create(None, None, None) == INVALID
    assert lib.mbd_plan_schedule(None, None, None, None) == INVALID
    assert lib.mbd_plan_set_state0(None, None) == INVALID
    assert lib.mbd_plan_run(None, None, None, None, None, None) == INVALID
    assert lib.mbd_plan_sample_rollout(None, 1, None, None, None, None, None) == INVALID
    assert lib.mbd_plan_score_update(None, 1, None, None, None, None, None, None, None) == INVALID
    assert lib.mbd_env_create_model(None, 0, None, None, C.c_float(0), None) == INVALID
    assert lib.mbd_env_xref_logpd(None, None, 1, 50, None, None) == INVALID
    assert lib.mbd_env_observe(None, None, None) == INVALID and lib.mbd_model_observe(None, None, None, None, None) == INVALID
    assert lib.mbd_env_get_model(None, None) == INVALID and lib.mbd_env_xref(None, None, 0, None) == INVALID
    assert lib.mbd_env_destroy(None) == 0 and lib.mbd_plan_destroy(None) == 0
    n = C.c_int(-1)
    assert lib.mbd_device_count(C.byref(n)) == 0 and n.value >= 0
    assert lib.mbd_version() >= 1


def test_round3_entry_points_fail_loudly_without_a_device_and_levers_work_on_the_host(lib):
    """The in-library exchange and the sweeps have no CPU path either: without a GPU their constructors return
    MBD_ERR_NO_DEVICE (argument errors first).  The debug levers are host state: the table is seeded from the
    environment once, mbd_debug_set / get work without a device, unknown names are rejected."""
    from mbd_hip import _capi
    h = C.c_void_p()
    assert lib.mbd_exchange_create(0, 0, 1, 1, 8, None) == _capi.MBD_ERR_INVALID
    assert lib.mbd_sweep_create(None, None, 2, None, C.byref(h)) == _capi.MBD_ERR_INVALID
    if _capi.device_count() == 0:
        assert lib.mbd_exchange_create(0, 0, 2, 1, 8, C.byref(h)) == _capi.MBD_ERR_NO_DEVICE
        assert b"no CPU fallback" in lib.mbd_last_error()
    assert set(_capi.LEVERS) >= {"MBD_PK2", "MBD_NO_DPP", "MBD_NO_LAZY"}
    for name in _capi.LEVERS:
        before = _capi.debug_get(name)
        _capi.debug_set(name, 1)
        assert _capi.debug_get(name) == 1
        _capi.debug_set(name, before)
    with pytest.raises(_capi.MbdError):
        _capi.debug_set("MBD_NO_SUCH_LEVER", 1)
    # the debug header declares exactly what the library exports beside the product ABI
    import os, re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "include", "mbd_hip_debug.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for n in sorted(set(re.findall(r"\b(mbd_debug_[a-z0-9_]+)\s*\(", text))):
        assert hasattr(lib, n), n
