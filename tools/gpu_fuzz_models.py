"""Fuzz beyond the committed seeds: random MJCF models (tests/random_models.py) through the general kernels against the checker,
rollouts bit for bit.  usage (GPU box): python tools/gpu_fuzz_models.py FIRST COUNT [planar | planar3d | spec | planarspec] [--bits W]
--bits W: every model carries the specification word W (under MBD_HIP_LIB=<a tuned variant of word W> the models then run that
build's TUNED general instantiations: the round-6 check of MBD_TUNED_SPEC beyond the built-in models)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from random_models import stable_random_model
from test_random_models import _comp
from mbd_hip import _capi
from mbd_hip.envs.base import RigidBodyEnv
from oracle import oracle as orc_mod
orc_mod.build()
orc = orc_mod.Oracle("f32")
first, count = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "3d"
forced_bits = int(sys.argv[sys.argv.index("--bits") + 1]) if "--bits" in sys.argv else None
bad = refused = touched = blown = ranaway = 0
ranaway_seeds = []
blown_seeds, blown_default = [], []
kinds = {}
for seed in range(first, first + count):
    bits = 4  # (model.DEFAULT_SPEC: the word the shipped library's tuned and general non-SPEC instantiations compile in)
    if mode.endswith("spec"):  # a random subset of the specification switches (include/mbd_hip.h mbd_model_flags)
        bits = int(np.random.default_rng(1000 + seed).integers(1, 64)) * 4
    if forced_bits is not None:
        bits = forced_bits
    if mode in ("3d", "spec"):
        _, m = stable_random_model(seed, lambda x: _comp(x, spec_flags=bits))
    elif mode == "planarspec":
        _, m = stable_random_model(seed, lambda x: _comp(x, env_name="halfcheetah", spec_flags=bits & (4 | 8 | 16 | 32)), planar=True, max_bodies=10)
    else:
        _, m = stable_random_model(seed, lambda x: _comp(x, env_name="halfcheetah", planar=None if mode == "planar" else False,
                                                         spec_flags=bits & (4 | 8 | 16 | 32)), planar=True, max_bodies=10)
    try:
        env = RigidBodyEnv("hopper" if mode in ("3d", "spec") else "halfcheetah", model=m)
    except Exception as e:  # a shape the library refuses (says so)
        refused += 1
        print(seed, "refused:", str(e)[:120])
        continue
    st = env.reset(_capi.prng_key(seed))
    us = np.clip(np.random.default_rng(seed).normal(size=(21, 90, env.action_size)) * 0.6, -1.3, 1.3).astype(np.float32)
    got = env.rollout(st, us).cpu().numpy()
    ref = orc.rollout(m.to_struct(), np.asarray(st.pipeline_state, np.float32), us)
    touched += int(np.abs(np.diff(ref, axis=1)).max() > 0.02)  # (a jump of the per-step reward: an impact)
    # (a model that the switches drive unstable — seed 151 of `spec`: friction as a velocity bound on links with two colliders,
    # Jacobi-summed since round 5 — blows up in the checker and in the kernel alike: NaN at the same places counts as equal)
    if not np.isfinite(ref).all():
        blown += 1
        blown_seeds.append((seed, bits))
        if bits == (4 if forced_bits is None else forced_bits):  # the DEFAULT specification (or the forced word) on a model the generator called stable: an instability of the default, not of a switch
            blown_default.append(seed)
    # RUNAWAYS: a model whose motion leaves the numerical contract's ranges (the kernels' short division / square-root
    # sequences are the correctly rounded results for magnitudes up to ~1e8 — a free slide under a strong motor passes 1e5 m
    # within the horizon, seed 3255) is compared up to the control step at which the CHECKER's reward first exceeds 1e4 in
    # magnitude; beyond it kernel and checker may differ (NaN against a large finite number).  Counted and listed.
    big = ~(np.abs(ref) <= 1e4)
    if big.any():
        ranaway += 1
        ranaway_seeds.append(seed)
        t_big = np.where(big.any(axis=1), big.argmax(axis=1), ref.shape[1])
        keep = np.arange(ref.shape[1])[None, :] < t_big[:, None]
        got, ref = np.where(keep, got, 0.0), np.where(keep, ref, 0.0)
    if not np.array_equal(got, ref, equal_nan=True):
        bad += 1
        print(seed, "MISMATCH max|d|", np.nanmax(np.abs(got - ref)), "links", m.n_links, "non-finite:", int((~np.isfinite(got)).sum()), int((~np.isfinite(ref)).sum()))
print(f"{mode} seeds {first}..{first + count - 1}: {bad} mismatches, {refused} refused, {touched} with an impact in the horizon, "
      f"{blown} non-finite in checker and kernel alike" + (f": (seed, flags) {blown_seeds}" if blown_seeds else "") +
      (f", {ranaway} ran away beyond |reward| 1e4 (compared up to there): seeds {ranaway_seeds}" if ranaway else ""))
# a NaN that checker and kernel share still counts as equal (the comparison is about the kernels) — but it is REPORTED, seed by
# seed, and a model WITHOUT switches that blows up fails the run: the default specification must hold every model the generator
# calls stable (round-5 advice: "0 mismatches" must not hide instabilities of a new default)
if bad or blown_default:
    print(f"FAIL: {bad} mismatches, default-specification blow-ups at seeds {blown_default}")
    sys.exit(1)
