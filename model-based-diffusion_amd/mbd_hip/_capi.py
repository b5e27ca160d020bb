"""ctypes binding of libmbd_hip.so (include/mbd_hip.h) — the stub INTEGRATION.md shows a maintainer.

The library is the product: if it is missing, or no gfx950 device is visible, calls fail loudly
(``MbdError``); there is no Python/CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .model import MbdModel

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "lib", "libmbd_hip.so")

MBD_OK = 0
MBD_ERR_INVALID, MBD_ERR_UNSUPPORTED, MBD_ERR_HIP, MBD_ERR_NO_DEVICE, MBD_ERR_STATE = -1, -2, -3, -4, -5
PRNG_LEGACY, PRNG_PARTITIONABLE = 0, 1


class MbdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmbd_hip error {code}: {msg}")
        self.code = code


class PlanConfig(C.Structure):
    _fields_ = [("Nsample", C.c_int32), ("Hsample", C.c_int32), ("Ndiffuse", C.c_int32),
                ("temp_sample", C.c_float), ("beta0", C.c_float), ("betaT", C.c_float),
                ("enable_demo", C.c_int32), ("prng_impl", C.c_int32), ("shard_begin", C.c_int32),
                ("shard_count", C.c_int32), ("literal_score", C.c_int32), ("update_method", C.c_int32),
                ("shares_device", C.c_int32), ("reserved", C.c_int32 * 3)]


EXPORTS = [
    "mbd_last_error", "mbd_version", "mbd_tuned_spec", "mbd_device_count", "mbd_prng_key", "mbd_prng_split",
    "mbd_env_create", "mbd_env_name", "mbd_builtin_model", "mbd_env_get_model", "mbd_env_xref", "mbd_env_xref_logpd",
    "mbd_env_observe", "mbd_model_observe", "mbd_model_forward", "mbd_env_create_car2d", "mbd_env_create_model", "mbd_env_destroy", "mbd_env_info", "mbd_env_reset", "mbd_env_pipeline_init",
    "mbd_env_step", "mbd_env_rew_xref", "mbd_env_rollout", "mbd_plan_create", "mbd_plan_destroy",
    "mbd_plan_schedule", "mbd_plan_set_state0", "mbd_plan_sample_rollout", "mbd_plan_prefetch_noise", "mbd_plan_score_update",
    "mbd_plan_set_sigma", "mbd_plan_get_sigma", "mbd_plan_reverse_once", "mbd_plan_run", "mbd_plan_eval", "mbd_plan_peek", "mbd_plan_kernel_time",
    "mbd_plan_enable_timing",
    "mbd_sweep_create", "mbd_sweep_destroy", "mbd_sweep_set_state0", "mbd_sweep_run", "mbd_sweep_kernel_time", "mbd_sweep_get_sigmas",
    "mbd_exchange_create", "mbd_exchange_destroy", "mbd_exchange_local_handle", "mbd_exchange_connect",
    "mbd_exchange_all_gather", "mbd_exchange_status", "mbd_exchange_fine_grained",
]

_lib = None
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_u32p = C.POINTER(C.c_uint32)
_fp = C.POINTER(C.c_float)


def load() -> C.CDLL:
    """dlopen the in-tree library. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(os.environ.get("MBD_HIP_LIB") or LIB_PATH)  # override: kernel A/B builds only
    if not os.path.exists(path):
        raise MbdError(MBD_ERR_STATE, f"{path} is missing: run `python __graft_entry__.py` (build()) first; "
                                      "mbd_hip has no CPU fallback")
    # torch (device memory / streams / torch.distributed plumbing) bundles its own HIP runtime with the
    # same SONAME as the system one: it has to be the first one mapped, or two runtimes fight over the
    # device ("No HIP GPUs are available")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.mbd_last_error.restype = C.c_char_p
    lib.mbd_device_count.argtypes = [C.POINTER(_i)]
    lib.mbd_prng_key.argtypes = [C.c_uint64, _u32p]
    lib.mbd_prng_split.argtypes = [_u32p, _i, _i, _u32p]
    lib.mbd_env_create.argtypes = [C.c_char_p, _i, C.POINTER(_vp)]
    lib.mbd_env_name.argtypes = [_i]
    lib.mbd_env_name.restype = C.c_char_p
    lib.mbd_builtin_model.argtypes = [C.c_char_p, C.POINTER(MbdModel)]
    lib.mbd_env_get_model.argtypes = [_vp, C.POINTER(MbdModel)]
    lib.mbd_env_xref.argtypes = [_vp, _vp, _i, C.POINTER(_i)]
    lib.mbd_env_xref_logpd.argtypes = [_vp, _vp, _i, _i, _vp, _vp]
    lib.mbd_env_observe.argtypes = [_vp, _vp, _vp]
    lib.mbd_model_observe.argtypes = [C.POINTER(MbdModel), _vp, _vp, _vp, _vp]
    lib.mbd_env_create_car2d.argtypes = [_i, _vp, C.POINTER(_vp)]
    lib.mbd_env_create_model.argtypes = [C.c_char_p, _i, C.POINTER(MbdModel), _vp, _f, C.POINTER(_vp)]
    lib.mbd_env_destroy.argtypes = [_vp]
    lib.mbd_env_info.argtypes = [_vp] + [C.POINTER(_i)] * 5 + [_fp]
    lib.mbd_env_reset.argtypes = [_vp, _u32p, _i, _vp]
    lib.mbd_model_forward.argtypes = [_vp, _vp, _vp, _vp]
    lib.mbd_env_pipeline_init.argtypes = [_vp, _vp, _i, _vp, _i, _vp]
    lib.mbd_env_step.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
    lib.mbd_env_rew_xref.argtypes = [_vp, _fp]
    lib.mbd_env_rollout.argtypes = [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]
    lib.mbd_plan_create.argtypes = [_vp, C.POINTER(PlanConfig), C.POINTER(_vp)]
    lib.mbd_plan_destroy.argtypes = [_vp]
    lib.mbd_plan_schedule.argtypes = [_vp, _vp, _vp, _vp]
    lib.mbd_plan_set_state0.argtypes = [_vp, _vp]
    lib.mbd_plan_sample_rollout.argtypes = [_vp, _i, _u32p, _vp, _vp, _vp, _vp]
    lib.mbd_plan_prefetch_noise.argtypes = [_vp, _u32p, _vp]
    lib.mbd_plan_score_update.argtypes = [_vp, _i, _u32p, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.mbd_plan_set_sigma.argtypes = [_vp, _f]
    lib.mbd_plan_get_sigma.argtypes = [_vp, _fp]
    lib.mbd_plan_reverse_once.argtypes = [_vp, _i, _u32p, _vp, _vp, _vp]
    lib.mbd_plan_run.argtypes = [_vp, _u32p, _vp, _vp, _fp, C.POINTER(C.c_double)]
    lib.mbd_plan_eval.argtypes = [_vp, _vp, _fp]
    lib.mbd_plan_peek.argtypes = [_vp, _vp, _vp, _vp]
    lib.mbd_plan_kernel_time.argtypes = [_vp, _fp, C.POINTER(_i), _i]
    lib.mbd_plan_enable_timing.argtypes = [_vp, _i]
    lib.mbd_sweep_create.argtypes = [_vp, C.POINTER(PlanConfig), _i, _vp, C.POINTER(_vp)]
    lib.mbd_sweep_destroy.argtypes = [_vp]
    lib.mbd_sweep_set_state0.argtypes = [_vp, _i, _vp]
    lib.mbd_sweep_run.argtypes = [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)]
    lib.mbd_sweep_kernel_time.argtypes = [_vp, _i, _fp, C.POINTER(_i)]
    lib.mbd_sweep_get_sigmas.argtypes = [_vp, _vp]
    lib.mbd_exchange_create.argtypes = [_i, _i, _i, _i, _i, C.POINTER(_vp)]
    lib.mbd_exchange_destroy.argtypes = [_vp]
    lib.mbd_exchange_local_handle.argtypes = [_vp, _vp]
    lib.mbd_exchange_connect.argtypes = [_vp, _vp]
    lib.mbd_exchange_all_gather.argtypes = [_vp, _vp, C.POINTER(_vp), _vp]
    lib.mbd_exchange_status.argtypes = [_vp]
    lib.mbd_exchange_fine_grained.argtypes = [_vp, C.POINTER(_i)]
    _lib = lib
    return lib


LEVERS = ("MBD_NO_DPP", "MBD_NO_NFR_CONST", "MBD_NO_REWARD_CONST", "MBD_NO_PLANAR_FLAGS", "MBD_NO_FAST_SLIDES",
          "MBD_NO_FUSED_NOISE", "MBD_NO_LAZY", "MBD_NO_PREFETCH", "MBD_NO_AUX", "MBD_WMEAN_SPLIT", "MBD_NO_FUSED_SCORE",
          "MBD_PK2", "MBD_WPB", "MBD_LDS_RESERVE", "MBD_NO_HELPERS")


def debug_set(name: str, value: int) -> None:
    """include/mbd_hip_debug.h: a test / A-B lever of the library (-1: not set).  The library reads the environment
    variables of the same names once, when it is first used; afterwards only this call changes a lever."""
    lib = load()
    lib.mbd_debug_set.argtypes = [C.c_char_p, _i]
    check(lib.mbd_debug_set(name.encode(), int(value)))


def debug_get(name: str) -> int:
    lib = load()
    lib.mbd_debug_get.argtypes = [C.c_char_p, C.POINTER(_i)]
    v = _i(0)
    check(lib.mbd_debug_get(name.encode(), C.byref(v)))
    return v.value


def check(rc: int) -> None:
    if rc != MBD_OK:
        raise MbdError(rc, load().mbd_last_error().decode())


def device_count() -> int:
    n = _i(0)
    check(load().mbd_device_count(C.byref(n)))
    return n.value


def key_array(key) -> "C.Array":
    k = np.ascontiguousarray(key, np.uint32).reshape(2)
    return (C.c_uint32 * 2)(int(k[0]), int(k[1]))


def prng_key(seed: int) -> np.ndarray:
    out = (C.c_uint32 * 2)()
    check(load().mbd_prng_key(int(seed), out))
    return np.array([out[0], out[1]], np.uint32)


def prng_split(key, num: int = 2, impl: int = PRNG_PARTITIONABLE) -> np.ndarray:
    out = (C.c_uint32 * (2 * num))()
    check(load().mbd_prng_split(key_array(key), num, impl, out))
    return np.array(list(out), np.uint32).reshape(num, 2)


def np_ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)
