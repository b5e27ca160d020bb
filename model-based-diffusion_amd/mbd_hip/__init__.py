"""mbd_hip — MI355X-native drop-in for the reverse-diffusion hot path of LeCAR-Lab/model-based-diffusion.

Mirrors the reference package layout for the path (``mbd.envs.get_env``, ``mbd.utils.rollout_us``,
``mbd.planners.mbd_planner.{Args, run_diffusion}``); all compute runs in libmbd_hip.so (HIP, gfx950).
"""
from . import _capi, envs, model, utils  # noqa: F401
from . import planners  # noqa: F401
from . import scripts  # noqa: F401

__all__ = ["envs", "utils", "planners", "model", "_capi"]
