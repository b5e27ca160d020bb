"""Env registry — drop-in for ``mbd.envs.get_env`` (mbd/envs/__init__.py:13-33) on the hot path."""
from __future__ import annotations

from . import specs
from .base import Car2d, RigidBodyEnv, State

__all__ = ["get_env", "State", "Car2d", "RigidBodyEnv"]


def get_env(env_name: str, device: int = 0):
    """Same contract as the reference: a string in, an env object out, ``ValueError`` on an unknown
    name.  Every name the reference's registry knows resolves — car2d, humanoidrun, humanoidtrack, humanoidstandup, hopper,
    halfcheetah, walker2d, cartpole, ant — except pushT (Brax's generalized backend: another physics engine)."""
    if env_name == "car2d":
        return Car2d(device=device)
    if env_name in specs.SPECS:
        return RigidBodyEnv(env_name, device=device)
    if env_name in specs.OUT_OF_SCOPE:
        raise ValueError(f"Environment {env_name!r} is known to the reference but outside the MI355X "
                         f"hot-path scope of mbd_hip (SURVEY.md §2)")
    raise ValueError(f"Unknown environment: {env_name}")
