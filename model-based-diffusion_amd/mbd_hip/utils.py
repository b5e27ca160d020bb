"""Drop-in for mbd/utils.py: ``rollout_us`` (utils.py:14-20) and ``eval_us`` (:6-12) over the GPU rollout."""
from __future__ import annotations

import numpy as np


def rollout_us(env, state, us):
    """Roll ONE action sequence ``us`` [H,Nu] out from ``state``.  Returns (rews [H], xpos) like the
    reference's ``(rews, pipline_states)``: of the pipeline states only the tracked link positions
    ([H,K,3]; car2d: q [H,3]) are materialised — the part eval_xref_logpd consumes.

    The reference's first argument is the jitted ``step_env``; here it is the env object (the step
    function is the HIP kernel)."""
    us = np.ascontiguousarray(us, np.float32)
    want = getattr(env, "xref", None) is not None
    out = env.rollout(state, us[None], want_xpos=want)
    if want:
        return out[0][0].cpu().numpy(), out[1][0].cpu().numpy()
    return out[0].cpu().numpy(), None


def eval_us(env, state, us):
    return rollout_us(env, state, us)[0]
