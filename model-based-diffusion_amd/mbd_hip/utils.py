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


def rollout_states(env, state, us):
    """The state after every control step of ONE action sequence (what the reference's render_us collects
    for the HTML viewer, utils.py:23-33), as plain arrays: pipeline states [H+1, ...] (incl. the initial
    one), rewards [H] and, for rigid-body envs, world link positions [H+1, L, 3] (x.pos)."""
    us = np.ascontiguousarray(us, np.float32)
    states, rews = [np.asarray(state.pipeline_state, np.float32).copy()], []
    st = state
    for t in range(us.shape[0]):
        st = env.step(st, us[t])
        states.append(np.asarray(st.pipeline_state, np.float32).copy())
        rews.append(np.float32(st.reward))
    out = dict(pipeline_states=np.stack(states), rewards=np.asarray(rews, np.float32))
    if hasattr(env, "link_positions"):
        out["link_positions"] = np.stack([env.link_positions(s) for s in states])
    return out
