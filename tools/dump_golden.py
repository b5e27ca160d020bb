"""Golden-vector escape hatch: run THIS under a real `pip install jax brax` next to the
reference checkout to dump what the reference itself computes; commit the .npz under tests/golden/ and
tests/test_golden.py will hold this repo to it.  It only *consumes* the reference (imports `mbd`).

    python tools/dump_golden.py /path/to/model-based-diffusion humanoidrun 64 50 3

Nothing here runs in the build container (jax/brax are absent) and no golden is ever fabricated.
"""
import sys

import numpy as np


def main():
    ref, env_name, N, H, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    sys.path.insert(0, ref)
    import functools
    import brax
    import jax
    from jax import numpy as jnp
    import mbd

    env = mbd.envs.get_env(env_name)
    Nu = env.action_size
    step_env = jax.jit(env.step)
    rollout_us = jax.jit(functools.partial(mbd.utils.rollout_us, step_env))
    rng = jax.random.PRNGKey(0)
    rng, rng_reset = jax.random.split(rng)
    state_init = jax.jit(env.reset)(rng_reset)
    Nd = 100
    betas = jnp.linspace(1e-4, 1e-2, Nd)
    alphas = 1.0 - betas
    alphas_bar = jnp.cumprod(alphas)
    sigmas = jnp.sqrt(1 - alphas_bar)
    rng_exp, rng = jax.random.split(rng)
    out = dict(jax_version=jax.__version__, brax_version=brax.__version__,
               threefry_partitionable=bool(jax.config.jax_threefry_partitionable),
               alphas_bar=np.asarray(alphas_bar), sigmas=np.asarray(sigmas),
               q0=np.asarray(state_init.pipeline_state.q), qd0=np.asarray(state_init.pipeline_state.qd),
               x0_pos=np.asarray(state_init.pipeline_state.x.pos), x0_rot=np.asarray(state_init.pipeline_state.x.rot))
    Ybar = jnp.zeros([H, Nu])
    r = rng_exp
    for k, i in enumerate(range(Nd - 1, Nd - 1 - steps, -1)):
        r, ks = jax.random.split(r)
        eps = jax.random.normal(ks, (N, H, Nu))
        Y0s = jnp.clip(eps * sigmas[i] + Ybar, -1.0, 1.0)
        rewss, qs = jax.vmap(rollout_us, in_axes=(None, 0))(state_init, Y0s)
        rews = rewss.mean(axis=-1)
        std = jnp.where(rews.std() < 1e-4, 1.0, rews.std())
        w = jax.nn.softmax((rews - rews.mean()) / std / 0.1)
        Ybar = jnp.einsum("n,nij->ij", w, Y0s)
        out.update({f"key_{k}": np.asarray(ks), f"eps_{k}": np.asarray(eps), f"Y0s_{k}": np.asarray(Y0s),
                    f"rewss_{k}": np.asarray(rewss), f"weights_{k}": np.asarray(w), f"Ybar_{k}": np.asarray(Ybar),
                    f"xpos_{k}": np.asarray(qs.x.pos)})
    np.savez_compressed(f"golden_{env_name}_N{N}_H{H}.npz", **out)
    print("wrote", f"golden_{env_name}_N{N}_H{H}.npz")


if __name__ == "__main__":
    main()
