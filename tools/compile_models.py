"""Compile the MJCF models and demo assets the GPU box needs into small committed data files.

/root/reference does not exist on the GPU box, so the compiled humanoid models (numbers only — what
brax.io.mjcf.load would hold in memory as `sys`) and the demo trajectories are generated HERE from the
reference's assets and committed under model-based-diffusion_amd/assets/compiled/.  No reference
source code is copied.  Usage:  python tools/compile_models.py [/root/reference]
"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
from mbd_hip import mjcf  # noqa: E402
from mbd_hip.envs import specs  # noqa: E402


class _NumpyUnpickler(pickle.Unpickler):
    """jog_xref.pkl holds pickled jax Arrays (mbd/envs/humanoidtrack.py:33-34); rebuild them as numpy."""

    def find_class(self, module, name):
        if module.startswith("jax") and name == "_reconstruct_array":
            def rebuild(fun, args, arr_state, aval_state):
                a = fun(*args)
                a.__setstate__(arr_state)
                return a
            return rebuild
        return super().find_class(module, name)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled")
    os.makedirs(out, exist_ok=True)
    own = os.path.join(ROOT, "model-based-diffusion_amd", "assets")
    for name, spec in specs.SPECS.items():
        xml = spec["xml"]
        path = os.path.join(ref, "mbd", "assets", xml) if spec["from_reference"] else os.path.join(own, xml)
        m = mjcf.load(path, env_name=name, n_frames=spec["n_frames"], drop_link_suffix=spec.get("drop_suffix"),
                      track_names=spec.get("track", ()), reset_noise=spec["reset_noise"],
                      reward_params=spec.get("reward_params", ()), dt_override=spec.get("dt_override"),
                      init_q_offset=spec.get("init_q_offset", ()),
                      gear_override=spec.get("gear_override", ()),
                      passive_joint_forces=spec.get("passive_joint_forces", True),
                      reset_quat_raw=spec.get("reset_quat_raw", False), planar=spec.get("planar"),
                      collide_all_capsules=spec.get("collide_all_capsules", False))
        with open(os.path.join(out, f"{name}.json"), "w") as f:
            f.write(m.to_json())
        print(f"{name}: L={m.n_links} nq={m.q_size()} nqd={m.qd_size()} nu={m.act_size()} "
              f"ncol={m.fields['n_col']} iso={m.fields['iso_inertia']} flags={m.fields['flags']} mass={m.masses.sum():.3f}")
    # demo trajectories
    xref = np.load(os.path.join(ref, "mbd", "assets", "car2d_xref.npy")).astype(np.float32)  # car2d.py:66
    np.save(os.path.join(out, "car2d_xref.npy"), xref)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with open(os.path.join(ref, "mbd", "assets", "jog_xref.pkl"), "rb") as f:
            d = _NumpyUnpickler(f).load()
    H = 50
    rows = []
    for body in specs.SPECS["humanoidtrack"]["track"]:  # humanoidtrack.py:36-43
        x = np.asarray(d[body], np.float32)
        if len(x) < H:
            x = np.concatenate([x, np.tile(x[-1:], (H - len(x), 1))], axis=0)
        else:
            x = x[70:H + 70]
        rows.append(x)
    jog = np.stack(rows, 0).astype(np.float32)
    np.save(os.path.join(out, "jog_xref.npy"), jog)
    print("car2d_xref", xref.shape, "jog_xref", jog.shape, jog[0, 0])


if __name__ == "__main__":
    main()
