"""Per-environment constants taken from the reference's env wrappers.

Data-level guesses live here as keys a golden vector can flip, read by tools/compile_models.py:
  collide_all_capsules (default False)   hopper / walker2d / halfcheetah: which geoms meet the floor.  Brax loads these three
      files from inside its wheel; the re-authored ones give only the FEET a contype (hopper.xml: foot_geom), so a body that
      tips over sinks its torso through the floor and keeps collecting forward reward (tools/model_guess_report.py: how often).
      True makes every capsule of every link collide (ends as spheres); hopper / walker2d stay on the planar kernels (two
      spheres per link), the halfcheetah's torso then carries four (the planar kernels' four-collider instantiations).
"""

SPECS = {
    # mbd/envs/humanoidrun.py:14-17,21-27
    "humanoidrun": dict(xml="humanoidrun.xml", from_reference=True, n_frames=7, reset_noise=0.01),
    # mbd/envs/humanoidtrack.py:15-46 (reset is deterministic, :48-61); the five *_ref marker links are
    # world-parented, collision-free and dynamically decoupled from the humanoid (humanoidtrack.xml:152-180),
    # and only overwritten for visualisation (:67-74): the compiled model drops them.
    "humanoidtrack": dict(xml="humanoidtrack.xml", from_reference=True, n_frames=5, reset_noise=0.0,
                          drop_suffix="_ref",
                          track=("torso", "left_thigh", "right_thigh", "left_shin", "right_shin")),
    # mbd/envs/hopper.py:12-18 (_reset_noise_scale 5e-3, n_frames 20); XML re-authored (see assets/hopper.xml)
    "hopper": dict(xml="hopper.xml", from_reference=False, n_frames=20, reset_noise=5e-3, reward_params=(1.0, 0.5)),
    # mbd/envs/walker2d.py:11-18,57-62 (same shape as the hopper; XML re-authored, see assets/walker2d.xml)
    "walker2d": dict(xml="walker2d.xml", from_reference=False, n_frames=20, reset_noise=5e-3,
                     reward_params=(1.1, 0.5)),
    # mbd/envs/humanoidstandup.py:14-27,50-56: the humanoid lying on the floor, 15 sphere colliders
    "humanoidstandup": dict(xml="humanoidstandup.xml", from_reference=True, n_frames=7, reset_noise=0.01),
    # mbd/envs/cartpole.py:11-37,45: the reference's own 2-link model; positional backend -> dt 0.005, n_frames 4;
    # reset adds [0, pi] to q (pole hanging down)
    "cartpole": dict(xml="cartpole.xml", from_reference=True, n_frames=4, reset_noise=0.01, dt_override=0.005,
                     init_q_offset=(0.0, 3.141592653589793)),
    # brax.envs.ant (absent; selected by name at mbd/envs/__init__.py:30-31 and the DEFAULT env_name of Args,
    # mbd_planner.py:25): positional backend dt 0.005, n_frames 10, reset noise 0.1 (uniform q, normal qd),
    # reward = forward velocity + healthy_reward - 0.5 |a|^2.  With the stock terminate_when_unhealthy=True the
    # healthy term is the CONSTANT healthy_reward (the z-range only sets `done`, which the planner never reads):
    # reward_params[5] = 1 selects that; 0 gives healthy_reward * (0.2 <= z <= 1.0).  The env class also replaces
    # every actuator gear by 200 on the positional backend.  Recollection of brax.envs.ant, unpinned.
    "ant": dict(xml="ant.xml", from_reference=False, n_frames=10, reset_noise=0.1,
                reward_params=(1.0, 0.5, 0.2, 1.0, 1.0, 1.0), gear_override=(200.0,) * 8),
    # brax.envs.half_cheetah (absent): n_frames 16 @ 0.003125 s, reset noise 0.1, forward_reward_weight 1,
    # ctrl_cost_weight 0.1; spring/positional backends replace the gears by [120, 90, 60, 120, 100, 100]
    #  — recollection, unpinned
    "halfcheetah": dict(xml="halfcheetah.xml", from_reference=False, n_frames=16, reset_noise=0.1,
                        reward_params=(1.0, 0.1), gear_override=(120.0, 90.0, 60.0, 120.0, 100.0, 100.0)),
}

# names mbd.envs.get_env knows (mbd/envs/__init__.py:13-33) that are outside the hot-path scope
OUT_OF_SCOPE = ("pushT",)
