"""Host logic: MJCF -> compiled model (mbd_hip/mjcf.py), against closed forms and SURVEY.md App. D.1."""
import math
import os

import numpy as np
import pytest

from conftest import ROOT, load_model
from test_oracle_physics import _compile

REF_XML = "/root/reference/mbd/assets/humanoidrun.xml"


def test_humanoid_topology_matches_survey_d1():
    m = load_model("humanoidrun")
    F = m.fields
    assert m.link_names == ["torso", "lwaist", "pelvis", "right_thigh", "right_shin", "left_thigh", "left_shin",
                            "right_upper_arm", "right_lower_arm", "left_upper_arm", "left_lower_arm"]
    assert F["parent"].tolist() == [-1, 0, 1, 2, 3, 2, 5, 0, 7, 0, 9]
    assert F["n_rot"].tolist() == [-1, 2, 1, 3, 1, 3, 1, 2, 1, 2, 1]
    assert (F["n_q"], F["n_qd"], F["n_act"], F["n_col"], F["n_frames"]) == (24, 23, 17, 2, 7)
    assert F["qd_idx"].tolist() == [0, 6, 8, 9, 12, 13, 16, 17, 19, 20, 22]
    # actuator k -> dof (note the abdomen_y / abdomen_z swap, humanoidrun.xml:141-142 vs :57-60)
    dof = [int(F["qd_idx"][l] + s) for l, s in zip(F["act_link"], F["act_slot"])]
    assert dof == [7, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22]
    assert np.abs(F["act_gear"]).tolist() == [350.0] * 11 + [100.0] * 6
    assert np.allclose(F["act_lo"], -0.4) and np.allclose(F["act_hi"], 0.4)
    assert abs(F["dt"] - 0.006) < 1e-9 and F["iso_inertia"] == 1
    # foot spheres fused into the shins: r = 0.075 at shin-local (0, 0, -0.35)
    assert F["col_link"].tolist() == [4, 6] and np.allclose(F["col_radius"], 0.075)
    assert np.allclose(F["col_pos"] + F["com"][[4, 6]], [[0, 0, -0.35]] * 2, atol=1e-6)
    # spring_inertia_scale = 1 -> identity inertia; spring_mass_scale = 0 -> physical masses (~40 kg)
    assert np.allclose(F["inv_inertia"][:11, :3], 1.0) and np.allclose(F["inv_inertia"][:11, 3:], 0.0)
    assert 38.0 < (1.0 / F["inv_mass"][:11]).sum() < 46.0
    assert np.allclose(F["init_q"][:7], [0, 0, 1.4, 1, 0, 0, 0])
    # knee range -160..-2 degrees about (0,-1,0)
    assert np.allclose(F["rot_lo"][4, 0], math.radians(-160)) and np.allclose(F["rot_hi"][4, 0], math.radians(-2))


def test_humanoid_masses_match_mujoco_body_mass():
    """MuJoCo's compile of the same humanoid (Gym/Brax `humanoid.xml`; `model.body_mass` as printed by
    mujoco and quoted in Gymnasium's Humanoid docs): torso 8.90746237, lwaist 2.26194671, pelvis 6.61619411,
    thigh 4.75175093, shin 2.75569617 + foot 1.76714587 (Brax fuses the foot sphere into the shin link),
    upper arm 1.66108048, lower arm 1.22954017.  The MJCF compiler here must reproduce them from the geoms
    (default density 1000)."""
    m = load_model("humanoidrun")
    want = [8.90746237, 2.26194671, 6.61619411, 4.75175093, 2.75569617 + 1.76714587, 4.75175093,
            2.75569617 + 1.76714587, 1.66108048, 1.22954017, 1.66108048, 1.22954017]
    assert np.allclose(1.0 / m.fields["inv_mass"][:11], want, rtol=2e-6)


def test_humanoidtrack_drops_marker_links_and_tracks_five_bodies():
    m = load_model("humanoidtrack")
    assert m.n_links == 11 and m.fields["n_frames"] == 5
    assert m.fields["track_link"].tolist() == [0, 5, 3, 6, 4]  # torso, l_thigh, r_thigh, l_shin, r_shin
    jog = np.load(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", "jog_xref.npy"))
    assert jog.shape == (5, 50, 3) and np.allclose(jog[0, 0], [0.035, 0.022, 1.214], atol=1e-3)
    assert np.array_equal(jog[:, 45], jog[:, 49])  # 46 frames padded by repeating the last row


def test_capsule_and_sphere_inertia_closed_form():
    xml = """<mujoco><compiler angle="radian"/><custom><numeric name="spring_inertia_scale" data="0"/></custom>
    <worldbody><body name="b" pos="0 0 1"><joint type="free"/>
    <geom type="capsule" fromto="0 0 -0.2 0 0 0.2" size="0.05"/></body></worldbody></mujoco>"""
    m = _compile(xml)
    r, h, rho = 0.05, 0.2, 1000.0
    mc, ms_ = rho * math.pi * r * r * 2 * h, rho * 4 / 3 * math.pi * r ** 3
    assert abs(m.masses[0] - (mc + ms_)) < 1e-9
    izz = mc * r * r / 2 + ms_ * 0.4 * r * r
    ixx = mc * ((2 * h) ** 2 / 12 + r * r / 4) + ms_ * (0.4 * r * r + h * h + 0.75 * h * r)
    assert np.allclose(np.diag(m.inertias[0]), [ixx, ixx, izz], rtol=1e-9)
    assert np.allclose(m.fields["inv_inertia"][0, :3], [1 / ixx, 1 / ixx, 1 / izz], rtol=1e-5)
    assert m.fields["iso_inertia"] == 0


def test_fusing_moves_geoms_and_parallel_axis():
    xml = """<mujoco><compiler angle="radian"/><worldbody><body name="a" pos="0 0 1"><joint type="free"/>
    <geom type="sphere" size="0.1"/><body name="fixed" pos="0.5 0 0"><geom type="sphere" size="0.1"/></body>
    </body></worldbody></mujoco>"""
    m = _compile(xml)
    ms_ = 1000 * 4 / 3 * math.pi * 1e-3
    assert m.n_links == 1 and abs(m.masses[0] - 2 * ms_) < 1e-9
    assert np.allclose(m.fields["com"][0], [0.25, 0, 0])
    iyy = 2 * (0.4 * ms_ * 0.01 + ms_ * 0.25 ** 2)
    assert abs(m.inertias[0][1, 1] - iyy) < 1e-9 and abs(m.inertias[0][0, 0] - 2 * 0.4 * ms_ * 0.01) < 1e-9


def test_unsupported_features_raise():
    with pytest.raises(ValueError):
        _compile("""<mujoco><worldbody><body name="a"><joint type="free"/><geom type="box" size="1 1 1"/>
        </body></worldbody></mujoco>""")
    with pytest.raises(ValueError):
        _compile("""<mujoco><worldbody><body name="a"><joint type="hinge" axis="1 0 0"/>
        <joint type="hinge" axis="1 1 0"/><geom type="sphere" size="1"/></body></worldbody></mujoco>""")


def test_json_roundtrip_is_exact():
    from mbd_hip.model import Model
    m = load_model("humanoidrun")
    m2 = Model.from_json(m.to_json())
    assert bytes(m.to_struct()) == bytes(m2.to_struct())


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference assets only exist in the build container")
def test_committed_models_match_a_fresh_compile_of_the_reference_assets():
    from mbd_hip import mjcf
    from mbd_hip.envs import specs
    for name in ("humanoidrun", "humanoidtrack"):
        sp = specs.SPECS[name]
        fresh = mjcf.load(f"/root/reference/mbd/assets/{sp['xml']}", env_name=name, n_frames=sp["n_frames"],
                          drop_link_suffix=sp.get("drop_suffix"), track_names=sp.get("track", ()),
                          reset_noise=sp["reset_noise"])
        assert bytes(fresh.to_struct()) == bytes(load_model(name).to_struct()), name


def test_stability_report_names_a_multi_dof_joint_that_can_reach_its_euler_pole():
    """Round-6 fuzz (seed 2123): a two-dof joint whose MIDDLE hinge is unlimited turned to -90 degrees and the step returned NaN —
    in checker and kernel alike: the joint's Euler angles are undefined there.  mjcf.stability_report says so for such a model
    (and stays empty for one whose middle hinge is limited inside (-90, 90), like every multi-dof joint of the reference's humanoids)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from mbd_hip import mjcf
    from test_oracle_physics import _compile
    xml = """<mujoco><compiler angle="degree" inertiafromgeom="true"/><option timestep="0.003"/>
<custom><numeric name="joint_scale_pos" data="0.5"/><numeric name="joint_scale_ang" data="0.2"/></custom>
<worldbody><body name="base" pos="0 0 1"><joint type="free"/><geom type="sphere" size="0.2"/>
 <body name="arm" pos="0.3 0 0"><joint type="hinge" axis="0 1 0" name="a" range="-60 60"/><joint type="hinge" axis="1 0 0" name="b"{rng}/>
  <geom type="capsule" fromto="0 0 0 0 0 -0.3" size="0.04"/></body></body></worldbody></mujoco>"""
    free = mjcf.stability_report(_compile(xml.format(rng=""), warn_unstable=False))
    assert len(free) == 1 and "middle hinge" in free[0] and "'arm'" in free[0]
    assert mjcf.stability_report(_compile(xml.format(rng=' range="-70 70"'), warn_unstable=False)) == []
