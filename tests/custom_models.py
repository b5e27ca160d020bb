"""A model none of the built-in ones resembles, for the kernels only custom MJCF files reach: ten links, four chains of
different lengths off a free root, full (non-diagonal) inertia tensors from skew capsules, 1/2/3-dof hinge joints with
limits, joint springs and dampers, a slide + hinge joint, a fused (jointless) body, two colliders on one link, actuators
in an order that differs from the joint order."""
CRAB = """<mujoco><compiler angle="radian"/><option timestep="0.004"/>
<custom><numeric name="spring_inertia_scale" data="0"/><numeric name="spring_mass_scale" data="0"/>
<numeric name="constraint_ang_damping" data="30"/><numeric name="constraint_vel_damping" data="0.5"/></custom>
<default><geom contype="0" conaffinity="0" density="900"/></default>
<worldbody><geom type="plane" size="10 10 1" contype="1" conaffinity="1" friction="0.9 0.005 0.0001"/>
<body name="root" pos="0 0 0.62"><joint type="free"/>
 <geom type="capsule" fromto="-0.25 -0.12 -0.03 0.22 0.1 0.06" size="0.07"/>
 <geom type="sphere" pos="0.05 0.2 0.05" size="0.06"/>
 <body name="leg_a" pos="0.22 0.1 0.02">
  <joint name="a_y" type="hinge" axis="0 1 0" range="-0.9 0.9" damping="0.4"/><joint name="a_x" type="hinge" axis="1 0 0" range="-0.7 0.6" stiffness="3"/>
  <geom type="capsule" fromto="0 0 0 0.12 0.07 -0.24" size="0.04"/>
  <body name="shin_a" pos="0.12 0.07 -0.24"><joint name="a_k" type="hinge" axis="0.6 0.8 0" range="-1.4 0.1"/>
   <geom type="capsule" fromto="0 0 0 0.03 -0.02 -0.26" size="0.035"/>
   <geom type="sphere" pos="0.03 -0.02 -0.26" size="0.05" contype="1" conaffinity="1"/></body></body>
 <body name="leg_b" pos="-0.25 -0.12 -0.02">
  <joint name="b_y" type="hinge" axis="0 1 0" range="-1.0 0.8"/>
  <geom type="capsule" fromto="0 0 0 -0.1 -0.08 -0.25" size="0.04"/>
  <body name="shin_b" pos="-0.1 -0.08 -0.25"><joint name="b_k" type="hinge" axis="1 0 0" range="-0.2 1.3" damping="0.2"/>
   <geom type="capsule" fromto="0 0 0 0 0.03 -0.27" size="0.035"/>
   <geom type="sphere" pos="0 0.03 -0.27" size="0.05" contype="1" conaffinity="1"/>
   <geom type="sphere" pos="0.05 0.0 -0.2" size="0.04" contype="1" conaffinity="1"/></body></body>
 <body name="leg_c" pos="0.0 0.2 0.0">
  <joint name="c_z" type="hinge" axis="0 0 1" range="-0.8 0.8"/><joint name="c_y" type="hinge" axis="0 1 0" range="-0.5 1.0"/><joint name="c_x" type="hinge" axis="1 0 0" range="-0.6 0.6"/>
  <geom type="capsule" fromto="0 0 0 0.02 0.18 -0.2" size="0.04"/>
  <body name="shin_c" pos="0.02 0.18 -0.2"><joint name="c_k" type="hinge" axis="1 0 0" range="-1.2 0.3"/>
   <geom type="capsule" fromto="0 0 0 0 -0.03 -0.3" size="0.035"/>
   <geom type="sphere" pos="0 -0.03 -0.3" size="0.05" contype="1" conaffinity="1"/>
   <body name="toe_c" pos="0 -0.03 -0.3"><joint name="c_t" type="hinge" axis="0 1 0" range="-0.5 0.5"/>
    <geom type="capsule" fromto="0 0 0 0.1 0.02 0" size="0.025"/>
    <geom type="sphere" pos="0.1 0.02 0" size="0.03" contype="1" conaffinity="1"/></body></body></body>
 <body name="tail" pos="-0.25 -0.12 0.0">
  <joint name="t_s" type="slide" axis="-0.8 -0.5 0.33" range="-0.05 0.15" damping="1.0"/><joint name="t_y" type="hinge" axis="0 1 0" range="-0.6 0.6"/>
  <geom type="capsule" fromto="0 0 0 -0.2 -0.1 0.08" size="0.03"/>
  <body name="stinger" pos="-0.2 -0.1 0.08"><geom type="sphere" size="0.04"/></body>
  <body name="arm_d" pos="-0.1 -0.05 0.06">
   <joint name="d_z" type="hinge" axis="0 0 1" range="-1 1"/>
   <geom type="capsule" fromto="0 0 0 0.18 0.05 0.07" size="0.03"/></body></body>
</body></worldbody>
<actuator><motor joint="a_y" gear="40" ctrlrange="-1 1"/><motor joint="a_x" gear="30" ctrlrange="-1 1"/><motor joint="a_k" gear="40" ctrlrange="-1 1"/>
<motor joint="b_y" gear="40" ctrlrange="-1 1"/><motor joint="b_k" gear="40" ctrlrange="-1 1"/>
<motor joint="c_z" gear="25" ctrlrange="-1 1"/><motor joint="c_y" gear="40" ctrlrange="-1 1"/><motor joint="c_x" gear="25" ctrlrange="-1 1"/><motor joint="c_k" gear="40" ctrlrange="-1 1"/><motor joint="c_t" gear="10" ctrlrange="-1 1"/>
<motor joint="t_s" gear="60" ctrlrange="-1 1"/><motor joint="t_y" gear="15" ctrlrange="-1 1"/><motor joint="d_z" gear="15" ctrlrange="-0.5 0.5"/></actuator></mujoco>"""

# A planar model the built-in ones do not cover: ten links (a 16-lane candidate group: the planar kernel's widest
# instantiation, shuffle exchange), three legs off the torso, joint springs AND a limited root slide (all the run-time
# switches of the planar kernel on at once), two sphere colliders per foot, a jointless (fused) body.
TRIPOD = """<mujoco model="tripod"><compiler angle="degree" inertiafromgeom="true"/>
<default><joint damping=".1" limited="true"/><geom conaffinity="0" contype="0" friction=".8 .1 .1"/>
<motor ctrllimited="true" ctrlrange="-1 1"/></default>
<option timestep="0.0025"/>
<custom><numeric data="0.5" name="joint_scale_pos"/><numeric data="0.2" name="joint_scale_ang"/>
<numeric data="2" name="constraint_ang_damping"/><numeric data="20" name="constraint_vel_damping"/>
<numeric data="0" name="spring_mass_scale"/><numeric data="0.5" name="spring_inertia_scale"/>
<numeric data="0.1" name="elasticity"/></custom>
<worldbody><geom conaffinity="1" name="floor" pos="0 0 0" size="40 40 40" type="plane" friction=".8 .1 .1"/>
<body name="torso" pos="0 0 1.1">
 <joint axis="1 0 0" damping="0" limited="false" name="rootx" pos="0 0 0" type="slide"/>
 <joint axis="0 0 1" damping="0" limited="true" range="-0.6 0.5" name="rootz" pos="0 0 0" type="slide"/>
 <joint axis="0 1 0" damping="0" limited="false" name="rooty" pos="0 0 0" type="hinge"/>
 <geom fromto="-0.35 0 0 0.35 0 0" size="0.06" type="capsule"/>
 <body name="mast" pos="0 0 0.1"><geom type="sphere" size="0.05"/></body>
 <body name="thigh_a" pos="0.35 0 0"><joint axis="0 -1 0" name="a1" pos="0 0 0" range="-100 20" stiffness="4" type="hinge"/>
  <geom fromto="0 0 0 0 0 -0.4" size="0.05" type="capsule"/>
  <body name="leg_a" pos="0 0 -0.4"><joint axis="0 -1 0" name="a2" pos="0 0 0" range="-140 0" type="hinge"/>
   <geom fromto="0 0 0 0 0 -0.45" size="0.04" type="capsule"/>
   <body name="foot_a" pos="0 0 -0.45"><joint axis="0 -1 0" name="a3" pos="0 0 0" range="-45 45" type="hinge"/>
    <geom contype="1" fromto="-0.05 0 0 0.15 0 0" size="0.05" type="capsule"/></body></body></body>
 <body name="thigh_b" pos="0 0 0"><joint axis="0 1 0" name="b1" pos="0 0 0" range="-60 60" type="hinge"/>
  <geom fromto="0 0 0 0 0 -0.42" size="0.05" type="capsule"/>
  <body name="leg_b" pos="0 0 -0.42"><joint axis="0 -1 0" name="b2" pos="0 0 0" range="-140 0" stiffness="2" type="hinge"/>
   <geom fromto="0 0 0 0 0 -0.43" size="0.04" type="capsule"/>
   <body name="foot_b" pos="0 0 -0.43"><joint axis="0 -1 0" name="b3" pos="0 0 0" range="-45 45" type="hinge"/>
    <geom contype="1" fromto="-0.1 0 0 0.1 0 0" size="0.05" type="capsule"/></body></body></body>
 <body name="thigh_c" pos="-0.35 0 0"><joint axis="0 -1 0" name="c1" pos="0 0 0" range="-20 100" type="hinge"/>
  <geom fromto="0 0 0 0 0 -0.4" size="0.05" type="capsule"/>
  <body name="leg_c" pos="0 0 -0.4"><joint axis="0 -1 0" name="c2" pos="0 0 0" range="0 140" type="hinge"/>
   <geom fromto="0 0 0 0 0 -0.45" size="0.04" type="capsule"/>
   <body name="foot_c" pos="0 0 -0.45"><joint axis="0 -1 0" name="c3" pos="0 0 0" range="-45 45" type="hinge"/>
    <geom contype="1" fromto="-0.15 0 0 0.05 0 0" size="0.05" type="capsule"/></body></body></body>
</body></worldbody>
<actuator><motor gear="80" joint="c1"/><motor gear="60" joint="c2"/><motor gear="30" joint="c3"/>
<motor gear="80" joint="a1"/><motor gear="60" joint="a2"/><motor gear="30" joint="a3"/>
<motor gear="80" joint="b1"/><motor gear="60" joint="b2"/><motor gear="30" joint="b3"/></actuator></mujoco>"""
