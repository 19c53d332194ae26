"""MJCF-subset compiler: sizes of the five reference models and the
hand-derivable constants K2/K3/K5/K6.  CPU only."""
import math

import numpy as np
import pytest

from mujoco_template_amd import mjcf
from tests.conftest import BASE_XML

SIZES = {  # nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nsensordata, nkey
    "pendulum": (1, 1, 1, 2, 1, 1, 1, 0, 0, 0),
    "cartpole": (2, 2, 1, 3, 2, 3, 1, 0, 0, 0),
    "humanoid": (28, 27, 21, 17, 22, 20, 0, 2, 0, 4),
    "drone2": (7, 6, 4, 2, 1, 11, 5, 0, 10, 1),
}


@pytest.mark.parametrize("name", list(SIZES))
def test_model_sizes(compiled, name):
    m = compiled(name)
    got = (m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nsite, m.ntendon, m.nsensordata, m.nkey)
    assert got == SIZES[name]


def test_base_xml_sizes():
    m = mjcf.compile_xml_string(BASE_XML)
    assert (m.nq, m.nv, m.nu, m.nsensordata, m.nsite) == (1, 1, 2, 1, 1)
    assert m.timestep == pytest.approx(0.005)
    assert list(m.actuator_group) == [0, 1]
    assert bool(m.actuator_forcelimited[0]) and bool(m.actuator_ctrllimited[1])
    # <position kp=1>: gain kp, bias -kp*q
    assert m.actuator_gainprm[1, 0] == 1.0 and m.actuator_biasprm[1, 1] == -1.0
    # default angle unit is degrees: range="-1 1" on a hinge is +-1 degree
    assert m.jnt_range[0] == pytest.approx([-math.pi / 180, math.pi / 180])


def test_k2_pendulum_mass_and_inertia(compiled):
    m = compiled("pendulum")
    assert m.body_mass[1] == pytest.approx(0.794195, rel=1e-6)
    assert m.qM0[0, 0] == pytest.approx(0.0680910, rel=1e-6)      # inertia about the hinge
    assert m.integrator == mjcf.INT_RK4 and m.timestep == 0.005
    assert m.jnt_range[0] == pytest.approx([-360.0, 360.0])        # radians: never active


def test_k3_cartpole_masses(compiled):
    m = compiled("cartpole")
    assert m.body_mass[1] == pytest.approx(4.8)
    assert m.body_mass[2] == pytest.approx(0.373064, rel=1e-6)
    assert m.qM0[1, 1] == pytest.approx(0.0461164, rel=1e-6)
    assert m.qM0[0, 0] == pytest.approx(4.8 + 0.373064, rel=1e-6)
    assert m.npair == 2    # floor-cart (box) and floor-pole (capsule); cart-pole is parent-child filtered
    assert list(m.dof_damping) == [1.0, 0.1]


def test_k5_humanoid_keyframes_and_structure(compiled):
    m = compiled("humanoid")
    assert m.key_qpos.shape == (4, 28)
    assert m.names[mjcf.OBJ_KEY] == ["squat", "stand_on_left_leg", "prone", "supine"]
    assert m.body_mass.sum() == pytest.approx(40.844, rel=1e-4)
    # hinge ranges are degrees in this file
    j = m.name2id(mjcf.OBJ_JOINT, "knee_right")
    assert m.jnt_range[j] == pytest.approx(np.deg2rad([-160, 2]))
    assert m.jnt_solimp[j] == pytest.approx([0, 0.99, 0.01, 0.5, 2])
    assert m.dof_armature[6:].min() == pytest.approx(0.01)
    # contact parameters mix: floor (default) x body capsules (solref .015, solimp .9 .99 .003, condim 1, friction .7)
    floor = m.name2id(mjcf.OBJ_GEOM, "floor")
    p = [i for i in range(m.npair) if m.pair_geom1[i] == floor][0]
    assert m.pair_condim[p] == 3
    assert m.pair_friction[p, 0] == pytest.approx(1.0)
    assert m.pair_solref[p] == pytest.approx([0.0175, 1.0])
    assert m.pair_solimp[p] == pytest.approx([0.9, 0.97, 0.002, 0.5, 2.0])
    # <exclude> and parent-child filtering: no pair between waist_lower and the thighs, none within a weld group
    wl = m.name2id(mjcf.OBJ_BODY, "waist_lower")
    for side in ("thigh_right", "thigh_left"):
        th = m.name2id(mjcf.OBJ_BODY, side)
        for i in range(m.npair):
            assert {m.geom_bodyid[m.pair_geom1[i]], m.geom_bodyid[m.pair_geom2[i]]} != {wl, th}
    head, torso = m.name2id(mjcf.OBJ_BODY, "head"), m.name2id(mjcf.OBJ_BODY, "torso")
    assert m.body_weldid[head] == torso
    # fixed tendons
    assert m.ntendon == 2 and list(m.tendon_num) == [2, 2] and list(m.wrap_prm) == [0.5, -0.5, 0.5, -0.5]


def test_k6_drone_inertia(compiled):
    m = compiled("drone2")
    assert m.body_mass[1] == pytest.approx(1.325)
    assert m.body_ipos[1] == pytest.approx([0, 0, 0.0539623], abs=1e-6)
    R = mjcf.quat_to_mat(m.body_iquat[1])
    I = R @ np.diag(m.body_inertia[1]) @ R.T
    expect = np.array([[0.03665, 0, -0.0021], [0, 0.02541, 0], [-0.0021, 0, 0.06053]])
    assert I == pytest.approx(expect, abs=2e-5)
    # site transmission with 6-D gear, ctrlrange via autolimits, hover keyframe = m g / 4
    assert m.actuator_trntype.tolist() == [mjcf.TRN_SITE] * 4
    assert m.actuator_gear[:, 5].tolist() == [-0.11, 0.11, -0.11, 0.11]
    assert bool(m.actuator_ctrllimited.all())
    assert m.key_ctrl[0] == pytest.approx([1.325 * 9.81 / 4] * 4)
    assert m.density == pytest.approx(1.225) and m.viscosity == pytest.approx(1.8e-5)
    assert m.npair == 8   # 4 boxes + 4 rotor ellipsoids vs the floor; visual geoms do not collide


def test_mass_matrix_jacobian_form_is_spd(compiled):
    for name in SIZES:
        M = compiled(name).qM0
        assert np.allclose(M, M.T)
        assert np.linalg.eigvalsh(M).min() > 0


def test_errors():
    with pytest.raises(mjcf.MjcfError):
        mjcf.compile_xml_string("<notmujoco/>")
    with pytest.raises(mjcf.MjcfError):
        mjcf.compile_xml_string("<mujoco><worldbody><body><joint type='ball'/><geom size='1'/></body></worldbody></mujoco>")
    with pytest.raises(mjcf.MjcfError):
        mjcf.compile_xml_path("/nonexistent/model.xml")
    with pytest.raises(mjcf.MjcfError):   # moving body without mass
        mjcf.compile_xml_string("<mujoco><worldbody><body><joint/></body></worldbody></mujoco>")


# ------------------------------------------------------------------------------------------------------------------------------
# Independent anchors for what the oracle and the HIP path SHARE (both consume mjcf.compile_xml_path's output, so a compiler error
# is invisible to every parity test; VERDICT r1 "Next round" #2).  Every expected value below is derived from numbers
# transcribed BY HAND from the XML with textbook closed forms written here — no compiler function is called to produce it.
# ------------------------------------------------------------------------------------------------------------------------------
def _capsule_mass(r, length, rho=1000.0):
    return rho * (math.pi * r * r * length + 4.0 / 3.0 * math.pi * r ** 3)


def _capsule_inertia_about_com(r, length, rho=1000.0):
    """(I_transverse, I_axial) of a solid capsule: cylinder of the given length + two hemispherical caps (textbook)."""
    mc, ms = rho * math.pi * r * r * length, rho * 4.0 / 3.0 * math.pi * r ** 3
    i_ax = 0.5 * mc * r * r + 0.4 * ms * r * r
    i_tr = mc * (length ** 2 / 12.0 + r * r / 4.0) + ms * (0.4 * r * r + length ** 2 / 4.0 + 3.0 / 8.0 * length * r)
    return i_tr, i_ax


def test_humanoid_masses_and_inertias_from_fromto_capsules_and_nested_defaults(compiled):
    """humanoid.xml:35-104: geoms inherit type=capsule from class "body" (childclass of the torso), their radius from the nested
    classes thigh / shin / foot / arm_upper / arm_lower, the shin and the feet even their fromto; hands are spheres (class hand)."""
    m = compiled("humanoid")
    seg = lambda a, b: math.dist(a, b)                                   # noqa: E731
    # (radius, length) transcribed from the XML, body by body; spheres as (radius, None)
    geoms = {
        "torso": [(0.07, 0.14), (0.06, 0.12)], "head": [(0.09, None)], "waist_lower": [(0.06, 0.12)], "pelvis": [(0.09, 0.14)],
        "thigh_right": [(0.06, seg((0, 0, 0), (0, 0.01, -0.34)))], "shin_right": [(0.049, 0.3)],
        "foot_right": [(0.027, seg((-0.07, -0.01, 0), (0.14, -0.03, 0))), (0.027, seg((-0.07, 0.01, 0), (0.14, 0.03, 0)))],
        "thigh_left": [(0.06, seg((0, 0, 0), (0, -0.01, -0.34)))], "shin_left": [(0.049, 0.3)],
        "foot_left": [(0.027, seg((-0.07, -0.01, 0), (0.14, -0.03, 0))), (0.027, seg((-0.07, 0.01, 0), (0.14, 0.03, 0)))],
        "upper_arm_right": [(0.04, seg((0, 0, 0), (0.16, -0.16, -0.16)))], "lower_arm_right": [(0.031, seg((0.01, 0.01, 0.01), (0.17, 0.17, 0.17)))],
        "hand_right": [(0.04, None)],
        "upper_arm_left": [(0.04, seg((0, 0, 0), (0.16, 0.16, -0.16)))], "lower_arm_left": [(0.031, seg((0.01, -0.01, 0.01), (0.17, -0.17, 0.17)))],
        "hand_left": [(0.04, None)],
    }
    total = 0.0
    for body, gl in geoms.items():
        mass = sum(_capsule_mass(r, ln) if ln is not None else 1000.0 * 4.0 / 3.0 * math.pi * r ** 3 for r, ln in gl)
        total += mass
        assert m.body_mass[m.name2id(mjcf.OBJ_BODY, body)] == pytest.approx(mass, rel=1e-12), body
    assert m.body_mass.sum() == pytest.approx(total, rel=1e-12) and total == pytest.approx(40.8446, abs=1e-3)
    # shin (class "shin": BOTH fromto and size come from the default class): principal inertia of one capsule along z
    b = m.name2id(mjcf.OBJ_BODY, "shin_right")
    i_tr, i_ax = _capsule_inertia_about_com(0.049, 0.3)
    assert sorted(m.body_inertia[b]) == pytest.approx(sorted([i_tr, i_tr, i_ax]), rel=1e-12)
    assert m.body_ipos[b] == pytest.approx([0, 0, -0.15], abs=1e-15)
    # hand: sphere through class "hand" (type overridden inside the capsule class), zaxis attribute irrelevant for a sphere
    b = m.name2id(mjcf.OBJ_BODY, "hand_left")
    assert m.body_inertia[b] == pytest.approx([0.4 * 1000 * 4 / 3 * math.pi * 0.04 ** 5] * 3, rel=1e-12)
    # foot: two capsules splayed by +-atan(0.02/0.21) about z, centres at (0.035, -+0.02, 0): parallel-axis theorem by hand
    b = m.name2id(mjcf.OBJ_BODY, "foot_right")
    ln = seg((-0.07, -0.01, 0), (0.14, -0.03, 0))
    mf = _capsule_mass(0.027, ln)
    i_tr, i_ax = _capsule_inertia_about_com(0.027, ln)
    th = math.atan2(0.02, 0.21)
    c2, s2 = math.cos(th) ** 2, math.sin(th) ** 2
    ixx = 2 * (i_ax * c2 + i_tr * s2 + mf * 0.02 ** 2)            # the xy products of the two mirrored capsules cancel
    iyy = 2 * (i_ax * s2 + i_tr * c2)
    izz = 2 * (i_tr + mf * 0.02 ** 2)
    assert m.body_ipos[b] == pytest.approx([0.035, 0, 0], abs=1e-15)
    assert sorted(m.body_inertia[b]) == pytest.approx(sorted([ixx, iyy, izz]), rel=1e-10)
    # upper arm along (1,-1,-1)/sqrt3: inertia frame = capsule frame, com at the segment midpoint
    b = m.name2id(mjcf.OBJ_BODY, "upper_arm_right")
    i_tr, i_ax = _capsule_inertia_about_com(0.04, seg((0, 0, 0), (0.16, -0.16, -0.16)))
    assert sorted(m.body_inertia[b]) == pytest.approx(sorted([i_tr, i_tr, i_ax]), rel=1e-10)
    assert m.body_ipos[b] == pytest.approx([0.08, -0.08, -0.08], abs=1e-15)


def test_cartpole_invweight0_closed_form(compiled):
    """dof_invweight0 = diag(M^-1), body_invweight0 = tr(J M^-1 J^T)/3 (translation, rotation) at qpos0 — the constants that scale
    every constraint row's regulariser R.  Cart-pole by hand (cartpole.xml:18-27): M = [[mc+mp, mp l], [mp l, I_h]] at theta = 0."""
    m = compiled("cartpole")
    mc = 500.0 * 0.24 * 0.4 * 0.1                                           # box 0.12 x 0.2 x 0.05 half-sizes, density 500
    r, ln, rho = 0.025, 0.6, 300.0
    mp = _capsule_mass(r, ln, rho)
    i_tr, _ = _capsule_inertia_about_com(r, ln, rho)
    l = 0.3                                                                  # com of the pole above the hinge
    ih = i_tr + mp * l * l
    mt = mc + mp
    det = mt * ih - (mp * l) ** 2
    minv = np.array([[ih, -mp * l], [-mp * l, mt]]) / det
    assert m.dof_invweight0 == pytest.approx([minv[0, 0], minv[1, 1]], rel=1e-10)
    cart, pole = m.name2id(mjcf.OBJ_BODY, "cart"), m.name2id(mjcf.OBJ_BODY, "pole")
    assert m.body_invweight0[cart, 0] == pytest.approx(minv[0, 0] / 3, rel=1e-10)          # J_p = [1 0]: x only
    jp = np.array([1.0, l])                                                  # com velocity of the pole along x: xdot + l thetadot
    assert m.body_invweight0[pole, 0] == pytest.approx(jp @ minv @ jp / 3, rel=1e-10)
    assert m.body_invweight0[pole, 1] == pytest.approx(minv[1, 1] / 3, rel=1e-10)          # rotation about y only
    assert m.body_invweight0[cart, 1] <= 1e-12                                              # the cart cannot rotate


def test_humanoid_tendon_limit_row_by_hand(compiled):
    """Fixed tendon hamstring_right (humanoid.xml:192-195): length = 0.5 q_hip_y - 0.5 q_knee, range [-0.3, 2].  With hip_y = -1 rad
    the lower limit is violated by 0.2: one row J = +[0.5, -0.5] on the two dofs, pos = -0.2, and with the default solref (0.02, 1) /
    solimp (0.9, 0.95, 0.001, 0.5, 2): K = 1 / (0.95^2 0.02^2), B = 2 / (0.95 0.02), impedance = dmax = 0.95 (|pos| >> width)
    => aref = -B J.v - K imp pos, R = (1 - imp) / imp * tendon_invweight0, tendon_invweight0 = J M0^-1 J^T."""
    from oracle import mjo

    m = compiled("humanoid")
    od = mjo.OracleData(mjo.OracleModel(m))
    jh, jk = m.name2id(mjcf.OBJ_JOINT, "hip_y_right"), m.name2id(mjcf.OBJ_JOINT, "knee_right")
    od.qpos[2] = 2.0                                                         # lift it: no floor contacts
    od.qpos[m.jnt_qposadr[jh]] = -1.0
    od.qvel[m.jnt_dofadr[jh]] = 0.4
    od.qvel[m.jnt_dofadr[jk]] = -0.6
    od.forward()
    typ = od.efc_type()
    rows = [i for i in range(len(typ)) if typ[i] == 1]                        # EFC_LIMIT_TENDON
    assert len(rows) == 1
    r = rows[0]
    J = od.efc_J.reshape(-1, m.nv)[r]
    want = np.zeros(m.nv); want[m.jnt_dofadr[jh]] = 0.5; want[m.jnt_dofadr[jk]] = -0.5
    assert J == pytest.approx(want, abs=1e-15)
    assert od.efc_pos[r] == pytest.approx(-0.5 + 0.3, abs=1e-15)
    K, B, imp = 1.0 / (0.95 ** 2 * 0.02 ** 2), 2.0 / (0.95 * 0.02), 0.95
    assert od.efc_aref[r] == pytest.approx(-B * (0.5 * 0.4 - 0.5 * -0.6) - K * imp * -0.2, rel=1e-12)
    tinv = want @ np.linalg.solve(m.qM0, want)                               # J M0^-1 J^T with the mass matrix at qpos0
    assert od.efc_D[r] == pytest.approx(imp / ((1 - imp) * tinv), rel=1e-9)


def test_base_xml_position_servo_by_hand():
    """<position joint="hinge" ctrlrange="-0.5 0.5"> (reference tests/test_mujoco_template.py:55): kp defaults to 1, force = kp (u - q)
    with u clamped to the ctrlrange; the motor beside it is force-limited to +-10.  qfrc_actuator = sum of both (gear 1)."""
    from oracle import mjo

    m = mjcf.compile_xml_string(BASE_XML)
    od = mjo.OracleData(mjo.OracleModel(m))
    od.qpos[0] = 0.01
    od.ctrl[:] = [25.0, 0.8]                                                 # motor saturates at 10, servo target clamps to 0.5
    od.forward()
    assert od.qfrc_actuator[0] == pytest.approx(10.0 + 1.0 * (0.5 - 0.01), rel=1e-14)
    od.ctrl[:] = [-3.0, -0.2]
    od.forward()
    assert od.qfrc_actuator[0] == pytest.approx(-3.0 + 1.0 * (-0.2 - 0.01), rel=1e-14)


_REJECTS = {
    "equality": ('<equality><weld body1="torso"/></equality>', "equality"),
    "flag": ('<option><flag gravity="disable"/></option>', "flag"),
    "contact flag": ('<option><flag contact="disable"/></option>', "flag"),
    "frictionloss": None,
    "noslip": ('<option noslip_iterations="3"/>', "noslip"),
    "override": ('<option o_margin="0.01"/>', "override"),
    "wind": ('<option wind="1 0 0"/>', "wind"),
    "eulerseq": ('<compiler eulerseq="zyx"/>', "eulerseq"),
    "settotalmass": ('<compiler settotalmass="5"/>', "settotalmass"),
    "global coordinates": ('<compiler coordinate="global"/>', "coordinate"),
    "unknown section": ('<deformable/>', "deformable"),
    "unknown option": ('<option banana="1"/>', "banana"),
    "contact pair": ('<contact><pair geom1="torso_geom" geom2="torso_geom"/></contact>', "pair"),
    "velocity actuator": ('<actuator><velocity joint="hinge"/></actuator>', "velocity"),
    "touch sensor": ('<sensor><touch site="tip"/></sensor>', "touch"),
}


@pytest.mark.parametrize("what", [k for k, v in _REJECTS.items() if v is not None])
def test_unsupported_mjcf_is_rejected_loudly(what):
    """anything outside the supported subset raises MjcfError naming it — never a silent drop (a model with <equality>,
    <flag gravity=disable> and frictionloss used to compile and simulate different physics without a word)."""
    snippet, needle = _REJECTS[what]
    xml = BASE_XML.replace("</mujoco>", snippet + "\n</mujoco>")
    with pytest.raises(mjcf.MjcfError, match=needle):
        mjcf.compile_xml_string(xml)


def test_unsupported_mjcf_attributes_are_rejected_loudly():
    for old, new, needle in (
        ('<joint name="hinge" type="hinge" axis="0 0 1"/>', '<joint name="hinge" type="hinge" axis="0 0 1" frictionloss="0.1"/>', "frictionloss"),
        ('<joint name="hinge" type="hinge" axis="0 0 1"/>', '<joint name="hinge" type="hinge" axis="0 0 1" wobble="1"/>', "wobble"),
        ('<geom name="torso_geom" type="capsule" size="0.04 0.2" pos="0 0 0"/>', '<geom name="torso_geom" type="capsule" size="0.04 0.2" fluidshape="ellipsoid"/>', "fluid"),
        ('<geom name="torso_geom" type="capsule" size="0.04 0.2" pos="0 0 0"/>', '<geom name="torso_geom" type="mesh" mesh="m" contype="0" conaffinity="0"/>', "mesh"),
        ('<body name="torso">', '<body name="torso" gravcomp="1">', "gravity compensation"),
        ('<body name="torso">', '<body name="torso" mocap="true">', "mocap"),
        ('<joint limited="true" range="-1 1"/>', '<joint limited="true" range="-1 1" frictionloss="2"/>', "frictionloss"),     # in <default>
    ):
        assert old in BASE_XML
        with pytest.raises(mjcf.MjcfError, match=needle):
            mjcf.compile_xml_string(BASE_XML.replace(old, new))
    # harmless values of the same attributes (MuJoCo's defaults) and rendering-only content still compile
    ok = BASE_XML.replace('<joint name="hinge" type="hinge" axis="0 0 1"/>', '<joint name="hinge" type="hinge" axis="0 0 1" frictionloss="0"/>')
    ok = ok.replace("<worldbody>", '<visual><global offwidth="800"/></visual><option><flag gravity="enable"/></option><worldbody><light pos="0 0 3"/>')
    ok = ok.replace('<geom name="torso_geom"', '<geom rgba="1 0 0 1" name="torso_geom"')
    assert mjcf.compile_xml_string(ok).nv == 1


def test_native_compiler_matches_the_python_restatement(models):
    """The product compiler is native (csrc/mjb_mjcf.cpp behind mjb_model_load_xml); tests/pymjcf.py is an independent pure-Python
    restatement of it.  Every field of every model must agree (1e-12 relative; principal frames through the inertia tensors, the
    eigenvector signs of a 3x3 eigen-decomposition being a free choice) — the one check of the compiler that does not pass through
    both the oracle and the HIP path."""
    from tests import pymjcf
    from tests.conftest import CAPSULES_XML

    cases = [(k, dict(path=v)) for k, v in models.items()] + [("base", dict(text=BASE_XML)), ("capsules", dict(text=CAPSULES_XML))]
    for name, kw in cases:
        a = pymjcf.compile_xml_path(kw["path"]) if "path" in kw else pymjcf.compile_xml_string(kw["text"])
        b = mjcf.compile_xml_path(kw["path"]) if "path" in kw else mjcf.compile_xml_string(kw["text"])
        for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ntendon", "nwrap", "nsensor", "nsensordata", "nkey", "npair", "nexclude",
                  "integrator", "iterations", "ls_iterations", "disableactuator", "name"):
            assert getattr(a, k) == getattr(b, k), (name, k)
        for k in ("timestep", "density", "viscosity", "impratio", "tolerance", "meaninertia"):
            assert getattr(a, k) == pytest.approx(getattr(b, k), rel=1e-13), (name, k)
        assert np.array_equal(a.gravity, b.gravity) and a.names == b.names and set(a.arrays) == set(b.arrays), name
        for k in a.arrays:
            x, y = np.asarray(a.arrays[k], dtype=float), np.asarray(b.arrays[k], dtype=float)
            assert x.shape == y.shape, (name, k)
            if k in ("body_iquat", "body_inertia") or x.size == 0:
                continue
            assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(x).max()), (name, k)
        for bb in range(a.nbody):
            ia = mjcf.quat_to_mat(a.body_iquat[bb]) @ np.diag(a.body_inertia[bb]) @ mjcf.quat_to_mat(a.body_iquat[bb]).T
            ib = mjcf.quat_to_mat(b.body_iquat[bb]) @ np.diag(b.body_inertia[bb]) @ mjcf.quat_to_mat(b.body_iquat[bb]).T
            assert np.abs(ia - ib).max() <= 1e-12 * max(1e-30, np.abs(ia).max()), (name, bb)
            assert np.sort(a.body_inertia[bb]) == pytest.approx(np.sort(b.body_inertia[bb]), rel=1e-12, abs=1e-18)


def test_model_load_xml_through_the_c_abi(models, tmp_path):
    """mjb_model_load_xml / mjb_model_load_xml_string (SURVEY §8(b) "model_load_xml"): a host in any language compiles MJCF through the
    C ABI; <include> resolves against the file's directory; errors come back as MJB_ERR_MODEL with the compiler's message."""
    import ctypes

    from mujoco_template_amd._capi import load_library

    L = load_library()
    m = ctypes.c_void_p()
    assert L.mjb_model_load_xml(models["drone2"].encode(), ctypes.byref(m)) == 0          # scene.xml includes x2.xml
    p, n, dt = ctypes.c_void_p(), ctypes.c_long(), ctypes.c_int()
    assert L.mjb_model_field(m, b"nq", ctypes.byref(p), ctypes.byref(n), ctypes.byref(dt)) == 0
    assert ctypes.cast(p, ctypes.POINTER(ctypes.c_int))[0] == 7 and L.mjb_model_name2id(m, 6, b"imu") == 0
    L.mjb_model_free(m)
    bad = tmp_path / "bad.xml"
    bad.write_text("<mujoco><equality/></mujoco>")
    assert L.mjb_model_load_xml(str(bad).encode(), ctypes.byref(m)) == -2 and b"equality" in L.mjb_last_error()
    assert L.mjb_model_load_xml(b"/nonexistent.xml", ctypes.byref(m)) == -2 and b"not found" in L.mjb_last_error()
    assert L.mjb_model_load_xml_string(b"<mujoco><worldbody><body><joint/></body>", b".", ctypes.byref(m)) == -2 and b"XML parse error" in L.mjb_last_error()
    assert L.mjb_model_load_xml_string(b"<mujoco><worldbody><body><joint/></body></worldbody></mujoco>", b".", ctypes.byref(m)) == -2
    assert b"zero mass" in L.mjb_last_error()
