"""The HIP kernel SOURCE (csrc/mjb_device.hpp) executed on the host — one thread per lane, real
barriers (tests/hostemu) — against the oracle.  Checks lane-strided indexing, scans, shuffles and
sync placement without a GPU; the GPU parity tests (-m gpu) run the real kernels.  CPU only."""
import numpy as np
import pytest

from oracle import mjo
from tests.hostemu.emu import EmuEnv


def _pair(compiled, name, G, use_double=True):
    cm = compiled(name)
    return cm, mjo.OracleData(mjo.OracleModel(cm)), EmuEnv(cm, G=G, use_double=use_double)


@pytest.mark.parametrize("name,G", [("pendulum", 8), ("cartpole", 8), ("drone2", 16), ("humanoid", 16), ("humanoid", 64)])
def test_forward_phases_match_oracle(compiled, name, G):
    cm, od, e = _pair(compiled, name, G)
    rng = np.random.default_rng(0)
    q = od.integrate_pos(cm.qpos0, rng.normal(size=cm.nv) * 0.1, 1.0)
    v = rng.normal(size=cm.nv) * 0.5
    u = rng.uniform(-1, 1, size=cm.nu)
    od.qpos[:] = q; od.qvel[:] = v; od.ctrl[:] = u
    e.qpos[:] = q; e.qvel[:] = v; e.ctrl[:cm.nu] = u
    od.forward(); e.forward()
    for k in ("xpos", "xipos", "subtree_com", "cdof", "cinert", "cvel", "qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "qfrc_constraint", "qacc"):
        a, b = getattr(e, k), getattr(od, k)
        scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
        assert np.abs(a[:b.size] - b).max() <= 1e-11 * scale, k
    assert e.counters[1] == od.counters()["nefc"]


@pytest.mark.parametrize("name,G,steps,scale", [("pendulum", 8, 60, 1.0), ("cartpole", 8, 60, 0.01), ("drone2", 16, 40, 0.3), ("humanoid", 16, 12, 1.0)])
def test_free_running_float64_tracks_oracle(compiled, name, G, steps, scale):
    cm, od, e = _pair(compiled, name, G)
    if name == "pendulum":
        od.qpos[0] = e.qpos[0] = 1.0
    for s in range(steps):
        u = od.random_ctrl(0, 0, s, scale)
        od.ctrl[:] = u; e.ctrl[:cm.nu] = u
        od.step(); e.step()
        c = od.counters()
        assert (e.counters[0], e.counters[1], e.counters[2]) == (c["ncon"], c["nefc"], c["solver_niter"])
    assert np.abs(e.qpos - od.qpos).max() < 1e-10
    assert np.abs(e.qvel - od.qvel).max() < 1e-8
    assert e.time[0] == pytest.approx(od.time)


def test_humanoid_contacts_wave64_and_fp32(compiled):
    """Standing humanoid: 8 foot contacts -> 32 pyramidal rows; G=64 uses the shuffle triangular solve."""
    cm, od, e = _pair(compiled, "humanoid", 64)
    od.forward(); e.forward()
    assert e.counters[0] == 8 and e.counters[1] == 32
    nv = cm.nv
    J_o = od.efc_J.reshape(-1, nv)
    J_e = e.efc_J[: 32 * nv].reshape(32, nv)
    assert np.abs(J_e - J_o).max() < 1e-12
    assert np.abs(e.efc_aref[:32] - od.efc_aref).max() < 1e-9
    assert np.abs(e.efc_D[:32] - od.efc_D).max() < 1e-9 * od.efc_D.max()
    assert np.abs(e.qacc - od.qacc).max() < 1e-9
    # one full step with the wavefront-wide (G = 64) path: register-tiled Cholesky of M, M + J^T D J and M + h D
    e.step(); od.step()
    assert np.abs(e.qpos - od.qpos).max() < 1e-12 and np.abs(e.qvel - od.qvel).max() < 1e-9
    od.reset(); od.forward()
    # fp32 instantiation of the same source: single-step error at fp32 level
    e32 = EmuEnv(cm, G=16, use_double=False)
    e32.step()
    od.reset(); od.step()
    assert np.abs(e32.qpos - od.qpos).max() < 2e-6


def test_row_cap_overflow_matches_oracle(compiled):
    """Constraint-row / contact caps: both sides drop the same tail rows and count them."""
    cm = compiled("humanoid")
    om = mjo.OracleModel(cm)
    om.set_limits(6, 20)
    od = mjo.OracleData(om)
    e = EmuEnv(cm, G=16, use_double=True, ncon_max=6, nefc_max=20)
    od.forward(); e.forward()
    c = od.counters()
    assert (c["ncon"], c["nefc"]) == (6, 20) and c["ncon_dropped"] == 2
    assert (e.counters[0], e.counters[1]) == (6, 20) and e.counters[3] == 2
    assert np.abs(e.qacc - od.qacc).max() < 1e-9


def test_bad_state_resets_like_mj_checkpos(compiled):
    cm, od, e = _pair(compiled, "cartpole", 8)
    od.qpos[0] = e.qpos[0] = np.nan
    od.step(); e.step()
    assert od.counters()["warn_badqpos"] == 1 and e.counters[5] == 1
    assert np.isfinite(e.qpos).all() and np.abs(e.qpos - od.qpos).max() < 1e-12


def test_drone_sensors_match_oracle(compiled):
    """gyro / accelerometer / framequat (x2.xml:83-87): emulated kernel vs oracle, in flight and after a step."""
    cm, od, e = _pair(compiled, "drone2", 16)
    rng = np.random.default_rng(4)
    q = od.integrate_pos(cm.qpos0, rng.normal(size=6) * 0.3, 1.0); q[2] += 1.0
    v = rng.normal(size=6)
    u = np.array([4.0, 3.0, 5.0, 2.0])
    od.qpos[:] = q; od.qvel[:] = v; od.ctrl[:] = u
    e.qpos[:] = q; e.qvel[:] = v; e.ctrl[:4] = u
    od.forward(); e.forward()
    assert np.abs(e.sensordata[:10] - od.sensordata).max() < 1e-11
    od.step(); e.step()                        # sensordata keeps the values of the forward pass inside the step
    assert np.abs(e.sensordata[:10] - od.sensordata).max() < 1e-11


@pytest.mark.parametrize("name,G", [("drone2", 16), ("humanoid", 64), ("cartpole", 8)])
def test_inverse_dynamics_matches_oracle(compiled, name, G):
    """mjb_inverse's device code (mode 2) vs the oracle's mj_inverse restatement, state with contacts / limits."""
    cm, od, e = _pair(compiled, name, G)
    od.rollout_random(40, seed=3, env=1, scale=0.3 if name != "cartpole" else 0.01)
    rng = np.random.default_rng(1)
    od.qacc[:] = rng.normal(size=cm.nv)
    e.qpos[:] = od.qpos; e.qvel[:] = od.qvel; e.qacc[:] = od.qacc
    od.inverse(); e.inverse()
    s = max(1.0, np.abs(od.qfrc_inverse).max())
    assert np.abs(e.qfrc_inverse - od.qfrc_inverse).max() < 1e-10 * s
    assert np.abs(e.actuator_moment[: cm.nu * cm.nv] - od.actuator_moment).max() < 1e-12
    assert np.abs(e.qacc - od.qacc).max() == 0            # state untouched


def test_more_bodies_than_lanes_takes_the_slow_tree_paths():
    """nbody > G: the per-lane register schedule of the tree passes (kinematics composition, velocity prefix sums, subtree
    rounds) falls back to its strided loop.  A branching tree of 13 bodies with 3 dofs on 8 lanes, vs the oracle."""
    from mujoco_template_amd import mjcf

    def limb(name, depth):
        if depth == 0:
            return ""
        return (f"<body name='{name}{depth}' pos='0.1 0.02 -0.15' euler='0 10 5'><geom type='capsule' size='0.02' fromto='0 0 0 0.1 0 -0.1'/>"
                + limb(name, depth - 1) + "</body>")

    xml = f"""<mujoco><option timestep='0.004' gravity='0 0 -9.81'/>
      <worldbody><geom name='floor' type='plane' size='2 2 0.1'/>
        <body name='root' pos='0 0 1.0'><joint name='j0' type='hinge' axis='0 1 0' damping='0.1'/>
          <geom type='capsule' size='0.03' fromto='0 0 0 0 0 -0.3'/>
          <body name='a' pos='0 0 -0.3'><joint name='j1' type='hinge' axis='1 0 0' damping='0.05' limited='true' range='-40 40'/>
            <geom type='sphere' size='0.05'/>{limb('l', 5)}{limb('r', 4)}
          </body>
          <body name='b' pos='0.1 0 -0.1'><joint name='j2' type='slide' axis='0 0 1' damping='0.2'/><geom type='box' size='0.03 0.03 0.03' contype='0' conaffinity='0'/></body>
        </body></worldbody>
      <actuator><motor joint='j0' gear='2'/><motor joint='j2' gear='1'/></actuator></mujoco>"""
    cm = mjcf.compile_xml_string(xml)
    assert cm.nbody == 13 and cm.nv == 3
    od = mjo.OracleData(mjo.OracleModel(cm))
    e = EmuEnv(cm, G=8)
    rng = np.random.default_rng(4)
    q = rng.normal(size=cm.nq) * 0.3; v = rng.normal(size=cm.nv); u = rng.uniform(-1, 1, size=cm.nu)
    od.qpos[:] = q; od.qvel[:] = v; od.ctrl[:] = u
    e.qpos[:] = q; e.qvel[:] = v; e.ctrl[:cm.nu] = u
    od.forward(); e.forward()
    for k in ("xpos", "xipos", "subtree_com", "cdof", "cinert", "cvel", "qM", "qfrc_bias", "qfrc_passive", "qacc"):
        a, b = getattr(e, k), getattr(od, k)
        assert np.abs(a[:b.size] - b).max() <= 1e-11 * max(1.0, float(np.abs(b).max())), k
    for _ in range(30):
        od.step(); e.step()
    assert np.abs(e.qpos - od.qpos).max() < 1e-10 and np.abs(e.qvel - od.qvel).max() < 1e-9


def test_parallel_capsules_two_contacts_in_the_kernel_source():
    """The kernel's mjraw_CapsuleCapsule parallel branch (two end-cap contacts) against the oracle and the hand-derived anchor."""
    from mujoco_template_amd import mjcf
    from tests.conftest import CAPSULES_XML

    cm = mjcf.compile_xml_string(CAPSULES_XML)
    od = mjo.OracleData(mjo.OracleModel(cm))
    e = EmuEnv(cm, G=16, use_double=True)
    od.forward(); e.forward()
    assert e.counters[0] == 2 and e.counters[1] == 8 == od.counters()["nefc"]
    assert np.abs(e.efc_J[: 8 * cm.nv] - od.efc_J).max() < 1e-12
    assert np.abs(e.qacc - od.qacc).max() < 1e-9 * max(1.0, np.abs(od.qacc).max())
    for s in range(5):
        od.step(); e.step()
    assert np.abs(e.qpos - od.qpos).max() < 1e-11


def test_fp32_mfma_solver_paths_run_on_the_cpu(compiled):
    """The fp32 hot path itself (one wavefront per environment, nv <= 32): sweep inverse of M and M + hD in the emulated 32x32x2 MFMA
    accumulator, MFMA Cholesky of the Newton Hessian with fused forward / blocked backward substitution, reuse solves from the packed
    factor, J^T f split over the wave halves — executed by the CPU suite through the emulated wave intrinsics (the
    host emulation used to switch exactly these paths off).  Against the float64 oracle at fp32 tolerances, humanoid with foot contacts."""
    cm = compiled("humanoid")
    od = mjo.OracleData(mjo.OracleModel(cm))
    e = EmuEnv(cm, G=64, use_double=False, ncon_max=24, nefc_max=64)
    rng = np.random.default_rng(2)
    for s in range(3):                                       # a few random-ctrl steps into the foot contacts, teacher-forced
        u = od.random_ctrl(0, 0, s, 1.0)
        od.ctrl[:] = u
        e.qpos[:] = od.qpos; e.qvel[:] = od.qvel; e.qacc_warmstart[:] = od.qacc_warmstart; e.ctrl[:cm.nu] = u
        if s == 2:
            od.forward(); e.forward()
            assert e.counters[1] == od.counters()["nefc"] >= 4                      # constraint rows: the Hessian path is taken
            scale = max(1.0, np.abs(od.qacc).max())
            assert np.abs(e.qacc_smooth - od.qacc_smooth).max() < 2e-5 * max(1.0, np.abs(od.qacc_smooth).max())      # sweep inverse of M
            assert np.abs(e.qacc - od.qacc).max() < 5e-5 * scale                                                      # MFMA Cholesky Newton steps
            assert np.abs(e.qfrc_constraint - od.qfrc_constraint).max() < 5e-5 * max(1.0, np.abs(od.qfrc_constraint).max())
        od.step(); e.step()
        assert np.abs(e.qpos - od.qpos).max() < 2e-6 and np.abs(e.qvel - od.qvel).max() < 2e-3     # one-step fp32 bounds (sweep of M + hD in Euler)
    assert e.counters[2] >= 1                                # Newton iterations were needed
    # free-running for a few steps: stays at the fp32 level
    for s in range(3, 8):
        u = od.random_ctrl(0, 0, s, 1.0)
        od.ctrl[:] = u; e.ctrl[:cm.nu] = u
        od.step(); e.step()
    assert np.abs(e.qpos - od.qpos).max() < 2e-5


@pytest.mark.parametrize("n", [32, 30])
def test_fp32_mfma_solves_at_other_matrix_sizes_on_the_cpu(n):
    """The elimination-order MFMA Cholesky / sweep inverse at sizes the reference's models do not have (tests.conftest.chain_xml): n = 32
    has no spare accumulator column for the right-hand side (both substitutions come from the packed factor instead), n = 30 is even
    (no half panel at the end).  Emulated wave intrinsics against the float64 oracle, teacher-forced, contacts and limits active."""
    from tests.conftest import chain_xml
    from mujoco_template_amd import mjcf
    cm = mjcf.compile_xml_string(chain_xml(n))
    od = mjo.OracleData(mjo.OracleModel(cm))
    e = EmuEnv(cm, G=64, use_double=False, ncon_max=16, nefc_max=48)
    od.qpos[:] = 0; od.qpos[0] = np.arcsin(0.3 / (0.1 * n))              # pitched down until the tip touches the floor
    worst = 0.0
    for s in range(3):
        u = od.random_ctrl(1, 0, s, 1.0)
        e.qpos[:] = od.qpos; e.qvel[:] = od.qvel; e.qacc_warmstart[:] = od.qacc_warmstart; e.ctrl[:cm.nu] = u
        od.ctrl[:] = u
        od.forward(); e.forward()
        assert e.counters[1] == od.counters()["nefc"] >= 8
        worst = max(worst, np.abs(e.qacc - od.qacc).max() / max(1.0, np.abs(od.qacc).max()),
                    np.abs(e.qacc_smooth - od.qacc_smooth).max() / max(1.0, np.abs(od.qacc_smooth).max()))
        od.step(); e.step()
        assert np.abs(e.qpos - od.qpos).max() < 2e-7 and np.abs(e.qvel - od.qvel).max() < 4e-5
    assert worst < 6e-4                                                  # measured 1.4e-4 (n = 32), 8.5e-5 (n = 30)


def test_fp32_joint_angles_beyond_pi_on_the_cpu(compiled):
    """The half-angle reduction by multiples of pi (three exact pieces, in front of the fp32 sin / cos polynomials) at hinge angles far
    outside +-pi: emulated kernel against the float64 oracle on the pendulum's site position."""
    cm = compiled("pendulum")
    od = mjo.OracleData(mjo.OracleModel(cm))
    e = EmuEnv(cm, G=8, use_double=False)
    for theta in (3.2, -3.5, 10.0, -50.0, 100.5, 355.0):
        t32 = float(np.float32(theta))
        od.qpos[:] = t32; od.qvel[:] = 0; od.forward()
        e.qpos[:] = t32; e.qvel[:] = 0; e.forward()
        assert np.abs(e.site_xpos.ravel() - od.site_xpos.ravel()).max() <= 2e-6 + 0.5 * 1.2e-7 * abs(theta), theta
