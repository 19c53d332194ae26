import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = {
    "pendulum": os.path.join(ROOT, "models", "pendulum.xml"),
    "cartpole": os.path.join(ROOT, "models", "cartpole.xml"),
    "humanoid": os.path.join(ROOT, "models", "humanoid.xml"),
    "drone2": os.path.join(ROOT, "models", "drone2", "scene.xml"),
}

# the reference's test model (reference tests/test_mujoco_template.py:40-61), reproduced as test data
BASE_XML = """
<mujoco model="template-test">
  <option timestep="0.005"/>
  <default>
    <joint limited="true" range="-1 1"/>
  </default>
  <worldbody>
    <body name="torso">
      <joint name="hinge" type="hinge" axis="0 0 1"/>
      <geom name="torso_geom" type="capsule" size="0.04 0.2" pos="0 0 0"/>
      <site name="tip" pos="0 0 0.2"/>
    </body>
  </worldbody>
  <actuator>
    <motor name="torque_act" joint="hinge" group="0" forcelimited="true" forcerange="-10 10"/>
    <position name="pos_act" joint="hinge" group="1" ctrllimited="true" ctrlrange="-0.5 0.5"/>
  </actuator>
  <sensor>
    <jointpos name="hinge_pos" joint="hinge"/>
  </sensor>
</mujoco>
"""


def chain_xml(n, link=0.1, z0=0.3):
    """A synthetic n-link chain (hinges about alternating axes, capsule links over a floor, a motor on every fourth joint): nv = n.  Not a
    reference model - it exists to run the fp32 MFMA solves (one wavefront per environment, nv <= 32) at matrix sizes the reference's
    models do not have: n = 32 (no spare column for the right-hand side), even n, n far below 32."""
    s = ['<mujoco model="chain"><option timestep="0.005"/>',
         '<default><joint type="hinge" armature="0.3" damping="0.3" limited="true" range="-50 50"/>',
         '<geom type="capsule" size="0.03" contype="1" conaffinity="0" density="800"/></default>',
         '<worldbody><geom name="floor" type="plane" size="5 5 0.1" contype="1" conaffinity="1"/>']
    for k in range(n):
        pos = f"0 0 {z0}" if k == 0 else f"{link} 0 0"
        ax = "0 1 0" if k % 2 == 0 else "0 0 1"
        ct = 1 if k >= n - 4 else 0                                # only the last four links collide with the floor: at most 8 contacts
        s.append(f'<body name="b{k}" pos="{pos}"><joint name="j{k}" axis="{ax}"/><geom fromto="0 0 0 {link} 0 0" contype="{ct}"/>')
    s.append("</body>" * n)
    s.append("</worldbody><actuator>")
    for k in range(0, n, 4):
        s.append(f'<motor name="m{k}" joint="j{k}" gear="2" ctrllimited="true" ctrlrange="-1 1"/>')
    s.append("</actuator></mujoco>")
    return "\n".join(s)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Measured parity errors next to their tolerances.  Every fp32 tolerance of the GPU suite goes through ``measured()``: the value is
# kept (dumped to gpurun_out/parity_measured.json when the session ends - the numbers DESIGN.md §7 quotes) and printed in the assertion
# message; tolerances are set to <= 3x what was measured.  MJB_PARITY_RECORD_ONLY=1 records without asserting (a calibration run).
_MEASURED: dict = {}


def measured(key: str, value, tol: float, what: str = "") -> None:
    value = float(value)
    prev = _MEASURED.get(key)
    _MEASURED[key] = {"measured": max(value, prev["measured"]) if prev else value, "tol": float(tol)}
    if os.environ.get("MJB_PARITY_RECORD_ONLY") == "1":
        return
    assert value <= tol, f"{key}: measured {value:.3e} > tolerance {tol:.3e} {what}"


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    import json

    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.json"), "w") as f:
            json.dump(dict(sorted(_MEASURED.items())), f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="session")
def models():
    return MODELS


# two capsules with PARALLEL axes, the shorter one (half-length 0.2, r 0.04) lying 5 mm inside the longer one (0.3, r 0.05),
# 0.05 off-centre: MuJoCo's parallel branch gives two contacts, at the two ends of the shorter capsule (hand-derived in
# tests/test_oracle_anchors.py::test_parallel_capsules_give_two_contacts)
CAPSULES_XML = """
<mujoco model="parallel-capsules">
  <option timestep="0.002"/>
  <worldbody>
    <body name="a" pos="0 0 1"><freejoint/><geom name="ga" type="capsule" size="0.05" fromto="-0.3 0 0 0.3 0 0"/></body>
    <body name="b" pos="0.05 0 1.085"><freejoint/><geom name="gb" type="capsule" size="0.04" fromto="-0.2 0 0 0.2 0 0"/></body>
  </worldbody>
</mujoco>
"""


@pytest.fixture(scope="session")
def compiled():
    from mujoco_template_amd.mjcf import compile_xml_path

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = compile_xml_path(MODELS[name])
        return cache[name]

    return get


@pytest.fixture(scope="session")
def oracle(compiled):
    from oracle import mjo

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = mjo.OracleModel(compiled(name))
        return cache[name], mjo.OracleData(cache[name])

    return get

# four non-plane primitive pairs, 10 m apart (hand-derived contact geometry in tests/test_oracle_anchors.py)
_ROT_Y = 'quat="0.7071067811865476 0 0.7071067811865476 0"'       # capsule axis z -> x
_ROT_X = 'quat="0.7071067811865476 -0.7071067811865476 0 0"'      # capsule axis z -> y
PAIRS_XML = f"""<mujoco><option gravity="0 0 0"/><worldbody>
  <body pos="0 0 0"><freejoint/><geom type="capsule" size="0.1 0.5" {_ROT_Y} contype="1" conaffinity="1"/></body>
  <body pos="0.2 0.1 0.15"><freejoint/><geom type="capsule" size="0.1 0.5" {_ROT_X} contype="1" conaffinity="1"/></body>
  <body pos="10 0 0"><freejoint/><geom type="capsule" size="0.1 0.5" {_ROT_Y} contype="2" conaffinity="2"/></body>
  <body pos="10.6 0 0.08"><freejoint/><geom type="capsule" size="0.05 0.5" {_ROT_X} contype="2" conaffinity="2"/></body>
  <body pos="20 0 0"><freejoint/><geom type="capsule" size="0.1 0.5" {_ROT_Y} contype="4" conaffinity="4"/></body>
  <body pos="20.3 0 0.17"><freejoint/><geom type="sphere" size="0.1" contype="4" conaffinity="4"/></body>
  <body pos="30 0 0"><freejoint/><geom type="sphere" size="0.1" contype="8" conaffinity="8"/></body>
  <body pos="30.15 0.2 0"><freejoint/><geom type="sphere" size="0.2" contype="8" conaffinity="8"/></body>
  </worldbody></mujoco>"""
