"""Host-side logic that needs no GPU: counter-based RNG parity (numpy vs C), sharding, obs layout,
controllers, manifold helpers, reference error behaviour, and the world_size-2 gloo all-gather."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mujoco_template_amd as mt
from mujoco_template_amd import mj, mjcf
from mujoco_template_amd.controllers import philox_uniform
from mujoco_template_amd.distributed import shard_range
from tests.conftest import BASE_XML

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    """A rendezvous port nobody is listening on right now (fixed numbers collide with whatever an earlier run left behind)."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return int(sock.getsockname()[1])


def test_philox_numpy_matches_oracle_c(oracle):
    m, d = oracle("humanoid")
    cm = m.compiled
    for seed, step in ((0, 0), (7, 123), (2**31 + 5, 99999)):
        u = philox_uniform(seed, np.arange(5) + 11, step, cm.nu)
        for e in range(5):
            ctrl = d.random_ctrl(seed, 11 + e, step, 1.0)
            lo, hi = cm.actuator_ctrlrange[:, 0], cm.actuator_ctrlrange[:, 1]
            assert 0.5 * (lo + hi) + 0.5 * (hi - lo) * (2 * u[e] - 1) == pytest.approx(ctrl, abs=1e-15)
    assert ((u >= 0) & (u < 1)).all()


def test_shard_range_partitions_batch():
    for B, W in ((4096, 8), (2048, 8), (10, 4), (3, 8), (4096, 1)):
        blocks = [shard_range(B, r, W) for r in range(W)]
        assert sum(c for _, c in blocks) == B
        pos = 0
        for e0, c in blocks:
            assert e0 == pos
            pos += c
    with pytest.raises(ValueError):
        shard_range(8, 3, 2)


def test_manifold_helpers_roundtrip():
    model = mj.MjModel.from_xml_path(os.path.join(ROOT, "models", "humanoid.xml"))
    rng = np.random.default_rng(0)
    q0 = np.array(model.qpos0)
    v = rng.normal(size=model.nv) * 0.4
    q1 = q0.copy()
    mj.mj_integratePos(model, q1, v, 0.5)
    assert np.linalg.norm(q1[3:7]) == pytest.approx(1.0)
    back = np.zeros(model.nv)
    mj.mj_differentiatePos(model, back, 0.5, q0, q1)
    assert back == pytest.approx(v, abs=1e-12)


def test_manifold_helpers_match_oracle(oracle):
    m, d = oracle("drone2")
    model = mj.MjModel(m.compiled)
    rng = np.random.default_rng(1)
    v = rng.normal(size=6)
    q = np.array(m.compiled.qpos0)
    mj.mj_integratePos(model, q, v, 0.3)
    assert q == pytest.approx(d.integrate_pos(m.compiled.qpos0, v, 0.3), abs=1e-14)


def test_name_lookup_and_model_tables():
    model = mj.MjModel.from_xml_string(BASE_XML)
    assert mj.mj_name2id(model, mj.mjtObj.mjOBJ_SITE, "tip") == 0
    assert mj.mj_name2id(model, mj.mjtObj.mjOBJ_SITE, "nope") == -1
    assert mj.mj_id2name(model, mj.mjtObj.mjOBJ_ACTUATOR, 1) == "pos_act"
    assert model.nu == 2 and model.actuator_ctrlrange.shape == (2, 2)
    with pytest.raises(mt.NameLookupError):
        mt.ObservationExtractor(model, mt.ObservationSpec(sites_pos=("missing",)))
    ex = mt.ObservationExtractor(model, mt.ObservationSpec(include_ctrl=True, include_time=True, sites_pos=("tip",), bodies_pos=("torso",)))
    assert ex.obs_dim == 1 + 1 + 2 + 1 + 3 + 3


def test_jacobian_token_grammar():
    from mujoco_template_amd.jacobians import _parse_jacobian_token

    assert _parse_jacobian_token("site:tip") == ("site", "tip")
    assert _parse_jacobian_token("subtreecom:torso") == ("subtreecom", "torso")
    assert _parse_jacobian_token("com") == ("com", None)
    with pytest.raises(mt.ConfigError):
        _parse_jacobian_token("frame:x")


def test_compat_report_matches_reference_rules():
    model = mj.MjModel.from_xml_string(BASE_XML)
    caps = mt.ControllerCapabilities(control_space=mt.ControlSpace.POSITION, actuator_groups=(1,))
    rep = mt.check_controller_compat(model, caps, np.array([True, True]))
    assert rep.ok
    assert any("lacks ctrlrange" in w for w in rep.warnings)          # actuator 0 is a torque motor
    assert any("beyond the controller request" in w for w in rep.warnings)
    rep = mt.check_controller_compat(model, caps, np.array([False, False]))
    assert not rep.ok
    with pytest.raises(mt.CompatibilityError):
        rep.assert_ok()
    with pytest.raises(mt.ConfigError):
        mt.check_controller_compat(model, caps, np.array([True]))


def test_controller_protocol_objects():
    z = mt.ZeroController()
    assert z.device_ctrl_mode == "zero" and z.capabilities.control_space == mt.ControlSpace.TORQUE
    r = mt.RandomCtrlController(seed=3, scale=0.5)
    assert r.device_ctrl_mode == "random"


def test_gloo_world_size_2_all_gather_obs(tmp_path):
    """The N>1 path (shard_range + all_gather_obs) under torch.distributed/gloo with two CPU ranks."""
    script = tmp_path / "worker.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from mujoco_template_amd.distributed import init_process_group, shard_range, all_gather_obs, world\n"
        "rank, ws, _ = world()\n"
        "assert init_process_group('gloo')\n"
        "B, dim = 10, 7\n"
        "full = torch.arange(3 * B * dim, dtype=torch.float32).reshape(3, B, dim)\n"
        "e0, c = shard_range(B, rank, ws)\n"
        "out = all_gather_obs(full[:, e0:e0 + c].clone())\n"
        "assert out.shape == full.shape and torch.equal(out, full), (rank, out.shape)\n"
        "e0, c = shard_range(9, rank, ws)                      # ragged shards: 5 + 4\n"
        "out = all_gather_obs(full[:, e0:e0 + c].clone())\n"
        "assert torch.equal(out, full[:, :9])\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        f"open(os.path.join({str(tmp_path)!r}, 'ok_%d' % rank), 'w').write('ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_gloo_world_size_2_shard_plan_gather_equals_the_one_rank_block(tmp_path):
    """What a sharded ``Env`` does on the N > 1 path, on CPU ranks (the Env itself needs a GPU: tests/test_gpu_env_api.py runs the same
    through ``Env.from_xml_path(..., shard=True)`` with two gloo ranks on the GPU box): ``ShardPlan.from_environment`` reads the
    torchrun environment, every rank fills ITS block of a table indexed by the GLOBAL environment index, ``ShardPlan.gather`` returns
    the whole table on every rank - bitwise the one-rank table - for equal (10) and ragged (9 = 5 + 4) global batches; a global batch
    smaller than the world is refused."""
    script = tmp_path / "worker_plan.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from mujoco_template_amd.distributed import ShardPlan, init_process_group\n"
        "assert init_process_group('gloo')\n"
        "for B in (10, 9):\n"
        "    plan = ShardPlan.from_environment(B)\n"
        "    assert plan.world_size == 2 and sum(plan.counts) == B and plan.env0 == (0 if plan.rank == 0 else plan.counts[0])\n"
        "    one = ShardPlan.from_environment(B, rank=0, world_size=1)\n"
        "    table = lambda e0, c: (torch.arange(e0, e0 + c, dtype=torch.float32)[None, :, None] * 1000 + torch.arange(3, dtype=torch.float32)[:, None, None] * 10 + torch.arange(7, dtype=torch.float32)[None, None, :])\n"
        "    full = one.gather(table(one.env0, one.count))               # one rank: the identity\n"
        "    out = plan.gather(table(plan.env0, plan.count))\n"
        "    assert out.shape == (3, B, 7) and torch.equal(out, full), (plan, out.shape)\n"
        "try:\n"
        "    ShardPlan.from_environment(1)\n"
        "    raise SystemExit('a global batch of 1 on 2 ranks must be refused')\n"
        "except ValueError:\n"
        "    pass\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        f"open(os.path.join({str(tmp_path)!r}, 'plan_ok_%d' % plan.rank), 'w').write('ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "plan_ok_0").exists() and (tmp_path / "plan_ok_1").exists()


def test_gloo_single_rank_group_issues_the_collective(tmp_path):
    """single_rank=True (the one-GPU rehearsal switch of bench.py under torchrun): a ONE-rank group is initialised and the all-gather is
    really issued instead of short-circuiting; without RANK in the environment nothing is initialised."""
    script = tmp_path / "worker1.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from mujoco_template_amd.distributed import init_process_group, all_gather_obs\n"
        "assert not init_process_group('gloo')                 # ws == 1, no single_rank: nothing initialised\n"
        "assert init_process_group('gloo', single_rank=True) and dist.get_world_size() == 1\n"
        "calls = []\n"
        "real = dist.all_gather_into_tensor\n"
        "dist.all_gather_into_tensor = lambda out, x: (calls.append(1), real(out, x))[1]\n"
        "x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)\n"
        "assert all_gather_obs(x, counts=[3]) is x and not calls\n"
        "out = all_gather_obs(x, counts=[3], single_rank=True)\n"
        "assert calls == [1] and out is not x and torch.equal(out, x)\n"
        "dist.destroy_process_group()\n"
        f"open(os.path.join({str(tmp_path)!r}, 'ok1'), 'w').write('ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok1").exists()
    env.pop("RANK")
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); "
                        "from mujoco_template_amd.distributed import init_process_group as f; assert not f('gloo', single_rank=True)"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr


def test_recorder_schema_and_trajectory_logger(tmp_path):
    """CSV schema of the reference's StateControlRecorder (logging.py:81-178): joint-major qpos/qvel columns with the
    per-joint-type component suffixes, ctrl columns by actuator name, probes last; TrajectoryLogger row-length check."""
    from mujoco_template_amd import mj
    from mujoco_template_amd.exceptions import ConfigError
    from mujoco_template_amd.logging import build_schema
    from mujoco_template_amd.runtime import TrajectoryLogger
    from tests.conftest import MODELS

    m = mj.MjModel.from_xml_path(MODELS["cartpole"])
    cols, qi, vi = build_schema(m, ["energy"])
    assert cols == ("time_s", "qpos[slider]", "qvel[slider]", "qpos[hinge]", "qvel[hinge]", "ctrl[cart_force]", "energy")
    assert qi == [0, 1] and vi == [0, 1]
    h = mj.MjModel.from_xml_path(MODELS["humanoid"])
    cols, qi, vi = build_schema(h)
    assert cols[1:8] == tuple(f"qpos[root].{s}" for s in ("pos_x", "pos_y", "pos_z", "quat_w", "quat_x", "quat_y", "quat_z"))
    assert cols[8:14] == tuple(f"qvel[root].{s}" for s in ("lin_x", "lin_y", "lin_z", "ang_x", "ang_y", "ang_z"))
    assert len(cols) == 1 + h.nq + h.nv + h.nu and sorted(qi) == list(range(h.nq)) and sorted(vi) == list(range(h.nv))
    assert cols[14].startswith("qpos[") and "." not in cols[14]          # hinge joints: one bare column
    nu0 = mj.MjModel.from_xml_string("<mujoco><worldbody><body><joint type='hinge'/><geom size='0.1'/></body></worldbody></mujoco>")
    assert build_schema(nu0)[0] == ("time_s", "qpos[joint_0]", "qvel[joint_0]", "ctrl[none]")
    path = tmp_path / "sub" / "log.csv"
    with TrajectoryLogger(path, ("a", "b"), lambda r: (r, 2 * r)) as lg:
        assert lg.enabled and lg.log(3) == (3, 6)
        with pytest.raises(ConfigError):
            lg.write_row((1, 2, 3))
    assert path.read_text().splitlines() == ["a,b", "3,6"]
    with pytest.raises(ConfigError):
        TrajectoryLogger(None, (), lambda r: ())


def test_bench_launcher_decision(monkeypatch):
    """`python bench.py --gpus N` from a plain shell starts its own N ranks as children; under torchrun it is a rank."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert "torch" not in vars(bench)                                       # the launcher stays GPU- and torch-free at import
    assert not bench.launcher_needed(1, {})
    assert bench.launcher_needed(2, {})
    assert bench.launcher_needed(8, {"HOME": "/x"})
    assert not bench.launcher_needed(8, {"WORLD_SIZE": "8", "RANK": "3"})  # already one of torchrun's ranks
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd and "29999" in cmd
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert bench.visible_gpu_count() == 2
    assert bench.launch_ranks(4, ["--gpus", "4"]) == 2                       # refuses loudly: fewer devices than ranks, nothing started


def test_bench_wrong_world_size_is_refused():
    """A rank whose WORLD_SIZE differs from --gpus must not silently run a different job (the r1 bench printed n_gpus=1 for --gpus 8)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_batched_manifold_helpers_through_the_c_abi_match_oracle(oracle):
    """mjb_integrate_pos / mjb_differentiate_pos: float64, [batch, nq] <-> [batch, nv], against the oracle's
    mjo_integrate_pos / mjo_differentiate_pos (reference call sites linearization.py:12,67,77; lqr.py:153)."""
    m, d = oracle("humanoid")
    model = mj.MjModel(m.compiled)
    rng = np.random.default_rng(5)
    B = 7
    q1 = np.stack([d.integrate_pos(m.compiled.qpos0, rng.normal(size=model.nv) * 0.5, 1.0) for _ in range(B)])
    v = rng.normal(size=(B, model.nv)) * 2.0
    q2 = q1.copy()
    mj.mj_integratePos(model, q2, v, 0.21)
    ref = np.stack([d.integrate_pos(q1[e], v[e], 0.21) for e in range(B)])
    assert np.abs(q2 - ref).max() < 1e-14
    dv = np.zeros((B, model.nv))
    mj.mj_differentiatePos(model, dv, 0.21, q1, q2)
    refd = np.stack([d.differentiate_pos(q1[e], q2[e], 0.21) for e in range(B)])
    assert np.abs(dv - refd).max() < 1e-13 and np.abs(dv - v).max() < 1e-12
    # a rotation beyond pi comes back as the shorter way round (MuJoCo's mju_subQuat convention)
    big = np.zeros(model.nv); big[3] = 4.0
    qa = np.array(m.compiled.qpos0); qb = qa.copy()
    mj.mj_integratePos(model, qb, big, 1.0)
    back = np.zeros(model.nv)
    mj.mj_differentiatePos(model, back, 1.0, qa, qb)
    assert back[3] == pytest.approx(4.0 - 2 * np.pi, abs=1e-12)
    with pytest.raises(mt.ConfigError):
        mj.mj_integratePos(model, q2.astype(np.float32), v, 0.1)           # in place on float64 only
    with pytest.raises(mt.ConfigError):
        mj.mj_integratePos(model, np.zeros(model.nq + 1), v[0], 0.1)


def test_binary_model_round_trip_and_names_through_the_c_abi(tmp_path):
    """mjb_model_save / mjb_model_load / mjb_model_field_at / mjb_model_name2id: ModelHandle.from_binary_path + save_binary
    (reference model.py:28-31, :45-51) on the engine's own flat table; a host without the Python compiler can load it."""
    model = mj.MjModel.from_xml_path(os.path.join(ROOT, "models", "drone2", "scene.xml"))
    path = str(tmp_path / "drone.mjbm")
    mj.mj_saveModel(model, path, None)
    again = mj.MjModel.from_binary_path(path)
    a, b = model.compiled, again.compiled
    assert (a.nq, a.nv, a.nu, a.nbody, a.npair, a.timestep, a.name) == (b.nq, b.nv, b.nu, b.nbody, b.npair, b.timestep, b.name)
    assert set(a.arrays) == set(b.arrays)
    for k in a.arrays:
        assert np.asarray(a.arrays[k]).shape == np.asarray(b.arrays[k]).shape, k
        assert np.array_equal(np.asarray(a.arrays[k], dtype=float), np.asarray(b.arrays[k], dtype=float)), k
    assert a.names == b.names
    for obj, name in ((mj.mjtObj.mjOBJ_SITE, "imu"), (mj.mjtObj.mjOBJ_BODY, "x2"), (mj.mjtObj.mjOBJ_KEY, "hover"), (mj.mjtObj.mjOBJ_ACTUATOR, "thrust1")):
        assert mj.mj_name2id(again, obj, name) == mj.mj_name2id(model, obj, name) >= 0
        assert mj.mj_id2name(again, obj, mj.mj_name2id(again, obj, name)) == name
    assert mj.mj_name2id(again, mj.mjtObj.mjOBJ_XBODY, "x2") == mj.mj_name2id(again, mj.mjtObj.mjOBJ_BODY, "x2")
    assert mj.mj_id2name(again, mj.mjtObj.mjOBJ_BODY, 10**6) is None
    bad = tmp_path / "bad.mjbm"
    bad.write_bytes(open(path, "rb").read()[:1000])
    with pytest.raises(ValueError, match="truncated|corrupt"):
        mj.MjModel.from_binary_path(str(bad))
    bad.write_bytes(b"not a model")
    with pytest.raises(ValueError, match="magic"):
        mj.MjModel.from_binary_path(str(bad))


def test_binary_model_keeps_edited_options_and_rejects_a_crafted_count(tmp_path):
    """Like mj_saveModel the file carries the options as they are when it is written (disableactuator / iterations / tolerance edited
    after compilation), and a crafted int64 element count must be refused instead of wrapping the byte size."""
    import struct

    model = mj.MjModel.from_xml_path(os.path.join(ROOT, "models", "cartpole.xml"))
    model.opt.iterations = 7
    model.opt.tolerance = 3e-5
    model.opt.disableactuator = 2
    path = str(tmp_path / "cp.mjbm")
    mj.mj_saveModel(model, path, None)
    again = mj.MjModel.from_binary_path(path)
    assert (again.opt.iterations, again.opt.tolerance, again.opt.disableactuator) == (7, 3e-5, 2)
    # corrupt the count of the first field: 2^61 elements of 8 bytes wraps to 0 bytes in size_t arithmetic
    raw = bytearray(open(path, "rb").read())
    nl = struct.unpack_from("<i", raw, 12)[0]
    off = 12 + 4 + nl + 4                                      # magic(8) nfield(4) | namelen(4) name dtype(4) count(8)
    for evil in (1 << 61, (1 << 62) + 1, (1 << 63) - 1):
        struct.pack_into("<q", raw, off, evil)
        bad = tmp_path / "evil.mjbm"
        bad.write_bytes(bytes(raw))
        with pytest.raises(ValueError, match="truncated|corrupt"):
            mj.MjModel.from_binary_path(str(bad))


def test_include_cycles_raise_instead_of_overflowing_the_stack(tmp_path):
    """A self-including file, a two-file cycle and a file included twice are rejected with the compiler's error (MuJoCo rejects
    repeated includes); before, the recursion ran until the process segfaulted."""
    (tmp_path / "self.xml").write_text('<mujoco><include file="self.xml"/><worldbody/></mujoco>')
    (tmp_path / "a.xml").write_text('<mujoco><include file="b.xml"/><worldbody/></mujoco>')
    (tmp_path / "b.xml").write_text('<mujoco><include file="a.xml"/></mujoco>')
    (tmp_path / "part.xml").write_text('<mujoco><worldbody><body><geom size="0.1"/><joint/></body></worldbody></mujoco>')
    (tmp_path / "twice.xml").write_text('<mujoco><include file="part.xml"/><include file="part.xml"/></mujoco>')
    (tmp_path / "once.xml").write_text('<mujoco><include file="part.xml"/></mujoco>')
    for name in ("self.xml", "a.xml", "twice.xml"):
        with pytest.raises(ValueError, match="more than once|cycle"):
            mj.MjModel.from_xml_path(str(tmp_path / name))
    assert mj.MjModel.from_xml_path(str(tmp_path / "once.xml")).nv == 1


def test_module_smoke_cli_fails_loudly_without_a_gpu():
    """No CPU fallback anywhere, the smoke CLI included: without a HIP device it exits non-zero and names the reason."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tests.conftest import MODELS

    r = subprocess.run([sys.executable, "-m", "mujoco_template_amd", MODELS["pendulum"], "--steps", "3", "--zero"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "HIP" in r.stderr or "device" in r.stderr.lower(), r.stderr[-400:]


def test_default_caps_keep_eight_humanoids_per_cu():
    """mjb_data_create's default caps (choose_caps): the largest row cap, then contact cap, whose LDS slice still lets eight fp32
    environments share one CU's 160 KB (two waves per SIMD); small models hold their own worst case.  Read off the specialised
    translation unit, which pins exactly these numbers (no GPU needed)."""
    import re

    from tests.conftest import MODELS

    def caps(name):
        src = mj.MjModel.from_xml_path(MODELS[name])._device_model().spec_source()
        got = dict(re.findall(r"\(m\)\.(n(?:con|efc)_max) == (\d+)", src))
        return int(got["nefc_max"]), int(got["ncon_max"]), int(re.search(r"\(L\)\.bytes == (\d+)", src).group(1))

    ne, nc, lds = caps("humanoid")
    assert (ne, nc) == (64, 23) and 8 * lds <= 160 * 1024 < 8 * (lds + 48)      # one more contact would not fit
    assert caps("cartpole")[:2] == (28, 6) and caps("drone2")[:2] == (82, 20)    # the models' own worst cases: nothing can be dropped
