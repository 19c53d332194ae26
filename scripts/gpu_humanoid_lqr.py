#!/usr/bin/env python3
"""The reference's humanoid example (one-leg balance under LQR) end to end on the batched engine.

Design = the recipe of ``/root/reference/examples/humanoid/controllers/lqr.py:34-145`` with ``humanoid_config.py:63-79``'s numbers,
written against THIS package's mirror of the ``mujoco`` surface (``mt.mj.*``), with the per-sample Python loops turned into batches:

1. height sweep: 2001 offsets of the ``stand_on_left_leg`` keyframe as ONE batch of 2001 environments -> one ``mj_inverse`` call,
   offset = argmin |vertical residual force| (``lqr.py:52-64``);
2. set-point: ``qfrc_inverse`` at that offset, ``ctrl0 = qfrc0 pinv(actuator_moment)`` (``lqr.py:66-85``);
3. ``(A, B)`` by ``linearize_discrete`` (float64 ``k_fd``), COM-over-foot balance cost from ``mj_jacSubtreeCom`` / ``mj_jacBodyCom``
   (``lqr.py:90-110``), ``K`` from scipy's DARE (``lqr.py:112-113``);
4. rollout: ``LinearFeedbackController`` with the law's smoothed ctrl noise (``lqr.py:175-216``), evaluated INSIDE the fused step kernel,
   every environment reading the noise table at its own phase -> ``--batch`` humanoids balancing at once in fp32.

Reported: design numbers, how many environments are still upright after ``--seconds``, and the rollout rate.  Behavioural anchor (not a
numeric one): DeepMind's LQR tutorial, which the reference's example transcribes, shows the humanoid keeping its balance under this
noise with this cost; a humanoid that falls here would point at contact / actuation / linearisation errors the oracle shares.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mujoco_template_amd as mt  # noqa: E402
from mujoco_template_amd import mj  # noqa: E402

CFG = dict(keyframe=1, height_offset_min_m=-1e-3, height_offset_max_m=1e-3, height_samples=2001, linearization_eps=1e-6,
           balance_cost=1000.0, balance_joint_cost=3.0, other_joint_cost=0.3, perturb_seed=1, perturb_duration_s=6.0,
           perturb_ctrl_rate_s=0.8, perturb_balance_std=0.01, perturb_other_std=0.08)


# DeepMind's LQR tutorial itself (the text the reference keeps as LQR.txt:266-300,354-398): balance joints = abdomen + LEFT leg without the
# z axes, noise from the legacy generator ``np.random.seed(1); randn`` over 12 s.  Its published result - the humanoid keeps its balance
# for the 12 s under this noise - is the behavioural known answer of real MuJoCo this script is checked against.
TUTORIAL = dict(CFG, balance="tutorial", rng="legacy", perturb_duration_s=12.0)


def balance_dofs(model, variant: str = "reference") -> np.ndarray:
    """reference: dofs of the joints named *hip* / *knee* / *ankle* (``lqr.py:218-238``); tutorial: abdomen + left leg, no z axes."""
    dofs = []
    for j in range(model.njnt):
        name = model.joint(j).name.lower()
        if int(model.jnt_type[j]) == mj.mjtJoint.mjJNT_FREE:
            continue
        leg = any(t in name for t in ("hip", "knee", "ankle"))
        if variant == "tutorial":
            take = "z" not in name and ("abdomen" in name or ("left" in name and leg))
        else:
            take = leg
        if take:
            dofs.append(int(model.jnt_dofadr[j]))
    return np.array(sorted(dofs), dtype=int)


def noise_table(model, cfg, bal, duration_s: float):
    """The law's smoothed ctrl noise (``lqr.py:175-216`` / LQR.txt:380-398): white noise per actuator, Gaussian window of
    ``perturb_ctrl_rate_s``, unit variance; per-actuator scale by joint class.  Returns (ctrl_std [nu], table [nsteps, nu])."""
    nu = model.nu
    act_dof = np.array([int(model.jnt_dofadr[int(model.actuator_trnid[a][0])]) for a in range(nu)])
    ctrl_std = np.where(np.isin(act_dof, bal), cfg["perturb_balance_std"], cfg["perturb_other_std"])
    dt = float(model.opt.timestep)
    nsteps = max(1, int(np.ceil(duration_s / dt)))
    if cfg.get("rng", "default_rng") == "legacy":
        pert = np.random.RandomState(cfg["perturb_seed"]).randn(nsteps, nu)        # = np.random.seed(1); np.random.randn(nsteps, nu)
        width = int(cfg["perturb_ctrl_rate_s"] / dt)
    else:
        pert = np.random.default_rng(cfg["perturb_seed"]).standard_normal((nsteps, nu))
        width = max(1, int(np.ceil(cfg["perturb_ctrl_rate_s"] / dt)))
    kern = np.exp(-0.5 * np.linspace(-3.0, 3.0, width) ** 2)
    kern /= np.linalg.norm(kern)
    for a in range(nu):
        pert[:, a] = np.convolve(pert[:, a], kern, mode="same")
    return ctrl_std, pert


def design(model, cfg=CFG, verbose=True):
    """Returns dict(qpos0, ctrl0, K, A, B, offset, forces, ctrl_std, perturbations)."""
    import scipy.linalg

    nv, nu = model.nv, model.nu
    offsets = np.linspace(cfg["height_offset_min_m"], cfg["height_offset_max_m"], cfg["height_samples"])
    sweep = mj.MjData(model, batch=len(offsets), dtype="float64")
    mj.mj_resetDataKeyframe(model, sweep, cfg["keyframe"])
    mj.mj_forward(model, sweep)
    sweep.qacc[...] = 0.0
    sweep.qpos[:, 2] += offsets
    mj.mj_inverse(model, sweep)
    forces = np.array(sweep.qfrc_inverse)[:, 2]
    best = int(np.argmin(np.abs(forces)))
    offset = float(offsets[best])
    del sweep

    work = mj.MjData(model, batch=1, dtype="float64")
    mj.mj_resetDataKeyframe(model, work, cfg["keyframe"])
    mj.mj_forward(model, work)
    work.qacc[...] = 0.0
    qp = np.atleast_2d(work.qpos)
    qp[:, 2] += offset
    mj.mj_inverse(model, work)
    qpos0 = np.array(work.qpos, dtype=float).reshape(model.nq)
    qfrc0 = np.array(work.qfrc_inverse, dtype=float).reshape(nv)
    moment = np.zeros((nu, nv))
    mj.mju_sparse2dense(moment, np.reshape(work.actuator_moment, (-1,)), work.moment_rownnz, work.moment_rowadr, np.reshape(work.moment_colind, (-1,)))
    ctrl0 = (np.atleast_2d(qfrc0) @ np.linalg.pinv(moment)).reshape(nu)
    work.qvel[...] = 0.0
    work.ctrl[...] = ctrl0
    mj.mj_forward(model, work)
    A, B = mt.linearize_discrete(model, work, eps=cfg["linearization_eps"])
    jac_com, jac_foot = np.zeros((3, nv)), np.zeros((3, nv))
    mj.mj_jacSubtreeCom(model, work, jac_com, model.body("torso").id)
    mj.mj_jacBodyCom(model, work, jac_foot, None, model.body("foot_left").id)
    jd = jac_com - jac_foot
    bal = balance_dofs(model, cfg.get("balance", "reference"))
    other = np.setdiff1d(np.arange(6, nv), bal)
    Qjoint = np.eye(nv)
    Qjoint[:6, :6] = 0.0
    Qjoint[bal, bal] = cfg["balance_joint_cost"]
    Qjoint[other, other] = cfg["other_joint_cost"]
    Q = np.zeros((2 * nv, 2 * nv))
    Q[:nv, :nv] = cfg["balance_cost"] * (jd.T @ jd) + Qjoint
    R = np.eye(nu)
    P = scipy.linalg.solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
    ctrl_std, pert = noise_table(model, cfg, bal, cfg["perturb_duration_s"])
    rho = float(np.abs(np.linalg.eigvals(A - B @ K)).max())
    if verbose:
        print(f"height sweep: {len(offsets)} offsets in one batched mj_inverse; |f_z| min {abs(forces[best]):.4f} N at offset {offset * 1e3:+.3f} mm "
              f"(f_z at -1 / 0 / +1 mm: {forces[0]:.2f} / {forces[len(forces) // 2]:.2f} / {forces[-1]:.2f} N)")
        print(f"set-point: |qfrc0[root]| max {np.abs(qfrc0[:6]).max():.3e} (unactuated residual), |ctrl0| max {np.abs(ctrl0).max():.4f}, "
              f"realised force error {np.abs(ctrl0 @ moment - qfrc0)[6:].max():.2e}")
        print(f"LQR: balance dofs {bal.tolist()}, |K| max {np.abs(K).max():.2f}, closed-loop spectral radius {rho:.6f} "
              f"(open loop {float(np.abs(np.linalg.eigvals(A)).max()):.4f})")
    return dict(qpos0=qpos0, ctrl0=ctrl0, K=K, A=A, B=B, offset=offset, forces=forces, ctrl_std=ctrl_std, perturbations=pert, rho=rho, bal=bal)


def balance(d, batch: int, seconds: float, dtype: str = "float32", env_stride: int = 7, chunk: int = 200, noise: bool = True, verbose=True):
    """Fused closed-loop rollout from the set-point; returns (upright fraction, min torso height, env-steps/s, env)."""
    ctl = mt.LinearFeedbackController(K=d["K"], ctrl0=d["ctrl0"], qpos_goal=d["qpos0"],
                                      ctrl_noise_std=d["ctrl_std"] if noise else None, perturbations=d["perturbations"] if noise else None,
                                      env_stride=env_stride)
    env = mt.Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), controller=ctl, keyframe=CFG["keyframe"], batch=batch, dtype=dtype)
    env.data.qpos[...] = d["qpos0"]
    env.data.qvel[...] = 0.0
    nsteps = int(round(seconds / float(env.model.opt.timestep)))
    z0 = float(d["qpos0"][2])
    zmin = np.full(batch, z0)
    env.rollout(1)                                       # first launch (kernel load) outside the timing
    t0 = time.perf_counter()
    done = 1
    while done < nsteps:
        n = min(chunk, nsteps - done)
        env.rollout(n)
        done += n
        z = np.array(env.data.qpos).reshape(batch, -1)[:, 2]
        zmin = np.minimum(zmin, z)
    el = time.perf_counter() - t0
    up = zmin > z0 - 0.15                                # a fallen humanoid's torso is > 0.5 m lower
    rate = batch * (nsteps - 1) / el
    if verbose:
        print(f"{dtype} batch {batch}: {nsteps} steps ({seconds:.1f} s simulated){' with ctrl noise' if noise else ''}: upright {int(up.sum())} / {batch}, "
              f"torso height min over run {zmin.min():.4f} m (set-point {z0:.4f}), |qvel| max at end {np.abs(np.array(env.data.qvel)).max():.3f}, "
              f"{rate / 1e6:.2f} M env-steps/s incl. the per-{chunk}-step host check")
    return float(up.mean()), float(zmin.min()), rate, env


def single_env_pair(d, seconds: float, verbose=True):
    """The tutorial's own run (ONE humanoid, the noise table from its start) in float64 and fp32 side by side, fused launches of one
    simulated second; returns (upright64, upright32, max |qpos32 - qpos64| per second)."""
    envs = {}
    for dtype in ("float64", "float32"):
        ctl = mt.LinearFeedbackController(K=d["K"], ctrl0=d["ctrl0"], qpos_goal=d["qpos0"], ctrl_noise_std=d["ctrl_std"],
                                          perturbations=d["perturbations"], env_stride=0)
        env = mt.Env.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"), controller=ctl, keyframe=CFG["keyframe"], batch=1, dtype=dtype)
        env.data.qpos[...] = d["qpos0"]
        env.data.qvel[...] = 0.0
        envs[dtype] = env
    per_s = int(round(1.0 / float(envs["float64"].model.opt.timestep)))
    z0, dev, up = float(d["qpos0"][2]), [], {"float64": True, "float32": True}
    for k in range(int(round(seconds))):
        q = {}
        for dtype, env in envs.items():
            env.rollout(per_s)
            q[dtype] = np.array(env.data.qpos, dtype=float).ravel()
            up[dtype] &= bool(q[dtype][2] > z0 - 0.15)
        dev.append(float(np.abs(q["float32"] - q["float64"]).max()))
        if verbose:
            print(f"  t = {k + 1:2d} s: torso z {q['float64'][2]:.4f} (float64) {q['float32'][2]:.4f} (fp32), joint excursion max "
                  f"{np.abs(q['float64'][7:] - d['qpos0'][7:]).max():.3f} rad, max |qpos32 - qpos64| {dev[-1]:.2e}")
    return up["float64"], up["float32"], dev


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=12.0)
    args = ap.parse_args()
    model = mj.MjModel.from_xml_path(os.path.join(ROOT, "models/humanoid.xml"))
    print("== DeepMind's tutorial recipe (LQR.txt): balance joints = abdomen + left leg, legacy seed-1 noise over 12 s ==")
    t0 = time.perf_counter()
    d = design(model, TUTORIAL)
    print(f"design took {time.perf_counter() - t0:.2f} s")
    balance(d, 16, args.seconds, dtype="float64", noise=False)
    print("one humanoid, the tutorial's noise sequence (published behaviour of real MuJoCo: keeps its balance for the 12 s):")
    u64, u32, dev = single_env_pair(d, args.seconds)
    print(f"  upright after {args.seconds:.0f} s: float64 {u64}, fp32 {u32}; fp32 vs float64 max |dqpos| over the closed-loop run {max(dev):.2e} "
          f"({int(round(args.seconds / float(model.opt.timestep)))} steps, contacts active throughout)")
    # a whole batch: every environment reads ITS OWN window of one long noise sequence (no wrap inside the run)
    stride = 23
    long_s = args.seconds + args.batch * stride * float(model.opt.timestep) + 1.0
    dl = dict(d)
    dl["ctrl_std"], dl["perturbations"] = noise_table(model, TUTORIAL, d["bal"], long_s)
    print(f"{args.batch} humanoids, each under its own {args.seconds:.0f} s window of a {long_s:.0f} s noise sequence (env_stride {stride} steps):")
    nb = min(args.batch, 512)
    *_, e64 = balance(dl, nb, args.seconds, dtype="float64", env_stride=stride)
    *_, e32 = balance(dl, nb, args.seconds, dtype="float32", env_stride=stride)
    z64, z32 = np.array(e64.data.qpos).reshape(nb, -1)[:, 2], np.array(e32.data.qpos).reshape(nb, -1)[:, 2]
    s64, s32 = z64 > d["qpos0"][2] - 0.15, z32 > d["qpos0"][2] - 0.15
    both = s64 & s32
    dq = np.abs(np.array(e32.data.qpos).reshape(nb, -1) - np.array(e64.data.qpos).reshape(nb, -1)).max(axis=1)
    print(f"  the same {nb} noise windows in both precisions: standing at the end float64 {int(s64.sum())}, fp32 {int(s32.sum())}, same outcome in {int((s64 == s32).sum())} / {nb}; "
          f"among the {int(both.sum())} standing in both: fp32 vs float64 |dqpos| median {np.median(dq[both]):.2e}, 90th pct {np.percentile(dq[both], 90):.2e}")
    del e64, e32
    balance(dl, args.batch, args.seconds, dtype="float32", env_stride=stride)
    print("== the reference's variant (lqr.py:218-238: balance joints = every hip / knee / ankle of BOTH legs, the abdomen among the noisy 'other' joints; default_rng(1), 6 s) ==")
    dr = design(model, CFG)
    u64, u32, dev = single_env_pair(dr, 6.0)
    print(f"  upright after 6 s: float64 {u64}, fp32 {u32}")


if __name__ == "__main__":
    main()
