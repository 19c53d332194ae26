"""What do the scheduler's code paths cost the STATIC map?  Spec-kernel variants (MJB_SPEC_FLAGS) timed with the static map, fair off."""
import sys, os, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from mujoco_template_amd.mjcf import compile_xml_path
    from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
    dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
    for B in (2048, 4096, 4097):
        if B == 4097:                                              # 4096 again with the engine's own policy (ticket map)
            B = 4096; os.environ.pop("MJB_CHUNK_STEPS", None); os.environ.pop("MJB_FAIR_BIT", None)
        sim = BatchSim(dm, B, dtype="float32")
        sim.rollout(100, CTRL_RANDOM, seed=1); sim.sync()
        ts = []
        for r in range(6):
            t = time.perf_counter(); sim.rollout(100, CTRL_RANDOM, seed=1, step0=100 + 100 * r); sim.sync(); ts.append(time.perf_counter() - t)
        print(f"   B={B} [{sim.schedule_info()['map']}]: best {min(ts)*1e3:.3f} ms median {sorted(ts)[3]*1e3:.3f} ms per 100-step launch", flush=True)
    sys.exit(0)
for flags in os.environ.get("ABLATE_FLAGS", "|-DMJB_NO_TICKETS").split("|"):
    env = dict(os.environ, MJB_SPEC_FLAGS=flags, MJB_CHUNK_STEPS="0", MJB_FAIR_BIT="0")
    print(f"spec flags '{flags}' (static map, fair off):", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
