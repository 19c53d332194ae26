"""Register budget of the specialised step kernel vs batch: MJB_SPEC_FLAGS=-DMJB_WPS=1 lets the compiler use 512 VGPRs (one wave per SIMD)."""
import sys, os, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from mujoco_template_amd.mjcf import compile_xml_path
    from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
    dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
    for B in (256, 512, 1024, 2048):
        sim = BatchSim(dm, B, dtype="float32")
        sim.rollout(100, CTRL_RANDOM, seed=1); sim.sync()
        ts = []
        for r in range(5):
            t = time.perf_counter(); sim.rollout(100, CTRL_RANDOM, seed=1, step0=100 + 100 * r); sim.sync(); ts.append(time.perf_counter() - t)
        print(f"   B={B}: best {min(ts)*1e3:.3f} ms per 100-step launch = {B*100/min(ts)/1e6:.2f} M env-steps/s", flush=True)
    sys.exit(0)
for flags in os.environ.get("WPS_FLAGS", "|-DMJB_WPS=1").split("|"):
    print(f"spec flags '{flags}':", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MJB_SPEC_FLAGS=flags), check=False)
