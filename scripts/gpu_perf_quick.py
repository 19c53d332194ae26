"""A/B timing of specialised-kernel build variants in ONE process on ONE box (box-to-box spread is +-2 %): for every variant (a string of
extra compile flags for the per-model specialised kernels, MJB_SPEC_FLAGS) the humanoid rollout at B = 4096 (ticket map; 100-step and
20-step launches) and at B = 512 / 1024 (the small-batch kernel), best and median of 7 launches, plus a checksum of the final state
(variants that only change code generation must agree bit for bit).

    python scripts/gpu_perf_quick.py "" "-DMJB_NO_MASK_OPAQUE" ...
"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_template_amd.mjcf import compile_xml_path
from mujoco_template_amd._capi import BatchSim, DeviceModel, CTRL_RANDOM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1:] or [""]
dm = DeviceModel(compile_xml_path(os.path.join(ROOT, "models/humanoid.xml")))
for rep in range(2):                                            # two rounds over the variants: drift of the box shows as a difference between rounds
    for flags in variants:
        os.environ["MJB_SPEC_FLAGS"] = flags
        line = f"[{flags or 'default':28s}]"
        for B, n in ((4096, 100), (4096, 20), (512, 100), (1024, 100)):
            sim = BatchSim(dm, B, dtype="float32")
            sim.rollout(150, CTRL_RANDOM, seed=1); sim.sync()
            ts = []
            for r in range(7):
                t = time.perf_counter(); sim.rollout(n, CTRL_RANDOM, seed=1, step0=150 + n * r); sim.sync(); ts.append(time.perf_counter() - t)
            crc = zlib.crc32(np.ascontiguousarray(sim.get("qpos")).tobytes())
            line += f"  B={B} x{n}: {B * n / min(ts) / 1e6:6.2f} M/s (med {B * n / np.median(ts) / 1e6:6.2f}; {min(ts) / n * 1e6:6.2f} us/step) crc {crc:08x}"
            del sim
        print(line, flush=True)
