import sys; sys.path.insert(0, '/root/repo')
import numpy as np
import mujoco_template_amd as mt
from tests.conftest import MODELS
rng = np.random.default_rng(4)
for dtype in ("float32", "float64"):
    h = mt.ModelHandle.from_xml_path(MODELS["drone2"], batch=2, dtype=dtype)
    m, d = h.model, h.data
    q = np.array(d.qpos); q[:, 2] += 0.6
    d.qpos[:] = q; d.qvel[:] = rng.normal(size=d.qvel.shape) * 0.3; d.ctrl[:] = 2.0
    h.forward()
    A1, B1 = mt.linearize_discrete(m, d, use_native=True)
    A2, B2 = mt.linearize_discrete(m, d, use_native=False)
    dA = np.abs(A1 - A2); i = np.unravel_index(dA.argmax(), dA.shape)
    print(dtype, "max dA", dA.max(), "at", i, A1[i], A2[i], "max dB", np.abs(B1 - B2).max())
    np.set_printoptions(precision=5, linewidth=200, suppress=True)
    print((A1 - A2)[0])
