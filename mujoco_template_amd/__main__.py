"""``python -m mujoco_template_amd model.xml --steps N --zero``: the reference's smoke run (``mujoco_template/__main__.py:10-33``) on the
batched engine.  Same flags; ``--batch`` / ``--dtype`` choose how many replicas are stepped and in which precision.  With a device-side
controller (``--zero``) the steps run as fused launches (``run_passive_headless``); without a controller the loop is the reference's
``for _ in env.passive(max_steps=...)``."""

from __future__ import annotations

import argparse

from .controllers import ZeroController
from .env import Env
from .runtime import run_passive_headless


def main() -> None:
    parser = argparse.ArgumentParser(description="Batched MuJoCo-template smoke test on MI355X (fail-fast)")
    parser.add_argument("xml", help="Path to the MJCF XML")
    parser.add_argument("--steps", type=int, default=300)
    parser.add_argument("--zero", action="store_true", help="Use ZeroController (evaluated on the device)")
    parser.add_argument("--groups", type=int, nargs="*", default=None, help="Enable only these actuator groups")
    parser.add_argument("--decim", type=int, default=1, help="Control decimation (>=1)")
    parser.add_argument("--batch", type=int, default=1, help="Independent replicas stepped together")
    parser.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    args = parser.parse_args()

    env = Env.from_xml_path(args.xml, controller=ZeroController() if args.zero else None, enabled_groups=args.groups,
                            control_decimation=args.decim, batch=args.batch, dtype=args.dtype)
    if args.zero and args.decim == 1:
        steps = run_passive_headless(env, max_steps=args.steps)
    else:
        steps = 0
        for _ in env.passive(max_steps=args.steps):
            steps += 1
    print(f"Completed {steps} steps.")


if __name__ == "__main__":
    main()
