"""``python -m mujoco_template_amd model.xml --steps N --zero``: the reference's smoke run (``mujoco_template/__main__.py:10-33``) on the
batched engine.  Same flags; ``--batch`` / ``--dtype`` choose how many replicas are stepped and in which precision.  With a device-side
controller (``--zero``) the steps run as fused launches (``run_passive_headless``); without a controller the loop is the reference's
``for _ in env.passive(max_steps=...)``."""

from __future__ import annotations

import argparse

from .controllers import ZeroController
from .env import Env
from .runtime import run_passive_headless

# (flag, argparse keywords): the reference's five, then the two this engine adds
_FLAGS = (
    ("xml", dict(help="MJCF file of the model")),
    ("--steps", dict(type=int, default=300, help="physics steps to run")),
    ("--zero", dict(action="store_true", help="drive with ZeroController (evaluated on the device)")),
    ("--groups", dict(type=int, nargs="*", default=None, help="actuator groups left enabled (default: all)")),
    ("--decim", dict(type=int, default=1, help="control decimation, >= 1")),
    ("--batch", dict(type=int, default=1, help="independent replicas stepped together")),
    ("--dtype", dict(default="float32", choices=("float32", "float64"), help="state precision on the device")),
)


def _run(opts: argparse.Namespace) -> int:
    env = Env.from_xml_path(opts.xml, controller=ZeroController() if opts.zero else None, enabled_groups=opts.groups,
                            control_decimation=opts.decim, batch=opts.batch, dtype=opts.dtype)
    if opts.zero and opts.decim == 1:
        return run_passive_headless(env, max_steps=opts.steps)             # fused launches
    return sum(1 for _ in env.passive(max_steps=opts.steps))               # the reference's loop, one Env.step per step


def main() -> None:
    cli = argparse.ArgumentParser(description="Batched MuJoCo-template smoke test on MI355X (fail-fast)")
    for flag, kw in _FLAGS:
        cli.add_argument(flag, **kw)
    print(f"Completed {_run(cli.parse_args())} steps.")


if __name__ == "__main__":
    main()
