"""
Launcher: run an UNMODIFIED example script of the reference against the MI355X backend.

    python -m graphbasedlocaltrajectoryplanner_amd.run [--ticks N] [--device D] [--mode planner|seams] [--extra-path DIR] main_std_example.py

What it does, in this order:
  1. puts the script's directory (the reference checkout: it holds the ``graph_ltpl`` package, ``params/``, ``inputs/``)
     and any ``--extra-path`` at the front of ``sys.path``;
  2. restores the two NumPy aliases the reference (pinned to numpy 1.18) still uses -- ``np.Inf``
     (main_online_path_gen.py:96) and ``np.object`` (main_offline_callback.py:160) -- when running on NumPy 2;
  3. imports ``graph_ltpl`` and calls ``install()`` (seams (1) and (2) now end in libltpl_hip.so);
  4. optionally bounds the stock ``while True`` loop (main_std_example.py:98) by raising ``SystemExit`` after ``N`` calls
     of ``Graph_LTPL.calc_vel_profile`` (one per tick);
  5. executes the script with ``runpy`` as ``__main__``.

The backend is selected by the launcher, not by an INI key, because the md5 of the offline INI keys the reference's
graph cache (main_offline_callback.py:57-68).
"""
import argparse
import os
import runpy
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m graphbasedlocaltrajectoryplanner_amd.run")
    ap.add_argument("--ticks", type=int, default=0, help="stop after N planning ticks (0 = run the script's own loop)")
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--mode", choices=("planner", "seams"), default="planner",
                    help="planner = the OnlineTrajectoryHandler state machine runs in C++ behind ltpl_planner_* (default); "
                         "seams = only main_online_path_gen / VpForwardBackward are replaced")
    ap.add_argument("--extra-path", action="append", default=[], help="additional sys.path entries (dependencies)")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)

    script = os.path.abspath(args.script)
    ref_root = os.path.dirname(os.path.realpath(script))
    for p in [ref_root] + [os.path.abspath(p) for p in args.extra_path]:
        if p not in sys.path:
            sys.path.insert(0, p)

    import numpy as np
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    if not hasattr(np, "object"):
        np.object = object

    import graph_ltpl
    from .install import install
    session = install(graph_ltpl, device=args.device, mode=args.mode)

    cls = graph_ltpl.Graph_LTPL.Graph_LTPL
    orig = cls.calc_vel_profile
    if args.ticks > 0:
        state = {"n": 0}

        def counted(self, *a, **kw):
            out = orig(self, *a, **kw)
            state["n"] += 1
            if state["n"] >= args.ticks:
                print("ltpl-hip launcher: %d ticks done" % state["n"])
                raise SystemExit(0)
            return out
        cls.calc_vel_profile = counted

    sys.argv = [script] + list(args.script_args)
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        cls.calc_vel_profile = orig          # the tick counter must not outlive the script (in-process callers, tests)
    return session


if __name__ == "__main__":
    main()
