#!/usr/bin/env python3
"""Run a script of this repository with module-level flags of coarse3d_amd set first (same-box A/B under a profiler):
    python tools/run_with_flag.py backbone.FUSE_BN_APPLY=0 [more.flags=1 ...] bench.py --steps 3 ...
(rocprofv3 wants the interpreter itself after `--`, so this is a plain in-process runpy, no re-exec)."""
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
while args and "=" in args[0] and not args[0].endswith(".py"):
    name, val = args.pop(0).split("=", 1)
    modname, attr = name.rsplit(".", 1)
    mod = importlib.import_module("coarse3d_amd." + modname)
    cur = getattr(mod, attr)
    setattr(mod, attr, type(cur)(int(val)) if isinstance(cur, (bool, int)) else type(cur)(val))
script = os.path.join(ROOT, args[0]) if not os.path.isabs(args[0]) else args[0]
sys.argv = [script] + args[1:]
runpy.run_path(script, run_name="__main__")
