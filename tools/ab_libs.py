"""Experiment script (GPU): the same frames with different builds of the library (RT_AMD_LIB), one process per build. Not a test."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs, rest = sys.argv[1].split(","), sys.argv[2:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default": env["RT_AMD_LIB"] = os.path.join(root, lib)
        print("#### lib", lib, flush=True)
        subprocess.run([sys.executable, "-u", os.path.join(root, "tools", "ab_traverse.py")] + rest, env=env, check=True, timeout=400)
