"""Experiment script (GPU): bench.py's collective modes on RCCL with a ONE-rank process group (all a one-GPU box allows):
the very code the driver's N > 1 runs take -- device tensors, pack -> gather -> unpack and the direct reduce-scatter on one
stream -- minus the peers.  Not a test."""
import os
import subprocess
import sys
env = dict(os.environ, BENCH_FORCE_DIST="1")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in (["--mode", "tiles"], ["--mode", "samples", "--workload", "soup_1m_1080p_4spp"]):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline"] + extra, env=env)
    if r.returncode:
        sys.exit(r.returncode)
