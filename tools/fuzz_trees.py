"""Experiment script (GPU): the fuzz case of tests/test_gpu_paths.py over many more seeds. Not a test."""
import os, sys, traceback
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle"), os.path.join(_ROOT, "tests")]
import test_gpu_paths as t
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, blank = [], []
for seed in range(lo, hi):
    try:
        t.test_fuzz_wide_tree_binary_tree_and_reference_loop_agree(seed)
    except AssertionError as e:
        bad.append(seed)
        tb = traceback.extract_tb(e.__traceback__)[-1]
        if "the camera does see the soup" in (tb.line or ""):
            bad.pop(); blank.append(seed)          # a frame that sees (almost) nothing: not a parity failure
        else:
            print("seed", seed, "FAILED at line", tb.lineno, ":", (tb.line or "")[:120], flush=True)
    if seed % 50 == 0: print("seed", seed, "done", flush=True)
print("parity failures:", bad, "| blank frames (skipped):", len(blank), "of", hi - lo)
