"""The long form of test_gpu_fuzz.py (not collected by pytest; run by hand on a GPU box):
    python tests/fuzz_vs_oracle.py [n_seeds] [first_seed]
every render path of render_all_paths() on n random scenes against the ORACLE's frame, bit for bit — the check to run
after a change to the arithmetic every path shares (the paths agreeing among themselves, scripts/fuzz_paths.py, says
nothing about that)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as oracle
from test_gpu_fuzz import fuzz_case, render_all_paths

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad = 0
for seed in range(first, first + n_seeds):
    desc = fuzz_case(seed, max_tris=1000, max_w=128, max_h=96)
    want = oracle.render_desc(desc, threads=16).view(np.uint32)
    frames, _ = render_all_paths(desc)
    for label, f in frames:
        diff = int((want != f.view(np.uint32)).any(axis=2).sum())
        if diff:
            bad += 1
            print("seed", seed, label, "differs from the oracle in", diff, "pixels", flush=True)
    if (seed - first) % 100 == 0:
        print("seed", seed, "ok so far, mismatches", bad, flush=True)
print("done:", n_seeds, "scenes x", len(frames), "paths against the oracle,", bad, "mismatches")
sys.exit(1 if bad else 0)
