"""Differential fuzz by the thousand: brute force vs LBVH per lane vs LBVH packets, lockstep and state-machine
kernels, on the random scenes of tests/test_gpu_fuzz.py (frames must agree bit for bit).
usage: python scripts/fuzz_paths.py [n_seeds]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_fuzz import fuzz_case, render_all_paths

bad = 0
hoisted_scenes = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(1000, 1000 + n_seeds):
    frames, hoisted = render_all_paths(fuzz_case(seed))
    hoisted_scenes += hoisted > 0
    a = frames[0][1].view(np.uint32)
    for label, f in frames[1:]:
        if not np.array_equal(a, f.view(np.uint32)):
            bad += 1
            print("seed", seed, label, "differs in", int((a != f.view(np.uint32)).any(axis=2).sum()), "pixels", flush=True)
    if seed % 500 == 0:
        print("seed", seed, "ok so far, mismatches", bad, flush=True)
print("done:", n_seeds, "scenes x", len(frames), "paths,", hoisted_scenes, "with hoisted triangles,", bad, "mismatches")
sys.exit(1 if bad else 0)
