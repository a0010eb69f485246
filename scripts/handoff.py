"""end-to-end ms per 800x800 frame of the hand-off variants (bench.py hand_off), for A/B runs of library builds."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from raytracing_amd import api
class A: size=None; frames_per_launch=0; traversal="auto"; kernel="auto"; tiles_per_wave=0
w = bench.Workload(api, torch, "room800", 0, A)
print(json.dumps({k: v for k, v in bench.hand_off(api, torch, w).items() if k != "note"}))
