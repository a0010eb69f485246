"""Wall time of rt_scene_commit for the 102k-triangle scene, three commits (the first loads code objects); with the experiment build
and RT355_LGRID_TIMES=1 the library prints its own laps to stderr."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from raytracing_amd import api, scenes
desc = scenes.spheres(20, 4, 800, 800)
with api.Scene.from_desc(desc) as sc:
    for k in range(4):
        t = time.perf_counter()
        sc.commit(0, api.ACCEL_LBVH)
        print("commit %d: %.2f ms wall, build_ms %.2f" % (k, (time.perf_counter() - t) * 1e3, sc.build_ms(0)), flush=True)
        sys.stderr.write("----\n"); sys.stderr.flush()
