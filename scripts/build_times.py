"""Tree build times of the 102k-triangle scene: RT_BUILD_FAST and RT_BUILD_SAH, three commits each (the first is cold)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from raytracing_amd import api, scenes
d = scenes.spheres(20, 4, 800, 800)
for q, name in ((api.BUILD_FAST, "fast"), (api.BUILD_SAH, "sah")):
    with api.Scene.from_desc(d) as sc:
        sc.set_build(q)
        for rep in range(3):
            t = time.perf_counter()
            sc.commit(0, api.ACCEL_LBVH)
            wall = (time.perf_counter() - t) * 1e3
            print(name, "commit %d: build_ms %.2f, commit wall %.1f ms, depth %s" % (rep, sc.build_ms(0), wall, sc.bvh_depth(0)), flush=True)
            sc.addLight((0, 0, 0), 1.0)       # touch the scene so the next commit rebuilds
