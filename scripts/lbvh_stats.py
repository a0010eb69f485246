"""Commit and re-commit wall time of the 102k-triangle scene and the work counters of one frame through the tree, per ray."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from raytracing_amd import api, scenes
desc = scenes.spheres(20, 4, 800, 800)
with api.Scene.from_desc(desc) as sc:
    t0 = time.time(); sc.commit(0, api.ACCEL_LBVH); print("commit s", time.time() - t0, "build ms", sc.build_ms(0))
    for rep in range(3):
        t0 = time.time(); sc.commit(0, api.ACCEL_LBVH); print("re-commit s %.4f build ms %.3f" % (time.time() - t0, sc.build_ms(0)))
    _, st = sc.render_stats(desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth, want_frame=False)
d = st.as_dict(); r = st.rays
print(d)
print("rays", r, "box/ray %.1f leaf/ray %.1f point/ray %.1f edge/ray %.2f flops/ray %.0f" % (d['n_box']/r, d['n_plane']/r, d['n_point']/r, d['n_edge']/r, st.flops/r))
