import sys, os
sys.path.insert(0, os.getcwd())
from raytracing_amd import api, scenes
for (w, h, tpw) in [(8, 8, 1), (16, 8, 2), (32, 8, 4), (64, 8, 8), (64, 64, 64)]:
    api.lib().rt_set_tiles_per_wave(tpw)
    desc = scenes.room(w, h)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        for rep in range(2):
            print(w, h, "tpw", tpw, "rep", rep, flush=True)
            sc.render_stats(desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth, want_frame=False)
