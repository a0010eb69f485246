"""Work counters of one frame of the 102k-triangle scene through the packet walk and per lane (rt_render_stats), per 8x8 tile."""
import sys, os
sys.path.insert(0, os.getcwd())
from raytracing_amd import api, scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 800)
desc = scenes.spheres(20, 4, W, H)
cams, trs = api.camera_sweep(64, desc.cam_pos, desc.cam_rot)
tiles = ((W + 7) // 8) * ((H + 7) // 8)
with api.Scene.from_desc(desc) as sc:
    sc.commit(0, api.ACCEL_LBVH)
    for mode, name in ((api.TRAVERSAL_PACKET, "packet"), (api.TRAVERSAL_PER_LANE, "per lane")):
        sc.set_traversal(mode)
        for pose in (0, 21, 42):
            _, st = sc.render_stats(cams[pose], trs[pose], W, H, desc.depth)
            d = st.as_dict()
            print(name, "pose", pose, {k: round(v / tiles, 1) for k, v in d.items() if isinstance(v, (int, float))}, flush=True)
