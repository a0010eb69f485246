"""102k triangles at 640x360, depth 4: per-lane == packets == brute force, bit for bit (quick check on a GPU box)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import api, scenes
desc = scenes.spheres(20, 4, 640, 360, depth=4)
fr = []
with api.Scene.from_desc(desc) as sc:
    tr = api.camera_trans(desc.cam_rot)
    sc.commit(0, api.ACCEL_LBVH)
    for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
        sc.set_traversal(mode); fr.append(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth))
    sc.set_traversal(api.TRAVERSAL_AUTO)
    sc.commit(0, api.ACCEL_BRUTE)
    fr.append(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth))
print("per-lane == packet:", np.array_equal(fr[0].view(np.uint32), fr[1].view(np.uint32)), " packet == brute:", np.array_equal(fr[1].view(np.uint32), fr[2].view(np.uint32)), "lit", int((fr[2].sum(axis=2) > 0).sum()))
