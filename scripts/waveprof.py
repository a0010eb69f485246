"""Per-wave stage counters of a render (RT_PROFILE build, RT355_LIB=build/librt355_prof.so): usage waveprof.py <scene> [w h]."""
import sys, os
sys.path.insert(0, os.getcwd())
from raytracing_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "room"
desc = scenes.get(name) if name != "mirrors" else scenes.mirrors()
if name.startswith("spheres"):
    desc = scenes.spheres(20, 4, 800, 800)
if len(sys.argv) > 3:
    desc = desc.with_size(int(sys.argv[2]), int(sys.argv[3]))
accel = api.ACCEL_LBVH if name.startswith('spheres') else api.ACCEL_BRUTE
with api.Scene.from_desc(desc) as sc:
    sc.commit(0, accel)
    _, st = sc.render_stats(desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth, want_frame=False)
d = st.as_dict()
names = ["trace", "group", "inbox", "tri", "maybe", "cand", "c0", "c1", "accept", "iter"]
vals = list(d.values())
waves = ((desc.width + 7) // 8) * ((desc.height + 7) // 8)
print(name, "waves", waves)
for n, v in zip(names, vals):
    print("  %-7s %10d  per wave %.1f" % (n, v, v / waves))
