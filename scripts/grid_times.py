"""us/frame of the 102k-triangle scene with the shadow grids on (and off), both traversals, 800x800 and 3840x2160; prints the grid's size.
RT355_LGRID_CELL (experiment builds) sets the cell size in mean triangle sizes as seen from the light."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import api, scenes
from scripts.build_times import timed


def main():
    desc = scenes.spheres(20, 4, 800, 800)
    with api.Scene.from_desc(desc) as sc:
        for on in ((True, False) if "--off" in sys.argv else (True,)):
            sc.set_shadow_grid(on)
            sc.commit(0, api.ACCEL_LBVH)
            print("grid", on, sc.shadow_grid(0, 0), flush=True)
            for (w, h, fpl) in ((800, 800, 32), (3840, 2160, 16)):
                cams, trs = api.camera_sweep(fpl, desc.cam_pos, desc.cam_rot)
                for mode, mname in ((api.TRAVERSAL_PER_LANE, "per lane"), (api.TRAVERSAL_PACKET, "packets")):
                    sc.set_traversal(mode)
                    print("   %dx%d %-9s %8.1f us/frame" % (w, h, mname, timed(sc, cams, trs, w, h, desc.depth)), flush=True)


if __name__ == "__main__":
    main()
