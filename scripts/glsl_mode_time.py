"""us/frame of RT_SHADE_GLSL (state-machine kernel) against the default mode, room 800x800, 32 frames per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import api, scenes

desc = scenes.room(800, 800)
W = H = 800
with api.Scene.from_desc(desc) as sc:
    sc.commit(0, api.ACCEL_BRUTE)
    cams, trs = api.camera_path(desc.cam_pos, desc.cam_rot, 32) if hasattr(api, "camera_path") else (None, None)
    if cams is None:
        cams = np.tile(np.asarray(desc.cam_pos, np.float32), (32, 1))
        trs = np.tile(api.camera_trans(desc.cam_rot).astype(np.float32).reshape(1, 16), (32, 1))
    buf = torch.empty(32 * H * W * 3, dtype=torch.float32, device="cuda:0")
    for name, mode in (("default", api.STRICT), ("glsl", api.STRICT | api.SHADE_GLSL)):
        for _ in range(3):
            sc.render_frames_device(0, cams, trs, W, H, 1, mode, buf.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            sc.render_frames_device(0, cams, trs, W, H, 1, mode, buf.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        print(name, "%.1f us/frame" % (e0.elapsed_time(e1) * 1000 / 320))
