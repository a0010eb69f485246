"""ctypes binding of oracle/librt_oracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (raytracing_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")
TRI_FLOATS = 28


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_bounce", "rays_shadow", "n_plane", "n_point", "n_edge",
        "px_hit", "px_mirror", "px_lit")]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self) -> int:
        return int(self.rays_primary + self.rays_bounce + self.rays_shadow)


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so"])
    return _LIB_PATH


_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_triangle_init.argtypes = [_fp, _fp, _fp, _fp, C.c_float, _fp]
        L.orc_quad_half.argtypes = [_fp, _fp, _fp, _fp, C.c_float, _fp]
        L.orc_camera_trans.argtypes = [_fp, _fp]
        L.orc_camera_init.argtypes = [_fp, _fp, _fp]
        L.orc_camera_step.argtypes = [_fp, C.POINTER(C.c_int), C.c_float]
        L.orc_cast_ray.argtypes = [_fp, C.c_int, _fp, _fp, C.c_int, _fp, C.POINTER(Counters)]
        L.orc_cast_ray.restype = C.c_int
        L.orc_render.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp,
                                 C.POINTER(Counters)]
        L.orc_render_rows.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, _fp, C.POINTER(Counters)]
        L.orc_render_mt.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                    C.c_int, _fp, C.POINTER(Counters)]
        L.orc_render_glsl.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, _fp]
        L.orc_quantize.argtypes = [_fp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.orc_flops.argtypes = [C.POINTER(Counters)]
        L.orc_flops.restype = C.c_double
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a: np.ndarray):
    return a.ctypes.data_as(_fp)


def triangle_init(p1, p2, p3, color, shin) -> np.ndarray:
    out = np.zeros(TRI_FLOATS, dtype=np.float32)
    a, b, c, d = _f32(p1), _f32(p2), _f32(p3), _f32(color)
    lib().orc_triangle_init(_p(a), _p(b), _p(c), _p(d), float(shin), _p(out))
    return out


def quad_half(t, r, s, color, shin) -> np.ndarray:
    out = np.zeros(TRI_FLOATS, dtype=np.float32)
    a, b, c, d = _f32(t), _f32(r), _f32(s), _f32(color)
    lib().orc_quad_half(_p(a), _p(b), _p(c), _p(d), float(shin), _p(out))
    return out


def camera_trans(rot_deg) -> np.ndarray:
    out = np.zeros(16, dtype=np.float32)
    r = _f32(rot_deg)
    lib().orc_camera_trans(_p(r), _p(out))
    return out


def camera_run(start6, ms, keys10) -> np.ndarray:
    """The reference's Camera driven like its frame loop (Src:1294-1295, 1307-1308): per frame the
    17 state floats after the update + the next frame's trans -> (F, 33)."""
    cam = np.zeros(17, dtype=np.float32)
    s = _f32(start6)
    pos, rot = np.ascontiguousarray(s[:3]), np.ascontiguousarray(s[3:6])
    lib().orc_camera_init(_p(cam), _p(pos), _p(rot))
    out = np.zeros((len(ms), 33), dtype=np.float32)
    for i in range(len(ms)):
        k = (C.c_int * 10)(*[int(x) for x in keys10[i]])
        lib().orc_camera_step(_p(cam), k, float(np.float32(ms[i])))
        out[i, :17] = cam
        out[i, 17:] = camera_trans(cam[3:6])
    return out


def build_scene(desc) -> Tuple[np.ndarray, np.ndarray]:
    """SceneDesc -> (tris (N,28) f32, lights (L,4) f32) through the oracle's own constructors."""
    tris, lights = [], []
    for kind, v in desc.ops:
        if kind == "quadhalf":
            tris.append(quad_half(v[0:3], v[3:6], v[6:9], v[9:12], v[12]))
        elif kind == "tri":
            tris.append(triangle_init(v[0:3], v[3:6], v[6:9], v[9:12], v[12]))
        elif kind == "light":
            lights.append(np.asarray(v, dtype=np.float32))
    T = np.stack(tris) if tris else np.zeros((0, TRI_FLOATS), np.float32)
    Ls = np.stack(lights) if lights else np.zeros((0, 4), np.float32)
    return np.ascontiguousarray(T, dtype=np.float32), np.ascontiguousarray(Ls, dtype=np.float32)


def cast_ray(tris, origin, direction, skip=-1):
    tris = _f32(tris)
    o, d = _f32(origin), _f32(direction)
    out = np.zeros(6, dtype=np.float32)
    lib().orc_cast_ray(_p(tris), tris.shape[0], _p(o), _p(d), int(skip), _p(out), None)
    return out


def render(tris, lights, cam_pos, trans16, width, height, depth=1, threads=1,
           counters: bool = False, rows: Optional[Tuple[int, int, int]] = None):
    """Full frame (H,W,3) float32.  rows=(y0,y1,step) renders only those rows (others stay 0)."""
    tris, lights = _f32(tris), _f32(lights)
    cam, tr = _f32(cam_pos), _f32(trans16)
    out = np.zeros((height, width, 3), dtype=np.float32)
    ctr = Counters() if counters else None
    cp = C.byref(ctr) if counters else None
    n, nl = tris.shape[0], lights.shape[0]
    if rows is not None:
        lib().orc_render_rows(_p(tris), n, _p(lights), nl, _p(cam), _p(tr), width, height, depth,
                              rows[0], rows[1], rows[2], _p(out), cp)
    elif threads > 1:
        lib().orc_render_mt(_p(tris), n, _p(lights), nl, _p(cam), _p(tr), width, height, depth,
                            threads, _p(out), cp)
    else:
        lib().orc_render(_p(tris), n, _p(lights), nl, _p(cam), _p(tr), width, height, depth, _p(out), cp)
    return (out, ctr) if counters else out


def render_glsl(tris, lights, cam_pos, trans16, width, height):
    """Full frame (H,W,3) float32 by the restated compute shader (Compute.shader:140-345; parity unpinned)."""
    tris, lights = _f32(tris), _f32(lights)
    cam, tr = _f32(cam_pos), _f32(trans16)
    out = np.zeros((height, width, 3), dtype=np.float32)
    lib().orc_render_glsl(_p(tris), tris.shape[0], _p(lights), lights.shape[0], _p(cam), _p(tr), width, height, _p(out))
    return out


def render_desc(desc, threads=1, counters=False):
    tris, lights = build_scene(desc)
    trans = camera_trans(desc.cam_rot)
    return render(tris, lights, desc.cam_pos, trans, desc.width, desc.height, desc.depth,
                  threads=threads, counters=counters)


def quantize(rgb) -> np.ndarray:
    a = _f32(rgb)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().orc_quantize(_p(a), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def flops(ctr: Counters) -> float:
    return float(lib().orc_flops(C.byref(ctr)))
