"""ctypes binding of librt355.so (include/rt355.h) with the reference's own vocabulary.

The reference has no plugin API: its "interface" is ``Triangle(p1,p2,p3,color,shininess)``
(Source.cpp:45), ``addTriangle(translate, rotate, scale, color, shininess)`` (Source.cpp:410),
``lights.push_back({pos, intensity})`` (Source.cpp:880), ``Camera(pos, rot)`` (Source.cpp:810),
the per-frame ``trans`` (Source.cpp:1145-1147) and the loop over ``getPixelData`` that fills
``float data[h*w*3]`` (Source.cpp:1248-1264).  :class:`Scene` offers exactly those, forwarding
every call to the C ABI; nothing is computed in Python and there is no CPU fallback — a missing
library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT355_LIB") or os.path.join(_HERE, "librt355.so")   # RT355_LIB: A/B builds of the same library

RT_OK = 0
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_LBVH = 0, 1, 2
TRAVERSAL_AUTO, TRAVERSAL_PER_LANE, TRAVERSAL_PACKET = 0, 1, 2
KERNEL_AUTO, KERNEL_STATE_MACHINE, KERNEL_LOCKSTEP = 0, 1, 2
BUILD_FAST, BUILD_SAH, BUILD_PLOC = 0, 1, 2
PRIMARY_AUTO, PRIMARY_WALK, PRIMARY_RASTER = 0, 1, 2
STRICT = 0            # RT_RENDER_STRICT: the only render mode (bit-identical frames)
OUT_RGBA8 = 0x100     # RT_OUT_RGBA8: the output buffer takes packed 8-bit pixels instead of three floats
SHADE_GLSL = 0x200    # RT_SHADE_GLSL: the colour arithmetic of the reference's compute shader (Compute.shader:181-219, 330)
STACK_LIMIT = 64      # RT_STACK_LIMIT
NOALLOC_MAX_RAYS = 16384
TRI_FLOATS = 28

ERROR_NAMES = {-1: "RT_ERR_INVALID", -2: "RT_ERR_STATE", -3: "RT_ERR_DEVICE", -4: "RT_ERR_NOMEM",
               -5: "RT_ERR_IO", -6: "RT_ERR_LIMIT"}

# every symbol include/rt355.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "rt_scene_create", "rt_scene_destroy", "rt_scene_clear", "rt_scene_add_triangle",
    "rt_scene_add_triangles", "rt_scene_add_quad_half", "rt_scene_add_light", "rt_scene_load_rto",
    "rt_scene_triangle_count", "rt_scene_light_count", "rt_scene_get_triangles", "rt_scene_commit",
    "rt_scene_build_ms", "rt_scene_get_bvh", "rt_scene_get_hoisted", "rt_scene_accel", "rt_scene_bvh_depth", "rt_camera_trans", "rt_camera_init", "rt_camera_rotate",
    "rt_camera_translate", "rt_camera_frame_time", "rt_camera_step", "rt_camera_path", "rt_render", "rt_render_device",
    "rt_render_frames_device", "rt_render_tiles_device", "rt_shard_rows", "rt_render_tiles", "rt_scatter_shards", "rt_scatter_shards_host", "rt_render_stats", "rt_render_multi", "rt_cast_rays",
    "rt_render_async", "rt_render_wait", "rt_frame_alloc", "rt_frame_free", "rt_cast_rays_noalloc",
    "rt_device_count", "rt_device_props", "rt_scene_set_tiles_per_wave", "rt_scene_set_traversal", "rt_scene_set_kernel", "rt_scene_set_build", "rt_scene_build_kind", "rt_scene_set_shadow_grid", "rt_scene_shadow_grid", "rt_scene_set_primary_raster", "rt_scene_traversal", "rt_last_error", "rt_version",
)


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "RT_ERR"), code, msg))
        self.code = code


class CameraState(C.Structure):
    """rt_camera (rt355.h)."""
    _fields_ = [("position", C.c_float * 3), ("rotations", C.c_float * 3), ("direction", C.c_float * 3),
                ("right", C.c_float * 3), ("up", C.c_float * 3), ("move_speed", C.c_float), ("rot_speed", C.c_float)]


class Stats(C.Structure):
    """rt_stats — work counters in the units SURVEY.md §8(d) prices."""
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_bounce", "rays_shadow", "n_plane", "n_point", "n_edge", "n_box",
        "px_hit", "px_mirror", "px_lit")]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self) -> int:
        return int(self.rays_primary + self.rays_bounce + self.rays_shadow)

    @property
    def flops(self) -> float:
        """14 plane + 6 point + 17 edge + 22 box + 32/29/30 per primary/bounce/shadow ray."""
        return (14.0 * self.n_plane + 6.0 * self.n_point + 17.0 * self.n_edge + 22.0 * self.n_box +
                32.0 * self.rays_primary + 29.0 * self.rays_bounce + 30.0 * self.rays_shadow)


_fp = C.POINTER(C.c_float)
_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME
    as /opt/rocm's); whichever copy is loaded first must serve both torch and librt355.so, or the
    second copy finds "no HIP GPUs".  If torch is installed but not imported yet, load ITS copy
    first so that librt355.so's DT_NEEDED entry binds to it and a later `import torch` re-uses it."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load librt355.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C raytracing_amd/csrc` "
                          "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.rt_scene_create.restype = vp
    L.rt_scene_create.argtypes = []
    L.rt_scene_destroy.restype = None
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_clear.argtypes = [vp]
    L.rt_scene_add_triangle.argtypes = [vp, _fp, _fp, _fp, _fp, cf]
    L.rt_scene_add_triangles.argtypes = [vp, ci, _fp, _fp, _fp, _fp, _fp]
    L.rt_scene_add_quad_half.argtypes = [vp, _fp, _fp, _fp, _fp, cf]
    L.rt_scene_add_light.argtypes = [vp, _fp, cf]
    L.rt_scene_load_rto.argtypes = [vp, C.c_char_p]
    L.rt_scene_triangle_count.argtypes = [vp]
    L.rt_scene_light_count.argtypes = [vp]
    L.rt_scene_get_triangles.argtypes = [vp, _fp, ci]
    L.rt_scene_commit.argtypes = [vp, ci, ci]
    L.rt_scene_build_ms.argtypes = [vp, ci, _fp]
    L.rt_scene_accel.argtypes = [vp, ci]
    L.rt_scene_get_bvh.argtypes = [vp, ci, _fp, ci]
    L.rt_scene_get_hoisted.argtypes = [vp, ci, C.POINTER(C.c_int), ci]
    L.rt_camera_trans.argtypes = [_fp, _fp]
    cp = C.POINTER(CameraState)
    L.rt_camera_init.argtypes = [cp, _fp, _fp]
    L.rt_camera_rotate.argtypes = [cp, ci, ci, ci, ci]
    L.rt_camera_translate.argtypes = [cp, ci, ci, ci, ci, ci, ci]
    L.rt_camera_frame_time.argtypes = [cp, cf]
    L.rt_camera_step.argtypes = [cp, C.c_uint, cf]
    L.rt_camera_path.argtypes = [cp, ci, C.POINTER(C.c_uint), _fp, _fp, _fp]
    L.rt_render.argtypes = [vp, _fp, _fp, ci, ci, ci, ci, vp]
    L.rt_render_device.argtypes = [vp, ci, _fp, _fp, ci, ci, ci, ci, vp, vp]
    L.rt_render_tiles_device.argtypes = [vp, ci, _fp, _fp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
    L.rt_render_frames_device.argtypes = [vp, ci, ci, _fp, _fp, ci, ci, ci, ci, vp, C.c_size_t, vp]
    L.rt_shard_rows.argtypes = [ci, ci, ci, ci]
    L.rt_render_tiles.argtypes = [vp, _fp, _fp, ci, ci, ci, ci, ci, ci, ci, vp]
    L.rt_scatter_shards.argtypes = [ci, vp, C.c_size_t, ci, ci, ci, ci, ci, vp, vp]
    L.rt_scatter_shards_host.argtypes = [vp, C.c_size_t, ci, ci, ci, ci, ci, vp]
    L.rt_render_stats.argtypes = [vp, _fp, _fp, ci, ci, ci, _fp, C.POINTER(Stats)]
    L.rt_render_multi.argtypes = [vp, C.POINTER(ci), ci, _fp, _fp, ci, ci, ci, ci, ci, _fp]
    L.rt_cast_rays.argtypes = [vp, ci, _fp, _fp]
    L.rt_device_count.argtypes = []
    L.rt_device_props.argtypes = [ci, C.POINTER(ci), C.POINTER(ci)]
    L.rt_scene_set_tiles_per_wave.argtypes = [vp, ci]
    L.rt_scene_set_traversal.argtypes = [vp, ci]
    L.rt_scene_set_kernel.argtypes = [vp, ci]
    L.rt_scene_set_build.argtypes = [vp, ci]
    L.rt_scene_build_kind.argtypes = [vp, ci]
    L.rt_scene_set_shadow_grid.argtypes = [vp, ci]
    L.rt_scene_set_primary_raster.argtypes = [vp, ci]
    L.rt_scene_shadow_grid.argtypes = [vp, ci, ci, C.POINTER(ci), C.POINTER(C.c_longlong), C.POINTER(ci), C.POINTER(cf)]
    L.rt_scene_bvh_depth.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
    L.rt_render_async.argtypes = [vp, _fp, _fp, ci, ci, ci, ci, vp]
    L.rt_render_wait.argtypes = [vp, ci]
    L.rt_frame_alloc.restype = vp
    L.rt_frame_alloc.argtypes = [C.c_size_t]
    L.rt_frame_free.restype = None
    L.rt_frame_free.argtypes = [vp]
    L.rt_cast_rays_noalloc.argtypes = [vp, ci, _fp, _fp]
    L.rt_scene_traversal.argtypes = [vp, ci, _fp, ci, ci]
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = []
    L.rt_version.restype = C.c_char_p
    L.rt_version.argtypes = []
    for name in SYMBOLS:
        getattr(L, name)          # AttributeError here == the library is older than the header
    _lib = L
    return L


def _check(rc: int) -> int:
    if rc < 0:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))
    return rc


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a: np.ndarray):
    return a.ctypes.data_as(_fp)


class HostFrame:
    """A pinned host framebuffer (rt_frame_alloc: what replaces `data = malloc(...)`, Source.cpp:1035)
    as an (H, W, 3) float32 array — or, rgba8=True, an (H, W, 4) uint8 array for OUT_RGBA8 frames;
    rt_render_async copies into it at PCIe speed."""

    def __init__(self, width: int, height: int, rgba8: bool = False):
        n = int(width) * int(height) * (1 if rgba8 else 3)
        self._p = lib().rt_frame_alloc(n)
        if not self._p:
            raise RtError(-4, lib().rt_last_error().decode())
        self.rgba8 = bool(rgba8)
        if rgba8:
            self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(n * 4,)).reshape(int(height), int(width), 4)
        else:
            self.array = np.ctypeslib.as_array(C.cast(self._p, _fp), shape=(n,)).reshape(int(height), int(width), 3)

    @property
    def ptr(self) -> int:
        return int(self._p)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            lib().rt_frame_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count() -> int:
    return int(lib().rt_device_count())


def camera_trans(rot_deg: Sequence[float]) -> np.ndarray:
    """trans = rotate(rotate(I, radians(rot.x), Y), radians(-rot.y), X) — Source.cpp:1145-1147."""
    r = _f32(rot_deg)
    out = np.zeros(16, dtype=np.float32)
    _check(lib().rt_camera_trans(_p(r), _p(out)))
    return out


def shard_rows(height: int, tile_rows: int, tile_first: int, tile_stride: int) -> int:
    return int(lib().rt_shard_rows(height, tile_rows, tile_first, tile_stride))


def scatter_shards(device: int, d_shards_ptr: int, pitch_bytes: int, n_shards: int, width: int, height: int, tile_rows: int,
                   mode: int, d_frame_ptr: int, stream: int = 0) -> None:
    """rt_scatter_shards: de-interleave the gathered compact shards (shard r at d_shards_ptr + r*pitch_bytes) into the
    frame at d_frame_ptr, one kernel on `stream` of `device`.  `mode` tells the pixel format (OUT_RGBA8 or float RGB)."""
    _check(lib().rt_scatter_shards(device, C.c_void_p(d_shards_ptr), C.c_size_t(pitch_bytes), n_shards, width, height, tile_rows,
                                   mode, C.c_void_p(d_frame_ptr), C.c_void_p(stream)))


def scatter_shards_host(shards_ptr: int, pitch_bytes: int, n_shards: int, width: int, height: int, tile_rows: int,
                        mode: int, frame_ptr: int) -> None:
    """rt_scatter_shards_host: the same on host memory."""
    _check(lib().rt_scatter_shards_host(C.c_void_p(shards_ptr), C.c_size_t(pitch_bytes), n_shards, width, height, tile_rows,
                                        mode, C.c_void_p(frame_ptr)))


def camera_sweep(n_frames: int = 64, pos=(-5.0, 5.0, 5.0), rot=(135.0, -45.0, 0.0), frame_ms: float = 16.0):
    """BASELINE config 5's deterministic camera path, walked by the reference's own controller
    (rt_camera_path: Camera::rotate / translate, Source.cpp:144-209, speeds from the frame time,
    Source.cpp:1307-1308): look right + move forward held down, every frame reported as `frame_ms`
    (the first step moves with the base speeds the reference starts with, Source.cpp:109-112).
    -> (cam_pos (n,3), trans (n,16)) float32, the pose each frame is drawn with."""
    return Camera(pos, rot).path([KEY_LOOK_RIGHT | KEY_FORWARD] * int(n_frames), [float(frame_ms)] * int(n_frames))


# key flags of Source.cpp:237-241 (rt355.h RT_KEY_*)
KEY_LEFT, KEY_RIGHT, KEY_UP, KEY_DOWN, KEY_FORWARD, KEY_BACKWARD = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
KEY_LOOK_LEFT, KEY_LOOK_RIGHT, KEY_LOOK_UP, KEY_LOOK_DOWN = 0x40, 0x80, 0x100, 0x200


class Camera:
    """Camera(position, rotations_deg) — Source.cpp:95-212: the pose the render takes plus the
    reference's controller (rotate / translate / frame-time speeds) through rt_camera_*."""

    def __init__(self, position=(-5.0, 5.0, 5.0), rotations=(135.0, -45.0, 0.0)):
        self._c = CameraState()
        p, r = _f32(position), _f32(rotations)
        _check(lib().rt_camera_init(C.byref(self._c), _p(p), _p(r)))

    position = property(lambda self: tuple(self._c.position))
    rotations = property(lambda self: tuple(self._c.rotations))
    direction = property(lambda self: tuple(self._c.direction))
    right = property(lambda self: tuple(self._c.right))
    up = property(lambda self: tuple(self._c.up))
    moveSpeed = property(lambda self: self._c.move_speed)
    rotSpeed = property(lambda self: self._c.rot_speed)

    def state(self) -> np.ndarray:
        """position, rotations, direction, right, up, moveSpeed, rotSpeed as 17 float32."""
        return np.frombuffer(bytes(self._c), dtype=np.float32).copy()

    def trans(self) -> np.ndarray:
        return camera_trans(self.rotations)

    def rotate(self, l=False, r=False, u=False, d=False) -> "Camera":                     # Source.cpp:181
        _check(lib().rt_camera_rotate(C.byref(self._c), int(l), int(r), int(u), int(d)))
        return self

    def translate(self, l=False, r=False, u=False, d=False, f=False, b=False) -> "Camera":  # Source.cpp:144
        _check(lib().rt_camera_translate(C.byref(self._c), int(l), int(r), int(u), int(d), int(f), int(b)))
        return self

    def frame_time(self, milis: float) -> "Camera":                                        # Source.cpp:1307-1308
        _check(lib().rt_camera_frame_time(C.byref(self._c), float(milis)))
        return self

    def step(self, key_mask: int, milis: float) -> "Camera":
        _check(lib().rt_camera_step(C.byref(self._c), int(key_mask), float(milis)))
        return self

    def path(self, key_masks, frame_ms):
        """Poses of a scripted path -> (cam_pos (F,3), trans (F,16)) for Scene.render_frames_device."""
        k = np.ascontiguousarray(np.asarray(key_masks, dtype=np.uint32))
        ms = _f32(frame_ms)
        if k.shape != ms.shape or k.ndim != 1:
            raise ValueError("key_masks and frame_ms must be 1-D and of equal length")
        pos = np.zeros((len(k), 3), dtype=np.float32)
        tr = np.zeros((len(k), 16), dtype=np.float32)
        _check(lib().rt_camera_path(C.byref(self._c), len(k), k.ctypes.data_as(C.POINTER(C.c_uint)), _p(ms), _p(pos), _p(tr)))
        return pos, tr


class Scene:
    """The reference's globals `triangles` / `lights` (Source.cpp:64-73) behind a handle."""

    def __init__(self):
        self._h = lib().rt_scene_create()
        if not self._h:
            raise RtError(-4, lib().rt_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- construction: the reference's entry points ------------------------------------------
    def Triangle(self, p1, p2, p3, color, shininess=0.0) -> "Scene":
        """triangles.push_back(Triangle(p1, p2, p3, color, shininess)) — Source.cpp:45-59."""
        a, b, c, d = _f32(p1), _f32(p2), _f32(p3), _f32(color)
        _check(lib().rt_scene_add_triangle(self._h, _p(a), _p(b), _p(c), _p(d), float(shininess)))
        return self

    def addTriangle(self, translate, rotate, scale, color, shininess=0.0) -> "Scene":
        """addTriangle(translate, rotate_deg, scale, color, shininess) — Source.cpp:410-439."""
        a, b, c, d = _f32(translate), _f32(rotate), _f32(scale), _f32(color)
        _check(lib().rt_scene_add_quad_half(self._h, _p(a), _p(b), _p(c), _p(d), float(shininess)))
        return self

    def addTriangles(self, p1, p2, p3, color, shininess) -> "Scene":
        a, b, c, d, e = _f32(p1), _f32(p2), _f32(p3), _f32(color), _f32(shininess)
        n = a.shape[0]
        if not (a.shape == b.shape == c.shape == d.shape == (n, 3) and e.shape == (n,)):
            raise ValueError("addTriangles wants (n,3) x4 and (n,)")
        _check(lib().rt_scene_add_triangles(self._h, n, _p(a), _p(b), _p(c), _p(d), _p(e)))
        return self

    def addLight(self, position, intensity) -> "Scene":
        """lights.push_back({position, intensity}) — Source.cpp:880."""
        a = _f32(position)
        _check(lib().rt_scene_add_light(self._h, _p(a), float(intensity)))
        return self

    def loadModel(self, path: str) -> int:
        """loadModel(path) — Source.cpp:631-657 (.rto text records)."""
        return _check(lib().rt_scene_load_rto(self._h, os.fsencode(path)))

    def clear(self) -> "Scene":
        _check(lib().rt_scene_clear(self._h))
        return self

    @classmethod
    def from_desc(cls, desc) -> "Scene":
        """Replays a scenes.SceneDesc (ordered reference calls) through the C ABI."""
        sc = cls()
        run = []   # consecutive `tri` ops go through the bulk entry point

        def flush():
            if run:
                v = np.asarray(run, dtype=np.float32)
                sc.addTriangles(v[:, 0:3], v[:, 3:6], v[:, 6:9], v[:, 9:12], v[:, 12])
                run.clear()

        for kind, v in desc.ops:
            if kind == "tri":
                run.append(v)
                continue
            flush()
            if kind == "quadhalf":
                sc.addTriangle(v[0:3], v[3:6], v[6:9], v[9:12], v[12])
            elif kind == "light":
                sc.addLight(v[0:3], v[3])
            else:
                raise ValueError("unknown scene op %r" % (kind,))
        flush()
        return sc

    # -- inspection -----------------------------------------------------------------------------
    @property
    def n_triangles(self) -> int:
        return _check(lib().rt_scene_triangle_count(self._h))

    @property
    def n_lights(self) -> int:
        return _check(lib().rt_scene_light_count(self._h))

    def triangles(self) -> np.ndarray:
        """(N, 28) float32 — the reference's Triangle records (p n p1 p2 p3 edge0..2 color shin)."""
        n = self.n_triangles
        out = np.zeros((n, TRI_FLOATS), dtype=np.float32)
        if n:
            _check(lib().rt_scene_get_triangles(self._h, _p(out), n))
        return out

    # -- device ---------------------------------------------------------------------------------
    def commit(self, device: int = 0, accel: int = ACCEL_AUTO) -> "Scene":
        _check(lib().rt_scene_commit(self._h, int(device), int(accel)))
        return self

    def accel(self, device: int = 0) -> int:
        return _check(lib().rt_scene_accel(self._h, int(device)))

    def set_traversal(self, mode: int) -> "Scene":
        """How this scene's LBVH is traversed (rt_scene_set_traversal): TRAVERSAL_AUTO | _PER_LANE | _PACKET."""
        _check(lib().rt_scene_set_traversal(self._h, int(mode)))
        return self

    def set_kernel(self, kind: int) -> "Scene":
        """Which form of the render kernel brute-force launches use (rt_scene_set_kernel)."""
        _check(lib().rt_scene_set_kernel(self._h, int(kind)))
        return self

    def set_build(self, quality: int) -> "Scene":
        """How the tree of the next LBVH commit is built (rt_scene_set_build): BUILD_PLOC (the default: Morton sort +
        locally-ordered clustering, all on the GPU), BUILD_FAST (Morton/Karras on the GPU) or BUILD_SAH (binned
        surface-area heuristic on host threads)."""
        _check(lib().rt_scene_set_build(self._h, int(quality)))
        return self

    def set_primary_raster(self, mode: int) -> "Scene":
        """How primary rays of LBVH commits find their hit (rt_scene_set_primary_raster): PRIMARY_AUTO | _WALK | _RASTER."""
        _check(lib().rt_scene_set_primary_raster(self._h, int(mode)))
        return self

    def set_shadow_grid(self, on: bool) -> "Scene":
        """Per-light direction grids for the shadow rays of LBVH commits (rt_scene_set_shadow_grid; on by default)."""
        _check(lib().rt_scene_set_shadow_grid(self._h, 1 if on else 0))
        return self

    def shadow_grid(self, device: int = 0, light: int = 0) -> dict:
        """{cells_per_edge, entries, always, build_ms} of one light's grid (rt_scene_shadow_grid); cells_per_edge 0: none."""
        r, e, a, ms = C.c_int(0), C.c_longlong(0), C.c_int(0), C.c_float(0)
        _check(lib().rt_scene_shadow_grid(self._h, int(device), int(light), C.byref(r), C.byref(e), C.byref(a), C.byref(ms)))
        return {"cells_per_edge": int(r.value), "entries": int(e.value), "always": int(a.value), "build_ms": float(ms.value)}

    def build_kind(self, device: int = 0) -> int:
        """BUILD_* the tree of the last LBVH commit was really built by (a PLOC commit that fell back reports BUILD_FAST)."""
        return _check(lib().rt_scene_build_kind(self._h, int(device)))

    def set_tiles_per_wave(self, tiles: int) -> "Scene":
        _check(lib().rt_scene_set_tiles_per_wave(self._h, int(tiles)))
        return self

    def bvh_depth(self, device: int = 0):
        """(binary depth, 4-wide depth) of the LBVH (rt_scene_bvh_depth); (0, 0) for brute force."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().rt_scene_bvh_depth(self._h, int(device), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def bvh(self, device: int = 0) -> np.ndarray:
        """(triangles in the tree - 1, 16) float32 LBVH nodes (see rt_scene_get_bvh); empty for a brute-force commit."""
        n = _check(lib().rt_scene_get_bvh(self._h, int(device), None, 0))
        out = np.zeros((n, 16), dtype=np.float32)
        if n:
            _check(lib().rt_scene_get_bvh(self._h, int(device), _p(out), n))
        return out

    def hoisted(self, device: int = 0) -> np.ndarray:
        """Indices of the triangles an LBVH commit kept out of the tree (rt_scene_get_hoisted)."""
        n = _check(lib().rt_scene_get_hoisted(self._h, int(device), None, 0))
        out = np.zeros(n, dtype=np.int32)
        if n:
            _check(lib().rt_scene_get_hoisted(self._h, int(device), out.ctypes.data_as(C.POINTER(C.c_int)), n))
        return out

    def build_ms(self, device: int = 0) -> float:
        ms = C.c_float(0)
        _check(lib().rt_scene_build_ms(self._h, int(device), C.byref(ms)))
        return float(ms.value)

    # -- render: replaces the getPixelData loop -------------------------------------------------
    def render(self, cam_pos, trans16, width: int, height: int, max_bounces: int = 1,
               mode: int = STRICT) -> np.ndarray:
        """(H, W, 3) float32, data[(y*w+x)*3+c], row 0 = bottom — Source.cpp:550, 612-614; with
        mode | OUT_RGBA8 the (H, W, 4) uint8 frame the reference's viewer displays (GL unorm, alpha 255); with
        mode | SHADE_GLSL the frame the reference's compute shader computes (no mirrors, no clamp; rt355.h)."""
        cam, tr = _f32(cam_pos), _f32(trans16)
        out = np.empty((height, width, 4), dtype=np.uint8) if mode & OUT_RGBA8 else np.empty((height, width, 3), dtype=np.float32)
        _check(lib().rt_render(self._h, _p(cam), _p(tr), width, height, max_bounces, mode, C.c_void_p(out.ctypes.data)))
        return out

    def render_async(self, cam_pos, trans16, width: int, height: int, max_bounces: int, out: "HostFrame",
                     mode: int = STRICT) -> int:
        """rt_render_async into a pinned HostFrame (or any array of the frame's size: float32 RGB, or uint8 RGBA
        for an OUT_RGBA8 frame) -> ticket."""
        cam, tr = _f32(cam_pos), _f32(trans16)
        ptr = out.ptr if isinstance(out, HostFrame) else out.ctypes.data
        rgba8 = out.rgba8 if isinstance(out, HostFrame) else out.dtype == np.uint8
        return _check(lib().rt_render_async(self._h, _p(cam), _p(tr), width, height, max_bounces, (mode & ~OUT_RGBA8) | (OUT_RGBA8 if rgba8 else 0), C.c_void_p(ptr)))

    def render_wait(self, ticket: int) -> None:
        _check(lib().rt_render_wait(self._h, int(ticket)))

    def render_stats(self, cam_pos, trans16, width: int, height: int, max_bounces: int = 1,
                     want_frame: bool = True):
        cam, tr = _f32(cam_pos), _f32(trans16)
        out = np.empty((height, width, 3), dtype=np.float32) if want_frame else None
        st = Stats()
        _check(lib().rt_render_stats(self._h, _p(cam), _p(tr), width, height, max_bounces,
                                     _p(out) if want_frame else None, C.byref(st)))
        return out, st

    def render_device(self, device: int, cam_pos, trans16, width: int, height: int, max_bounces: int,
                      mode: int, d_out_ptr: int, stream: int = 0) -> None:
        """Asynchronous launch into caller-owned device memory (raw pointer, e.g. tensor.data_ptr())."""
        cam, tr = _f32(cam_pos), _f32(trans16)
        _check(lib().rt_render_device(self._h, device, _p(cam), _p(tr), width, height, max_bounces, mode,
                                      C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def traversal(self, device: int, trans16, width: int, height: int) -> int:
        """TRAVERSAL_PACKET / TRAVERSAL_PER_LANE a frame of this size and pose gets (0: brute force)."""
        tr = _f32(trans16)
        return _check(lib().rt_scene_traversal(self._h, device, _p(tr), width, height))

    def render_frames_device(self, device: int, cam_pos3, trans16, width: int, height: int, max_bounces: int,
                             mode: int, d_out_ptr: int, frame_stride_floats: int = 0, stream: int = 0) -> None:
        """n frames (cam_pos3 (n,3), trans16 (n,16)) in as few launches as possible, asynchronously."""
        cam, tr = _f32(cam_pos3).reshape(-1, 3), _f32(trans16).reshape(-1, 16)
        if cam.shape[0] != tr.shape[0]:
            raise ValueError("cam_pos3 and trans16 disagree on the frame count")
        stride = frame_stride_floats or width * height * (1 if mode & OUT_RGBA8 else 3)
        _check(lib().rt_render_frames_device(self._h, device, cam.shape[0], _p(cam), _p(tr), width, height,
                                             max_bounces, mode, C.c_void_p(d_out_ptr), stride, C.c_void_p(stream)))

    def render_tiles_device(self, device: int, cam_pos, trans16, width: int, height: int, max_bounces: int,
                            mode: int, tile_rows: int, tile_first: int, tile_stride: int,
                            d_out_ptr: int, stream: int = 0) -> None:
        cam, tr = _f32(cam_pos), _f32(trans16)
        _check(lib().rt_render_tiles_device(self._h, device, _p(cam), _p(tr), width, height, max_bounces, mode,
                                            tile_rows, tile_first, tile_stride,
                                            C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def render_tiles(self, cam_pos, trans16, width: int, height: int, max_bounces: int, mode: int,
                     tile_rows: int, tile_first: int, tile_stride: int, out=None) -> np.ndarray:
        """rt_render_tiles: this shard's rows into host memory (rows, width, 3) float32, or (rows, width, 4) uint8 with OUT_RGBA8."""
        cam, tr = _f32(cam_pos), _f32(trans16)
        rows = shard_rows(height, tile_rows, tile_first, tile_stride)
        if out is None:
            out = np.empty((rows, width, 4), dtype=np.uint8) if mode & OUT_RGBA8 else np.empty((rows, width, 3), dtype=np.float32)
        ptr = out.ctypes.data if isinstance(out, np.ndarray) else int(out)
        _check(lib().rt_render_tiles(self._h, _p(cam), _p(tr), width, height, max_bounces, mode, tile_rows, tile_first, tile_stride,
                                     C.c_void_p(ptr)))
        return out

    def render_multi(self, devices: Iterable[int], cam_pos, trans16, width: int, height: int,
                     max_bounces: int = 1, mode: int = STRICT, tile_rows: int = 16) -> np.ndarray:
        """rt_render_multi -> (height, width, 3) float32, or (height, width, 4) uint8 with OUT_RGBA8."""
        devs_l = [int(x) for x in devices]
        devs = (C.c_int * len(devs_l))(*devs_l)
        cam, tr = _f32(cam_pos), _f32(trans16)
        out = np.empty((height, width, 4), dtype=np.uint8) if mode & OUT_RGBA8 else np.empty((height, width, 3), dtype=np.float32)
        _check(lib().rt_render_multi(self._h, devs, len(devs), _p(cam), _p(tr), width, height, max_bounces,
                                     mode, tile_rows, C.cast(C.c_void_p(out.ctypes.data), _fp)))
        return out

    def cast_rays(self, rays7) -> np.ndarray:
        """castRay for K rays (Source.cpp:277-405): rays7 (K,7) = origin, dir, skip index (<0 none)
        -> (K,6) = didHit, index|-1, distance, point."""
        r = _f32(rays7).reshape(-1, 7)
        out = np.zeros((r.shape[0], 6), dtype=np.float32)
        _check(lib().rt_cast_rays(self._h, r.shape[0], _p(r), _p(out)))
        return out

    def cast_rays_noalloc(self, rays7) -> np.ndarray:
        """cast_rays through rt_cast_rays_noalloc (staging inside the library image, no device allocation);
        at most NOALLOC_MAX_RAYS rays."""
        r = _f32(rays7).reshape(-1, 7)
        out = np.zeros((r.shape[0], 6), dtype=np.float32)
        _check(lib().rt_cast_rays_noalloc(self._h, r.shape[0], _p(r), _p(out)))
        return out


def render_desc(desc, device: int = 0, accel: int = ACCEL_AUTO, mode: int = STRICT,
                traversal: int = TRAVERSAL_AUTO, shadow_grid: bool = True) -> np.ndarray:
    """Scene description -> frame, the way the reference's main() would: build, commit, render."""
    with Scene.from_desc(desc) as sc:
        sc.set_traversal(traversal)
        sc.set_shadow_grid(shadow_grid)
        sc.commit(device, accel)
        return sc.render(desc.cam_pos, camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth, mode)


def quantize(rgb: np.ndarray) -> np.ndarray:
    """The GL unorm rule both sides of every 8-bit comparison use:
    q = (int)(c*255.0f + 0.5f), c<0 or NaN -> 0, c>1 -> 255."""
    c = np.asarray(rgb, dtype=np.float32)
    c = np.where(c > 0, c, np.float32(0))          # NaN and negatives -> 0
    c = np.minimum(c, np.float32(1))
    return (c * np.float32(255.0) + np.float32(0.5)).astype(np.int32).astype(np.uint8)


def frame_diff(a: np.ndarray, b: np.ndarray) -> dict:
    """The north_star comparison of two frames (float (H,W,3) or already 8-bit): per 8-bit channel
    differences after the GL unorm rule, pixels beyond 1 LSB, and hit/miss flips (a pixel that is
    black — the reference's miss colour, Source.cpp:623-625 — in exactly one of the two)."""
    qa = a if a.dtype == np.uint8 else quantize(a)
    qb = b if b.dtype == np.uint8 else quantize(b)
    if qa.shape != qb.shape or qa.ndim != 3 or qa.shape[2] != 3:
        raise ValueError("frame_diff wants two (H,W,3) frames of one size")
    d = np.abs(qa.astype(np.int16) - qb.astype(np.int16)).max(axis=2)
    black_a, black_b = ~qa.any(axis=2), ~qb.any(axis=2)
    n = d.size
    over = int((d > 1).sum())
    out = dict(pixels=n, differing=int((d > 0).sum()), over_1_lsb=over, max_lsb=int(d.max()) if n else 0,
               flips=int((black_a != black_b).sum()), within_1_lsb_fraction=float(1.0 - over / n) if n else 1.0)
    out["meets_north_star"] = out["within_1_lsb_fraction"] >= 0.999
    if a.dtype != np.uint8 and b.dtype != np.uint8:
        fa, fb = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        out["bit_identical"] = bool(np.array_equal(fa.view(np.uint32), fb.view(np.uint32)))
    return out
