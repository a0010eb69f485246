"""Scene descriptions for the render loop: the data the reference hard-codes in ``main``.

A scene here is only *data* — an ordered list of the calls the reference makes on its own scene
entry points (``addTriangle`` Source.cpp:410, ``Triangle(...)`` Source.cpp:45,
``lights.push_back`` Source.cpp:880) plus a camera pose (Source.cpp:810), a frame size
(Source.cpp:887-888) and a bounce depth (Source.cpp:567).  Turning the calls into triangle
records is done by the C-ABI library (product) or by the oracle (tests); nothing is computed here.

The same list can be written as a small text file (format: oracle/SCENE_FORMAT.md) that the
headless harness, the oracle driver and the tests all read.
"""
from __future__ import annotations

import dataclasses
from typing import Iterable, List, Sequence, Tuple

import numpy as np

Op = Tuple[str, Tuple[float, ...]]


@dataclasses.dataclass
class SceneDesc:
    """Ordered scene-construction calls + camera + frame parameters."""

    name: str
    ops: List[Op] = dataclasses.field(default_factory=list)
    cam_pos: Tuple[float, float, float] = (-5.0, 5.0, 5.0)       # Source.cpp:810
    cam_rot: Tuple[float, float, float] = (135.0, -45.0, 0.0)    # 180-45, -45, 0
    width: int = 800                                             # Source.cpp:887
    height: int = 800                                            # Source.cpp:888
    depth: int = 1                                               # Source.cpp:567 maxBounces

    # -- builders mirroring the reference's entry points -----------------------------------
    def quad_half(self, t, r, s, color, shin=0.0) -> "SceneDesc":
        """addTriangle(translate, rotate_deg, scale, color, shininess) — Source.cpp:410."""
        self.ops.append(("quadhalf", tuple(map(float, (*t, *r, *s, *color, shin)))))
        return self

    def quad(self, t, r, s, color, shin=0.0) -> "SceneDesc":
        """The reference builds every quad as two halves, the second with Rz + 180
        (Source.cpp:815-816 and every pair after it)."""
        self.quad_half(t, r, s, color, shin)
        self.quad_half(t, (r[0], r[1], r[2] + 180.0), s, color, shin)
        return self

    def tri(self, p1, p2, p3, color, shin=0.0) -> "SceneDesc":
        """Triangle(p1, p2, p3, color, shininess) — Source.cpp:45."""
        self.ops.append(("tri", tuple(map(float, (*p1, *p2, *p3, *color, shin)))))
        return self

    def light(self, pos, intensity) -> "SceneDesc":
        """lights.push_back({pos, intensity}) — Source.cpp:880."""
        self.ops.append(("light", tuple(map(float, (*pos, intensity)))))
        return self

    # -- counts -----------------------------------------------------------------------------
    @property
    def n_triangles(self) -> int:
        return sum(1 for k, _ in self.ops if k in ("quadhalf", "tri"))

    @property
    def n_lights(self) -> int:
        return sum(1 for k, _ in self.ops if k == "light")

    # -- text form --------------------------------------------------------------------------
    def to_text(self) -> str:
        def f(x: float) -> str:
            return "%.9g" % float(np.float32(x))

        lines = ["# scene %s — see oracle/SCENE_FORMAT.md" % self.name]
        lines.append("size %d %d" % (self.width, self.height))
        lines.append("depth %d" % self.depth)
        lines.append("camera " + " ".join(f(x) for x in (*self.cam_pos, *self.cam_rot)))
        for kind, vals in self.ops:
            lines.append(kind + " " + " ".join(f(x) for x in vals))
        return "\n".join(lines) + "\n"

    @staticmethod
    def from_text(text: str, name: str = "file") -> "SceneDesc":
        sc = SceneDesc(name)
        for lineno, raw in enumerate(text.splitlines(), 1):
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            parts = line.split()
            kw, vals = parts[0], tuple(float(x) for x in parts[1:])
            if kw in ("quadhalf", "tri") and len(vals) == 13:
                sc.ops.append((kw, vals))
            elif kw == "light" and len(vals) == 4:
                sc.ops.append((kw, vals))
            elif kw == "camera" and len(vals) == 6:
                sc.cam_pos, sc.cam_rot = vals[:3], vals[3:]
            elif kw == "size" and len(vals) == 2:
                sc.width, sc.height = int(vals[0]), int(vals[1])
            elif kw == "depth" and len(vals) == 1:
                sc.depth = int(vals[0])
            else:
                raise ValueError("%s:%d: bad scene line %r" % (name, lineno, raw))
        return sc

    def with_size(self, w: int, h: int) -> "SceneDesc":
        return dataclasses.replace(self, ops=list(self.ops), width=w, height=h)

    def with_camera(self, pos, rot) -> "SceneDesc":
        return dataclasses.replace(self, ops=list(self.ops), cam_pos=tuple(pos), cam_rot=tuple(rot))


# ------------------------------------------------------------------------------------------
# The reference's own scene (Source.cpp:815-851, 880) and the BASELINE.json variants of it.
# ------------------------------------------------------------------------------------------

_ROOM = [  # (translate, rotate, scale, color) of the first half of each wall, Source.cpp:815-830
    ((0, 0, 10), (0, 0, 0), (255, 0, 0)),
    ((0, 0, -10), (0, 0, 0), (255, 0, 128)),
    ((10, 0, 0), (0, 90, 0), (0, 255, 0)),
    ((-10, 0, 0), (0, 90, 0), (0, 255, 128)),
    ((0, -10, 0), (90, 0, 0), (0, 0, 255)),
    ((0, 10, 0), (90, 0, 0), (255, 0, 255)),
]
_CUBE = [  # Source.cpp:836-851
    ((0, 0, 1), (0, 0, 0)),
    ((0, 0, -1), (0, 0, 0)),
    ((1, 0, 0), (0, 90, 0)),
    ((-1, 0, 0), (0, 90, 0)),
    ((0, -1, 0), (90, 0, 0)),
    ((0, 1, 0), (90, 0, 0)),
]


def _room(sc: SceneDesc, shin: Sequence[float] = (0, 0, 0, 0, 0, 0)) -> SceneDesc:
    for (t, r, col), s in zip(_ROOM, shin):
        sc.quad(t, r, (10, 10, 10), col, s)
    return sc


def _cube(sc: SceneDesc) -> SceneDesc:
    for t, r in _CUBE:
        sc.quad(t, r, (1, 1, 1), (255, 255, 255), 0)
    return sc


def stock(w: int = 800, h: int = 800) -> SceneDesc:
    """The unmodified v1 scene: 24 triangles, one light (Source.cpp:815-851, 880).  Config 1b."""
    sc = SceneDesc("stock", width=w, height=h)
    _cube(_room(sc))
    return sc.light((0, 8, 0), 4096)


def room(w: int = 800, h: int = 800) -> SceneDesc:
    """BASELINE configs 1/2: the README-style room.  The two walls facing the stock camera
    (z=-10 and x=+10) are mirrors (shininess 1), the white cube stays, and a 2x2 blue quad
    hangs between the ceiling light and the floor."""
    sc = SceneDesc("room", width=w, height=h)
    _cube(_room(sc, shin=(0, 1, 1, 0, 0, 0)))
    sc.quad((0, 6, 0), (90, 0, 0), (1, 1, 1), (0, 0, 255), 0)
    return sc.light((0, 8, 0), 4096)


def mirrors(w: int = 3840, h: int = 2160, depth: int = 16) -> SceneDesc:
    """BASELINE config 3: two facing 20x20 mirrors (shininess 1) inside the matte room, camera
    between them looking at one mirror slightly off-axis, reflection depth 16.  The mirrors stand
    at z=+-3 so that the per-pixel loop trip count really spans 0..16 across the 90-degree view
    (lateral drift per bounce is 12|u|, so |u| < 0.05 reaches the depth limit)."""
    sc = SceneDesc("mirrors", width=w, height=h, depth=depth,
                   cam_pos=(0.5, -1.0, 0.25), cam_rot=(7.0, 3.0, 0.0))
    _room(sc)
    sc.quad((0, 0, 3), (0, 0, 0), (10, 10, 10), (200, 220, 255), 1)
    sc.quad((0, 0, -3), (0, 0, 0), (10, 10, 10), (255, 220, 200), 1)
    return sc.light((0, 8, 0), 4096)


def mirror_checker(w: int = 800, h: int = 800, depth: int = 8, n: int = 16) -> SceneDesc:
    """Config 3 with bounce depths that diverge INSIDE 8x8 tiles: the two facing mirrors of mirrors(), each
    overlaid with an n x n checker of matte patches (every other cell, a hair in front of the glass).  A ray
    ends on the first patch it meets, so neighbouring pixels stop after different numbers of bounces — the
    patches' images shrink with every reflection (an image k bounces deep is 1/(2k+1) the size), down to a
    few pixels.  2 n^2 + 16 triangles."""
    sc = SceneDesc("mirror_checker", width=w, height=h, depth=depth, cam_pos=(0.5, -1.0, 0.25), cam_rot=(7.0, 3.0, 0.0))
    _room(sc)
    sc.quad((0, 0, 3), (0, 0, 0), (10, 10, 10), (200, 220, 255), 1)
    sc.quad((0, 0, -3), (0, 0, 0), (10, 10, 10), (255, 220, 200), 1)
    cell = 20.0 / n
    for side, z in ((0, 3.0 - 0.01), (1, -3.0 + 0.01)):
        for j in range(n):
            for i in range(n):
                if (i + j + side) % 2:
                    continue
                cx, cy = -10.0 + (i + 0.5) * cell, -10.0 + (j + 0.5) * cell
                col = (230, 60 + (37 * i) % 160, 60 + (53 * j) % 160) if side == 0 else (60 + (41 * i) % 160, 230, 60 + (29 * j) % 160)
                sc.quad((cx, cy, z), (0, 0, 0), (cell / 2, cell / 2, 1), col, 0)
    return sc.light((0, 8, 0), 4096)


def two_lights(w: int = 100, h: int = 100) -> SceneDesc:
    """Stock scene plus the second light the reference keeps commented out (Source.cpp:881)."""
    sc = stock(w, h)
    sc.name = "two_lights"
    return sc.light((5, 8, 5), 64)


# ------------------------------------------------------------------------------------------
# BASELINE config 4: tessellated spheres inside the stock room.
# ------------------------------------------------------------------------------------------

class _LCG:
    """Fixed-seed 32-bit LCG (Numerical Recipes constants) -> floats in [0,1)."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFF

    def next(self) -> float:
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return (self.s >> 8) / float(1 << 24)


def icosphere(subdiv: int) -> Tuple[np.ndarray, np.ndarray]:
    """Unit icosphere: (V,3) float64 vertices, (F,3) int faces; F = 20 * 4**subdiv."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache = {}

        def mid(a: int, b: int) -> int:
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts), np.array(faces, dtype=np.int64)


def spheres_arrays(n_spheres: int = 20, subdiv: int = 4, seed: int = 0x5EED):
    """Triangle soup for config 4 as arrays: (p1,p2,p3,color) each (N,3) float32, shin (N,)."""
    rng = _LCG(seed)
    verts, faces = icosphere(subdiv)
    P1, P2, P3, COL, SH = [], [], [], [], []
    for _ in range(n_spheres):
        c = np.array([rng.next() * 16 - 8, rng.next() * 16 - 8, rng.next() * 16 - 8])
        r = 0.3 + 0.9 * rng.next()
        col = np.array([np.floor(rng.next() * 256), np.floor(rng.next() * 256), np.floor(rng.next() * 256)])
        shin = 1.0 if rng.next() < 0.1 else 0.0
        w = (verts * r + c).astype(np.float32)
        P1.append(w[faces[:, 0]]); P2.append(w[faces[:, 1]]); P3.append(w[faces[:, 2]])
        COL.append(np.tile(col.astype(np.float32), (len(faces), 1)))
        SH.append(np.full(len(faces), shin, dtype=np.float32))
    return (np.concatenate(P1), np.concatenate(P2), np.concatenate(P3),
            np.concatenate(COL), np.concatenate(SH))


def spheres(n_spheres: int = 20, subdiv: int = 4, w: int = 800, h: int = 800,
            seed: int = 0x5EED, depth: int = 1) -> SceneDesc:
    """BASELINE config 4: the stock room (12 wall triangles) + ``n_spheres`` icospheres given to
    ``Triangle(p1,p2,p3,...)`` directly.  Defaults: 20 x 5120 + 12 = 102 412 triangles."""
    sc = SceneDesc("spheres_%dx%d" % (n_spheres, 20 * 4 ** subdiv), width=w, height=h, depth=depth)
    _room(sc)
    p1, p2, p3, col, sh = spheres_arrays(n_spheres, subdiv, seed)
    for i in range(len(p1)):
        sc.ops.append(("tri", (*map(float, p1[i]), *map(float, p2[i]), *map(float, p3[i]),
                               *map(float, col[i]), float(sh[i]))))
    return sc.light((0, 8, 0), 4096)


# BASELINE config 5's deterministic camera path is walked by the reference's own controller:
# api.camera_sweep (rt_camera_path).  Nothing is computed here.


BUILTIN = {
    "stock": stock,
    "room": room,
    "mirrors": mirrors,
    "two_lights": two_lights,
    "mirror_checker": mirror_checker,
}


def get(name: str, **kw) -> SceneDesc:
    if name.startswith("spheres"):
        return spheres(**kw)
    return BUILTIN[name](**kw)
