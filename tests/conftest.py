"""pytest configuration: markers, paths and the golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(("rays_", "camera_", "rto_")))


def load_golden(name):
    """-> dict(desc=SceneDesc, tris, trans, sha256 hex, frame|None, q8|None)"""
    from raytracing_amd import scenes
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    text = bytes(z["scene"]).decode()
    return dict(desc=scenes.SceneDesc.from_text(text, name), text=text, tris=z["tris"], trans=z["trans"],
                sha256=bytes(z["sha256"]).hex(),
                frame=z["frame"] if "frame" in z.files else None,
                q8=z["q8"] if "q8" in z.files else None)


def rto_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("rto_") and f.endswith(".npz"))


def load_rto_golden(name):
    """Fixtures of the reference's own loadModel (Source.cpp:631-657; make_golden.py rto_cases):
    -> dict(desc = the scene WITHOUT the model, rto = bytes of the model file, tris, trans, frame, sha256)"""
    from raytracing_amd import scenes
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return dict(desc=scenes.SceneDesc.from_text(bytes(z["scene"]).decode(), name), rto=bytes(z["rto"]), tris=z["tris"],
                trans=z["trans"], frame=z["frame"], sha256=bytes(z["sha256"]).hex())


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
