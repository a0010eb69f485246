"""A fingerprint of the sources the shared library is built from (raytracing_amd/csrc/*, include/rt355.h): what the
numbers kept under profiles/ are stamped with (tools/profile.sh), so that bench.py can tell a profile of THIS code from
a profile of an earlier state (the GPU box has no .git to ask)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    pats = ("raytracing_amd/csrc/*.hip", "raytracing_amd/csrc/*.inl", "raytracing_amd/csrc/*.h", "raytracing_amd/csrc/*.cpp",
            "raytracing_amd/csrc/Makefile", "include/*.h")
    out = []
    for p in pats:
        out += glob.glob(os.path.join(ROOT, p))
    return sorted(out)


def source_hash():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
