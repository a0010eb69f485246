"""tools/keep_profiles.py <tag> — copy what tools/profile.sh left under gpurun_out/profiles_<tag>/ into profiles/ (the
tracked, judged directory): the per-workload kernel-trace statistics, PMC summaries and bench lines as they are, and the
per-workload entries of traffic.json / issued.json / issue.json merged into the files of the same name (each entry carries
the fingerprint of the sources it was measured on: raytracing_amd/srchash.py)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
try:
    git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    git = ""
for f in glob.glob(os.path.join(src, tag + "_*")):
    shutil.copy(f, dst)
for name in ("traffic.json", "issued.json", "issue.json"):
    new = json.load(open(os.path.join(src, name)))
    path = os.path.join(dst, name)
    try:
        old = json.load(open(path))
    except (OSError, ValueError):
        old = {}
    for w, v in new.items():
        if isinstance(v, dict):
            v["git_head_when_kept"] = git      # (the commit the tree was at or just after; the fingerprint is what is checked)
        old[w] = v
    json.dump(old, open(path, "w"), indent=1)
    print(name, sorted(new))
