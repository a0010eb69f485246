#!/bin/bash
# tools/refresh_evidence.sh [tag] — run HERE (the build container): re-takes the profiles of every bench workload on a GPU box,
# keeps them under profiles/ (stamped with the fingerprint of the current sources), then runs the default bench against them
# and keeps its line as profiles/<tag>_bench.json.  Two gpurun calls, ~4 GPU-minutes.
tag=${1:-r03}
cd "$(dirname "$0")/.."
/usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/profile.sh $tag \"room800 spheres800 spheres4k mirrors4k sweep8k sweep8k_spheres stock800\" > gpurun_out/prof_$tag.log 2>&1; echo profile rc=\$?" 2>&1 | tee /dev/stderr | grep -q "profile rc=0" || { echo "profile step did not run to the end (a transient refusal: wait and run again)"; exit 1; }
python tools/keep_profiles.py $tag > /dev/null || exit 1
/usr/local/graft/bin/gpurun --timeout 900 -- "timeout -k 10 800 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo bench rc=\$?" 2>&1 | tee /dev/stderr | grep -q "bench rc=0" || { echo "bench step did not run to the end (a transient refusal: wait and run again)"; exit 1; }
cp gpurun_out/bench_$tag.json profiles/${tag}_bench.json
fp=$(python raytracing_amd/srchash.py)
sed -i "s/The entries here are fingerprint \`[0-9a-f]*\`/The entries here are fingerprint \`$fp\`/" profiles/README.md
python - <<'PY'
import json
d = json.loads(open("profiles/r03_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", round(d["value"]), "Mrays/s", round(d["ms_per_frame"] * 1e3, 2), "us/frame frac", round(r["frac"], 3), "stale", r["profile_stale"], "issue", r["valu_issue_use"] and (round(r["valu_issue_use"]["low"], 2), round(r["valu_issue_use"]["cycles_per_valu_if_issue_bound"], 2)))
for c in d["configs"]:
    u = c.get("valu_issue_use")
    print(c["workload"], c.get("build"), round(c["ms_per_frame"] * 1e3, 1), "frac %.3f" % c["roofline_frac"], "build", c.get("build_ms") and round(c["build_ms"], 2), u and round(u["cycles_per_valu_if_issue_bound"], 2))
h = d["hand_off"]; h.pop("note"); print(h)
PY
