#!/bin/bash
# The sweeps DESIGN 5.3.R "The launch shape" cites, in one log (profiles/r03_launch_shape_sweeps.txt): run on a GPU box.
cd /root/repo
echo "== source fingerprint $(python3 raytracing_amd/srchash.py); us/frame, 102 412 triangles unless named otherwise"
echo "== XCD bands off (0) | on (1), experiment build"
bash scripts/bands_sweep.sh || exit 1
echo "== tiles per wavefront (0 = the launcher's choice)"
WLS="spheres800 spheres4k sweep8k_spheres" TPWS="0 1 2 3 4 6" bash scripts/tpw_sweep.sh || exit 1
SIZES="1200 1600 2400" TPWS="1 2 4" bash scripts/tpw_size_sweep.sh || exit 1
echo "== frames per launch"
bash scripts/fpl_sweep.sh || exit 1
echo "== brute force, tiles per wavefront"
WLS="room800 mirrors4k sweep8k" TPWS="0 4 8 16" bash scripts/tpw_sweep.sh || exit 1
