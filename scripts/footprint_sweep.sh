# packets | per lane on the 100k-triangle scene over frame sizes (the table behind use_packets' threshold)
# usage: bash scripts/footprint_sweep.sh [ploc|fast|sah]
b=${1:-ploc}
for s in "300 300" "400 400" "500 500" "600 600" "800 800" "1000 1000" "1200 1200" "1600 1600" "2400 2400" "3840 2160"; do for tr in packet per-lane; do
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload spheres800 --size $s --traversal $tr --build $b | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$s $b $tr', round(d['ms_per_frame']*1000,1), 'us/frame', round(d['value']), 'Mrays/s')" || exit 1
done; done
