# packets | per lane on the 102k-triangle scene around the switch-over (launch_render: a tile's footprint of 6 triangle sizes), 12 timed launches each
for s in ${SIZES:-"500 500" "600 600" "700 700" "800 800" "900 900" "1000 1000"}; do for tr in packet per-lane; do
  timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu --no-configs --workload spheres800 --size $s --traversal $tr | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$s $tr', round(d['ms_per_frame']*1000,2), 'us/frame')" || exit 1
done; done
