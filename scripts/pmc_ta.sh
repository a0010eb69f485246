# texture-addresser / vector L1 busy and stall counters of spheres800's render kernel, one rocprofv3 --pmc pass each -> gpurun_out/pmc3/
export TMPDIR=/tmp
o=gpurun_out/pmc3; rm -rf $o; mkdir -p $o
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?\s*:?\s*(TA_|TCP_)" | head -60 > $o/list.txt; wc -l $o/list.txt
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d $o/p$i -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload spheres800 > /dev/null 2> $o/p$i.err || { echo "pass $i failed"; tail -2 $o/p$i.err; }
done
python3 - $o "render_kernel<2, false>" <<'PY'
import csv, sys, glob, os, collections
o, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(o, "p*", "p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-40s %16.0f" % (k, sum(acc[k]) / len(acc[k])))
PY
