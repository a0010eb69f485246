# usage: pmc2.sh <kernel-substring> <bench args...>
pat=$1; shift
export TMPDIR=/tmp
o=gpurun_out/pmc2; rm -rf $o; mkdir -p $o
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d $o/p$i -o p -- python3 bench.py --steps 16 --warmup 16 --no-cpu "$@" > /dev/null 2> $o/p$i.err || { echo "pass $i failed"; tail -3 $o/p$i.err; }
done
python3 - $o "$pat" <<'PY'
import csv, sys, glob, os, collections
o, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(o, "p*", "p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print("%-32s %16.0f" % (k, m[k]))
w = m.get("SQ_WAVES", 1)
print("per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f cycles %.0f" % (m.get("SQ_INSTS_VALU",0)/w, m.get("SQ_INSTS_SALU",0)/w, m.get("SQ_INSTS_LDS",0)/w, m.get("SQ_INSTS_VMEM_RD",0)/w, 4*m.get("SQ_WAVE_CYCLES",0)/w))
if "TCC_HIT_sum" in m: print("L2 hit rate %.3f" % (m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"])))
PY
