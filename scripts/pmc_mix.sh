# usage: bash scripts/pmc_mix.sh <workload> <kernel substring>   -> VALU instruction mix per wavefront
export TMPDIR=/tmp
w=$1; pat=$2
o=gpurun_out/pmc_mix_$w; rm -rf $o; mkdir -p $o
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $o/p1 -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w > /dev/null 2> $o/p1.err || { echo "pass 1 failed"; tail -3 $o/p1.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $o/p2 -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w > /dev/null 2> $o/p2.err || { echo "pass 2 failed"; tail -3 $o/p2.err; exit 1; }
python3 - $o "$pat" <<'PY'
import csv, sys, glob, os, collections
o, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(o, "p*", "p_counter_collection.csv")):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for d in per.values():
        w = d.get("SQ_WAVES", 0)
        if w >= 100000:                      # the 16-frame launches only
            for k, v in d.items(): acc[k].append(v / w)
for k in sorted(acc): print("%-28s %10.1f per wave" % (k, sum(acc[k]) / len(acc[k])))
PY
