# usage: bash scripts/pmc_kcache.sh <workload> [bench args]  -> scalar-cache and L2 hit rates of the render kernels
export TMPDIR=/tmp
w=$1; shift
o=gpurun_out/pmc_kcache_$w; rm -rf $o; mkdir -p $o
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL --output-format csv -d $o/p1 -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w "$@" > /dev/null 2> $o/p1.err || { echo "pass 1 failed"; tail -3 $o/p1.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $o/p2 -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w "$@" > /dev/null 2> $o/p2.err || { echo "pass 2 failed"; tail -3 $o/p2.err; exit 1; }
python3 - $o <<'PY'
import csv, sys, glob, os, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(o, "p*", "p_counter_collection.csv")):
    per = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"][:40]
    for d, c in per.items():
        w = c.get("SQ_WAVES", 0)
        if w >= 100000:
            for k, v in c.items(): acc[names[d]][k].append(v / w)
for kn, a in acc.items():
    print(kn)
    for k in sorted(a): print("   %-28s %10.2f per wave" % (k, sum(a[k]) / len(a[k])))
PY
