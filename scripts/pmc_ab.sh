# usage: bash scripts/pmc_ab.sh <workload> <tag> [<tag> ...]  — instruction counters per wave of the render kernel for
# library builds build/librt355_<tag>.so (tag "cur" = the in-tree library), one --pmc pass each -> gpurun_out/pmcab_<tag>/
w=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  lib=$PWD/build/librt355_$v.so; [ "$v" = cur ] && lib=$PWD/raytracing_amd/librt355.so
  out=gpurun_out/pmcab_$v; mkdir -p $out
  RT355_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $out -o a -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w > $out/a.json 2> $out/a.err || exit 1
  python3 - $out/a_counter_collection.csv $v <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'render_' not in k: continue
    m={c:sum(x)/len(x) for c,x in v.items()}
    w=m.get('SQ_WAVES',1)
    print(sys.argv[2], k, ' '.join('%s %.1f'%(c.replace('SQ_',''),m[c]/w) for c in sorted(m) if c!='SQ_WAVES'), 'waves %d'%w)
PY
done
