# usage: bash scripts/pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out -o a -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs "$@" > $out/a.json 2> $out/a.err || exit 1
python3 - $out/a_counter_collection.csv <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'render_kernel' not in k: continue
    print(k)
    m={c:sum(x)/len(x) for c,x in v.items()}
    for c,x in m.items(): print('   %-22s %14.0f'%(c,x))
    w=m.get('SQ_WAVES',1)
    print('   VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f  wave_cycles/wave(x4) %.0f'%(m['SQ_INSTS_VALU']/w, m['SQ_INSTS_SALU']/w, m['SQ_INSTS_LDS']/w, 4*m['SQ_WAVE_CYCLES']/w))
PY
