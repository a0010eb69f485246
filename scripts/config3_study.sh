#!/bin/bash
# Config 3 (facing mirrors, 3840x2160, depth 16) per pose batch, both forms of the brute-force kernel:
# kernel time per launch from the kernel trace, VALU instructions, lanes active per VALU instruction and issue use from a PMC pass.
# Run on the GPU box; writes gpurun_out/config3_study.txt.
export TMPDIR=/tmp
out=gpurun_out/config3
mkdir -p $out
for k in lockstep state-machine; do
  args="--workload mirrors4k --kernel $k --no-cpu --no-configs --steps 8 --warmup 2"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$k -o t -- python3 bench.py $args > $out/bench_$k.json 2>/dev/null || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $out/pmc_$k -o p -- python3 bench.py $args > /dev/null 2>&1 || exit 1
done
timeout -k 10 300 python3 scripts/kernel_forms.py > $out/kernel_forms.txt 2>/dev/null || exit 1
python3 - $out <<'PY' | tee gpurun_out/config3_study.txt
import csv, sys, json, os, collections
out = sys.argv[1]
print("config 3: mirrors 3840x2160 depth 16, 32 frames per launch; launches alternate between poses 0-31 (batch A) and 32-63 (batch B)")
print("(lanes active = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; issue slots per VALU = 1024 SIMDs x kernel cycles at 2.4 GHz / SQ_INSTS_VALU;")
print(" the issue rates measured on this GPU are 2.5 cycles for register-only FP32 arithmetic, 4.1 with an SGPR operand and for compares/selects/min/max)")
for k in ("lockstep", "state-machine"):
    b = json.loads(open(os.path.join(out, "bench_%s.json" % k)).read().strip().splitlines()[-1])
    rows = [r for r in csv.DictReader(open(os.path.join(out, "trace_%s" % k, "t_kernel_trace.csv"))) if "render_" in r["Kernel_Name"] and ", true>(" not in r["Kernel_Name"] and "true>" not in r["Kernel_Name"].split("(")[0][-6:]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    name = rows[0]["Kernel_Name"].split("(")[0]
    a, bb = dur[0::2], dur[1::2]
    pm = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(out, "pmc_%s" % k, "p_counter_collection.csv"))):
        if name in r["Kernel_Name"]:
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
    def mean(v, sl): 
        v = v[sl]; return sum(v) / max(1, len(v))
    print("%s  %s" % (k, name))
    print("  bench: %.1f us/frame, algorithmic fraction of the FP32 peak %.3f" % (b["ms_per_frame"] * 1e3, b["roofline"]["frac"]))
    for label, sl, d in (("batch A", slice(0, None, 2), a), ("batch B", slice(1, None, 2), bb)):
        valu, act, thr, waves = (mean(pm[c], sl) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES"))
        ms = sum(d) / len(d)
        cyc = ms * 1e-3 * 2.4e9 * 1024
        print("  %s: kernel %.2f ms/launch (%.1f us/frame; min %.2f max %.2f over %d launches)  VALU/launch %.3g  VALU/wave %.0f  lanes active per VALU instruction %.1f of 64  issue slots per VALU %.2f cycles" % (
            label, ms, ms / 32 * 1e3, min(d), max(d), len(d), valu, valu / waves, thr / act if act else 0, cyc / valu))
print(open(os.path.join(out, "kernel_forms.txt")).read())
PY
