"""From a rocprofv3 --kernel-trace csv of scripts/single_frame.py: the render kernel's own duration and the gap between one
launch's end and the next one's start, per phase (one stream | two streams).  usage: single_frame_gap.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "render_lockstep_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
n = len(rows)
print("launches", n)
for a in range(0, n - 1999, 2000):
    seg = range(a + 200, a + 1999)
    dur = sum(en[i] - st[i] for i in seg) / len(seg)
    step = (st[a + 1999] - st[a + 200]) / (1999 - 200)
    print("launches %5d..: kernel %.2f us, start-to-start %.2f us, idle between %.2f us" % (a, dur / 1e3, step / 1e3, (step - dur) / 1e3))
