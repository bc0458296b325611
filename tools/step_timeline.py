#!/usr/bin/env python3
"""The kernels of the LAST bench step from a rocprofv3 --kernel-trace CSV, in launch order: start offset, duration and the gap to the
kernel before (us) - where a step's wall-clock goes between the kernels.  usage: step_timeline.py trace.csv [steps_back]
(steps_back: 1 = the last step - in bench.py the one instrumented with HIP events - 3 = a step of the timed region)"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# a step ends with k_select_rank: take the kernels between the last two of them
ends = [i for i, r in enumerate(rows) if "k_select_rank" in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
t0, prev = rows[lo][0], rows[lo][0]
busy = 0.0
for s, e, n in rows[lo:hi]:
    print("%9.1f  %8.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:70]))
    busy += (e - s) / 1e3
    prev = e
print("step: %.1f us from first start to last end, %.1f us inside kernels, %d launches" % ((rows[hi - 1][1] - t0) / 1e3, busy, hi - lo))
