#!/usr/bin/env python3
"""Per-kernel dispatch durations from a rocprofv3 --kernel-trace CSV: for each kernel the number of dispatches and the
durations (us) of its longest ones - the --stats average mixes batch sizes.  Usage: kernel_trace_summary.py trace.csv [N]"""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort(reverse=True)
    print("%-60s n=%6d total %10.1f us  longest: %s" % (k[:60], len(v), sum(v), " ".join("%.1f" % x for x in v[:top])))
