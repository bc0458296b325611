#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/wallclock_round.sh on the GPU box) -> profiles/<name>_*: the rocprofv3 kernel statistics of
the million-sequence NJ phase and of the protein pipeline as text, the comparisons with the compiled reference, the tick counters
of the quartet kernel.  usage: profile_collect_wallclock.py <tag> <name>"""
import csv, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def stats_txt(csv_path, out_path, title, per=None):
    rows = list(csv.DictReader(open(csv_path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out_path, "w") as f:
        f.write("# %s\n# rocprofv3 --kernel-trace --stats --output-format csv; all kernels together: %.2f s\n" % (title, total / 1e9))
        f.write("%-100s %9s %10s %12s %12s %12s %6s%s\n" % ("kernel", "calls", "total_s", "avg_us", "min_us", "max_us", "%", "  us_per_join" if per else ""))
        for r in rows:
            f.write("%-100s %9s %10.3f %12.2f %12.2f %12.2f %6s%s\n" % (
                r["Name"].split("(")[0].replace("void ", "")[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e9, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"], "  %10.2f" % (float(r["TotalDurationNs"]) / 1e3 / per) if per else ""))


stats_txt(os.path.join(src, "engine_1M_kernel_stats.csv"), os.path.join(dst, name + "_engine_1M_kernel_stats.txt"),
          "python3 tools/nj_gpu_only.py 1000000 200 mu=0.02 gap=0.01 seed=4 (config C4's alignment, default settings, 1 x MI355X): 999 014 joins", per=999014)
stats_txt(os.path.join(src, "aa_3000_kernel_stats.csv"), os.path.join(dst, name + "_aa_3000_kernel_stats.txt"),
          "python3 tools/full_pipeline_aa_once.py 3000 300 (-lg -double-precision, the complete default protein pipeline, 1 x MI355X)")
for f, out in (("compare_c2.txt", "_compare_c2.txt"), ("compare_aa_3000.txt", "_compare_aa_3000.txt"), ("ml_ticks_aa_1500.txt", "_ml_ticks.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, name + out))
with open(os.path.join(dst, name + "_nj_wallclock_runs.txt"), "w") as out:
    out.write("# tools/wallclock_round.sh on 1 x MI355X (tools/nj_gpu_only.py N L [fastest] mu=0.02 gap=0.01 seed=4)\n")
    for f in sorted(os.listdir(src)):
        if f.startswith("nj_") and f.endswith(".log") and "rocprof" not in f:
            for line in open(os.path.join(src, f)):
                if line.startswith("GPU NJ phase") or "[host]" in line or "[count]" in line or line.startswith("  vft_"):
                    out.write(line)
print(sorted(f for f in os.listdir(dst) if f.startswith(name)))
