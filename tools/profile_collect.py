#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) -> profiles/<name>_*: the bench lines, the rocprofv3
kernel statistics as text, the per-kernel averages of the two PMC passes, the per-operator tables, and profiles/traffic.json
(HBM bytes per launch of the sweep kernels: FETCH_SIZE (KB) x 2 - the gfx950 correction for wide coalesced reads,
MI355X_MICROARCH.md, HBM section - + WRITE_SIZE (KB), x 1024).  usage: profile_collect.py <tag> <name>"""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def stats_txt(csv_path, out_path, title):
    rows = list(csv.DictReader(open(csv_path)))
    with open(out_path, "w") as f:
        f.write("# %s\n# rocprofv3 --kernel-trace --stats --output-format csv\n" % title)
        f.write("%-120s %7s %14s %12s %12s %12s %6s\n" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
        for r in rows:
            f.write("%-120s %7s %14s %12.0f %12s %12s %6s\n" % (r["Name"][:120], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                                            r["MinNs"], r["MaxNs"], r["Percentage"]))


def pmc_avg(csv_path):
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(csv_path)):
        a = acc[(r["Kernel_Name"], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: (v[0] / v[1], v[1], v[2] / v[1]) for k, v in acc.items()}


for f in ("bench_line.json", "bench_line_under_rocprof.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (name, f)))
stats_txt(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, name + "_bench_kernel_stats.txt"),
          "python3 bench.py --steps 10 --warmup 2 --no-e2e --no-cpu-baseline (1 x MI355X)")
for a in ("nt", "aa"):
    shutil.copy(os.path.join(src, "ops_table_%s.txt" % a), os.path.join(dst, "%s_ops_table_%s.txt" % (name, a)))
    stats_txt(os.path.join(src, "ops_%s_kernel_stats.csv" % a), os.path.join(dst, "%s_ops_%s_kernel_stats.txt" % (name, a)),
              "python3 tools/bench_ops.py %s (1 x MI355X)" % a)
fetch, write = pmc_avg(os.path.join(src, "bench_pmc_fetch.csv")), pmc_avg(os.path.join(src, "bench_pmc_write.csv"))
with open(os.path.join(dst, name + "_bench_pmc.txt"), "w") as f:
    f.write("# python3 bench.py --steps 3 --warmup 1 --no-e2e --no-dense --no-cpu-baseline, two rocprofv3 passes:\n"
            "# --pmc FETCH_SIZE and --pmc WRITE_SIZE (KB per dispatch, averaged per kernel)\n")
    f.write("%-110s %-11s %7s %16s %12s\n" % ("kernel", "counter", "calls", "avg_value_kb", "avg_ns"))
    for table in (fetch, write):
        for (k, cn), (v, n, ns) in sorted(table.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            f.write("%-110s %-11s %7d %16.1f %12.0f\n" % (k[:110], cn, n, v, ns))


def kern(prefix):
    out = {}
    for (k, cn), (v, n, ns) in list(fetch.items()) + list(write.items()):
        if k.startswith(prefix):
            out.setdefault(k, {})[cn] = v
    return out


line = json.load(open(os.path.join(src, "bench_line.json")))
traffic = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (profiles/%s_bench_pmc.txt, bench.py --steps 3): FETCH_SIZE(KB) x 2 "
                       "(gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE(KB), x 1024. Workload "
                       "c4_1M_x200_nt_tophits, 1 GPU, tile streams; same commit as profiles/%s_bench_line.json. A step's sweeps are two launches: "
                       "k_sweep_nt_leafq_multi<float, 4> for its four leaf seeds (ONE pass over the targets for four sweeps) and "
                       "k_sweep_nt_profq_multi<float, 4> for its four profile seeds (one pass over the internal targets + a table walk over the leaf "
                       "targets per seed); sweep_launch_average is their mean, i.e. the traffic of FOUR sweeps." % (name, name)}
sw = kern("void k_sweep_nt")
inst = {k: v for k, v in sw.items() if "k_sweep_nt_leafq_multi<float, 4>" in k or "k_sweep_nt_profq_multi<float, 4>" in k}
if inst:
    fk = sum(v["FETCH_SIZE"] for v in inst.values()) / len(inst)
    wk = sum(v["WRITE_SIZE"] for v in inst.values()) / len(inst)
    entry = {"fetch_size_kb": round(fk, 1), "write_size_kb": round(wk, 1), "bytes_per_launch": int((2 * fk + wk) * 1024),
             "sweeps_per_launch": 4,
             "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"]}
    for k, v in inst.items():
        entry["leaf_seed_instance" if "leafq" in k else "profile_seed_instance"] = {
            "fetch_size_kb": round(v["FETCH_SIZE"], 1), "write_size_kb": round(v["WRITE_SIZE"], 1),
            "bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
    traffic["sweep_launch_average"] = entry
# the figure belongs to the sweep kernels as they were when the counters were read: bench.py prints it only while these files are unchanged
sys.path.insert(0, ROOT)
from bench import sweep_kernel_hash
traffic["kernel_sources_sha256"] = sweep_kernel_hash()
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=2)
print(json.dumps(traffic, indent=1))
