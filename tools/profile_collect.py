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


# instruction issue of the same kernels (profile_round.sh's third and fourth passes): per-kernel averages as text, and for the two
# multi-seed sweep kernels the figures bench.py prints (roofline.valu)
valu = pmc_avg(os.path.join(src, "bench_pmc_valu.csv")) if os.path.exists(os.path.join(src, "bench_pmc_valu.csv")) else {}
mix = pmc_avg(os.path.join(src, "bench_pmc_mix.csv")) if os.path.exists(os.path.join(src, "bench_pmc_mix.csv")) else {}
if valu:
    with open(os.path.join(dst, name + "_bench_pmc_valu.txt"), "w") as f:
        f.write("# python3 bench.py --steps 3 --warmup 1 --no-e2e --no-dense --no-cpu-baseline under rocprofv3 --pmc, two passes:\n"
                "# SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY\n"
                "# GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS\n"
                "# (averages per dispatch; SQ_ACTIVE_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over the wavefronts;\n"
                "#  valu_lane_frac = SQ_INSTS_VALU x 64 lanes / duration / (256 CUs x 64 lanes x 2.4 GHz);\n"
                "#  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))\n")
        kernels = sorted({k for k, _ in valu}, key=lambda k: -max(v[2] * v[1] for (kk, c), v in valu.items() if kk == k))
        for k in kernels:
            row = {c: v for (kk, c), v in valu.items() if kk == k}
            row.update({c: v for (kk, c), v in mix.items() if kk == k})
            calls, ns = row["SQ_WAVES"][1], row["SQ_WAVES"][2]
            f.write("%s\n    calls %d  avg_us %.1f\n" % (k[:150], calls, ns / 1e3))
            for cn in sorted(row):
                f.write("    %-22s %.5g\n" % (cn, row[cn][0]))
            iv, av, wc = row["SQ_INSTS_VALU"][0], row["SQ_ACTIVE_INST_VALU"][0], max(row["SQ_WAVE_CYCLES"][0], 1.0)
            busy = av * 4 / (1024 * row["GRBM_GUI_ACTIVE"][0] / 8) if "GRBM_GUI_ACTIVE" in row and row["GRBM_GUI_ACTIVE"][0] > 0 else float("nan")
            f.write("    -> valu_lane_frac %.3f  valu_busy %.3f  wave cycles: issuing %.2f  waiting (waitcnt / barrier) %.2f  issue-stalled %.2f\n" % (
                iv * 64 / (ns * 1e-9) / (256 * 64 * 2.4e9), busy, row["SQ_ACTIVE_INST_ANY"][0] / wc, row["SQ_WAIT_ANY"][0] / wc, row["SQ_WAIT_INST_ANY"][0] / wc))


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
    vsum, vn = 0.0, 0
    for k, v in inst.items():
        rec = {"fetch_size_kb": round(v["FETCH_SIZE"], 1), "write_size_kb": round(v["WRITE_SIZE"], 1),
               "bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
        if (k, "SQ_INSTS_VALU") in valu:   # wavefront-level VALU instructions per launch; the busy fraction from the counters themselves
            rec["valu_wave_insts_per_launch"] = round(valu[(k, "SQ_INSTS_VALU")][0], 1)
            vsum += valu[(k, "SQ_INSTS_VALU")][0]
            vn += 1
            if (k, "GRBM_GUI_ACTIVE") in mix and mix[(k, "GRBM_GUI_ACTIVE")][0] > 0:
                rec["valu_busy_frac_pmc"] = round(valu[(k, "SQ_ACTIVE_INST_VALU")][0] * 4 / (1024 * mix[(k, "GRBM_GUI_ACTIVE")][0] / 8), 4)
        entry["leaf_seed_instance" if "leafq" in k else "profile_seed_instance"] = rec
    if vn:
        entry["valu_wave_insts_per_launch"] = round(vsum / vn, 1)
    traffic["sweep_launch_average"] = entry
# the figure belongs to the sweep kernels as they were when the counters were read: bench.py prints it only while these files are unchanged
sys.path.insert(0, ROOT)
from bench import sweep_kernel_hash
traffic["kernel_sources_sha256"] = sweep_kernel_hash()
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=2)
print(json.dumps(traffic, indent=1))
