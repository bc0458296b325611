#!/usr/bin/env python3
"""bench.py — profile-ops/sec of the MI355X profile-operations backend on BASELINE.json's headline workload.

Workload (config.workload = "c4_1M_x200_nt_tophits"): the 1M-taxa x 200 bp nucleotide alignment of
BASELINE.json configs[3] (random-descent generator, mu 0.02, gaps 0.01, seed 4) in a mid-NJ "top-hits" state:
250k sibling pairs already joined (250k internal profiles, 500k active leaves, 750k active nodes).
One STEP = one pass of the hot path over one batch of 8 seeds (4 leaves + 4 internal nodes): for each seed the
lazy out-distance refresh, the one-vs-all sweep over every active node (seqDist / profileDist + criterion,
NJ.tcc:3571-3646) and the top-2m selection in the reference's sort order (m = 1000), hits returned to the host.
One profile-op = one seqDist/profileDist evaluation (the reference's seqOps + profileOps counters).

With --gpus N > 1 (torch.distributed / RCCL, one rank per GPU) the target id range of every sweep is sharded over
the ranks (strong scaling: same 1M problem), each rank selects its local top-2m and the lists are all-gathered
and merged with the (criterion asc, id desc) rule.

Smaller problems for quick checks: --n-seqs / --n-pos.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Multi-GPU runs only: small device->host copies (the merged hit lists of a step) go through blit kernels instead of the
# SDMA engines, whose fixed latency is hundreds of microseconds on this platform; must be set before HIP initialises.
# The single-GPU path returns its small results through host-mapped memory and does not need it (measured: no change).
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("HSA_ENABLE_SDMA", "0")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
T_START = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-seqs", type=int, default=1000000)
    ap.add_argument("--n-pos", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end trees (wall-clock half of the metric)")
    ap.add_argument("--no-e2e-c4", action="store_true", help="skip the 1M x 200 end-to-end tree (minutes), keep C3's")
    ap.add_argument("--no-dense", action="store_true", help="skip the phi ~ 1 roofline of the sweep kernel")
    ap.add_argument("--no-e2e-full", action="store_true", help="skip the complete pipelines of configs C2, C5 and C4 (~17 minutes)")
    ap.add_argument("--no-e2e-c4-full", action="store_true", help="skip the complete -nt pipeline of config C4 (1M x 200; ~9 minutes on one GPU)")
    ap.add_argument("--no-e2e-c5-one-thread", action="store_true", help="skip config C5's complete pipeline in the reference's one-thread order (~6 minutes)")
    ap.add_argument("--time-budget", type=float, default=1500.0, help="seconds after which the remaining complete pipelines are skipped (and reported as skipped)")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)   # (internal: "main" = the step measurement alone, or the key of one end-to-end leg)
    ap.add_argument("--in-process", action="store_true", help="run the end-to-end legs inside this process instead of one child process each")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) and the
    CPUs its affinity mask allows: a pod that gets a slice of the box cannot scale beyond its quota whatever the thread count."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = None
    load = None
    try:
        load = float(open("/proc/loadavg").read().split()[0])
    except (OSError, ValueError):
        pass
    return dict(cgroup_quota_cpus=quota, affinity_cpus=allowed, loadavg_1min_before=load)


def cpu_baseline(state, codes, ops, budget_s=6.0):
    """SURVEY.md 8d "CPU baseline beside it": this repo's AVX2 + OpenMP restatement of the one-vs-all sweep
    (oracle/vft_oracle_avx2.c: -O3 -mavx2 -mfma -fopenmp, bit-identical to the scalar oracle that is pinned to the
    reference) timed on the GPU box's host cores over a bounded sample of the same workload - the benchmark's seeds
    against 170 000 active leaves + 80 000 internal profiles of the same alignment (1 GB in the oracle's dense layout).
    The timed leg allocates and first-touches its own copy of the sample inside the OpenMP team, static blocks of 64
    targets per thread, so that every thread streams memory of its own NUMA node; run on every core, on half of them
    (one socket's worth) and on one thread, ~budget_s seconds each.  `value` is the rate on all the cores the container may use
    (its cgroup CPU quota and affinity mask, reported in `host`: the pool's GPU pods get 16 CPUs' worth of a 256-thread box)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import Oracle, avx2_max_threads
    orc = Oracle(ops.dt)
    n = state.n_seqs
    leaf_ids = state.active[state.active < n][:170000]
    int_ids = state.active[state.active >= n][:80000]
    L = codes.shape[1]
    m = len(leaf_ids) + len(int_ids)
    W = np.zeros((m, L), ops.dt)
    Cc = np.full((m, L), 127, np.uint8)
    F = np.zeros((m, L, 4), ops.dt)
    Cc[:len(leaf_ids)] = codes[leaf_ids]
    W[:len(leaf_ids)] = (codes[leaf_ids] != 127)
    for t, v in enumerate(int_ids):
        w, c, f = ops.profile_download(int(v))
        W[len(leaf_ids) + t], Cc[len(leaf_ids) + t], F[len(leaf_ids) + t] = w, c, f
    outp, _ = ops.out_profile_download()
    z = np.zeros(m, ops.dt)
    st = orc.state(len(leaf_ids), W, Cc, F, np.full(m, -1), z, W.sum(1).astype(ops.dt), z, state.totdiam, outp)
    od = np.zeros(m, ops.dt)
    na = np.full(m, state.n_active)
    cores = avx2_max_threads()
    # alternate internal / leaf seeds like the GPU step
    queries = np.array([(len(leaf_ids) + q // 2) % m if q % 2 == 0 else q // 2 for q in range(16)], np.int64)
    host = cpu_quota_cores()
    # a container with a CPU quota (the pool's GPU pods: 16 CPUs' worth of a 256-thread box) cannot use more cores than that:
    # a team of 128 threads only fights over the quota's time slices (measured: 128 threads slower than 64, 5-9x one thread)
    if host["cgroup_quota_cpus"]:
        cores = max(1, min(cores, int(host["cgroup_quota_cpus"] + 0.5)))
    if host["affinity_cpus"]:
        cores = max(1, min(cores, host["affinity_cpus"]))
    rates = {}
    for threads in sorted({cores, max(1, cores // 2), 1}, reverse=True):
        secs, done, _ = orc.avx2_sweep_bench(st, queries, state.n_active, od, na, threads=threads,
                                             budget=budget_s if threads > 1 else min(budget_s, 4.0))
        rates[threads] = (done * m / secs, done)
    return dict(value=rates[cores][0], unit="profile-ops/s", cores=cores, kind="port", cpu=cpu_model(),
                value_1_thread=rates[1][0], value_half_cores=rates[max(1, cores // 2)][0],
                scaling_all_cores_vs_1_thread=rates[cores][0] / rates[1][0], host=host,
                sample="%d sweeps on %d threads, %d on %d, %d on 1 thread, each of one seed vs %d active leaves + %d internal "
                       "profiles of the same alignment (team-allocated, first-touched copy; static blocks of 64 targets; all repetitions inside one parallel region); "
                       "AVX2 + OpenMP restatement (oracle/vft_oracle_avx2.c)"
                       % (rates[cores][1], cores, rates[max(1, cores // 2)][1], max(1, cores // 2), rates[1][1], len(leaf_ids), len(int_ids)))


def dense_profile_roofline(ops, state, n, L, groups=None):
    """roofline of k_sweep_nt at vector density phi ~ 1 (SURVEY 8d asks for phi in {0, 0.25, 1}; the headline state has
    0.28): `groups` internal profiles that each average 8 unrelated leaves, built in the free id space above the
    benchmark state and swept alone (target range = those nodes).  Returns a dict or None when the ids do not fit."""
    base = ((state.maxnode + 63) // 64) * 64
    if groups is None:   # as many as the free id space holds (a sweep needs >> 1024 tiles of 64 targets to fill 256 CUs)
        groups = min(131072, ((ops.max_nodes - base) // 7) // 8192 * 8192)
    if groups < 8192 or base + 7 * groups > ops.max_nodes or 8 * groups > n:
        return None
    rng = np.random.default_rng(99)
    leaves = rng.permutation(n)[:8 * groups].astype(np.int64)
    ops.set_max_node(base + 7 * groups)
    # three levels of averages; the final `groups` nodes start at a tile boundary
    l1 = base + 3 * groups + np.arange(4 * groups, dtype=np.int64)
    l2 = base + groups + np.arange(2 * groups, dtype=np.int64)
    l3 = base + np.arange(groups, dtype=np.int64)
    step = 1 << 15
    for out, a, b in ((l1, leaves[0::2], leaves[1::2]), (l2, l1[0::2], l1[1::2]), (l3, l2[0::2], l2[1::2])):
        for k0 in range(0, len(out), step):
            ops.averageProfile(out[k0:k0 + step], a[k0:k0 + step], b[k0:k0 + step])
    par = np.full(7 * groups, 1, np.int64)     # only the dense nodes are active targets
    par[:groups] = -1
    ops.set_parents(base, par)
    ops.set_out_distances(base, np.zeros(7 * groups, ops.dt), np.full(7 * groups, state.n_active, np.int64))
    nvec = ops.profile_nvectors(base, groups)
    S = ops.dt.itemsize
    alg = groups * (L * (S + 1) + 2 * S + S + 8) + int(nvec.sum()) * 4 * S
    pad = ((L + 15) // 16) * 16
    moved = groups * (pad + (24 * pad) // 64 + 4 + S + 4 + S + 3 * S) + int(nvec.sum()) * 4 * S
    ops.set_shard(base, base + groups)
    q = int(l3[5])
    for _ in range(3):
        ops.setBestHit(q, state.n_active, state.n_diff_allow, state.totdiam, 0, want_best=False, want_hits=False)
    ops.synchronize()
    ops.timer_start()
    for _ in range(10):
        ops.setBestHit(q, state.n_active, state.n_diff_allow, state.totdiam, 0, want_best=False, want_hits=False)
    ops.timer_stop_ms()
    ms, launches = ops.sweep_kernel_ms()
    ops.set_parents(base, np.full(7 * groups, 1, np.int64))   # retire them again
    ops.set_max_node(state.maxnode)
    if ms <= 0:
        return None
    ach, mov = alg / (ms * 1e-3) / 1e9, moved / (ms * 1e-3) / 1e9
    return dict(kernel="k_sweep_nt<float,MODE_CRIT>", phi=float(nvec.mean()) / L, targets=groups, launches=int(launches),
                avg_launch_ms=ms, algorithmic_bytes_per_launch=int(alg), moved_bytes_per_launch=int(moved),
                achieved=ach, frac=ach / HBM_PEAK_GBS, achieved_moved_gbs=mov, frac_moved=mov / HBM_PEAK_GBS)


def sweep_kernel_hash():
    """SHA-256 over the sources of the sweep kernels and of the layout they read: profiles/traffic.json (HBM bytes per launch
    from rocprofv3 PMC passes, tools/profile_collect.py) carries the hash it was measured at, and `roofline.traffic` is
    printed only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for f in ("vft_kernels_nj.h", "vft_layout.h", "vft_device.h"):
        h.update(open(os.path.join(ROOT, "veryfasttree_amd", "csrc", f), "rb").read())
    return h.hexdigest()


E2E = {
    # name: (n, L, mu, gap, seed, fastest, golden file, the reference's flags)
    "c3": (100000, 500, 0.03, 0.01, 3, True, "bb_c3_crc.npz", "-nt -fastest -noml -nome -nosupport"),
    "c4": (1000000, 200, 0.02, 0.01, 4, False, "bb_c4_crc.npz", "-nt -noml -nome -nosupport"),
}


def end_to_end(which, device, comm=None):
    """Wall-clock to a tree for BASELINE config C3 (100 000 nt x 500, `-nt -fastest`: top hits with the second-level lists,
    as the reference runs it at one thread) or C4 (1 000 000 nt x 200, `-nt`: the headline alignment): NJ phase, root,
    minimum-evolution branch lengths, Newick - what `VeryFastTree <flags> -noml -nome -nosupport` prints.  newick_crc is
    compared with the reference's own output for the same alignment (tests/golden/bb_c{3,4}_crc.npz, from
    oracle/_ref/VeryFastTree at one thread)."""
    import zlib
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick
    n, L, mu, gap, seed, fastest, golden, flags = E2E[which]
    codes = synth.random_descent_codes(n, L, 4, mu, gap, seed=seed)
    names = ["s%d" % k for k in range(n)]
    if comm is not None:   # count this tree's collectives
        comm.calls = comm.bytes = comm.device_calls = comm.device_bytes = 0
    t0 = time.perf_counter()
    tree = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 4, np.float32, max_nodes=3 * m, device=device), codes, names,
                     fastest=fastest, me_lengths=True, comm=comm)
    wall = time.perf_counter() - t0
    crc = zlib.crc32(tree.encode())
    out = dict(workload="%s_%dk_x%d_nt_%snj_tree" % (which, n // 1000, L, "fastest_" if fastest else ""), reference_flags=flags,
               wall_s=round(wall, 2), unique_seqs=int(len(np.unique(codes, axis=0))), newick_bytes=len(tree), newick_crc=crc)
    if comm is not None:   # what was actually split over the ranks: sweeps + leaf blocks (the join loop itself is replicated)
        out["allgathers"] = int(comm.calls)
        out["allgather_bytes"] = int(comm.bytes)
        out["allgather_bytes_by_phase"] = {"sweep_lists_device": int(comm.device_bytes), "leaf_blocks_host": int(comm.bytes - comm.device_bytes)}
    ref = os.path.join(ROOT, "tests", "golden", golden)
    if os.path.exists(ref):
        g = np.load(ref)
        want = int(g["newick_crc"])
        out["reference_newick_crc"] = want
        out["identical_to_reference"] = bool(want == crc)
        if "flags" in g:
            out["reference_tree_flags"] = bytes(g["flags"]).decode()
        if "reference_wall_s" in g:
            out["reference_wall_s_1_thread"] = round(float(g["reference_wall_s"]), 1)
    t6 = os.path.join(ROOT, "tests", "golden", "bb_%s_t6.npz" % which)
    if os.path.exists(t6):   # C4: the reference at six threads - a different tree than its own one-thread run (oracle/gen_fixtures.py c4_tree)
        g = np.load(t6)
        out["reference_tree_at_6_threads"] = dict(newick_crc=int(g["newick_crc"]), identical=bool(int(g["newick_crc"]) == crc),
                                                  one_thread_traced_joins_not_in_it=int(len(g["one_thread_traced_joins_not_in_this_tree"])))
    # the join order against the reference's `Join` trace as far as the reference got (C4: the one-thread reference needs more than
    # half a day for the million-sequence NJ phase; tests/golden/bb_c4_prefix.npz holds CRC-32s per 10 000 joins of what it wrote)
    pre = os.path.join(ROOT, "tests", "golden", "bb_%s_prefix.npz" % which)
    if os.path.exists(pre):
        from veryfasttree_amd.backend import last_join_crcs
        g = np.load(pre)
        chunk, n_joins, crcs = last_join_crcs()
        complete = bool(int(g["complete"])) if "complete" in g else False   # the finished trace: its last, shorter chunk is there too
        k = min(len(crcs), len(g["join_chunk_crc"])) if chunk == int(g["join_chunk"]) else 0
        if not complete and n_joins % chunk:   # (a prefix holds complete chunks only; this run's last entry is its shorter tail)
            k = min(k, len(crcs) - 1)
        out["reference_joins_compared"] = int(min(k * chunk, int(g["n_joins"])))
        out["reference_trace_complete"] = complete
        out["join_order_identical_to_reference_prefix"] = bool(k > 0 and np.array_equal(crcs[:k].astype(np.int64), g["join_chunk_crc"][:k]))
        if complete:
            out["join_order_identical_to_reference"] = bool(out["join_order_identical_to_reference_prefix"] and n_joins == int(g["n_joins"]))
    return out


E2E_FULL = {
    # BASELINE configs 2 and 5: the complete default pipeline (NJ, ME NNIs + 2 SPR rounds, ML NNIs with 20 CAT categories, model
    # fit, SH-like supports).  threads: the reference's `-threads T` schedule this backend follows for the refinement stages
    # (host/MLLengths.h "the subtree schedule": the walks of T-thread partitions advanced in lockstep, batches of quartets on the GPU).
    "c2": dict(n=10000, L=1000, nc=4, seed=2, dtype="float32", gtr=True, aa=None, threads=64, flags="-nt -gtr", golden="bb_c2_crc.npz"),
    # (C5 = SURVEY.md 8(d)'s alignment: mu 0.08, gaps 0.02, seed 5; rounds 4-5 ran C2's generator parameters here by mistake)
    "c5": dict(n=50000, L=300, nc=20, seed=5, mu=0.08, gap=0.02, dtype="float64", gtr=False, aa="lg", threads=128, flags="-lg -double-precision", golden="bb_c5_crc.npz"),
    # config C4's generator at 100 000 sequences with its real flags on the 64-thread schedule: the largest complete pipeline of the reference
    # this backend is pinned to byte for byte (tests/golden/thr_c4_100k_t64_crc.npz: `VeryFastTree -nt -threads 64`, 985 s on eight cores)
    "c4s": dict(n=100000, L=200, nc=4, seed=4, mu=0.02, dtype="float32", gtr=False, aa=None, threads=64, flags="-nt", golden=None,
                golden_threads="thr_c4_100k_t64_crc.npz"),
    # config C4 with its real flags (9 minutes)
    "c4": dict(n=1000000, L=200, nc=4, seed=4, mu=0.02, dtype="float32", gtr=False, aa=None, threads=1024, flags="-nt", golden=None),
}


def end_to_end_full(which, device, one_thread, comm=None):
    """Wall-clock to the final tree of BASELINE config C2 (10 000 nt x 1 000, `-nt -gtr`, float32) or C5 (50 000 aa x 300, `-lg`, float64):
    everything `VeryFastTree <flags>` does, supports included.  one_thread: the reference's one-thread order (its deterministic path;
    C2's tree is compared with tests/golden/bb_c2_crc.npz, the reference binary's own output); otherwise the schedule of a T-thread
    run of the reference (byte-identical to `VeryFastTree -threads T` under Jukes-Cantor, tests/test_gpu_threads.py; with a matrix
    model the reference's threaded runs are not reproducible and there is nothing to pin)."""
    import zlib
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import nj_newick, last_stage_seconds
    cfg = E2E_FULL[which]
    dt = np.float64 if cfg["dtype"] == "float64" else np.float32
    codes = synth.random_descent_codes(cfg["n"], cfg["L"], cfg["nc"], cfg.get("mu", 0.03), cfg.get("gap", 0.01), seed=cfg["seed"])
    names = ["s%d" % k for k in range(cfg["n"])]
    T = 1 if one_thread else cfg["threads"]
    kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, return_loglk=True, threads=T)
    if cfg["aa"]:
        kw["aa_model"] = cfg["aa"]
    if cfg["gtr"]:
        kw["gtr"] = True
    if comm is not None:   # sharded: the NJ sweeps and the lanes of the subtree schedule are split over the ranks
        kw["comm"] = comm
        comm.calls = comm.bytes = comm.device_calls = comm.device_bytes = 0
    # C2 in the one-thread order: the compiled reference itself (oracle/_ref/VeryFastTree, when the box has it) runs the same alignment with
    # the same flags on ONE host core beside this leg - a wall-clock of the reference measured in this run, on this box (~110 s)
    ref_run = None
    refbin = os.path.join(ROOT, "oracle", "_ref", "VeryFastTree")
    if one_thread and which == "c2" and comm is None and os.path.exists(refbin) and not os.environ.get("VFT_BENCH_NO_REFERENCE_RUN"):
        import subprocess, tempfile, threading
        tmpd = tempfile.mkdtemp(prefix="vft_bench_ref_")
        fa = os.path.join(tmpd, "c2.fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
        ref_run = dict(t0=time.perf_counter(), t1=None, tmpd=tmpd, out=os.path.join(tmpd, "c2.tree"))
        try:
            ref_run["proc"] = subprocess.Popen([refbin] + cfg["flags"].split() + ["-threads", "1", "-seed", "1", fa], stdout=open(ref_run["out"], "wb"),
                                               stderr=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS="1"))

            def _wait():
                ref_run["proc"].wait()
                ref_run["t1"] = time.perf_counter()
            ref_run["thread"] = threading.Thread(target=_wait, daemon=True)
            ref_run["thread"].start()
        except OSError:   # (a binary this box cannot run: the leg goes on without the side-by-side number)
            ref_run = None
    t0 = time.perf_counter()
    tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, cfg["nc"], dt, max_nodes=3 * m, device=device), codes, names, **kw)
    wall = time.perf_counter() - t0
    out = dict(workload="%s_%dk_x%d_%s_full_pipeline" % (which, cfg["n"] // 1000, cfg["L"], "nt" if cfg["nc"] == 4 else "aa"),
               reference_flags=cfg["flags"] + " -threads %d" % T, schedule_threads=T, dtype=cfg["dtype"], wall_s=round(wall, 2),
               tree_loglk=[round(float(x), 4) for x in loglk], newick_bytes=len(tree), newick_crc=zlib.crc32(tree.encode()),
               stages_s=last_stage_seconds())
    if comm is not None:
        from veryfasttree_amd.backend import last_lane_exchange
        lane_calls, lane_bytes = last_lane_exchange()
        out["allgathers"] = int(comm.calls)
        out["allgather_bytes_by_phase"] = {"sweep_lists_device": int(comm.device_bytes), "ml_lanes": int(lane_bytes),
                                           "other_host": int(comm.bytes - comm.device_bytes - lane_bytes)}
        out["ml_lane_allgathers"] = int(lane_calls)
    # The compiled reference with the same flags, timed by the builder on the host cores of ANOTHER box of this pool (tools/reference_walls.py
    # in round 5 -> the file named in `source`): a record carried along, not a measurement of this run.
    walls = os.path.join(ROOT, "profiles", "r05_reference_walls_gpu_box.json")
    if os.path.exists(walls):
        w = json.load(open(walls))
        rec = w.get("%s_threads_%d" % (which, T))
        if rec and not (which == "c5" and w.get("c5_alignment") != "random_descent_codes(50000, 300, 20, 0.08, 0.02, seed=5)"):
            out["reference_on_a_pool_box_builder_run"] = dict(wall_s=rec["wall_s"], threads=T, cpu=w.get("cpu"), cores=w.get("cores"),
                                                              source="profiles/r05_reference_walls_gpu_box.json (not measured in this run)",
                                                              same_tree_as_this_run=bool(rec.get("newick_crc") == out["newick_crc"]))
    if ref_run is not None:
        import shutil
        ref_run["thread"].join(timeout=900)
        if ref_run["t1"] is not None and ref_run["proc"].returncode == 0:
            rtree = open(ref_run["out"], "rb").read().decode().strip()
            out["reference_measured_in_this_run"] = dict(wall_s=round(ref_run["t1"] - ref_run["t0"], 1), threads=1, cpu=cpu_model(),
                                                         binary="oracle/_ref/VeryFastTree " + cfg["flags"] + " -threads 1 -seed 1",
                                                         beside="this leg's own run (one host core)",
                                                         same_tree_as_this_run=bool(zlib.crc32(rtree.encode()) == out["newick_crc"]))
        else:
            ref_run["proc"].kill()
        shutil.rmtree(ref_run["tmpd"], ignore_errors=True)
    gt = cfg.get("golden_threads")
    if not one_thread and gt and os.path.exists(os.path.join(ROOT, "tests", "golden", gt)):   # a complete run of the reference at T threads
        g = np.load(os.path.join(ROOT, "tests", "golden", gt))
        if int(g["threads"]) == T:
            out["reference_newick_crc"] = int(g["newick_crc"])
            out["identical_to_reference"] = bool(int(g["newick_crc"]) == out["newick_crc"] and int(g["newick_bytes"]) == len(tree))
            out["reference_wall_s_%d_threads_build_container_8_cores" % T] = round(float(g["reference_wall_s"]), 1)
            out["reference_nj_equals_its_one_thread_nj"] = bool(int(g["reference_nj_equals_its_one_thread_nj"]))
            want = g["loglk"]
            if len(want) and len(loglk):
                out["final_loglk_rel_diff"] = float(abs(loglk[-1] - want[-1]) / abs(want[-1]))
    if one_thread and cfg["golden"] and os.path.exists(os.path.join(ROOT, "tests", "golden", cfg["golden"])):
        g = np.load(os.path.join(ROOT, "tests", "golden", cfg["golden"]))
        out["reference_newick_crc"] = int(g["newick_crc"])
        out["identical_to_reference"] = bool(int(g["newick_crc"]) == out["newick_crc"])
        out["reference_wall_s_1_thread_build_container"] = round(float(g["reference_wall_s"]), 1)
        want = g["loglk"]
        if len(want) and len(loglk):
            out["final_loglk_rel_diff"] = float(abs(loglk[-1] - want[-1]) / abs(want[-1]))
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as CHILD processes (torch.distributed.run,
    one per GPU, rendezvous on 127.0.0.1) before anything in this process touches the GPU, relay what they print and exit
    with their code.  Never exec: this process may hold profiler / runtime state."""
    import subprocess
    # --standalone: the rendezvous store listens on a port the launcher itself picks and holds (binding a socket here, closing it
    # and passing the number on can lose the port to another launch on the same box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, env=env)
    sys.exit(res.returncode)


# the end-to-end legs of a single-GPU run: key -> (kind, config, one-thread order)
LEGS = [("e2e", "tree", "c3", False), ("e2e_c4", "tree", "c4", False), ("e2e_c2", "full", "c2", True), ("e2e_c2_threads", "full", "c2", False),
        ("e2e_c5_threads", "full", "c5", False), ("e2e_c4s_threads", "full", "c4s", False), ("e2e_c4_full_threads", "full", "c4", False),
        ("e2e_c5", "full", "c5", True)]


def wanted_legs(args):
    out = []
    for key, kind, which, one in LEGS:
        if kind == "tree" and which == "c4" and args.no_e2e_c4:
            continue
        if kind == "full" and (args.no_e2e_full or (key == "e2e_c4_full_threads" and args.no_e2e_c4_full) or (key == "e2e_c5" and args.no_e2e_c5_one_thread)):
            continue
        out.append((key, kind, which, one))
    return out


def run_leg(key):
    """one end-to-end leg in this process; its record as a JSON line on stdout (the child of launch_legs)"""
    kind, which, one = next((k, w, o) for kk, k, w, o in LEGS if kk == key)
    rec = end_to_end(which, 0) if kind == "tree" else end_to_end_full(which, 0, one)
    print(json.dumps(rec))


# what a leg took on the driver's box last round (BENCH_r05.json; C5's legs are on SURVEY 8(d)'s alignment from round 6 on and cost more):
# a leg is skipped when the time used so far plus this figure would pass --time-budget, so that the driver's clock cannot run out inside it
LEG_WALL_S = {"e2e": 25.0, "e2e_c4": 130.0, "e2e_c2": 130.0, "e2e_c2_threads": 15.0, "e2e_c5_threads": 70.0, "e2e_c4s_threads": 55.0, "e2e_c4_full_threads": 540.0, "e2e_c5": 430.0}


def launch_legs(args):
    """N = 1 with end-to-end legs: this process never touches the GPU.  The step measurement runs as one child process, every end-to-end
    tree / complete pipeline as a child of its own (a fresh process per tree, as a user would run them; a leg that fails - or takes the
    process down with it - costs its own record, not the line).  The merged JSON line is printed after the step measurement and AGAIN
    after every leg, each time a superset of the one before (the last complete line is the result): whatever ends this process early,
    the headline and every leg finished by then are on stdout."""
    import subprocess
    me = os.path.abspath(__file__)
    passed = [a for a in sys.argv[1:]]
    res = subprocess.run([sys.executable, me] + passed + ["--child", "main"], stdout=subprocess.PIPE)
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        sys.stdout.write(res.stdout.decode())
        sys.exit(res.returncode or 1)
    line = json.loads(lines[-1])
    legs = wanted_legs(args)
    line["legs_pending"] = [k for k, _, _, _ in legs]
    print(json.dumps(line), flush=True)
    for key, kind, which, one in legs:
        line["legs_pending"].remove(key)
        used = time.perf_counter() - T_START
        if used + LEG_WALL_S.get(key, 60.0) > args.time_budget:
            line[key] = {"workload": which, "skipped": "%.0f s used, the leg needs ~%.0f s, time budget %.0f s" % (used, LEG_WALL_S.get(key, 60.0), args.time_budget)}
            print(json.dumps(line), flush=True)
            continue
        print("bench.py: %.0f s - %s" % (used, key), file=sys.stderr, flush=True)
        r = subprocess.run([sys.executable, me, "--child", key], stdout=subprocess.PIPE)
        recs = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode == 0 and recs:
            line[key] = json.loads(recs[-1])
        else:
            line[key] = {"workload": which, "error": "the leg's process ended with code %d" % r.returncode}
        print(json.dumps(line), flush=True)


def launch_only(args, world, rank):
    """test hook (VFT_BENCH_LAUNCH_ONLY=1, CPU): the launcher, the rendezvous and one collective, no GPU work"""
    import torch
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("VFT_BENCH_BACKEND", "gloo"))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    assert dist.get_world_size() == args.gpus == world
    if rank == 0:
        print(json.dumps(dict(metric="profile-ops/sec", n_gpus=world, world=dist.get_world_size(), launch_only=True,
                              rank_sum=float(t.item()))))
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("VFT_BENCH_SAME_DEVICE"):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("VFT_BENCH_LAUNCH_ONLY"):
        return launch_only(args, world, int(os.environ.get("RANK", "0")))
    single = world == 1 and not os.environ.get("VFT_BENCH_FORCE_DIST")
    if args.child and os.environ.get("VFT_BENCH_FAKE_CHILD"):   # test hook (CPU): canned records, the leg named by the variable dies
        if args.child == os.environ["VFT_BENCH_FAKE_CHILD"]:
            os._exit(139)
        print(json.dumps({"metric": "profile-ops/sec", "value": 1.0} if args.child == "main" else {"workload": args.child, "wall_s": 1.0}))
        return
    if args.child and args.child != "main":
        return run_leg(args.child)
    # (the end-to-end legs belong to the headline workload: a reduced --n-seqs / --n-pos run is the step measurement alone, in this process)
    if single and args.child is None and not args.in_process and not args.no_e2e and (args.n_seqs, args.n_pos) == (1000000, 200) and wanted_legs(args):
        return launch_legs(args)
    if args.child == "main":
        args.no_e2e = True
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (not used by the driver): run the multi-rank flow on ONE GPU with gloo + host tensors
    backend = os.environ.get("VFT_BENCH_BACKEND", "nccl")
    if os.environ.get("VFT_BENCH_SAME_DEVICE"):
        local_rank = 0
    # test hook: take the multi-rank code path (RCCL all-gather + device merge) even with one rank
    use_dist = world > 1 or bool(os.environ.get("VFT_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.workload import TopHitsState, merge_hits, shard_range, sweep_cost_weights

    n, L = args.n_seqs, args.n_pos
    n_join = n // 4
    m = int(0.5 + np.sqrt(n))
    k = 2 * m
    t_setup = time.perf_counter()
    codes = synth.random_descent_codes(n, L, 4, 0.02, 0.01, seed=4)
    ops = HipProfileOps(n, L, 4, np.float32, device=local_rank)
    state = TopHitsState(ops, codes, n_join)
    # shard the target id range over the ranks at tile boundaries
    # (by cost, not by id count: the joined leaves sit at the low ids and an internal profile costs ~5.5 leaves)
    lo, hi = shard_range(state.maxnode, rank, world, sweep_cost_weights(state.parent, n))
    ops.set_shard(lo, hi)
    seeds = []
    leaf_act = state.active[state.active < n]
    int_act = state.active[state.active >= n]
    for s in range(4):
        seeds.append(int(leaf_act[(s * 7919 + 13) % len(leaf_act)]))
        seeds.append(int(int_act[(s * 104729 + 7) % len(int_act)]))
    setup_s = time.perf_counter() - t_setup

    hit_dt = ops.hit_dtype
    S = len(seeds)
    seeds_arr = np.asarray(seeds, np.int64)
    if use_dist:
        # per step: this rank's [S][k] records, the ranks' blocks all-gathered ([world][S][k]), one batched merge
        rec = S * k * hit_dt.itemsize
        gdev = "cuda" if backend == "nccl" else "cpu"
        d_mine = torch.zeros(rec, dtype=torch.uint8, device="cuda")
        d_all = torch.zeros(world * rec, dtype=torch.uint8, device=gdev)
        ops.set_stream(torch.cuda.current_stream().cuda_stream)

    def one_step():
        # the S seeds of a step go down in one call (vft_sweep_batch): S sweeps back to back on the stream, one batched
        # top-k selection, one host wait - how the NJ driver refreshes the top-hit lists of a batch of seeds
        if not use_dist:
            # (the records are read where the selection leaves them - the host-mapped result blocks - as NJDriver::seedSweep reads them)
            hits, _ = ops.setBestHitBatch(seeds_arr, state.n_active, state.n_diff_allow, state.totdiam, k, view=not os.environ.get("VFT_BENCH_COPY_HITS"))
            return hits
        ops.setBestHitBatch(seeds_arr, state.n_active, state.n_diff_allow, state.totdiam, k, d_hits=d_mine.data_ptr(),
                            want_hits=False)
        if backend == "nccl":
            # RCCL over xGMI: S*k records per rank, one all-gather per step; gather and merge are stream-ordered and
            # the merged lists come back through the mapped result block with a single wait
            dist.all_gather_into_tensor(d_all, d_mine)
            return ops.merge_hits_batch(d_all.data_ptr(), world, S, k)
        torch.cuda.current_stream().synchronize()
        dist.all_gather_into_tensor(d_all, d_mine.cpu())
        gathered = d_all.cuda()
        return ops.merge_hits_batch(gathered.data_ptr(), world, S, k)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    last = [np.array(h) for h in last]   # (views of the result blocks: the passes below sweep again)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ops_per_step = len(seeds) * state.n_active
    value = ops_per_step * args.steps / elapsed

    # A sweep is ONE launch over all targets (vft_kernels_nj.h): k_sweep_nt<MODE_CRIT_LEAFQ> for a leaf seed, k_sweep_nt_both for a
    # profile seed - the profile x profile ProfileDist over the internal targets with the table walk over the leaf targets riding on
    # the same CUs.  Timed with HIP events on the kernels' own stream over one more pass per kind of seed (events between the
    # launches would open gaps in the timed region above); the roofline prices the average launch against its algorithmic bytes.
    seeds_leaf = np.asarray([q for q in seeds if q < n], np.int64)
    seeds_prof = np.asarray([q for q in seeds if q >= n], np.int64)

    def timed_pass(which):
        ops.timer_start()
        ops.setBestHitBatch(which, state.n_active, state.n_diff_allow, state.totdiam, k)
        ops.timer_stop_ms()
        return ops.sweep_kernel_ms() + (ops.sweep_kernel_sweeps(),)

    # (a launch can stand for several sweeps: runs of leaf seeds share a pass over the targets, k_sweep_nt_leafq_multi)
    leaf_ms, n_leaf_launches, n_leaf_sweeps = timed_pass(seeds_leaf) if len(seeds_leaf) and not use_dist else (0.0, 0, 0)
    prof_ms, n_prof_launches, n_prof_sweeps = timed_pass(seeds_prof) if len(seeds_prof) and not use_dist else (0.0, 0, 0)
    if use_dist:   # (sharded: one instrumented step as it is)
        ops.timer_start()
        one_step()
        ops.timer_stop_ms()
        kern_ms, launches = ops.sweep_kernel_ms()
        sweeps = ops.sweep_kernel_sweeps()
    else:
        launches = n_leaf_launches + n_prof_launches
        sweeps = n_leaf_sweeps + n_prof_sweeps
        kern_ms = (leaf_ms * n_leaf_launches + prof_ms * n_prof_launches) / max(launches, 1)
    ab = state.algorithmic_bytes_per_sweep(lo, hi)
    sweeps_per_launch = sweeps / max(launches, 1)
    alg_sweep = float(ab["leaf"] + ab["internal"])            # the algorithm's bytes of ONE seed's sweep
    alg_main = alg_sweep * sweeps_per_launch                  # ... times the sweeps an average launch processes
    moved_main = float(ab["moved_leaf"] + ab["moved_internal"])   # what one pass over the targets moves, however many seeds ride on it
    # Counters of the same kernels from the rocprofv3 PMC passes kept under profiles/ (rocprofv3 cannot run inside this process;
    # tools/profile_round.sh + tools/profile_collect.py -> profiles/traffic.json, used only while the hash of the sweep kernels'
    # sources recorded there still matches - counters read from other kernel sources say nothing: null):
    #   HBM bytes per launch = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE
    #   VALU instructions per launch = SQ_INSTS_VALU (wavefront instructions; x 64 lanes)
    traffic = None
    pmc = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and (n, L, world) == (1000000, 200, 1):
        t = json.load(open(tpath))
        if t.get("kernel_sources_sha256") == sweep_kernel_hash():
            pmc = t.get("sweep_launch_average", {})
            traffic = pmc.get("bytes_per_launch")
    # peak VALU issue: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz (MI355X_MICROARCH.md) = 3.93e13 lane-instructions/s
    VALU_PEAK = 256 * 64 * 2.4e9

    def per_kind(ms, nl, ns, key):
        spl = ns / max(nl, 1)
        rec = pmc.get(key, {})
        out = dict(avg_launch_ms=ms, sweeps_per_launch=spl,
                   algorithmic_equiv_gbs=alg_sweep * spl / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                   moved_model_gbs=moved_main / (ms * 1e-3) / 1e9 if ms > 0 else 0.0)
        if rec.get("bytes_per_launch") and ms > 0:
            out["hbm_bytes_per_launch_pmc"] = int(rec["bytes_per_launch"])
            out["hbm_frac_physical"] = rec["bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if rec.get("valu_wave_insts_per_launch") and ms > 0:
            out["valu_lane_insts_per_launch"] = int(rec["valu_wave_insts_per_launch"] * 64)
            out["valu_frac"] = rec["valu_wave_insts_per_launch"] * 64 / (ms * 1e-3) / VALU_PEAK
            if rec.get("valu_busy_frac_pmc") is not None:
                out["valu_busy_frac_pmc"] = rec["valu_busy_frac_pmc"]
        return out

    # `achieved` / `frac` price what the memory system SEES: the bytes one launch moves (the PMC figure while it belongs to these
    # kernel sources, else the layout model of workload.py) / the launch's duration - never above the peak.  With S seeds sharing a
    # pass (vft_sweep_batch) the algorithm's own byte count, SURVEY 8(d)'s per-sweep figure x the sweeps a launch carries, is an
    # EQUIVALENT rate only (`algorithmic_equiv_gbs`: what a seed-by-seed stream would have to sustain) and is not divided by the peak.
    phys = float(traffic) if traffic else moved_main
    achieved = phys / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    valu = None
    if pmc.get("valu_wave_insts_per_launch") and kern_ms > 0:
        li = pmc["valu_wave_insts_per_launch"] * 64.0
        valu = dict(lane_insts_per_launch=int(li), achieved_lane_insts_per_s=li / (kern_ms * 1e-3), peak_lane_insts_per_s=VALU_PEAK,
                    frac=li / (kern_ms * 1e-3) / VALU_PEAK, source="profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU)")
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, bytes_source="pmc" if traffic else "layout_model",
                    limited_by="valu (instruction issue: S seeds share one pass over the targets, the stream is read once per S sweeps)",
                    valu=valu,
                    kernel="k_sweep_nt_profq_multi<float,4> (four profile seeds per launch) / k_sweep_nt_leafq_multi<float,4> (four leaf seeds per launch)",
                    launches=int(launches), sweeps=int(sweeps), sweeps_per_launch=sweeps_per_launch, avg_launch_ms=kern_ms,
                    algorithmic_bytes_per_sweep=int(alg_sweep), algorithmic_bytes_per_launch=int(alg_main),
                    algorithmic_equiv_gbs=alg_main / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0,
                    moved_bytes_per_launch_model=int(moved_main),
                    phi=ab["phi"],
                    profile_seed_launch=per_kind(prof_ms, n_prof_launches, n_prof_sweeps, "profile_seed_instance") if not use_dist else None,
                    leaf_seed_launch=per_kind(leaf_ms, n_leaf_launches, n_leaf_sweeps, "leaf_seed_instance") if not use_dist else None,
                    step=dict(algorithmic_bytes=int(alg_sweep * len(seeds)), algorithmic_equiv_gbs=alg_sweep * len(seeds) / (elapsed / args.steps) / 1e9))
    assert roofline["frac"] <= 1.0, "a physical fraction above the peak: the byte count is wrong"

    line = dict(metric="profile-ops/sec", value=value, unit="profile-ops/s", n_gpus=world, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                scaling="strong", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload="c4_1M_x200_nt_tophits" if (n, L) == (1000000, 200) else "custom_tophits",
                            n_seqs=n, n_pos=L, internal_profiles=n_join, active_nodes=int(state.n_active),
                            seeds_per_step=len(seeds), top_k=k, sharding="target-range x%d" % world,
                            setup_s=round(setup_s, 1)),
                roofline=roofline)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N = 1 only (the other ranks would wait for it)
        line["cpu_baseline"] = cpu_baseline(state, codes, ops)
    elif rank == 0:
        line["cpu_baseline"] = None
    if world == 1 and not args.no_dense and (n, L) == (1000000, 200):
        ops.set_shard(0, state.maxnode)
        line["roofline"]["dense_profiles"] = dense_profile_roofline(ops, state, n, L)
        ops.set_shard(lo, hi)
    # size-independent sanity of the last step: sorted order and no duplicate ids
    h = last[0]
    hv = h[h["j"] >= 0]
    assert np.all(np.diff(hv["criterion"]) >= 0) and len(np.unique(hv["j"])) == len(hv)
    # a checksum of the last step's lists (every seed: ids and criteria), so that runs with different rank counts can
    # be compared: the sharded + merged lists must be the unsharded ones
    import zlib
    crc = 0
    for hits in last:
        crc = zlib.crc32(np.ascontiguousarray(hits["j"]).tobytes(), crc)
        crc = zlib.crc32(np.ascontiguousarray(hits["criterion"]).tobytes(), crc)
    line["hits_crc"] = crc
    line["world"] = world
    if use_dist:   # what each rank swept (target ids [lo, hi) of every sweep) and what the exchange moved per step
        sh = torch.tensor([float(lo), float(hi)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        allsh = [torch.zeros_like(sh) for _ in range(world)]
        dist.all_gather(allsh, sh)
        line["shards"] = [[int(x[0].item()), int(x[1].item())] for x in allsh]
        line["allgathers_per_step"] = 1
        line["allgather_bytes_per_step"] = int(world * rec)
    if not args.no_e2e and (n, L) == (1000000, 200):
        # the other half of BASELINE's metric: wall-clock to a tree.  With several ranks the C++ driver runs on every rank
        # (replicated join decisions) and splits its sweeps and leaf blocks over them (include/vft_host.h, vft_comm)
        ops.close()
        del state
        comm = None
        if use_dist:
            from veryfasttree_amd.backend import TorchComm
            comm = TorchComm(dist, local_rank)
            barrier()
        only = [k for k in os.environ.get("VFT_BENCH_ONLY_LEGS", "").split(",") if k]   # (tools: a subset of the legs, in-process runs)
        for which, key in (("c3", "e2e"), ("c4", "e2e_c4")):
            if which == "c4" and args.no_e2e_c4:
                continue
            if only and key not in only:
                continue
            e2e, ok = None, 1.0
            if rank == 0:
                print("bench.py: %.0f s - end-to-end tree %s" % (time.perf_counter() - T_START, which), file=sys.stderr, flush=True)
            try:
                e2e = end_to_end(which, local_rank, comm)
            except Exception as exc:   # the headline line must still be printed
                e2e, ok = {"workload": which, "error": repr(exc)}, 0.0
                if use_dist and comm is not None:
                    # The other ranks are inside the C++ driver's all-gathers for this tree: joining the all-reduces below from
                    # here would pair them with collectives of another kind and hang the job until the backend's timeout.  Say
                    # what happened and leave with an error code - the launcher (torch.distributed.run) then ends the job.
                    print("bench.py: rank %d failed in the sharded end-to-end run (%r); ending the job" % (rank, exc), file=sys.stderr, flush=True)
                    os._exit(17)
            if use_dist:
                # every rank takes part in both collectives whatever happened on it: first whether all succeeded, then the
                # slowest rank's wall-clock (a failure on one rank must not leave the others waiting in an all-reduce)
                gdev = "cuda" if backend == "nccl" else "cpu"
                t = torch.tensor([ok], dtype=torch.float64, device=gdev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                w = torch.tensor([float(e2e.get("wall_s", 0.0))], dtype=torch.float64, device=gdev)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                c = torch.tensor([float(e2e.get("newick_crc", -1)), -float(e2e.get("newick_crc", -1))], dtype=torch.float64, device=gdev)
                dist.all_reduce(c, op=dist.ReduceOp.MAX)
                if float(t.item()) < 1.0 and ok:
                    e2e = {"workload": which, "error": "another rank failed"}
                elif ok:
                    e2e["wall_s"] = round(float(w.item()), 2)
                    e2e["same_crc_on_all_ranks"] = bool(c[0].item() == -c[1].item())
            line[key] = e2e
        if not args.no_e2e_full and not use_dist:
            # the complete pipelines (refinement + maximum likelihood + supports): not sharded, rank 0 only.  C2 and C5 in the
            # reference's one-thread order (its deterministic path: trees compared with the reference binary's own output) and on
            # the schedule of a T-thread run; C4 with its real flags on the schedule of 1 024 threads.
            legs = [("e2e_c2", "c2", True), ("e2e_c2_threads", "c2", False), ("e2e_c5_threads", "c5", False), ("e2e_c4s_threads", "c4s", False)]
            if not args.no_e2e_c4_full:
                legs.append(("e2e_c4_full_threads", "c4", False))
            if not args.no_e2e_c5_one_thread:
                legs.append(("e2e_c5", "c5", True))
            for key, which, one in legs:
                if only and key not in only:
                    continue
                if time.perf_counter() - T_START > args.time_budget:
                    line[key] = {"workload": which, "skipped": "time budget of %.0f s used up" % args.time_budget}
                    continue
                print("bench.py: %.0f s - complete pipeline %s" % (time.perf_counter() - T_START, key), file=sys.stderr, flush=True)
                try:
                    line[key] = end_to_end_full(which, local_rank, one)
                except Exception as exc:
                    line[key] = {"workload": which, "error": repr(exc)}
        if not args.no_e2e_full and use_dist:
            # sharded: config C2's complete pipeline on the 64-thread schedule with the lanes split over the ranks (host/MLLengths.h
            # "lanes across ranks"); every rank runs it, the slowest rank's wall-clock counts
            try:
                leg = end_to_end_full("c2", local_rank, False, comm)
            except Exception as exc:
                print("bench.py: rank %d failed in the sharded complete pipeline (%r); ending the job" % (rank, exc), file=sys.stderr, flush=True)
                os._exit(18)
            gdev = "cuda" if backend == "nccl" else "cpu"
            w = torch.tensor([float(leg["wall_s"])], dtype=torch.float64, device=gdev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            c2 = torch.tensor([float(leg["newick_crc"]), -float(leg["newick_crc"])], dtype=torch.float64, device=gdev)
            dist.all_reduce(c2, op=dist.ReduceOp.MAX)
            leg["wall_s"] = round(float(w.item()), 2)
            leg["same_crc_on_all_ranks"] = bool(c2[0].item() == -c2[1].item())
            line["e2e_c2_threads"] = leg
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
