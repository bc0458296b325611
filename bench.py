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
# small device->host copies (merged hit lists in the multi-GPU path) go through blit kernels instead of the SDMA
# engines, whose fixed latency is hundreds of microseconds on this platform; must be set before HIP initialises
os.environ.setdefault("HSA_ENABLE_SDMA", "0")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-seqs", type=int, default=1000000)
    ap.add_argument("--n-pos", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(state, codes, ops, budget_s=15.0):
    """The oracle (oracle/vft_oracle.c, a scalar C port pinned to the reference) timed on one host core over a
    bounded sample of the same workload: one internal-node seed against 4000 active leaves + 2000 internal
    profiles, repeated until ~budget_s of CPU time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import Oracle
    orc = Oracle(ops.dt)
    n_leaf, n_int = 4000, 2000
    n = state.n_seqs
    leaf_ids = state.active[state.active < n][:n_leaf]
    int_ids = state.active[state.active >= n][:n_int]
    profs = [orc.leaf_profile(codes[i], ops.n_codes) for i in leaf_ids]
    profs += [ops.profile_download(int(v)) for v in int_ids]
    W = np.stack([p[0] for p in profs]); Cc = np.stack([p[1] for p in profs]); F = np.stack([p[2] for p in profs])
    m = len(profs)
    outp, _ = ops.out_profile_download()
    z = np.zeros(m, ops.dt)
    st = orc.state(len(leaf_ids), W, Cc, F, np.full(m, -1), z, W.sum(1).astype(ops.dt), z, state.totdiam, outp)
    od = np.zeros(m, ops.dt)
    na = np.full(m, state.n_active)
    t0 = time.perf_counter()
    done = 0
    q = 0
    while time.perf_counter() - t0 < budget_s:
        query = (len(leaf_ids) + q) if q % 2 == 0 else q   # alternate internal / leaf seeds like the GPU step
        orc.set_best_hit(st, query % m, state.n_active, state.n_diff_allow, od, na)
        done += m
        q += 1
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="profile-ops/s", cores=1, kind="port",
                sample="%d sweeps of one seed vs %d leaves + %d internal profiles (same alignment), oracle C port, "
                       "1 thread" % (q, len(leaf_ids), len(int_ids)))


def main():
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
            hits, _ = ops.setBestHitBatch(seeds_arr, state.n_active, state.n_diff_allow, state.totdiam, k)
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
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ops_per_step = len(seeds) * state.n_active
    value = ops_per_step * args.steps / elapsed

    # A sweep is two kernels (vft_kernels_nj.h): k_sweep_nt - every target of a leaf seed, the internal-profile
    # targets of a profile seed - and k_sweep_nt_table - the leaf targets of a profile seed.  Both are timed with HIP
    # events on their own stream over one more step; k_sweep_nt is the dominant one (roofline), the other is reported
    # beside it, and "sweep" prices the whole sweep (both kernels) against the same peak.
    ops.timer_start()
    one_step()
    ops.timer_stop_ms()
    kern_ms, launches = ops.sweep_kernel_ms()
    tab_ms, _ = ops.sweep_table_kernel_ms()
    ab = state.algorithmic_bytes_per_sweep(lo, hi)
    n_leaf_seeds = sum(1 for q in seeds if q < n)
    n_prof_seeds = len(seeds) - n_leaf_seeds
    # leaf seed: k_sweep_nt sees all targets; profile seed: the internal ones (+ < 1024 leaves of the range remainder)
    alg_main = (n_leaf_seeds * (ab["leaf"] + ab["internal"]) + n_prof_seeds * ab["internal"]) / float(len(seeds))
    moved_main = (n_leaf_seeds * (ab["moved_leaf"] + ab["moved_internal"]) + n_prof_seeds * ab["moved_internal"]) / float(len(seeds))
    alg_tab = n_prof_seeds * ab["leaf"] / float(len(seeds))          # tab_ms is averaged over all sweeps as well
    achieved = alg_main / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # HBM traffic of the same kernel from the PMC passes kept under profiles/ (rocprofv3 cannot run inside this
    # process): FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, bytes per launch
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and (n, L, world) == (1000000, 200, 1):
        traffic = json.load(open(tpath)).get("k_sweep_nt<float,MODE_CRIT>", {}).get("bytes_per_launch")
    both_ms = kern_ms + tab_ms
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, kernel="k_sweep_nt<float,MODE_CRIT>", launches=int(launches),
                    avg_launch_ms=kern_ms, algorithmic_bytes_per_launch=int(alg_main),
                    moved_bytes_per_launch=int(moved_main),
                    achieved_moved_gbs=moved_main / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0,
                    phi=ab["phi"],
                    table_kernel=dict(kernel="k_sweep_nt_table<float,MODE_CRIT>", avg_ms_per_sweep=tab_ms,
                                      algorithmic_bytes_per_sweep=int(alg_tab), bound="lds",
                                      achieved_gbs=alg_tab / (tab_ms * 1e-3) / 1e9 if tab_ms > 0 else 0.0),
                    sweep=dict(avg_ms=both_ms, algorithmic_bytes=int(ab["leaf"] + ab["internal"]),
                               achieved_gbs=(ab["leaf"] + ab["internal"]) / (both_ms * 1e-3) / 1e9 if both_ms > 0 else 0.0,
                               frac=(ab["leaf"] + ab["internal"]) / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if both_ms > 0 else 0.0))

    line = dict(metric="profile-ops/sec", value=value, unit="profile-ops/s", n_gpus=world, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                scaling="strong", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload="c4_1M_x200_nt_tophits" if (n, L) == (1000000, 200) else "custom_tophits",
                            n_seqs=n, n_pos=L, internal_profiles=n_join, active_nodes=int(state.n_active),
                            seeds_per_step=len(seeds), top_k=k, sharding="target-range x%d" % world,
                            setup_s=round(setup_s, 1)),
                roofline=roofline)
    if rank == 0 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(state, codes, ops)
    elif rank == 0:
        line["cpu_baseline"] = None
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
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
