"""The N>1 path on CPU: two gloo ranks shard the target id range of a sweep, each keeps its local top-k in the
reference's order, the lists are all-gathered and merged — the result must equal the single-rank top-k.
Per-shard sweeps are produced by the oracle here (no GPU in this suite); on the GPU the same records come from
vft_sweep(d_hits=...) and the same merge runs (bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as G
from oracle import Oracle
from veryfasttree_amd.backend import HIT_F32
from veryfasttree_amd.workload import merge_hits, shard_range


def _local_topk(crit, dist_, weight, lo, hi, k):
    orc = Oracle(np.float32)
    sub = crit[lo:hi]
    order = orc.sort_hits(sub)
    order = order[sub[order] < np.float32(1e20)][:k] + lo
    h = np.zeros(k, HIT_F32)
    h["j"] = -1
    h["criterion"] = 1e20
    h["dist"] = 1e20
    n = len(order)
    h["j"][:n] = order
    h["criterion"][:n] = crit[order]
    h["dist"][:n] = dist_[order]
    h["weight"][:n] = weight[order]
    return h


def _worker(rank, world, port, k, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = G.load("wb_nt_f32")
    key = "mid1.sweep1"
    crit, dd, ww = d[key + ".crit"], d[key + ".dist"], d[key + ".weight"]
    lo, hi = shard_range(len(crit), rank, world)
    mine = _local_topk(crit, dd, ww, lo, hi, k)
    t_mine = torch.from_numpy(mine.view(np.uint8).copy())
    t_all = torch.zeros(world * t_mine.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(t_all, t_mine)
    allh = t_all.numpy().view(HIT_F32).reshape(world, k)
    merged = merge_hits(list(allh), k)
    if rank == 0:
        np.save(out, merged)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_topk_equals_single_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    k = 24
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(2, port, k, out), nprocs=2, join=True)
    merged = np.load(out)
    d = G.load("wb_nt_f32")
    key = "mid1.sweep1"
    crit = d[key + ".crit"]
    want = d[key + ".sorted_j"]          # the reference's own sort of the whole sweep
    want = want[crit[want] < np.float32(1e20)][:k]
    assert np.array_equal(merged["j"][:len(want)], want)
    assert np.array_equal(merged["criterion"][:len(want)], crit[want])


def test_shard_ranges_tile_aligned_and_cover():
    for maxnode in (1, 63, 64, 65, 1000, 1500000):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_range(maxnode, r, world)
                assert lo % 64 == 0 and lo == prev and hi >= lo
                prev = hi
            assert prev == maxnode


def test_merge_breaks_ties_by_descending_id():
    a = np.zeros(3, HIT_F32)
    a["j"] = [5, 9, -1]
    a["criterion"] = [0.5, 0.7, 1e20]
    b = np.zeros(3, HIT_F32)
    b["j"] = [70, 64, 66]
    b["criterion"] = [0.5, 0.5, 0.9]
    m = merge_hits([a, b], 4)
    assert list(m["j"]) == [70, 64, 5, 9]


def _batch_worker(rank, world, port, k, out):
    """The batched exchange of bench.py: every rank holds [seeds][k] records, ONE all-gather gives [ranks][seeds][k]."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = G.load("wb_nt_f32")
    keys = ["mid1.sweep0", "mid1.sweep1", "mid1.sweep2"]
    n = len(d[keys[0] + ".crit"])
    lo, hi = shard_range(n, rank, world)
    mine = np.stack([_local_topk(d[key + ".crit"], d[key + ".dist"], d[key + ".weight"], lo, hi, k) for key in keys])
    t_mine = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
    t_all = torch.zeros(world * t_mine.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(t_all, t_mine)
    allh = t_all.numpy().view(HIT_F32).reshape(world, len(keys), k)      # vft_merge_hits_batch's input layout
    merged = np.stack([merge_hits([allh[r, s] for r in range(world)], k) for s in range(len(keys))])
    if rank == 0:
        np.save(out, merged)
    dist.barrier()
    dist.destroy_process_group()


def test_batched_exchange_of_several_seeds(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    k = 24
    out = str(tmp_path / "merged_batch.npy")
    mp.spawn(_batch_worker, args=(2, port, k, out), nprocs=2, join=True)
    merged = np.load(out)
    d = G.load("wb_nt_f32")
    for s, key in enumerate(["mid1.sweep0", "mid1.sweep1", "mid1.sweep2"]):
        crit = d[key + ".crit"]
        want = d[key + ".sorted_j"]
        want = want[crit[want] < np.float32(1e20)][:k]
        assert np.array_equal(merged[s]["j"][:len(want)], want), key
        assert np.array_equal(merged[s]["criterion"][:len(want)], crit[want]), key


class _ShardedOps:
    """OracleOps whose one-vs-all sweep is split over the gloo ranks exactly the way the C++ driver splits it over GPUs
    (host/NJDriver.h::sweep): this rank's tiles of the target ids, its local top-k, one all-gather, the merge in the
    reference's order.  Everything else - the lazy out-distance refreshes included - is replicated."""

    def __init__(self, ops, rank, world):
        self._ops, self._rank, self._world = ops, rank, world
        self.exchanges = 0

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def setBestHit(self, query, n_active, n_diff_allow, totdiam, k, want_best=True, d_hits=None, want_hits=True):
        ops = self._ops
        n = ops.maxnode
        # replicated: every rank evaluates (and lazily refreshes) like a single rank would ...
        crit = np.full(n, 1e20, ops.dt)
        dd = np.full(n, 1e20, ops.dt)
        ww = np.zeros(n, ops.dt)
        for j in range(n):
            if ops.parent[j] >= 0:
                continue
            dd[j], ww[j], crit[j] = ops._dist_crit(query, j, n_active, n_diff_allow, totdiam)
        # ... but only its own tiles of the result enter its list
        tiles = (n + 63) // 64
        per = (tiles + self._world - 1) // self._world
        lo, hi = min(n, self._rank * per * 64), min(n, (self._rank + 1) * per * 64)
        mine = _local_topk(crit, dd, ww, lo, hi, k)
        t_mine = torch.from_numpy(mine.view(np.uint8).copy())
        t_all = torch.zeros(self._world * t_mine.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(t_all, t_mine)
        self.exchanges += 1
        allh = t_all.numpy().view(HIT_F32).reshape(self._world, k)
        return merge_hits(list(allh), k), -1


def _driver_worker(rank, world, port, name, fastest, second, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_ops import OracleOps
    from test_nj_driver_cpu import unique_codes
    from nj_driver_py import NJDriver
    d = G.load(name)
    codes = unique_codes(d["codes"])
    ops = _ShardedOps(OracleOps(codes.shape[0], codes.shape[1], 4, np.float32), rank, world)
    drv = NJDriver(ops, codes, fastest=fastest, use_tophits_2nd=second)
    if fastest:
        drv.tophits_refresh = 0.5
    joins = drv.run()
    got = np.array([(a, b, c) for a, b, c, _ in joins], dtype=np.int64)
    np.save(out % rank, got)
    assert ops.exchanges > 0
    dist.barrier()
    dist.destroy_process_group()


def test_nj_driver_join_order_does_not_depend_on_the_rank_count(tmp_path):
    """The NJ driver (Python prototype of host/NJDriver.h on the CPU oracle) with its sweeps split over two gloo ranks:
    both ranks reproduce the reference's join order, i.e. what one rank produces."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "joins_%d.npy")
    mp.spawn(_driver_worker, args=(2, port, "bb_nt_200", False, False, out), nprocs=2, join=True)
    want = G.load("bb_nt_200")["joins"]
    for r in range(2):
        assert np.array_equal(np.load(out % r), want)


def _lane_worker(rank, world, port, out):
    """The lanes-across-ranks exchange of host/MLLengths.h over gloo, with the C++ layout functions themselves (vft_nj_lane_share): a
    batch of K verdict records, every rank fills in its share [k0, k1), sends its padded share, and finds item t of the batch at record
    laneRecord(t, per) of what it received."""
    import ctypes as C
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from veryfasttree_amd.backend import load_host_library
    lib = load_host_library()
    rec = 32 + 5 * 4   # a vft_quartet_nni + five float lengths, as runNNILanes packs them
    ok = True
    for K in (1, 2, 3, 5, 64, 65, 513):
        o = np.zeros(4, np.int64)
        assert lib.vft_nj_lane_share(C.c_int64(K), C.c_int32(world), C.c_int32(rank), C.c_int64(0), o.ctypes.data_as(C.c_void_p)) == 0
        per, k0, k1 = int(o[0]), int(o[1]), int(o[2])
        truth = np.arange(K * rec, dtype=np.int64).astype(np.uint8).reshape(K, rec) ^ np.uint8(K & 0xFF)   # the record every rank would compute for item t
        send = np.zeros((per, rec), np.uint8)
        send[:k1 - k0] = truth[k0:k1]
        t_all = torch.zeros(world * per * rec, dtype=torch.uint8)
        dist.all_gather_into_tensor(t_all, torch.from_numpy(send.reshape(-1).copy()))
        got = t_all.numpy().reshape(world * per, rec)
        covered = np.zeros(K, bool)
        for r in range(world):   # the shares tile the batch
            assert lib.vft_nj_lane_share(C.c_int64(K), C.c_int32(world), C.c_int32(r), C.c_int64(0), o.ctypes.data_as(C.c_void_p)) == 0
            assert int(o[0]) == per and 0 <= o[1] <= o[2] <= K
            covered[int(o[1]):int(o[2])] = True
        ok &= bool(covered.all())
        for t in range(K):
            assert lib.vft_nj_lane_share(C.c_int64(K), C.c_int32(world), C.c_int32(rank), C.c_int64(t), o.ctypes.data_as(C.c_void_p)) == 0
            ok &= bool(np.array_equal(got[int(o[3])], truth[t]))
    if rank == 0:
        np.save(out, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_lane_exchange_layout_over_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "lanes.npy")
    mp.spawn(_lane_worker, args=(2, port, out), nprocs=2, join=True)
    assert bool(np.load(out)[0])


def _out_profile_worker(rank, world, port, out):
    """The out-profile in P blocks over gloo (vft_nj_options.out_profile_parts; SURVEY.md 8e: per-GPU partial sums, all-gather, fixed-order
    sum) with the C++ layout function itself (vft_nj_out_profile_block): every rank sums the blocks it owns - numpy restatement of
    k_outprofile_partial's running sums, one numeric_t rounding per step -, sends its share, puts the blocks back in block order and adds
    them up as k_outprofile_finish does."""
    import ctypes as C
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from veryfasttree_amd.backend import load_host_library
    lib = load_host_library()
    rng = np.random.default_rng(5)
    n, width = 1037, 40
    addends = rng.random((n, width)).astype(np.float32)   # what each active node adds per (column, weight / frequency) cell

    def block_sum(i0, i1):
        acc = np.zeros(width, np.float32)
        for t in range(i0, i1):
            acc = (acc + addends[t]).astype(np.float32)
        return acc

    results = {}
    for P in (2, 3, 4, 7):
        o = np.zeros(5, np.int64)
        lay = []
        for b in range(P):
            assert lib.vft_nj_out_profile_block(C.c_int32(P), C.c_int32(world), C.c_int32(b), C.c_int64(n), o.ctypes.data_as(C.c_void_p)) == 0
            lay.append(tuple(int(x) for x in o))
        slots = lay[0][2]
        mine = np.zeros((slots, width), np.float32)
        for b, (owner, slot, _, i0, i1) in enumerate(lay):
            if owner == rank:
                mine[slot] = block_sum(i0, i1)
        t_all = torch.zeros(world * slots * width, dtype=torch.float32)
        dist.all_gather_into_tensor(t_all, torch.from_numpy(mine.reshape(-1).copy()))
        got = t_all.numpy().reshape(world, slots, width)
        total = np.zeros(width, np.float32)
        for b, (owner, slot, _, i0, i1) in enumerate(lay):
            total = (total + got[owner, slot]).astype(np.float32)
        # what ONE rank computes for the same partition
        want = np.zeros(width, np.float32)
        covered = 0
        for b, (_, _, _, i0, i1) in enumerate(lay):
            want = (want + block_sum(i0, i1)).astype(np.float32)
            covered += i1 - i0
        assert covered == n and lay[0][3] == 0 and lay[-1][4] == n
        results[P] = bool(np.array_equal(total, want))
    if rank == 0:
        np.save(out, np.array([all(results.values())]))
    dist.barrier()
    dist.destroy_process_group()


def test_out_profile_blocks_over_two_ranks_equal_one_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "outprofile.npy")
    mp.spawn(_out_profile_worker, args=(2, port, out), nprocs=2, join=True)
    assert bool(np.load(out)[0])
