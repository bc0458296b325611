"""The whole `bench.py --gpus 8` flow on ONE device (VFT_BENCH_SAME_DEVICE=1, gloo): eight ranks started by torch.distributed.run as
the driver starts them, the target range of every sweep sharded at tile boundaries by cost, one all-gather of the ranks' top-k blocks
per step, the batched merge - the merged lists must be the single-rank lists (hit CRC), whatever the shard layout.  RCCL itself cannot
run with more than one rank on a one-GPU box; everything around the collective does."""
import json
import os
import subprocess
import sys

import pytest

from conftest import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, launcher=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=env, check=True, stdout=subprocess.PIPE, timeout=1500).stdout.decode()
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("world", [8, 3])
def test_ranks_on_one_device_give_the_single_rank_lists(world):
    common = ["--n-seqs", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-dense", "--no-e2e"]
    one = _bench(common)
    launcher = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(world)]
    many = _bench(["--gpus", str(world)] + common, {"VFT_BENCH_SAME_DEVICE": "1", "VFT_BENCH_BACKEND": "gloo"}, launcher)
    assert many["n_gpus"] == world and many["world"] == world
    assert many["hits_crc"] == one["hits_crc"], "the sharded + merged lists differ from the unsharded ones"
    sh = many["shards"]
    assert len(sh) == world and sh[0][0] == 0
    maxnode = one["config"]["n_seqs"] + one["config"]["internal_profiles"]
    assert sh[-1][1] == maxnode
    for (lo, hi), (lo2, _) in zip(sh, sh[1:] + [[maxnode, maxnode]]):
        assert lo % 64 == 0 and lo <= hi and hi == lo2, sh   # tile-aligned, contiguous, disjoint
    assert many["allgathers_per_step"] == 1
    k, seeds = one["config"]["top_k"], one["config"]["seeds_per_step"]
    assert many["allgather_bytes_per_step"] == world * seeds * k * 16   # vft_hit_f32 records


def test_out_profile_in_parts_depends_on_the_partition_only():
    """vft_out_profile_partial / vft_out_profile_finish (per-GPU partial sums, all-gather, fixed-order sum): one block = the
    whole active list is k_outprofile_full's arithmetic bit for bit; four blocks finished in order equal the same blocks with the first
    two added up on the host in numeric_t (the finish is a left-to-right numeric_t sum); and the driver's join order with
    out_profile_parts = 4 is the same on one rank and on two (both on this box's one GPU, gloo) - while it may differ from the one-pass
    order's (P = 0), as the reference's threaded outProfile differs from its one-thread one."""
    import re
    import numpy as np
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.workload import TopHitsState
    n, L = 4000, 90
    codes = synth.random_descent_codes(n, L, 4, 0.05, 0.03, seed=31)
    for dt in (np.float32, np.float64):
        ops = HipProfileOps(n, L, 4, dt)
        st = TopHitsState(ops, codes, 1200)
        act = st.active
        ops.debug_option(4, 1)          # vft_out_profile_full through the one-thread-per-column kernel (the reference's loop)
        ops.outProfile(act)
        ops.debug_option(4, 0)
        dl = lambda: ops.out_profile_download()[0]   # (weights, codes, frequencies)
        want = dl()
        ops.out_profile_finish([ops.out_profile_partial(len(act), act)])
        got = dl()
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        per = (len(act) + 3) // 4
        parts = [ops.out_profile_partial(len(act), act[b * per:(b + 1) * per]) for b in range(4)]
        ops.out_profile_finish(parts)
        four = dl()
        ops.out_profile_finish([(parts[0] + parts[1]).astype(dt), parts[2], parts[3]])
        for a, b in zip(four, dl()):
            assert np.array_equal(a, b)
        assert np.allclose(four[0], want[0], rtol=1e-4) and np.allclose(four[2], want[2], atol=1e-4)   # (the same profile up to the rounding of the sums)
        ops.close()
    script = os.path.join(ROOT, "tests", "run_nj_ranks.py")
    args = ["3000", "150", "parts=4"]
    one = subprocess.run([sys.executable, script] + args, check=True, stdout=subprocess.PIPE, timeout=600).stdout.decode()
    env = dict(os.environ, VFT_SAME_DEVICE="1", VFT_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), script] + args, check=True, stdout=subprocess.PIPE, env=env,
                         timeout=900).stdout.decode()
    want = re.search(r"rank 0 crc (\d+) joins (\d+)", one).groups()
    got = re.findall(r"rank (\d) crc (\d+) joins (\d+) allgathers (\d+)", two)
    assert len(got) == 2
    for r, crc, nj, calls in got:
        assert (crc, nj) == want, (r, crc, nj, want)
