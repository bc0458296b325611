#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the C++ NJ driver on WORLD_SIZE ranks (torch.distributed; all ranks may share GPU 0 with gloo:
VFT_SAME_DEVICE=1) - every rank prints `rank crc32(joins) n_joins allgathers`.  run_nj_ranks.py N L [fastest] [second] [parts=P]
(parts=P: full out-profile recomputations in P blocks split over the ranks, vft_nj_options.out_profile_parts)"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    n, L = int(sys.argv[1]), int(sys.argv[2])
    fastest = "fastest" in sys.argv[3:]
    second = "second" in sys.argv[3:]
    parts = next((int(a[6:]) for a in sys.argv[3:] if a.startswith("parts=")), 0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("VFT_SAME_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    from veryfasttree_amd import HipProfileOps, synth
    from veryfasttree_amd.backend import TorchComm, nj_run
    comm = None
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get("VFT_BACKEND", "nccl"))
        comm = TorchComm(dist, local)
    codes = synth.random_descent_codes(n, L, 4, 0.04, 0.02, seed=17)
    _, first = np.unique(codes, axis=0, return_index=True)
    codes = codes[np.sort(first)]
    ops = HipProfileOps(codes.shape[0], L, 4, np.float32, device=local)
    joins, crit = nj_run(ops, codes, fastest=fastest, second_level=second, comm=comm, out_profile_parts=parts)
    # (one write per rank: the ranks share a pipe)
    sys.stdout.write("rank %d crc %d joins %d allgathers %d\n" % (comm.rank if comm else 0, zlib.crc32(joins.tobytes()), len(joins),
                                                                  comm.calls if comm else 0))
    sys.stdout.flush()
    ops.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
