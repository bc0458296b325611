#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: a complete pipeline of a fixture on WORLD_SIZE ranks (torch.distributed; all ranks may share GPU 0 with gloo:
VFT_SAME_DEVICE=1) - the NJ sweeps AND the lanes of the subtree schedule split over the ranks (vft_comm).  Every rank prints
`rank crc32(tree) bytes lane_allgathers lane_bytes loglk <the TreeLogLk lines as hex floats: treeLogLk's pair likelihoods are split over the ranks too> end`.  run_pipeline_ranks.py <fixture name>"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import golden_util as G
    from test_gpu_threads import AA
    name = sys.argv[1]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("VFT_SAME_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    from veryfasttree_amd import HipProfileOps
    from veryfasttree_amd.backend import TorchComm, nj_newick, last_lane_exchange
    comm = None
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get("VFT_BACKEND", "nccl"))
        comm = TorchComm(dist, local)
    d = G.load(name)
    flags = bytes(d["flags"]).decode().split()
    codes_all = d["codes"]
    nt = "-nt" in flags
    dt = np.float64 if "-double-precision" in flags else np.float32
    names = ["s%d" % k for k in range(len(codes_all))]
    make = lambda n, L: HipProfileOps(n, L, 4 if nt else 20, dt, max_nodes=3 * n, device=local)
    kw = dict(dtype=dt, me_lengths=True, threads=int(d["threads"]), comm=comm)
    if not nt:
        kw["aa_model"] = next((AA[f] for f in flags if f in AA), "jtt")
    if "-noml" in flags:
        kw.update(me_nni=True, spr=2)
    elif "-mllen" in flags:
        kw.update(mllen=20)
    else:
        kw.update(me_nni="-nome" not in flags, spr=0 if "-nome" in flags else 2, ml_nni=20, gtr="-gtr" in flags)
    ml = "-noml" not in flags
    tree = nj_newick(make, codes_all, names, return_loglk=ml, **kw)
    loglk = []
    if ml:
        tree, loglk = tree
    calls, nbytes = last_lane_exchange()
    # (one write per rank: the ranks share a pipe, and print() hands text and newline over separately - lines interleaved mid-way)
    sys.stdout.write("rank %d crc %d bytes %d lane_allgathers %d lane_bytes %d loglk %s end\n" % (comm.rank if comm else 0, zlib.crc32(tree.encode()), len(tree), calls, nbytes,
                                                                                                ",".join(float(x).hex() for x in loglk)))
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
