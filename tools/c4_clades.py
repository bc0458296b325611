#!/usr/bin/env python3
"""Config C4's tree against what the reference has left (round 4): (1) every clade the ONE-thread reference formed in the 594 643 joins
it traced (oracle/_ref/c4_joins_t1_r03.txt) must be a split of this backend's tree; (2) the same check on the tree of the reference
at six threads (oracle/_ref/c4_nj_t6.tree) - which fails for 14 of them: at a million sequences the reference's NJ phase depends on
the thread count, so that tree cannot pin the one-thread order.  Needs this backend's tree as gpurun_out/c4tree/ours.tree.gz
(nj_newick on the alignment below, `fastest=False, me_lengths=True`).  Output: profiles/r04_c4_pin.txt."""
import gzip, re, sys, numpy as np
sys.path.insert(0, '/root/repo')
from veryfasttree_amd import synth
M = (1 << 64) - 1
codes = synth.random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)
# unique sequences in order of first occurrence (the reference's node numbering)
_, first, inv = np.unique(codes, axis=0, return_index=True, return_inverse=True)
order = np.argsort(first)            # unique id (sorted) -> rank by first occurrence
rank = np.empty_like(order); rank[order] = np.arange(len(order))
node_of_seq = rank[inv.ravel()]      # sequence index -> node index
nU = len(first)
print("unique", nU)
rng = np.random.default_rng(7)
leafh = rng.integers(0, 1 << 63, size=1000000, dtype=np.uint64)
nodeh = np.zeros(2 * nU, dtype=np.uint64)
np.add.at(nodeh, node_of_seq, leafh)     # wraps mod 2^64
total = int(leafh.sum(dtype=np.uint64))
def tree_splits(t):
    tok = re.findall(r"[(),;]|[^(),;]+", t)
    stack = []; cur = []; out = set(); last = None
    for x in tok:
        if x == "(":
            stack.append(cur); cur = []
        elif x == ")":
            h = 0
            for c in cur: h = (h + c) & M
            cur = stack.pop(); cur.append(h); out.add(min(h, (total - h) & M))
        elif x == "," : pass
        elif x == ";": break
        else:
            if x.startswith(":") or x[0] == ")" : continue
            name = x.partition(":")[0]
            if name.startswith("s") and name[1:].isdigit():
                cur.append(int(leafh[int(name[1:])]))
    return out
ours = tree_splits(gzip.open('/root/repo/gpurun_out/c4tree/ours.tree.gz','rt').read())
t6 = tree_splits(open('/root/repo/oracle/_ref/c4_nj_t6.tree').read())
print("splits ours", len(ours), "t6", len(t6), "common", len(ours & t6))
# clades of the one-thread reference's traced joins
n = 0; miss_ours = []; miss_t6 = []
for line in open('/root/repo/oracle/_ref/c4_joins_t1_r03.txt'):
    f = line.rstrip("\n").split("\t")
    if len(f) < 11 or not line.endswith("\n"): break
    i, j, new = int(f[1]), int(f[2]), int(f[10])
    nodeh[new] = nodeh[i] + nodeh[j]
    h = int(nodeh[new]); key = min(h, (total - h) & M)
    if key not in ours: miss_ours.append(n)
    if key not in t6: miss_t6.append(n)
    n += 1
print("traced joins of the one-thread reference:", n)
print("clades missing in OUR tree:", len(miss_ours), miss_ours[:10])
print("clades missing in the 6-thread reference tree:", len(miss_t6), miss_t6[:20])
