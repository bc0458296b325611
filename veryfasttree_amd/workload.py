"""Synthetic mid-run NJ states for bench.py and the full-size GPU tests.

A "top-hits state" is what the join loop of fastNJ (NJ.tcc:2857) looks like part-way through: some leaves have
been joined into internal nodes (here: sibling pairs 2k, 2k+1 of the random-descent alignment), the out-profile
and all out-distances are fresh, and a one-vs-all sweep (setBestHit, NJ.tcc:3571) has leaf targets
(seqDist / profile-vs-leaf) as well as internal-profile targets (profileDist).
"""
import numpy as np

NOCODE = 127


class TopHitsState:
    def __init__(self, ops, codes, n_join):
        n, n_pos = codes.shape
        assert 2 * n_join <= n
        self.ops, self.n_seqs, self.n_pos, self.n_join = ops, n, n_pos, n_join
        ops.upload_leaves(codes)
        dt = ops.dt
        a = 2 * np.arange(n_join, dtype=np.int64)
        b = a + 1
        new = n + np.arange(n_join, dtype=np.int64)
        # leaf scalars (NJ.tcc:245-252): selfweight = ungapped columns, selfdist = diameter = 0
        selfweight = (codes != NOCODE).sum(axis=1).astype(dt)
        ops.set_node_scalars(0, np.zeros(n, dt), selfweight, np.zeros(n, dt))
        step = 1 << 16
        for k0 in range(0, n_join, step):
            ops.averageProfile(new[k0:k0 + step], a[k0:k0 + step], b[k0:k0 + step])
        self.maxnode = n + n_join
        ops.set_max_node(self.maxnode)
        parent = np.full(self.maxnode, -1, np.int64)
        parent[a] = new
        parent[b] = new
        ops.set_parents(0, parent)
        self.parent = parent
        self.active = np.nonzero(parent < 0)[0]
        self.n_active = len(self.active)
        # diameters of the new nodes as the join would set them with equal branch lengths (NJ.tcc:3003):
        # half the raw %-different distance of the two children
        ca, cb = codes[a], codes[b]
        both = (ca != NOCODE) & (cb != NOCODE)
        nuse = np.maximum(both.sum(axis=1), 1)
        dist = ((ca != cb) & both).sum(axis=1) / nuse
        diam = (0.25 * dist).astype(dt)
        ops.set_node_scalars(n, diameter=diam)
        self.totdiam = float(diam.astype(np.float64).sum())
        ops.outProfile(self.active)
        ops.set_out_distances(0, np.zeros(self.maxnode, dt), np.full(self.maxnode, 10 * n, np.int64))
        ops.setOutDistance(None, self.n_active, self.totdiam)
        ops.synchronize()
        self.n_diff_allow = int(self.n_active * 0.01)   # Options::staleOutLimit, NJ.tcc:1091

    def algorithmic_bytes_per_sweep(self, lo=0, hi=None):
        """SURVEY.md §8(d), target side + output only, for the targets in [lo, hi).
        seqDist / profile-vs-leaf: nPos*1 + 2S; profileDist: nPos*(S + 1 + phi*V) + 2S; + (S + 8) for the fused
        criterion epilogue (outDistance + nOutDistActive) per target.
        Returns a dict: leaf / internal = algorithmic bytes of the active leaf / internal targets, moved_leaf /
        moved_internal = what the kernels really read and write for them, phi = mean vector density."""
        S = self.ops.dt.itemsize
        V = self.ops.n_codes * S
        hi = self.maxnode if hi is None else hi
        act = self.active[(self.active >= lo) & (self.active < hi)]
        leaves = int((act < self.n_seqs).sum())
        internal = act[act >= self.n_seqs]
        nvec_all = self.ops.profile_nvectors(self.n_seqs, self.n_join)
        nvec = nvec_all[internal - self.n_seqs]
        per_leaf = self.n_pos * 1 + 2 * S + S + 8
        b_leaf = leaves * per_leaf
        b_int = int(len(internal) * (self.n_pos * (S + 1) + 2 * S + S + 8) + int(nvec.sum()) * V)
        phi = float(nvec.mean()) / self.n_pos if len(internal) else 0.0
        # bytes the kernels really move per target (tile streams: only existing vectors and explicit weights are
        # read; codes are padded to 16 columns; 24 bytes of masks + offsets per 64 targets and column):
        # epilogue reads parent(4) outDist(S) nOutActive(4) diameter(S), writes dist/weight/criterion (3S)
        E = 4 + S + 4 + S + 3 * S
        pad_pos = ((self.n_pos + 15) // 16) * 16
        m_leaf = leaves * (pad_pos + E)
        m_int = len(internal) * (pad_pos + (24 * pad_pos) // 64 + E) + int(nvec.sum()) * V
        return dict(leaf=b_leaf, internal=b_int, moved_leaf=m_leaf, moved_internal=m_int, phi=phi,
                    n_leaf=leaves, n_internal=int(len(internal)))


def shard_range(maxnode, rank, world, weights=None):
    """Tile-aligned target id range [lo, hi) of `rank` (what vft_set_shard takes).
    weights (optional, one cost per node id): boundaries are placed so that every rank gets the same total cost
    instead of the same number of ids - an internal profile costs ~5.5x a leaf in the sweep and joined nodes nothing,
    and late in a run the low ids are almost all joined."""
    tiles = (maxnode + 63) // 64
    if weights is None:
        lo = (tiles * rank // world) * 64
        hi = min((tiles * (rank + 1) // world) * 64, maxnode)
        return lo, hi
    w = np.zeros(tiles * 64, np.float64)
    w[:maxnode] = np.asarray(weights, np.float64)[:maxnode]
    cum = np.concatenate([[0.0], np.cumsum(w.reshape(tiles, 64).sum(axis=1))])
    cuts = [int(np.searchsorted(cum, cum[-1] * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, tiles
    for r in range(1, world + 1):          # monotone, tile granularity
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts[rank] * 64, min(cuts[rank + 1] * 64, maxnode)


def sweep_cost_weights(parent, n_seqs):
    """Relative cost of each node id as a sweep target (measured): joined 0.02, active leaf 1,
    active internal profile 5.5."""
    parent = np.asarray(parent)
    w = np.where(parent < 0, 1.0, 0.02)
    w[n_seqs:] = np.where(parent[n_seqs:] < 0, 5.5, 0.02)
    return w


def merge_hits(all_hits, k):
    """Merge per-shard sorted hit lists (numpy structured arrays) with the reference's order:
    ascending criterion, ties by descending node id."""
    h = np.concatenate(all_hits)
    h = h[h["j"] >= 0]
    order = np.lexsort((-h["j"].astype(np.int64), h["criterion"]))
    out = h[order][:k]
    if len(out) < k:
        pad = np.zeros(k - len(out), h.dtype)
        pad["j"] = -1
        pad["dist"] = 1e20
        pad["criterion"] = 1e20
        out = np.concatenate([out, pad])
    return out
