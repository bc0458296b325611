"""TEST INFRASTRUCTURE — the backend interface of veryfasttree_amd.backend.HipProfileOps implemented with the CPU
oracle, so that the host NJ driver can be exercised (and pinned against the reference's Join lines) without a GPU.
Never imported by the product."""
import numpy as np

from oracle import Oracle, tolerances
from veryfasttree_amd.backend import HIT_F32, HIT_F64

NOCODE = 127


class OracleOps:
    def __init__(self, n_seqs, n_pos, n_codes=4, dtype=np.float32, max_nodes=None):
        self.dt = np.dtype(dtype)
        self.orc = Oracle(dtype)
        self.n_seqs, self.n_pos, self.n_codes = n_seqs, n_pos, n_codes
        self.max_nodes = max_nodes or 2 * n_seqs
        self.maxnode = n_seqs
        self.profs = [None] * self.max_nodes
        self.parent = np.full(self.max_nodes, -1, np.int64)
        self.diameter = np.zeros(self.max_nodes, self.dt)
        self.selfweight = np.zeros(self.max_nodes, self.dt)
        self.selfdist = np.zeros(self.max_nodes, self.dt)
        self.out_dist = np.zeros(self.max_nodes, self.dt)
        self.n_out = np.zeros(self.max_nodes, np.int64)
        self.outp = None
        self.tol = tolerances(dtype)[2]
        self.hit_dtype = HIT_F32 if self.dt == np.float32 else HIT_F64

    def upload_leaves(self, codes):
        for i in range(self.n_seqs):
            self.profs[i] = self.orc.leaf_profile(codes[i], self.n_codes)

    def set_max_node(self, maxnode):
        self.maxnode = maxnode

    def set_parents(self, first, parent):
        parent = np.asarray(parent, np.int64)
        self.parent[first:first + len(parent)] = parent

    def set_node_scalars(self, first, diameter=None, selfweight=None, selfdist=None):
        for arr, src in ((self.diameter, diameter), (self.selfweight, selfweight), (self.selfdist, selfdist)):
            if src is not None:
                src = np.asarray(src, self.dt)
                arr[first:first + len(src)] = src

    def set_out_distances(self, first, out_dist, n_out_active):
        out_dist = np.asarray(out_dist, self.dt)
        self.out_dist[first:first + len(out_dist)] = out_dist
        self.n_out[first:first + len(out_dist)] = np.asarray(n_out_active, np.int64)

    def get_out_distances(self, first, count):
        return self.out_dist[first:first + count].copy(), self.n_out[first:first + count].copy()

    def _pack(self, ids):
        W = np.stack([self.profs[v][0] for v in ids])
        C = np.stack([self.profs[v][1] for v in ids])
        F = np.stack([self.profs[v][2] for v in ids])
        return W, C, F

    def outProfile(self, active_ids):
        W, C, F = self._pack(list(active_ids))
        self.outp, _ = self.orc.out_profile(W, C, F, None, self.tol)

    def updateOutProfile(self, old1, old2, new, n_active_old):
        self.outp, _ = self.orc.update_out_profile(self.outp, None, self.profs[old1], self.profs[old2],
                                                   self.profs[new], n_active_old, None, self.tol)

    def averageProfile(self, out, a, b, bionj_weight=None):
        for k in range(len(out)):
            p = self.orc.average_profile(self.profs[int(a[k])], self.profs[int(b[k])], -1.0, None, self.tol)
            self.profs[int(out[k])] = p
            d, w = self.orc.profiledist(p, p)
            self.selfdist[int(out[k])], self.selfweight[int(out[k])] = d, w

    def _refresh(self, v, n_active, totdiam):
        if self.n_out[v] == n_active:
            return
        d, w = self.orc.profiledist(self.profs[v], self.outp)
        self.out_dist[v] = self.orc.out_distance(d, w, n_active, self.selfweight[v], self.selfdist[v],
                                                  self.diameter[v], totdiam)
        self.n_out[v] = n_active

    def setOutDistance(self, ids, n_active, totdiam):
        if ids is None:
            ids = [v for v in range(self.maxnode) if self.parent[v] < 0]
        for v in ids:
            self._refresh(int(v), n_active, totdiam)

    def _dist_crit(self, i, j, n_active, n_diff_allow, totdiam):
        if i < self.n_seqs and j < self.n_seqs:
            d, w = self.orc.seqdist(self.profs[i][1], self.profs[j][1], self.n_codes)
        else:
            d, w = self.orc.profiledist(self.profs[i], self.profs[j])
            d = self.dt.type(d - self.dt.type(self.diameter[i] + self.diameter[j]))
        for v in (i, j):
            if self.n_out[v] - n_active > n_diff_allow:
                self._refresh(v, n_active, totdiam)
        c = self.orc.criterion(d, self.out_dist[i], self.n_out[i], self.out_dist[j], self.n_out[j], n_active)
        return d, w, c

    def setDistCriterion(self, i, j, n_active, n_diff_allow, totdiam):
        n = len(i)
        d, w, c = (np.zeros(n, self.dt) for _ in range(3))
        for t in range(n):
            d[t], w[t], c[t] = self._dist_crit(int(i[t]), int(j[t]), n_active, n_diff_allow, totdiam)
        return d, w, c

    def setBestHit(self, query, n_active, n_diff_allow, totdiam, k, want_best=True, d_hits=None, want_hits=True):
        n = self.maxnode
        crit = np.full(n, 1e20, self.dt)
        dist = np.full(n, 1e20, self.dt)
        weight = np.zeros(n, self.dt)
        for j in range(n):
            if self.parent[j] >= 0:
                continue
            dist[j], weight[j], crit[j] = self._dist_crit(query, j, n_active, n_diff_allow, totdiam)
        order = self.orc.sort_hits(crit)
        order = order[crit[order] < self.dt.type(1e20)][:k]
        hits = np.zeros(k, self.hit_dtype)
        hits["j"] = -1
        hits["dist"] = 1e20
        hits["criterion"] = 1e20
        m = len(order)
        hits["j"][:m] = order
        hits["dist"][:m] = dist[order]
        hits["weight"][:m] = weight[order]
        hits["criterion"][:m] = crit[order]
        best = -1
        bc = self.dt.type(1e20)
        for j in range(n):
            if j != query and crit[j] < bc:
                best, bc = j, crit[j]
        return hits, best
