"""TEST INFRASTRUCTURE: a Python restatement of the neighbour-joining driver (the product's driver is C++:
veryfasttree_amd/host/NJDriver.h, entry point vft_nj_run).

The top-hits bookkeeping of fastNJ (NJ.tcc:2796-3155, 3746-4833) against the backend interface (`HipProfileOps` on the
GPU, the oracle-backed stand-in of tests/oracle_ops.py on the CPU), so that the reference's `Join` lines can be checked
without a GPU and the multi-rank exchange with gloo.  All profile arithmetic happens behind `ops`; what lives here is
control flow and the handful of scalar formulas the reference evaluates on the host (criterion, branch lengths,
diameters), with the reference's float/double mix reproduced through numpy scalars.  Deterministic single-thread
semantics, default options and `-fastest -no2nd` (no second-level top hits, no constraints, no BIONJ weighting).
"""
import math

import numpy as np

F32 = np.float32


class Besthit:
    __slots__ = ("i", "j", "weight", "dist", "criterion")

    def __init__(self, i=-1, j=-1, weight=0.0, dist=1e20, criterion=1e20):
        self.i, self.j, self.weight, self.dist, self.criterion = i, j, weight, dist, criterion

    def copy(self):
        return Besthit(self.i, self.j, self.weight, self.dist, self.criterion)


def sort_by_criterion(hits):
    """psort + CompareHitsByCriterion (Utils.h:126-146, NJ.tcc:7301): ascending criterion, ties by DESCENDING
    original position."""
    order = sorted(range(len(hits)), key=lambda t: (hits[t].criterion, -t))
    return [hits[t] for t in order]


def sort_by_ij(hits):
    """psort + CompareHitsByIJ (NJ.tcc:7309): ascending (i, j), ties by descending original position."""
    order = sorted(range(len(hits)), key=lambda t: (hits[t].i, hits[t].j, -t))
    return [hits[t] for t in order]


class NJDriver:
    def __init__(self, ops, codes, fastest=False, tophits_mult=1.0, tophits_close=-1.0, tophits_refresh=0.8,
                 topvisible_mult=1.5, stale_out_limit=0.01, n_reset_out_profile=200, f_reset_out_profile=0.02,
                 use_tophits_2nd=False, tophits2_mult=1.0, tophits2_safety=3, tophits2_refresh=0.6):
        self.ops = ops
        self.dt = np.dtype(ops.dt).type
        self.n_seqs, self.n_pos = codes.shape
        self.maxnodes = 2 * self.n_seqs
        self.maxnode = self.n_seqs
        self.fastest = fastest
        self.tophits_mult, self.tophits_close, self.tophits_refresh = tophits_mult, tophits_close, tophits_refresh
        self.topvisible_mult, self.stale_out_limit = topvisible_mult, stale_out_limit
        self.n_reset_out_profile, self.f_reset_out_profile = n_reset_out_profile, f_reset_out_profile
        # second-level top hits: -fastest at one thread (main.cpp:339-343, VeryFastTree.cpp:87-91, Options.h:31-35)
        self.use_tophits_2nd, self.tophits2_mult = use_tophits_2nd, tophits2_mult
        self.tophits2_safety, self.tophits2_refresh = tophits2_safety, tophits2_refresh
        n = self.n_seqs
        dt = self.dt
        self.parent = np.full(self.maxnodes, -1, np.int64)
        self.child = {}
        self.diameter = np.zeros(self.maxnodes, dt)
        self.branchlength = np.zeros(self.maxnodes, dt)
        self.selfweight_leaf = (codes != 127).sum(axis=1).astype(dt)
        self.totdiam = 0.0
        self.joins = []
        # NJ constructor (NJ.tcc:210-272)
        ops.upload_leaves(codes)
        ops.set_node_scalars(0, np.zeros(n, dt), self.selfweight_leaf, np.zeros(n, dt))
        ops.set_max_node(n)
        ops.outProfile(np.arange(n))
        ops.set_out_distances(0, np.zeros(n, dt), np.full(n, 10 * n, np.int64))
        ops.setOutDistance(None, n, 0.0)
        self.out_dist = np.zeros(self.maxnodes, dt)
        self.n_out = np.full(self.maxnodes, 10 * n, np.int64)
        self._sync_out(n)

    # ---- mirrors of the device-side lazy out-distance state
    def _sync_out(self, upto=None):
        upto = self.maxnode if upto is None else upto
        od, na = self.ops.get_out_distances(0, upto)
        self.out_dist[:upto] = od
        self.n_out[:upto] = na

    def n_diff_allow(self, n_active):
        return int(n_active * self.stale_out_limit) if self.tophits_mult > 0 else 0

    def set_out_distance(self, node, n_active):
        if self.n_out[node] == n_active:
            return
        self.ops.setOutDistance([node], n_active, self.totdiam)
        od, na = self.ops.get_out_distances(node, 1)
        self.out_dist[node], self.n_out[node] = od[0], na[0]

    def set_criterion(self, n_active, hit):
        """NJ.tcc:1085-1113."""
        if hit.i < 0 or hit.j < 0 or self.parent[hit.i] >= 0 or self.parent[hit.j] >= 0:
            return
        allow = self.n_diff_allow(n_active)
        if self.n_out[hit.i] - n_active > allow:
            self.set_out_distance(hit.i, n_active)
        if self.n_out[hit.j] - n_active > allow:
            self.set_out_distance(hit.j, n_active)
        out_i = float(self.out_dist[hit.i])
        if self.n_out[hit.i] != n_active:
            out_i *= (n_active - 1) / float(self.n_out[hit.i] - 1)
        out_j = float(self.out_dist[hit.j])
        if self.n_out[hit.j] != n_active:
            out_j *= (n_active - 1) / float(self.n_out[hit.j] - 1)
        hit.criterion = self.dt(float(hit.dist) - (out_i + out_j) / float(n_active - 2))

    def set_dist_criterion_batch(self, n_active, hits):
        """setDistCriterion (NJ.tcc:1115-1124) for a list of Besthit, one device call."""
        if not hits:
            return
        i = np.array([h.i for h in hits], np.int64)
        j = np.array([h.j for h in hits], np.int64)
        d, w, c = self.ops.setDistCriterion(i, j, n_active, self.n_diff_allow(n_active), self.totdiam)
        for t, h in enumerate(hits):
            h.dist, h.weight, h.criterion = d[t], w[t], c[t]
        self._sync_out()

    def active_ancestor(self, node):
        if node < 0:
            return node
        while self.parent[node] >= 0:
            node = self.parent[node]
        return int(node)

    # ---- top-hits structures (NJ.h:206-248)
    def init_tophits(self, m):
        self.m = m
        self.q = int(0.5 + self.tophits2_mult * math.sqrt(m))   # NJ.tcc:199-204
        if not self.use_tophits_2nd or self.q >= m:
            self.q = 0
        self.hits = [[] for _ in range(self.maxnodes)]          # list of [j, dist]
        self.hit_source = [-1] * self.maxnodes
        self.age = [0] * self.maxnodes
        self.visible = [[-1, self.dt(1e20)] for _ in range(self.maxnodes)]
        self.topvisible = [-1] * int(0.5 + self.topvisible_mult * m)
        self.topvisible_age = 0

    def hits_to_besthits(self, hits, node):
        return [Besthit(node, int(h[0]), -1.0, h[1], self.dt(1e20)) for h in hits]

    def sort_save_best_hits(self, node, besthits, n_in, n_out, sort=True):
        """NJ.tcc:4535-4578."""
        if sort:
            besthits[:] = sort_by_criterion(besthits)
        saved = []
        j_last = -1
        for b in besthits[:n_in]:
            if len(saved) >= n_out:
                break
            if b.i < 0:
                continue
            if b.j != node and b.j != j_last and b.j >= 0:
                saved.append([b.j, b.dist])
                j_last = b.j
        # the reference's second pass does not test i < 0; entries with i < 0 carry j values that were already
        # filtered by the first pass only through i, so mirror the second pass exactly:
        out = []
        j_last = -1
        for b in besthits[:n_in]:
            if len(out) >= len(saved):
                break
            if b.j != node and b.j != j_last and b.j >= 0:
                out.append([b.j, b.dist])
                j_last = b.j
        self.hits[node] = out

    def transfer_best_hits(self, n_active, node, oldhits, n_old, update_distances):
        """NJ.tcc:4580-4613."""
        new = []
        todo_dist, todo_crit = [], []
        for old in oldhits[:n_old]:
            h = Besthit(node, self.active_ancestor(old.j), old.weight, old.dist, old.criterion)
            if h.j < 0 or h.j == node:
                h.weight, h.dist, h.criterion = 0, self.dt(-1e20), self.dt(1e20)
            elif h.i != old.i or h.j != old.j:
                if update_distances:
                    todo_dist.append(h)
                else:
                    h.dist, h.criterion = self.dt(-1e20), self.dt(1e20)
            else:
                if update_distances:
                    todo_crit.append(h)
                else:
                    h.criterion = self.dt(1e20)
            new.append(h)
        self.set_dist_criterion_batch(n_active, todo_dist)
        for h in todo_crit:
            self.set_criterion(n_active, h)
        return new

    def update_best_hit(self, n_active, hit, update_dist, todo):
        """NJ.tcc:1626-1648; distance recomputation is deferred into `todo`."""
        i, j = self.active_ancestor(hit.i), self.active_ancestor(hit.j)
        if i < 0 or j < 0 or i == j:
            hit.i = hit.j = -1
            hit.weight, hit.dist, hit.criterion = 0, self.dt(1e20), self.dt(1e20)
            return False
        if i != hit.i or j != hit.j:
            hit.i, hit.j = i, j
            if update_dist:
                todo.append(hit)
            else:
                hit.dist, hit.criterion = self.dt(-1e20), self.dt(1e20)
        return True

    def unique_best_hits(self, n_active, combined):
        """NJ.tcc:4786-4833."""
        for h in combined:
            self.update_best_hit(n_active, h, False, None)
        combined = sort_by_ij(combined)
        out = []
        last = None
        for h in combined:
            if h.i < 0 or h.j < 0:
                continue
            if last is not None and last.i == h.i and last.j == h.j:
                continue
            out.append(h)
            last = h
        todo = [h for h in out if h.dist < 0.0]
        done = set(id(h) for h in todo)
        self.set_dist_criterion_batch(n_active, todo)
        for h in out:
            if id(h) not in done:
                self.set_criterion(n_active, h)
        return out

    def get_visible(self, n_active, node):
        """NJ.tcc:546-557."""
        if node < 0 or self.parent[node] >= 0:
            return None
        v = self.visible[node]
        if v[0] < 0 or self.parent[v[0]] >= 0:
            return None
        b = Besthit(node, int(v[0]), -1.0, v[1], self.dt(1e20))
        self.set_criterion(n_active, b)
        return b

    def update_top_visible(self, n_active, i_in, hit):
        """NJ.tcc:4660-4726."""
        tv = self.topvisible
        placed = False
        for t in range(len(tv)):
            node = tv[t]
            if node == i_in:
                placed = True
                break
            if node < 0 or self.parent[node] >= 0:
                tv[t] = i_in
                placed = True
                break
        pos_worst, crit_worst = -1, -1e20
        if not placed:
            for t in range(len(tv)):
                node = tv[t]
                vis = self.get_visible(n_active, node)
                if vis is None:
                    tv[t] = i_in
                    placed = True
                    break
                if vis.i == hit[0] and vis.j == i_in:
                    placed = True
                    break
                if vis.criterion >= crit_worst:
                    pos_worst, crit_worst = t, vis.criterion
        if not placed and pos_worst >= 0:
            b = Besthit(i_in, int(hit[0]), -1.0, hit[1], self.dt(1e20))
            self.set_criterion(n_active, b)
            if b.criterion < crit_worst:
                tv[pos_worst] = i_in

    def update_visible(self, n_active, tophits_node):
        """NJ.tcc:4633-4657."""
        for hit in tophits_node:
            if hit.i < 0:
                continue
            vis = self.get_visible(n_active, hit.j)
            if vis is None or hit.criterion < vis.criterion:
                v = self.visible[hit.j]
                v[0], v[1] = hit.i, hit.dist
                self.update_top_visible(n_active, hit.j, v)

    def reset_top_visible(self, n_active):
        """NJ.tcc:4728-4784."""
        vis = []
        for node in range(self.maxnode):
            if self.parent[node] >= 0:
                continue
            v = self.get_visible(n_active, node)
            if v is not None:
                vis.append(v)
        n_visible = len(vis)
        # The reference sorts a value-initialised array of nActive records of which only the first nVisible are
        # filled (NJ.tcc:4729-4744): the zero records (i = j = 0, criterion 0) take part in the sort and only the
        # first nVisible positions of the sorted array are then considered.  Reproduced literally.
        vis = vis + [Besthit(0, 0, 0.0, self.dt(0), self.dt(0)) for _ in range(n_active - n_visible)]
        vis = sort_by_criterion(vis)[:n_visible]
        in_top = {}
        save = []
        for v in vis:
            if len(save) >= len(self.topvisible):
                break
            if in_top.get(v.i, -1) != v.j:
                save.append(v.i)
                in_top[v.i] = v.j
                in_top[v.j] = v.i
        self.topvisible = save + [-1] * (len(self.topvisible) - len(save))
        self.topvisible_age = 0

    # ---- setAllLeafTopHits (NJ.tcc:3746-4119), threads == 1 branch without 2nd-level lists
    def set_all_leaf_top_hits(self):
        n, m = self.n_seqs, self.m
        close = self.tophits_close
        if close < 0:
            if self.fastest and n >= 50000:
                close = 0.99
            else:
                log_n = math.log(float(n)) / math.log(2.0)
                close = log_n / (log_n + 2.0)
        n_gaps = [int(0.5 + self.n_pos - float(self.selfweight_leaf[i])) for i in range(n)]
        # CompareSeeds (NJ.tcc:7285-7299): fewer gaps, then smaller out-distance, ties by descending position
        seeds = sorted(range(n), key=lambda s: (n_gaps[s], float(self.out_dist[s]), -s))
        visited = [False] * n
        allow = self.n_diff_allow(n)
        for seed in seeds:
            if visited[seed]:
                continue
            visited[seed] = True
            hits, _ = self.ops.setBestHit(seed, n, allow, self.totdiam, 2 * m, want_best=False)
            self._sync_out()
            best_seed = [Besthit(seed, int(h["j"]), h["weight"], h["dist"], h["criterion"]) for h in hits]
            self.sort_save_best_hits(seed, list(best_seed), len(best_seed), m, sort=False)
            neardist = float(best_seed[2 * m - 1].dist) * close
            nearweight = 0.0
            for t in range(2 * m):
                nearweight += float(best_seed[t].weight)
            nearweight = nearweight / (2.0 * m)
            nearweight *= (1.0 - 2.0 * neardist / 3.0)
            nearcover = 1.0 - neardist / 2.0
            for i_close in range(m):
                ch = best_seed[i_close]
                cn = ch.j
                if visited[cn]:
                    continue
                is_close = float(ch.dist) <= neardist and (float(ch.weight) >= nearweight or
                                                            float(ch.weight) >= (self.n_pos - n_gaps[cn]) * nearcover)
                identical = (float(ch.dist) < 1e-6 and abs(float(ch.weight) - (self.n_pos - n_gaps[seed])) < 1e-5
                             and abs(float(ch.weight) - (self.n_pos - n_gaps[cn])) < 1e-5)
                if self.use_tophits_2nd and i_close < self.q and (is_close or identical):
                    n_use = min(self.q * self.tophits2_safety, 2 * m)
                    bc = self.transfer_best_hits(n, cn, best_seed, n_use, True)
                    visited[cn] = True
                    self.sort_save_best_hits(cn, bc, n_use, self.q)
                    self.hit_source[cn] = seed
                elif is_close or identical or (self.fastest and i_close < (self.q + 1) // 2):
                    nb = self.transfer_best_hits(n, cn, best_seed, 2 * m, True)
                    visited[cn] = True
                    self.sort_save_best_hits(cn, nb, 2 * m, m)      # sorts nb in place, like the reference
                    # second level of transfer (NJ.tcc:3993-4012): no closeness test, q is small
                    for i_close2 in range(min(self.q, 2 * m)):
                        cn2 = nb[i_close2].j
                        if cn2 >= 0 and not visited[cn2]:
                            n_use = min(self.q * self.tophits2_safety, 2 * m)
                            bc2 = self.transfer_best_hits(n, cn2, nb, n_use, True)
                            visited[cn2] = True
                            self.sort_save_best_hits(cn2, bc2, n_use, self.q)
                            self.hit_source[cn2] = cn
        for node in range(n):
            self.visible[node] = list(self.hits[node][0])
        # checking phase (NJ.tcc:4052-4119)
        n_check = self.q if self.q > 0 else int(0.5 + 2.0 * math.sqrt(m))
        for node in range(n):
            l_node = self.hits[node]
            for i_hit in range(min(n_check, len(l_node))):
                bh = Besthit(node, int(l_node[i_hit][0]), -1.0, l_node[i_hit][1], self.dt(1e20))
                self.set_criterion(n, bh)
                l_target = self.hits[bh.j]
                chk = Besthit(bh.j, int(l_target[n_check - 1][0]), -1.0, l_target[n_check - 1][1], self.dt(1e20))
                self.set_criterion(n, chk)
                if chk.criterion < bh.criterion:
                    continue
                if any(h[0] == node for h in l_target):
                    continue
                i_worst, d_worst = -1, -1e20
                for t, h in enumerate(l_target):
                    b2 = Besthit(bh.j, int(h[0]), -1.0, h[1], self.dt(1e20))
                    self.set_criterion(n, b2)
                    if b2.criterion > d_worst:
                        i_worst, d_worst = t, b2.criterion
                if d_worst > bh.criterion:
                    l_target[i_worst][0], l_target[i_worst][1] = node, bh.dist
                    v = self.get_visible(n, bh.j)
                    if bh.criterion < v.criterion:
                        self.visible[bh.j] = list(l_target[i_worst])

    # ---- topHitNJSearch (NJ.tcc:4137-4262)
    def get_best_from_top_hits(self, node, n_active):
        if not self.fastest:
            self.set_out_distance(node, n_active)
        best = Besthit(-1, -1, 0, self.dt(1e20), self.dt(1e20))
        cand = []
        todo = []
        for h in self.hits[node]:
            bh = Besthit(node, int(h[0]), -1.0, h[1], self.dt(1e20))
            if self.update_best_hit(n_active, bh, True, todo):
                cand.append(bh)
        self.set_dist_criterion_batch(n_active, todo)
        for bh in cand:
            self.set_criterion(n_active, bh)
            if bh.criterion < best.criterion:
                best = bh
        return best

    def top_hit_nj_search(self, n_active):
        while True:
            n_cand, best_node, best_crit = 0, -1, 1e20
            for node in self.topvisible:
                v = self.get_visible(n_active, node)
                if v is not None:
                    n_cand += 1
                    if best_node < 0 or v.criterion < best_crit:
                        best_node, best_crit = node, v.criterion
            self.topvisible_age += 1
            if 2 * self.topvisible_age > self.m or (3 * n_cand < len(self.topvisible) and 3 * n_cand < n_active):
                if self.topvisible_age <= 2:
                    for node in range(self.maxnode):
                        if self.parent[node] >= 0:
                            continue
                        v = self.visible[node]
                        newj = self.active_ancestor(v[0])
                        if newj >= 0 and newj != v[0]:
                            if newj == node:
                                newj = 0
                                while self.parent[newj] >= 0 or newj == node:
                                    newj += 1
                            bh = Besthit(node, newj)
                            self.set_dist_criterion_batch(n_active, [bh])
                            v[0], v[1] = newj, bh.dist
                self.reset_top_visible(n_active)
                continue
            break
        join = self.get_visible(n_active, best_node)
        if self.fastest:
            return join
        join2 = join.copy()
        while True:
            changed = False
            best = self.get_best_from_top_hits(join2.i, n_active)
            if best.j != join2.j and best.criterion < join2.criterion:
                changed = True
                join2 = best
            best = self.get_best_from_top_hits(join2.j, n_active)
            if best.j != join2.i and best.criterion < join2.criterion:
                changed = True
                join2 = best
            join = join2
            if not changed:
                break
        return join

    # ---- topHitJoin (NJ.tcc:4306-4533)
    def top_hit_join(self, newnode, n_active):
        c0, c1 = self.child[newnode]
        combined = self.hits_to_besthits(self.hits[c0], c0) + self.hits_to_besthits(self.hits[c1], c1)
        unique = self.unique_best_hits(n_active, combined)
        n_unique = len(unique)
        self.hits[c0], self.hits[c1] = [], []
        self.age[newnode] = (self.age[c0] + self.age[c1] + 1) // 2 + 1
        age_limit = max(1, int(0.5 + math.log(float(self.m)) / math.log(2.0)))
        second = self.hit_source[c0] >= 0 and self.hit_source[c1] >= 0
        need = int(0.5 + self.tophits2_refresh * self.q) if second else int(0.5 + self.m * self.tophits_refresh)
        use_unique = n_unique == n_active - 1 or (self.age[newnode] <= age_limit and n_unique >= need)
        if not use_unique and second and self.age[newnode] <= age_limit:
            # switch from 2nd-level to 1st-level top hits (NJ.tcc:4364-4410)
            source = self.active_ancestor(self.hit_source[c0])
            if source == newnode:
                source = self.active_ancestor(self.hit_source[c1])
            if source != newnode and source >= 0 and self.hit_source[source] < 0:
                merge = [h.copy() for h in unique]
                first = Besthit(newnode, source)
                self.set_dist_criterion_batch(n_active, [first])
                merge.append(first)
                more = self.hits_to_besthits(self.hits[source], newnode)
                self.set_dist_criterion_batch(n_active, more)
                merge += more
                unique = self.unique_best_hits(n_active, merge)
                # the reference tests the OLD nUnique here (NJ.tcc:4402: nUnique is not refreshed after the merge)
                use_unique = n_unique >= int(0.5 + self.m * self.tophits_refresh)
                second = False
        if use_unique:
            if second:
                self.hit_source[newnode] = self.hit_source[c0]
            n_save = min(n_unique, self.q if second else self.m)
            self.sort_save_best_hits(newnode, unique, n_unique, n_save)
            self.visible[newnode] = list(self.hits[newnode][0])
            self.update_top_visible(n_active, newnode, self.visible[newnode])
            self.update_visible(n_active, unique[:n_save])
            return
        # refresh
        self.age[newnode] = 0
        if self.fastest:
            for node in range(self.maxnode):
                if self.parent[node] < 0:
                    self.set_criterion(n_active, Besthit(node, node, 0, self.dt(0), 0))
        else:
            self.ops.setOutDistance(None, n_active, self.totdiam)
            self._sync_out()
        hits, _ = self.ops.setBestHit(newnode, n_active, self.n_diff_allow(n_active), self.totdiam, 2 * self.m,
                                      want_best=False)
        self._sync_out()
        allhits = [Besthit(newnode if h["j"] >= 0 else -1, int(h["j"]), h["weight"], h["dist"], h["criterion"])
                   for h in hits]
        self.sort_save_best_hits(newnode, list(allhits), len(allhits), self.m, sort=False)
        for i_hit in range(min(self.m, len(allhits))):
            if allhits[i_hit].i < 0:
                continue
            node = allhits[i_hit].j
            if self.parent[node] >= 0:
                continue
            old = self.hits[node]
            n_old = len(old)
            self.age[node] = 0
            both = self.hits_to_besthits(old, node)
            for b in both:
                self.set_criterion(n_active, b)
            if n_active <= 2 * self.m:
                self.hit_source[node] = -1     # abandon the 2nd-level heuristic
            n_new = self.q if self.hit_source[node] >= 0 else self.m
            both += self.transfer_best_hits(n_active, node, allhits, 2 * n_new, False)
            unique2 = self.unique_best_hits(n_active, both[:n_old + 2 * n_new])
            self.sort_save_best_hits(node, unique2, len(unique2), n_new)
            self.visible[node] = list(self.hits[node][0])
        self.reset_top_visible(n_active)

    # ---- fastNJ (NJ.tcc:2796-3155)
    def run(self, max_joins=None):
        n = self.n_seqs
        dt = self.dt
        m = int(0.5 + self.tophits_mult * math.sqrt(n)) if self.tophits_mult > 0 else 0
        if m < 4 or 2 * m >= n:
            raise NotImplementedError("top-hits are off for this size (m=%d): the visible-set path is not ported" % m)
        self.init_tophits(m)
        self.set_all_leaf_top_hits()
        self.reset_top_visible(n)
        n_active_reset = n
        n_active = n
        while n_active > 3:
            if max_joins is not None and len(self.joins) >= max_joins:
                break
            join = self.top_hit_nj_search(n_active)
            self.set_out_distance(join.i, n_active)
            self.set_out_distance(join.j, n_active)
            self.set_dist_criterion_batch(n_active, [join])
            newnode = self.maxnode
            self.maxnode += 1
            i, j = join.i, join.j
            self.parent[i] = self.parent[j] = newnode
            self.child[newnode] = (min(i, j), max(i, j))
            self.joins.append((min(i, j), max(i, j), newnode, float(join.criterion)))
            dist_ij = float(join.dist)
            delta = float(dt(self.out_dist[i] - self.out_dist[j])) / float(n_active - 2)
            self.branchlength[i] = dt((dist_ij + delta) / 2)
            self.branchlength[j] = dt((dist_ij - delta) / 2)
            bw = 0.5
            self.diameter[newnode] = dt(bw * float(dt(self.branchlength[i] + self.diameter[i])) +
                                        (1 - bw) * float(dt(self.branchlength[j] + self.diameter[j])))
            ops = self.ops
            ops.set_max_node(self.maxnode)
            ops.averageProfile([newnode], [i], [j])
            ops.set_parents(i, [newnode])
            ops.set_parents(j, [newnode])
            ops.set_node_scalars(newnode, diameter=np.array([self.diameter[newnode]], dt))
            changed = n_active_reset - (n_active - 1)
            if changed >= self.n_reset_out_profile and changed >= self.f_reset_out_profile * n_active_reset:
                active = np.nonzero(self.parent[:self.maxnode] < 0)[0]
                tot = 0.0
                for v in active:
                    tot += float(self.diameter[v])
                self.totdiam = tot
                ops.outProfile(active)
                n_active_reset = n_active - 1
            else:
                ops.updateOutProfile(i, j, newnode, n_active)
                self.totdiam += float(dt(dt(self.diameter[newnode] - self.diameter[i]) - self.diameter[j]))
            self.out_dist[newnode] = 0
            self.n_out[newnode] = 10 * n
            ops.set_out_distances(newnode, np.zeros(1, dt), np.array([10 * n]))
            self.top_hit_join(newnode, n_active - 1)
            n_active -= 1
        return self.joins
