"""TEST INFRASTRUCTURE - CPU restatement of the reference's ML branch-length pass.

Only tests/ may import this.  The per-pair arithmetic (pairLogLk, posteriorProfile) is oracle/vft_oracle.c through
tests/oracle.py; this file restates the control flow around it in plain Python (small cases only):

  min_branch_length            onedimenmin + brent       NeighbourJoining.tcc:7025-7178
  optimize_all_branch_lengths  optimizeAllBranchLengths  NeighbourJoining.tcc:5006-5113
                               (traversePostorder :3343-3380, getUpProfile useML :3382-3434, setupABCD :1942-1975,
                                recomputeProfile useML :3436-3473, MLPairOptimize :1790-1803)

Pinned by tests/test_oracle_golden.py against the branch lengths and tree likelihoods the reference produces on the
white-box fixtures (<model>.opt1/.opt2 in tests/golden/wb_*.npz, written by oracle/whitebox.cpp).
"""
import numpy as np

MAX_BRANCH_LENGTH = 6.0  # xmax of MLPairOptimize, NJ.tcc:1795


def min_branch_length(f, xmin, xguess, xmax, ftol, atol):
    """onedimenmin (NJ.tcc:7025-7080) followed by Brent's method (NJ.tcc:7099-7178).  Returns the minimiser."""
    if xguess == xmin:
        ax, bx, cx = xmin, 2.0 * xguess, 10.0 * xguess
    elif xguess <= 2.0 * xmin:
        ax, bx, cx = xmin, xguess, 5.0 * xguess
    else:
        ax, bx, cx = 0.5 * xguess, xguess, 2.0 * xguess
    if cx > xmax:
        cx = xmax
    if bx >= cx:
        bx = 0.5 * (ax + cx)
    fa, fb, fc = f(ax), f(bx), f(cx)
    while fa < fb and ax > xmin:
        ax = (ax + xmin) / 2.0
        if ax < 2.0 * xmin:
            ax = xmin
        fa = f(ax)
    while fc < fb and cx < xmax:
        cx = (cx + xmax) / 2.0
        if cx > xmax * 0.95:
            cx = xmax
        fc = f(cx)
    # Brent
    cgold, zeps = 0.3819660, 1.0e-10
    a, b = min(ax, cx), max(ax, cx)
    x, fx = bx, fb
    if fa < fc:
        w, fw, v, fv = ax, fa, cx, fc
    else:
        w, fw, v, fv = cx, fc, ax, fa
    d = e = 0.0
    for _ in range(100):
        xm = 0.5 * (a + b)
        tol1 = ftol * abs(x)
        tol2 = 2.0 * (tol1 + zeps)
        if abs(x - xm) <= (tol2 - 0.5 * (b - a)) or abs(a - b) < atol:
            return x
        if abs(e) > tol1:
            r = (x - w) * (fx - fv)
            q = (x - v) * (fx - fw)
            p = (x - v) * q - (x - w) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = abs(q)
            etemp = e
            e = d
            if abs(p) >= abs(0.5 * q * etemp) or p <= q * (a - x) or p >= q * (b - x):
                e = a - x if x >= xm else b - x
                d = cgold * e
            else:
                d = p / q
                u = x + d
                if u - a < tol2 or b - u < tol2:
                    d = abs(tol1) if xm - x >= 0.0 else -abs(tol1)
        else:
            e = a - x if x >= xm else b - x
            d = cgold * e
        if abs(d) >= tol1:
            u = x + d
        else:
            u = x + (abs(tol1) if d >= 0.0 else -abs(tol1))
        fu = f(u)
        if fu <= fx:
            if u >= x:
                a = x
            else:
                b = x
            v, w, x = w, x, u
            fv, fw, fx = fw, fx, fu
        else:
            if u < x:
                a = u
            else:
                b = u
            if fu <= fw or w == x:
                v, w, fv, fw = w, u, fw, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    return x


def postorder(child, n_child, root):
    """traversePostorder (NJ.tcc:3343-3380) without topology changes: children in stored order, then the node."""
    out, stack = [], [(root, 0)]
    while stack:
        node, k = stack.pop()
        if k < n_child[node]:
            stack.append((node, k + 1))
            stack.append((int(child[node][k]), 0))
        else:
            out.append(node)
    return out


def optimize_all_branch_lengths(orc, profs, child, n_child, parent, root, bl, rates, ratecat, tm, min_len, min_rel,
                                ftol, atol):
    """One call of optimizeAllBranchLengths (NJ.tcc:5065-5113, one thread).  profs (list indexed by node) and bl
    (numeric_t array) are updated in place.  Returns the number of likelihood evaluations."""
    up = {}
    evals = [0]

    def posterior(p1, p2, l1, l2):
        return orc.posterior_profile(p1, p2, float(l1), float(l2), rates, ratecat, tm, min_len, min_rel)

    def cd_of(node):
        """C and D of setupABCD (NJ.tcc:1942-1975): profiles and the nodes whose branch lengths go with them."""
        p = int(parent[node])
        if p == root:
            sibs = [int(c) for c in child[root][:3] if int(c) != node]
            return (profs[sibs[0]], sibs[0]), (profs[sibs[1]], sibs[1])
        sib = [int(c) for c in child[p][:2] if int(c) != node][0]
        return (profs[sib], sib), (up_profile(p), p)

    def up_profile(node):
        """getUpProfile(useML = true), NJ.tcc:3382-3434: cached; missing ancestors are filled from the root down."""
        if node in up:
            return up[node]
        path = [node]
        while int(parent[path[-1]]) != root:
            path.append(int(parent[path[-1]]))
        for x in reversed(path):
            if x not in up:
                (pc, nc), (pd, nd) = cd_of(x)
                up[x] = posterior(pc, pd, bl[nc], bl[nd])
        return up[node]

    for node in postorder(child, n_child, root):
        if n_child[node] == 0:
            continue
        kids = [int(c) for c in child[node][:n_child[node]]]
        if n_child[node] == 3:
            nodes3 = kids
            profiles3 = [profs[k] for k in kids]
        else:
            nodes3 = kids + [node]
            profiles3 = [profs[kids[0]], profs[kids[1]], up_profile(node)]
        for _ in range(2):
            for i in range(3):
                b1, b2 = (i + 1) % 3, (i + 2) % 3
                pb = posterior(profiles3[b1], profiles3[b2], bl[nodes3[b1]], bl[nodes3[b2]])
                length = float(bl[nodes3[i]])
                if length < min_len:
                    length = min_len

                def neg_loglk(x, pa=profiles3[i], pb=pb):
                    evals[0] += 1
                    return -orc.pair_loglk(pa, pb, x, rates, ratecat, tm, min_rel)

                bl[nodes3[i]] = min_branch_length(neg_loglk, min_len, length, MAX_BRANCH_LENGTH, ftol, atol)
        if node != root:
            profs[node] = posterior(profs[kids[0]], profs[kids[1]], bl[kids[0]], bl[kids[1]])   # recomputeProfile
            up.pop(node, None)
    return evals[0]


def tree_loglk(orc, profs, child, root, bl, rates, ratecat, tm, min_len, min_rel, leaf_codes, nocode=127):
    """treeLogLk (NJ.tcc:5114-5259) without site likelihoods."""
    n_seqs = leaf_codes.shape[0]
    total = 0.0
    for v in list(range(n_seqs, root)) + [root]:
        a, b = int(child[v][0]), int(child[v][1])
        length = float(bl.dtype.type(bl[a]) + bl.dtype.type(bl[b]))
        total += orc.pair_loglk(profs[a], profs[b], length, rates, ratecat, tm, min_rel)
    r0, r1, r2 = [int(c) for c in child[root][:3]]
    ab = orc.posterior_profile(profs[r0], profs[r1], float(bl[r0]), float(bl[r1]), rates, ratecat, tm, min_len, min_rel)
    total += orc.pair_loglk(ab, profs[r2], float(bl[r2]), rates, ratecat, tm, min_rel)
    if tm is None:
        gaps = int((leaf_codes == nocode).sum())
        total += (gaps - leaf_codes.shape[1]) * np.log(4.0)
    return total


# ---------------------------------------------------------------------------------------------------------------------
# Quartet likelihoods (the evaluation behind testSplitsML and an ML NNI)
#   quartet_loglk      MLQuartetLogLk     NeighbourJoining.tcc:5412-5427
#   quartet_optimize   MLQuartetOptimize  NeighbourJoining.tcc:1650-1788 (no star test: testSplitsML passes nullptr)
#   split_test         the per-split body of traverseTestSplitsML, NeighbourJoining.tcc:6885-6925
# Pinned by tests/test_oracle_golden.py on the <model>.quartet<k>.* entries of the white-box fixtures.
CLOSE_LOGLK_LIMIT = 5.0  # Constants::closeLogLkLimit


def quartet_loglk(orc, pa, pb, pc, pd, lens, rates, ratecat, tm, min_len, min_rel, site=None):
    post = lambda x, y, l1, l2: orc.posterior_profile(x, y, float(l1), float(l2), rates, ratecat, tm, min_len, min_rel)
    pll = lambda x, y, l: orc.pair_loglk(x, y, float(l), rates, ratecat, tm, min_rel, site)
    ab, cd = post(pa, pb, lens[0], lens[1]), post(pc, pd, lens[2], lens[3])
    return pll(pa, pb, lens[0] + lens[1]) + pll(pc, pd, lens[2] + lens[3]) + pll(ab, cd, lens[4])


def quartet_optimize(orc, pa, pb, pc, pd, lens, rates, ratecat, tm, min_len, min_rel, ftol, atol, site=None):
    """lens = [A, B, C, D, I] is updated in place; returns the quartet log-likelihood."""
    post = lambda x, y, l1, l2: orc.posterior_profile(x, y, float(l1), float(l2), rates, ratecat, tm, min_len, min_rel)
    for j in range(5):
        if lens[j] < min_len:
            lens[j] = min_len
    last = {}

    def search(p1, p2, slot):
        def f(x):
            last["v"] = -orc.pair_loglk(p1, p2, x, rates, ratecat, tm, min_rel)
            last["x"] = x
            return last["v"]
        vals = {}

        def g(x):
            v = f(x)
            vals[x] = v
            return v
        lens[slot] = min_branch_length(g, min_len, lens[slot], MAX_BRANCH_LENGTH, ftol, atol)
        return vals[lens[slot]]     # brent returns f at the point it returns

    ab, cd = post(pa, pb, lens[0], lens[1]), post(pc, pd, lens[2], lens[3])
    search(ab, cd, 4)
    search(pa, post(pb, cd, lens[1], lens[4]), 0)
    search(pb, post(pa, cd, lens[0], lens[4]), 1)
    ab = post(pa, pb, lens[0], lens[1])
    search(pc, post(ab, pd, lens[4], lens[3]), 2)
    abc = post(ab, pc, lens[4], lens[2])
    negloglk = search(pd, abc, 3)
    if site is not None:
        site[:] = 1.0
        orc.pair_loglk(abc, pd, float(lens[3]), rates, ratecat, tm, min_rel, site)
    return (-negloglk + orc.pair_loglk(ab, pc, float(lens[4] + lens[2]), rates, ratecat, tm, min_rel, site)
            + orc.pair_loglk(pa, pb, float(lens[0] + lens[1]), rates, ratecat, tm, min_rel, site))


def split_test(orc, pa, pb, pc, pd, lens, rates, ratecat, tm, min_len, min_rel, ftol, atol):
    """Returns (loglk[3], lenAC, lenAD) as testSplitsML computes them for one split."""
    l_ab = [lens[0], lens[1], lens[2], lens[3], lens[4]]
    l_ac = [lens[0], lens[2], lens[1], lens[3], lens[4]]
    l_ad = [lens[0], lens[3], lens[2], lens[1], lens[4]]
    args = (rates, ratecat, tm, min_len, min_rel)
    loglk = [quartet_loglk(orc, pa, pb, pc, pd, l_ab, *args),
             quartet_optimize(orc, pa, pc, pb, pd, l_ac, *args, ftol, atol),
             quartet_optimize(orc, pa, pd, pc, pb, l_ad, *args, ftol, atol)]
    if loglk[1] > loglk[2]:
        if loglk[1] > loglk[0] - CLOSE_LOGLK_LIMIT:
            loglk[1] = quartet_optimize(orc, pa, pc, pb, pd, l_ac, *args, ftol, atol)
    elif loglk[2] > loglk[0] - CLOSE_LOGLK_LIMIT:
        loglk[2] = quartet_optimize(orc, pa, pd, pc, pb, l_ad, *args, ftol, atol)
    return loglk, l_ac, l_ad
