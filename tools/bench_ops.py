#!/usr/bin/env python3
"""Per-operator throughput table (profile-ops/s per kernel) on one MI355X.

One profile-op = one call-equivalent of seqDist / profileDist / averageProfile / pairLogLk / posteriorProfile.
Every line is a whole C-ABI call timed with HIP events on the context's stream (inputs resident, ids uploaded inside
the call), algorithmic bytes as in SURVEY 8d.  Usage: bench_ops.py [nt|aa] -> text table on stdout."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.workload import TopHitsState

which = sys.argv[1] if len(sys.argv) > 1 else "nt"
if which == "nt":
    n, L, NC, dt, mu, gap, seed = 1000000, 200, 4, np.float32, 0.02, 0.01, 4
else:
    n, L, NC, dt, mu, gap, seed = 50000, 300, 20, np.float64, 0.08, 0.02, 5
S = np.dtype(dt).itemsize
V = NC * S
codes = synth.random_descent_codes(n, L, NC, mu, gap, seed=seed)
ops = HipProfileOps(n, L, NC, dt)
if which == "aa":
    d = np.load(os.path.join(ROOT, "tests", "golden", "wb_aa_f64.npz"))
    ops.set_distance_matrix(d["dmat.distances"], d["dmat.codefreq"], d["dmat.eigenval"], d["dmat.eigentot"])
    k = "lg.tm."
    ops.set_transition_matrix(d[k + "stat"], d[k + "statinv"], d[k + "eigenval"], d[k + "codefreq"],
                              d[k + "eigeninv"], d[k + "eigeninvT"])
nj = n // 4
st = TopHitsState(ops, codes, nj)
rates = np.exp(-np.log(20.0) + np.arange(20) * 2 * np.log(20.0) / 19)
ops.set_rates(rates, np.random.default_rng(1).integers(0, 20, L))
nvec = ops.profile_nvectors(n, nj)
phi = float(nvec.mean()) / L
print("# %s: %d x %d, %s, %d internal profiles (phi = %.3f), 1 x MI355X" % (which, n, L, np.dtype(dt).name, nj, phi))
print("%-44s %10s %12s %14s %10s" % ("operator (C-ABI call)", "batch", "us / call", "profile-ops/s", "GB/s (alg)"))


def timed(label, batch, alg_bytes, fn, reps=5):
    fn()
    ops.synchronize()
    ops.timer_start()
    for _ in range(reps):
        fn()
    ms = ops.timer_stop_ms() / reps
    print("%-44s %10d %12.1f %14.3e %10.0f" % (label, batch, ms * 1e3, batch / (ms * 1e-3), alg_bytes / (ms * 1e-3) / 1e9))


leaf_act = st.active[st.active < n]
int_act = st.active[st.active >= n]
per_leaf = L + 3 * S + 8
per_int = L * (S + 1) + 3 * S + 8
b_leaf = len(leaf_act) * per_leaf
b_int = len(int_act) * per_int + int(nvec.sum()) * V
ql, qi = int(leaf_act[17]), int(int_act[23])
kw = dict(want_best=False, want_hits=False)
timed("sweep, leaf seed (seqDist + profileDist)", st.n_active, b_leaf + b_int,
      lambda: ops.setBestHit(ql, st.n_active, st.n_diff_allow, st.totdiam, 0, **kw))
timed("sweep, profile seed (profileDist)", st.n_active, b_leaf + b_int,
      lambda: ops.setBestHit(qi, st.n_active, st.n_diff_allow, st.totdiam, 0, **kw))
m2 = 2 * int(0.5 + np.sqrt(n))
timed("sweep + top-%d select, profile seed" % m2, st.n_active, b_leaf + b_int,
      lambda: ops.setBestHit(qi, st.n_active, st.n_diff_allow, st.totdiam, m2, want_best=True))
rng = np.random.default_rng(3)
pi = np.full(m2, qi, np.int64)
pj = rng.choice(st.active, m2, replace=False)
pb = int((pj < n).sum()) * per_leaf + int((pj >= n).sum()) * per_int + int(nvec[pj[pj >= n] - n].sum()) * V
timed("pair list (transferBestHits), 2m pairs", m2, pb,
      lambda: ops.setDistCriterion(pi, pj, st.n_active, st.n_diff_allow, st.totdiam))
# averageProfile: rewrite a block of existing internal nodes from their own children (same result, same cost)
blk = min(16384, nj)
out = n + np.arange(blk, dtype=np.int64)
a, b = 2 * np.arange(blk, dtype=np.int64), 2 * np.arange(blk, dtype=np.int64) + 1
avg_bytes = blk * (2 * L + L * (S + 1)) + int(nvec[:blk].sum()) * V
timed("averageProfile (+ tile commit, self dist)", blk, avg_bytes, lambda: ops.averageProfile(out, a, b), reps=3)
timed("outProfile, all active nodes", st.n_active, b_leaf + b_int, lambda: ops.outProfile(st.active), reps=2)
# ML operators.  ML profiles live in the transition matrix's eigenbasis, so they are built here the way the reference
# does (posterior of the children, NJ.tcc:3516-3539): level 1 from leaf pairs, level 2 from level-1 pairs; the timed
# posterior is level 2 (both children carry vectors), the timed pairLogLk runs between level-1 profiles.
# Row mode (vft_set_profile_rows), as the ML stage of the pipeline runs (host/MLLengths.h): every internal profile a dense row, posteriors
# write rows.  (Until round 6 this table timed the tile-stream path - stash, tile commit - which no ML stage takes.)
ops.set_profile_rows(True)
blk = min(16384, (nj // 3) * 2)
half = blk // 2
lv1 = n + np.arange(blk, dtype=np.int64)
ops.posteriorProfile(lv1, 2 * np.arange(blk, dtype=np.int64), 2 * np.arange(blk, dtype=np.int64) + 1,
                     np.full(blk, 0.05), np.full(blk, 0.07))
# (vector density of the level-1 posteriors from the rows themselves: vft_profile_nvectors reads the tile streams' masks)
samp = [ops.profile_download(int(v)) for v in lv1[:: max(1, blk // 64)][:64]]
phi1 = float(np.mean([((w > 0) & (cc == 127)).mean() for w, cc, f in samp]))
side = L * (S + 1) + phi1 * L * V
model = "JC" if which == "nt" else "LG"
lv2 = n + blk + np.arange(half, dtype=np.int64)
ln = np.full(half, 0.1)
timed("posteriorProfile (%s), phi_in = %.2f" % (model, phi1), half, half * (3 * side + L),
      lambda: ops.posteriorProfile(lv2, lv1[:half], lv1[half:], ln, ln), reps=2)
timed("pairLogLk (%s), phi = %.2f" % (model, phi1), half, half * (2 * side + L + 8),
      lambda: ops.pairLogLk(lv1[:half], lv1[half:], ln), reps=3)
