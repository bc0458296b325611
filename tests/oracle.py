"""ctypes front-end to oracle/libvft_oracle.so — TEST INFRASTRUCTURE (checker only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libvft_oracle.so")
NOCODE = 127


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "port"], check=True)


def _load():
    src_time = max(os.path.getmtime(os.path.join(ORACLE_DIR, f))
                   for f in ("vft_oracle.c", "vft_oracle.h", "vft_oracle_impl.h", "vft_oracle_sort.h"))
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < src_time:
        build()
    return C.CDLL(LIB_PATH)


_lib = _load()
_avx2_lib = None


def _avx2():
    """oracle/libvft_oracle_avx2.so: the CPU baseline of bench.py (AVX2 + OpenMP), built by `make -C oracle port`."""
    global _avx2_lib
    if _avx2_lib is None:
        path = os.path.join(ORACLE_DIR, "libvft_oracle_avx2.so")
        src = os.path.join(ORACLE_DIR, "vft_oracle_avx2.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _avx2_lib = C.CDLL(path)
    return _avx2_lib


def avx2_max_threads():
    return int(_avx2().vfto_avx2_max_threads())
P = C.c_void_p
I64 = C.c_int64


def _ptr(a):
    return None if a is None else a.ctypes.data_as(P)


class _DMat(C.Structure):
    _fields_ = [("distances", P), ("codefreq", P), ("eigenval", P), ("eigentot", P)]


class _TMat(C.Structure):
    _fields_ = [("stat", P), ("statinv", P), ("eigenval", P), ("codefreq", P), ("eigeninv", P), ("eigeninvT", P)]


class _State(C.Structure):
    _fields_ = [("nSeqs", I64), ("maxnode", I64), ("nPos", I64), ("nCodes", C.c_int), ("W", P), ("C", P), ("F", P),
                ("parent", P), ("diameter", P), ("selfweight", P), ("selfdist", P), ("totdiam", C.c_double),
                ("out_w", P), ("out_c", P), ("out_f", P), ("out_cd", P), ("dm", P)]


class Oracle:
    """All oracle entry points for one precision (np.float32 or np.float64)."""

    def __init__(self, dtype):
        self.dt = np.dtype(dtype)
        self.suf = "f32" if self.dt == np.float32 else "f64"
        self.creal = C.c_float if self.dt == np.float32 else C.c_double
        self._keep = []

    def fn(self, name, restype=None):
        f = getattr(_lib, "%s_%s" % (name, self.suf))
        f.restype = restype
        return f

    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dt)

    # ---- model tables
    def dmat(self, distances, codefreq, eigenval, eigentot):
        arrs = [self.arr(x) for x in (distances, codefreq, eigenval, eigentot)]
        s = _DMat(*[_ptr(a) for a in arrs])
        self._keep.append((arrs, s))
        return s

    def tmat(self, stat, statinv, eigenval, codefreq, eigeninv, eigeninvT):
        arrs = [self.arr(x) for x in (stat, statinv, eigenval, codefreq, eigeninv, eigeninvT)]
        s = _TMat(*[_ptr(a) for a in arrs])
        self._keep.append((arrs, s))
        return s

    @staticmethod
    def _ref(s):
        return None if s is None else C.byref(s)

    # ---- distances
    def seqdist(self, c1, c2, n_codes=4, distances=None):
        c1 = np.ascontiguousarray(c1, np.uint8)
        c2 = np.ascontiguousarray(c2, np.uint8)
        d, w = self.creal(), self.creal()
        dist = None if distances is None else self.arr(distances)
        self.fn("vfto_seqdist")(_ptr(c1), _ptr(c2), I64(len(c1)), C.c_int(n_codes), _ptr(dist), C.byref(d), C.byref(w))
        return self.dt.type(d.value), self.dt.type(w.value)

    def profiledist(self, p1, p2, cd2=None, dm=None):
        w1, c1, f1 = p1
        w2, c2, f2 = p2
        n_pos, n_codes = f1.shape
        d, w = self.creal(), self.creal()
        cd = None if cd2 is None else self.arr(cd2)
        self.fn("vfto_profiledist")(_ptr(w1), _ptr(c1), _ptr(f1), _ptr(w2), _ptr(c2), _ptr(f2), _ptr(cd), I64(n_pos),
                                    C.c_int(n_codes), self._ref(dm), C.byref(d), C.byref(w))
        return self.dt.type(d.value), self.dt.type(w.value)

    def new_profile(self, n_pos, n_codes):
        return (np.zeros(n_pos, self.dt), np.full(n_pos, NOCODE, np.uint8), np.zeros((n_pos, n_codes), self.dt))

    def leaf_profile(self, codes, n_codes):
        codes = np.ascontiguousarray(codes, np.uint8)
        return ((codes != NOCODE).astype(self.dt), codes.copy(), np.zeros((len(codes), n_codes), self.dt))

    def average_profile(self, p1, p2, bionj_weight=-1.0, dm=None, tol=1e-10):
        n_pos, n_codes = p1[2].shape
        out = self.new_profile(n_pos, n_codes)
        self.fn("vfto_average_profile")(_ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(p1[0]), _ptr(p1[1]),
                                        _ptr(p1[2]), _ptr(p2[0]), _ptr(p2[1]), _ptr(p2[2]), I64(n_pos),
                                        C.c_int(n_codes), C.c_double(bionj_weight), self._ref(dm), C.c_double(tol))
        return out

    def out_profile(self, W, Cc, F, dm=None, tol=1e-10):
        n, n_pos, n_codes = F.shape
        out = self.new_profile(n_pos, n_codes)
        cd = np.zeros((n_pos, n_codes), self.dt) if dm is not None else None
        self.fn("vfto_out_profile")(_ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(cd), _ptr(W), _ptr(Cc), _ptr(F),
                                    I64(n), I64(n_pos), C.c_int(n_codes), self._ref(dm), C.c_double(tol))
        return out, cd

    def update_out_profile(self, out, cd, p1, p2, pn, n_active_old, dm=None, tol=1e-10):
        wo, co, fo = out[0].copy(), out[1].copy(), out[2].copy()
        cdo = None if cd is None else cd.copy()
        n_pos, n_codes = fo.shape
        self.fn("vfto_update_out_profile")(_ptr(wo), _ptr(fo), _ptr(cdo), _ptr(co), _ptr(p1[0]), _ptr(p1[1]),
                                           _ptr(p1[2]), _ptr(p2[0]), _ptr(p2[1]), _ptr(p2[2]), _ptr(pn[0]),
                                           _ptr(pn[1]), _ptr(pn[2]), I64(n_active_old), I64(n_pos),
                                           C.c_int(n_codes), self._ref(dm), C.c_double(tol))
        return (wo, co, fo), cdo

    def out_distance(self, dist, weight, n_active, selfweight, selfdist, diameter, totdiam):
        f = self.fn("vfto_out_distance", self.creal)
        return self.dt.type(f(self.creal(dist), self.creal(weight), I64(n_active), self.creal(selfweight),
                              self.creal(selfdist), self.creal(diameter), C.c_double(totdiam)))

    def criterion(self, dist, out_i, n_out_i, out_j, n_out_j, n_active):
        f = self.fn("vfto_criterion", self.creal)
        return self.dt.type(f(self.creal(dist), self.creal(out_i), I64(n_out_i), self.creal(out_j), I64(n_out_j),
                              I64(n_active)))

    def state(self, n_seqs, W, Cc, F, parent, diameter, selfweight, selfdist, totdiam, outp, out_cd=None, dm=None):
        maxnode, n_pos, n_codes = F.shape
        arrs = [self.arr(W), np.ascontiguousarray(Cc, np.uint8), self.arr(F), np.ascontiguousarray(parent, np.int64),
                self.arr(diameter), self.arr(selfweight), self.arr(selfdist), self.arr(outp[0]),
                np.ascontiguousarray(outp[1], np.uint8), self.arr(outp[2]),
                None if out_cd is None else self.arr(out_cd)]
        s = _State(n_seqs, maxnode, n_pos, n_codes, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]),
                   _ptr(arrs[4]), _ptr(arrs[5]), _ptr(arrs[6]), float(totdiam), _ptr(arrs[7]), _ptr(arrs[8]),
                   _ptr(arrs[9]), _ptr(arrs[10]), None if dm is None else C.cast(C.pointer(dm), P))
        self._keep.append((arrs, s, dm))
        return s

    def set_best_hit(self, st, node, n_active, n_diff_allow, out_dist, n_out_active):
        n = st.maxnode
        od = self.arr(out_dist).copy()
        na = np.ascontiguousarray(n_out_active, np.int64).copy()
        hi, hj = np.zeros(n, np.int64), np.zeros(n, np.int64)
        hw, hd, hc = np.zeros(n, self.dt), np.zeros(n, self.dt), np.zeros(n, self.dt)
        best = I64(-1)
        self.fn("vfto_set_best_hit")(C.byref(st), I64(node), I64(n_active), I64(n_diff_allow), _ptr(od), _ptr(na),
                                     _ptr(hi), _ptr(hj), _ptr(hw), _ptr(hd), _ptr(hc), C.byref(best))
        return dict(i=hi, j=hj, weight=hw, dist=hd, crit=hc, best_j=best.value, outdist=od, noutactive=na)

    def avx2_sweep(self, st, node, n_active, out_dist, n_out_active, threads=0, repeat=1):
        """The AVX2 + OpenMP restatement of the sweep (oracle/vft_oracle_avx2.c; nucleotides, no matrix, float32,
        out-distances taken as fresh): dict of weight / dist / crit per target.  threads <= 0: every core."""
        assert self.dt == np.float32
        n = st.maxnode
        od = self.arr(out_dist)
        na = np.ascontiguousarray(n_out_active, np.int64)
        hw, hd, hc = np.zeros(n, self.dt), np.zeros(n, self.dt), np.zeros(n, self.dt)
        fn = _avx2().vfto_avx2_sweep_f32
        fn.restype = None
        for _ in range(repeat):
            fn(C.byref(st), I64(node), I64(n_active), _ptr(od), _ptr(na), _ptr(hw), _ptr(hd), _ptr(hc), C.c_int(threads))
        return dict(weight=hw, dist=hd, crit=hc)

    def avx2_sweep_bench(self, st, queries, n_active, out_dist, n_out_active, threads=0, budget=5.0):
        """bench.py's timed CPU leg (vfto_avx2_sweep_bench_f32): the sweeps of `queries`, round-robin for `budget` seconds,
        over a copy of the state that the OpenMP team allocates and first-touches itself.  Returns (seconds inside the
        sweeps, sweeps done, dict of the last sweep's weight / dist / crit)."""
        assert self.dt == np.float32
        n = st.maxnode
        od = self.arr(out_dist)
        na = np.ascontiguousarray(n_out_active, np.int64)
        q = np.ascontiguousarray(queries, np.int64)
        hw, hd, hc = np.zeros(n, self.dt), np.zeros(n, self.dt), np.zeros(n, self.dt)
        done = I64(0)
        fn = _avx2().vfto_avx2_sweep_bench_f32
        fn.restype = C.c_double
        secs = fn(C.byref(st), _ptr(q), I64(len(q)), I64(n_active), _ptr(od), _ptr(na), C.c_int(threads), C.c_double(budget),
                  _ptr(hw), _ptr(hd), _ptr(hc), C.byref(done))
        return float(secs), int(done.value), dict(weight=hw, dist=hd, crit=hc)

    def sort_hits(self, crit):
        crit = self.arr(crit)
        order = np.zeros(len(crit), np.int64)
        self.fn("vfto_sort_hits")(_ptr(crit), I64(len(crit)), _ptr(order))
        return order

    # ---- likelihood
    def pair_loglk(self, p1, p2, length, rates, ratecat, tm=None, min_rel=2.5e-4, site_lk=None):
        n_pos, n_codes = p1[2].shape
        rates = self.arr(rates)
        ratecat = np.ascontiguousarray(ratecat, np.int64)
        f = self.fn("vfto_pair_loglk", C.c_double)
        return f(_ptr(p1[0]), _ptr(p1[1]), _ptr(p1[2]), _ptr(p2[0]), _ptr(p2[1]), _ptr(p2[2]), I64(n_pos),
                 C.c_int(n_codes), C.c_double(length), _ptr(rates), C.c_int(len(rates)), _ptr(ratecat),
                 self._ref(tm), C.c_double(min_rel), _ptr(site_lk))

    def posterior_profile(self, p1, p2, len1, len2, rates, ratecat, tm=None, min_len=5e-4, min_rel=2.5e-4):
        n_pos, n_codes = p1[2].shape
        rates = self.arr(rates)
        ratecat = np.ascontiguousarray(ratecat, np.int64)
        out = self.new_profile(n_pos, n_codes)
        self.fn("vfto_posterior_profile")(_ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(p1[0]), _ptr(p1[1]),
                                          _ptr(p1[2]), _ptr(p2[0]), _ptr(p2[1]), _ptr(p2[2]), I64(n_pos),
                                          C.c_int(n_codes), C.c_double(len1), C.c_double(len2), _ptr(rates),
                                          C.c_int(len(rates)), _ptr(ratecat), self._ref(tm), C.c_double(min_len),
                                          C.c_double(min_rel))
        return out

    def profile_hash(self, p):
        n_pos, n_codes = p[2].shape
        _lib.vfto_profile_hash.restype = I64
        return _lib.vfto_profile_hash(_ptr(p[0]), _ptr(p[1]), _ptr(p[2]), I64(n_pos), C.c_int(n_codes),
                                      C.c_int(self.dt.itemsize))


def tolerances(dtype):
    """(MLMinBranchLength, MLMinRelBranchLength, fPostTotalTolerance): Constants.h:26-39."""
    if np.dtype(dtype) == np.float32:
        return 5.0e-4, 2.5e-4, 1.0e-10
    return 5.0e-9, 2.5e-9, 1.0e-20


def knuth_stream(n):
    """The reference's knuth_rand() stream from its default state (oracle/vft_knuth.h)."""
    out = np.zeros(n, np.float64)
    _lib.vfto_knuth_stream(_ptr(out), I64(n))
    return out


def knuth_selftest(rounds, length):
    _lib.vfto_knuth_selftest.restype = C.c_long
    return int(_lib.vfto_knuth_selftest(C.c_int(rounds), C.c_int(length)))
