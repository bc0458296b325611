#!/usr/bin/env python3
"""Wall-clock of the NJ phase: C++ host driver on the HIP backend vs the reference binary (when oracle/_ref travelled),
same alignment, and join-order comparison through the reference's Join lines.
usage: nj_wallclock.py N L [fastest] [threads]"""
import os, re, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth
from veryfasttree_amd.backend import nj_run

n, L = int(sys.argv[1]), int(sys.argv[2])
fastest = len(sys.argv) > 3 and sys.argv[3] == "fastest"
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 1
codes = synth.random_descent_codes(n, L, 4, 0.03, 0.01, seed=3)
_, first = np.unique(codes, axis=0, return_index=True)
codes = codes[np.sort(first)]
print("unique sequences:", codes.shape[0])
ops = HipProfileOps(codes.shape[0], L, 4, np.float32)
t0 = time.perf_counter()
joins, crit = nj_run(ops, codes, fastest=fastest, second_level=False)
t_gpu = time.perf_counter() - t0
print("GPU NJ phase: %.2f s for %d joins" % (t_gpu, len(joins)))
ref = os.path.join(ROOT, "oracle", "_ref", "VeryFastTree")
if os.path.exists(ref):
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "a.fa")
        synth.codes_to_fasta(codes, fa)
        flags = ["-nt"] + (["-fastest", "-no2nd"] if fastest else [])
        for th, verbose in ((1, True), (threads, False)):
            if th == 1 and not verbose:
                continue
            log = os.path.join(tmp, "log%d" % th)
            cmd = [ref] + flags + ["-threads", str(th), "-noml", "-nome", "-nosupport", "-log", log] + (["-verbose", "3"] if verbose else []) + [fa]
            t0 = time.perf_counter()
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            print("reference (%d thread%s%s): %.2f s  -> speedup %.1fx" % (th, "s" if th > 1 else "", ", verbose 3" if verbose else "", dt, dt / t_gpu))
            if verbose:
                text = open(log).read() + res.stderr.decode(errors="replace")
                rj = {}
                for mm in re.finditer(r"^Join\t(\d+)\t(\d+)\t\S+\tlambda\t\S+\tselfw\t\S+\t\S+\tnew\t(\d+)", text, re.M):
                    rj[int(mm.group(3))] = (int(mm.group(1)), int(mm.group(2)))
                same = sum(1 for a, b, c in joins if rj.get(int(c)) == (int(a), int(b)))
                print("join order identical to the reference for %d of %d joins" % (same, len(joins)))
