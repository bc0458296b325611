#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not collected by pytest; run on the GPU box):

    python tests/compare_with_reference_run.py N L [flags...] [--aa] [--mu M] [--gap G] [--seed S] [--threads T] [--out FILE]

One whole-pipeline run of this backend and one of the compiled reference (oracle/_ref/VeryFastTree, -threads 1 unless
--threads) on the same synthetic alignment, compared: TreeLogLk lines, differing splits, branch lengths, supports, and
the wall-clock of both.  flags: -gtr -lg -wag -double-precision -fastest (passed to the reference; mapped for the
backend).  BASELINE config C2 = `10000 1000 -gtr`, C5 = `50000 300 --aa -lg -double-precision`."""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps, synth  # noqa: E402
from veryfasttree_amd.backend import nj_newick  # noqa: E402

REFBIN = os.path.join(ROOT, "oracle", "_ref", "VeryFastTree")


def splits(newick):
    names = re.findall(r"[(,]([^(),:;]+):", newick)
    first, allset = names[0], frozenset(names)
    stack, out = [], set()
    for tok in re.findall(r"\(|\)|[^(),:;]+(?=:)|,", newick):
        if tok == "(":
            stack.append(set())
        elif tok == ")":
            top = stack.pop()
            if stack:
                stack[-1] |= top
            side = frozenset(top) if first not in top else allset - frozenset(top)
            if 1 < len(side) < len(allset) - 1:
                out.add(side)
        elif tok != ",":
            if tok in allset and stack:
                stack[-1].add(tok)
    return out


def main():
    args = sys.argv[1:]
    n, L = int(args[0]), int(args[1])
    opt = dict(mu=0.03, gap=0.01, seed=2, threads=1, out=None)
    flags, aa = [], False
    i = 2
    while i < len(args):
        a = args[i]
        if a == "--aa":
            aa = True
        elif a == "--noref":
            opt["noref"] = True
        elif a in ("--mu", "--gap", "--seed", "--threads", "--out"):
            opt[a[2:]] = args[i + 1]
            i += 1
        else:
            flags.append(a)
        i += 1
    mu, gap, seed, threads = float(opt["mu"]), float(opt["gap"]), int(opt["seed"]), int(opt["threads"])
    dt = np.float64 if "-double-precision" in flags else np.float32
    codes = synth.random_descent_codes(n, L, 20 if aa else 4, mu, gap, seed=seed)
    names = ["s%d" % k for k in range(n)]
    lines = []
    say = lambda s: (lines.append(s), print(s, flush=True))
    say("alignment %d x %d %s, mu %g gap %g seed %d; flags %s; %s" % (n, L, "aa" if aa else "nt", mu, gap, seed, " ".join(flags) or "(default)",
                                                                      "float64" if dt == np.float64 else "float32"))
    # --threads T: both sides run the T-thread schedule (the reference with -threads T, this backend with threads = T: the
    # lanes of treePartitioning in lockstep); T = 1 is the one-thread order
    kw = dict(dtype=dt, me_lengths=True, me_nni=True, spr=2, ml_nni=20, n_bootstrap=1000, return_loglk=True, fastest="-fastest" in flags,
              threads=threads, second_level="-fastest" in flags and threads == 1)
    if aa:
        kw["aa_model"] = "lg" if "-lg" in flags else "wag" if "-wag" in flags else "jtt"
    elif "-gtr" in flags:
        kw["gtr"] = True
    t0 = time.perf_counter()
    tree, loglk = nj_newick(lambda m, Lp: HipProfileOps(m, Lp, 20 if aa else 4, dt, max_nodes=3 * m), codes, names, **kw)
    t_gpu = time.perf_counter() - t0
    say("this backend, 1 x MI355X, schedule of %d thread(s): %.1f s; TreeLogLk %s" % (threads, t_gpu, " ".join("%.4f" % x for x in loglk)))
    if opt.get("noref"):
        return
    with tempfile.TemporaryDirectory() as tmp:
        fa, log = os.path.join(tmp, "a.fa"), os.path.join(tmp, "a.log")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_AA if aa else synth.ALPHABET_NT)
        cmd = [REFBIN] + ([] if aa else ["-nt"]) + flags + ["-threads", str(threads), "-seed", "1", "-log", log, fa]
        t0 = time.perf_counter()
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=dict(os.environ, OMP_WAIT_POLICY="passive"))
        t_ref = time.perf_counter() - t0
        text = open(log).read()
    ref = res.stdout.decode().strip()
    rll = [float(m.group(1)) for m in re.finditer(r"^TreeLogLk\tML_NNI\d+\t(\S+)", text, re.M)]
    m2 = re.search(r"^TreeLogLk\tML_Lengths2\t(\S+)", text, re.M)
    if m2:
        rll.append(float(m2.group(1)))
    say("reference, %d CPU thread(s) of %d: %.1f s; TreeLogLk %s" % (threads, os.cpu_count(), t_ref, " ".join("%.4f" % x for x in rll)))
    a, b = splits(tree), splits(ref)
    say("splits differing: %d of %d" % (len(a ^ b) // 2, len(b)))
    if a == b:
        strip = lambda t: re.sub(r"\)[0-9.]+:", "):", re.sub(r":[0-9.eE+-]+", ":", t))
        if strip(tree) == strip(ref):
            gl = np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", tree)])
            rl = np.array([float(x) for x in re.findall(r":([0-9.eE+-]+)", ref)])
            gs = np.array([float(x) for x in re.findall(r"\)([0-9.]+):", tree)])
            rs = np.array([float(x) for x in re.findall(r"\)([0-9.]+):", ref)])
            say("printed branch lengths differing: %d of %d (max %.3g); supports differing: %d of %d (max %.3g)" % (
                int((gl != rl).sum()), len(rl), np.abs(gl - rl).max(), int((gs != rs).sum()), len(rs), np.abs(gs - rs).max() if len(rs) else 0))
        else:
            say("same splits, different child order")
    say("byte-identical output: %s" % ("YES" if tree == ref else "no"))
    if len(loglk) == len(rll):
        say("TreeLogLk max relative difference: %.3g" % max(abs(x - y) / abs(y) for x, y in zip(loglk, rll)))
    if opt["out"]:
        os.makedirs(os.path.dirname(os.path.abspath(opt["out"])), exist_ok=True)
        open(opt["out"], "a").write("\n".join(lines) + "\n\n")


if __name__ == "__main__":
    main()
