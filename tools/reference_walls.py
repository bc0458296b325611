#!/usr/bin/env python3
"""Wall-clock of the compiled reference (oracle/_ref/VeryFastTree, built by oracle/Makefile from /root/reference) on THIS box's host cores
for the complete pipelines bench.py times on the GPU: config C2 (`-nt -gtr`) and C5 (`-lg -double-precision`) at one thread and at the
thread count of the schedule the GPU legs follow.  One JSON object; profiles/r05_reference_walls_gpu_box.json holds a run on a GPU box of
the pool (bench.py quotes it next to its own wall-clock).  usage: reference_walls.py [c2] [c5] [--threads-only]"""
import json, os, subprocess, sys, tempfile, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import synth
REF = os.path.join(ROOT, "oracle", "_ref", "VeryFastTree")
CASES = {"c2": (10000, 1000, 4, ["-nt", "-gtr"], 64), "c5": (50000, 300, 20, ["-lg", "-double-precision"], 128)}
which = [a for a in sys.argv[1:] if a in CASES] or ["c2", "c5"]
out = dict(cores=os.cpu_count(), cpu=next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown"))
try:
    out["cores_allowed"] = len(os.sched_getaffinity(0))
except AttributeError:
    pass
with tempfile.TemporaryDirectory() as tmp:
    jobs = []
    for w in which:
        n, L, nc, flags, T = CASES[w]
        fa = os.path.join(tmp, w + ".fa")
        synth.codes_to_fasta(synth.random_descent_codes(n, L, nc, 0.03, 0.01, seed=2), fa, synth.ALPHABET_NT if nc == 4 else synth.ALPHABET_AA)
        for threads in ([T] if "--threads-only" in sys.argv else [1, T]):
            jobs.append((w, threads, flags, fa))
    # the one-thread runs side by side (one core each), then the threaded runs one at a time
    running = []
    for w, threads, flags, fa in [j for j in jobs if j[1] == 1]:
        t0 = time.time()
        p = subprocess.Popen([REF] + flags + ["-threads", "1", "-seed", "1", fa], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        running.append((w, threads, p, t0))
    for w, threads, p, t0 in running:
        tree = p.communicate()[0]
        out["%s_threads_%d" % (w, threads)] = dict(wall_s=round(time.time() - t0, 1), newick_crc=zlib.crc32(tree.decode().strip().encode()), flags=" ".join(CASES[w][3]))
    for w, threads, flags, fa in [j for j in jobs if j[1] != 1]:
        t0 = time.time()
        tree = subprocess.run([REF] + flags + ["-threads", str(threads), "-seed", "1", fa], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        out["%s_threads_%d" % (w, threads)] = dict(wall_s=round(time.time() - t0, 1), newick_crc=zlib.crc32(tree.decode().strip().encode()), flags=" ".join(flags))
print(json.dumps(out))
