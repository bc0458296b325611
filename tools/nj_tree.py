#!/usr/bin/env python3
"""FASTA (nucleotides) -> Newick on one MI355X: the tree `VeryFastTree -nt [-fastest] -noml -nome [-nosupport]` prints,
or - with -mllen - the tree of `VeryFastTree -nt -nome -mllen [-nocat | -cat N] [-nosupport]` (Jukes-Cantor).

    python tools/nj_tree.py in.fasta [-fastest] [-double] [-nosupport] [-nj-lengths] [-mllen [-nocat | -cat N]] > tree.nwk
    python tools/nj_tree.py in.fasta -full [-gtr] [-double] [-nosupport] > tree.nwk     # what plain `VeryFastTree -nt [-gtr]` prints
    python tools/nj_tree.py in.fasta -full -lg -double > tree.nwk     # proteins: `VeryFastTree -lg -double-precision` (-aa / -jtt, -wag, -lg)
    python tools/nj_tree.py in.fasta -full -threads 64 [-gamma] [-spr N] > tree.nwk   # the schedule of `VeryFastTree -threads 64`; -gamma; -spr N rounds

Neighbour joining with top hits on the device (veryfasttree_amd/host/NJDriver.h), the root, minimum-evolution branch
lengths (updateBranchLengths), local-bootstrap supports (1000 resamples, reliabilityNJ) and printNJ; -nj-lengths keeps
the NJ branch lengths and prints no supports (the reference's "NJ" log line).  -mllen: maximum-likelihood branch
lengths on that topology (optimizeAllBranchLengths rounds, CAT rate categories unless -nocat) and SH-like supports
(testSplitsML, 1000 resamples) unless -nosupport; the TreeLogLk of every round goes to stderr.
Sequence normalisation and uniquify follow Alignment.cpp:453-526 (U -> T, '.' -> '-', duplicates by sequence string in
first-occurrence order; N -> X for nucleotides)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veryfasttree_amd import HipProfileOps
from veryfasttree_amd.backend import nj_newick
from veryfasttree_amd.synth import ALPHABET_AA, ALPHABET_NT, NOCODE


def read_fasta(path):
    names, seqs, cur = [], [], []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if names:
                    seqs.append("".join(cur))
                names.append(line[1:].split()[0] if line[1:].split() else "")
                cur = []
            elif names:
                cur.append(line.strip())
    if names:
        seqs.append("".join(cur))
    return names, seqs


def main():
    args = sys.argv[1:]
    if not args or args[0].startswith("-"):
        sys.exit(__doc__)
    fastest, double, nj_len = "-fastest" in args, "-double" in args, "-nj-lengths" in args
    mllen = 0
    if "-mllen" in args:
        mllen = 1 if "-nocat" in args else (int(args[args.index("-cat") + 1]) if "-cat" in args else 20)
    n_boot = 0 if ("-nosupport" in args or nj_len) else 1000
    extra = dict(me_nni=True, spr=2, ml_nni=20) if "-full" in args else {}
    if "-gtr" in args:
        extra["gtr"] = True   # ME NNIs + SPRs, ML NNIs, CAT, SH supports
    if "-spr" in args and extra:
        extra["spr"] = int(args[args.index("-spr") + 1])
    if "-threads" in args:   # the refinement stages on the schedule of a T-thread run of the reference (include/vft_host.h, vft_nj_options.threads)
        extra["threads"] = int(args[args.index("-threads") + 1])
    if "-gamma" in args and extra:
        extra["gamma"] = True
    names, seqs = read_fasta(args[0])
    if len({len(s) for s in seqs}) != 1:
        sys.exit("sequences have different lengths: not an alignment")
    aa = None
    for flag, model in (("-aa", "jtt"), ("-jtt", "jtt"), ("-wag", "wag"), ("-lg", "lg")):
        if flag in args:
            aa = model
    # Alignment.cpp:453-471: '.' -> '-' always; for nucleotides U -> T and N -> X (before the sequences are uniquified)
    seqs = [s.upper().replace(".", "-") for s in seqs]
    if aa is None:
        seqs = [s.replace("U", "T").replace("N", "X") for s in seqs]
    first_of, last, unique_first = {}, {}, []
    aln_next = np.full(len(seqs), -1, np.int64)
    for k, s in enumerate(seqs):
        if s not in first_of:
            first_of[s] = k
            unique_first.append(k)
        else:
            aln_next[last[s]] = k
        last[s] = k
    lut = np.full(256, NOCODE, np.uint8)
    for i, ch in enumerate(ALPHABET_AA if aa else ALPHABET_NT):
        lut[ord(ch)] = i
    codes_all = np.stack([lut[np.frombuffer(s.encode("ascii", "replace"), np.uint8)] for s in seqs])
    n_unique = len(unique_first)
    if n_unique < 3:
        sys.exit("fewer than 3 unique sequences")
    dt = np.float64 if double else np.float32
    if aa:
        extra["aa_model"] = aa
    tree, loglk = nj_newick(lambda n, L: HipProfileOps(n, L, 20 if aa else 4, dt, max_nodes=3 * n), codes_all, names, fastest=fastest,
                            dtype=dt, me_lengths=not nj_len, unique=(np.array(unique_first, np.int64), aln_next),
                            n_bootstrap=n_boot, mllen=mllen, return_loglk=True, **extra)
    for k, ll in enumerate(loglk):
        sys.stderr.write("TreeLogLk\t%s%d\t%.4f\n" % ("Round" if extra else "Length", k + 1, ll))
    print(tree)


if __name__ == "__main__":
    main()
