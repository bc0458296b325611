"""Deterministic synthetic alignments (SURVEY.md §8d "random-descent" generator).

seq_0 is i.i.d. uniform over the alphabet; seq_k copies a uniformly chosen earlier sequence and
re-draws each site with probability ``mu``; finally every site becomes a gap with probability ``gap``.
The descent forest is evaluated level by level so that 1M x 200 takes seconds in numpy.

Codes follow the reference's leaf encoding (NJ.tcc:415-457, TransitionMatrix.h:7): 0..nCodes-1 in
alphabet order, 127 (NOCODE) for a gap.
"""
import numpy as np

NOCODE = 127
ALPHABET_NT = "ACGT"
ALPHABET_AA = "ARNDCQEGHILKMFPSTWYV"


def random_descent_codes(n_seq, n_pos, n_codes=4, mu=0.03, gap=0.01, seed=1):
    """Return uint8 codes[n_seq, n_pos]; depths of the descent forest come from pointer jumping."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parent = np.zeros(n_seq, dtype=np.int64)
    parent[1:] = (rng.random(n_seq - 1) * np.arange(1, n_seq)).astype(np.int64)
    depth = np.zeros(n_seq, dtype=np.int64)
    anc = parent.copy()
    alive = np.arange(n_seq) > 0
    while alive.any():
        depth[alive] += 1
        alive = alive & (anc > 0)
        anc = parent[anc]
    codes = np.empty((n_seq, n_pos), dtype=np.uint8)
    codes[0] = rng.integers(0, n_codes, size=n_pos, dtype=np.uint8)
    order = np.argsort(depth, kind="stable")
    bounds = np.searchsorted(depth[order], np.arange(1, depth.max() + 2))
    for d in range(1, int(depth.max()) + 1):
        idx = order[bounds[d - 1]:bounds[d]]
        block = codes[parent[idx]]
        mut = rng.random(block.shape) < mu
        block[mut] = rng.integers(0, n_codes, size=int(mut.sum()), dtype=np.uint8)
        codes[idx] = block
    if gap > 0:
        codes[rng.random(codes.shape) < gap] = NOCODE
    return codes


def codes_to_fasta(codes, path, alphabet=ALPHABET_NT):
    lut = np.full(256, ord("-"), dtype=np.uint8)
    for i, ch in enumerate(alphabet):
        lut[i] = ord(ch)
    with open(path, "w") as fh:
        for k, row in enumerate(codes):
            fh.write(">s%d\n%s\n" % (k, lut[row].tobytes().decode("ascii")))


def fasta_to_codes(path, alphabet=ALPHABET_NT):
    """Minimal FASTA reader with the reference's normalisation (Alignment.cpp:453-473: U->T for nt,
    '.'->'-', unknown -> gap) and first-occurrence uniquify (Alignment.cpp:494-526)."""
    lut = np.full(256, NOCODE, dtype=np.uint8)
    for i, ch in enumerate(alphabet):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    if alphabet == ALPHABET_NT:
        lut[ord("U")] = lut[ord("T")]
        lut[ord("u")] = lut[ord("T")]
    names, seqs, cur = [], [], []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if names:
                    seqs.append("".join(cur))
                names.append(line[1:].split()[0])
                cur = []
            else:
                cur.append(line.strip())
    if names:
        seqs.append("".join(cur))
    seen, uniq = {}, []
    for s in seqs:
        if s not in seen:
            seen[s] = len(uniq)
            uniq.append(s)
    arr = np.stack([lut[np.frombuffer(s.encode("ascii"), dtype=np.uint8)] for s in uniq])
    return names, arr
