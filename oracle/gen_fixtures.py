#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerates tests/golden/*.npz from the compiled reference.

Run in the authoring container only (needs /root/reference and `make -C oracle ref whitebox`):

    python oracle/gen_fixtures.py

White-box fixtures: oracle/_ref/whitebox (oracle/whitebox.cpp linked against the reference's own
objects) dumps inputs/outputs of the reference's private hot-path members on small synthetic alignments.
Black-box fixtures: oracle/_ref/VeryFastTree runs with `-threads 1 -verbose 3 -log` and the join order /
stage log-likelihoods are parsed from its log.

Only data (inputs + expected outputs) is written to tests/golden/.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from veryfasttree_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
WHITEBOX = os.path.join(HERE, "_ref", "whitebox")
REFBIN = os.path.join(HERE, "_ref", "VeryFastTree")

DTYPES = {b"f": np.float32, b"d": np.float64, b"i": np.int32, b"q": np.int64, b"b": np.uint8}


def read_vfx(path):
    out = {}
    with open(path, "rb") as fh:
        assert fh.read(4) == b"VFX1"
        while True:
            head = fh.read(4)
            if not head:
                break
            (nl,) = struct.unpack("<I", head)
            name = fh.read(nl).decode()
            dt = DTYPES[fh.read(1)]
            (nd,) = struct.unpack("<I", fh.read(4))
            dims = struct.unpack("<%dQ" % nd, fh.read(8 * nd)) if nd else ()
            count = int(np.prod(dims)) if nd else 1
            data = np.frombuffer(fh.read(count * np.dtype(dt).itemsize), dtype=dt)
            out[name] = data.reshape(dims) if nd else data.reshape(())
    return out


WHITEBOX_CASES = [
    # name, mode, n_seq, n_pos, n_codes, mu, gap, seed
    ("wb_nt_f32", "nt_f32", 160, 96, 4, 0.06, 0.04, 11),
    ("wb_nt_f32_gappy", "nt_f32", 48, 40, 4, 0.15, 0.30, 12),
    ("wb_nt_f64", "nt_f64", 40, 48, 4, 0.08, 0.05, 13),
    ("wb_aa_f32", "aa_f32", 56, 48, 20, 0.10, 0.05, 14),
    ("wb_aa_f64", "aa_f64", 56, 48, 20, 0.10, 0.05, 15),
]


def gen_whitebox(tmp):
    for name, mode, n, L, nc, mu, gap, seed in WHITEBOX_CASES:
        codes = synth.random_descent_codes(n, L, nc, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_AA if nc == 20 else synth.ALPHABET_NT)
        vfx = os.path.join(tmp, name + ".vfx")
        subprocess.run([WHITEBOX, mode, fa, vfx, str(seed)], check=True)
        d = read_vfx(vfx)
        d["input.codes"] = codes  # the alignment as generated (before the reference's uniquify)
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, **d)
        print("%-20s %4d arrays  %7.1f KiB" % (name, len(d), os.path.getsize(dst) / 1024.0))


PARTITION_CASES = [("wb_partition_40", 40, 60, 0.10, 91), ("wb_partition_300", 300, 100, 0.06, 92), ("wb_partition_2000", 2000, 60, 0.04, 93)]


def gen_partition(tmp):
    """White box: the reference's own treePartitioning(penalty) (NJ.tcc:5540-5750, a private member called through oracle/whitebox.cpp)
    on the NJ tree of three alignments, penalty 1 and 2, for 2 ... 64 threads - what vft_tree_partitioning has to return."""
    for name, n, L, mu, seed in PARTITION_CASES:
        codes = synth.random_descent_codes(n, L, 4, mu, 0.01, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
        vfx = os.path.join(tmp, name + ".vfx")
        subprocess.run([WHITEBOX, "nt_f32:partition", fa, vfx, str(seed)], check=True)
        d = read_vfx(vfx)
        keep = {k: v for k, v in d.items() if k.startswith("part.") or k in ("nj.child", "nj.parent", "nj.root", "nj.nchild", "nSeqs")}
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, **keep)
        print("%-20s %4d arrays  %7.1f KiB; subtrees at 8 threads, penalty 2: %d" % (name, len(keep), os.path.getsize(dst) / 1024.0,
                                                                                     int((keep["part.T8.p2"] >= 0).sum())))


def gen_knuth(tmp):
    """5000 values of the reference's knuth_rand() stream (Knuth.cpp; never re-seeded by the pipeline)."""
    vfx = os.path.join(tmp, "knuth.vfx")
    subprocess.run([WHITEBOX, "knuth", "-", vfx], check=True)
    d = read_vfx(vfx)
    dst = os.path.join(GOLDEN, "wb_knuth.npz")
    np.savez_compressed(dst, **d)
    print("%-20s %4d arrays  %7.1f KiB" % ("wb_knuth", len(d), os.path.getsize(dst) / 1024.0))


def gen_tables(tmp):
    """The amino-acid model tables the reference builds (JTT / WAG / LG transition tables, BLOSUM45 distance tables) in
    both precisions; the raw constants are not stored here (they are the product's data header)."""
    vfx = os.path.join(tmp, "tables.vfx")
    subprocess.run([WHITEBOX, "tables", "-", vfx], check=True)
    d = {k: v for k, v in read_vfx(vfx).items() if ".raw." not in k}
    dst = os.path.join(GOLDEN, "wb_aa_tables.npz")
    np.savez_compressed(dst, **d)
    print("%-20s %4d arrays  %7.1f KiB" % ("wb_aa_tables", len(d), os.path.getsize(dst) / 1024.0))


BLACKBOX_CASES = [
    # name, flags, n_seq, n_pos, n_codes, mu, gap, seed
    ("bb_nt_c1", ["-nt", "-fastest"], 16, 100, 4, 0.05, 0.0, 1),  # BASELINE config 1
    ("bb_nt_200", ["-nt"], 200, 120, 4, 0.05, 0.02, 21),
    ("bb_nt_600_fastest", ["-nt", "-fastest"], 600, 100, 4, 0.04, 0.02, 22),
    ("bb_nt_600_fastest_no2nd", ["-nt", "-fastest", "-no2nd"], 600, 100, 4, 0.04, 0.02, 22),
    ("bb_nt_1500", ["-nt"], 1500, 80, 4, 0.03, 0.01, 23),
    ("bb_nt_300_double", ["-nt", "-double-precision"], 300, 90, 4, 0.05, 0.03, 24),
    ("bb_aa_300", [], 300, 80, 20, 0.10, 0.03, 25),   # protein: BLOSUM45-derived distance matrix (the default)
    # tiny inputs: no top hits (m < 4), the visible-set search of fastNJ (NJ.tcc:2846-2852, 3049-3090, 3686-3744)
    ("bb_nt_10", ["-nt"], 10, 60, 4, 0.15, 0.02, 26),
    ("bb_nt_5", ["-nt"], 5, 40, 4, 0.2, 0.0, 27),
    ("bb_nt_12_fastest", ["-nt", "-fastest"], 12, 50, 4, 0.15, 0.05, 28),
    # top hits switched off on an input with more nodes than the device's sorted-hit buffer (8192): allhits[] in full
    ("bb_nt_4400_notop", ["-nt", "-notop"], 4400, 64, 4, 0.08, 0.02, 29),
]


def gen_blackbox(tmp, only=None):
    for name, flags, n, L, nc, mu, gap, seed in BLACKBOX_CASES:
        if only and name not in only:
            continue
        codes = synth.random_descent_codes(n, L, nc, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_AA if nc == 20 else synth.ALPHABET_NT)
        log = os.path.join(tmp, name + ".log")
        cmd = [REFBIN] + flags + ["-threads", "1", "-seed", "1", "-verbose", "3", "-noml", "-nome", "-nosupport",
                                  "-log", log, fa]
        res = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        joins = []
        text = open(log).read() + "\n" + res.stderr.decode(errors="replace")
        for m in re.finditer(r"^Join\t(\d+)\t(\d+)\t(\S+)\tlambda\t\S+\tselfw\t\S+\t\S+\tnew\t(\d+)", text, re.M):
            joins.append((int(m.group(1)), int(m.group(2)), int(m.group(4)), float(m.group(3))))
        seen, uniq = set(), []
        for j in joins:  # log + stderr may both carry the lines
            if j[2] not in seen:
                seen.add(j[2])
                uniq.append(j)
        joins = uniq
        assert joins, "no Join lines for " + name
        # the tree as fastNJ leaves it, NJ branch lengths included (logTree("NJ"), VeryFastTreeImpl.tcc:143)
        mnj = re.search(r"^NJ\t(\(.*;)\s*$", text, re.M)
        assert mnj, "no NJ tree line for " + name
        # the same pipeline with the default local-bootstrap supports (1000 resamples, reliabilityNJ)
        cmd2 = [REFBIN] + flags + ["-threads", "1", "-seed", "1", "-noml", "-nome", fa]
        res2 = subprocess.run(cmd2, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        ja = np.array([(a, b, c) for a, b, c, _ in joins], dtype=np.int64)
        jc = np.array([c for _, _, _, c in joins], dtype=np.float64)
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, codes=codes, joins=ja, join_criterion=jc,
                            newick=np.frombuffer(res.stdout, dtype=np.uint8),
                            nj_newick=np.frombuffer(mnj.group(1).encode(), dtype=np.uint8),
                            newick_support=np.frombuffer(res2.stdout, dtype=np.uint8),
                            flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        print("%-20s %5d joins  %7.1f KiB" % (name, len(joins), os.path.getsize(dst) / 1024.0))


MLLEN_CASES = [
    # name, flags, n_seq, n_pos, mu, gap, seed: `-nome -mllen -nosupport` + flags (Jukes-Cantor)
    ("ml_nt_200", ["-nt", "-nocat"], 200, 120, 0.05, 0.02, 21),            # same alignment as bb_nt_200
    ("ml_nt_400_double", ["-nt", "-double-precision", "-nocat"], 400, 150, 0.06, 0.03, 31),
    ("ml_nt_30", ["-nt", "-nocat"], 30, 300, 0.10, 0.05, 32),
    # the default CAT approximation: 20 rate categories fitted after the first round (setMLRates)
    ("ml_nt_200_cat", ["-nt"], 200, 120, 0.05, 0.02, 21),
    ("ml_nt_300_cat", ["-nt"], 300, 400, 0.08, 0.02, 33),
    ("ml_nt_150_double_cat", ["-nt", "-double-precision", "-cat", "8"], 150, 200, 0.08, 0.04, 34),
    # `-gtr`: rates and base frequencies fitted after the first round (setMLGtr), then CAT
    ("ml_nt_200_gtr", ["-nt", "-gtr"], 200, 120, 0.05, 0.02, 21),
    ("ml_nt_150_double_gtr", ["-nt", "-gtr", "-double-precision"], 150, 200, 0.08, 0.04, 34),
]


def parse_gtr(text):
    """(rates[6], freq[4]) from the 'GTR rates' / 'GTR Frequencies' lines, or zeros."""
    mr = re.search(r"^GTR rates\(ac ag at cg ct gt\)((?: \S+){6})", text, re.M)
    mf = re.search(r"^GTR Frequencies:((?: \S+){4})", text, re.M)
    if not mr:
        return np.zeros(6), np.zeros(4)
    return np.array([float(x) for x in mr.group(1).split()]), np.array([float(x) for x in mf.group(1).split()])


def gen_mllen(tmp):
    """Black box: NJ + ME lengths + ML lengths on the fixed topology; pins TreeLogLk per round and the final tree."""
    for name, flags, n, L, mu, gap, seed in MLLEN_CASES:
        codes = synth.random_descent_codes(n, L, 4, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
        log = os.path.join(tmp, name + ".log")
        cmd = [REFBIN] + flags + ["-threads", "1", "-seed", "1", "-nome", "-mllen", "-nosupport", "-log", log, fa]
        res = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(log).read()
        ll = [float(m.group(1)) for m in re.finditer(r"^TreeLogLk\tLength\d+\t(\S+)\tMaxChange", text, re.M)]
        assert ll, "no TreeLogLk lines for " + name
        rates = [float(x) for x in re.search(r"^Rates((?: \S+)+)$", text, re.M).group(1).split()]
        cats = [int(x) - 1 for x in re.search(r"^SiteCategories((?: \d+)+)$", text, re.M).group(1).split()]
        assert len(cats) == L
        # the same with the default SH-like supports (testSplitsML, 1000 resamples)
        cmd2 = [REFBIN] + flags + ["-threads", "1", "-seed", "1", "-nome", "-mllen", fa]
        res2 = subprocess.run(cmd2, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        mb = re.search(r"Bad splits: (\d+)/(\d+)", res2.stderr.decode(errors="replace"))
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, codes=codes, loglk=np.array(ll), newick=np.frombuffer(res.stdout, dtype=np.uint8),
                            newick_support=np.frombuffer(res2.stdout, dtype=np.uint8),
                            gtr_rates=parse_gtr(text)[0], gtr_freq=parse_gtr(text)[1],
                            bad_splits=np.array([int(mb.group(1)), int(mb.group(2))], dtype=np.int64),
                            rates=np.array(rates), ratecat=np.array(cats, dtype=np.int32),
                            flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        print("%-20s %2d rounds  final logLk %.4f  %7.1f KiB" % (name, len(ll), ll[-1], os.path.getsize(dst) / 1024.0))


MENNI_CASES = [
    # name, flags, n_seq, n_pos, n_codes, mu, gap, seed: `-noml -spr 0` = NJ + minimum-evolution NNIs + ME lengths
    ("nni_nt_200", ["-nt"], 200, 120, 4, 0.05, 0.02, 21),          # same alignment as bb_nt_200
    ("nni_nt_500", ["-nt"], 500, 100, 4, 0.10, 0.05, 41),
    ("nni_nt_300_double", ["-nt", "-double-precision"], 300, 90, 4, 0.08, 0.03, 42),
    ("nni_nt_12", ["-nt"], 12, 60, 4, 0.2, 0.05, 43),
    # `-noml` with the default two SPR rounds between the NNI rounds
    ("spr_nt_200", ["-nt", "-spr", "2"], 200, 120, 4, 0.05, 0.02, 21),
    ("spr_nt_500", ["-nt", "-spr", "2"], 500, 100, 4, 0.10, 0.05, 41),
    ("spr_nt_300_double", ["-nt", "-double-precision", "-spr", "2"], 300, 90, 4, 0.08, 0.03, 42),
    ("spr_nt_12", ["-nt", "-spr", "2"], 12, 60, 4, 0.2, 0.05, 43),
]


def gen_menni(tmp):
    """Black box: the ME_NNI<k> trees of every round (topology pins), the final tree without and with supports."""
    for name, flags, n, L, nc, mu, gap, seed in MENNI_CASES:
        codes = synth.random_descent_codes(n, L, nc, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_AA if nc == 20 else synth.ALPHABET_NT)
        log = os.path.join(tmp, name + ".log")
        base = [REFBIN] + flags + ["-threads", "1", "-seed", "1", "-noml"] + ([] if "-spr" in flags else ["-spr", "0"])
        res = subprocess.run(base + ["-nosupport", "-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(log).read()
        rounds = re.findall(r"^ME_NNI\d+\t(\(.*;)\s*$", text, re.M)
        mm = re.search(r"^NNI: (\d+) SPR: (\d+)", res.stderr.decode(errors="replace") + text, re.M)
        nni, nspr = int(mm.group(1)), int(mm.group(2))
        res2 = subprocess.run(base + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, codes=codes, n_nni=np.int64(nni), n_spr=np.int64(nspr), n_rounds=np.int64(len(rounds)),
                            last_round=np.frombuffer(rounds[-1].encode(), dtype=np.uint8),
                            newick=np.frombuffer(res.stdout, dtype=np.uint8),
                            newick_support=np.frombuffer(res2.stdout, dtype=np.uint8),
                            flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        print("%-20s %2d rounds %4d NNIs %3d SPRs  %7.1f KiB" % (name, len(rounds), nni, nspr, os.path.getsize(dst) / 1024.0))


MLNNI_CASES = [
    # name, flags, n_seq, n_pos, mu, gap, seed: maximum-likelihood NNIs under Jukes-Cantor, no SPR moves
    ("mlnni_nt_200_nocat", ["-nt", "-nome", "-nocat"], 200, 120, 0.05, 0.02, 21),   # NJ topology + ML NNIs, constant rates
    ("mlnni_nt_200", ["-nt", "-nome"], 200, 120, 0.05, 0.02, 21),                   # + CAT rates
    ("mlnni_nt_300_spr0", ["-nt", "-spr", "0"], 300, 200, 0.08, 0.03, 51),          # ME NNIs + ML NNIs + CAT: default minus SPR
    ("mlnni_nt_150_double", ["-nt", "-nome", "-double-precision"], 150, 150, 0.10, 0.04, 52),
    ("mlnni_nt_20", ["-nt", "-spr", "0"], 20, 100, 0.15, 0.05, 53),
    # the reference's complete default pipeline for nucleotides (Jukes-Cantor + CAT): NJ, ME NNIs + 2 SPR rounds, ML NNIs
    ("full_nt_200", ["-nt"], 200, 120, 0.05, 0.02, 21),
    ("full_nt_300", ["-nt"], 300, 200, 0.08, 0.03, 51),
    ("full_nt_250_double", ["-nt", "-double-precision"], 250, 150, 0.10, 0.04, 54),
    ("full_nt_200_gtr", ["-nt", "-gtr"], 200, 120, 0.05, 0.02, 21),                 # BASELINE C2's model: GTR + CAT
    ("full_nt_250_double_gtr", ["-nt", "-gtr", "-double-precision"], 250, 150, 0.10, 0.04, 54),
    # 16S-like length: more than 1024 columns (eight columns per thread in the quartet kernels)
    ("full_nt_40_x1500", ["-nt"], 40, 1500, 0.05, 0.02, 71),
    ("full_nt_30_x1300_gtr_double", ["-nt", "-gtr", "-double-precision"], 30, 1300, 0.06, 0.02, 72),
    # beyond the register-resident line-search kernels (2 048 columns): the workspace kernels (csrc/vft_kernels_ml_long.h)
    ("full_nt_40_x3000", ["-nt"], 40, 3000, 0.05, 0.02, 73),
    ("full_nt_30_x2500_gtr", ["-nt", "-gtr"], 30, 2500, 0.06, 0.02, 74),
    ("full_nt_24_x5000_double", ["-nt", "-double-precision"], 24, 5000, 0.05, 0.02, 75),
]


def gen_mlnni(tmp, only=None):
    """Black box: TreeLogLk after every ML NNI round and after the final length pass, NNI counts, final trees."""
    for name, flags, n, L, mu, gap, seed in MLNNI_CASES:
        if only and name not in only:
            continue
        codes = synth.random_descent_codes(n, L, 4, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
        log = os.path.join(tmp, name + ".log")
        base = [REFBIN] + flags + ["-threads", "1", "-seed", "1"]
        res = subprocess.run(base + ["-nosupport", "-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(log).read()
        ll = [float(m.group(1)) for m in re.finditer(r"^TreeLogLk\tML_NNI\d+\t(\S+)\tMaxChange", text, re.M)]
        ll.append(float(re.search(r"^TreeLogLk\tML_Lengths2\t(\S+)", text, re.M).group(1)))
        err = res.stderr.decode(errors="replace") + text
        m = re.search(r"^NNI: (\d+) SPR: (\d+) ML-NNI: (\d+)", err, re.M)
        res2 = subprocess.run(base + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, codes=codes, loglk=np.array(ll), n_me_nni=np.int64(m.group(1)), n_ml_nni=np.int64(m.group(3)),
                            gtr_rates=parse_gtr(text)[0], gtr_freq=parse_gtr(text)[1],
                            newick=np.frombuffer(res.stdout, dtype=np.uint8),
                            newick_support=np.frombuffer(res2.stdout, dtype=np.uint8),
                            flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        print("%-22s %2d ML NNI rounds %4d ML NNIs %4d ME NNIs  final logLk %.4f" % (name, len(ll) - 1, int(m.group(3)), int(m.group(1)), ll[-1]))


AA_CASES = [
    # name, kind, flags, n_seq, n_pos, mu, gap, seed.  Protein alignments (BASELINE config C5's path): the BLOSUM45-derived
    # distances in the NJ / ME phase, JTT (default) / WAG / LG + CAT in the ML phase.
    ("nni_aa_150", "menni", ["-spr", "2"], 150, 80, 0.10, 0.03, 61),                       # -noml: ME NNIs + SPRs, scoredist
    ("ml_aa_100_lg_double", "mllen", ["-lg", "-double-precision"], 100, 90, 0.10, 0.03, 62),   # -nome -mllen
    ("ml_aa_120_jtt", "mllen", [], 120, 70, 0.12, 0.04, 63),
    ("ml_aa_80_wag_nocat", "mllen", ["-wag", "-nocat"], 80, 60, 0.10, 0.02, 64),
    ("full_aa_120_lg_double", "mlnni", ["-lg", "-double-precision"], 120, 100, 0.10, 0.03, 65),   # C5's exact flags
    ("full_aa_150_lg", "mlnni", ["-lg"], 150, 80, 0.12, 0.04, 66),
    ("full_aa_100_jtt", "mlnni", [], 100, 90, 0.10, 0.03, 67),                              # the reference's protein default
    ("full_aa_90_wag_double", "mlnni", ["-wag", "-double-precision"], 90, 70, 0.08, 0.05, 68),
    ("full_aa_30_x2200_lg_double", "mlnni", ["-lg", "-double-precision"], 30, 2200, 0.10, 0.03, 69),   # > 2 048 columns: the workspace line searches
]


def gen_aa(tmp, only=None):
    """Black box, proteins: the same three kinds of runs as gen_menni / gen_mllen / gen_mlnni."""
    for name, kind, flags, n, L, mu, gap, seed in AA_CASES:
        if only and name not in only:
            continue
        codes = synth.random_descent_codes(n, L, 20, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_AA)
        log = os.path.join(tmp, name + ".log")
        base = [REFBIN] + flags + ["-threads", "1", "-seed", "1"]
        if kind == "menni":
            base += ["-noml"]
        elif kind == "mllen":
            base += ["-nome", "-mllen"]
        res = subprocess.run(base + ["-nosupport", "-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(log).read()
        err = res.stderr.decode(errors="replace") + text
        res2 = subprocess.run(base + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out = dict(codes=codes, newick=np.frombuffer(res.stdout, dtype=np.uint8),
                   newick_support=np.frombuffer(res2.stdout, dtype=np.uint8),
                   flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        m = re.search(r"^NNI: (\d+) SPR: (\d+)(?: ML-NNI: (\d+))?", err, re.M)
        nni, nspr = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        out["n_me_nni"], out["n_spr"] = np.int64(nni), np.int64(nspr)
        if kind == "mllen":
            ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\tLength\d+\t(\S+)\tMaxChange", text, re.M)]
        elif kind == "mlnni":
            ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\tML_NNI\d+\t(\S+)\tMaxChange", text, re.M)]
            ll.append(float(re.search(r"^TreeLogLk\tML_Lengths2\t(\S+)", text, re.M).group(1)))
            out["n_ml_nni"] = np.int64(m.group(3))
        else:
            ll = []
        out["loglk"] = np.array(ll)
        if kind != "menni":
            out["rates"] = np.array([float(x) for x in re.search(r"^Rates((?: \S+)+)$", text, re.M).group(1).split()])
            out["ratecat"] = np.array([int(x) - 1 for x in re.search(r"^SiteCategories((?: \d+)+)$", text, re.M).group(1).split()], dtype=np.int32)
            mb = re.search(r"Bad splits: (\d+)/(\d+)", res2.stderr.decode(errors="replace"))
            out["bad_splits"] = np.array([int(mb.group(1)), int(mb.group(2))], dtype=np.int64)
        dst = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(dst, **out)
        print("%-24s %-6s NNI %3d SPR %2d  loglk %s  %6.1f KiB" % (name, kind, nni, nspr,
                                                                    " ".join("%.3f" % x for x in ll[-2:]), os.path.getsize(dst) / 1024.0))


THREADS_CASES = [
    # name, alphabet, flags, n_seq, n_pos, mu, gap, seed, threads.  The reference with `-threads T` (default -threads-level 3,
    # deterministic mode): NNI rounds and ML length rounds over the subtrees of treePartitioning, no star test in the serial
    # part of an ML NNI round - a different (still deterministic) schedule than one thread, a function of T.
    ("thr_menni_nt_400_t4", 4, ["-nt", "-noml"], 400, 150, 0.06, 0.02, 81, 4),             # ME NNIs + 2 SPR rounds
    ("thr_mllen_nt_300_t4", 4, ["-nt", "-nome", "-mllen"], 300, 120, 0.06, 0.02, 82, 4),   # ML lengths + CAT on the NJ topology
    ("thr_full_nt_400_t4", 4, ["-nt"], 400, 150, 0.06, 0.02, 81, 4),
    ("thr_full_nt_600_t8", 4, ["-nt"], 600, 120, 0.05, 0.02, 83, 8),
    ("thr_full_nt_300_double_t3", 4, ["-nt", "-double-precision"], 300, 150, 0.08, 0.03, 84, 3),
    # (`-gtr` with threads > 1 cannot be pinned: two runs of the reference fit different GTR rates - 0.9348 vs 0.9397 for ac on
    #  500 x 200 at -threads 4 - a race in its threaded setMLGtr path; Jukes-Cantor and the protein models are deterministic)
    ("thr_full_nt_500_t16", 4, ["-nt"], 500, 200, 0.05, 0.02, 85, 16),
    ("thr_full_nt_1500_t32", 4, ["-nt"], 1500, 100, 0.04, 0.01, 86, 32),
    # Matrix models (GTR, JTT / WAG / LG) with threads > 1: the reference is NOT reproducible - two runs of the same command give
    # different trees (observed at every -threads-level, 0 included; TreeLogLk differs by ~2e-5 relative after the first ML NNI
    # round: a race in its threaded likelihood code).  These fixtures keep the TreeLogLk lines of three runs; the test asks for
    # the north star's 1e-4 against them, and for this backend's own two runs to be identical.
    ("thrx_full_aa_300_lg_double_t8", 20, ["-lg", "-double-precision"], 300, 100, 0.10, 0.03, 87, 8),   # BASELINE C5's flags
    ("thrx_full_nt_500_gtr_t16", 4, ["-nt", "-gtr"], 500, 200, 0.05, 0.02, 85, 16),                     # BASELINE C2's flags
]


def gen_threads(tmp, only=None):
    """Black box, `-threads T`: every TreeLogLk line, the final tree without and with supports."""
    env = dict(os.environ, OMP_WAIT_POLICY="passive")
    for name, nc, flags, n, L, mu, gap, seed, threads in THREADS_CASES:
        if only and name not in only:
            continue
        codes = synth.random_descent_codes(n, L, nc, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT if nc == 4 else synth.ALPHABET_AA)
        log = os.path.join(tmp, name + ".log")
        base = [REFBIN] + flags + ["-threads", str(threads), "-seed", "1"]
        res = subprocess.run(base + ["-nosupport", "-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        text = open(log).read()
        ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", text, re.M)]
        if name.startswith("thrx_"):
            runs = [ll]
            trees = {res.stdout}
            for _ in range(2):
                r = subprocess.run(base + ["-nosupport", "-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
                runs.append([float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", open(log).read(), re.M)])
                trees.add(r.stdout)
            n_lines = min(len(r) for r in runs)
            np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), codes=codes, threads=np.int64(threads),
                                loglk_runs=np.array([r[:n_lines] for r in runs]), n_lines=np.array([len(r) for r in runs]),
                                distinct_trees=np.int64(len(trees)), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
            print("%-30s T=%2d  3 runs of the reference: %d distinct trees, final TreeLogLk %s" % (name, threads, len(trees), [r[-1] for r in runs]))
            continue
        res2 = subprocess.run(base + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        again = subprocess.run(base + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert again.stdout == res2.stdout, name + ": two runs of the reference differ"
        one = subprocess.run([REFBIN] + flags + ["-threads", "1", "-seed", "1", fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), codes=codes, threads=np.int64(threads), loglk=np.array(ll),
                            newick=np.frombuffer(res.stdout, dtype=np.uint8), newick_support=np.frombuffer(res2.stdout, dtype=np.uint8),
                            differs_from_one_thread=np.int64(one.stdout != res2.stdout),
                            flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        print("%-30s T=%2d  %2d TreeLogLk lines, final %.4f, %s the one-thread tree" % (name, threads, len(ll), ll[-1] if ll else 0.0,
                                                                                      "differs from" if one.stdout != res2.stdout else "EQUALS"))


def gen_threads_big(tmp, n=200000, threads=64):
    """The subtree schedule far beyond toy size: config C4's generator at n sequences, `VeryFastTree -nt -threads T -seed 1` (the whole
    default pipeline, Jukes-Cantor + CAT, SH-like supports; under Jukes-Cantor the reference's threaded runs are reproducible, see
    THREADS_CASES).  Kept: CRC-32 and length of the tree, the TreeLogLk lines -> thr_c4_<n/1000>k_t<T>_crc.npz.  The files are written under
    oracle/_ref/ first (`thrbigh:<n>:<T>` harvests a finished run)."""
    import time
    codes = synth.random_descent_codes(n, 200, 4, 0.02, 0.01, seed=4)
    tag = "thr_c4_%dk_t%d" % (n // 1000, threads)
    fa = os.path.join(HERE, "_ref", tag + ".fa")
    synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
    del codes
    env = dict(os.environ, OMP_WAIT_POLICY="passive")
    log = os.path.join(HERE, "_ref", tag + ".log")
    t0 = time.time()
    with open(os.path.join(HERE, "_ref", tag + ".tree"), "wb") as out, open(os.path.join(HERE, "_ref", tag + ".err"), "wb") as err:
        subprocess.run([REFBIN, "-nt", "-threads", str(threads), "-seed", "1", "-log", log, fa], check=True, stdout=out, stderr=err, env=env)
    open(os.path.join(HERE, "_ref", tag + ".wall"), "w").write("%.1f" % (time.time() - t0))
    os.remove(fa)
    harvest_threads_big(n, threads)


def harvest_threads_big(n, threads):
    import zlib
    tag = "thr_c4_%dk_t%d" % (n // 1000, threads)
    tree = open(os.path.join(HERE, "_ref", tag + ".tree"), "rb").read().decode().strip()
    assert tree.endswith(";"), "the reference has not finished"
    wall = float(open(os.path.join(HERE, "_ref", tag + ".wall")).read())
    ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", open(os.path.join(HERE, "_ref", tag + ".log")).read(), re.M)]
    flags = ["-nt", "-threads", str(threads), "-seed", "1"]
    # Does the reference's NJ phase at T threads give its one-thread NJ tree for this alignment?  (`VeryFastTree -nt -noml -nome -nosupport
    # -threads T` against `-threads 1`, both kept under oracle/_ref/ as <tag>_nj.tree / <tag>_nj_t1.tree; -1: not run.)  At 200 000 sequences
    # and 64 threads it does NOT (one split and 22 branch lengths differ: its threaded outProfile sums per-thread partials, NJ.tcc:763-783),
    # and a tree that starts from another NJ tree cannot be this backend's, whose NJ phase follows the one-thread order.
    nj_equal = -1
    pT, p1 = os.path.join(HERE, "_ref", tag + "_nj.tree"), os.path.join(HERE, "_ref", tag + "_nj_t1.tree")
    if os.path.exists(pT) and os.path.exists(p1):
        nj_equal = 1 if open(pT, "rb").read().strip() == open(p1, "rb").read().strip() else 0
    np.savez_compressed(os.path.join(GOLDEN, tag + "_crc.npz"), newick_crc=np.int64(zlib.crc32(tree.encode())), newick_bytes=np.int64(len(tree)),
                        reference_nj_equals_its_one_thread_nj=np.int64(nj_equal),
                        loglk=np.array(ll), threads=np.int64(threads), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(("random_descent_codes(%d, 200, 4, 0.02, 0.01, seed=4)" % n).encode(), dtype=np.uint8),
                        reference_wall_s=np.float64(wall))
    print("%s_crc: %d bytes of Newick, crc %d, %d TreeLogLk lines, %.0f s" % (tag, len(tree), zlib.crc32(tree.encode()), len(ll), wall))


def gen_c3_full(tmp):
    """Config C3's alignment (100 000 x 500 nt) through the COMPLETE default pipeline with its own flag: `VeryFastTree -nt -fastest -threads 1
    -seed 1` - top hits with the second-level lists in the NJ phase (the reference's one-thread `-fastest`), then ME NNIs + SPRs, ML NNIs
    under Jukes-Cantor + CAT, SH-like supports.  Kept: CRC-32 and length of the tree, the TreeLogLk lines -> bb_c3_full_crc.npz."""
    import time, zlib
    codes = synth.random_descent_codes(100000, 500, 4, 0.03, 0.01, seed=3)
    fa = os.path.join(HERE, "_ref", "c3_full.fa")
    synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
    del codes
    log = os.path.join(HERE, "_ref", "c3_full.log")
    flags = ["-nt", "-fastest", "-threads", "1", "-seed", "1"]
    t0 = time.time()
    with open(os.path.join(HERE, "_ref", "c3_full.tree"), "wb") as out, open(os.path.join(HERE, "_ref", "c3_full.err"), "wb") as err:
        subprocess.run([REFBIN] + flags + ["-log", log, fa], check=True, stdout=out, stderr=err)
    wall = time.time() - t0
    os.remove(fa)
    tree = open(os.path.join(HERE, "_ref", "c3_full.tree"), "rb").read().decode().strip()
    ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", open(log).read(), re.M)]
    np.savez_compressed(os.path.join(GOLDEN, "bb_c3_full_crc.npz"), newick_crc=np.int64(zlib.crc32(tree.encode())), newick_bytes=np.int64(len(tree)),
                        loglk=np.array(ll), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(b"random_descent_codes(100000, 500, 4, 0.03, 0.01, seed=3)", dtype=np.uint8),
                        reference_wall_s=np.float64(wall))
    print("bb_c3_full_crc: %d bytes of Newick, crc %d, %d TreeLogLk lines, %.0f s" % (len(tree), zlib.crc32(tree.encode()), len(ll), wall))


GAMMA_CASES = [
    # name, alphabet size, flags, n, L, mu, gap, seed
    ("gamma_nt_200", 4, ["-nt", "-gamma"], 200, 120, 0.05, 0.02, 21),
    ("gamma_nt_300_gtr", 4, ["-nt", "-gtr", "-gamma"], 300, 200, 0.05, 0.02, 22),
    ("gamma_aa_150_lg_double", 20, ["-lg", "-gamma", "-double-precision"], 150, 120, 0.08, 0.02, 23),
]


def gen_gamma(tmp):
    """Black box, `-gamma` at one thread: the complete default pipeline, then the Gamma(20) fit and the rescaled lengths
    (branchlengthScale, NJ.tcc:297-308).  Kept: the final tree with supports, every TreeLogLk line, the three numbers of the
    "Gamma(20) LogLk = .. alpha = .. rescaling lengths by .." line."""
    for name, nc, flags, n, L, mu, gap, seed in GAMMA_CASES:
        codes = synth.random_descent_codes(n, L, nc, mu, gap, seed)
        fa = os.path.join(tmp, name + ".fa")
        synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT if nc == 4 else synth.ALPHABET_AA)
        log = os.path.join(tmp, name + ".log")
        res = subprocess.run([REFBIN] + flags + ["-threads", "1", "-seed", "1", "-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(log).read()
        ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", text, re.M)]
        text += res.stderr.decode(errors="replace")
        m = re.search(r"Gamma\(20\) LogLk = (\S+) alpha = (\S+) rescaling lengths by (\S+)", text)
        assert m, name + ": no Gamma(20) line"
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), codes=codes, loglk=np.array(ll), newick=np.frombuffer(res.stdout, dtype=np.uint8),
                            gamma=np.array([float(m.group(1)), float(m.group(2)), float(m.group(3))]),
                            flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8))
        print("%-26s %d TreeLogLk lines, Gamma(20) LogLk %s alpha %s rescale %s" % (name, len(ll), m.group(1), m.group(2), m.group(3)))


def gen_c3(tmp):
    """BASELINE config C3 at full size (100 000 x 500 nt, `-nt -fastest` at one thread, i.e. with the second-level top
    hits): only the CRC-32 and length of the reference's `-noml -nome -nosupport` tree are kept (the tree is 2.5 MB).
    Takes ~11 minutes of one core."""
    import zlib
    codes = synth.random_descent_codes(100000, 500, 4, 0.03, 0.01, seed=3)
    fa = os.path.join(tmp, "c3.fa")
    synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
    flags = ["-nt", "-fastest", "-noml", "-nome", "-nosupport", "-threads", "1", "-seed", "1"]
    import time
    t0 = time.time()
    res = subprocess.run([REFBIN] + flags + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    tree = res.stdout.decode().strip()
    np.savez_compressed(os.path.join(GOLDEN, "bb_c3_crc.npz"), newick_crc=np.int64(zlib.crc32(tree.encode())),
                        newick_bytes=np.int64(len(tree)), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(b"random_descent_codes(100000, 500, 4, 0.03, 0.01, seed=3)", dtype=np.uint8),
                        reference_wall_s=np.float64(time.time() - t0))
    print("bb_c3_crc: %d bytes of Newick, crc %d" % (len(tree), zlib.crc32(tree.encode())))


def gen_c2(tmp):
    """BASELINE config C2 at full size (10 000 x 1 000 nt, `-nt -gtr`, one thread, the complete default pipeline with supports):
    CRC-32 and length of the reference's tree and its TreeLogLk lines - what bench.py's e2e_c2 compares its one-thread-order run with."""
    import time
    import zlib
    codes = synth.random_descent_codes(10000, 1000, 4, 0.03, 0.01, seed=2)
    fa = os.path.join(tmp, "c2.fa")
    synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
    del codes
    flags = ["-nt", "-gtr", "-threads", "1", "-seed", "1"]
    log = os.path.join(tmp, "c2.log")
    t0 = time.time()
    res = subprocess.run([REFBIN] + flags + ["-log", log, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.time() - t0
    tree = res.stdout.decode().strip()
    ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", open(log).read(), re.M)]
    np.savez_compressed(os.path.join(GOLDEN, "bb_c2_crc.npz"), newick_crc=np.int64(zlib.crc32(tree.encode())), newick_bytes=np.int64(len(tree)),
                        loglk=np.array(ll), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(b"random_descent_codes(10000, 1000, 4, 0.03, 0.01, seed=2)", dtype=np.uint8),
                        reference_wall_s=np.float64(wall))
    print("bb_c2_crc: %d bytes of Newick, crc %d, %d TreeLogLk lines, %.0f s" % (len(tree), zlib.crc32(tree.encode()), len(ll), wall))


C5_PARAMS = {
    # tag prefix -> (mu, gap, seed): "c5" is SURVEY.md 8(d)'s config C5; "c5mu03" are rounds 4-5's fixtures, made on C2's generator
    # parameters by mistake (less divergence, half the gaps) and kept as extra cases
    "c5": (0.08, 0.02, 5),
    "c5mu03": (0.03, 0.01, 2),
}


def c5_tag(n, kind="c5"):
    return kind if n == 50000 else "%s_%dk" % (kind, n // 1000)


def gen_c5(tmp, n=50000, kind="c5"):
    """BASELINE config C5 (50 000 aa x 300, mu = 0.08, g = 0.02, seed 5; `-lg -double-precision`, one thread, the complete default
    pipeline with supports), or the same generator at a smaller n: CRC-32 and length of the reference's tree and its TreeLogLk lines -
    what bench.py's e2e_c5 leg and tests/test_gpu_fullsize.py compare the one-thread-order run with.  bb_c5_crc.npz for n = 50 000,
    bb_c5_<n/1000>k_crc.npz otherwise (kind "c5mu03": the earlier rounds' parameters, bb_c5mu03_*).  Tens of minutes of one core at
    full size; the files are written under oracle/_ref/ first so that a run that outlives this script can be harvested (`c5h:<n>`)."""
    import time
    mu, gap, seed = C5_PARAMS[kind]
    codes = synth.random_descent_codes(n, 300, 20, mu, gap, seed=seed)
    tag = c5_tag(n, kind)
    fa = os.path.join(HERE, "_ref", tag + ".fa")
    synth.codes_to_fasta(codes, fa, synth.ALPHABET_AA)
    del codes
    flags = ["-lg", "-double-precision", "-threads", "1", "-seed", "1"]
    log = os.path.join(HERE, "_ref", tag + ".log")
    t0 = time.time()
    with open(os.path.join(HERE, "_ref", tag + ".tree"), "wb") as out, open(os.path.join(HERE, "_ref", tag + ".err"), "wb") as err:
        subprocess.run([REFBIN] + flags + ["-log", log, fa], check=True, stdout=out, stderr=err)
    open(os.path.join(HERE, "_ref", tag + ".wall"), "w").write("%.1f" % (time.time() - t0))
    os.remove(fa)
    harvest_c5(n, kind)


def harvest_c5(n, kind="c5"):
    import zlib
    mu, gap, seed = C5_PARAMS[kind]
    tag = c5_tag(n, kind)
    tree = open(os.path.join(HERE, "_ref", tag + ".tree"), "rb").read().decode().strip()
    assert tree.endswith(";"), "the reference has not finished"
    wall = float(open(os.path.join(HERE, "_ref", tag + ".wall")).read())
    ll = [float(x.group(1)) for x in re.finditer(r"^TreeLogLk\t\S+\t(\S+)", open(os.path.join(HERE, "_ref", tag + ".log")).read(), re.M)]
    flags = ["-lg", "-double-precision", "-threads", "1", "-seed", "1"]
    np.savez_compressed(os.path.join(GOLDEN, "bb_%s_crc.npz" % tag), newick_crc=np.int64(zlib.crc32(tree.encode())), newick_bytes=np.int64(len(tree)),
                        loglk=np.array(ll), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(("random_descent_codes(%d, 300, 20, %g, %g, seed=%d)" % (n, mu, gap, seed)).encode(), dtype=np.uint8),
                        reference_wall_s=np.float64(wall))
    print("bb_%s_crc: %d bytes of Newick, crc %d, %d TreeLogLk lines, %.0f s" % (tag, len(tree), zlib.crc32(tree.encode()), len(ll), wall))


def gen_c4(tmp):
    """BASELINE config C4 at full size (1 000 000 x 200 nt, `-nt` at one thread): CRC-32 and length of the reference's
    `-noml -nome -nosupport` tree, plus the join order as one CRC-32 per 10 000 `Join` lines (`i j new`, NJ.tcc:2993-3001)
    so that a divergence can be located.  The complete join list stays in oracle/_ref/c4_joins.npy (git-ignored).
    Takes hours of one core and tens of GB of memory."""
    import time
    import zlib
    codes = synth.random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)
    fa = os.path.join(tmp, "c4.fa")
    synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
    del codes
    flags = ["-nt", "-noml", "-nome", "-nosupport", "-threads", "1", "-seed", "1"]
    joins_txt = os.path.join(HERE, "_ref", "c4_joins.txt")
    tree_path = os.path.join(HERE, "_ref", "c4_nj.tree")
    t0 = time.time()
    with open(tree_path, "wb") as out:
        ref = subprocess.Popen([REFBIN] + flags + ["-verbose", "3", fa], stdout=out, stderr=subprocess.PIPE)
        with open(joins_txt, "wb") as jt:
            flt = subprocess.Popen(["grep", "^Join"], stdin=ref.stderr, stdout=jt)
            ref.stderr.close()
            rc = ref.wait()
            flt.wait()
    wall = time.time() - t0
    assert rc == 0, "reference exited with %d" % rc
    tree = open(tree_path, "rb").read().decode().strip()
    joins = []
    for line in open(joins_txt):
        f = line.split("\t")
        joins.append((int(f[1]), int(f[2]), int(f[10])))
    ja = np.array(joins, dtype=np.int64)
    np.save(os.path.join(HERE, "_ref", "c4_joins.npy"), ja.astype(np.int32))
    chunk = 10000
    crcs = np.array([zlib.crc32(ja[k:k + chunk].astype("<i4").tobytes()) for k in range(0, len(ja), chunk)], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLDEN, "bb_c4_crc.npz"), newick_crc=np.int64(zlib.crc32(tree.encode())),
                        newick_bytes=np.int64(len(tree)), n_joins=np.int64(len(ja)), join_chunk=np.int64(chunk),
                        join_chunk_crc=crcs, flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(b"random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)", dtype=np.uint8),
                        reference_wall_s=np.float64(wall))
    print("bb_c4_crc: %d bytes of Newick, crc %d, %d joins, %.0f s" % (len(tree), zlib.crc32(tree.encode()), len(ja), wall))


def gen_c4_tree(tmp):
    """bb_c4_t6.npz: the tree of `VeryFastTree -nt -noml -nome -nosupport -threads 6 -seed 1` for config C4's alignment (the reference
    binary itself, 12 200 s on six cores; oracle/_ref/c4_nj_t6.tree).  NOT a pin of the one-thread order: at a million sequences the
    reference's NJ phase depends on the thread count - 14 of the 594 643 joins its own one-thread run traced in round 3
    (oracle/_ref/c4_joins_t1_r03.txt) are not clades of this tree, the first one join 293 793 (tools/c4_clades.py,
    profiles/r04_c4_pin.txt; at 200 000 sequences the two thread counts still give the same splits and lengths, see gen_c4_scaled).
    Kept as a record: CRC and length of the 26 MB tree, and the per-10 000 CRCs of the one-thread `Join` trace (the same numbers as
    bb_c4_prefix.npz, from the finished file)."""
    import zlib
    tree = open(os.path.join(HERE, "_ref", "c4_nj_t6.tree"), "rb").read().decode().strip()
    assert tree.endswith(";") and tree.count(",") > 900000, "the reference has not finished"
    joins = []
    for line in open(os.path.join(HERE, "_ref", "c4_joins_t1_r03.txt")):
        f = line.rstrip("\n").split("\t")
        if len(f) < 11 or not line.endswith("\n"):
            break
        joins.append((int(f[1]), int(f[2]), int(f[10])))
    chunk = 10000
    ja = np.array(joins[:len(joins) // chunk * chunk], dtype=np.int64)
    crcs = np.array([zlib.crc32(ja[k:k + chunk].astype("<i4").tobytes()) for k in range(0, len(ja), chunk)], dtype=np.int64)
    flags = ["-nt", "-noml", "-nome", "-nosupport", "-threads", "6", "-seed", "1"]
    np.savez_compressed(os.path.join(GOLDEN, "bb_c4_t6.npz"), newick_crc=np.int64(zlib.crc32(tree.encode())), newick_bytes=np.int64(len(tree)),
                        n_joins=np.int64(len(ja)), join_chunk=np.int64(chunk), join_chunk_crc=crcs,
                        one_thread_traced_joins_not_in_this_tree=np.array([293793, 299441, 323356, 335439, 367072, 383275, 386516, 391329,
                                                                           401229, 428835, 439930, 487421, 527439, 545714], dtype=np.int64),
                        flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        join_trace_flags=np.frombuffer(b"-nt -noml -nome -nosupport -threads 1 -seed 1 -verbose 3", dtype=np.uint8),
                        alignment=np.frombuffer(b"random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)", dtype=np.uint8))
    print("bb_c4_t6: %d bytes of Newick, crc %d; join order over %d joins" % (len(tree), zlib.crc32(tree.encode()), len(ja)))
    # The pin proper: the ONE-thread run's tree, if a run was allowed to finish (`VeryFastTree -nt -noml -nome -nosupport -threads 1
    # -seed 1 c4.fa > oracle/_ref/c4_nj_t1.tree`, ~17 h; one was started detached at the end of round 4) -> bb_c4_crc.npz, which
    # bench.py's e2e_c4 compares with (identical_to_reference).
    t1 = os.path.join(HERE, "_ref", "c4_nj_t1.tree")
    if os.path.exists(t1) and os.path.getsize(t1) > 20000000:
        tree1 = open(t1, "rb").read().decode().strip()
        if tree1.endswith(";"):
            f1 = ["-nt", "-noml", "-nome", "-nosupport", "-threads", "1", "-seed", "1"]
            # (round 6's run is oracle/_ref/njtrace - the reference's sources with the `Join` lines switched on, oracle/njtrace.cpp - which
            #  prints the same tree on stdout and the trace on stderr; its last progress line is the NJ phase's wall-clock)
            wall = 0.0
            trace = os.path.join(HERE, "_ref", "c4_joins_t1_r05.txt")
            if os.path.exists(trace):
                with open(trace, "rb") as fh:
                    fh.seek(max(0, os.path.getsize(trace) - 4096))
                    for line in fh.read().decode(errors="replace").splitlines():
                        if "seconds: Joined" in line:
                            wall = float(line.split()[0])
            np.savez_compressed(os.path.join(GOLDEN, "bb_c4_crc.npz"), newick_crc=np.int64(zlib.crc32(tree1.encode())), newick_bytes=np.int64(len(tree1)),
                                flags=np.frombuffer(" ".join(f1).encode(), dtype=np.uint8), reference_wall_s=np.float64(wall),
                                alignment=np.frombuffer(b"random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)", dtype=np.uint8))
            print("bb_c4_crc: %d bytes of Newick, crc %d (the one-thread reference, %.0f s)" % (len(tree1), zlib.crc32(tree1.encode()), wall))


def gen_c4_scaled(tmp, sizes=(200000, 400000)):
    """Config C4's generator and flags at sizes the one-thread reference finishes in an hour or two (200 000: 2 080 s; 400 000:
    see reference_wall_s): `VeryFastTree -nt -noml -nome -nosupport -threads 1 -seed 1`, CRC and length of the tree
    (bb_c4_200k_crc.npz, bb_c4_400k_crc.npz).  A finished tree under oracle/_ref/c4_<n>k_t1.tree is reused (its wall-clock then comes
    from the log line next to it, or 0)."""
    import time
    import zlib
    for n in sizes:
        keep = os.path.join(HERE, "_ref", "c4_%dk_t1.tree" % (n // 1000))
        flags = ["-nt", "-noml", "-nome", "-nosupport", "-threads", "1", "-seed", "1"]
        wall = 0.0
        if os.path.exists(keep) and os.path.getsize(keep) > 0:
            tree = open(keep).read().strip()
            if os.path.exists(keep + ".wall"):
                wall = float(open(keep + ".wall").read())
        else:
            if n > 200000 and "c4_scaled_all" not in sys.argv:
                print("c4 at %d sequences: no finished tree under oracle/_ref (add c4_scaled_all to run the reference: more than an hour)" % n)
                continue
            codes = synth.random_descent_codes(n, 200, 4, 0.02, 0.01, seed=4)
            fa = os.path.join(tmp, "c4s.fa")
            synth.codes_to_fasta(codes, fa, synth.ALPHABET_NT)
            t0 = time.time()
            res = subprocess.run([REFBIN] + flags + [fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.time() - t0
            tree = res.stdout.decode().strip()
            open(keep, "w").write(tree + "\n")
            open(keep + ".wall", "w").write("%.1f" % wall)
        assert tree.endswith(";")
        np.savez_compressed(os.path.join(GOLDEN, "bb_c4_%dk_crc.npz" % (n // 1000)), newick_crc=np.int64(zlib.crc32(tree.encode())),
                            newick_bytes=np.int64(len(tree)), flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                            alignment=np.frombuffer(("random_descent_codes(%d, 200, 4, 0.02, 0.01, seed=4)" % n).encode(), dtype=np.uint8),
                            reference_wall_s=np.float64(wall))
        print("bb_c4_%dk_crc: %d bytes of Newick, crc %d" % (n // 1000, len(tree), zlib.crc32(tree.encode())))


def gen_c4_prefix(tmp):
    """The part of gen_c4's run that exists so far (oracle/_ref/c4_joins.txt grows while the reference works: the 1M-taxa NJ
    phase takes it more than half a day at one thread): CRC-32 per 10 000 complete `Join` lines.  Pins the join order of
    config C4 up to the join the reference had reached when this was written."""
    import zlib
    # the longest trace there is: round 3's (c4_joins.txt, the reference binary at -verbose 3) or a later run of oracle/njtrace at one
    # thread (c4_joins_t1_r05.txt: its log keeps the progress lines too) - the same joins line for line as far as both go
    best = []
    for name in ("c4_joins.txt", "c4_joins_t1_r05.txt"):
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            continue
        joins = []
        for line in open(path):
            if not line.startswith("Join"):
                continue
            f = line.rstrip("\n").split("\t")
            if len(f) < 11 or not line.endswith("\n"):
                break
            joins.append((int(f[1]), int(f[2]), int(f[10])))
        if best and joins[:min(len(best), len(joins))] != best[:min(len(best), len(joins))]:
            raise SystemExit("the two traces of the one-thread reference disagree: %s" % name)
        if len(joins) > len(best):
            best = joins
    joins = best
    chunk = 10000
    # a FINISHED trace (the tree of the same run is there) keeps its last, shorter chunk too: the join order to the last join
    t1 = os.path.join(HERE, "_ref", "c4_nj_t1.tree")
    complete = os.path.exists(t1) and os.path.getsize(t1) > 20000000 and open(t1, "rb").read().rstrip().endswith(b";")
    ja = np.array(joins if complete else joins[:len(joins) // chunk * chunk], dtype=np.int64)
    crcs = np.array([zlib.crc32(ja[k:k + chunk].astype("<i4").tobytes()) for k in range(0, len(ja), chunk)], dtype=np.int64)
    flags = ["-nt", "-noml", "-nome", "-nosupport", "-threads", "1", "-seed", "1"]
    np.savez_compressed(os.path.join(GOLDEN, "bb_c4_prefix.npz"), n_joins=np.int64(len(ja)), complete=np.int64(1 if complete else 0), join_chunk=np.int64(chunk), join_chunk_crc=crcs,
                        flags=np.frombuffer(" ".join(flags).encode(), dtype=np.uint8),
                        alignment=np.frombuffer(b"random_descent_codes(1000000, 200, 4, 0.02, 0.01, seed=4)", dtype=np.uint8))
    print("bb_c4_prefix: %d joins in %d chunks" % (len(ja), len(crcs)))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    which = sys.argv[1:] or ["whitebox", "blackbox", "knuth", "tables", "mllen", "menni", "mlnni", "aa", "threads", "partition", "gamma"]
    with tempfile.TemporaryDirectory() as tmp:
        if "whitebox" in which:
            gen_whitebox(tmp)
        if "blackbox" in which:
            gen_blackbox(tmp)
        if any(w.startswith("bb_") for w in which):
            gen_blackbox(tmp, [w for w in which if w.startswith("bb_")])
        if "partition" in which:
            gen_partition(tmp)
        if "knuth" in which:
            gen_knuth(tmp)
        if "tables" in which:
            gen_tables(tmp)
        if "mllen" in which:
            gen_mllen(tmp)
        if "menni" in which:
            gen_menni(tmp)
        if "mlnni" in which:
            gen_mlnni(tmp)
        if any(w.startswith("mlnni:") for w in which):
            gen_mlnni(tmp, [w[6:] for w in which if w.startswith("mlnni:")])
        if "aa" in which:
            gen_aa(tmp)
        if any(w.startswith("aa:") for w in which):
            gen_aa(tmp, [w[3:] for w in which if w.startswith("aa:")])
        if "threads" in which:
            gen_threads(tmp)
        if any(w.startswith("threads:") for w in which):
            gen_threads(tmp, [w[8:] for w in which if w.startswith("threads:")])
        if "gamma" in which:
            gen_gamma(tmp)
        if "c3" in which:   # not part of the default set: ~11 minutes
            gen_c3(tmp)
        if "c3_full" in which:   # not part of the default set: an hour of one core
            gen_c3_full(tmp)
        if "c2" in which:   # not part of the default set: minutes
            gen_c2(tmp)
        for w in which:     # `c5:<n>`: config C5's generator at n sequences (50000 = the config itself, hours); `c5h:<n>` harvests a finished run
            if w.startswith("c5:"):
                gen_c5(tmp, int(w[3:]))
            if w.startswith("c5h:"):
                harvest_c5(int(w[4:]))
            if w.startswith("thrbig:"):      # `thrbig:<n>:<T>`: the threaded schedule at config C4's generator (an hour or more)
                gen_threads_big(tmp, *[int(x) for x in w.split(":")[1:]])
            if w.startswith("thrbigh:"):
                harvest_threads_big(*[int(x) for x in w.split(":")[1:]])
            if w.startswith("c5mu03:"):
                gen_c5(tmp, int(w[7:]), "c5mu03")
            if w.startswith("c5mu03h:"):
                harvest_c5(int(w[8:]), "c5mu03")
        if "c4" in which:   # not part of the default set: hours
            gen_c4(tmp)
        if "c4_tree" in which:   # from the files a finished reference run left under oracle/_ref/
            gen_c4_tree(tmp)
        if "c4_scaled" in which or "c4_scaled_all" in which:   # 35 minutes of one core (200 000), more with c4_scaled_all
            gen_c4_scaled(tmp)
        if "c4_prefix" in which:   # from the Join lines a running gen_c4 has produced so far
            gen_c4_prefix(tmp)


if __name__ == "__main__":
    main()
