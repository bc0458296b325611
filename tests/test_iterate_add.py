"""vft_iterate_add (veryfasttree_amd/csrc/vft_iterate_add.h): the closed form of outProfile's weight chain over the leaves
(numeric_t weight; repeat count times: weight = (numeric_t)((double) weight + inweight), NJ.tcc:738-745) must equal the
loop bit for bit - float and double, constants 1/n, constants that tie in many binades, counts up to two million."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_closed_form_equals_the_loop(tmp_path):
    exe = str(tmp_path / "iacheck")
    subprocess.run(["g++", "-O2", "-std=c++11", "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "iterate_add_check.cpp"),
                    "-o", exe], check=True)
    for seed in ("1", "2"):
        res = subprocess.run([exe, "1500", "2000000", seed], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert res.returncode == 0 and res.stdout.decode().strip().endswith("mismatches 0"), res.stdout.decode()[-2000:]
