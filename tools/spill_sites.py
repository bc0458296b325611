#!/usr/bin/env python3
"""Where a line-search kernel's scratch traffic sits: compiles ONE instance of k_ml_quartet to assembly (hipcc -S, the library's flags)
and attributes every scratch load / store to the innermost loop that holds it (the assembler comments name each block's loop and depth).
The evaluation loop of Brent's minimiser is the deepest large loop; spills outside it cost a few stores per search, not per evaluation.
usage: spill_sites.py REAL NC CPT QUAD   e.g.  spill_sites.py float 4 4 false   |   spill_sites.py double 20 5 true"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
real, nc, cpt, quad = sys.argv[1:5]
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "q.hip")
    open(src, "w").write('#include <hip/hip_runtime.h>\n#include <cstdint>\n#include "vft_layout.h"\n#include "vft_device.h"\n#include "vft_kernels_profile.h"\n'
                         '#include "vft_kernels_ml.h"\nVFT_ML_QUARTET_INSTANCE(, %s, %s, %s, %s)\n' % (real, nc, cpt, quad))
    asm = os.path.join(tmp, "q.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "veryfasttree_amd", "csrc"),
                    "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z12k_ml_quartet\w+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
cur, scratch, size = ("outside any loop", 0), Counter(), Counter()
for i in range(start, end):
    l = lines[i]
    if re.match(r"^\.LBB\d+_\d+:", l):
        m2, m3 = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l), None
        if m2:
            cur = (m2.group(1), int(m2.group(2)))
        else:
            cur = ("outside any loop", 0)
            for j in range(i, min(i + 8, end)):
                m3 = re.search(r"Loop Header: Depth=(\d+)", lines[j])
                if m3:
                    cur = (l.split(":")[0].lstrip(".L"), int(m3.group(1)))
                    break
                if j > i and not lines[j].strip().startswith(";"):
                    break
    if l.startswith("\t") and not l.strip().startswith((".", ";")):
        size[cur] += 1
        if "scratch_" in l:
            scratch[cur] += 1
meta = "\n".join(lines[end:])
meta = meta[meta.index(".name:           _Z12k_ml_quartet") - 3000:]   # (the kernel's own metadata block: its fields precede and follow .name)
blk = re.split(r"\n  - \.agpr_count:", meta)
blk = next(b for b in blk if "_Z12k_ml_quartet" in b)
g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, ".agpr_count:" + blk) or re.search(r"(0)", "0")).group(1)
print("k_ml_quartet<%s, %s, %s, %s>: %s VGPRs + %s AGPRs, %s spilled VGPRs, %s bytes of scratch per lane, %d instructions"
      % (real, nc, cpt, quad, g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), sum(size.values())))
print("%-22s %5s %12s %14s" % ("loop (header block)", "depth", "instructions", "scratch ld/st"))
for k in sorted(size, key=lambda k: (-scratch[k], -size[k])):
    if scratch[k] or size[k] > 1000:
        print("%-22s %5d %12d %14d" % (k[0], k[1], size[k], scratch[k]))
by_depth = Counter()
for k, v in scratch.items():
    by_depth[k[1]] += v
print("scratch instructions by loop depth:", ", ".join("depth %d: %d" % (d, by_depth[d]) for d in sorted(by_depth)))
