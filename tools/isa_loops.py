#!/usr/bin/env python3
"""Loops of one kernel of the built library, by instruction class - what a column of a sweep costs in issue slots.

    python tools/isa_loops.py <mangled-name substring> [object]     (default object: build/obj/vft_api.hip.o)

Disassembles the gfx950 code object (llvm-objdump --offloading + -d), finds the kernel, and for every backward branch prints the loop
body's size: VALU / SALU / SMEM / VMEM / LDS instructions and waits.  Nested loops are listed inner first."""
import collections, os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
name = sys.argv[1]
obj = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "build", "obj", "vft_api.hip.o")
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(obj, os.path.join(tmp, "o.o"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "o.o"], check=True, stdout=subprocess.DEVNULL, cwd=tmp)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], check=True, stdout=subprocess.PIPE).stdout.decode().split("\n")
starts = [i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <", l)]
for si, i in enumerate(starts):
    if name not in text[i]:
        continue
    body = text[i + 1:starts[si + 1] if si + 1 < len(starts) else len(text)]
    ins = []
    for l in body:
        if "//" not in l:
            continue
        t, c = l.split("//", 1)
        ins.append((int(c.split(":")[0].strip(), 16), t.strip()))
    by = {a: k for k, (a, _) in enumerate(ins)}
    print(text[i].split("<")[1][:100], "-", len(ins), "instructions")
    def cls(x):
        op = x.split()[0]
        if op.startswith("s_waitcnt"): return "wait"
        if op.startswith("v_"): return "VALU"
        if op.startswith("s_load") or op.startswith("s_buffer_load"): return "SMEM"
        if op.startswith("s_"): return "SALU"
        if op.startswith("ds_"): return "LDS"
        if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "VMEM"
        return "other"
    loops = []
    for k, (a, t) in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\d+)", t)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + off * 4
        if tgt <= a and tgt in by:
            loops.append((by[tgt], k))
    for j, k in sorted(loops, key=lambda p: p[1] - p[0]):
        c = collections.Counter(cls(x) for _, x in ins[j:k + 1])
        ops = collections.Counter(x.split()[0] for _, x in ins[j:k + 1] if x.startswith("v_"))
        print("  loop %x..%x: %5d instructions  %s" % (ins[j][0], ins[k][0], k - j + 1, dict(c)))
        print("      VALU by opcode:", ", ".join("%s %d" % p for p in ops.most_common(14)))
