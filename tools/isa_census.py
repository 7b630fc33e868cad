#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S file:  tools/isa_census.py file.s <substring of the mangled name> [--dump out.s]
Prints totals and the counts inside the streaming loop (last depth-1 loop header to the end of the kernel) by class."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
starts = [i for i, l in enumerate(src) if l.startswith("_Z") and key in l and ": " in l]
if not starts:
    sys.exit("kernel not found")
st = starts[0]
en = next(i for i in range(st, len(src)) if src[i].startswith(".Lfunc_end"))
body = src[st:en]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"
def census(lines):
    c = {}
    for l in lines:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        k = cls(op)
        c[k] = c.get(k, 0) + 1
    return dict(sorted(c.items()))
hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and "=>This" in l]
print("kernel lines", len(body), "total", census(body))
meta = "\n".join(src[en:en + 80])
for k in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "SGPRSpill", "VGPRSpill"):
    m = re.search(rf"; {k}[^\n]*", meta)
    if m: print(m.group(0))
if hdr:
    lo = hdr[-1]
    print("streaming loop:", census(body[lo:]))
