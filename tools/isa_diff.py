#!/usr/bin/env python3
"""compare two llvm-objdump disassemblies kernel by kernel, ignoring addresses / encodings:  tools/isa_diff.py a.s b.s"""
import re, sys
def load(p):
    ks, cur = {}, None
    for l in open(p):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            cur = m.group(1); ks[cur] = []; continue
        if cur is None or not l.strip():
            continue
        l = re.sub(r"//.*$", "", l).strip()
        l = re.sub(r"<[^>]+>", "", l)          # branch target symbols
        ks[cur].append(l)
    return ks
a, b = load(sys.argv[1]), load(sys.argv[2])
same = diff = 0
for k in sorted(set(a) | set(b)):
    if a.get(k) == b.get(k):
        same += 1
    else:
        diff += 1
        print("DIFF", k[:120], len(a.get(k, [])), len(b.get(k, [])))
print(f"{same} kernels identical, {diff} differ")
