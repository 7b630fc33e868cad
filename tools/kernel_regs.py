#!/usr/bin/env python3
"""Register / scratch / LDS usage of every gfx950 kernel in a hipcc object:  tools/kernel_regs.py <file.o> [filter]"""
import os, re, subprocess, sys, tempfile
o = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = tempfile.mkdtemp()
os.system(f"objcopy -O binary --only-section=.hip_fatbin {o} {d}/fb.bin && /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o "
          f"--targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input={d}/fb.bin --output={d}/co.co --unbundle")
out = subprocess.run(f"/opt/rocm/lib/llvm/bin/llvm-readelf --notes {d}/co.co", shell=True, capture_output=True, text=True).stdout
blocks = out.split("- .agpr_count:")[1:]
for b in blocks:
    b = ".agpr_count:" + b
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", b) or [None, "?"])[1]
    name = g("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size')}")
