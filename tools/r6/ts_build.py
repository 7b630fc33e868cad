#!/usr/bin/env python3
"""Lab: an instrumented copy of the decode kernel (time stamps per wave) WITHOUT touching hqq_amd/csrc: the kernel text is copied to tools/r6/ts/, stamps are
spliced in at anchor lines, and a library tools/r6/libhqq_hip_ts.so is built whose gemv kernels write 8 x s_memrealtime (100 MHz) per wave into a buffer set with
hqq_lab_set_ts(ptr).   python tools/r6/ts_build.py"""
import os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(root, "hqq_amd", "csrc"), os.path.join(root, "tools", "r6", "ts")
shutil.rmtree(dst, ignore_errors=True); os.makedirs(dst)
for f in os.listdir(src):
    if f.endswith((".h", ".inc", ".hip")): shutil.copy(os.path.join(src, f), dst)
# the public header is included as ../../include/hqq_hip.h from csrc: same depth from tools/r6/ts? no -> fix the include path
p = os.path.join(dst, "hqq_common.h"); s = open(p).read().replace('#include "../../include/hqq_hip.h"', '#include "../../../include/hqq_hip.h"'); open(p, "w").write(s)
k = open(os.path.join(dst, "gemv_kernel.inc")).read()
def splice(text, anchor, add, after=True, count=1):
    i = text.index(anchor)
    j = text.index("\n", i) + 1 if after else i
    return text[:j] + add + "\n" + text[j:]
k = splice(k, "  const GvIn a = GV_IN_PACK;", "  unsigned long long t_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n_cons = 0; t_[0] = __builtin_amdgcn_s_memrealtime();\n#define TS(i) do { if (t_[i] == 0) t_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)")
k = splice(k, "  issue(ring[0], lc, cp, cu, live0);", "  TS(1);")
k = k.replace("  __syncthreads();\n\n  // FACTORED: one fp32 partial per", "  TS(2);\n  __syncthreads();\n  TS(3);\n\n  // FACTORED: one fp32 partial per")
k = k.replace("        if (rp[f] < total) consume(ring[f], rp[f], ru[f]);", "        if (rp[f] < total) { TS(4); consume(ring[f], rp[f], ru[f]); TS(5); ++n_cons; t_[6] = __builtin_amdgcn_s_memrealtime(); }")
# the end: after the loop
k = k.replace("    } while (more);\n  }\n}", "    } while (more);\n  }\n  t_[7] = __builtin_amdgcn_s_memrealtime();\n  if (lane == 0 && g_lab_ts_dev) { unsigned long long* q = g_lab_ts_dev + (static_cast<size_t>(blockIdx.x) * WPG + wave) * 11; for (int i = 0; i < 8; ++i) q[i] = t_[i]; q[8] = n_cons; q[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15; q[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4); }\n}")
open(os.path.join(dst, "gemv_kernel.inc"), "w").write(k)
g = open(os.path.join(dst, "gemv.hip")).read()
g = g.replace('#include "w3s.h"\n', '#include "w3s.h"\n__device__ unsigned long long* g_lab_ts_dev = nullptr;\nextern "C" int hqq_lab_set_ts(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_ts_dev), &p, sizeof(p)); }\n', 1)
open(os.path.join(dst, "gemv.hip"), "w").write(g)
obj = os.path.join(dst, "gemv.o")
name = sys.argv[1] if len(sys.argv) > 1 else "ts"
preload = sys.argv[2] if len(sys.argv) > 2 else "16"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm",
                       "-amdgpu-kernarg-preload-count=" + preload, "-fno-slp-vectorize"] + sys.argv[3:] + ["-c", os.path.join(dst, "gemv.hip"), "-o", obj], stderr=subprocess.DEVNULL)
objs = [os.path.join(src, "build", f) for f in os.listdir(os.path.join(src, "build")) if f.endswith(".o") and f != "gemv.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "tools", "r6", "libhqq_hip_%s.so" % name)] + objs + [obj])
print("built tools/r6/libhqq_hip_%s.so" % name)
