"""VALU instructions per element-iteration of the HQ solver's fast path, read off the ISA of the committed build
(hipcc -S of hqq_amd/csrc/quantize.hip, kernel solve_kernel<half, 8>: group_size 64, 8 lanes per group, 8 elements per lane).
Writes profiles/solver_valu.json, which bench.py turns into `quantize.layers[*].valu_frac`.

Fast path = what practically every wave runs: the double-precision pow of shrink_lp_op is skipped unless a lane's
|e| reaches 0.9 a*; the blocks that hold it (v_*_f64) are not on the path and not counted."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "hqq_amd", "csrc", "quantize.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "q.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
m = re.search(r'^(_ZN3hqq12solve_kernelIDF16_Li8EE\S*):', s, re.M)
b = s[m.end():]
b = b[:b.index('s_endpgm')]
blocks, cur = [], ['entry', []]
for l in b.split('\n'):
    l = l.strip()
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        blocks.append(cur); cur = [mm.group(1), []]; continue
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    cur[1].append(l.split(';')[0].strip())
blocks.append(cur)
valu = lambda ins: [i for i in ins if i.startswith('v_')]
elem = [i for i, (n, ins) in enumerate(blocks) if any(x.startswith('v_rndne_f32') for x in ins) and not any('f64' in x for x in ins)]
# the loop: from the first block that rounds a level to the block that branches back to it; the pow blocks (f64) are off the fast path
idx = {n: i for i, (n, _) in enumerate(blocks)}
back = [(i, idx[x.split()[-1]]) for i, (n, ins) in enumerate(blocks) for x in ins
        if (x.startswith('s_cbranch') or x.startswith('s_branch')) and x.split()[-1] in idx and idx[x.split()[-1]] <= i]
# the iteration loop is the backward branch whose span holds the eight per-element blocks (each rounds one level)
i_b, t = max(back, key=lambda bt: sum(1 for e in elem if bt[1] <= e <= bt[0]))
on_path = [(n, ins) for (n, ins) in blocks[t:i_b + 1] if not any('f64' in x for x in ins)]
# inside the loop: the latch block, and the blocks that round a level (one per element; the last also holds the row sum).  The blocks
# between them are the pow and its tail (clamp / sign / multiply), reached only when a lane's |e| asks for the pow: off the fast path
counted = [(n, ins) for k, (n, ins) in enumerate(on_path) if k == 0 or k == len(on_path) - 1 or any(x.startswith("v_rndne_f32") for x in ins)]   # + the row-sum block that closes the loop
nv = sum(len(valu(ins)) for _, ins in counted)
trans = sum(1 for _, ins in counted for x in valu(ins) if x.split()[0].startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_")))
EPL = 8
res = {"kernel": m.group(1), "elements_per_lane_iteration": EPL, "valu_instr_per_lane_iteration": nv, "valu_instr_per_element_iteration": round(nv / EPL, 2),
       "transcendental_per_element_iteration": round(trans / EPL, 2),
       # one VALU wave-instruction occupies a SIMD for 4 cycles (quarter-rate ones — v_rcp_f32 of the IEEE division — for 16)
       "issue_cycles_per_element_iteration": round(((nv - trans) * 4 + trans * 16) / EPL / 64, 5),
       "blocks_counted": [n for n, _ in counted],
       "note": "fast path of solve_kernel<half, 8> (pow blocks and their tails excluded); peak element-iterations/s = 1024 SIMDs x 2.4e9 Hz / issue_cycles_per_element_iteration"}
res["peak_G_element_iters_per_s"] = round(1024 * 2.4e9 / res["issue_cycles_per_element_iteration"] / 1e9, 1)
json.dump(res, open(os.path.join(ROOT, "profiles", "solver_valu.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
