#!/usr/bin/env python3
"""Per-launch durations of the 7B stack's four launches (q|k|v, o, gate|up, down) from a rocprofv3 kernel trace:
   tools/r6/per_launch.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, statistics as st
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "gemv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the steady part: the last 128 * 20 launches
rows = rows[-128 * 20:]
names = ["q|k|v", "o", "gate|up", "down"]
dur = {n: [] for n in names}
gap = {n: [] for n in names}
for i, r in enumerate(rows):
    n = names[i % 4]
    dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if i:
        gap[n].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
mb = {"q|k|v": 28.36, "o": 9.45, "gate|up": 50.79, "down": 25.39}
for n in names:
    d, g = st.mean(dur[n]), st.mean(gap[n])
    print(f"{n:8s} kernel {d:6.2f} us  (min {min(dur[n]):5.2f})  gap before {g:5.2f} us   {mb[n]:5.2f} MB  -> {mb[n] / d:5.2f} TB/s in-kernel, {mb[n] / (d + g):5.2f} with the gap   grid {rows[names.index(n)]['Grid_Size_X'] if 'Grid_Size_X' in rows[0] else ''} wg {rows[names.index(n)].get('Workgroup_Size_X','')}")
import collections
for n in names:
    h = collections.Counter(int(d * 4) / 4 for d in dur[n])
    print(n, "histogram (us: count)", " ".join(f"{k:.2f}:{v}" for k, v in sorted(h.items())))
# by position in the step (block index) for o
nb = len(dur["o"]) // 20
for n in ("o", "down"):
    per_block = [st.mean(dur[n][b::32]) for b in range(32)]
    print(n, "mean by block index", " ".join(f"{x:.2f}" for x in per_block))
tot = sum(st.mean(dur[n]) + st.mean(gap[n]) for n in names) * 32
print(f"sum over 32 blocks: {tot / 1e3:.4f} ms")
