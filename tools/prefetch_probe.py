#!/usr/bin/env python3
"""Does pulling the NEXT launch's weights through the memory-side cache on a side stream, while the current launch streams, shorten a chain of
dependent GEMV launches?  (development aid; needs an MI355X)   python tools/prefetch_probe.py [N K] [workgroups]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops  # noqa: E402

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
WG = int(sys.argv[3]) if len(sys.argv) > 3 else 64
NL = 64 if N * K <= 4096 * 4096 else 24
g = torch.Generator().manual_seed(0)
R = N * K // 64
layers = []
for i in range(NL):
    U = torch.randint(0, 16, (R, 64), generator=g, dtype=torch.uint8).cuda()
    P = ops.pack(4, U)
    s = (torch.rand(R, 1, generator=g) * 0.004 + 0.001).half().cuda()
    z = (torch.rand(R, 1, generator=g) * 15).round().half().cuda()
    layers.append((P, s, z))
opts = ops.OPT_META_SCALABLE
x = torch.randn(1, K, generator=g).half().cuda()
y = torch.empty(1, N, dtype=torch.float16, device="cuda")
side = torch.cuda.Stream()


def chain(prefetch, same=False):
    main = torch.cuda.current_stream()
    for i in range(NL):
        P, s, z = layers[0 if same else i]
        if prefetch and i + 1 < NL:
            side.wait_stream(main)                      # after launch i - 1 (everything enqueued so far)
            with torch.cuda.stream(side):
                ops.prefetch(layers[i + 1][0], WG)
        ops.gemv(x, P, s, z, None, N, K, 64, 4, out=y, opts=opts)
    if prefetch:
        main.wait_stream(side)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            fn()
    torch.cuda.synchronize()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / NL * 1e3


print(f"{N}x{K} int4, {NL} distinct layers ({NL * N * K / 2 / 1e6:.0f} MB), us per launch:")
print("  distinct layers, no prefetch     ", round(timed(lambda: chain(False)), 2))
print("  same layer every launch (cached) ", round(timed(lambda: chain(False, same=True)), 2))
print(f"  distinct layers, prefetch ({WG} wg)", round(timed(lambda: chain(True)), 2))
