"""What one hqq_hip_exchange kernel costs on ONE device (no xGMI): world = 1 (store to self, flag, wait satisfied at once) and eight ranks
of one process with the waits disabled (spin_limit 1: eight workgroups, every store local).  hipGraph of 200 kernels alternating between
the four exchange points of a Llama-2-70B block, us per kernel.  A floor for the kernel itself, not a multi-GPU measurement."""
import sys, torch
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else ".")
from hqq_amd.shard import PeerExchange
points = [[8192, 1024, 1024], [8192], [28672, 28672], [8192]]
for world, spin in ((1, 0), (8, 1)):
    grp = PeerExchange.local_group(points, 4, torch.float16, "cuda", world, spin_limit=spin)
    px = grp[0]
    ys = [[torch.randn(1, N // world, device="cuda").half() for N in pt] for pt in points]
    def step():
        for _ in range(50):
            for e in range(4):
                px.run(e, ys[e])
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    print(f"world={world} ({'real wait, satisfied by the own flag' if spin == 0 else 'waits disabled'}): {best:.2f} us per exchange kernel (dependent, graph replay)")
