import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqq_amd import ops
nbits=4
dims=[(1024, [512, 1024]), (1024, [2048]), (2048, [1024])]
def qlayer(N,K,seed,cpu_gen):
    if cpu_gen:
        W = (torch.randn(N, K, generator=torch.Generator().manual_seed(seed)) * (1.0 / K ** 0.5)).half().cuda()
    else:
        g = torch.Generator(device="cuda").manual_seed(seed)
        W = (torch.randn(N, K, device="cuda", generator=g) * (1.0 / K ** 0.5)).half()
    Wq, s, z = ops.quantize(W, nbits=nbits, group_size=64, round_zero=True)
    return Wq, s.half(), z.half()
for cpu_gen in (False, True):
  for nruns in (1, 3):
    torch.manual_seed(1)
    x0 = torch.randn(1, 1024, device="cuda").half()
    stages, layers, x = [], [], x0
    for si,(K,Ns) in enumerate(dims):
        Ls=[]
        for j,N in enumerate(Ns):
            Wq,s,z = qlayer(N,K,100*si+j,cpu_gen)
            y = torch.full((1,N), float("nan"), device="cuda", dtype=torch.float16)
            Ls.append((Wq,s,z,None,N,y))
        stages.append((x,Ls)); layers.append(Ls); x = Ls[-1][5]
    plan = ops.DecodePlan(stages, nbits, opts=0, grid=8)
    for r in range(nruns):
        plan.run()
        torch.cuda.synchronize()
        print(f"cpu_gen={cpu_gen} nruns={nruns} run {r}: status={plan.status()} nan counts:", [[int(torch.isnan(L[5]).sum()) for L in Ls] for Ls in layers], flush=True)
print("NaN positions stage0 layer0:", torch.nonzero(torch.isnan(layers[0][0][5][0])).flatten().tolist())
print("NaN positions stage0 layer1:", torch.nonzero(torch.isnan(layers[0][1][5][0])).flatten().tolist())
print("NaN positions stage1:", torch.nonzero(torch.isnan(layers[1][0][5][0])).flatten().tolist())
