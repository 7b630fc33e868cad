import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from hqq_amd import ops
from tools.microbench import rand_layer
M, N, K = 8192, 4096, 4096
Wq, s, z = rand_layer(N, K, 4)
x = torch.randn(M, K, device="cuda", dtype=torch.float16)
y = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(3):
    ops.gemm(x, Wq, s, z, None, N, K, 64, 4, out=y)
torch.cuda.synchronize()
