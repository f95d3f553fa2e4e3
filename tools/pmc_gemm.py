"""A few launches of NT GEMM shapes so rocprofv3 --pmc can attribute counters per kernel (k256 vs k128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
T, H = 17536, 4096
for M, N, K in [(T, H, H), (3 * H, H, T)]:
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for fg in (0, 2):
        for _ in range(3):
            hip.gemm(a, b, out=out, force_generic=fg)
torch.cuda.synchronize()
