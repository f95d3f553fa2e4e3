"""Run a few launches of one GEMM shape per (mode, kernel) so rocprofv3 --pmc can attribute counters per kernel.
Usage: python tools/pmc_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
T, H = 17536, 4096
for am, bm, M, N, K in [(0, 0, T, H, H), (0, 1, T, H, 3 * H), (1, 1, 3 * H, H, T)]:
    a = torch.randn((M, K) if am == 0 else (K, M), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K) if bm == 0 else (K, N), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for fg in (0, 2):
        for _ in range(3):
            hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=fg)
torch.cuda.synchronize()
