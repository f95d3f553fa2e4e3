"""Per-round overhead of the 256x256 kernels: time of M=16384, N=4096 (exactly 4 rounds of 256 tiles) against K -> slope (us per K-tile
of one round) and intercept (prologue + epilogue + launch per round). Experiment build (force_generic 4 = gemm8w with direct stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")
M, N = 16384, 4096
fgs = [("k256", 3)] + ([("a8w", 4)] if hip.lib().mla_query(3) == 1 else [])
res = {n: [] for n, _ in fgs}
Ks = (1024, 2048, 4096, 8192, 16384)
for K in Ks:
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K), device=dev).to(torch.bfloat16)
    o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for n, fg in fgs:
        res[n].append(min(timeit(lambda: hip.gemm(a, b, out=o, force_generic=fg), iters=20) for _ in range(3)) * 1e3)
for n, _ in fgs:
    t = res[n]
    slope = (t[-1] - t[1]) / ((Ks[-1] - Ks[1]) / 64) / 4
    icpt = [(t[i] / 4 - slope * Ks[i] / 64) for i in range(len(Ks))]
    print(f"{n:5s} kloop={hip.gemm_kloop(-1)} " + " ".join(f"K={k}: {v:7.1f}us" for k, v in zip(Ks, t)) + f" | slope {slope:.3f} us/K-tile/round, intercept per round " + " ".join(f"{v:5.1f}" for v in icpt))
