"""Times the experimental 4-wave asm GEMM alone (timing-only generator variants give wrong results: never use them for anything else)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")
out = []
for M, N, K in ((4096, 4096, 17536), (17536, 4096, 4096)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K), device=dev).to(torch.bfloat16)
    o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    t = timeit(lambda: hip.gemm(a, b, out=o, force_generic=5), iters=30)
    out.append(f"{M}x{N}x{K}: {t * 1e3:7.1f} us {fl / t / 1e9:6.0f} TF")
print(f"[{os.environ.get('MLA_HIP_LIB', 'product').split('/')[-2]:>8s}] " + " | ".join(out))
