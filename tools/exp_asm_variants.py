"""Times gemm256 / asm8w / asm4w / hipBLASLt on a few shapes for whichever library MLA_HIP_LIB points at (experiment builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")
out = []
for M, N, K in ((16384, 4096, 4096), (17536, 4096, 4096), (4096, 11008, 17536), (17536, 12288, 4096)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K), device=dev).to(torch.bfloat16)
    o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    t = {n: timeit(lambda: hip.gemm(a, b, out=o, force_generic=fg), iters=20) for n, fg in (("k256", 3), ("a8w", 4), ("a4w", 5))}
    t["blaslt"] = timeit(lambda: torch.matmul(a, b.t(), out=o), iters=20)
    out.append(f"{M}x{N}x{K}: " + " ".join(f"{n} {fl / v / 1e9:6.0f}" for n, v in t.items()))
print(f"[{os.environ.get('MLA_HIP_LIB', 'product').split('/')[-2]:>5s}] " + " | ".join(out))
