"""Cost of the residual epilogue (fp32 staging, two passes) vs the plain bf16 epilogue on the two residual GEMMs of a layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
T, H, I = 17536, 4096, 11008
dev = torch.device("cuda:0")
for name, M, N, K in (("o fwd", T, H, H), ("down fwd", T, H, I)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    t0 = timeit(lambda: hip.gemm(a, b, out=out), iters=20)
    t1 = timeit(lambda: hip.gemm(a, b, out=out, residual=r), iters=20)
    fl = 2.0 * M * N * K
    print(f"{name}: plain {t0*1e3:.1f} us ({fl/t0/1e9:.0f} TF/s) | +residual {t1*1e3:.1f} us ({fl/t1/1e9:.0f} TF/s) | residual costs {(t1-t0)*1e3:.1f} us")
