"""Fused d(act) GEMM + SwiGLU-backward epilogue vs the two separate launches at the 7B shape (T = 17536, I = 11008, K = 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
T, H, I = 17536, 4096, 11008
dev = torch.device("cuda:0")
dy = torch.randn(T, H, device=dev).to(torch.bfloat16)
wT = (torch.randn(I, H, device=dev) * 0.02).to(torch.bfloat16)
gu = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
def sep():
    dact = hip.gemm(dy, wT)
    return hip.swiglu_bwd_t(dact, gu)
t_sep = timeit(sep, iters=10)
t_gemm = timeit(lambda: hip.gemm(dy, wT), iters=10)
t_fused = timeit(lambda: hip.gemm_dact_swiglu_bwd(dy, wT, gu), iters=10)
print(f"separate {t_sep*1e3:.0f} us (gemm alone {t_gemm*1e3:.0f}) | fused {t_fused*1e3:.0f} us | saved {(t_sep-t_fused)*1e3:.0f} us per layer")
