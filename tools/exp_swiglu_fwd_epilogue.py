"""Fused gate|up GEMM + SwiGLU epilogue vs GEMM + swiglu_fwd_dual at the 7B shape (T = 17536, I = 11008, K = 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
T, H, I = 17536, 4096, 11008
dev = torch.device("cuda:0")
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w = (torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)
def sep():
    gu = hip.gemm(x, w)
    return hip.swiglu_fwd_dual(gu)
t_sep = timeit(sep, iters=10)
t_gemm = timeit(lambda: hip.gemm(x, w), iters=10)
t_fused = timeit(lambda: hip.gemm_gateup_swiglu(x, w, True), iters=10)
t_fused_nt = timeit(lambda: hip.gemm_gateup_swiglu(x, w, False), iters=10)
print(f"separate {t_sep*1e3:.0f} us (gemm alone {t_gemm*1e3:.0f}) | fused {t_fused*1e3:.0f} us (without act^T {t_fused_nt*1e3:.0f}) | saved {(t_sep-t_fused)*1e3:.0f} us per layer")
