"""Fused-epilogue GEMM launches (gate|up + SwiGLU, d(act) + SwiGLU backward) at the 7B shapes with the first-round stagger experiment
(MLA_GEMM_STAGGER=<mode>:<ticks of 10 ns>, read once per process: run this script once per setting). Prints us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
T, H, I = 17536, 4096, 11008
dev = torch.device("cuda:0")
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w = (torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)
wT = (torch.randn(I, H, device=dev) * 0.02).to(torch.bfloat16)
gu = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
t_f = timeit(lambda: hip.gemm_gateup_swiglu(x, w, True), iters=20)
t_b = timeit(lambda: hip.gemm_dact_swiglu_bwd(x, wT, gu), iters=20)
t_p = timeit(lambda: hip.gemm(x, wT), iters=20)
print(f"MLA_GEMM_STAGGER={os.environ.get('MLA_GEMM_STAGGER', '-'):>8s}: gate|up+swiglu {t_f*1e3:7.1f} us | dact+swiglu_bwd {t_b*1e3:7.1f} us | plain [T,I,K=H] {t_p*1e3:7.1f} us")
