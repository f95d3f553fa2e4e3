"""RMSNorm forward / backward at the 7B shape (T = 17536, H = 4096): microseconds and effective TB/s. Usage: python tools/bench_rmsnorm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
T, H = 17536, 4096
dev = torch.device("cuda:0")
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
dy = torch.randn(T, H, device=dev).to(torch.bfloat16)
dres = torch.randn(T, H, device=dev).to(torch.bfloat16)
w = torch.ones(H, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(H, device=dev, dtype=torch.float32)
y, rstd = hip.rmsnorm_fwd(x, w, 1e-5)
u = T * H * 2
f = timeit(lambda: hip.rmsnorm_fwd(x, w, 1e-5), iters=20)
b = timeit(lambda: hip.rmsnorm_bwd(dy, x, w, rstd, dres=dres, dw_out=dw), iters=20)
print(f"rmsnorm fwd {f*1e3:.1f} us ({2*u/f/1e9:.2f} TB/s) | bwd (+dres, +dw) {b*1e3:.1f} us ({4*u/b/1e9:.2f} TB/s) | nblocks bwd = {hip.lib().mla_rmsnorm_bwd_blocks(T)}")
