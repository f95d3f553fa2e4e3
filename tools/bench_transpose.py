"""Bandwidth of the tile-transpose kernels on the shapes of a decoder layer's backward (MLA_TRANSPOSE_TILE=64 for the old tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
T, H, I = 17536, 4096, 11008
for name, R, C in (("dY gate|up", T, 2 * I), ("dY qkv", T, 3 * H), ("dY o/down", T, H), ("W gate|up", 2 * I, H), ("W down", H, I)):
    x = torch.randn(R, C, device=dev).to(torch.bfloat16)
    ms = timeit(lambda: hip.transpose(x), iters=20)
    print(f"{name:12s} [{R:6d},{C:6d}] {ms*1e3:8.1f} us  {R*C*4/ms/1e9:6.2f} TB/s")
gu = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
ms = timeit(lambda: hip.swiglu_fwd_t(gu), iters=20)
print(f"swiglu_fwd_t [{T},{I}] {ms*1e3:8.1f} us  {T*I*6/ms/1e9:6.2f} TB/s")
