"""Tile-quantisation experiment: the 256x256 kernel on N=4096 shapes with token counts that give whole / fractional rounds of
256 CUs. Usage: python tools/exp_quant.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit  # noqa

dev = torch.device("cuda:0")
for N, K in ((4096, 4096), (4096, 11008), (4096, 12288), (12288, 4096)):
    for T in (16384, 17408, 17536, 18432, 20480):
        a = torch.randn((T, K), device=dev).to(torch.bfloat16)
        b = torch.randn((N, K), device=dev).to(torch.bfloat16)
        out = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        ms = timeit(lambda: hip.gemm(a, b, out=out))
        tiles = ((T + 255) // 256) * (N // 256)
        fl = 2.0 * T * N * K
        print(f"N={N:6d} K={K:6d} T={T:6d} tiles={tiles:5d} rounds={tiles/256:5.2f}  {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s  ms/round-ceil {ms/-(-tiles//256):.4f}", flush=True)
