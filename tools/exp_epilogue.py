"""How much of a gemm256 launch is the epilogue?  Runs each 7B shape with the normal library call and with alpha < 0, which the
experiment build (build_noep/, gemm256.hip patched to return before the epilogue stores when alpha < 0) turns into "main loop only".
Usage: MLA_HIP_LIB=build_noep/libmla_hip.so python tools/exp_epilogue.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
T, H, I = 17536, 4096, 11008
dev = torch.device("cuda:0")
shapes = [("qkv fwd", T, 3 * H, H, 0), ("o fwd", T, H, H, 0), ("gu fwd", T, 2 * I, H, 0), ("down fwd", T, H, I, 0),
          ("down dgrad", T, I, H, 0), ("gu dgrad", T, H, 2 * I, 0), ("gu wgrad f32", 2 * I, H, T, 1), ("o wgrad f32", H, H, T, 1)]
for name, M, N, K, f32 in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    od = torch.float32 if f32 else torch.bfloat16
    full = timeit(lambda: hip.gemm(a, b, out=out, out_dtype=od))
    noep = timeit(lambda: hip.gemm(a, b, out=out, out_dtype=od, alpha=-1.0))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = tiles / 256
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d} tiles {tiles:5d} ({rounds:5.2f} rounds): full {full*1e3:7.1f} us {fl/full/1e9:6.0f} TF/s | "
          f"no epilogue {noep*1e3:7.1f} us {fl/noep/1e9:6.0f} TF/s | epilogue = {(full-noep)*1e3:6.1f} us = {(full-noep)*1e3/rounds:5.2f} us per tile round")
