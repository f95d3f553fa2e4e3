"""Fixed cost per tile of the 256x256 GEMM: time(K) at fixed M, N for the one-workgroup-per-tile launch, the persistent walk and
hipBLASLt; the intercept of the linear fit / (tiles / 256) is the per-tile overhead (launch + pipeline fill + epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
for M, N in ((17536, 4096), (16384, 4096), (17536, 12288)):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rows = []
    for K in (512, 1024, 2048, 4096, 8192, 16384):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        t0 = timeit(lambda: hip.gemm(a, b, out=out), iters=20)
        t1 = timeit(lambda: hip.gemm(a, b, out=out, force_generic=0x1000), iters=20)
        t2 = timeit(lambda: torch.matmul(a, b.t()), iters=20)
        rows.append((K, t0, t1, t2))
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K:6d}: per-tile {t0*1e3:8.1f} us {fl/t0/1e9:7.1f} TF/s | persistent {t1*1e3:8.1f} us {fl/t1/1e9:7.1f} | hipBLASLt {t2*1e3:8.1f} us {fl/t2/1e9:7.1f}", flush=True)
    r = np.array(rows)
    for name, col in (("per-tile", 1), ("persistent", 2), ("hipBLASLt", 3)):
        slope, icpt = np.polyfit(r[2:, 0], r[2:, col], 1)
        rounds = tiles / 256
        print(f"   {name:10s}: {slope*64*1e3/rounds:6.3f} us per K-tile per CU-round, intercept {icpt*1e3:7.1f} us = {icpt*1e3/rounds:6.2f} us per tile round ({tiles} tiles, {rounds:.2f} rounds)")
