"""Is the gemm256 epilogue cost a fixed latency or store-burst congestion?  One tile round with 16..256 concurrent tiles, bf16 and fp32
output, full launch vs the experiment build's "no epilogue stores" (alpha < 0). Usage: MLA_HIP_LIB=build_noep/libmla_hip.so python tools/exp_epilogue2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")
K = 4096
for f32 in (0, 1):
    for mt, nt in ((4, 4), (8, 4), (8, 8), (16, 8), (16, 16), (32, 16), (32, 32)):
        M, N = mt * 256, nt * 256
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        od = torch.float32 if f32 else torch.bfloat16
        out = torch.empty((M, N), dtype=od, device=dev)
        full = timeit(lambda: hip.gemm(a, b, out=out, out_dtype=od, force_generic=3), iters=20)
        noep = timeit(lambda: hip.gemm(a, b, out=out, out_dtype=od, alpha=-1.0, force_generic=3), iters=20)
        print(f"{'fp32' if f32 else 'bf16'} out, {mt * nt:4d} tiles ({mt * nt / 256:4.2f} rounds), K={K}: full {full * 1e3:7.1f} us | no epilogue {noep * 1e3:7.1f} us | "
              f"difference {(full - noep) * 1e3:6.1f} us")
