import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
T, H = 17536, 4096
def timeit(fn, iters=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, am, bm, M, N, K in [("warm", 0, 0, T, 3*H, H), ("NT qkv", 0, 0, T, 3*H, H), ("NT wgradqkv", 0, 0, 3*H, H, T)]:
    a = torch.randn((M, K) if am == 0 else (K, M), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K) if bm == 0 else (K, N), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    for dbg in (0, 4, 8, 12, 16, 32, 48, 60):
        ms = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=dbg << 4))
        print(f"{name:8s} debug={dbg} {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s", flush=True)
