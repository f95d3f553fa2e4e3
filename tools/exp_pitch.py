"""Experiment: does a non-power-of-two row pitch change the 256-kernel's speed (L2/HBM channel camping)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
T, H = 17536, 4096


def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, am, bm, M, N, K in [("NT o", 0, 0, T, H, H), ("NN qkv", 0, 1, T, H, 3 * H), ("TN qkv", 1, 1, 3 * H, H, T), ("TN down", 1, 1, H, 11008, T)]:
    for pad in (0, 64, 136):
        ar, ac = (M, K) if am == 0 else (K, M)
        br, bc = (N, K) if bm == 0 else (K, N)
        a = torch.randn((ar, ac + pad), device=dev).to(torch.bfloat16)[:, :ac]
        b = torch.randn((br, bc + pad), device=dev).to(torch.bfloat16)[:, :bc]
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        for fg in (0, 2):
            ms = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=fg))
            print(f"{name:8s} pad={pad:4d} kernel={'256' if fg == 0 else '128'} {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s", flush=True)
