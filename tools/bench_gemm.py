"""Micro-benchmark of the hand-written MFMA GEMM on the Llama-2-7B shapes (random data), next to torch.matmul
(hipBLASLt) as an on-box reference ceiling. Usage: python tools/bench_gemm.py [tokens]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip

T = int(sys.argv[1]) if (__name__ == "__main__" and len(sys.argv) > 1) else 17536
dev = torch.device("cuda:0")
H, I = 4096, 11008
shapes = [  # name, M, N, K, a_mode, b_mode -- every large GEMM of a decoder layer in the all-NT formulation
    ("qkv fwd", T, 3 * H, H, 0, 0), ("o fwd", T, H, H, 0, 0), ("gu fwd", T, 2 * I, H, 0, 0), ("down fwd", T, H, I, 0, 0),
    ("qkv dgrad", T, H, 3 * H, 0, 0), ("o dgrad", T, H, H, 0, 0), ("gu dgrad", T, H, 2 * I, 0, 0), ("down dgrad", T, I, H, 0, 0),
    ("qkv wgrad", 3 * H, H, T, 0, 0), ("o wgrad", H, H, T, 0, 0), ("gu wgrad", 2 * I, H, T, 0, 0), ("down wgrad", H, I, T, 0, 0),
    ("qkv wgrad TN(tr)", 3 * H, H, T, 1, 1),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, M, N, K, am, bm in (shapes if __name__ == "__main__" else []):
    a = torch.randn((M, K) if am == 0 else (K, M), device=dev).to(torch.bfloat16)
    b = torch.randn((N, K) if bm == 0 else (K, N), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm))
    ms128 = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=2))
    ms8w = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=4)) if (am == 0 and bm == 0 and K % 128 == 0) else float("nan")
    ms4w = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=5)) if (am == 0 and bm == 0 and K % 128 == 0) else float("nan")
    A = a if am == 0 else a.t()
    Bt = b.t() if bm == 0 else b
    ms_ref = timeit(lambda: torch.matmul(A, Bt))
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d}  k256 {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s | asm8w {fl/ms8w/1e9:7.1f} | asm4w {fl/ms4w/1e9:7.1f} | k128 {ms128:7.3f} ms {fl/ms128/1e9:7.1f} TF/s | hipBLASLt {ms_ref:8.3f} ms {fl/ms_ref/1e9:8.1f} TF/s", flush=True)
