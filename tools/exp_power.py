"""Is the GEMM power-bound? Same kernel, same shapes, operands of decreasing bit activity: N(0,1) bf16, small-range integers, constants,
zeros. A kernel limited by instruction issue runs equally fast on all of them; one limited by the power cap speeds up as toggling drops."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
T, H, I = 17536, 4096, 11008
for name, M, N, K in (("gu dgrad", T, H, 2 * I), ("qkv fwd", T, 3 * H, H), ("o wgrad", H, H, T)):
    for kind in ("randn", "randint4", "ones", "zeros"):
        def mk(r, c):
            if kind == "randn":
                return torch.randn(r, c, device=dev).to(torch.bfloat16)
            if kind == "randint4":
                return torch.randint(-2, 2, (r, c), device=dev).to(torch.bfloat16)
            return (torch.ones if kind == "ones" else torch.zeros)(r, c, device=dev, dtype=torch.bfloat16)
        a, b = mk(M, K), mk(N, K)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ms = timeit(lambda: hip.gemm(a, b, out=out), iters=20)
        ms_ref = timeit(lambda: torch.matmul(a, b.t()), iters=20)
        fl = 2.0 * M * N * K
        print(f"{name:10s} {kind:9s} gemm256 {fl/ms/1e9:7.1f} TF/s | hipBLASLt {fl/ms_ref/1e9:7.1f} TF/s", flush=True)
