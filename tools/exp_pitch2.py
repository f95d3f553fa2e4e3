"""Channel camping test (round 3): the same GEMM with the leading dimensions of C, A and B padded away from multiples of 8 KiB.
HBM channels interleave at 256 B; a row pitch of 8 / 24 KiB puts the same-numbered chunk of every row on 4 of 128 channel groups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")


def padded(rows, cols, pad, scale=1.0):
    t = torch.empty((rows, cols + pad), dtype=torch.bfloat16, device=dev)
    t[:, :cols].copy_((torch.randn(rows, cols, device=dev) * scale).to(torch.bfloat16))
    return t[:, :cols]


for name, M, N, K in (("qkv fwd", 17536, 12288, 4096), ("o fwd", 17536, 4096, 4096), ("down dgrad", 17536, 11008, 4096), ("down fwd", 17536, 4096, 11008),
                      ("o wgrad", 4096, 4096, 17536)):
    fl = 2.0 * M * N * K
    res = []
    for pc, pa, pb in ((0, 0, 0), (64, 0, 0), (128, 0, 0), (0, 64, 64), (64, 64, 64), (128, 128, 128), (1088, 0, 0)):
        a, b, c = padded(M, K, pa), padded(N, K, pb, 0.02), padded(M, N, pc)
        t = timeit(lambda: hip.gemm(a, b, out=c), iters=20)
        res.append(f"C+{pc} A+{pa} B+{pb}: {fl / t / 1e9:5.0f}")
    print(f"{name:10s} M={M} N={N} K={K} | " + " | ".join(res), flush=True)
