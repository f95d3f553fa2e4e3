"""AdamW kernel bandwidth: 16-B path (aligned shard) vs the scalar path (shard offset by one element)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
n = 202_383_360 + 8          # one decoder layer's parameters
bufs = [torch.randn(n, device=dev) for _ in range(4)]
p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, device=dev)
for off, name in ((0, "aligned (vec4)"), (1, "offset by 1 (scalar)")):
    p, g, m, v = (b[off:off + n - 8] for b in bufs)
    v.abs_()
    q = p16[off * 4:off * 4 + n - 8] if off == 0 else p16[off:off + n - 8]
    ms = timeit(lambda: hip.adamw_step(p, g, m, v, q, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, coef), iters=20)
    print(f"{name:22s} {ms:7.3f} ms  {(n - 8) * 30 / ms / 1e9:6.2f} TB/s (30 B/element)")
