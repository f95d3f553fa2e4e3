"""AdamW kernel bandwidth: (a) one shard re-used every iteration, 16-B path vs scalar path (shard offset by one element);
(b) the in-step pattern: every launch works on a different 202 M-element slice of 6.5 GB buffers, right after a GEMM burst."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
n = 202_383_360          # one decoder layer's parameters
L = 8
bufs = [torch.randn(n * L + 8, device=dev) for _ in range(4)]
bufs[3].abs_()
p16 = torch.empty(n * L + 8, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, device=dev)
for off, name in ((0, "aligned (vec4)"), (1, "offset by 1 (scalar)")):
    p, g, m, v = (b[off:off + n] for b in bufs)
    q = p16[off:off + n]
    ms = timeit(lambda: hip.adamw_step(p, g, m, v, q, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, coef), iters=20)
    print(f"{name:22s} {ms:7.3f} ms  {n * 30 / ms / 1e9:6.2f} TB/s (30 B/element)")


def sweep():
    for l in range(L):
        p, g, m, v = (b[l * n:(l + 1) * n] for b in bufs)
        hip.adamw_step(p, g, m, v, p16[l * n:(l + 1) * n], 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, coef)


ms = timeit(sweep, iters=5)
print(f"{L} distinct slices per sweep: {ms / L:7.3f} ms per slice  {n * 30 / (ms / L) / 1e9:6.2f} TB/s")
a = torch.randn(17536, 4096, device=dev).to(torch.bfloat16)
w = torch.randn(22016, 4096, device=dev).to(torch.bfloat16)


def hot_sweep():
    for _ in range(12):
        hip.gemm(a, w)
    sweep()


for burst in (12, 250, 600):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for _ in range(burst):
        hip.gemm(a, w)
    s.record()
    sweep()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    print(f"same sweep right after {burst * 2.4:6.0f} ms of GEMMs: {ms / L:7.3f} ms per slice  {n * 30 / (ms / L) / 1e9:6.2f} TB/s")
