"""AdamW bandwidth against where the five streams of a unit live (round 4): the in-step optimizer runs at 5.4 TB/s while the same kernel
reaches 6.2 TB/s on tools/bench_adamw.py's buffers. (a) per-unit allocations in FlatUnit's order (grad32, bf16 replica, master, m, v
of unit 0, then unit 1, ...) as ShardedModel makes them; (b) one arena per stream kind, units as slices of it. Same kernel, same sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip

dev = torch.device("cuda:0")
n = 202_383_360          # one decoder layer's parameters
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
coef = torch.ones(1, device=dev)


def sweep(units):
    for g, p16, p, m, v in units:
        hip.adamw_step(p, g, m, v, p16, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, coef)


def timeit(units, iters=4):
    sweep(units)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        sweep(units)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters / len(units)
    return ms, n * 30 / ms / 1e9


def make_separate():
    units = []
    for _ in range(L):
        g = torch.zeros(n, dtype=torch.float32, device=dev)
        p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        p = torch.zeros(n, dtype=torch.float32, device=dev)
        m = torch.zeros(n, dtype=torch.float32, device=dev)
        v = torch.zeros(n, dtype=torch.float32, device=dev)
        units.append((g, p16, p, m, v))
    return units


def make_arena():
    G = torch.zeros(n * L, dtype=torch.float32, device=dev)
    P16 = torch.zeros(n * L, dtype=torch.bfloat16, device=dev)
    P = torch.zeros(n * L, dtype=torch.float32, device=dev)
    M = torch.zeros(n * L, dtype=torch.float32, device=dev)
    V = torch.zeros(n * L, dtype=torch.float32, device=dev)
    return [(G[i * n:(i + 1) * n], P16[i * n:(i + 1) * n], P[i * n:(i + 1) * n], M[i * n:(i + 1) * n], V[i * n:(i + 1) * n]) for i in range(L)]


def make_arena_aligned(skews=(0, 0, 0, 0, 0)):
    """arena per kind, every unit's slice on a 2 MiB boundary (+ a per-kind skew in bytes)"""
    def arena(dtype, skew):
        es = 2 if dtype == torch.bfloat16 else 4
        per = ((n * es + (2 << 20) - 1) // (2 << 20)) * (2 << 20) // es
        A = torch.zeros(per * L + (4 << 20) // es, dtype=dtype, device=dev)
        base = (-A.data_ptr()) % (2 << 20) // es + skew // es
        return [A[base + i * per: base + i * per + n] for i in range(L)]
    G, P16, P, M, V = (arena(torch.float32, skews[0]), arena(torch.bfloat16, skews[1]), arena(torch.float32, skews[2]),
                       arena(torch.float32, skews[3]), arena(torch.float32, skews[4]))
    return list(zip(G, P16, P, M, V))


def make_separate_skewed(skews):
    units = []
    for _ in range(L):
        bufs = []
        for dtype, sk in zip((torch.float32, torch.bfloat16, torch.float32, torch.float32, torch.float32), skews):
            es = 2 if dtype == torch.bfloat16 else 4
            t = torch.zeros(n + sk // es, dtype=dtype, device=dev)
            bufs.append(t[sk // es:])
        units.append(tuple(bufs))
    return units


K = 1 << 10
for name, mk in (("separate allocations per unit", make_separate), ("one arena per stream kind (unaligned slices)", make_arena),
                 ("arenas, slices 2 MiB aligned", make_arena_aligned),
                 ("arenas, aligned + skew 0/256K/512K/768K/1M", lambda: make_arena_aligned((0, 256 * K, 512 * K, 768 * K, 1024 * K))),
                 ("arenas, aligned + skew 0/4K/8K/12K/16K", lambda: make_arena_aligned((0, 4 * K, 8 * K, 12 * K, 16 * K))),
                 ("separate + skew 0/256K/512K/768K/1M", lambda: make_separate_skewed((0, 256 * K, 512 * K, 768 * K, 1024 * K))),
                 ("separate allocations per unit", make_separate)):
    units = mk()
    for g, p16, p, m, v in units:
        g.normal_(); p.normal_(); v.uniform_()
    ms, tb = timeit(units)
    print(f"{name:48s} {L} units: {ms:.3f} ms per unit  {tb:.2f} TB/s", flush=True)
    del units
    torch.cuda.empty_cache()
