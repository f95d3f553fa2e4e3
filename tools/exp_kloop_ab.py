"""A/B of gemm256's two main loops (mla_gemm_kloop 1 = assembly, 0 = compiler-scheduled) on the k-contiguous launches of the 7B step:
TFLOP/s per launch, both modes in one process on the same buffers (alternating, so box / clock drift hits both alike)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")
T, H, I = 17536, 4096, 11008
BF = torch.bfloat16
r = lambda *s: (torch.randn(s, device=dev) * 0.05).to(BF)
cases = []
x, wqkv, wo, wgu, wd = r(T, H), r(3 * H, H), r(H, H), r(2 * I, H), r(H, I)
act, dy = r(T, I), r(T, H)
cos, sin = torch.randn(548, 64, device=dev), torch.randn(548, 64, device=dev)
oq = torch.empty((T, 3 * H), dtype=BF, device=dev)
cases.append(("qkv + rope   17536x12288x4096", 2.0 * T * 3 * H * H, lambda: hip.gemm_qkv_rope(x, wqkv, oq, cos, sin, 548, 2 * H)))
oo = torch.empty((T, H), dtype=BF, device=dev)
cases.append(("o / plain    17536x4096x4096", 2.0 * T * H * H, lambda: hip.gemm(x, wo, out=oo)))
cases.append(("o + residual 17536x4096x4096", 2.0 * T * H * H, lambda: hip.gemm(x, wo, out=oo, residual=dy)))
cases.append(("gate|up+swiglu 17536x22016x4096", 2.0 * T * 2 * I * H, lambda: hip.gemm_gateup_swiglu(x, wgu, True)))
od = torch.empty((T, H), dtype=BF, device=dev)
cases.append(("down fwd     17536x4096x11008", 2.0 * T * H * I, lambda: hip.gemm(act, wd, out=od)))
wdT = r(I, H)
gu = r(T, 2 * I)
cases.append(("dact+swiglu' 17536x11008x4096", 2.0 * T * I * H, lambda: hip.gemm_dact_swiglu_bwd(dy, wdT, gu)))
dyT, xT = r(H, T), r(H, T)
g32 = torch.zeros((H, H), dtype=torch.float32, device=dev)
cases.append(("wgrad (kept T) 4096x4096x17536 f32+=", 2.0 * H * H * T, lambda: hip.gemm(dyT, xT, out=g32, accumulate=True)))
actT = r(I, T)
g32d = torch.zeros((H, I), dtype=torch.float32, device=dev)
cases.append(("down wgrad   4096x11008x17536 f32+=", 2.0 * H * I * T, lambda: hip.gemm(dyT, actT, out=g32d, accumulate=True)))
prev = hip.gemm_kloop(-1)
for name, fl, fn in cases:
    t = {0: [], 1: []}
    for rep in range(3):
        for mode in (1, 0):
            hip.gemm_kloop(mode)
            t[mode].append(timeit(fn, iters=15))
    a, c = min(t[1]), min(t[0])
    print(f"{name:40s} asm {a * 1e3:8.1f} us {fl / a / 1e9:6.0f} TF | compiler {c * 1e3:8.1f} us {fl / c / 1e9:6.0f} TF | {100 * (c / a - 1):+5.1f} %")
hip.gemm_kloop(prev)
