"""Per-launch times of the folded-RMSNorm GEMM forms next to the launches they replace, at the 7B benchmark shapes (T = 17 536)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mla_amd import hip

dev = torch.device("cuda:0")
BF = torch.bfloat16
T, H, I = 17536, 4096, 11008
g = torch.Generator().manual_seed(0)
def r(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(BF).to(dev)
x, o, act = r(T, H), r(T, H), r(T, I)
wqkv, wo, wgu, wd = r(3 * H, H, sc=0.02), r(H, H, sc=0.02), r(2 * I, H, sc=0.02), r(H, I, sc=0.02)
ln = (1 + 0.1 * torch.randn(H, generator=g)).to(BF).to(dev)
cos = torch.randn(548, 64, generator=g).to(dev); sin = torch.randn(548, 64, generator=g).to(dev)
qkv = torch.empty(T, 3 * H, dtype=BF, device=dev)

def timeit(name, fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:46s} {e0.elapsed_time(e1) / n * 1000:9.1f} us", flush=True)

xg, rstd = hip.rmsnorm_prep(x, ln, 1e-5)
h, xg2, ss = hip.gemm_res_norm(o, wo, x, ln)
for rep in range(2):
    timeit("rmsnorm_fwd", lambda: hip.rmsnorm_fwd(x, ln, 1e-5))
    timeit("rmsnorm_prep", lambda: hip.rmsnorm_prep(x, ln, 1e-5))
    timeit("o_proj + residual", lambda: hip.gemm(o, wo, residual=x))
    timeit("o_proj + residual + norm outputs", lambda: hip.gemm_res_norm(o, wo, x, ln))
    timeit("down_proj + residual", lambda: hip.gemm(act, wd, residual=x))
    timeit("down_proj + residual + norm outputs", lambda: hip.gemm_res_norm(act, wd, x, ln))
    timeit("qkv + rope", lambda: hip.gemm_qkv_rope(x, wqkv, qkv, cos, sin, 548, 2 * H))
    timeit("qkv + rope, rstd from 16 partials", lambda: hip.gemm_qkv_rope(xg, wqkv, qkv, cos, sin, 548, 2 * H, norm=(ss, None, 1e-5)))
    timeit("qkv + rope, rstd given", lambda: hip.gemm_qkv_rope(xg, wqkv, qkv, cos, sin, 548, 2 * H, norm=(None, rstd, 1e-5)))
    timeit("gate|up + swiglu", lambda: hip.gemm_gateup_swiglu(x, wgu, True))
    timeit("gate|up + swiglu, rstd from 16 partials", lambda: hip.gemm_gateup_swiglu(xg, wgu, True, norm=(ss, None, 1e-5)))
    timeit("gate|up + swiglu, rstd given", lambda: hip.gemm_gateup_swiglu(xg, wgu, True, norm=(None, rstd, 1e-5)))
