"""Do two independent GEMMs finish sooner when the second runs on a low-priority stream and fills the CUs the first one leaves idle in
its last, partial round of tiles? (fused gate|up + SwiGLU: 5934 tiles = 23.2 rounds, no split-K tail; d(act) + SwiGLU backward: 2967 =
11.6 rounds.) Pairs as the backward / forward of a decoder layer could issue them; sequential = one stream, back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
T, H, I = 17536, 4096, 11008
BF = torch.bfloat16
r = lambda *s: (torch.randn(s, device=dev) * 0.05).to(BF)
x, wgu, dy, wdT, gu = r(T, H), r(2 * I, H), r(T, H), r(I, H), r(T, 2 * I)
dyT, actT, xT = r(H, T), r(I, T), r(H, T)
g_down = torch.zeros((H, I), dtype=torch.float32, device=dev)
g_o = torch.zeros((H, H), dtype=torch.float32, device=dev)
pairs = {
    "dact+swiglu' | down wgrad": (lambda: hip.gemm_dact_swiglu_bwd(dy, wdT, gu), lambda: hip.gemm(dyT, actT, out=g_down, accumulate=False)),
    "gate|up+swiglu | o wgrad": (lambda: hip.gemm_gateup_swiglu(x, wgu, True), lambda: hip.gemm(dyT, xT, out=g_o, accumulate=False)),
}
hi, lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, (f1, f2) in pairs.items():
    def seq():
        f1(); f2()

    def par():
        cur = torch.cuda.current_stream()
        hi.wait_stream(cur); lo.wait_stream(cur)
        with torch.cuda.stream(hi):
            f1()
        with torch.cuda.stream(lo):
            f2()
        cur.wait_stream(hi); cur.wait_stream(lo)

    res = []
    for rep in range(3):
        res.append((timed(seq), timed(par)))
    s, p = min(a for a, _ in res), min(b for _, b in res)
    print(f"{name:28s} sequential {s:8.1f} us | high + low priority streams {p:8.1f} us | {100 * (s / p - 1):+5.1f} %")
