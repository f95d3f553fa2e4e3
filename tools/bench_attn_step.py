"""Attention kernels exactly as the training step calls them (fused RoPE backward, transposed dqkv / o copies written by the kernels):
forward and backward per-launch times from HIP events + error vs an fp32 reference on two (b, h) pairs.
Usage: python tools/bench_attn_step.py [S] [B] [ragged]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip

S = int(sys.argv[1]) if len(sys.argv) > 1 else 548
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ragged = len(sys.argv) > 3 and sys.argv[3] == "1"
H, D = 32, 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
T = B * S
qkv = (torch.randn(T, 3 * H * D, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
scale = D ** -0.5
seqlens = None
if ragged:
    seqlens = torch.full((B,), S, dtype=torch.int32, device=dev)
    seqlens[1::2] = S - 37
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


o, lse = hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, seqlens, scale)
do = (torch.randn(T, H * D, device=dev) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:]
tr = (torch.empty((3 * H * D, T), dtype=torch.bfloat16, device=dev), torch.empty((H * D, T), dtype=torch.bfloat16, device=dev)) if S % 4 == 0 else None
fwd = timeit(lambda: hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, seqlens, scale))
bwd = timeit(lambda: hip.attn_bwd(q, k, v, o, do, lse, seqlens, dq, dk, dv, B, S, H, D, 3 * H * D, scale, rope_cos=cos, rope_sin=sin, transposed=tr))
fl = 4.0 * B * H * (S * S / 2) * D
print(f"S={S} B={B} ragged={int(ragged)}: fwd {fwd*1e3:7.1f} us {fl/fwd/1e12:6.3f} PF/s | bwd(+rope, +T) {bwd*1e3:7.1f} us {2.5*fl/bwd/1e12:6.3f} PF/s   [lib {os.environ.get('MLA_HIP_LIB', 'product')[-40:]}]")
# fp32 reference on (b, h) = (0, 0) and (B - 1, H - 1), WITHOUT rope on the gradients (plain backward call)
hip.attn_bwd(q, k, v, o, do, lse, seqlens, dq, dk, dv, B, S, H, D, 3 * H * D, scale)
for b, h in ((0, 0), (B - 1, H - 1)):
    L = int(seqlens[b]) if seqlens is not None else S
    sl = slice(b * S, b * S + L)
    qf, kf, vf = (t[sl, h * D:(h + 1) * D].float().requires_grad_(True) for t in (q, k, v))
    sc = (qf @ kf.t()) * scale + torch.full((L, L), float("-inf"), device=dev).triu(1)
    ref = torch.softmax(sc, -1) @ vf
    ref.backward(do[sl, h * D:(h + 1) * D].float())
    e = lambda a, r: float((a.float() - r).norm() / r.norm())   # noqa: E731
    print(f"  (b={b}, h={h}) rel err: o {e(o[sl, h * D:(h + 1) * D], ref):.2e} dq {e(dq[sl, h * D:(h + 1) * D], qf.grad):.2e} "
          f"dk {e(dk[sl, h * D:(h + 1) * D], kf.grad):.2e} dv {e(dv[sl, h * D:(h + 1) * D], vf.grad):.2e}")
