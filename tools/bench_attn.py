"""Micro-benchmark of the causal flash-attention kernels at the 7B shapes. Usage: python tools/bench_attn.py [S] [B] [pad]
(pad = extra elements per token row of the packed q|k|v buffer: the token stride decides how K / V rows spread over the L2 channels)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip

S = int(sys.argv[1]) if len(sys.argv) > 1 else 548
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
PAD = int(sys.argv[3]) if len(sys.argv) > 3 else 0
H, D = 32, 128
LD = 3 * H * D + PAD
dev = torch.device("cuda:0")
qkv = (torch.randn(B * S, LD, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:3 * H * D]
scale = D ** -0.5


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


o, lse = hip.attn_fwd(q, k, v, B, S, H, D, LD, None, scale)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:3 * H * D]
fwd = timeit(lambda: hip.attn_fwd(q, k, v, B, S, H, D, LD, None, scale))
bwd = timeit(lambda: hip.attn_bwd(q, k, v, o, do, lse, None, dq, dk, dv, B, S, H, D, LD, scale))
fl = 4.0 * B * H * (S * S / 2) * D
print(f"S={S} B={B} pad={PAD}: fwd {fwd*1e3:.0f} us {fl/fwd/1e9:.0f} TF/s | bwd {bwd*1e3:.0f} us {2.5*fl/bwd/1e9:.0f} TF/s (causal flops)")
# reference check vs torch SDPA math in fp32 on one (b, h)
qf = q[:S, :D].float(); kf = k[:S, :D].float(); vf = v[:S, :D].float()
sc = (qf @ kf.t()) * scale + torch.full((S, S), float("-inf"), device=dev).triu(1)
ref = torch.softmax(sc, -1) @ vf
got = o.view(B, S, H, D)[0, :, 0].float()
print("fwd rel err", float((got - ref).norm() / ref.norm()))
