"""Cycle stamps inside the attention forward loop (experiment build build_tr/: attention.hip patched to write s_memtime-style cycle
counters of one block -- batch 5, head 0, second-heaviest row block -- into the buffer passed as `delta`)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
S = int(sys.argv[1]) if len(sys.argv) > 1 else 548
B, H, D = 32, 32, 128
dev = torch.device("cuda:0")
qkv = (torch.randn(B * S, 3 * H * D, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
o = torch.empty(B * S, H * D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
ts = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
L = hip.lib()
from ctypes import c_void_p, c_int, c_longlong, c_float
L.mla_attn_fwd_trace.argtypes = [c_void_p] * 6 + [c_int] * 4 + [c_longlong, c_longlong, c_float, c_void_p, c_void_p]
for it in range(3):
    ts.zero_()
    L.mla_attn_fwd_trace(hip._p(q), hip._p(k), hip._p(v), hip._p(o), hip._p(lse), None, B, S, H, D, 3 * H * D, H * D, D ** -0.5, hip._p(ts), hip._stream())
    torch.cuda.synchronize()
t = ts.cpu().view(8, 256)
for w in (0, 3, 7):
    r = t[w]
    print(f"wave {w}: block start->loop end {int(r[251] - r[250])} cycles")
    for kt in range(12):
        x = r[kt * 8:kt * 8 + 8]
        if int(x[0]) == 0:
            break
        print(f"   kt {kt}: vmcnt wait {int(x[6]-x[5]):6d} | barrier {int(x[0]-x[6]):6d} | stage-issue+QK {int(x[2]-x[0]):6d} | softmax {int(x[3]-x[2]):6d} | PV {int(x[4]-x[3]):6d} | total {int(x[4]-x[5]):6d}")
