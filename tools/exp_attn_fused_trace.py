"""Phase stamps of the fused backward kernel (MLA_ATTN_BWD_FUSED=8|4, experiment build with -DMLA_ATTN_BTRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mla_amd import hip
S, B, H, D = 548, 32, 32, 128
LD = 3 * H * D
dev = torch.device("cuda:0")
qkv = (torch.randn(B * S, LD, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
o, lse = hip.attn_fwd(q, k, v, B, S, H, D, LD, None, D ** -0.5)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
tr = (torch.empty((3 * H * D, B * S), dtype=torch.bfloat16, device=dev), torch.empty((H * D, B * S), dtype=torch.bfloat16, device=dev))
cos = torch.ones(S, D // 2, device=dev); sin = torch.zeros(S, D // 2, device=dev)
for _ in range(3):
    hip.attn_bwd(q, k, v, o, do, lse, None, dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:], B, S, H, D, LD, D ** -0.5, rope_cos=cos, rope_sin=sin, transposed=tr)
torch.cuda.synchronize()
N = 16384
buf = np.zeros(2 * N * 8, dtype=np.uint64)
lib = ctypes.CDLL(os.environ["MLA_HIP_LIB"])
assert lib.mla_attn_btrace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
a = buf.reshape(2, N, 8)[0][:B * H].astype(np.int64)
tot = a[:, 3] - a[:, 0]
print(f"fused kernel, {len(a)} heads: total {tot.mean():.0f} cycles per head; prologue {np.mean(a[:,1]-a[:,0]):.0f}; main loop incl. key-tile epilogues {np.mean(a[:,2]-a[:,1]):.0f} "
      f"(phase 1 {a[:,5].mean():.0f}, phase 2 {a[:,6].mean():.0f}, key-tile epilogues {a[:,4].mean():.0f}); dQ epilogues {np.mean(a[:,3]-a[:,2]):.0f}")
print(f"per pair (45): phase 1 {a[:,5].mean()/45:.0f}, phase 2 {a[:,6].mean()/45:.0f}; per key-tile epilogue pair (9): {a[:,4].mean()/9:.0f}; per dQ epilogue (9): {np.mean(a[:,3]-a[:,2])/9:.0f}")
