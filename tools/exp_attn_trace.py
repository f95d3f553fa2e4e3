"""Per-iteration cycle stamps of one block of the assembly attention forward: where a wave's iteration goes -- wait for the staged
tile (vmcnt), barrier, C++ glue, tile statement. Needs an experiment build that records them:
    bash tools/build_attn_variant.sh trace "-DMLA_ATTN_TRACE=0"          (block 0 = the heaviest row block of the first (batch, head))
    MLA_HIP_LIB=mla_amd/csrc/build_exp/trace/libmla_hip.so MLA_ATTN_FWD=1 [MLA_ATTN_LDS_EXTRA=16384] python tools/exp_attn_trace.py [S] [B]
MLA_ATTN_LDS_EXTRA=16384 leaves one block per CU (one wave per SIMD: the statement's time without a competing wave). Each stamp
(s_memtime + a global store) costs the interval it closes ~300 cycles. Numbers: HISTORY.md "Round 4"."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mla_amd import hip

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, D = 32, 128
dev = torch.device("cuda:0")
qkv = (torch.randn(B * S, 3 * H * D, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
for _ in range(3):
    hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, None, D ** -0.5)
torch.cuda.synchronize()
buf = np.zeros(4 * 64 * 8, dtype=np.uint64)
lib = ctypes.CDLL(os.environ.get("MLA_HIP_LIB") or os.path.join(os.path.dirname(hip.__file__), "libmla_hip.so"))
rc = lib.mla_attn_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
t = buf.reshape(4, 64, 8).astype(np.int64)
n = int((t[3, :, 0] > 0).sum())
print(f"S={S} B={B}: {n} iterations traced (block {os.environ.get('TRACE_BLOCK', '?')}); cycles per iteration: wait-vm | barrier | glue | statement | total")
for w in range(4):
    tw = t[w, :n]
    wait, bar, glue, stmt = tw[:, 1] - tw[:, 0], tw[:, 2] - tw[:, 1], tw[:, 3] - tw[:, 2], tw[:, 4] - tw[:, 3]
    tot = np.diff(tw[:, 0])
    print(f"wave {w}: mean wait {wait[1:].mean():7.0f} barrier {bar[1:].mean():7.0f} glue {glue[1:].mean():6.0f} statement {stmt[1:].mean():7.0f} | iteration {tot.mean():7.0f}")
    if w == 3:
        for i in range(min(n, 40)):
            print(f"   it {i - 1:3d}: wait {wait[i]:6d} bar {bar[i]:6d} glue {glue[i]:5d} stmt {stmt[i]:6d}")
print("whole block, wave 3:", int(t[3, n - 1, 4] - t[3, 0, 0]), "cycles")
