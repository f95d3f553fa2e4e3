"""Block-phase cycle stamps of the two attention-backward kernels (experiment build with -DMLA_ATTN_BTRACE):
    bash tools/build_attn_variant.sh btrace "-DMLA_ATTN_BTRACE"
    MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace/libmla_hip.so python tools/exp_attn_btrace.py [S] [B]
Per kernel: mean cycles of prologue (entry -> loads issued), first-tile wait, main loop (per tile), epilogue staging, store issue, grouped by the
block's tile count; and per-CU timelines: how long a CU slot stays empty between one block's last stamp and the next block's entry."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mla_amd import hip

S = int(sys.argv[1]) if len(sys.argv) > 1 else 548
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H, D = 32, 128
LD = 3 * H * D
dev = torch.device("cuda:0")
qkv = (torch.randn(B * S, LD, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
scale = D ** -0.5
o, lse = hip.attn_fwd(q, k, v, B, S, H, D, LD, None, scale)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:]
tr = (torch.empty((3 * H * D, B * S), dtype=torch.bfloat16, device=dev), torch.empty((H * D, B * S), dtype=torch.bfloat16, device=dev)) if S % 4 == 0 else None
cos = torch.ones(S, D // 2, device=dev); sin = torch.zeros(S, D // 2, device=dev)
for _ in range(3):
    hip.attn_bwd(q, k, v, o, do, lse, None, dq, dk, dv, B, S, H, D, LD, scale, rope_cos=cos, rope_sin=sin, transposed=tr)
torch.cuda.synchronize()
N = 16384
buf = np.zeros(2 * N * 8, dtype=np.uint64)
lib = ctypes.CDLL(os.environ.get("MLA_HIP_LIB") or os.path.join(os.path.dirname(hip.__file__), "libmla_hip.so"))
assert lib.mla_attn_btrace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
t = buf.reshape(2, N, 8)
for kern, name in ((0, "dQ"), (1, "dK.dV")):
    a = t[kern]
    live = a[:, 0] > 0
    a = a[live].astype(np.int64)
    full = a[:, 5] > 0            # padding-only blocks return early
    a = a[full]
    tiles = a[:, 6]
    pro, first, loop, stage, store = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3], a[:, 5] - np.where(a[:, 4] > 0, a[:, 4], a[:, 3])
    if kern == 0:
        stage = np.zeros_like(stage)   # the dQ epilogue is stamped as one interval (3 -> 5)
        store = a[:, 5] - a[:, 3]
    span = a[:, 5].max() - a[:, 0].min()
    print(f"== {name}: {len(a)} blocks traced (S={S} B={B}); kernel span {span} cycles; block total mean {np.mean(a[:, 5] - a[:, 0]):.0f}")
    print("   tiles  blocks   prologue  first-tile-wait   loop  (per tile)   epilogue-stage  store/epilogue   total")
    for nt in sorted(set(tiles.tolist())):
        m = tiles == nt
        print(f"   {nt:5d} {m.sum():7d} {pro[m].mean():10.0f} {first[m].mean():16.0f} {loop[m].mean():7.0f} {loop[m].mean() / max(nt, 1):10.0f} {stage[m].mean():15.0f} {store[m].mean():15.0f} {np.mean(a[m, 5] - a[m, 0]):8.0f}")
    print(f"   all: prologue {pro.sum() / (a[:, 5] - a[:, 0]).sum():.1%}, first-tile wait {first.sum() / (a[:, 5] - a[:, 0]).sum():.1%}, loop {loop.sum() / (a[:, 5] - a[:, 0]).sum():.1%}, "
          f"epilogue {(stage + store).sum() / (a[:, 5] - a[:, 0]).sum():.1%} of the traced blocks' stamped time")
    # per-CU timelines: HW_ID bits (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (+ xcc in the upper word)
    hw = a[:, 7]
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (((hw >> 32) & 0xF) << 8)
    gaps, busy, lifetimes = [], [], []
    for c in np.unique(cu):
        rows = a[cu == c]
        rows = rows[np.argsort(rows[:, 0])]
        ends = np.sort(rows[:, 5])
        # resident blocks per CU = 2: the k-th entry pairs with the (k-2)-th end
        for i in range(2, len(rows)):
            gaps.append(rows[i, 0] - ends[i - 2])
        lifetimes.append(rows[:, 5].max() - rows[:, 0].min())
        busy.append((rows[:, 5] - rows[:, 0]).sum() / 2)
    gaps = np.array(gaps)
    print(f"   {len(np.unique(cu))} CUs seen; slot turnover (next entry - end of the block whose slot it takes): median {np.median(gaps):.0f}, mean {gaps.mean():.0f}, p90 {np.percentile(gaps, 90):.0f} cycles; "
          f"stamped block time / (2 slots x CU lifetime) = {np.sum(busy) / np.sum(lifetimes):.1%}")
