"""Dump attention forward / backward outputs (one seeded input per shape) so that two library builds can be compared bit for bit:
MLA_HIP_LIB=<a> python tools/exp_attn_bits.py out_a.pt;  MLA_HIP_LIB=<b> python tools/exp_attn_bits.py out_b.pt;  ... cmp (both given: compares)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print(f"{len(a)} tensors compared, {len(bad)} differ: {bad[:8]}")
    sys.exit(1 if bad else 0)
from mla_amd import hip
dev = torch.device("cuda:0")
out = {}
H, D = 4, 128
for S, B, lens in ((548, 3, None), (548, 3, [548, 300, 37]), (100, 2, [100, 64]), (36, 2, None), (1024, 2, [1024, 999]), (132, 2, None)):
    torch.manual_seed(S + B)
    qkv = (torch.randn(B * S, 3 * H * D, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    sl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=dev)
    o, lse = hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, sl, D ** -0.5)
    do = torch.randn_like(o)
    dqkv = torch.zeros_like(qkv)
    dq, dk, dv = dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:]
    tr = (torch.zeros((3 * H * D, B * S), dtype=torch.bfloat16, device=dev), torch.zeros((H * D, B * S), dtype=torch.bfloat16, device=dev)) if S % 4 == 0 else None
    hip.attn_bwd(q, k, v, o, do, lse, sl, dq, dk, dv, B, S, H, D, 3 * H * D, D ** -0.5, transposed=tr)
    torch.cuda.synchronize()
    tag = f"S{S}_B{B}_{'ragged' if lens else 'full'}"
    out[tag + "_o"], out[tag + "_lse"], out[tag + "_dqkv"] = o.cpu(), lse.cpu(), dqkv.cpu()
    if tr is not None:
        out[tag + "_dqkvT"], out[tag + "_oT"] = tr[0].cpu(), tr[1].cpu()
    # the form the training step calls: RoPE backward fused into the epilogues (+ the transposed copies)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
    dqkv2 = torch.zeros_like(qkv)
    tr2 = (torch.zeros_like(tr[0]), torch.zeros_like(tr[1])) if tr is not None else None
    hip.attn_bwd(q, k, v, o, do, lse, sl, dqkv2[:, :H * D], dqkv2[:, H * D:2 * H * D], dqkv2[:, 2 * H * D:], B, S, H, D, 3 * H * D, D ** -0.5,
                 rope_cos=cos, rope_sin=sin, transposed=tr2)
    torch.cuda.synchronize()
    out[tag + "_rope_dqkv"] = dqkv2.cpu()
    if tr2 is not None:
        out[tag + "_rope_dqkvT"], out[tag + "_rope_oT"] = tr2[0].cpu(), tr2[1].cpu()
torch.save(out, sys.argv[1])
print("saved", len(out), "tensors")
