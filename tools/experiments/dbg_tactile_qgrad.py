"""Debug (round 6): where does the 12 % error on tactile_gen_module.decoder.layers.0.multihead_attn.in_proj_weight (query rows) come
from? Runs the tactile e2e with the attention core of the generation heads replaced by torch math at different precisions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import math
import numpy as np
import torch
from mla_amd import ops
import parity_util as P
import test_tactile_gpu as TT

dev = torch.device("cuda", 0)
KEY = "vlm.generation_manager.tactile_gen_module.decoder.layers.0.multihead_attn.in_proj_weight"
orig = ops.mha_core


def torch_core(dtype, round_p=False, round_dp=False):
    def core(qsrc, kvsrc, nheads, nvalid, p, training):
        B, Sq = qsrc.shape[0], qsrc.shape[1]
        if kvsrc is None:
            E = qsrc.shape[2] // 3
            q, k, v = qsrc[..., :E], qsrc[..., E:2 * E], qsrc[..., 2 * E:]
        else:
            E = qsrc.shape[2]
            q, k, v = qsrc, kvsrc[..., :E], kvsrc[..., E:]
        k, v = k[:, :nvalid], v[:, :nvalid]
        hd = E // nheads
        sp = lambda t: t.to(dtype).view(B, -1, nheads, hd).transpose(1, 2)
        s = (sp(q) @ sp(k).transpose(-1, -2)) / math.sqrt(hd)
        pr = torch.softmax(s.float(), -1).to(dtype)
        o = pr @ sp(v)
        return o.transpose(1, 2).reshape(B, Sq, E).to(qsrc.dtype)
    return core


for tag, fn in (("hip kernels (default)", None), ("torch fp32 core", torch_core(torch.float32)), ("torch bf16 core (autograd through bf16 matmuls)", torch_core(torch.bfloat16))):
    ops.mha_core = fn or orig
    m, ld, gold = TT.run_tactile_e2e(dev)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    rows = {r["name"]: r for r in P.grad_sample_rows(grads, gold)}
    r = rows[KEY]
    over = [n for n, x in rows.items() if x["ratio"] > 2]
    print(f"{tag:55s} {KEY.split('tactile_gen_module.')[1]}: err hip {r['hip']:.4f} | C {r['C']:.4f} | ratio {r['ratio']:.2f}; tensors over 2x: {len(over)}")
    g = grads[KEY].float().cpu()
    print("      |g q-rows|", float(g[:128].norm()), "A sample norm", r["normA"])
ops.mha_core = orig
