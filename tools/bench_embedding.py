"""Embedding backward (grad[ids[t]] += dy[t]) timing + exactness against a sequential fp32 accumulation.
Usage: [MLA_EMB_SLICES=n] python tools/bench_embedding.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mla_amd import hip

dev = torch.device("cuda:0")
V, H = 32064, 4096
g = torch.Generator().manual_seed(0)
for T, S in ((16384, 2048), (65536, 2048)):
    for pad_frac in (0.0, 0.3):
        ids = torch.randint(0, 32000, (T // S, S), generator=g)
        npad = int(S * pad_frac)
        if npad:
            ids[:, S - npad:] = 32000
        ids = ids.reshape(-1)
        dy = (torch.randn(T, H, generator=g) * 0.01).to(torch.bfloat16)
        grad = torch.zeros(V, H, device=dev)
        idd, dyd = ids.to(dev), dy.to(dev)
        hip.embedding_bwd(idd, dyd, grad)
        if T == 16384:
            ref = torch.zeros(V, H)
            dyf = dy.float()
            for t in range(T):  # ascending token order, the order the kernel promises
                ref[ids[t]] += dyf[t]
            print("exact:", torch.equal(grad.cpu(), ref), "max rel", float((grad.cpu() - ref).abs().max() / ref.abs().max()))
            g2 = torch.zeros(V, H, device=dev)
            hip.embedding_bwd(idd, dyd, g2)
            print("deterministic:", torch.equal(g2, grad))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(10):
            hip.embedding_bwd(idd, dyd, grad)
        ev[1].record()
        torch.cuda.synchronize()
        print(f"T={T} pad={pad_frac}: {ev[0].elapsed_time(ev[1]) / 10 * 1e3:.1f} us")
