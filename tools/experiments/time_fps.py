"""FPS kernel time at the benchmark's shapes (32 clouds: 1024 -> 256 and 256 -> 128 points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
for N, G in ((1024, 256), (256, 128), (1024, 512)):
    xyz = torch.randn(32, N, 3, device=dev)
    start = torch.randint(0, N, (32,), device=dev)
    for _ in range(3): idx = hip.fps(xyz, start, G)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): idx = hip.fps(xyz, start, G)
    e1.record(); torch.cuda.synchronize()
    # plain-torch restatement of the loop (fp32, same expression order) on the first two clouds
    ok = True
    for b in range(2):
        p = xyz[b].cpu(); far = int(start[b]); dist = torch.full((N,), 1e10); got = idx[b].cpu()
        for it in range(G):
            ok &= int(got[it]) == far
            d = ((p - p[far]) ** 2); d = (d[:, 0] + d[:, 1]) + d[:, 2]
            dist = torch.minimum(dist, d); far = int(torch.argmax(dist))
    print(f"fps N={N} -> {G}: {e0.elapsed_time(e1) / 20 * 1000:7.1f} us   indices match the torch loop: {ok}")
