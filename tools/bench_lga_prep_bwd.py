"""Timing of the deterministic lga_prep backward (mla_amd/csrc/pointcloud.hip, round 4: one workgroup per target point, fixed-order
gather over the batch's whole kNN list) at the point tower's real shapes, B = 32 = 8 samples x 4 diffusion repeats (advisor, round 4:
"record a before/after timing"). The scatter kernel it replaced is gone from the source; the stand-in for "before" is the same
reduction as a torch index_add_ (fp32 atomics, non-deterministic), which moves the same bytes.
Usage: python tools/bench_lga_prep_bwd.py > gpurun_out/lga_prep_bwd_timing.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip

dev = torch.device("cuda:0")


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
    return s.elapsed_time(e) / iters * 1e3


for name, B, N, G, K, C in (("stage 0", 32, 1024, 512, 81, 96), ("stage 1", 32, 512, 256, 81, 192)):
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz = torch.rand(B, N, 3, generator=g)
    fps_idx = torch.stack([torch.randperm(N, generator=g)[:G] for _ in range(B)]).to(dev)
    d = torch.cdist(xyz[torch.arange(B)[:, None], fps_idx.cpu()], xyz)
    knn_idx = d.topk(K, largest=False).indices.to(torch.int32).to(dev)              # a true kNN list (no repeated point per group)
    rows = B * G * K
    drows = (torch.randn(rows, 2 * C, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    t = timeit(lambda: hip.lga_prep_bwd(drows, fps_idx, knn_idx, B, N, C))
    out = hip.lga_prep_bwd(drows, fps_idx, knn_idx, B, N, C)
    # the same reduction as an atomic scatter: neighbour half -> feats[knn], centre half -> feats[fps]
    flat_knn = (knn_idx.long() + (torch.arange(B, device=dev) * N)[:, None, None]).reshape(-1)
    flat_ctr = (fps_idx.long() + (torch.arange(B, device=dev) * N)[:, None]).repeat_interleave(K, dim=1).reshape(-1)
    d32 = drows.float()

    def scatter():
        o = torch.zeros(B * N, C, device=dev)
        o.index_add_(0, flat_knn, d32[:, :C])
        o.index_add_(0, flat_ctr, d32[:, C:])
        return o
    ts = timeit(scatter, 5)
    ref = scatter().view(B, N, C)
    rel = float((out - ref).norm() / ref.norm())
    print(f"{name}: B={B} N={N} G={G} K={K} C={C}: lga_prep_bwd (deterministic gather) {t:8.1f} us | torch index_add_ stand-in (atomics, incl. its fp32 cast input) {ts:8.1f} us | "
          f"rows read {rows * 2 * C * 2 / 1e6:.0f} MB, index list {B * G * K * 4 / 1e6:.1f} MB re-scanned by {N} workgroups per batch row; agreement rel {rel:.1e}")
