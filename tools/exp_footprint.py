import os, sys
sys.path.insert(0, os.getcwd())
import torch
from mla_amd import hip
from tools.bench_gemm import timeit
dev = torch.device("cuda:0")
n = 202_383_360
coef = torch.ones(1, device=dev)
def run(tag):
    bufs = [torch.randn(n, device=dev) for _ in range(4)]
    bufs[3].abs_()
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: hip.adamw_step(bufs[0], bufs[1], bufs[2], bufs[3], p16, 1e-4, 0.9, 0.999, 1e-8, 0.0, 3, coef), iters=20)
    print(f"{tag:40s} {ms:7.3f} ms {n*30/ms/1e9:6.2f} TB/s  free={torch.cuda.mem_get_info()[0]/2**30:.0f} GiB", flush=True)
    del bufs, p16
run("empty process")
hold = []
for gb in (32, 64, 96, 128):
    while sum(t.numel() for t in hold) * 4 < gb * 2**30:
        hold.append(torch.zeros(2**28, device=dev))      # 1 GiB pieces, touched
    run(f"{gb} GiB held in 1 GiB tensors")
del hold
torch.cuda.empty_cache()
run("after freeing")
