"""What makes AdamW's five streams fast or slow? Same kernel, same sizes: separately allocated tensors vs views into one allocation,
in one process; prints the virtual addresses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
n = 202_383_360
coef = torch.ones(1, device=dev)


def run(tag, views):
    for t in views[:4]:
        t.normal_()
    views[3].abs_()
    ms = timeit(lambda: hip.adamw_step(views[0], views[1], views[2], views[3], views[4], 1e-4, 0.9, 0.999, 1e-8, 0.0, 3, coef), iters=10)
    print(f"{tag:34s} {ms:7.3f} ms {n * 30 / ms / 1e9:5.2f} TB/s  VA GiB: " + " ".join(f"{t.data_ptr() / 2**30:9.4f}" for t in views), flush=True)


sep = [torch.empty(n, device=dev) for _ in range(4)] + [torch.empty(n, dtype=torch.bfloat16, device=dev)]
run("separate tensors (first)", sep)
arena = torch.empty(8 * 2**30, dtype=torch.uint8, device=dev)
slot = ((n * 4 + (1 << 21) - 1) >> 21) << 21


def carve(start, stride):
    out = []
    for j in range(5):
        off = start + j * stride
        nb = n * (2 if j == 4 else 4)
        out.append(arena[off:off + nb].view(torch.bfloat16 if j == 4 else torch.float32))
    return out


run("arena views, stride = slot", carve(0, slot))
run("arena views, stride = 1 GiB", carve(0, 1 << 30))
run("arena views, stride = 1.5 GiB", carve(0, 3 << 29))
run("separate tensors (again)", sep)
sep2 = [torch.empty(n, device=dev) for _ in range(4)] + [torch.empty(n, dtype=torch.bfloat16, device=dev)]
run("separate tensors (second set)", sep2)
big = [torch.empty(n * 2, device=dev) for _ in range(4)] + [torch.empty(n * 2, dtype=torch.bfloat16, device=dev)]
run("first halves of 2x tensors", [t[:n] for t in big])
run("second halves of 2x tensors", [t[n:] for t in big])
# single-stream read bandwidth (grad-norm kernel) per tensor of the slow and the fast set, and a 2-stream copy
out = torch.zeros(1, device=dev)
for tag, ts in (("slow set", sep), ("fast set", sep2)):
    bw = []
    for t in ts[:4]:
        ms = timeit(lambda: hip.sumsq(t, out, False), iters=10)
        bw.append(n * 4 / ms / 1e9)
    ms = timeit(lambda: ts[1].copy_(ts[0]), iters=10)
    print(f"{tag}: read-only TB/s per tensor " + " ".join(f"{b:5.2f}" for b in bw) + f" | copy t0->t1 {n * 8 / ms / 1e9:5.2f} TB/s")
