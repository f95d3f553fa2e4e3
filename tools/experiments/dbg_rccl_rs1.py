"""Debug (round 6): RCCL reduce_scatter_tensor at world 1, out of place, AVG vs SUM, odd tails."""
import os, torch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for n in (8, 1000, 4096 + 8, 1 << 20, (1 << 20) + 8, 12345672, 5_000_000 + 24):
    x = torch.randn(n, device=dev)
    for op, name in ((dist.ReduceOp.AVG, "AVG"), (dist.ReduceOp.SUM, "SUM")):
        y = torch.zeros(n, device=dev)
        dist.reduce_scatter_tensor(y, x, op=op)
        torch.cuda.synchronize()
        bad = (y != x).nonzero().flatten()
        print(n, name, "equal" if bad.numel() == 0 else f"{bad.numel()} differ, first {int(bad[0])} last {int(bad[-1])}, y there {y[bad[:4]].tolist()} x {x[bad[:4]].tolist()}")
    s = torch.cuda.Stream()
    y = torch.zeros(n, device=dev)
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s):
        s.wait_event(ev)
        dist.reduce_scatter_tensor(y, x, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    print(n, "AVG on a side stream:", "equal" if torch.equal(x, y) else "DIFFER")
dist.destroy_process_group()
