"""In-situ timing of the optimizer phase (grad-norm + AdamW over the 7B flat buffers) and of one unit's AdamW launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mla_amd import hip
from mla_amd.strategy import FSDPStrategy

dev = torch.device("cuda", 0)
torch.manual_seed(42)


def ev():
    return torch.cuda.Event(enable_timing=True)


if os.environ.get("ARENA_GIB"):
    # one pristine hipMalloc handed to torch's caching allocator: every later allocation is carved out of it
    _arena = torch.empty(int(os.environ["ARENA_GIB"]) * 2**30, dtype=torch.uint8, device=dev)
    del _arena
N_EARLY = 202_383_360
early = [torch.randn(N_EARLY, device=dev) for _ in range(4)]
early[3].abs_()
early16 = torch.empty(N_EARLY, dtype=torch.bfloat16, device=dev)
one = torch.ones(1, device=dev)


def time_early(tag):
    for rep in range(3):
        s, e = ev(), ev()
        s.record()
        hip.adamw_step(early[0], early[1], early[2], early[3], early16, 2e-5, 0.9, 0.999, 1e-8, 0.0, 5, one)
        e.record()
        torch.cuda.synchronize()
    print(f"buffers allocated at process start, {tag}: {s.elapsed_time(e) * 1e3:7.1f} us  {N_EARLY * 30 / s.elapsed_time(e) / 1e9:5.2f} TB/s", flush=True)


time_early("before the model exists")
mla = bench.build(dev, 1)
time_early("after build()")
strat = FSDPStrategy(mla, 0, stage="finetune", global_batch_size=8, per_device_batch_size=8, learning_rate=2e-5, weight_decay=0.0,
                     max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=4)
strat.run_setup(n_train_examples=10_000)
time_early("after run_setup()")
print("torch reserved GiB", torch.cuda.memory_reserved() / 2**30, "allocated GiB", torch.cuda.memory_allocated() / 2**30)
sm = strat.sharded
for u in sm.units:
    if u.trainable:
        u.grad32.normal_()
big = [u for u in sm.units if u.trainable and u.n_train > 150_000_000][3]
print("unit", big.name, "n_train", big.n_train, "ptr % 4096:", [t.data_ptr() % 4096 for t in (big.master_train, big.grad32, big.exp_avg, big.exp_avg_sq, big.flat16)])
print("ptr >> 21 (2 MiB page index) mod 64:", [(t.data_ptr() >> 21) % 64 for t in (big.master_train, big.grad32, big.exp_avg, big.exp_avg_sq, big.flat16)])

for rep in range(3):
    s, e = ev(), ev()
    torch.cuda.synchronize()
    s.record()
    strat.clip_grad_norm()
    sm.optimizer_step(2e-5, weight_decay=0.0)
    e.record()
    torch.cuda.synchronize()
    print(f"clip + optimizer_step: {s.elapsed_time(e):7.2f} ms")
n = big.n_train
for rep in range(3):
    s, e = ev(), ev()
    s.record()
    hip.adamw_step(big.master_train, big.gshard, big.exp_avg, big.exp_avg_sq, big.flat16[:n], 2e-5, 0.9, 0.999, 1e-8, 0.0, 5, sm._coef)
    e.record()
    torch.cuda.synchronize()
    print(f"one unit ({n} elements): {s.elapsed_time(e) * 1e3:7.1f} us  {n * 30 / s.elapsed_time(e) / 1e9:5.2f} TB/s")
# the same launch on freshly allocated buffers of the same size
fresh = [torch.randn(n, device=dev) for _ in range(4)]
fresh[3].abs_()
f16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
for rep in range(3):
    s, e = ev(), ev()
    s.record()
    hip.adamw_step(fresh[0], fresh[1], fresh[2], fresh[3], f16, 2e-5, 0.9, 0.999, 1e-8, 0.0, 5, sm._coef)
    e.record()
    torch.cuda.synchronize()
    print(f"fresh buffers          : {s.elapsed_time(e) * 1e3:7.1f} us  {n * 30 / s.elapsed_time(e) / 1e9:5.2f} TB/s")
print("fresh ptr >> 21 mod 64:", [(t.data_ptr() >> 21) % 64 for t in fresh + [f16]])
