"""How far does the host run ahead of the GPU in the 7B step? Per step: host time spent inside train_step() (launch only, no
explicit sync) and the caching-allocator traffic that would force device-wide synchronisations (hipMalloc / hipFree / retries).
Usage: python tools/host_lead.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sys.argv = [sys.argv[0]]
device = torch.device("cuda:0")
torch.cuda.set_device(0)
from mla_amd.strategy import FSDPStrategy
from mla_amd.synthetic import make_batch

torch.manual_seed(42)
mla = bench.build(device, 1, False, use_pointcloud=True, generation=False, stage="finetune")
strat = FSDPStrategy(mla, 0, stage="finetune", global_batch_size=bench.B_PER_GPU, per_device_batch_size=bench.B_PER_GPU,
                     learning_rate=2e-5, weight_decay=0.0, max_grad_norm=1.0, lr_scheduler_type="constant",
                     enable_gradient_checkpointing=False, repeated_diffusion_steps=bench.R_DIFF)
strat.run_setup(n_train_examples=10_000)
batch = make_batch(B=bench.B_PER_GPU, L_text=bench.L_TEXT, seed=42, device=device, use_pointcloud=True, with_next=False)
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "num_ooms")
for _ in range(2):
    strat.train_step(batch)
torch.cuda.synchronize()
prev = torch.cuda.memory_stats()
t_all = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    strat.train_step(batch)
    host = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {i}: host time in train_step {host * 1e3:7.1f} ms; allocator deltas "
          + ", ".join(f"{k}={st.get(k, 0) - prev.get(k, 0)}" for k in keys)
          + f"; reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB")
    prev = st
torch.cuda.synchronize()
print(f"wall per step {(time.perf_counter() - t_all) / steps * 1e3:.1f} ms")
