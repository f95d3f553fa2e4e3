"""Soak run: N optimizer steps of BASELINE configs[1] (7B, 8 samples x 4 repeats x 548 tokens) on ONE fixed synthetic batch with fresh
noise / timesteps every step: losses must stay finite and the diffusion + contrastive objective must go down (the model can only
over-fit the batch). Prints one line every 10 steps and a JSON summary. Usage: python tools/soak.py [steps] [lr]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mla_amd.strategy import FSDPStrategy
from mla_amd.synthetic import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-5
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(42)
mla = bench.build(dev, 1)
strat = FSDPStrategy(mla, 0, stage="finetune", global_batch_size=8, per_device_batch_size=8, learning_rate=lr, weight_decay=0.0,
                     max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=4)
strat.run_setup(n_train_examples=10_000)
batch = make_batch(B=8, L_text=32, seed=42, device=dev, use_pointcloud=True)
hist = []
t0 = time.time()
for i in range(steps):
    ld = strat.train_step(batch)
    if i % 10 == 0 or i == steps - 1:
        rec = {"step": i, "total": float(ld["total_loss"]), "contrastive": float(ld["img_pc_contrastive_loss"]),
               "grad_norm": float(strat.sharded._norm)}
        rec["diff_mse"] = rec["total"] - rec["contrastive"]
        hist.append(rec)
        print(rec, flush=True)
torch.cuda.synchronize()
ok = all(map(lambda r: all(abs(v) < 1e6 for v in r.values()), hist))
print(json.dumps({"steps": steps, "lr": lr, "wall_s": round(time.time() - t0, 1), "finite": ok, "first": hist[0], "last": hist[-1],
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
