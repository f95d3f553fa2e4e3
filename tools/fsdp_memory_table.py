"""Per-rank memory plan of the sharded 7B model at world 1 / 2 / 4 / 8 -- no allocation, the model lives on the meta device
(mla_amd.fsdp.plan_sharded_layout). Reproduces the table in DESIGN.md section 4; tests/test_fsdp_layout_7b.py asserts its invariants.
Usage: python tools/fsdp_memory_table.py [config: 1 | 3 | 4]"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")


def build_meta(config=1):
    import bench
    gen = config == 3
    stage = "post-training" if gen else ("pretrain" if config == 4 else "finetune")
    return bench.build("meta", 1, use_pointcloud=config in (1, 3), generation=gen, stage=stage)


def table(config=1, worlds=(1, 2, 4, 8)):
    from mla_amd.fsdp import plan_sharded_layout
    mla = build_meta(config)
    policy = mla.vlm.get_fsdp_wrapping_policy() if hasattr(mla, "vlm") and hasattr(mla.vlm, "get_fsdp_wrapping_policy") else mla.get_fsdp_wrapping_policy()
    rows = []
    for w in worlds:
        units = plan_sharded_layout(mla, policy, w)
        tot = {k: sum(u[k] for u in units) for k in ("bytes_bf16_replica", "bytes_grad32", "bytes_master", "bytes_moments")}
        layer = next(u for u in units if u["is_layer"])
        rows.append(dict(world=w, units=len(units), layers=sum(u["is_layer"] for u in units), **tot,
                         total=sum(tot.values()), layer_shard_bytes_bf16=2 * layer["shard_train"],
                         rs_out_bytes=sum(u["bytes_grad32"] for u in units) * (w - 1) // w,
                         ag_in_bytes=sum(2 * u["n_train"] for u in units) * (w - 1) // w))
    return rows


if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    G = 2.0 ** 30
    print(f"configs[{cfg if cfg != 1 else '1/2'}] per-rank persistent state (GiB), plan_sharded_layout on the meta device")
    print("world | bf16 replica | fp32 grad buffer | fp32 masters (1/N) | AdamW m+v (1/N) | state total | reduce-scatter out / step | all-gather in / step")
    for r in table(cfg):
        print(f"{r['world']:5d} | {r['bytes_bf16_replica'] / G:12.2f} | {r['bytes_grad32'] / G:16.2f} | {r['bytes_master'] / G:18.2f} | "
              f"{r['bytes_moments'] / G:15.2f} | {r['total'] / G:11.2f} | {r['rs_out_bytes'] / G:25.2f} | {r['ag_in_bytes'] / G:20.2f}")
