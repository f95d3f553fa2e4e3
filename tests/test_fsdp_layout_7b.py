"""CPU, no allocation: the 7B ShardedModel layout on the META device at world 2 / 4 / 8 (VERDICT r3 next #3b) -- the real module
structure (32 decoder layers as units, vision tower, point tower, projectors, root unit with embeddings / heads / embedders), audited
with the same arithmetic ShardedModel uses (mla_amd.fsdp.plan_sharded_layout / plan_flat_layout / discover_units):
  * every parameter starts on an 8-element boundary, every region boundary is a multiple of 8 x world, so every rank's shard of the
    trainable AND of the frozen region starts 16-B aligned in bf16 (RCCL in-place all-gather / reduce-scatter slices, 16-B vector
    accesses of the fused AdamW) -- for every real unit, not for a toy;
  * a rank whose shard straddles the decay boundary gets [decayed | not decayed] back to back with a boundary that is a multiple of 4
    (the one-launch mla_adamw_step_groups form), ranks entirely inside one region get one range, padding-only tails are covered;
  * q|k|v and gate|up are adjacent in every decoder layer (the fused QKV / gate|up GEMM operands), all 32 layers share one layout;
  * the per-rank memory plan (tools/fsdp_memory_table.py) leaves room for the activations inside 288 GB.
Reference: training/strategies/fsdp.py:181-209 (wrapping policy), :231-257 (parameter groups)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

H, I = 4096, 11008


@pytest.fixture(scope="module", params=[1, 3, 4])
def model(request):
    from fsdp_memory_table import build_meta
    return request.param, build_meta(request.param)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_7b_layout_invariants(model, world):
    from mla_amd.fsdp import plan_sharded_layout
    cfg, mla = model
    units = plan_sharded_layout(mla, mla.vlm.get_fsdp_wrapping_policy(), world)
    layers = [u for u in units if u["is_layer"]]
    assert len(layers) == 32
    names = [u["name"] for u in units]
    assert "<root>" in names and names.index("<root>") < names.index(layers[0]["name"])       # embeddings gathered before layer 0
    seen = set()
    n_straddle = 0
    for u in units:
        # -- parameters: disjoint, 8-element aligned, inside their region
        end_prev = 0
        for n, p, off in u["params"]:
            assert id(p) not in seen, f"{n} claimed by two units"
            seen.add(id(p))
            assert off % 8 == 0 and off >= end_prev, (u["name"], n, off)
            end_prev = off + p.numel()
            region_end = u["n_decay"] if (p.requires_grad and p.ndim > 1 and not n.endswith(".bias")) else (u["n_train"] if p.requires_grad else u["n_total"])
            assert end_prev <= region_end, (u["name"], n)
        # -- regions and shards
        for b in (u["n_decay"], u["n_train"], u["n_total"]):
            assert b % (8 * world) == 0
        assert u["shard_train"] * world == u["n_train"] and u["shard_frozen"] * world == u["n_total"] - u["n_train"]
        for rank in range(world):
            assert (rank * u["shard_train"] * 2) % 16 == 0 and (u["n_train"] * 2 + rank * u["shard_frozen"] * 2) % 16 == 0
            rr = u["shard_ranges"][rank]
            if u["shard_train"] == 0:
                assert rr == []
                continue
            # the ranges tile the local shard [0, shard_train) in order, decayed first
            assert rr[0][0] == 0 and rr[-1][1] == u["shard_train"] and all(rr[i][1] == rr[i + 1][0] for i in range(len(rr) - 1))
            assert [d for *_, d in rr] in ([True], [False], [True, False])
            for ls, le, g0, dec in rr:
                assert ls % 4 == 0 and le % 4 == 0 and g0 == rank * u["shard_train"] + ls       # vec4 AdamW + 16-B bf16 write-back
            n_straddle += len(rr) == 2
    assert len(seen) == sum(1 for _ in mla.parameters())
    assert n_straddle >= 1 or world == 2, "expected at least one rank whose shard holds the decay boundary"
    # -- decoder layers: one layout, fused-GEMM operands adjacent
    ref = [(n[len(layers[0]["name"]):], off, p.numel()) for n, p, off in layers[0]["params"]]
    for u in layers[1:]:
        assert [(n[len(u["name"]):], off, p.numel()) for n, p, off in u["params"]] == ref
    offs = {n[len(layers[0]["name"]) + 1:]: off for n, p, off in layers[0]["params"]}
    assert offs["self_attn.k_proj.weight"] == offs["self_attn.q_proj.weight"] + H * H
    assert offs["self_attn.v_proj.weight"] == offs["self_attn.k_proj.weight"] + H * H
    assert offs["mlp.up_proj.weight"] == offs["mlp.gate_proj.weight"] + I * H
    # a decoder layer is 202.4 M parameters: its bf16 shard per rank is what one in-place all-gather call moves
    assert layers[0]["n_train"] >= 4 * H * H + 3 * H * I + 2 * H


def test_7b_memory_plan_fits_the_part():
    from fsdp_memory_table import table
    G = 2.0 ** 30
    for cfg in (1, 3):
        rows = {r["world"]: r for r in table(cfg)}
        # persistent state shrinks with the world size only in its sharded part; replica + fp32 gradient buffer stay whole
        pad = 3 * 8 * 8 * rows[8]["units"] * 4          # three regions per unit, each padded to 8 x world elements, fp32
        assert 0 <= rows[8]["bytes_bf16_replica"] - rows[1]["bytes_bf16_replica"] <= pad and 0 <= rows[8]["bytes_grad32"] - rows[1]["bytes_grad32"] <= pad
        for w in (2, 4, 8):
            assert abs(rows[w]["bytes_moments"] * w - rows[1]["bytes_moments"]) <= 2 * pad
        # configs[1]: measured single-GPU peak 170 GB = 116 GiB state + ~57 GiB activations; at 8 ranks the state is
        # 48.5 GiB, so the same activations fit with > 150 GiB to spare -- no activation checkpointing needed at any world size
        assert rows[8]["total"] / G < 60 and rows[1]["total"] / G < 145
        # per step and rank: reduce-scatter sends (N-1)/N of the fp32 gradient buffer, the all-gather receives (N-1)/N of the bf16 weights
        assert abs(rows[8]["rs_out_bytes"] - rows[8]["bytes_grad32"] * 7 // 8) < 1024
        assert rows[8]["ag_in_bytes"] * 2 <= rows[8]["rs_out_bytes"] + 1024
