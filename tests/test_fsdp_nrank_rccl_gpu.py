"""N = 2 / 4 / 8 ranks, ONE MI355X EACH, backend nccl (= RCCL over xGMI): the first test that executes the branches of
mla_amd/fsdp.py that will run on the 8-GPU node -- the in-place SUM `reduce_scatter_tensor` (recvbuff == sendbuff + rank * count) and
the in-place `all_gather_into_tensor` -- with world > 1, against the SAME ranks over gloo (all-reduce + slice, list all-gather: the
path every other N > 1 test of this repo takes). Skipped on boxes with fewer than N devices (the 1-GPU box of the round-end
`pytest -m gpu`); `tools/first_node_run.sh` runs it first thing on the node.

Checks per N, tiny MLA, one sample per rank, two optimizer steps (the accumulation-window scenario of test_fsdp_8rank_gpu.py):
  (1) RCCL's reduced gradient shards of step 0 == float64 mean of the ranks' local fp32 buffers within 4 fp32 ulp, and == gloo's reduced
      shards within 4 ulp (bit for bit at N = 2, where a sum has one order);
  (2) under RCCL every rank ends with bit-identical fp32 masters and bf16 replicas (the in-place all-gather delivered every slice);
  (3) RCCL vs gloo: fp32 masters after the first optimizer step to 1e-7 relative Frobenius, same gradient norms to 2e-5;
  (4) MLA_FSDP_INPLACE_RS=0 (out-of-place SUM reduce-scatter into separate shard buffers, the fallback knob; ncclAvg is never used: round 6) gives the same shards within 4 ulp.
Reference: training/strategies/fsdp.py:181-209 (FSDP full-shard wrapping), :308-310 (clip over the sharded gradients)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, backend, env, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_fsdp_8rank_gpu as t8
    t8.WORLD = world                                # the scenario's global batch = one sample per rank
    torch.cuda.set_device(rank)                     # one device per rank for BOTH backends: same kernels on the same hardware
    dev = torch.device("cuda", rank)
    if backend == "nccl":
        from mla_amd.fsdp import apply_rccl_env
        apply_rccl_env()
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = t8._run(rank, world, dev)
    finally:
        dist.destroy_process_group()


def _ranks(world, backend, env=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, env or {}, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0, f"{backend} rank process failed (exit code {p.exitcode})"
    return [ret[r] for r in range(world)]


def _ulp_err(got, want, scale):
    ulp = np.spacing(np.maximum(scale, np.finfo(np.float32).tiny).astype(np.float32)).astype(np.float64)
    return float((np.abs(got.astype(np.float64) - want) / ulp).max())


def _shards(ranks, name):
    return np.concatenate([r["grads0"]["shards"][name] for r in ranks])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_match_gloo_ranks(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL: one rank per device); this box has {torch.cuda.device_count()}")
    rccl = _ranks(world, "nccl")
    gloo = _ranks(world, "gloo")
    rccl_avg = _ranks(world, "nccl", {"MLA_FSDP_INPLACE_RS": "0"})
    worst = dict(mean=0.0, gloo=0.0, avg=0.0)
    for name in rccl[0]["local0"]:
        loc = np.stack([r["local0"][name] for r in rccl]).astype(np.float64)
        scale = np.abs(loc).sum(0) / world
        got = _shards(rccl, name)
        worst["mean"] = max(worst["mean"], _ulp_err(got, loc.sum(0) / world, scale))
        worst["gloo"] = max(worst["gloo"], _ulp_err(got, _shards(gloo, name).astype(np.float64), scale))
        worst["avg"] = max(worst["avg"], _ulp_err(got, _shards(rccl_avg, name).astype(np.float64), scale))
        if world == 2:
            assert np.array_equal(got, _shards(gloo, name)), f"unit {name}: two-rank RCCL and gloo sums differ"
    print(f"RCCL x{world}: reduced shards vs float64 mean {worst['mean']:.2f} ulp, vs gloo {worst['gloo']:.2f} ulp, "
          f"in-place vs out-of-place SUM {worst['avg']:.2f} ulp")
    assert max(worst.values()) <= 4.0, worst
    for r in rccl[1:]:                                                     # (2)
        assert r["norms"] == rccl[0]["norms"]
        for k in rccl[0]["weights"]:
            assert np.array_equal(r["weights"][k], rccl[0]["weights"][k]), k
            assert np.array_equal(r["compute"][k], rccl[0]["compute"][k]), k
    a, b = rccl[0]["weights1"], gloo[0]["weights1"]                        # (3)
    num = sum(float(((a[k].astype(np.float64) - b[k]) ** 2).sum()) for k in a) ** 0.5
    den = sum(float((b[k].astype(np.float64) ** 2).sum()) for k in a) ** 0.5
    print(f"RCCL x{world} vs gloo x{world}: fp32 masters after step 1 rel {num / den:.2e}, norms {rccl[0]['norms']} vs {gloo[0]['norms']}")
    assert num / den < 1e-7, num / den
    assert abs(rccl[0]["norms"][0] - gloo[0]["norms"][0]) < 2e-5 * gloo[0]["norms"][0]
