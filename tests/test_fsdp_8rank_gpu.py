"""EIGHT ranks on ONE GPU (gloo backend, CUDA tensors) running the real HIP kernels on the tiny MLA -- first-run insurance for the
8 x MI355X node (VERDICT r3 next #3a): the module structure the 7B run shards (decoder layers as units with backward hooks, vision
tower, projectors, root unit) has only ever been sharded two ways before this test; here every unit is cut into 8 shards (padding
shards, shards that straddle the decay boundary, shards of a frozen region all occur with the tiny model's odd sizes).

Checks, for a global batch of 8 samples (one per rank), two optimizer steps:
  (1) the reduced gradient shards of step 0, concatenated in rank order, equal the mean over the 8 ranks' local fp32 gradient buffers
      taken at reduce-scatter time: the collective + shard bookkeeping isolated from any kernel noise. gloo's all-reduce sums the
      eight buffers in its own (chunk-dependent ring) order, so "equal" is: every element within 4 fp32 ulps of the float64 mean
      (with two ranks the sum has one order and tests/test_fsdp_2rank_gpu.py asserts bit equality);
  (2) every rank ends with the same fp32 masters and bf16 replicas (bit for bit), and reports the same gradient norms;
  (3) against ONE process that takes the same 8 samples as an 8-micro-batch accumulation window (same per-rank shapes -> same kernels
      on the same rows; only the place and order of the fp32 sum differ): reduced gradients to 1e-6 (whole and worst matrix; measured 6e-8),
      fp32 masters after the first optimizer step to 1e-7 relative Frobenius (measured 1e-8) -- fp32-rounding level, nothing bf16-sized may hide
      here -- and after the second step at the bf16 level the second forward re-introduces.
Reference: training/strategies/fsdp.py:181-209 (FSDP full-shard wrapping), :308-310 (clip over the sharded gradients)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD, STEPS, R = 8, 2, 2


def _build(dev):
    from test_fsdp_2rank_gpu import _build as b2
    return b2(dev)


def _run(rank, world, dev):
    """world 8: this rank's sample; world 1: all 8 samples as 8 micro-batches of one accumulation window."""
    from mla_amd.strategy import FSDPStrategy
    from oracle import recipe
    m = _build(dev)
    strat = FSDPStrategy(m, dev.index or 0, global_batch_size=WORLD, per_device_batch_size=1, learning_rate=1e-3, weight_decay=0.01,
                         max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=R)
    assert strat.grad_accumulation_steps == (1 if world == WORLD else WORLD)
    strat.run_setup(100)
    batch, draws = recipe.make_batch(B=WORLD, R=R, ragged=False)
    orig = m.forward

    def micro(part):
        sel = slice(part, part + 1)
        rows = torch.tensor([part + WORLD * r for r in range(R)])          # rows of the R-tiled draws that belong to sample `part`
        b = {k: (v[sel] if torch.is_tensor(v) else v) for k, v in batch.items() if k not in ("images", "point_cloud")}
        b["images"] = {"front_image": batch["images"]["front_image"][sel]}
        m.forward = lambda **kw: orig(**kw, noise=draws["noise"][rows].to(dev), timestep=draws["timestep"][rows].to(dev))
        return strat.train_step(b)
    local0 = {}
    if world > 1:
        inner = strat.sharded._reduce_scatter

        def spy(u):
            if u.name not in local0:
                local0[u.name] = u.grad32.detach().clone()
            inner(u)
        strat.sharded._reduce_scatter = spy
    losses, norms, grads0 = [], [], None
    for _ in range(STEPS):
        if world > 1:
            out = micro(rank)
            losses.append(float(out["total_loss"]))
        else:
            tot = 0.0
            for part in range(WORLD):
                tot += float(micro(part)["total_loss"])
            losses.append(tot / WORLD)
        norms.append(float(strat.sharded._norm))
        if grads0 is None:
            torch.cuda.synchronize()
            div = np.float32(strat.sharded.grad_div)
            grads0 = dict(shards={u.name: u.gshard.detach().cpu().numpy().copy() / div for u in strat.sharded.units if u.trainable},
                          where={n: (u.name, o, p.numel()) for u in strat.sharded.units if u.trainable for n, p, o in u.params
                                 if p.requires_grad},
                          layout={u.name: dict(n_total=u.n_total, n_train=u.n_train, n_decay=u.n_decay, shard_train=u.shard_train)
                                  for u in strat.sharded.units})
            strat.synchronize()
            trainable = {k for k, p in m.named_parameters() if p.requires_grad}
            weights1 = {k: v.cpu().numpy() for k, v in strat.sharded.full_state_dict_fp32().items() if k in trainable}
    assert strat.step == STEPS
    full = strat.sharded.full_state_dict_fp32()
    compute = dict(m.named_parameters())
    keys = [k for k, p in compute.items() if p.requires_grad]
    return dict(losses=losses, norms=norms, grads0=grads0, local0={k: v.cpu().numpy() for k, v in local0.items()}, weights1=weights1,
                weights={k: full[k].cpu().numpy() for k in keys},
                compute={k: compute[k].detach().float().cpu().numpy() for k in keys})


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.set_device(0)                       # all eight ranks share device 0 (gloo moves the buffers through the host)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _run(rank, world, dev)
    finally:
        dist.destroy_process_group()


def _param_grads(*ranks):
    full = {name: np.concatenate([r["grads0"]["shards"][name] for r in ranks]) for name in ranks[0]["grads0"]["shards"]}
    return {n: full[u][o:o + k] for n, (u, o, k) in ranks[0]["grads0"]["where"].items()}


def test_eight_ranks_on_one_gpu_match_an_accumulation_window(dev):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, ret)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0, f"rank process failed (exit code {p.exitcode})"
    ranks = [ret[r] for r in range(WORLD)]

    # the layout really is an 8-way cut with the awkward cases present
    lay = ranks[0]["grads0"]["layout"]
    assert len(lay) >= 12, len(lay)
    assert all(v["n_total"] % (8 * WORLD) == 0 or v["n_total"] % WORLD == 0 for v in lay.values())
    straddle = [k for k, v in lay.items() if v["shard_train"] and 0 < v["n_decay"] < v["n_train"] and v["n_decay"] % v["shard_train"] != 0]
    assert straddle, "expected at least one unit whose decay boundary falls inside a rank's shard"

    # (1) the collective, isolated
    n_units, worst_ulp = 0, 0.0
    for name in ranks[0]["local0"]:
        loc = np.stack([r["local0"][name] for r in ranks]).astype(np.float64)
        want = loc.sum(0) / WORLD
        got = np.concatenate([r["grads0"]["shards"][name] for r in ranks]).astype(np.float64)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        # 7 fp32 additions in some order: error <= ~7 half-ulps of the largest partial sum; scale by the sum of magnitudes
        scale = np.abs(loc).sum(0) / WORLD
        ulp = np.spacing(np.maximum(scale, np.finfo(np.float32).tiny).astype(np.float32)).astype(np.float64)
        err = np.abs(got - want) / ulp
        worst_ulp = max(worst_ulp, float(err.max()))
        assert float(err.max()) <= 4.0, f"unit {name}: reduced shard differs from the rank mean by {float(err.max()):.1f} ulp"
        n_units += 1
    assert n_units >= 10, n_units
    print(f"reduce-scatter(mean) over 8 ranks: {n_units} units, reduced shards == float64 mean of the ranks' local fp32 buffers within {worst_ulp:.2f} fp32 ulp")

    # (2) every rank holds the same model
    for r in ranks[1:]:
        assert r["norms"] == ranks[0]["norms"]
        for k in ranks[0]["weights"]:
            assert np.array_equal(r["weights"][k], ranks[0]["weights"][k]), k
            assert np.array_equal(r["compute"][k], ranks[0]["compute"][k]), k

    # (3) one process, the same 8 samples as an accumulation window
    single = _run(0, 1, dev)
    ref, got = _param_grads(single), _param_grads(*ranks)
    assert ref.keys() == got.keys()
    num = sum(float(((got[n].astype(np.float64) - ref[n]) ** 2).sum()) for n in ref) ** 0.5
    den = sum(float((ref[n].astype(np.float64) ** 2).sum()) for n in ref) ** 0.5
    worst_n, worst = "", 0.0
    for n in ref:
        if ref[n].size < 4096:
            continue
        e = float(np.linalg.norm(got[n].astype(np.float64) - ref[n]) / (np.linalg.norm(ref[n]) + 1e-30))
        if e > worst:
            worst_n, worst = n, e
    def wrel(key):
        a = sum(float(((ranks[0][key][k].astype(np.float64) - single[key][k]) ** 2).sum()) for k in single[key]) ** 0.5
        return a / sum(float((single[key][k].astype(np.float64) ** 2).sum()) for k in single[key]) ** 0.5
    w1, w2 = wrel("weights1"), wrel("weights")
    dp_loss = [sum(r["losses"][st] for r in ranks) / WORLD for st in range(STEPS)]
    print(f"8 ranks vs one-process accumulation window of 8: reduced gradients Frobenius rel {num / den:.2e} (worst matrix {worst:.2e}, {worst_n}), "
          f"fp32 masters rel {w1:.2e} after the first optimizer step, {w2:.2e} after {STEPS}, grad norms {ranks[0]['norms']} vs {single['norms']}, losses {dp_loss} vs {single['losses']}")
    assert num / den < 1e-6 and worst < 1e-6, (num / den, worst, worst_n)          # measured 6.0e-8 / 7.6e-8
    # After ONE optimizer step the masters agree at fp32-rounding level. The second step starts from bf16 replicas of those masters:
    # a master that differs in its last fp32 bits can round to the neighbouring bf16 value, that moves activations by a bf16 ulp, and
    # AdamW's sign-like second step turns the resulting gradient noise into lr-sized differences on a few elements -- the same
    # mechanism as in the two-rank test's comparison (3); bound = bf16 level, measured 6.4e-5.
    assert w1 < 1e-7, w1                                                            # measured 1.0e-8
    assert w2 < 3e-4, w2
    for st in range(STEPS):
        # step 0: same kernels on the same rows, fp32-level; step 1 runs from bf16 replicas that may differ by an ulp (see above)
        tol = 2e-5 if st == 0 else 1e-2
        assert abs(ranks[0]["norms"][st] - single["norms"][st]) < tol * single["norms"][st], (st, ranks[0]["norms"], single["norms"])
        assert abs(dp_loss[st] - single["losses"][st]) < tol * max(1.0, abs(single["losses"][st])), (st, dp_loss, single["losses"])
