"""Two ranks on ONE GPU (gloo backend, CUDA tensors): the sharded training step with the real HIP kernels -- separate master / gradient
shards, side-stream reduce-scatter + all-gather with per-unit waits, sharded fused AdamW, global grad-norm -- against a
single-process run on the same global batch.

The model is the tiny MLA without the point tower (BatchNorm there normalises over the per-rank batch, so data-parallel and
single-process runs legitimately differ, in the reference as well) and without the contrastive head; every remaining op is
per-sample, so rank-averaged gradients must equal the single-process gradients up to bf16 rounding (kernel choice and split-K
depend on the number of rows, which moves individual activations by one bf16 ulp: measured 0.3 % on the loss after 9 layers).
(RCCL needs one device per rank; gloo moves the same buffers through the host, which is enough to validate the logic.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 2


def _build(dev):
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from oracle import recipe
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=1), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=False, use_contrastive=False,
                       use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=False, use_contrastive=False)
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.freeze_backbones("finetune")
    return m


def _run(rank, world, dev, accumulate=False):
    """Runs STEPS optimizer steps on this rank's slice of the global batch of 2; returns (losses, grad norms, weights).
    accumulate=True: one process, per-device batch 1, the two samples as two micro-batches of one accumulation window."""
    from mla_amd.strategy import FSDPStrategy
    from oracle import recipe
    R = 2
    m = _build(dev)
    split = world > 1 or accumulate
    strat = FSDPStrategy(m, dev.index or 0, global_batch_size=2, per_device_batch_size=1 if split else 2, learning_rate=1e-3, weight_decay=0.01,
                         max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=R)
    assert strat.grad_accumulation_steps == (2 if accumulate else 1)
    strat.run_setup(100)
    batch, draws = recipe.make_batch(B=2, R=R, ragged=False)
    orig = m.forward

    def micro(part):
        """part None: both samples; 0 / 1: that sample (and its rows of the tiled draws: order [s0, s1, s0, s1])."""
        sel = slice(None) if part is None else slice(part, part + 1)
        rows = torch.arange(2 * R) if part is None else torch.tensor([part, part + 2])
        b = {k: (v[sel] if torch.is_tensor(v) else v) for k, v in batch.items() if k not in ("images", "point_cloud")}
        b["images"] = {"front_image": batch["images"]["front_image"][sel]}
        m.forward = lambda **kw: orig(**kw, noise=draws["noise"][rows].to(dev), timestep=draws["timestep"][rows].to(dev))
        return strat.train_step(b)
    losses, norms = [], []
    grads0 = None
    local0 = {}
    if world > 1:
        # snapshot of every unit's LOCAL fp32 gradient buffer at the moment its reduce-scatter is launched (first step only): the
        # parent checks that the reduced shards are exactly the rank mean of these -- the collective itself, isolated from bf16 noise
        inner = strat.sharded._reduce_scatter

        def spy(u):
            if u.name not in local0:
                local0[u.name] = u.grad32.detach().clone()
            inner(u)
        strat.sharded._reduce_scatter = spy
    for _ in range(STEPS):
        if accumulate:
            first = micro(0)
            assert strat.step == len(losses)                      # no optimizer step inside the window
            out = micro(1)
            losses.append(0.5 * (float(first["total_loss"]) + float(out["total_loss"])))
        else:
            out = micro(None if world == 1 else rank)
            losses.append(float(out["total_loss"]))
        norms.append(float(strat.sharded._norm))
        if grads0 is None:
            # the REDUCED gradients of the first step, as the collectives left them: this rank's 1/world shard of every unit's fp32
            # gradient buffer (world 1: the whole buffer) + where each parameter sits in the unsharded buffer. AdamW only reads them.
            torch.cuda.synchronize()
            div = np.float32(strat.sharded.grad_div)     # RCCL path: in-place SUM shards, the mean's 1 / world lives in the scales
            grads0 = dict(shards={u.name: u.gshard.detach().cpu().numpy().copy() / div for u in strat.sharded.units if u.trainable},
                          where={n: (u.name, o, p.numel()) for u in strat.sharded.units if u.trainable for n, p, o in u.params
                                 if p.requires_grad})
        if world == 1:
            # the clipping norm is assembled from the wgrad epilogues' sum-of-squares partials (superseded per micro-batch inside an
            # accumulation window) + a pass over what they did not write: it must equal the norm of the fp32 gradient buffers
            direct = sum(float((u.grad32.double() ** 2).sum()) for u in strat.sharded.units if u.trainable) ** 0.5
            assert abs(norms[-1] - direct) < 1e-5 * direct, (norms[-1], direct)
    assert strat.step == STEPS
    full = strat.sharded.full_state_dict_fp32()
    keys = ("vlm.llm_backbone.llm.model.layers.3.mlp.down_proj.weight", "vlm.llm_backbone.llm.model.layers.0.self_attn.q_proj.weight",
            "vlm.projector_2d.mlp.2.weight", "vlm.final_layer.mlp.fc1.weight", "vlm.llm_backbone.llm.model.norm.weight",
            "vlm.x_embedder.mlp.fc1.bias")
    compute = dict(m.named_parameters())
    return dict(losses=losses, norms=norms, grads0=grads0, local0={k: v.cpu().numpy() for k, v in local0.items()},
                weights={k: full[k].cpu().numpy() for k in keys},
                compute={k: compute[k].detach().float().cpu().numpy() for k in keys})


def _worker(rank, world, port, ret, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    local = rank if backend == "nccl" else 0          # RCCL: one device per rank; gloo: both ranks share device 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _run(rank, world, dev)
    finally:
        dist.destroy_process_group()


def _two_ranks(backend):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, backend)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank process failed (exit code {p.exitcode})"
    return ret[0], ret[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per device (the 1-GPU box runs the gloo variant)")
def test_two_rccl_ranks_match_single_process(dev):
    """The same comparison over the REAL collective path: two processes, one MI355X each, backend nccl (= RCCL over xGMI):
    reduce_scatter_tensor(SUM, the mean folded into the scales) launched from the backward on the side stream, in-place all_gather_into_tensor behind the optimizer,
    scalar all-reduce of the gradient norm (training/strategies/fsdp.py:181-209, 308-310)."""
    r0, r1 = _two_ranks("nccl")
    _check_collective_exact(r0, r1)
    _compare_grads(_run(0, 1, dev, accumulate=True), (r0, r1), "rccl, vs accumulation window", ACC_GRAD_FRO, ACC_GRAD_FRO_WORST)
    _compare(_run(0, 1, dev), r0, r1, tag="rccl")


def test_two_ranks_match_single_process(dev):
    r0, r1 = _two_ranks("gloo")
    _check_collective_exact(r0, r1)
    _compare_grads(_run(0, 1, dev, accumulate=True), (r0, r1), "gloo, vs accumulation window", ACC_GRAD_FRO, ACC_GRAD_FRO_WORST)
    _compare(_run(0, 1, dev), r0, r1, tag="gloo")


# Bounds = 1.5 x the values measured on MI355X (printed on every run). Rank-averaged and single-process runs execute DIFFERENT kernel
# paths on purpose (half the rows per rank: other tile counts, other split-K tails, other attention grids), so individual bf16
# activations move by one ulp and the comparison is at bf16 level, not bit level; AdamW's first steps are sign-like (update ~ lr * sign(g)
# while v is small), which turns a gradient entry that flips sign within that noise into a 2 * lr difference of the weight.
# Measured (round 2, two kernel revisions -- the values move with every change of a kernel's rounding order, they are not noise of
# one build): loss 3.2e-3 / 4.6e-3, grad norm 3.0e-3 / 2.4e-3, min cos(update) 0.9913 / 0.9934, update norm 1.8e-3 / 7.3e-3.
# Bounds = 2-3 x the larger observation (round 1 asserted 1e-2 / 5e-2 / 0.9 / 0.1).
DP_LOSS_REL, DP_NORM_REL, DP_COS_MIN, DP_UPD_NORM_REL = 1.0e-2, 8e-3, 0.975, 2e-2
# The reduced gradients themselves -- what reduce-scatter(mean) / the accumulation window produce, before AdamW's sign-like first steps
# amplify anything: Frobenius distance of the WHOLE gradient (all trainable parameters) and of the worst single matrix, relative.
# Three comparisons, from exact to bf16-level:
#  (1) reduced shards == rank mean of the ranks' local buffers at reduce-scatter time: EXACT (the collective, isolated);
#  (2) reduced shards vs the one-process ACCUMULATION window over the same two samples -- the micro-batches have the per-rank shapes,
#      so both sides run the same kernels on the same data and differ only in where the fp32 sum is formed (GEMM epilogue `+= C` vs
#      the collective): fp32-rounding level, bound 1e-5 / 1e-4 (whole gradient / worst matrix);
#  (3) reduced shards vs the single-process run on the batch of two: different tile counts / split-K tails / attention grids move
#      individual bf16 activations by an ulp, and nine layers of that is a 1 % gradient difference. Measured on MI355X (round 3):
#      1.01e-2 whole gradient, 2.78e-2 worst matrix (layers.3 q_proj); the same figure for one decoder layer against the fp32
#      oracle is 0.7-1.1e-2 (test_decoder_layer_at_7b_dimensions), i.e. this is the bf16 floor, not a sharding artefact.
DP_GRAD_FRO, DP_GRAD_FRO_WORST = 2e-2, 5e-2
ACC_GRAD_FRO, ACC_GRAD_FRO_WORST = 1e-5, 1e-4


def _param_grads(*ranks):
    """Per-parameter fp32 gradients of step 0 from the ranks' shards (concatenated in rank order = the unsharded buffer)."""
    full = {name: np.concatenate([r["grads0"]["shards"][name] for r in ranks]) for name in ranks[0]["grads0"]["shards"]}
    return {n: full[u][o:o + k] for n, (u, o, k) in ranks[0]["grads0"]["where"].items()}


def _check_collective_exact(r0, r1):
    """(1) above: concat(rank shards) == (local_0 + local_1) / 2 bit for bit, for every trainable unit."""
    n = 0
    for name, l0 in r0["local0"].items():
        want = (l0.astype(np.float32) + r1["local0"][name]) * np.float32(0.5)
        got = np.concatenate([r0["grads0"]["shards"][name], r1["grads0"]["shards"][name]])
        assert np.array_equal(got, want), f"unit {name}: reduce-scatter(mean) result differs from the mean of the local buffers"
        n += 1
    assert n >= 10, n
    print(f"reduce-scatter(mean): {n} units, reduced shards == mean of the ranks' local fp32 buffers, bit for bit")


def _compare_grads(single, ranks, tag, fro=None, fro_worst=None):
    fro = DP_GRAD_FRO if fro is None else fro
    fro_worst = DP_GRAD_FRO_WORST if fro_worst is None else fro_worst
    ref, got = _param_grads(single), _param_grads(*ranks)
    assert ref.keys() == got.keys()
    num = sum(float(((got[n].astype(np.float64) - ref[n]) ** 2).sum()) for n in ref) ** 0.5
    den = sum(float((ref[n].astype(np.float64) ** 2).sum()) for n in ref) ** 0.5
    worst_n, worst = "", 0.0
    for n in ref:
        if ref[n].size < 4096:
            continue                                   # vectors (norm weights, biases): covered by the global figure
        e = float(np.linalg.norm(got[n].astype(np.float64) - ref[n]) / (np.linalg.norm(ref[n]) + 1e-30))
        if e > worst:
            worst_n, worst = n, e
    print(f"reduced gradients ({tag}) vs single-process fp32 gradient buffer: Frobenius rel {num / den:.2e} over {len(ref)} tensors, "
          f"worst matrix {worst:.2e} ({worst_n})")
    assert num / den < fro and worst < fro_worst, (num / den, worst, worst_n)


def _compare(single, r0, r1, tag):
    from oracle import recipe
    _compare_grads(single, (r0,) if r0 is r1 else (r0, r1), tag)
    # every rank ends with the same weights (fp32 masters after gathering the shards, and the bf16 compute copies)
    for k in r0["weights"]:
        assert np.array_equal(r0["weights"][k], r1["weights"][k]), k
        assert np.array_equal(r0["compute"][k], r1["compute"][k]), k
        assert np.allclose(r0["compute"][k], r0["weights"][k], rtol=1e-2, atol=1e-3), k     # bf16 copy of the master
    assert r0["norms"] == r1["norms"]
    # data-parallel == single process: mean of the per-rank losses, global gradient norm, updated weights
    worst = dict(loss=0.0, norm=0.0, cos=1.0, upd=0.0)
    for st in range(STEPS):
        dp_loss = 0.5 * (r0["losses"][st] + r1["losses"][st])
        worst["loss"] = max(worst["loss"], abs(dp_loss - single["losses"][st]) / max(1.0, abs(single["losses"][st])))
        worst["norm"] = max(worst["norm"], abs(r0["norms"][st] - single["norms"][st]) / single["norms"][st])
    init = {k: recipe.det_weight(k, v.shape).numpy() for k, v in r0["weights"].items()}
    for k in r0["weights"]:
        du2, du1 = r0["weights"][k] - init[k], single["weights"][k] - init[k]
        cos = float((du2 * du1).sum() / (np.linalg.norm(du2) * np.linalg.norm(du1) + 1e-30))
        worst["cos"] = min(worst["cos"], cos)
        worst["upd"] = max(worst["upd"], abs(np.linalg.norm(du2) / np.linalg.norm(du1) - 1))
    print(f"2 ranks ({tag}) vs single process over {STEPS} steps: loss rel {worst['loss']:.2e}, grad-norm rel {worst['norm']:.2e}, "
          f"min cos(update) {worst['cos']:.4f}, update-norm rel {worst['upd']:.2e}")
    assert worst["loss"] < DP_LOSS_REL and worst["norm"] < DP_NORM_REL and worst["cos"] > DP_COS_MIN and worst["upd"] < DP_UPD_NORM_REL, worst


def test_gradient_accumulation_matches_one_big_batch(dev):
    """grad_accumulation_steps = 2 (base_strategy_mla.py:100, :365-377): two micro-batches of one sample, loss / 2, one clip + AdamW
    step per window == one step on both samples, up to bf16 rounding (same bounds as the data-parallel comparison above)."""
    single = _run(0, 1, dev)
    acc = _run(0, 1, dev, accumulate=True)
    _compare(single, acc, acc, tag="accumulation window of 2")


def test_accumulation_window_with_changing_token_counts_keeps_the_clip_norm_right(dev):
    """Gradient-norm partials vs accumulation (round-2 advisor finding): micro-batch 1 has T = 2 x 544 = 1088 tokens (T % 64 == 0, so the
    wgrad launches leave sum(dW^2) partials), micro-batch 2 is a shorter sample whose wgrad launches are plain accumulates WITHOUT
    partials (until round 5 because its T = 1080 was not a multiple of 64; round 6 pads a layer's token rows to 64, so the test
    switches the partials off for that micro-batch). The partials of micro-batch 1 describe values that micro-batch 2
    has since added to -- they must be dropped, and the clipping norm must equal the norm of the fp32 gradient buffers."""
    from mla_amd import ops
    from mla_amd.strategy import FSDPStrategy
    from oracle import recipe
    R = 2
    m = _build(dev)
    strat = FSDPStrategy(m, dev.index or 0, global_batch_size=2, per_device_batch_size=1, learning_rate=1e-3, weight_decay=0.01,
                         max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=R)
    assert strat.grad_accumulation_steps == 2
    strat.run_setup(100)
    batch, draws = recipe.make_batch(B=2, L=28, R=R, ragged=False)
    lens = [28, 24]
    orig = m.forward

    def micro(part):
        L = lens[part]
        b = {k: (v[part:part + 1] if torch.is_tensor(v) else v) for k, v in batch.items() if k not in ("images", "point_cloud")}
        for k in ("input_ids", "attention_mask", "labels"):
            b[k] = b[k][:, :L].clone()
        b["input_ids"][0, L - 1] = 2                      # </s> closes the (shortened) prompt
        b["labels"][:] = -100
        b["labels"][0, L - 1] = 2
        b["images"] = {"front_image": batch["images"]["front_image"][part:part + 1]}
        rows = torch.tensor([part, part + 2])
        m.forward = lambda **kw: orig(**kw, noise=draws["noise"][rows].to(dev), timestep=draws["timestep"][rows].to(dev))
        return strat.train_step(b)

    for step in range(2):
        micro(0)
        layer_units = [u for u in strat.sharded.units if u.name.endswith("layers.3")]
        assert layer_units and len(layer_units[0].sq_entries) >= 4, "micro-batch 1 should have registered wgrad-epilogue partials"
        ops._WGRAD_SQ = False                             # micro-batch 2: plain accumulate launches (no sum(dW^2) partials)
        try:
            micro(1)
        finally:
            ops._WGRAD_SQ = True
        assert not layer_units[0].sq_entries, "micro-batch 2 wrote the same ranges without partials: the entries must be gone"
        got = float(strat.sharded._norm)
        direct = sum(float((u.grad32.double() ** 2).sum()) for u in strat.sharded.units if u.trainable) ** 0.5
        print(f"step {step}: clip norm {got:.6f} vs norm of the gradient buffers {direct:.6f}")
        assert abs(got - direct) < 1e-5 * direct, (got, direct)
