"""RCCL-over-xGMI replacement for torch FSDP on the MLA training path.

Reference: training/strategies/fsdp.py:176-306 wraps the model in torch FSDP (HYBRID/FULL shard, MixedPrecision(param
bf16, reduce fp32), use_orig_params, one unit per LlamaDecoderLayer + tokenizers + projectors, model_mla.py:279-303).

MI355X-first redesign (DESIGN.md "sharding"):
* every unit owns ONE flat bf16 compute buffer (all ranks hold the full copy while computing -- 13.5 GB of 288 GB), one
  flat fp32 gradient buffer that the wgrad GEMM epilogues write into directly (``param.main_grad`` views; no zero-fill,
  no bf16->fp32 grad cast pass) and a 1/world shard of fp32 master weights + AdamW moments;
* per step and unit there is ONE bf16 all-gather (after the optimizer step, prefetched in forward order on a side stream
  and awaited right before the unit's first use) instead of FSDP's two (forward + pre-backward), and ONE fp32
  reduce-scatter (mean) launched on the side stream the moment the unit's backward kernels have been enqueued;
* q/k/v (and gate/up) weights are laid out back-to-back so the fused QKV / gate-up GEMMs read one contiguous operand;
* the fused AdamW kernel updates the fp32 shard and writes the refreshed bf16 shard in the same pass.

Local arithmetic (cast, AdamW, sum of squares) goes through an injected ``LocalOps``: the product uses HipLocalOps (HIP
kernels, no fallback); the CPU/gloo unit tests inject a torch implementation from tests/ to exercise the sharding logic.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


class HipLocalOps:
    """Product implementation: every call is a libmla_hip.so kernel on the current stream."""

    def __init__(self):
        from . import hip
        self.hip = hip

    def cast_to_bf16(self, src32, dst16):
        self.hip.cast_f32_to_bf16(src32, dst16)

    def adamw(self, p32, g32, m, v, p16, lr, betas, eps, wd, step, grad_scale):
        self.hip.adamw_step(p32, g32, m, v, p16, lr, betas[0], betas[1], eps, wd, step, grad_scale)

    def adamw_groups(self, p32, g32, m, v, p16, n_decay, lr, betas, eps, wd, step, grad_scale):
        self.hip.adamw_step_groups(p32, g32, m, v, p16, n_decay, lr, betas[0], betas[1], eps, wd, step, grad_scale)

    def sumsq(self, x32, out1, accumulate):
        self.hip.sumsq(x32, out1, accumulate)

    def sum_partials(self, partials, n, out1, accumulate):
        self.hip.sum_partials(partials, n, out1, accumulate)

    def clip_coef(self, sumsq1, max_norm, coef1, norm1):
        self.hip.clip_coef(sumsq1, max_norm, coef1, norm1)

    def stream(self, device):
        return torch.cuda.Stream(device=device)


class _SqArena:
    """Bump allocator over one fp32 buffer: every wgrad launch of a step takes the slice its sum-of-squares partials go to; the
    gradient norm then sums the used prefix in one launch. Reset at the start of every step."""

    def __init__(self, n: int, device):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.used = 0

    def take(self, count: int):
        if self.used + count > self.buf.numel():
            return None
        part = self.buf[self.used:self.used + count]
        self.used += count
        return part


def _round_up(n, m):
    return ((n + m - 1) // m) * m


def apply_rccl_env(environ=None) -> Dict[str, str]:
    """Knobs for the collectives' footprint on the chip, to be applied BEFORE the process group is created (bench.py does; a training
    script would call it next to its init_process_group). RCCL's kernels run on the side stream underneath the backward GEMMs, which
    launch one 512-thread workgroup per CU: every CU a channel occupies is a CU the GEMM rounds do not get.
      MLA_RCCL_MAX_CHANNELS=n  -> NCCL_MAX_NCHANNELS=n (RCCL honours the NCCL_* names): caps the workgroups per collective kernel;
      MLA_GEMM_CUS=n           -> read by libmla_hip.so itself: the GEMMs plan their split-K tails for n CUs instead of all of them.
    Returns what was set (for the bench line)."""
    import os as _os
    env = _os.environ if environ is None else environ
    done = {}
    v = env.get("MLA_RCCL_MAX_CHANNELS")
    if v:
        env["NCCL_MAX_NCHANNELS"] = str(int(v))
        done["NCCL_MAX_NCHANNELS"] = env["NCCL_MAX_NCHANNELS"]
    for k in ("MLA_GEMM_CUS", "MLA_FSDP_INPLACE_RS", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS"):
        if env.get(k) is not None:
            done[k] = env[k]
    return done


def default_no_decay(n: str, p) -> bool:
    """The reference's two AdamW parameter groups (training/strategies/fsdp.py:236-256): vectors and biases are not decayed."""
    return p.ndim <= 1 or n.endswith(".bias")


def plan_flat_layout(named_params: Sequence[Tuple[str, nn.Parameter]], world: int, no_decay: Callable = default_no_decay):
    """Where every parameter of ONE sharding unit sits in the unit's flat buffers -- pure arithmetic on shapes and requires_grad (works
    on meta-device modules): returns ``(params, n_decay, n_train, n_total)`` with ``params = [(name, parameter, element offset)]``.
    Order: [trainable & decayed | trainable & not decayed | frozen], each in module order (keeps q|k|v and gate|up adjacent); every
    tensor starts on an 8-element boundary (16 B in bf16, 32 B in fp32) and every region ends on a multiple of 8 x world, so that a
    1/world shard of the trainable or of the frozen region starts 16-B aligned in bf16 on every rank."""
    decay = [(n, p) for n, p in named_params if p.requires_grad and not no_decay(n, p)]
    nodecay = [(n, p) for n, p in named_params if p.requires_grad and no_decay(n, p)]
    frozen = [(n, p) for n, p in named_params if not p.requires_grad]
    align = 8 * world
    params: List[Tuple[str, nn.Parameter, int]] = []
    off = 0
    bounds = []
    for group in (decay, nodecay, frozen):
        for n, p in group:
            params.append((n, p, off))
            off += _round_up(p.numel(), 8)
        off = _round_up(off, align)
        bounds.append(off)
    return params, bounds[0], bounds[1], bounds[2]


def discover_units(model: nn.Module, unit_policy: Callable[[nn.Module], bool]):
    """The sharding units of a model in the order ShardedModel creates them: ``[(unit name, module or None, [(param name, param)])]``.
    Outermost matches of the policy are units; what no unit claims folds into the root unit; forward order = non-decoder units, root
    (the embeddings are needed first), decoder layers."""
    unit_mods: List[Tuple[str, nn.Module]] = []

    def walk(prefix, mod):
        for cn, child in mod.named_children():
            full = f"{prefix}.{cn}" if prefix else cn
            if unit_policy(child):
                unit_mods.append((full, child))
            else:
                walk(full, child)
    walk("", model)
    claimed = set()
    pending = []
    for name, mod in unit_mods:
        named = [(f"{name}.{n}", p) for n, p in mod.named_parameters() if id(p) not in claimed]
        claimed.update(id(p) for _, p in named)
        pending.append((name, mod, named))
    root_named = [(n, p) for n, p in model.named_parameters() if id(p) not in claimed]
    early = [x for x in pending if not hasattr(x[1], "self_attn")]
    layers = [x for x in pending if hasattr(x[1], "self_attn")]
    out = list(early)
    if root_named:
        out.append(("<root>", None, root_named))
    out.extend(layers)
    return [x for x in out if x[2]]


def plan_sharded_layout(model: nn.Module, unit_policy: Callable[[nn.Module], bool], world: int, no_decay: Callable = default_no_decay,
                        inplace_reduce: bool = True):
    """Per-unit layout and per-rank memory of ``ShardedModel(model, unit_policy)`` at ``world`` ranks WITHOUT allocating anything (the
    model may live on the meta device): the audit the 8-GPU run is planned with. Bytes per rank and unit:
    bf16 replica (whole unit: the all-gathered weights stay resident), fp32 gradient buffer (whole trainable region: the in-place
    reduce-scatter leaves the rank's shard inside it), fp32 master + two AdamW moments (1/world of the trainable region), fp32
    master of the frozen region (1/world). ``inplace_reduce=False`` adds the separate fp32 gradient shard of the gloo / out-of-place path."""
    units = []
    for name, mod, named in discover_units(model, unit_policy):
        params, n_decay, n_train, n_total = plan_flat_layout(named, world, no_decay)
        shard_train, shard_frozen = n_train // world, (n_total - n_train) // world
        ranges = []
        for rank in range(world):
            lo, hi = rank * shard_train, (rank + 1) * shard_train
            ranges.append([(max(a, lo) - lo, min(b, hi) - lo, max(a, lo), dec) for a, b, dec in ((0, n_decay, True), (n_decay, n_train, False))
                           if min(b, hi) > max(a, lo)])
        units.append(dict(name=name, is_layer=mod is not None and hasattr(mod, "self_attn"), params=params, n_decay=n_decay, n_train=n_train,
                          n_total=n_total, shard_train=shard_train, shard_frozen=shard_frozen, shard_ranges=ranges,
                          bytes_bf16_replica=2 * n_total, bytes_grad32=4 * n_train,
                          bytes_master=4 * (shard_train + shard_frozen), bytes_moments=8 * shard_train,
                          bytes_grad_shard=0 if (inplace_reduce or world == 1) else 4 * shard_train))
    return units


class _Arenas:
    """One allocation per KIND of buffer for all units of a model (bf16 replicas, fp32 gradient buffers, masters, the two AdamW moments,
    separate gradient shards): a unit's buffers are consecutive slices of them. Measured on MI355X (tools/exp_adamw_placement.py): the
    fused AdamW streams five buffers of a unit at once, and with the units' buffers allocated one by one in unit order (the five of a
    unit adjacent in the address space) it runs at 5.5 TB/s; with one arena per kind (the five streams of a unit several GB apart, as
    in the kernel's stand-alone benchmark) at 6.0 TB/s -- the in-step optimizer was the slow case in rounds 1-3."""

    ALIGN_BYTES = 2 << 20      # every slice starts on a 2 MiB boundary, like a separate allocation would: a weight matrix whose rows do
    #                            not start on 128-B lines costs the GEMMs' 1-KiB LDS-DMA reads an extra line each (measured: -3 % on the step)

    def __init__(self, sizes: Dict[str, int], device, slices: int = 0):
        dt = {"flat16": torch.bfloat16}
        pad = {k: slices * (self.ALIGN_BYTES // (2 if k == "flat16" else 4)) for k in sizes}
        self.buf = {k: torch.zeros(max(n + pad[k], 1), dtype=dt.get(k, torch.float32), device=device) for k, n in sizes.items() if n > 0}
        self.used = {k: 0 for k in self.buf}

    def has(self, kind: str) -> bool:
        return kind in self.buf

    def take(self, kind: str, n: int):
        if n == 0:
            return torch.zeros(0, dtype=torch.bfloat16 if kind == "flat16" else torch.float32, device=next(iter(self.buf.values())).device)
        gran = self.ALIGN_BYTES // self.buf[kind].element_size()
        o = (self.used[kind] + gran - 1) // gran * gran
        assert o + n <= self.buf[kind].numel(), f"arena {kind}: {o} + {n} > {self.buf[kind].numel()}"
        self.used[kind] = o + n
        return self.buf[kind][o:o + n]


class FlatUnit:
    """One sharding unit. Parameter order inside the flat buffers: [trainable & decayed | trainable & not decayed |
    frozen], each in module order (keeps q|k|v and gate|up adjacent), every tensor padded to 8 elements (plan_flat_layout)."""

    def __init__(self, name: str, named_params: Sequence[Tuple[str, nn.Parameter]], device, world: int, rank: int, ops,
                 no_decay: Callable[[str, nn.Parameter], bool], process_group=None, sync_from_rank0: bool = True,
                 collectives: Optional[bool] = None, inplace_reduce: bool = False, arenas: Optional["_Arenas"] = None):
        self.name, self.world, self.rank, self.ops, self.device = name, world, rank, ops, device
        # buffers come from per-kind arenas when the owner provides them (ShardedModel), else from torch one by one
        def take(kind, n, dtype):
            if arenas is not None and arenas.has(kind):
                return arenas.take(kind, n)
            return torch.zeros(n, dtype=dtype, device=device)
        coll = (world > 1) if collectives is None else collectives   # separate shard buffers + real collectives
        self.params, self.n_decay, self.n_train, self.n_total = plan_flat_layout(named_params, world, no_decay)
        self.shard_total = self.n_total // world
        self.shard_train = self.n_train // world
        self.trainable = self.n_train > 0

        # ---- buffers
        full32 = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        for n, p, o in self.params:
            full32[o:o + p.numel()].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
        if coll and sync_from_rank0:
            dist.broadcast(full32, src=0, group=process_group)      # every rank starts from rank 0's weights (FSDP sync_module_states)
        self.flat16 = take("flat16", self.n_total, torch.bfloat16)
        ops.cast_to_bf16(full32, self.flat16)
        # fp32 master weights: the trainable region and the frozen region are each sharded 1/world, so that the weights,
        # gradient shard and AdamW moments of one element always live on the same rank
        nf = (self.n_total - self.n_train) // world
        if not coll and not (arenas is not None and arenas.has("master")):
            self.master_train, self.master_frozen = full32[:self.n_train], full32[self.n_train:]
        else:
            r = rank if coll else 0
            st, sf = (self.shard_train, nf) if coll else (self.n_train, self.n_total - self.n_train)
            self.master_train = take("master", st, torch.float32)
            self.master_frozen = take("master", sf, torch.float32)
            self.master_train.copy_(full32[r * st:(r + 1) * st])
            self.master_frozen.copy_(full32[self.n_train + r * sf:self.n_train + (r + 1) * sf])
        del full32
        self.grad32 = take("grad32", self.n_train, torch.float32) if self.trainable else None
        # inplace_reduce (RCCL): the reduce-scatter writes this rank's shard INTO its own slice of the gradient buffer (NCCL's in-place
        # form, recvbuff == sendbuff + rank * count) as a SUM; the 1 / world of the mean is folded into the norm and into AdamW's
        # gradient scale (ShardedModel.grad_div). No separate shard buffer (27 GB / world at 7B), no copy when world == 1.
        self.inplace_reduce = bool(coll and inplace_reduce)
        if self.trainable:
            if not coll:
                self.gshard = self.grad32
            elif self.inplace_reduce:
                self.gshard = self.grad32[rank * self.shard_train:(rank + 1) * self.shard_train]
            else:
                self.gshard = take("gshard", self.shard_train, torch.float32)
            self.exp_avg = take("exp_avg", self.shard_train, torch.float32)
            self.exp_avg_sq = take("exp_avg_sq", self.shard_train, torch.float32)
        # ---- re-point the module parameters at the bf16 compute storage; install fp32 main_grad views
        # gradient-norm partials delivered by the wgrad GEMM epilogues (ops._gemm_into_main_grad): offset -> (numel, partials, count).
        # Only without collectives: under FSDP the norm is taken over the REDUCED shards, which no local epilogue has seen.
        self.sq_entries: Dict[int, Tuple[int, int, int]] = {}      # offset -> (numel, arena offset, partial count)
        self.sq_arena = None                                        # set by ShardedModel (one arena for all units)
        self._offset_of = {id(p): o for _, p, o in self.params}
        for n, p, o in self.params:
            p.data = self.flat16[o:o + p.numel()].view(p.shape)
            if p.requires_grad:
                p.main_grad = self.grad32[o:o + p.numel()].view(p.shape)
                p._mg_touched = False
                p._mg_dirty = False
                if not coll and hasattr(ops, "sum_partials"):
                    p._sq_sink = self._sq_sink
                    p._sq_invalidate = self._sq_invalidate_param
            p.grad = None
        self.gather_event = None
        self.rs_event = None

    def sq_invalidate(self, off: int, end: int) -> None:
        """main_grad[off:end) is about to be written by something that leaves no sum(dW^2) partials (a plain GEMM, an
        accumulate on an odd shape, an embedding / vector gradient): every recorded range that INTERSECTS it is stale -- dropped
        and its arena slice zeroed, so the norm pass reads those elements from the buffer again (uncovered_ranges)."""
        if not self.sq_entries:
            return
        for k in [k for k, e in self.sq_entries.items() if k < end and off < k + e[0]]:
            e = self.sq_entries.pop(k)
            self.sq_arena.buf[e[1]:e[1] + e[2]].zero_()

    def _sq_invalidate_param(self, p) -> None:
        off = self._offset_of.get(id(p))
        if off is not None:
            self.sq_invalidate(off, off + p.numel())

    def _sq_sink(self, weights, count):
        """A wgrad launch is about to write main_grad of `weights` (adjacent in the flat buffer) and to leave `count` sum(dW^2)
        partials of the final values: returns where they go (a slice of the model-wide arena, summed by ONE launch at clipping
        time) or None when the arena is full. A later launch that touches any part of the range (gradient accumulation, or a
        per-weight launch after a q|k|v-wide one) supersedes the earlier entries: their slices are zeroed."""
        arena = self.sq_arena
        if arena is None:
            return None
        off = self._offset_of[id(weights[0])]
        end = self._offset_of[id(weights[-1])] + weights[-1].numel()
        self.sq_invalidate(off, end)
        part = arena.take(count)
        if part is None:
            return None
        self.sq_entries[off] = (end - off, part.storage_offset(), count)
        return part

    def uncovered_ranges(self) -> List[Tuple[int, int]]:
        """[a, b) pieces of the trainable gradient buffer that no sq_entries range covers (norm weights, biases, embeddings, ...)."""
        out, pos = [], 0
        for off in sorted(self.sq_entries):
            n = self.sq_entries[off][0]
            assert off >= pos, "overlapping gradient-norm ranges"
            if off > pos:
                out.append((pos, off))
            pos = off + n
        if pos < self.n_train:
            out.append((pos, self.n_train))
        return out

    # shard-local [lo, hi) intersections with the decay / no-decay regions (optimizer launches)
    def _shard_ranges(self):
        lo = self.rank * self.shard_train
        hi = lo + self.shard_train
        out = []
        for a, b, decayed in ((0, self.n_decay, True), (self.n_decay, self.n_train, False)):
            s, e = max(a, lo), min(b, hi)
            if e > s:
                out.append((s - lo, e - lo, s, decayed))
        return out

    def begin_step(self):
        self.sq_entries = {}
        for _, p, _ in self.params:
            if p.requires_grad:
                p._mg_touched = False
                p._mg_locked = False
                if hasattr(p, "_mg_regions"):
                    p._mg_regions = {}       # per-region state of parameters used through ops.param_view

    def zero_stale_and_lock(self):
        """Called when the unit's reduce-scatter is about to be launched from its backward hook (see ShardedModel._launch_reduce_scatter)."""
        for _, p, _ in self.params:
            if not p.requires_grad:
                continue
            touched = p._mg_touched or any(getattr(p, "_mg_regions", {}).values())
            if not touched and p._mg_dirty and p.grad is None:
                p.main_grad.zero_()
                p._mg_dirty = False
            p._mg_locked = True

    def collect_autograd_grads(self):
        """Gradients delivered by autograd into ``.grad`` (small broadcast parameters: queries, mask token, position tables) move
        into main_grad; called after every backward so that accumulation over micro-batches happens in fp32."""
        for _, p, o in self.params:
            if p.requires_grad and p.grad is not None:
                self.sq_invalidate(o, o + p.numel())   # an autograd gradient lands on top of a wgrad epilogue's values
                if p._mg_touched:
                    p.main_grad.add_(p.grad.to(torch.float32))
                else:
                    p.main_grad.copy_(p.grad)
                p._mg_touched = True
                p.grad = None

    def finish_backward(self, already_reduced: bool = False):
        """Parameters that received no gradient this step but hold a stale one from an earlier step are zeroed
        (torch FSDP with use_orig_params presents zero gradients for them, and AdamW still applies to them).

        ``already_reduced``: the unit's reduce-scatter was launched from its backward hook (decoder layers). With the in-place
        collective the reduced SUM shard lives INSIDE grad32, so nothing may write main_grad any more -- a late autograd ``.grad``
        or a stale gradient that would need zeroing is an error here, not something to patch up on top of (or racing with) the
        collective's output (advisor, round 3). Only the dirty-flag bookkeeping runs."""
        if already_reduced:
            for n, p, _ in self.params:
                if not p.requires_grad:
                    continue
                if p.grad is not None:
                    raise RuntimeError(f"{self.name}: {n} received an autograd gradient after the unit's reduce-scatter was launched")
                if not p._mg_touched and any(getattr(p, "_mg_regions", {}).values()):
                    p._mg_touched = True
                if not p._mg_touched and p._mg_dirty:        # cannot happen since zero_stale_and_lock(): kept as an invariant check
                    raise RuntimeError(f"{self.name}: {n} holds a stale gradient that was reduced with this step's (untouched this step, "
                                       "but the unit's reduce-scatter fired from its backward hook)")
                if p._mg_touched:
                    p._mg_dirty = True
            return
        self.collect_autograd_grads()
        for _, p, _ in self.params:
            if p.requires_grad:
                if not p._mg_touched and any(getattr(p, "_mg_regions", {}).values()):
                    p._mg_touched = True     # written through its views (packed in_proj weights, conv weights used as matrices)
                if not p._mg_touched and p._mg_dirty:
                    p.main_grad.zero_()
                    p._mg_dirty = False
                elif p._mg_touched:
                    p._mg_dirty = True

    def state_bytes(self):
        n = self.flat16.numel() * 2 + (self.master_train.numel() + self.master_frozen.numel()) * 4
        if self.trainable:
            n += self.grad32.numel() * 4 + (self.exp_avg.numel() + self.exp_avg_sq.numel()) * 4
            if self.gshard is not self.grad32 and not self.inplace_reduce:
                n += self.gshard.numel() * 4
        return n


class ShardedModel:
    """Owns the FlatUnits of a model and runs the step-level collectives + optimizer."""

    def __init__(self, model: nn.Module, unit_policy: Callable[[nn.Module], bool], device, ops=None,
                 process_group=None, no_decay: Optional[Callable[[str, nn.Parameter], bool]] = None,
                 inplace_reduce: Optional[bool] = None):
        self.model, self.device = model, device
        self.pg = process_group
        import os
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # MLA_FORCE_COLLECTIVES=1 runs the sharded code path (separate shards, RCCL calls, side stream) even with one rank:
        # used to validate the collective plumbing on a single-GPU box
        self.coll = self.world > 1 or (os.environ.get("MLA_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized())
        self.ops = ops if ops is not None else HipLocalOps()
        # RCCL: in-place SUM reduce-scatter, mean folded into the scales (see FlatUnit). gloo (CPU tests) keeps separate mean shards.
        # (inplace_reduce=True with gloo is for the CPU tests: it runs the same SUM-shard bookkeeping over an in-place all-reduce)
        # MLA_FSDP_INPLACE_RS=0: back to an out-of-place reduce-scatter into separate shard buffers (the round-2 layout; a SUM like the
        # in-place form since round 6) -- a pre-wired fallback for the first multi-GPU run, should RCCL's in-place form misbehave there
        if inplace_reduce is None and os.environ.get("MLA_FSDP_INPLACE_RS") == "0":
            inplace_reduce = False
        self.inplace_reduce = bool(self.coll and (dist.get_backend(process_group) == "nccl" if inplace_reduce is None else inplace_reduce))
        # RCCL reduce-scatters are always SUMs, in place or not (round 6): ncclAvg is not used at all. The world-1 RCCL test
        # (tests/test_fsdp_rccl_world1_gpu.py) caught this image's RCCL 2.26.6 dropping the last 8 elements of an out-of-place AVG
        # reduce-scatter of 2^20 + 8 floats (output left untouched there; SUM of the same buffer is exact --
        # profiles/r6_rccl_avg_tail.txt), which silently froze the last bias of the root unit under MLA_FSDP_INPLACE_RS=0.
        self.sum_shards = bool(self.inplace_reduce or (self.coll and dist.get_backend(process_group) == "nccl"))
        self.grad_div = float(self.world) if self.sum_shards else 1.0         # gshard holds grad_div x the mean gradient
        no_decay = no_decay or default_no_decay   # fsdp.py:236-256
        # ---- unit discovery (outermost matches of the policy; the remainder folds into the root unit); forward order: non-decoder
        # units and the root first (embeddings are needed first), decoder layers after -- discover_units()
        self.units: List[FlatUnit] = []
        self.unit_of_module: Dict[int, FlatUnit] = {}
        found = discover_units(model, unit_policy)
        # one arena per kind of buffer, sized from the layout plan (see _Arenas: +10 % AdamW bandwidth against per-unit allocations);
        # MLA_FSDP_ARENAS=0 = the per-unit allocations of rounds 1-3 (A/B switch)
        self._arenas = None
        kinds = [k for k in os.environ.get("MLA_FSDP_ARENAS", "master,exp_avg,exp_avg_sq").split(",") if k and k != "0"]
        if kinds:
            sizes = dict(flat16=0, grad32=0, master=0, exp_avg=0, exp_avg_sq=0, gshard=0)
            for name, mod, named in found:
                _, n_decay, n_train, n_total = plan_flat_layout(named, self.world, no_decay)
                w = self.world if self.coll else 1
                sizes["flat16"] += n_total
                sizes["grad32"] += n_train
                sizes["master"] += n_total // w
                sizes["exp_avg"] += n_train // self.world
                sizes["exp_avg_sq"] += n_train // self.world
                if self.coll and not self.inplace_reduce:
                    sizes["gshard"] += n_train // self.world
            self._arenas = _Arenas({k: v for k, v in sizes.items() if k in kinds}, device, slices=2 * len(found))
        for name, mod, named in found:
            self._add_unit(name, mod, named, no_decay)
        # buffers (BatchNorm statistics, ...) just move to the device in fp32 (FSDP buffer_dtype fp32)
        for b in model.buffers():
            b.data = b.data.to(device)
        self.step_count = 0
        self.defer_reduce = False     # gradient accumulation: micro-batches before the last one only add into main_grad
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=device)
        # arena of the wgrad epilogues' sum-of-squares partials (FlatUnit._sq_sink): ~1.4 K floats per decoder-layer GEMM at 7B
        self._sq_arena = _SqArena(1 << 20, device) if (not self.coll and hasattr(self.ops, "sum_partials") and
                                                       self.device.type == "cuda") else None
        for u in self.units:
            u.sq_arena = self._sq_arena
        self._coef = torch.ones(1, dtype=torch.float32, device=device)
        self._norm = torch.zeros(1, dtype=torch.float32, device=device)
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            # every launch of libmla_hip.so goes to the CURRENT device's current stream (hip._stream): the owner of this object
            # must have made `device` current (FSDPStrategy.__init__ does), otherwise kernels would run on another device's stream
            assert torch.cuda.current_device() == (self.device.index or 0), \
                f"torch.cuda.set_device({self.device}) must be called before sharding a model onto it"
        # side stream for the collectives (reduce-scatter launched per unit from the backward, all-gather behind the optimizer).
        # NB (measured, round 2): running AdamW itself on a side stream underneath the next forward -- whole, per layer gated on the
        # forward reaching the layer in front, low stream priority, grid capped to 128..1024 workgroups -- never changed the step
        # time: the GEMMs slow down by exactly the AdamW time hidden (the chip is power-bound either way), so it stays inline.
        self.comm_stream = self.ops.stream(device) if self.coll else None
        self.wait_profile = None      # list of (kind, event, event) when a caller wants the main stream's collective stalls timed
        # consumption: a unit's forward waits for ITS all-gather only, so later layers keep streaming in behind the forward pass
        # instead of being waited for up front. Units without a module (root: embeddings, heads, embedders) are waited for in
        # begin_step AND by a pre-hook on every module that directly owns one of their parameters, so an eval forward /
        # predict_action_diff right after optimizer_step never reads half-gathered weights.
        if self.on_gpu and self.comm_stream is not None:
            for u in self.units:
                mod = getattr(u, "module", None)
                if mod is not None:
                    mod.register_forward_pre_hook(lambda m, inp, uu=u: self.wait_unit(uu))
                    if hasattr(mod, "_gather_wait"):      # the previous decoder layer reads this one's input_layernorm weight (ops.NormFoldIO)
                        mod._gather_wait = (lambda uu=u: self.wait_unit(uu))
                else:
                    owned = {id(p) for _, p, _ in u.params}
                    for m in model.modules():
                        if any(id(p) in owned for p in m.parameters(recurse=False)):
                            m.register_forward_pre_hook(lambda mm, inp, uu=u: self.wait_unit(uu))
        # reduce-scatter launch hooks on the decoder layers (fires when the layer's backward has been enqueued)
        for u in self.units:
            mod = getattr(u, "module", None)
            if mod is not None and hasattr(mod, "_grad_hook") and u.trainable and self.coll:
                mod._grad_hook = (lambda uu=u: self._launch_reduce_scatter(uu))

    def _add_unit(self, name, mod, named, no_decay):
        if not named:
            return
        u = FlatUnit(name, named, self.device, self.world, self.rank, self.ops, no_decay, self.pg, collectives=self.coll,
                     inplace_reduce=self.inplace_reduce, arenas=self._arenas)
        u.module = mod
        self.units.append(u)

    # ------------------------------------------------------------------------------------------ collectives
    def _launch_reduce_scatter(self, u: FlatUnit):
        if not self.coll or not u.trainable or u.rs_event is not None or self.defer_reduce:
            return
        # (advisor, round 4) Before the collective may read grad32: a parameter of this unit that got NO gradient in this window but
        # still holds one from an earlier step is zeroed HERE, on the compute stream, in front of the event the side stream waits for --
        # torch FSDP with use_orig_params presents zero gradients for unused parameters, and zeroing after the launch would race with
        # the in-place collective. From here on the unit's main_grad is locked: a later write (the same layer run through backward a
        # second time in one step, a late autograd .grad) raises in ops._touch / finish_backward instead of corrupting the reduced shard.
        u.zero_stale_and_lock()
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if cur is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._reduce_scatter(u)
                u.rs_event = torch.cuda.Event()
                u.rs_event.record(self.comm_stream)
        else:
            self._reduce_scatter(u)
            u.rs_event = True

    def _reduce_scatter(self, u: FlatUnit):
        backend = dist.get_backend(self.pg)
        if backend == "nccl":
            # SUM either way (in place: gshard = this rank's own slice of grad32; MLA_FSDP_INPLACE_RS=0: a separate shard buffer); the
            # mean's 1 / world lives in grad_div. Never ncclAvg: see __init__.
            dist.reduce_scatter_tensor(u.gshard, u.grad32, op=dist.ReduceOp.SUM, group=self.pg)
        else:  # gloo (CPU tests): all-reduce + slice
            dist.all_reduce(u.grad32, op=dist.ReduceOp.SUM, group=self.pg)
            if not u.inplace_reduce:            # (in place: gshard IS the rank's slice of grad32 and now holds the SUM)
                u.gshard.copy_(u.grad32[self.rank * u.shard_train:(self.rank + 1) * u.shard_train] / self.world)

    def _all_gather(self, u: FlatUnit):
        """bf16 all-gather of the trainable region (frozen weights never change)."""
        region = u.flat16[:u.n_train]
        shard = region[self.rank * u.shard_train:(self.rank + 1) * u.shard_train]
        if dist.get_backend(self.pg) == "nccl":
            # in place: the input is this rank's slice of the output (NCCL/RCCL in-place all-gather: sendbuff == recvbuff +
            # rank * count) -- no per-unit shard copy / allocation on the side stream every step
            dist.all_gather_into_tensor(region, shard, group=self.pg)
        else:
            parts = [torch.empty_like(shard) for _ in range(self.world)]
            dist.all_gather(parts, shard.clone(), group=self.pg)
            region.copy_(torch.cat(parts))

    def _timed_wait(self, kind: str, event) -> None:
        """main stream waits for `event`; with ``wait_profile`` set (bench.py --gpus N) the stall is bracketed by two events on the
        main stream, so the first multi-GPU run says how long the compute stream sat behind each kind of collective."""
        cur = torch.cuda.current_stream(self.device)
        prof = self.wait_profile
        if prof is None:
            cur.wait_event(event)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        cur.wait_event(event)
        e1.record(cur)
        prof.append((kind, e0, e1))

    def wait_profile_ms(self) -> Dict[str, float]:
        """Sum of the recorded stalls per kind (ms); clears the list. Call after a device synchronisation."""
        out: Dict[str, float] = {}
        for kind, e0, e1 in (self.wait_profile or []):
            out[kind] = out.get(kind, 0.0) + e0.elapsed_time(e1)
        if self.wait_profile is not None:
            self.wait_profile = []
        return out

    def wait_unit(self, u: FlatUnit):
        if u.gather_event is not None and u.gather_event is not True:
            self._timed_wait("gather_event", u.gather_event)
        u.gather_event = None

    def wait_all(self):
        """Main stream waits for every outstanding optimizer update / all-gather (checkpointing, evaluation, state access)."""
        for u in self.units:
            self.wait_unit(u)

    # ------------------------------------------------------------------------------------------ step API
    def begin_step(self):
        if self._sq_arena is not None:
            self._sq_arena.used = 0
        for u in self.units:
            u.begin_step()
            u.rs_event = None
        # the bf16 all-gathers of the previous optimizer step were issued in unit order on the side stream; units that own a
        # module wait in their forward pre-hook, the rest (root unit: embeddings, heads) here
        for u in self.units:
            if getattr(u, "module", None) is None or self.device.type != "cuda":
                self.wait_unit(u)

    def finish_micro_backward(self):
        """End of a backward that is not the last of its accumulation window: no reduction, no optimizer."""
        for u in self.units:
            if u.trainable:
                u.collect_autograd_grads()

    def finish_backward(self):
        if self.on_gpu:
            self.wait_all()   # normally long satisfied (every used unit waited in its forward)
        for u in self.units:
            if u.trainable:
                u.finish_backward(already_reduced=u.rs_event is not None)
        if self.coll:
            for u in self.units:
                if u.trainable and u.rs_event is None:
                    self._launch_reduce_scatter(u)
            if self.device.type == "cuda":
                cur = torch.cuda.current_stream(self.device)
                for u in self.units:
                    if u.trainable and u.rs_event is not None and u.rs_event is not True:
                        self._timed_wait("rs_event", u.rs_event)

    def grad_norm_and_clip(self, max_norm: Optional[float]):
        """Global L2 norm over the reduced gradient shards (+ scalar all-reduce), clip coefficient kept on device."""
        first = True
        small: List[torch.Tensor] = []
        for u in self.units:
            if not u.trainable:
                continue
            if u.sq_entries and not self.coll:
                # the big matrices' contributions come from their wgrad epilogues (arena, below); only what no such launch wrote
                # is read here: large pieces (embeddings) in place, the small ones (norm weights, biases) of all units together
                for a, b in u.uncovered_ranges():
                    if b - a > (1 << 20):
                        self.ops.sumsq(u.gshard[a:b], self._sumsq, not first)
                        first = False
                    else:
                        small.append(u.gshard[a:b])
            else:
                self.ops.sumsq(u.gshard, self._sumsq, not first)
                first = False
        if small:
            self.ops.sumsq(torch.cat(small) if len(small) > 1 else small[0], self._sumsq, not first)
            first = False
        if self._sq_arena is not None and self._sq_arena.used:
            self.ops.sum_partials(self._sq_arena.buf, self._sq_arena.used, self._sumsq, not first)    # every wgrad epilogue's partials
            first = False
        if self.coll:
            dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=self.pg)
        if self.grad_div != 1.0:
            self._sumsq.mul_(1.0 / (self.grad_div * self.grad_div))     # the shards hold SUMS over ranks: norm of the mean gradient
        self.ops.clip_coef(self._sumsq, float(max_norm) if max_norm is not None else 3.0e38, self._coef, self._norm)
        if self.grad_div != 1.0:
            self._coef.mul_(1.0 / self.grad_div)                        # AdamW's gradient scale = clip coefficient x 1 / world
        return self._norm

    def optimizer_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.step_count += 1

        for u in self.units:
            if not u.trainable:
                continue
            ranges = u._shard_ranges()
            if hasattr(self.ops, "adamw_groups") and ranges and ranges[0][0] == 0 and \
                    all(ranges[i][1] == ranges[i + 1][0] and ranges[i][2] + (ranges[i][1] - ranges[i][0]) == ranges[i + 1][2]
                        for i in range(len(ranges) - 1)):
                # the local shard is [decayed | not decayed] back to back: both parameter groups in one launch
                n_local = ranges[-1][1]
                n_dec = sum(le - ls for ls, le, _, dec in ranges if dec)
                g0 = ranges[0][2]
                self.ops.adamw_groups(u.master_train[:n_local], u.gshard[:n_local], u.exp_avg[:n_local], u.exp_avg_sq[:n_local],
                                      u.flat16[g0:g0 + n_local], n_dec, lr, betas, eps, weight_decay, self.step_count, self._coef)
            else:
                for ls, le, g0, decayed in ranges:
                    self.ops.adamw(u.master_train[ls:le], u.gshard[ls:le], u.exp_avg[ls:le], u.exp_avg_sq[ls:le],
                                   u.flat16[g0:g0 + (le - ls)], lr, betas, eps, weight_decay if decayed else 0.0, self.step_count,
                                   self._coef)
            if self.coll:
                if self.on_gpu:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self.comm_stream):
                        self.comm_stream.wait_event(ev)
                        self._all_gather(u)
                        u.gather_event = torch.cuda.Event()
                        u.gather_event.record(self.comm_stream)
                else:
                    self._all_gather(u)

    # ------------------------------------------------------------------------------------------ checkpoint helpers
    def iter_full_state_fp32(self, to_cpu_on_rank0: bool = False):
        """Yields ``(name, fp32 tensor)`` for every parameter, ONE UNIT AT A TIME: each unit's master shards are gathered (all ranks
        take part in the collectives), handed out, and freed before the next unit is touched, so the peak extra device memory is
        one unit (<= 0.8 GB for a 7B decoder layer), not the +27 GB of a whole fp32 model. With ``to_cpu_on_rank0`` only rank 0
        receives tensors (already copied to host memory); the other ranks get ``(name, None)`` -- the reference's
        FullStateDictConfig(offload_to_cpu=True, rank0_only=True), training/strategies/fsdp.py:107-110."""
        if self.on_gpu:
            self.wait_all()
        for u in self.units:
            full = self._gather_master(u)
            keep = (not to_cpu_on_rank0) or self.rank == 0
            for n, p, o in u.params:
                if not keep:
                    yield n, None
                    continue
                t = full[o:o + p.numel()].view(p.shape)
                yield n, (t.cpu() if to_cpu_on_rank0 else t.clone())
            del full

    def full_state_dict_fp32(self) -> Dict[str, torch.Tensor]:
        """Full fp32 state dict on the device of every rank (tests / small models; checkpoints use iter_full_state_fp32)."""
        return dict(self.iter_full_state_fp32())

    def _gather_master(self, u: FlatUnit):
        if not self.coll:
            return torch.cat([u.master_train, u.master_frozen])
        full = torch.empty(u.n_total, dtype=torch.float32, device=self.device)
        parts_t = [torch.empty(u.shard_train, dtype=torch.float32, device=self.device) for _ in range(self.world)]
        dist.all_gather(parts_t, u.master_train, group=self.pg)
        full[:u.n_train] = torch.cat(parts_t)
        nf = (u.n_total - u.n_train) // self.world
        if nf:
            parts_f = [torch.empty(nf, dtype=torch.float32, device=self.device) for _ in range(self.world)]
            dist.all_gather(parts_f, u.master_frozen, group=self.pg)
            full[u.n_train:] = torch.cat(parts_f)
        return full

    def state_bytes(self):
        return sum(u.state_bytes() for u in self.units)
