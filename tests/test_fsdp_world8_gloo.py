"""CPU, world_size = 8 over gloo: the 8-way sharding of mla_amd.fsdp.ShardedModel (BASELINE configs[2]: FSDP full-shard over the
8 GPUs of one node, training/strategies/fsdp.py:181-209, 308-310) checked BIT FOR BIT against an unsharded single-process AdamW.

What only an 8-way run exercises: `align = 8 * world` padding of the three regions, shard ranges that cross the decayed /
not-decayed boundary (FlatUnit._shard_ranges), shards that hold nothing but padding, 1/8 shards of the frozen region, the
reduce-scatter(mean) + scalar all-reduce + in-place all-gather sequence with eight participants, and the rank-0 streamed
checkpoint gather.

How bit-level equality is possible over a collective whose summation order is not specified: the per-rank gradients are small
integers / 64, so every partial sum (over ranks, and of squares for the norm) is exact in fp32 -- any order gives the same bits.
AdamW itself is elementwise, so running the SAME update function on the unsharded buffers is the reference. A second phase uses real
autograd gradients (order-dependent rounding) at allclose level, as the world-2 test does."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from test_fsdp_gloo import TorchLocalOps

WORLD = 8


class GroupedOps(TorchLocalOps):
    """+ the two-group launch the product's HipLocalOps offers: [decayed | not decayed] of one shard in a single call."""

    def adamw_groups(self, p32, g32, m, v, p16, n_decay, lr, betas, eps, wd, step, grad_scale):
        self.adamw(p32[:n_decay], g32[:n_decay], m[:n_decay], v[:n_decay], p16[:n_decay], lr, betas, eps, wd, step, grad_scale)
        self.adamw(p32[n_decay:], g32[n_decay:], m[n_decay:], v[n_decay:], p16[n_decay:], lr, betas, eps, 0.0, step, grad_scale)


class Block(nn.Module):
    """d = 20: the unit's decayed region is 2 x 400 -> padded to 832, the not-decayed one 3 x 24 -> n_train = 960, shard 120:
    rank 6's shard [720, 840) straddles the decay boundary at 832, rank 7's is 8 real + padding."""

    def __init__(self, d):
        super().__init__()
        self.self_attn = nn.Identity()
        self.a = nn.Linear(d, d, bias=False)
        self.b = nn.Linear(d, d, bias=True)
        self.norm = nn.LayerNorm(d)
        self._grad_hook = None

    def forward(self, x):
        return x + self.b(torch.tanh(self.a(self.norm(x))))


class Tower(nn.Module):
    """a unit that is entirely frozen, and one with a frozen tail of awkward size (13 x 7 = 91 elements)"""

    def __init__(self, d):
        super().__init__()
        self.lin = nn.Linear(7, d)
        self.tail = nn.Linear(7, 13)

    def forward(self, x):
        return self.lin(x) + self.tail(x).sum(-1, keepdim=True)


class Toy(nn.Module):
    def __init__(self, d=20, n=3):
        super().__init__()
        self.frozen_tower = Tower(d)
        self.tower = Tower(d)
        self.inp = nn.Linear(7, d)
        self.scale = nn.Parameter(torch.ones(3))            # 3 elements: a not-decayed root parameter smaller than one 8-slot
        self.layers = nn.ModuleList([Block(d) for _ in range(n)])
        self.out = nn.Linear(d, 5)

    def forward(self, x):
        h = self.inp(x) * self.scale.sum() + self.tower(x) + self.frozen_tower(x)
        for l in self.layers:
            h = l(h)
        return self.out(h)


def _make(seed=0):
    torch.manual_seed(seed)
    m = Toy()
    m.frozen_tower.requires_grad_(False)
    m.tower.tail.requires_grad_(False)
    return m


def _no_decay(n, p):
    return p.ndim <= 1 or n.endswith(".bias")


def _int_grad(shape, step, rank, pidx):
    g = torch.Generator().manual_seed(1000003 * step + 7919 * rank + pidx)
    return torch.randint(-4, 5, shape, generator=g).to(torch.float32) / 64.0


def _worker(rank, world, port, grouped, inplace, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mla_amd.fsdp import ShardedModel
        ops = GroupedOps() if grouped else TorchLocalOps()
        model = _make()
        sm = ShardedModel(model, lambda mod: isinstance(mod, (Block, Tower)), torch.device("cpu"), ops=ops, no_decay=_no_decay,
                          inplace_reduce=inplace)
        assert sm.grad_div == (float(world) if inplace else 1.0)
        names = [u.name for u in sm.units]
        assert names == ["frozen_tower", "tower", "<root>", "layers.0", "layers.1", "layers.2"], names
        assert not sm.units[0].trainable and sm.units[1].trainable
        # the layout facts this test is about
        blk = sm.units[3]
        assert (blk.n_decay, blk.n_train, blk.n_total, blk.shard_train) == (832, 960, 960, 120)
        lo = rank * blk.shard_train
        rr = blk._shard_ranges()
        if rank == 6:
            assert [(a, b, d) for a, b, _, d in rr] == [(0, 112, True), (112, 120, False)], rr     # straddles the boundary
        if rank == 7:
            assert [(a, b, d) for a, b, _, d in rr] == [(0, 120, False)], rr
        for u in sm.units:
            assert u.n_decay % (8 * world) == 0 and u.n_train % (8 * world) == 0 and u.n_total % (8 * world) == 0
            assert u.master_train.numel() * world == u.n_train and u.master_frozen.numel() * world == u.n_total - u.n_train
        tw = sm.units[1]
        assert tw.n_total - tw.n_train == 128 and tw.master_frozen.numel() == 16      # 91 + 13 frozen elements -> 2 x 64

        # ---- unsharded reference: fp32 masters + moments per parameter, the same elementwise update
        ref = _make()
        rp = {n: p.detach().clone().float() for n, p in ref.named_parameters()}
        rm = {n: torch.zeros_like(t) for n, t in rp.items()}
        rv = {n: torch.zeros_like(t) for n, t in rp.items()}
        order = [(n, p) for n, p in model.named_parameters()]
        lr, wd, betas, eps, max_norm = 1e-2, 0.1, (0.9, 0.999), 1e-8, 0.05
        ref_ops = TorchLocalOps()
        for step in range(1, 4):
            sm.begin_step()
            for pidx, (n, p) in enumerate(order):
                if p.requires_grad:
                    p.main_grad.copy_(_int_grad(p.shape, step, rank, pidx))
                    p._mg_touched = True
            sm.finish_backward()
            # what the reduce-scatter(mean) must have left in this rank's gradient shard: exact, so compare bits
            mean = {n: sum(_int_grad(p.shape, step, r, pidx) for r in range(world)) / world
                    for pidx, (n, p) in enumerate(order) if p.requires_grad}
            for u in sm.units:
                if not u.trainable:
                    continue
                full = torch.zeros(u.n_train)
                for n, p, o in u.params:
                    if p.requires_grad:
                        full[o:o + p.numel()] = mean[n].reshape(-1)
                # (in-place form: the shard is the rank's slice of the gradient buffer and holds the SUM; 1 / world lives in the scales)
                assert torch.equal(u.gshard / sm.grad_div, full[rank * u.shard_train:(rank + 1) * u.shard_train]), (u.name, "reduced shard")
            norm = sm.grad_norm_and_clip(max_norm)
            want_norm = torch.sqrt(sum((g.double() ** 2).sum() for g in mean.values()).float())
            assert torch.equal(norm.reshape(()), want_norm.reshape(())), (float(norm), float(want_norm))
            assert float(sm._coef) * sm.grad_div < 1.0                     # the clip is active
            sm.optimizer_step(lr, betas=betas, eps=eps, weight_decay=wd)
            for n, p in order:
                if not p.requires_grad:
                    continue
                p16 = torch.empty(p.shape, dtype=torch.bfloat16)
                ref_ops.adamw(rp[n], mean[n], rm[n], rv[n], p16, lr, betas, eps, 0.0 if _no_decay(n, p) else wd, step,
                              sm._coef * sm.grad_div)
            full = sm.full_state_dict_fp32()
            for n, p in order:
                assert torch.equal(full[n], rp[n]), (step, n, "fp32 master")
                assert p.dtype == torch.bfloat16 and torch.equal(p.detach(), rp[n].to(torch.bfloat16)), (step, n, "bf16 replica")
        # rank-0 streamed gather (checkpoint path): rank 0 gets host tensors, the others placeholders, same bits
        got = dict(sm.iter_full_state_fp32(to_cpu_on_rank0=True))
        if rank == 0:
            assert all(torch.equal(got[n], rp[n]) for n, _ in order)
        else:
            assert all(v is None for v in got.values())

        # ---- phase 2: real autograd gradients through the bf16 replica (order-dependent rounding: allclose)
        torch.manual_seed(0)
        opt_params = {n: rp[n].clone().requires_grad_(p.requires_grad) for n, p in order}
        decay = [opt_params[n] for n, p in order if p.requires_grad and not _no_decay(n, p)]
        nodecay = [opt_params[n] for n, p in order if p.requires_grad and _no_decay(n, p)]
        opt = torch.optim.AdamW([{"params": decay, "weight_decay": wd}, {"params": nodecay, "weight_decay": 0.0}], lr=lr)
        # (fresh moments on the reference side would diverge from the sharded state: rebuild the sharded model from the same weights)
        model2 = _make()
        model2.load_state_dict({n: rp[n] for n, _ in order})
        sm2 = ShardedModel(model2, lambda mod: isinstance(mod, (Block, Tower)), torch.device("cpu"), ops=ops, no_decay=_no_decay,
                           inplace_reduce=inplace)
        for step in range(2):
            sm2.begin_step()
            xs = [torch.randn(5, 7, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
            model2(xs[rank].to(torch.bfloat16)).float().pow(2).mean().backward()
            for p in model2.parameters():
                if p.requires_grad:
                    p.main_grad.copy_(p.grad.float())
                    p._mg_touched = True
                    p.grad = None
            sm2.finish_backward()
            norm = sm2.grad_norm_and_clip(0.5)
            sm2.optimizer_step(lr, weight_decay=wd)
            grads = None
            for r in range(world):
                shadow = _make()
                shadow.load_state_dict({n: t.detach() for n, t in opt_params.items()})
                shadow.to(torch.bfloat16)
                shadow(xs[r].to(torch.bfloat16)).float().pow(2).mean().backward()
                g = {n: (p.grad.float() if p.grad is not None else None) for n, p in shadow.named_parameters()}
                grads = g if grads is None else {n: (grads[n] + g[n] if g[n] is not None else None) for n in g}
            live = []
            for n, t in opt_params.items():
                t.grad = None if (grads[n] is None or not t.requires_grad) else grads[n] / world
                if t.grad is not None:
                    live.append(t)
            total = torch.nn.utils.clip_grad_norm_(live, 0.5)
            opt.step()
            assert abs(float(norm) - float(total)) < 1e-4 * max(1.0, float(total)), (float(norm), float(total))
            full = sm2.full_state_dict_fp32()
            for n, t in opt_params.items():
                assert torch.allclose(full[n], t.detach(), rtol=2e-5, atol=2e-6), (step, n)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grouped,inplace", [(False, False), (True, False), (True, True)],
                         ids=["adamw_per_range-mean_shards", "adamw_groups-mean_shards", "adamw_groups-inplace_sum_shards"])
def test_sharded_model_world8_gloo_bit_exact(grouped, inplace):
    """inplace = the RCCL path's bookkeeping (reduce-scatter as an in-place SUM into the rank's slice of the gradient buffer, 1 / world
    folded into the norm and into AdamW's gradient scale) run over gloo."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(WORLD, port, grouped, inplace, ret), nprocs=WORLD, join=True)
    assert dict(ret) == {r: "ok" for r in range(WORLD)}
