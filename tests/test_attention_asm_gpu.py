"""The opt-in 32-rows-per-wave attention forward with the generated assembly iteration body (MLA_ATTN_FWD=1, attn_fwd32p_kernel,
tools/gen_attn_asm.py) against the same oracle and tolerances as the default kernel. The variant is read once per process, so the
parity cases of tests/test_kernels_gpu.py (forward AND the backward that consumes this forward's o / lse; ragged lengths, padding-only
blocks, S = 2048) run in a child process with the variable set; a second child proves the variable really selects another kernel
(same results within bf16 rounding, not the same bits).
Reference: transformers modeling_llama.py:371-380 (scaled causal softmax attention of LlamaAttention.forward)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DUMP = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from mla_amd import hip
dev = torch.device("cuda:0")
out = {}
for S, B, lens in ((548, 3, None), (548, 3, [548, 17, 300]), (1024, 2, [1024, 700]), (132, 2, None), (36, 2, None)):
    H, D = 3, 128
    g = torch.Generator().manual_seed(S + B)
    qkv = (torch.randn(B * S, 3 * H * D, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    sl = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    o, lse = hip.attn_fwd(qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:], B, S, H, D, 3 * H * D, sl, D ** -0.5)
    out[f"{S}_{B}_{'ragged' if lens else 'full'}"] = (o.cpu(), lse.cpu())
torch.save(out, sys.argv[1])
"""


def _child(env_extra, args, timeout=900):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_assembly_forward_passes_the_attention_parity_cases(dev):
    r = _child({"MLA_ATTN_FWD": "1"}, ["-m", "pytest", os.path.join("tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu", "-k",
                                       "test_attention_fwd_bwd or test_attention_full_size_config4_properties or test_attention_bwd_five_product"])
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail, tail


def test_the_variable_selects_another_kernel_with_the_same_results(dev, tmp_path):
    outs = {}
    for variant in ("0", "1"):
        f = tmp_path / f"v{variant}.pt"
        r = _child({"MLA_ATTN_FWD": variant}, ["-c", DUMP, str(f), ROOT])
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        outs[variant] = torch.load(f)
    differ = 0
    for key, (o0, l0) in outs["0"].items():
        o1, l1 = outs["1"][key]
        assert torch.equal(torch.isfinite(l0), torch.isfinite(l1)), key                    # padded rows: lse = +inf in both
        fin = torch.isfinite(l0)
        assert float((l0[fin] - l1[fin]).abs().max()) < 5e-3, key                           # log-sum-exp: fp32 either way
        rel = float((o0.float() - o1.float()).norm() / o0.float().norm())
        assert rel < 3e-3, (key, rel)                                                       # measured 1.3e-3 .. 2.0e-3: two bf16 roundings
        assert torch.equal(o0 == 0, o1 == 0) or float(((o0 == 0) != (o1 == 0)).float().mean()) < 1e-3, key
        differ += int(not torch.equal(o0, o1))
    assert differ > 0, "MLA_ATTN_FWD=1 produced the default kernel's bits: the assembly forward did not run"


def test_fused_backward_is_bit_identical_to_the_two_kernel_form(dev, tmp_path):
    """The opt-in one-workgroup-per-head backward (MLA_ATTN_BWD_FUSED=8, attn_bwd_fused_kernel: five products, every operand read once,
    S <= 576) against the default two-kernel, seven-product backward: tools/exp_attn_bits.py dumps dq | dk | dv, their transposes, o^T
    and the RoPE-fused forms for full, ragged and padding-only shapes in two child processes (the variant is read once per process);
    the reduction orders are the same by construction, so the comparison is bit for bit. Shapes above 576 fall back to the two-kernel form."""
    from mla_amd import hip
    if hip.lib().mla_query(3) != 1:
        pytest.skip("experiment kernels are not in the product build (mla_amd/csrc/build.sh with MLA_EXPERIMENTAL=1)")
    a, b = str(tmp_path / "two.pt"), str(tmp_path / "fused.pt")
    tool = os.path.join("tools", "exp_attn_bits.py")
    r0 = _child({"MLA_ATTN_BWD_FUSED": "0", "MLA_ATTN_BWD_MERGED": "0"}, [tool, a], timeout=600)
    r1 = _child({"MLA_ATTN_BWD_FUSED": "8"}, [tool, b], timeout=600)
    assert r0.returncode == 0 and r1.returncode == 0, (r0.stderr[-500:], r1.stderr[-500:])
    x, y = torch.load(a), torch.load(b)
    assert x.keys() == y.keys() and len(x) >= 40
    bad = [k for k in x if not torch.equal(x[k], y[k])]
    assert not bad, bad


def test_merged_backward_launch_is_bit_identical_to_two_launches(dev, tmp_path):
    """The default backward is ONE launch (attn_bwd_merged_kernel: the dQ blocks of a head publish delta through a per-head counter, the
    dK / dV blocks of the same head -- later workgroup ids on the same XCD -- wait for it) running the same block bodies as the two-launch
    form (MLA_ATTN_BWD_MERGED=0). Compared bit for bit on every tensor tools/exp_attn_bits.py dumps (full / ragged / padding-only
    shapes, RoPE-fused and transposed outputs; many launches of changing shape per process on ONE caller-owned counter buffer, which
    the self-resetting hand-off must leave zero every time), for the default order, for lag 1 (the consumers really do spin; a requested
    lag 0 is clamped to 1) and for the non-interleaved order."""
    tool = os.path.join("tools", "exp_attn_bits.py")
    ref = str(tmp_path / "two.pt")
    r0 = _child({"MLA_ATTN_BWD_MERGED": "0"}, [tool, ref], timeout=600)
    assert r0.returncode == 0, r0.stderr[-500:]
    x = torch.load(ref)
    assert len(x) >= 40
    for tag, env in (("default", {}), ("lag1", {"MLA_ATTN_BWD_MERGED": "1"}), ("lag3_plain", {"MLA_ATTN_BWD_MERGED": "3"}),
                     ("lag0_clamped_to_1", {"MLA_ATTN_BWD_MERGED": "100"})):
        out = str(tmp_path / f"{tag}.pt")
        r = _child(env, [tool, out], timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-500:])
        y = torch.load(out)
        assert x.keys() == y.keys()
        bad = [k for k in x if not torch.equal(x[k], y[k])]
        assert not bad, (tag, bad)


SEQ = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from mla_amd import hip
dev = torch.device("cuda:0")
out = {}
D = 128
# same padded head-group count with different real head counts, repeats of one shape, shrinking and growing shapes, ragged rows
shapes = [(2, 64, 2, None), (2, 100, 2, None), (2, 100, 3, [70, 100]), (2, 100, 3, [70, 100]), (1, 100, 3, None), (2, 100, 3, None),
          (4, 200, 8, [200, 1, 0, 137]), (1, 548, 2, None), (1, 548, 2, None), (5, 132, 7, None), (2, 100, 3, [100, 3]), (1, 1100, 9, None)]
for i, (B, S, H, lens) in enumerate(shapes):
    g = torch.Generator().manual_seed(100 + i)
    qkv = (torch.randn(B * S, 3 * H * D, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    do = torch.randn(B * S, H * D, generator=g).to(torch.bfloat16).to(dev)
    sl = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    o, lse = hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, sl, D ** -0.5)
    dqkv = torch.full_like(qkv, float("nan"))
    hip.attn_bwd(q, k, v, o, do, lse, sl, dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:], B, S, H, D, 3 * H * D, D ** -0.5)
    out[f"{i}_{B}_{S}_{H}"] = dqkv.cpu()
torch.save(out, sys.argv[1])
"""


def test_merged_launch_counters_survive_shape_changes(dev, tmp_path):
    """The merged launch's per-head counters (round 6: caller-owned, self-resetting -- the last consumer of a head stores zero to the
    head's pair) under a sequence of launches whose PADDED head-group count stays the same while the real head count changes (the case
    that left two heads' counters behind in round 5's first version: the consumers of those heads spun until the watchdog trap),
    repeats, shrinking and growing shapes, ragged and empty rows -- every gradient finite and bit-identical to the two-launch form."""
    outs = {}
    for tag, env in (("two", {"MLA_ATTN_BWD_MERGED": "0"}), ("merged", {})):
        f = tmp_path / f"{tag}.pt"
        r = _child(env, ["-c", SEQ, str(f), ROOT], timeout=600)
        assert r.returncode == 0, (tag, r.stdout[-500:] + r.stderr[-1500:])
        outs[tag] = torch.load(f)
    assert outs["two"].keys() == outs["merged"].keys()
    for key, a in outs["two"].items():
        assert torch.isfinite(a.float()).all(), key
        assert torch.equal(a, outs["merged"][key]), key


def _attn_case(dev, B, S, H, lens, seed):
    D = 128
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B * S, 3 * H * D, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    do = torch.randn(B * S, H * D, generator=g).to(torch.bfloat16).to(dev)
    sl = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    return qkv, do, sl


def test_dispatch_probe_holds_on_this_device(dev):
    """What attn_bwd_merged_kernel assumes and HIP does not promise (include/mla_hip.h: mla_dispatch_probe): 8 XCDs, workgroup L on XCD
    L & 7, workgroups started in id order per XCD (no start ticket more than two residency rounds of 64 slots out of place). hip.py hands
    head counters to mla_attn_bwd only where this holds; on MI355X it must, or the default backward silently became two launches."""
    from mla_amd import hip
    ok, info = hip.dispatch_probe(dev)
    print("dispatch probe:", info)
    assert info["tickets_complete"] and info["xcds"] == 8 and info["xcc_is_id_mod_8"], info
    assert info["worst_start_displacement"] <= 128, info
    assert ok
    # a second, much longer-resident grid: the order is a property of the queue, not of short workgroups
    ok2, info2 = hip.dispatch_probe(dev, blocks=2048, hold_us=200)
    assert ok2, info2


def test_merged_backward_in_process_counters_return_to_zero(dev):
    """merged=True vs merged=False in ONE process on one counter buffer: bit-identical gradients for full / ragged / empty rows and
    changing shapes, and the caller-owned head_sync buffer is all zeros after every launch (the self-resetting hand-off) -- which is what
    makes the entry point stateless (no library-owned memory, no epoch on the host)."""
    from mla_amd import hip
    D = 128
    shapes = [(2, 100, 3, [70, 100]), (4, 200, 8, [200, 1, 0, 137]), (1, 548, 2, None), (3, 548, 5, [548, 0, 64]), (1, 1100, 9, None), (2, 64, 2, None)]
    for i, (B, S, H, lens) in enumerate(shapes):
        qkv, do, sl = _attn_case(dev, B, S, H, lens, 300 + i)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
        o, lse = hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, sl, D ** -0.5)
        res = {}
        for merged in (False, True):
            dqkv = torch.full_like(qkv, float("nan"))
            hip.attn_bwd(q, k, v, o, do, lse, sl, dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:], B, S, H, D, 3 * H * D, D ** -0.5,
                         merged=merged)
            res[merged] = dqkv
        assert torch.isfinite(res[True].float()).all(), (B, S, H)
        assert torch.equal(res[True], res[False]), (B, S, H, lens)
        bufs = [b for (d, _), b in hip._HEAD_SYNC.items() if d == torch.cuda.current_device()]
        assert bufs and all(int(b.abs().sum()) == 0 for b in bufs), "head_sync not back to zero after the launch"


def test_merged_backward_replays_from_a_captured_graph(dev):
    """Advisor (round 5): the first merged launcher passed a host-computed epoch target as a kernel argument, so a captured launch
    replayed with a stale target and the dK / dV blocks stopped waiting -- silently wrong gradients. The round-6 hand-off has no
    host-side state: capture one backward, replay it three times over fresh inputs, every replay bit-identical to two launches."""
    from mla_amd import hip
    D, B, S, H = 128, 3, 548, 4
    qkv, do, sl = _attn_case(dev, B, S, H, [548, 300, 17], 77)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    o, lse = hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, sl, D ** -0.5)
    dqkv = torch.full_like(qkv, float("nan"))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hip.attn_bwd(q, k, v, o, do, lse, sl, dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:], B, S, H, D, 3 * H * D, D ** -0.5,
                     merged=True)                                     # warm-up on the capture stream: probe + counter buffer exist now
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hip.attn_bwd(q, k, v, o, do, lse, sl, dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:], B, S, H, D, 3 * H * D, D ** -0.5,
                     merged=True)
    for rep in range(3):
        qkv2, do2, _ = _attn_case(dev, B, S, H, None, 500 + rep)
        qkv.copy_(qkv2)
        do.copy_(do2)
        o2, lse2 = hip.attn_fwd(q, k, v, B, S, H, D, 3 * H * D, sl, D ** -0.5)
        o.copy_(o2)
        lse.copy_(lse2)
        dqkv.fill_(float("nan"))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = torch.full_like(qkv, float("nan"))
        hip.attn_bwd(q, k, v, o, do, lse, sl, want[:, :H * D], want[:, H * D:2 * H * D], want[:, 2 * H * D:], B, S, H, D, 3 * H * D, D ** -0.5,
                     merged=False)
        assert torch.equal(dqkv, want), rep
