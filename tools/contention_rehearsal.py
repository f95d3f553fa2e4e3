"""CU-contention rehearsal for the 8-GPU run, on ONE GPU (VERDICT r3 next #3c; writes profiles/r4_contention.txt via gpurun_out/).

At N = 8 every backward sends 7/8 of the fp32 gradient buffer (23.6 GB at 7B) through RCCL's reduce-scatter kernels on the side stream,
per decoder layer, while the main stream runs the backward GEMMs one 512-thread workgroup per CU. This tool puts a stand-in for those
kernels (mla_side_traffic: k resident workgroups streaming a + b -> out over the layer's gradient buffer, throttled so that a layer's
transfer lasts about what a link-bound ring step would) exactly where ShardedModel launches its reduce-scatters -- from the decoder
layers' backward hooks, on a side stream, behind an event of the main stream, waited for before the gradient norm -- and measures the
configs[1] step for k in {0, 8, 16, 32} CUs, with the workgroups either SHARING their CUs with GEMM workgroups or TAKING them (160 KiB
of LDS each), and with the GEMMs planning their split-K tails for all 256 CUs or for the 256 - k that are left (mla_gemm_cus).
Round 6 (VERDICT r5 next #1c): --quick runs the six (k, sharing / taking) cases with the GEMMs planning for 256 CUs only and prints the
loss and gradient norm of the last step of every case as hex floats: the one-launch attention backward (attn_bwd_merged_kernel, whose
dK.dV workgroups WAIT for dQ workgroups of the same launch) runs next to the resident side-stream workgroups here for the first time.
tools/contention_merged.sh runs it twice (default and MLA_ATTN_BWD_MERGED=0) and compares -> profiles/r6_contention_merged.txt.
Usage: python tools/contention_rehearsal.py [--steps 6] [--ms-per-layer 2.5] [--quick]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mla_amd import hip
from mla_amd.strategy import FSDPStrategy
from mla_amd.synthetic import make_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--ms-per-layer", type=float, default=2.5, help="target duration of one layer's stand-in transfer (80 ms / 32 layers)")
    ap.add_argument("--quick", action="store_true", help="GEMM plan 256 only + hex loss / norm per case (merged-launch check)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(42)
    mla = bench.build(dev, 1)
    strat = FSDPStrategy(mla, 0, stage="finetune", global_batch_size=8, per_device_batch_size=8, learning_rate=2e-5, weight_decay=0.0,
                         max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=4)
    strat.run_setup(n_train_examples=10_000)
    sm = strat.sharded
    batch = make_batch(B=8, L_text=32, seed=42, device=dev, use_pointcloud=True)
    layers = [u for u in sm.units if getattr(u, "module", None) is not None and hasattr(u.module, "_grad_hook")]
    assert len(layers) == 32, len(layers)
    n_layer = layers[0].grad32.numel()
    n_move = (n_layer * 7 // 8) & ~3                       # what one rank of eight sends per layer
    other = torch.zeros(n_move, dtype=torch.float32, device=dev)
    sink = torch.empty(n_move, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)
    cfg = dict(k=0, lds=0, sleep=0)
    events, spans = [], []

    def hook(u):
        if cfg["k"] == 0:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            side.wait_event(ev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            hip.side_traffic(u.grad32[:n_move], other, sink, cfg["k"], cfg["lds"], cfg["sleep"])
            e1.record(side)
            spans.append((e0, e1))
            events.append(e1)
    for u in layers:
        u.module._grad_hook = (lambda uu=u: hook(uu))
    inner_finish = sm.finish_backward

    def finish():
        cur = torch.cuda.current_stream(dev)
        for e in events:
            cur.wait_event(e)
        events.clear()
        inner_finish()
    sm.finish_backward = finish

    last = {}

    def run(steps):
        spans.clear()
        strat.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last["out"] = strat.train_step(batch)
        strat.synchronize()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        busy = sum(a.elapsed_time(b) for a, b in spans) / steps if spans else 0.0
        return ms, busy

    def alone_ms(k, lds, ticks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        hip.side_traffic(layers[0].grad32[:n_move], other, sink, k, lds, ticks)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    def calibrate(k, lds):
        """largest sleep (ticks per 64 KiB chunk) that keeps one layer's transfer within ~ms_per_layer on an otherwise idle chip"""
        alone_ms(k, lds, 0)
        best, t_best = 0, alone_ms(k, lds, 0)
        for ticks in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):
            t = alone_ms(k, lds, ticks)
            if t > args.ms_per_layer:
                break
            best, t_best = ticks, t
        return best, t_best

    for _ in range(2):
        strat.train_step(batch)
    lines = []
    base = []
    for rep in range(2):
        cfg.update(k=0)
        base.append(run(args.steps)[0])
    lines.append(f"baseline (no side traffic): {base[0]:.1f} / {base[1]:.1f} ms per step")
    results = []
    for lds, label in ((0, "sharing CUs"), (160 * 1024, "taking CUs (160 KiB LDS each)")):
        for k in (8, 16, 32):
            ticks, t_alone = calibrate(k, lds)
            for plan in ((0,) if args.quick else (0, 256 - k)):
                hip.gemm_cus(plan)
                cfg.update(k=k, lds=lds, sleep=ticks)
                ms, busy = run(args.steps)
                hip.gemm_cus(0)
                if args.quick:
                    lines.append(f"    last step of this case: total_loss {float(last['out']['total_loss']).hex()} grad_norm {float(sm._norm).hex()}")
                results.append(dict(k=k, mode=label, gemm_planned_cus=plan or 256, sleep_ticks=ticks, layer_transfer_alone_ms=round(t_alone, 2), ms_per_step=round(ms, 1),
                                    side_stream_busy_ms_per_step=round(busy, 1), slowdown_pct=round(100 * (ms / min(base) - 1), 2)))
                lines.append(f"k = {k:2d} workgroups {label:30s} GEMMs planning for {plan or 256:3d} CUs: {ms:7.1f} ms per step ({100 * (ms / min(base) - 1):+5.2f} %), "
                             f"side stream busy {busy:6.1f} ms per step ({n_move * 12 * 32 / 1e9:.1f} GB of HBM traffic; one layer alone {t_alone:.2f} ms, sleep ticks {ticks})")
    cfg.update(k=0)
    tail = run(args.steps)[0]
    lines.append(f"baseline again: {tail:.1f} ms per step")
    print("\n".join(lines))
    lines.append(f"attention backward form: {'one launch (merged), dispatch probe ' + str(hip._DISPATCH_OK) if hip.ATTN_BWD_MERGED else 'two launches (MLA_ATTN_BWD_MERGED=0)'}")
    print(lines[-1])
    print(json.dumps(dict(baseline_ms=base + [tail], results=results, bytes_moved_per_layer=n_move * 4, note=__doc__.split("Usage")[0].strip()[:400])))


if __name__ == "__main__":
    main()
