#!/usr/bin/env python
"""Latency of the inference path at 7B scale: MLA.predict_action_diff = 8-step DDIM, batch 1,
548-token sequence per step (672x672 image + 1024 points + prompt). Random-init weights, synthetic inputs.
    python tools/bench_infer.py [--steps 8] [--iters 5]
Prints one JSON line (not the driver's bench contract -- that is bench.py)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=1, help="future_action_window_size + 1")
    ap.add_argument("--no-reuse-prefix", action="store_true", help="the reference's control flow: a whole forward per DDIM step")
    args = ap.parse_args()
    from bench import build
    from mla_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = build(dev, 1)
    m.future_action_window_size = m.vlm.future_action_window_size = args.chunk - 1
    m.eval()
    for p in m.parameters():
        p.data = p.data.to(torch.bfloat16)
    b = make_batch(B=1, device=dev)
    ids = torch.cat([b["input_ids"][:, :-4], torch.tensor([[29871]], device=dev)], dim=1)   # prompt + the '▁' tag the splice looks for
    kw = dict(image=b["images"]["front_image"][0], pointcloud=b["point_cloud"][0], cur_robot_state=b["proprio"][0, 0].cpu().numpy(),
              input_ids=ids, num_ddim_steps=args.steps, reuse_prefix=not args.no_reuse_prefix)
    m.predict_action_diff(**kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        act = m.predict_action_diff(**kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    parts = {}
    if not args.no_reuse_prefix:
        # where the cached path's time goes: the prefill (encoders + one 545-row pass) and one graph replay over the suffix rows
        from mla_amd.infer import PrefixCachedEps
        mk = dict(input_ids=ids, images=b["images"]["front_image"][:1], point_cloud=b["point_cloud"][:1], camera_name="rlbench_front",
                  proprio=b["proprio"][:1])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        eng = PrefixCachedEps.for_inputs(m.vlm, n_action_rows=args.chunk, **mk)
        ev[1].record()
        x = torch.randn(1, args.chunk, 7, device=dev)
        t = torch.tensor([91], device=dev)
        eng(x, t)
        ev[2].record()
        for _ in range(8):
            eng._run()
        ev[3].record()
        torch.cuda.synchronize()
        parts = {"prefill_ms": round(ev[0].elapsed_time(ev[1]), 2), "one_eps_call_ms": round(ev[1].elapsed_time(ev[2]), 2),
                 "suffix_pass_graph_replay_ms": round(ev[2].elapsed_time(ev[3]) / 8, 3),
                 "weights_streamed_per_pass_gb": round(sum(p.numel() for l in m.vlm.llm_backbone.llm.model.layers for p in l.parameters()) * 2 / 1e9, 2),
                 "suffix_pass_weight_stream_tbps": round(sum(p.numel() for l in m.vlm.llm_backbone.llm.model.layers for p in l.parameters()) * 2 / 1e12 /
                                                         (ev[2].elapsed_time(ev[3]) / 8 * 1e-3), 2)}
    print(json.dumps({"metric": "predict_action_diff latency, MLA-Llama2-7B bf16, batch 1", "value": round(ms, 1), "unit": "ms",
                      "ddim_steps": args.steps, "ms_per_ddim_step": round(ms / args.steps, 1), "seq_len": int(ids.shape[1]) + 513 + 2 + args.chunk,
                      "action_chunk": args.chunk, "reuse_prefix": not args.no_reuse_prefix, **parts, "action": [round(float(v), 4) for v in act.reshape(-1)[:7]], "data": "synthetic"}))


if __name__ == "__main__":
    main()
