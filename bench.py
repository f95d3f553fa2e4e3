#!/usr/bin/env python
"""Headline benchmark: MLA-Llama2-7B SFT training step (bf16) on N MI355X -- BASELINE.json configs[1] (N=1) / [2] (N=8).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one micro-step of the reference's training loop (base_strategy_mla.py:303-379): forward + backward of the whole
MLA model on a synthetic batch of 8 samples per GPU (x4 diffusion repeats = 32 sequences of 548 tokens), global grad-norm
clip, AdamW step. Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (MFMA GEMM family, timed
with HIP events around every launch on the launch stream) and `cpu_baseline` (the oracle's decoder layer on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
S_FUSED, L_TEXT, R_DIFF, B_PER_GPU = 513, 32, 4, 8


def model_flops_per_sample(S, H=4096, I=11008, L=32, V=32064, R=R_DIFF):
    """per token per layer fwd = 8 H^2 + 6 H I + causal attention 2*2*(S/2)*H; model FLOPs = 3x fwd
    (no recompute credit); + lm_head fwd. Returns (decoder_flops, total_flops) per dataset sample."""
    per_tok_layer = 8 * H * H + 6 * H * I + 2 * 2 * (S / 2) * H
    dec = 3 * per_tok_layer * L * S * R
    lm = 2 * H * V * S * R
    return dec, dec + lm


class BoxSampler:
    """Shader clock and socket power of this rank's GPU while the timed steps run (~20 Hz, a daemon thread): what makes one box's
    `value` comparable with another's (VERDICT r5 next #4 -- the step's GEMMs run at the board power cap, and the same build spans
    567-598 ms per step across the pool's boxes). Source: the amdsmi Python binding of this image (gpu_metrics: current_gfxclk,
    current / average socket power), else the amdgpu hwmon files in sysfs; `source` and any error travel into the `box` block."""

    def __init__(self, dev_index=0, period_s=0.05):
        import threading
        self.dev_index, self.period = dev_index, period_s
        self.rows, self.source, self.error = [], None, None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._read = self._open()

    def _open(self):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[self.dev_index]

            def read():
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clk = m.get("current_gfxclk")
                if not isinstance(clk, (int, float)) or clk <= 0 or clk >= 65535:
                    cl = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 65535]
                    clk = sum(cl) / len(cl) if cl else None
                pw = next((m[k] for k in ("current_socket_power", "average_socket_power") if isinstance(m.get(k), (int, float)) and 0 < m[k] < 65535), None)
                return clk, pw
            read()
            self.source = "amdsmi gpu_metrics (current_gfxclk MHz, socket power W)"
            return read
        except Exception as e:   # noqa: BLE001 -- any failure of the binding falls through to sysfs
            self.error = f"amdsmi: {e!r}"[:200]
        try:
            import glob
            hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            hw = [d for d in hw if os.path.exists(os.path.join(d, "freq1_input"))]
            d = hw[min(self.dev_index, len(hw) - 1)]
            pfile = next(f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f)))

            def read():
                return int(open(os.path.join(d, "freq1_input")).read()) / 1e6, int(open(os.path.join(d, pfile)).read()) / 1e6
            read()
            self.source = f"sysfs {d} (freq1_input, {pfile})"
            return read
        except Exception as e:   # noqa: BLE001
            self.error = (self.error or "") + f" | sysfs: {e!r}"[:200]
        return None

    def _run(self):
        while not self._stop.is_set():
            try:
                self.rows.append(self._read())
            except Exception as e:   # noqa: BLE001
                self.error = f"read: {e!r}"[:200]
                return
            self._stop.wait(self.period)

    def start(self):
        if self._read is not None:
            self._thread.start()

    def stop(self):
        import statistics
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(1.0)
        clk = [c for c, _ in self.rows if c]
        pw = [p for _, p in self.rows if p]
        return {"sclk_mhz_median": round(statistics.median(clk)) if clk else None, "sclk_mhz_min_max": [round(min(clk)), round(max(clk))] if clk else None,
                "socket_power_w_median": round(statistics.median(pw)) if pw else None, "socket_power_w_max": round(max(pw)) if pw else None,
                "samples": len(self.rows), "period_s": self.period, "source": self.source, **({"sampler_error": self.error} if self.error else {})}


def step_weighted_mfma_util(ms_per_step):
    """sum over kernels of MfmaUtil x time, over the step time: the matrix pipe's busy share of the WHOLE step. MfmaUtil and the
    per-kernel durations come from the newest committed counter table (profiles/r*_pmc_table.json, separate rocprofv3 --pmc passes over
    this command: tools/collect_counters.sh -> tools/pmc_table.py), the step time is this run's; kernels the table does not list and idle
    time count as zero. `stale` says whether the loaded library is the one the counters were collected on."""
    import glob
    import hashlib
    import re

    def round_key(f):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(f))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_table.json")), key=round_key, reverse=True)
    if not files:
        return None
    with open(files[0]) as fh:
        tj = json.load(fh)
    from mla_amd import hip
    with open(hip._LIB_PATH, "rb") as fh:
        lib_id = hashlib.sha256(fh.read()).hexdigest()[:16]
    busy_ms = sum(r["mfma_util_pct"] / 100.0 * r["us"] * r["n_per_step"] * 1e-3 for r in tj["rows"] if r.get("mfma_util_pct") == r.get("mfma_util_pct"))
    return {"value": round(busy_ms / ms_per_step, 4), "mfma_busy_ms_per_step_from_table": round(busy_ms, 1),
            "source": f"profiles/{os.path.basename(files[0])} (library {tj.get('library_id')}, gemm256 source {tj.get('gemm_source_id')})",
            "stale": tj.get("library_id") != lib_id}


def build(device, save_level, tiny=False, use_pointcloud=True, generation=False, stage=None):
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    with torch.device(device):
        if tiny:
            cfg = LlamaConfig(vocab_size=32000, hidden_size=512, intermediate_size=1024, num_hidden_layers=9, num_attention_heads=4,
                              activation_save_level=save_level)
        else:
            cfg = LlamaConfig(activation_save_level=save_level)   # Llama-2-7b
        bb = LLaMa2LLMBackbone("llama2-7b-pure", config=cfg)
        gen = dict(use_generation=generation, gen_image=generation, use_roi=False, gen_pointcloud=generation, gen_tactile=False)
        vlm = PrismaticVLM("mla-7b", bb, token_size=cfg.hidden_size, action_dim=7, use_diff=True, use_pointcloud=use_pointcloud,
                           use_contrastive=use_pointcloud, future_action_window_size=0, **gen)
        mla = MLA(vlm, None, token_size=cfg.hidden_size, action_dim=7, future_action_window_size=0, use_diff=True,
                  use_pointcloud=use_pointcloud, use_contrastive=use_pointcloud, **gen)
        # <BOD>, <EOD> added by scripts/train.py:132-155 stay inside the 32064 rows; give final_layer a non-zero read-out
        torch.nn.init.normal_(mla.vlm.final_layer.mlp.fc2.weight, std=0.02)
    mla.freeze_backbones(stage or ("post-training" if generation else "finetune"))
    return mla


def cpu_baseline(seconds_budget=28.0):
    """BASELINE.md section 3, timed on the GPU box's host cores in the same run as the GPU numbers, on a bounded sample:
      (i)   one Llama-2-7B decoder layer fwd+bwd (fp32 oracle, true dims, 2 x 548 tokens), median of the iterations after the warm-up
            one, for each thread count of the sweep {64, all cores} (oversubscribing a many-core host slows torch's CPU GEMMs down);
      (ii)  encoders / heads forward at true dims on 2 samples: vision tokenizer 672^2 -> 4096-d tokens, point tokenizer 1024 points,
            3-D projector, contrastive head on 256+256 tapped tokens, embedders + FinalLayer (timm 0.9.10 RmsNorm), lm_head + shifted CE;
      (iii) the tiny end-to-end step of configs[0] (oracle mla_forward + backward) measured directly.
    value = 8 samples / (17 536 tokens x 32 layers at (i)'s rate + 32 tiled encoder passes and the heads at (ii)'s rate): the 7B CPU
    step is EXTRAPOLATED from per-layer timing (a real fp32 7B step needs ~110 GB and tens of minutes)."""
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import mla_oracle, recipe
    from oracle import torch_oracle as O
    H, I, nh, S, Bs = 4096, 11008, 32, L_TEXT + S_FUSED + 3, 2
    g = torch.Generator().manual_seed(0)
    names = ["input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
             "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
             "mlp.down_proj.weight"]
    shapes = [(H,), (H, H), (H, H), (H, H), (H, H), (H,), (I, H), (I, H), (H, I)]
    p = {n: (torch.randn(*s, generator=g) * 0.02).requires_grad_(True) for n, s in zip(names, shapes)}
    x = torch.randn(Bs, S, H, generator=g).requires_grad_(True)
    cos, sin = O.rope_tables(S, H // nh)
    ncpu = os.cpu_count() or 1
    t_start = time.time()
    sweep = {}
    for nthreads in sorted({min(64, ncpu), ncpu}):
        torch.set_num_threads(nthreads)
        times = []
        for it in range(4):
            t0 = time.time()
            O.decoder_layer(x, p, cos, sin, nh, 1e-5).sum().backward()
            times.append(time.time() - t0)
            best_so_far = min((v[0] for v in sweep.values()), default=None)
            if best_so_far is not None and times[-1] > 3 * best_so_far:
                break                                  # hopeless thread count (oversubscribed host): one iteration is evidence enough
            if time.time() - t_start > 0.55 * seconds_budget and len(times) >= 2:
                break
        sweep[nthreads] = (statistics.median(times[1:]) if len(times) > 1 else times[0], len(times))
    nthreads = min(sweep, key=lambda k: sweep[k][0])
    t_layer, n_it = sweep[nthreads]
    torch.set_num_threads(nthreads)
    del p, x

    # (ii) encoders / heads at true dims (forward; the towers are frozen in SFT, the trainable heads are < 0.3 % of the step)
    from tests_shapes import MLA_TINY_SHAPES
    tower = {k: v for k, v in MLA_TINY_SHAPES.items() if k.startswith(("vlm.vision_tower_2d.", "vlm.vision_tower_3d."))}
    sd = recipe.make_state_dict(tower)
    rnd = lambda *sh: torch.randn(*sh, generator=g) * 0.02   # noqa: E731
    proj2d = dict(w0=rnd(H, 1024), b0=rnd(H), w2=rnd(H, H), b2=rnd(H))
    proj3d = dict(w0=rnd(H, 768), b0=rnd(H), w2=rnd(H, H), b2=rnd(H))
    img = torch.randn(Bs, 4, 672, 672, generator=g)
    img[:, 3] = 1.0
    pc = torch.rand(Bs, 1024, 3, generator=g)
    heads = {f"{a}_{n}_{w}": (rnd(H, H) if (n == 0 and w == "w") else rnd(256, H) if w == "w" else rnd(H if n == 0 else 256))
             for a in ("img", "pc") for n in (0, 2) for w in ("w", "b")}
    lm_w, hid = rnd(32064, H), torch.randn(Bs, S, H, generator=g)
    labels = torch.randint(0, 32000, (Bs, S), generator=g)
    parts = {}
    with torch.no_grad():
        def timed(name, fn):
            t0 = time.time()
            out = fn()
            parts[name] = time.time() - t0
            return out
        vt = timed("vision_tokenizer", lambda: O.vision_tokenizer(img, mla_oracle.vision_weights(sd), proj2d))
        ptok = timed("point_tokenizer", lambda: O.point_tokenizer(pc, mla_oracle.point_weights(sd), [torch.zeros(Bs, dtype=torch.long)] * 2))
        timed("projector_3d", lambda: O.mlp_projector(ptok[0], proj3d["w0"], proj3d["b0"], proj3d["w2"], proj3d["b2"]))
        timed("contrastive_head", lambda: O.coordinate_contrastive_loss(hid[:, 257:513], hid[:, 1:257], torch.zeros(Bs, 256, 2, dtype=torch.long),
                                                                        torch.ones(Bs, 256, dtype=torch.bool), heads))
        ew = [rnd(H, 7), rnd(H), rnd(H, H), rnd(H), rnd(H, 256), rnd(H), rnd(H, H), rnd(H), rnd(H, H), rnd(H), rnd(7, H), rnd(7)]
        timed("embedders_final_layer", lambda: (O.mlp_gelu_tanh(torch.randn(Bs * R_DIFF, 7), *ew[0:4]),
                                                O.timestep_embedder(torch.arange(Bs * R_DIFF), *ew[4:8]),
                                                O.final_layer(hid[:, -1], torch.ones(H), *ew[8:12])))
        timed("lm_head_ce", lambda: O.shifted_cross_entropy(torch.nn.functional.linear(hid, lm_w), labels))
    del lm_w, hid, heads, proj2d, proj3d
    # (iii) configs[0]: the tiny end-to-end step, forward + backward through the whole oracle
    tiny_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in recipe.make_state_dict(MLA_TINY_SHAPES).items()}
    batch, draws = recipe.make_batch(R=2)
    t0 = time.time()
    out = mla_oracle.mla_forward(tiny_sd, batch, draws, 9, 2, 1e-5, 2)
    out["total_loss"].backward()
    t_tiny = time.time() - t0

    tok_step = B_PER_GPU * R_DIFF * S
    per_seq = {"vision_tokenizer": parts["vision_tokenizer"] / Bs, "point_tokenizer": (parts["point_tokenizer"] + parts["projector_3d"]) / Bs,
               "heads": (parts["contrastive_head"] + parts["embedders_final_layer"] + parts["lm_head_ce"]) / Bs}
    t_enc = B_PER_GPU * R_DIFF * sum(per_seq.values())                      # the reference runs every encoder on the R-tiled batch
    t_step = 32 * t_layer * tok_step / (Bs * S) + t_enc
    return {"value": B_PER_GPU / t_step, "unit": "samples/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle (fp32 CPU restatement of the reference step) on {ncpu} host CPUs: LlamaDecoderLayer fwd+bwd at 7B dims on "
                      f"{Bs}x{S} tokens = {t_layer:.2f} s (median of {max(n_it - 1, 1)} after warm-up, {nthreads} threads) x 32 layers x "
                      f"{tok_step // (Bs * S)} token blocks + encoders/heads forward at true dims on {Bs} samples x {B_PER_GPU * R_DIFF // Bs} "
                      f"({t_enc:.1f} s per step); 7B step EXTRAPOLATED to {t_step:.0f} s",
            "threads_sweep_s_per_layer_iter": {str(k): round(v[0], 3) for k, v in sweep.items()},
            "components_s_on_2_samples": {k: round(v, 3) for k, v in parts.items()},
            "tiny_e2e_step_s": round(t_tiny, 3),
            "tiny_e2e_note": "BASELINE.json configs[0] (9-layer 256-d Llama, 4 sequences): oracle forward + backward, measured directly"}


def secondary_configs(steps=3, warmup=1, timeout_s=420):
    """configs[3] (post-training) and configs[4] (stage 'pretrain', S = 2048, --keep-layers 0 = every decoder layer checkpointed like
    the reference, training/strategies/fsdp.py:211-223) through this same script, one child process each, after the headline run has
    released the device. Returns {"config3": {...}, "config4": {...}}: ms_per_step, value, the child's own roofline object and peak memory.
    A child that fails is reported as {"error": ...}; it never takes the headline line down."""
    import subprocess
    res = {}
    for cfg, extra in ((3, []), (4, ["--keep-layers", "0"]), (4, ["--keep-layers", "0", "--share-prefix"]), (1, ["--readout-rows"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", str(cfg), "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-secondary", "--no-box"] + extra
        t0 = time.time()
        try:
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            key = f"config{cfg}" + ("_shared_prefix" if "--share-prefix" in extra else "") + ("_readout_rows" if "--readout-rows" in extra else "")
            if cp.returncode != 0 or not line:
                res[key] = {"error": f"rc {cp.returncode}", "stderr_tail": cp.stderr[-400:]}
                continue
            j = json.loads(line[-1])
            r = j.get("roofline") or {}
            res[key] = {
                **({"share_prefix": j["share_prefix"]} if "share_prefix" in j else {}),
                **({"readout_rows": j["readout_rows"]} if "readout_rows" in j else {}),
                "workload": j["config"]["workload"], "steps": j["steps"], "warmup": j["warmup"], "ms_per_step": j["ms_per_step"],
                "value": j["value"], "unit": j["unit"], "seq_len": j["config"]["seq_len"],
                **({"activation_policy": j["config"]["activation_policy"]} if "activation_policy" in j["config"] else {}),
                "model_tflop_per_sample": j["model_tflop_per_sample"], "peak_mem_gb": j["peak_mem_gb"],
                "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "all_gemm_frac", "whole_step_mfu",
                                                   "launches_per_step", "avg_launch_ms", "gemm_ms_per_step", "traffic")} if r else None,
                "loss": j["loss"], "wall_s_incl_model_build": round(time.time() - t0, 1)}
        except subprocess.TimeoutExpired:
            res[f"config{cfg}" + ("_shared_prefix" if "--share-prefix" in extra else "") + ("_readout_rows" if "--readout-rows" in extra else "")] = {"error": f"timeout after {timeout_s} s"}
        except Exception as e:   # noqa: BLE001 -- the headline line must survive anything the secondary runs do
            res[f"config{cfg}" + ("_shared_prefix" if "--share-prefix" in extra else "") + ("_readout_rows" if "--readout-rows" in extra else "")] = {"error": repr(e)}
    # latency of MLA.predict_action_diff (8-step DDIM, batch 1, 7B) -- with the prefix computed once per action chunk
    # (round 6, mla_amd/infer.py) and with the reference's control flow (a whole forward per DDIM step)
    for tag, extra in (("inference_predict_action_diff", []), ("inference_whole_forward_per_step", ["--no-reuse-prefix"])):
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_infer.py"), "--iters", "5"] + extra, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, timeout=timeout_s, text=True)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            res[tag] = json.loads(line[-1]) if (cp.returncode == 0 and line) else {"error": f"rc {cp.returncode}", "stderr_tail": cp.stderr[-400:]}
        except Exception as e:   # noqa: BLE001
            res[tag] = {"error": repr(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--save-level", type=int, default=1, help="activation policy: 2 keep all, 1 recompute cheap elementwise, 0 full recompute")
    ap.add_argument("--tiny", action="store_true", help="small model for smoke runs (NOT the benchmark config)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 3, 4],
                    help="BASELINE.json configs index: 1 = 7B SFT (the headline metric; also configs[2] when --gpus 8), "
                         "3 = post-training (image + point-cloud generation heads on top of config 1), "
                         "4 = pretrain shape, use_pointcloud=False, S=2048, activation checkpointing")
    ap.add_argument("--keep-layers", type=int, default=0,
                    help="config 4: decoder layers (the last N) that keep their activations instead of being checkpointed. 0 (default) = "
                         "the reference's policy and what BASELINE.json configs[4] names: every layer checkpointed "
                         "(training/strategies/fsdp.py:211-223). -1 = opt-in MIXED policy: as many layers as fit next to the MEASURED "
                         "peak of an all-checkpointed step in --mem-frac of this device's memory (torch.cuda.mem_get_info)")
    ap.add_argument("--keep-level", type=int, default=3, choices=[1, 2, 3], help="save level of the kept layers (3 = level 1 without act^T)")
    ap.add_argument("--mem-frac", type=float, default=0.91, help="share of the device's total memory the automatic --keep-layers -1 choice may plan for")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-profile", action="store_true")
    ap.add_argument("--share-prefix", action="store_true",
                    help="config 4 only, OPT-IN (reported next to the reference-layout run, never instead of it): the 4 diffusion copies of a "
                         "sample share one prefix -- [prefix 2045 | 4 x 3 suffix rows] = 2 057 executed rows per sample instead of 4 x 2 048 "
                         "(mla_amd/prismatic.py: forward_shared_prefix; same mathematics, tests/test_pretrain_gpu.py)")
    ap.add_argument("--eager-lm-head", action="store_true",
                    help="compute lm_head + cross entropy inside every forward like the reference (modeling_llama.py:1255-1269) instead of on "
                         "first access of output.logits / output.loss (the default since round 6: the trainer discards `output`, SURVEY App. A #7)")
    ap.add_argument("--readout-rows", action="store_true",
                    help="opt-in (round 6, configs 1 / 4): MLA.readout_rows_only -- the last decoder layer runs its row-wise half (o_proj, MLP) on the "
                         "action read-out rows only, the one thing the diffusion objective reads of the final hidden state")
    ap.add_argument("--no-box", action="store_true", help="skip the `box` block (sclk / power sampler + the two 300 ms MFMA calibrations)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (configs[3] and configs[4] at 3 timed steps each) the default 1-GPU configs[1] run appends")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if "RANK" not in os.environ:
        # the parent (plain `python bench.py --gpus N`): never a silent smaller job, N ranks need N devices (one RCCL rank per GPU)
        if ndev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible")
    elif os.environ.get("MLA_BENCH_REHEARSAL") == "1":
        # rehearsal of the N-rank control flow on ONE GPU (tools/rehearse_bench_ranks.sh): every rank uses device 0 and the collectives
        # go through gloo. Only with --tiny (N replicas of the 7B model do not fit one GPU); the numbers mean nothing, the JSON line says so
        if not args.tiny:
            raise SystemExit("MLA_BENCH_REHEARSAL=1 needs --tiny")
    elif local_rank >= max(ndev, 1) and ndev != 1:
        # a launcher's worker: launchers that bind ONE visible device per rank (per-rank HIP_VISIBLE_DEVICES, SLURM --gpus-per-task=1)
        # show ndev == 1 with WORLD_SIZE == --gpus, which is fine; what must not happen is a rank without a device of its own
        raise SystemExit(f"bench.py rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    if args.gpus != world:
        if "RANK" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: re-launch ourselves as N ranks (one process per GPU) under torch.distributed.run --
            # exactly the command the driver uses; rank 0 of the children prints the JSON line
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                      "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    # stdout carries exactly ONE line, the JSON: native libraries that print to fd 1 (RCCL's version banner at communicator
    # creation) are pointed at stderr for the whole run, the line itself goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rehearsal = os.environ.get("MLA_BENCH_REHEARSAL") == "1" and "RANK" in os.environ
    dev_index = local_rank if (local_rank < ndev and not rehearsal) else 0      # one-visible-device-per-rank launchers: every rank's device is index 0
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_knobs = {}
    if world > 1 or "RANK" in os.environ:       # launched through torch.distributed.run (also with --gpus 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from mla_amd.fsdp import apply_rccl_env
        coll_knobs = apply_rccl_env()           # MLA_RCCL_MAX_CHANNELS / MLA_GEMM_CUS / MLA_FSDP_INPLACE_RS
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from mla_amd import hip
    from mla_amd.strategy import FSDPStrategy
    from mla_amd.synthetic import make_batch

    l_text = L_TEXT if args.config in (1, 3) else 2048 - S_FUSED - 3
    pc_on, gen_on = args.config in (1, 3), args.config == 3
    if args.config == 4:
        args.save_level = 0                    # the config names activation checkpointing (4x the tokens of config 1)
    torch.manual_seed(42)                      # identical initial weights on every rank (scripts/train.py:76 seed)
    stage = "post-training" if gen_on else ("pretrain" if args.config == 4 else "finetune")   # config 4: the vision tokenizer trains too
    mla = build(device, args.save_level, args.tiny, use_pointcloud=pc_on, generation=gen_on, stage=stage)
    mla.vlm.llm_backbone.llm.config.lazy_lm_head = not args.eager_lm_head
    if args.readout_rows:
        if args.config == 3:
            raise SystemExit("--readout-rows: the generation heads of config 3 read the whole final hidden state")
        mla.readout_rows_only = True
    if args.share_prefix:
        if args.config != 4:
            raise SystemExit("--share-prefix applies to config 4 (use_pointcloud=False): with a point cloud the FPS start indices differ per copy")
        mla.share_prefix = True
    torch.manual_seed(42 + rank)               # rank-local noise / timesteps / FPS starts, like the reference's per-rank RNG
    strat = FSDPStrategy(mla, dev_index, stage=stage, global_batch_size=B_PER_GPU * world, per_device_batch_size=B_PER_GPU,
                         learning_rate=2e-5, weight_decay=0.0, max_grad_norm=1.0, lr_scheduler_type="constant",
                         enable_gradient_checkpointing=False, repeated_diffusion_steps=R_DIFF)
    strat.run_setup(n_train_examples=10_000)
    keep_layers = 0
    if args.config == 4 and not args.tiny:
        # Mixed activation policy: the config names activation checkpointing because the reference targets 80 GB parts; with 288 GB the
        # LAST k decoder layers keep their activations (level 3: h, qkv, o, lse, h_mid, gate|up = 93 KB per token and layer) and only the
        # others recompute their forward. Bit-identical results for every k (tests/test_model_gpu.py::test_decoder_stack_mixed_...).
        keep_layers = max(0, args.keep_layers)
        mla.vlm.llm_backbone.set_activation_policy(keep_layers, keep_level=args.keep_level, rest_level=0)
    batch = make_batch(B=B_PER_GPU, L_text=l_text, seed=42 + rank, device=device, use_pointcloud=pc_on, with_next=gen_on)
    S = l_text + S_FUSED + 3

    def sync():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    if args.config == 4 and not args.tiny and args.keep_layers < 0:
        # opt-in mixed policy, sized from THIS rank's measurements (advisor, round 3: no constants): one all-checkpointed step gives the
        # base peak (weights, optimizer shards, gradients, one layer's recompute, logits); every kept layer adds its saved activations
        strat.train_step(batch)
        strat.synchronize()
        torch.cuda.synchronize()
        base = torch.cuda.max_memory_allocated()                 # bytes
        total = torch.cuda.mem_get_info()[1]                     # bytes, this device
        tok = B_PER_GPU * ((2048 - 3) + (R_DIFF + 1) * 3 if args.share_prefix else R_DIFF * 2048)   # executed rows per rank
        per_layer = ({1: (6 * 4096 + 3 * 11008) * 2, 3: (6 * 4096 + 2 * 11008) * 2, 2: (8 * 4096 + 4 * 11008) * 2}[args.keep_level]
                     - 4096 * 2) * tok                           # bytes; minus the layer input a checkpointed layer keeps anyway
        keep_layers = max(0, min(32, int((args.mem_frac * total - base) // per_layer)))
        if world > 1:                                            # every rank must run the same policy: take the smallest choice
            kt = torch.tensor([keep_layers], device=device)
            dist.all_reduce(kt, op=dist.ReduceOp.MIN)
            keep_layers = int(kt.item())
        print(f"# mixed activation policy: base peak {base / 2**30:.1f} GiB, device {total / 2**30:.1f} GiB, {per_layer / 2**30:.2f} GiB per kept "
              f"layer -> keep {keep_layers}", file=sys.stderr)
        mla.vlm.llm_backbone.set_activation_policy(keep_layers, keep_level=args.keep_level, rest_level=0)
        losses = None
        torch.cuda.empty_cache()                 # the probe step's cached blocks have the all-checkpointed step's shapes
        torch.cuda.reset_peak_memory_stats()
    box = None
    if rank == 0 and not args.no_box:
        # (cold) calibration before the warm-up, 300 ms: the random-operand MFMA stream from a chip that has been idle
        strat.synchronize()
        torch.cuda.synchronize()
        box = {"mfma_random_pflops_cold": round(hip.calib_mfma(device)["pflops"], 4)}
    for _ in range(args.warmup):
        losses = strat.train_step(batch)
    prof = None if args.no_gemm_profile else []
    sampler = BoxSampler(dev_index) if box is not None else None
    strat.synchronize()                        # flush the warm-up's deferred optimizer updates: the timed region owns exactly K of them
    sync()
    if strat.sharded.coll:
        # time every stall of the compute stream behind a reduce-scatter / all-gather event -- switched on AFTER the flush above, so the
        # warm-up's exposed all-gather waits are not divided into the K timed steps (advisor, round 4)
        strat.sharded.wait_profile = []
    hip.GEMM_PROFILE = prof
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = strat.train_step(batch)
    strat.synchronize()                        # the last step's AdamW (normally overlapped with the next forward) completes inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    hip.GEMM_PROFILE = None
    if sampler is not None:
        box.update(sampler.stop())
        # (hot) calibration right behind the timed steps, outside the timed region: the same stream on the chip in the thermal / power
        # state the steps left it in -- the denominator of roofline.frac_of_box_ceiling
        cal = hip.calib_mfma(device)
        # second yardstick: the step's own GEMM kernel on random operands, alone on the chip, at the step's dominant shape
        # (tokens x 4096 x 4096): what the in-step launches lose against it is the step's doing (cache state, neighbours), not the box's
        ga = torch.randn(B_PER_GPU * R_DIFF * S if not args.tiny else 1024, 4096, device=device).to(torch.bfloat16)
        gb = torch.randn(4096, 4096, device=device).to(torch.bfloat16)
        gc = torch.empty(ga.shape[0], 4096, dtype=torch.bfloat16, device=device)
        for _ in range(20):
            hip.gemm(ga, gb, out=gc)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_g = 400
        g0.record()
        for _ in range(n_g):
            hip.gemm(ga, gb, out=gc)
        g1.record()
        torch.cuda.synchronize()
        box["gemm256_standalone_pflops"] = round(2.0 * ga.shape[0] * 4096 * 4096 * n_g / (g0.elapsed_time(g1) * 1e-3) / 1e15, 4)
        del ga, gb, gc
        box.update({"mfma_random_pflops": round(cal["pflops"], 4), "mfma_calibration": f"{cal['launches']} launches x {cal['ms_per_launch']:.1f} ms on "
                    f"{cal['blocks']} workgroups x 8 waves, v_mfma_f32_16x16x32_bf16 on N(0,1) operands from registers (mla_calib_mfma), last 200 ms of 300",
                    "device": torch.cuda.get_device_name(dev_index)})
    per_rank = None
    if strat.sharded.coll:
        # first-run diagnosis for N > 1 (VERDICT r3 #3d): every rank's own step time and how long ITS compute stream sat behind
        # collectives -- rs_event = the backward's reduce-scatters not finished when clipping starts (exposed communication),
        # gather_event = a layer's bf16 all-gather (issued behind AdamW) not finished when the next forward reaches the layer
        waits = strat.sharded.wait_profile_ms()
        mine = torch.tensor([elapsed / args.steps * 1e3, waits.get("rs_event", 0.0) / args.steps, waits.get("gather_event", 0.0) / args.steps],
                            dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
        if world > 1:
            dist.all_gather(allr, mine)
        rows = [[round(float(x), 3) for x in r.tolist()] for r in allr]
        # self-check of the collectives (round 6: RCCL's out-of-place ncclAvg dropped a tail at world 1; nothing can be tested at N > 1
        # without the node): after K optimizer steps every rank's bf16 replica must be the SAME bits -- one float64 checksum per sharding
        # unit, MIN- and MAX-reduced over the ranks; a difference means an all-gather (or the sharded update in front of it) went wrong
        strat.synchronize()
        torch.cuda.synchronize()
        sums = torch.stack([u.flat16.double().sum() + u.flat16[::97].double().abs().sum() for u in strat.sharded.units])
        lo, hi = sums.clone(), sums.clone()
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_equal = bool(torch.equal(lo, hi)) and bool(torch.isfinite(sums).all())
        per_rank = {"replicas_bit_identical_after_the_run": replicas_equal, "step_ms": [r[0] for r in rows], "step_ms_min": min(r[0] for r in rows), "step_ms_max": max(r[0] for r in rows),
                    "rs_event_wait_ms_per_step": [r[1] for r in rows], "gather_event_wait_ms_per_step": [r[2] for r in rows],
                    "note": "main-stream stall behind the side-stream collectives, bracketed by HIP events around each wait_event"}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B_PER_GPU * args.steps / elapsed
        dec_fl, tot_fl = model_flops_per_sample(S)
        if not args.eager_lm_head:
            tot_fl = dec_fl                    # lazy lm_head: the 2 H V flops per token are not executed in a training step, so not counted
        ref_layout_fl = tot_fl
        readout_fl = None
        if args.readout_rows and not args.share_prefix:
            # executed work: the last layer's o_proj + MLP (2 H^2 + 6 H I of its 8 H^2 + 6 H I per token, x 3 for forward + backward)
            # run on the B' x T read-out rows instead of B' x S
            H_, I_ = 4096, 11008
            readout_fl = 3 * (2 * H_ * H_ + 6 * H_ * I_) * (S - 1) * R_DIFF
            tot_fl -= readout_fl
        if args.share_prefix:
            # MFU figures are about EXECUTED work: one (S - 3) + 4 x 3 row sequence per sample (its causal attention priced as fully causal)
            tot_fl = model_flops_per_sample((S - 3) + (R_DIFF + 1) * 3, R=1)[0]      # (+ one dummy group: 2 060 rows, a multiple of 4)
        heads_fl = 0.0
        if prof and args.config == 3 and not args.tiny:
            # configs[3]: the generation heads' GEMM work is not in the decoder formula -- take it from the launches themselves:
            # (sum of 2MNK over every GEMM launch of a step) - (the decoder's + lm_head's linear layers, forward + backward)
            H_, I_, L_, V_ = 4096, 11008, 32, 32064
            tok = B_PER_GPU * R_DIFF * S
            dec_gemm = 3.0 * (8 * H_ * H_ + 6 * H_ * I_) * L_ * tok + (2.0 * H_ * V_ * tok if args.eager_lm_head else 0.0)
            heads_fl = max(0.0, sum(fl for _, _, fl, _ in prof) / args.steps - dec_gemm) / B_PER_GPU
            tot_fl += heads_fl
        roof = None
        if prof and any(fl > 1e11 for _, _, fl, _ in prof):       # (the tiny rehearsal model has no launch of that size since lm_head went lazy)
            big = [(e0.elapsed_time(e1), fl, key) for e0, e1, fl, key in prof if fl > 1e11]
            tsum = sum(t for t, _, _ in big) * 1e-3
            fsum = sum(fl for _, fl, _ in big)
            ach = fsum / tsum / 1e12
            # HBM bytes per launch come from separate rocprofv3 --pmc passes over this same command (PMC collection perturbs timing,
            # so it is not done inline): profiles/r1_gemm256_hbm_traffic.json, FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE
            traffic = None
            tprov = None
            hit = None
            stale = None
            # newest record first (parsed round tag: r10 > r4b > r4 > r3d ...)
            import glob
            import re

            def round_key(f):                      # "r10b_..." -> (10, "b"): numeric round first, so r10 sorts above r4b
                m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(f))
                return (int(m.group(1)), m.group(2)) if m else (-1, "")
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm256_hbm_traffic.json")), key=round_key, reverse=True)
            for tpath in tfiles:
                if args.config == 1 and not args.tiny:
                    with open(tpath) as fh:
                        tj = json.load(fh)
                    traffic = round(tj["hbm_bytes_per_launch"])
                    hit = tj.get("tcc_hit_rate")
                    # the counters describe the kernel they were collected on: a record without the library's gemm256 source id, or with
                    # another one, is reported as stale instead of silently standing for the current kernel
                    stale = tj.get("gemm_source_id") != hip.gemm_source_id()
                    tprov = (f"profiles/{os.path.basename(tpath)} (collected {tj.get('collected', 'round 1')} on gemm256 source id "
                             f"{tj.get('gemm_source_id', 'unrecorded')}, loaded library {hip.gemm_source_id()}; separate --pmc passes over this command, not this run)")
                    break
            if os.environ.get("MLA_BENCH_GEMM_SHAPES"):
                by = {}
                for t, fl, key in big:
                    e = by.setdefault(key, [0, 0.0, fl])
                    e[0] += 1
                    e[1] += t
                for key, (n, t, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                    print(f"# gemm (a_mode, b_mode, M, N, K)={key}: {n // args.steps}/step, {t / n:.4f} ms avg, {fl * n / t / 1e9:.0f} TFLOP/s, "
                          f"{t / args.steps:.1f} ms/step", file=sys.stderr)
            # The fused gate|up + SwiGLU and d(act) + SwiGLU-backward launches are separate kernels (gemm256_kernel<0,0,1> / <0,0,2>:
            # a GEMM plus an HBM-bound elementwise pass in the epilogue); the roofline object is about the dominant kernel, the
            # plain gemm256_kernel<0,0,0> (incl. the RoPE epilogue of the QKV projection), and reports the fused ones next to it
            fused = [(t, fl) for t, fl, k in big if len(k) >= 6 and k[5] != "rope_epilogue"]
            big = [(t, fl, k) for t, fl, k in big if not (len(k) >= 6 and k[5] != "rope_epilogue")]
            ach_all = (fsum / tsum) / 1e12
            tsum = sum(t for t, _, _ in big) * 1e-3
            fsum = sum(fl for _, fl, _ in big)
            ach = fsum / tsum / 1e12
            ach_fused = (sum(fl for _, fl in fused) / (sum(t for t, _ in fused) * 1e-3) / 1e12) if fused else None
            abytes = sum(2.0 * (k[2] * k[4] + k[3] * k[4]) + 2.0 * k[2] * k[3] for _, _, k in big) / len(big)
            roof = {"bound": "mfma", "kernel": "gemm256_kernel<0,0,0> (+ gemm128_kernel for small shapes): bf16 MFMA GEMM launches >= 0.1 TFLOP",
                    "main_loop": ("hand-scheduled assembly (gemm256_kloop.inc)" if hip.gemm_kloop(-1) == 1 else "compiler-scheduled (MLA_GEMM_KLOOP=0)"),
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "fused_swiglu_epilogue_kernels": ({"kernel": "gemm256_kernel<0,0,1> (gate|up + SwiGLU) and <0,0,2> (d(act) + SwiGLU backward): GEMM flops "
                                                                 "over a duration that also contains the HBM-bound SwiGLU pass",
                                                       "achieved_gemm_flops_only": round(ach_fused, 1), "launches_per_step": len(fused) // args.steps}
                                                      if fused else None),
                    "all_gemm_launches_achieved": round(ach_all, 1),
                    # the two numbers that bracket `frac` live INSIDE the roofline object -- every GEMM launch
                    # >= 0.1 TFLOP incl. the fused-epilogue kernels, and the whole step as model FLOPs (no recompute credit)
                    "all_gemm_frac": round(ach_all / PEAK_BF16_TFLOPS, 4),
                    "whole_step_mfu": round(tot_fl * B_PER_GPU / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "traffic_stale": stale, "traffic_source": tprov,
                    "traffic_unit": "HBM+MALL bytes per launch (fabric-side counters), avg over the plain gemm256_kernel<0,0,0> launches >= 0.1 TFLOP "
                                    "(the population of algorithmic_bytes_per_launch)",
                    "l2_hit_rate": (round(hit, 4) if hit is not None else None),
                    "algorithmic_bytes_per_launch_2B_outputs": round(abytes), "launches_per_step": len(big) // args.steps, "avg_launch_ms": round(tsum / len(big) * 1e3, 4),
                    "gemm_ms_per_step": round(tsum / args.steps * 1e3, 1)}
            if box and box.get("mfma_random_pflops"):
                # the fraction a slow box and a fast box agree on -- the plain GEMM launches against what THIS box's
                # matrix cores sustain on random operands under its power cap (measured right behind the timed steps)
                roof["frac_of_box_ceiling"] = round(ach / 1e3 / box["mfma_random_pflops"], 4)
                roof["all_gemm_frac_of_box_ceiling"] = round(ach_all / 1e3 / box["mfma_random_pflops"], 4)
                roof["frac_of_standalone_gemm"] = round(ach / 1e3 / box["gemm256_standalone_pflops"], 4)
            if args.config == 1 and not args.tiny:
                roof["step_weighted_mfma_util"] = step_weighted_mfma_util(ms)
        out = {"metric": "training samples/sec + step-time, MLA-Llama2-7B bf16", "value": round(value, 3), "unit": "samples/s",
               "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 0, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": ("BASELINE.json configs[1]: MLA-Llama2-7B SFT, use_pointcloud+use_contrastive, 672x672(+mask) image + "
                                       "1024 points + 32 text tokens, per-GPU batch 8 x 4 diffusion repeats = 32 x 548 tokens" if args.config == 1 else
                                       "BASELINE.json configs[3]: MLA post-training = configs[1] + image (128 queries, 2+3 decoder layers, d=4096) and "
                                       "point-cloud (4 blocks, d=1024) generation heads, use_roi=False, dropout 0.1 active; 32 x 548 tokens" if args.config == 3 else
                                       "BASELINE.json configs[4]: MLA stage 'pretrain' (vision tokenizer + projector + LLM trainable), use_pointcloud=False, S=2048, activation checkpointing, "
                                       "per-GPU batch 8 x 4 diffusion repeats = 32 x 2048 tokens")
                                      + (" [TINY SMOKE MODEL - not the benchmark]" if args.tiny else ""),
                          "model": "mla-llama2-7b" if not args.tiny else "tiny", "global_batch": B_PER_GPU * world, "seq_len": S,
                          "parallelism": f"fsdp-rccl x{world}" if world > 1 else "single-gpu", "activation_save_level": args.save_level,
                          **({"activation_policy": ("every decoder layer checkpointed (the reference's policy, fsdp.py:211-223); a layer's recomputation stops in front of its down projection, whose output the backward never reads" if keep_layers == 0 else
                                                    f"MIXED (opt-in, not the reference's policy): last {keep_layers} of 32 decoder layers keep "
                                                    f"activations (level {args.keep_level}), {32 - keep_layers} checkpointed (level 0)")}
                             if args.config == 4 and not args.tiny else {}),
                          "optimizer": "fused AdamW + grad clip inside the timed region",
                          "lm_head": ("eager: lm_head + shifted CE inside every forward, like the reference" if args.eager_lm_head else
                                      "lazy: lm_head(h).float() + shifted CE (modeling_llama.py:1255-1269) run on first access of output.logits / "
                                      "output.loss; the training loop never reads them (base_strategy_mla.py:307,334), so they are neither executed "
                                      "in the timed steps nor counted in model_tflop_per_sample")},
               "model_tflop_per_sample": round(tot_fl / 1e12, 2),
               **({"share_prefix": {"executed_rows_per_sample": (S - 3) + (R_DIFF + 1) * 3, "reference_layout_rows_per_sample": R_DIFF * S,
                                    "executed_tflop_per_sample": round(tot_fl / 1e12, 2),
                                    "reference_layout_tflop_per_sample": round(ref_layout_fl / 1e12, 2),
                                    "note": "opt-in: [prefix | 4 suffix groups] per sample (suffix rows attend to the prefix and their own copy, at the "
                                            "reference's positions). `value` counts the same dataset samples per second as the reference-layout run; "
                                            "model_tflop_per_sample / mfu / whole_step_mfu in THIS line are the EXECUTED work (the causal attention of "
                                            "the 2 060-row sequence (2 045 prefix + 4 suffix groups + 1 dummy group of 3 rows) priced as fully causal), so the speed-up over config4 is an algorithmic saving, "
                                            "not a kernel rate"}}
                  if args.share_prefix else {}),
               **({"readout_rows": {"executed_tflop_per_sample": round(tot_fl / 1e12, 2), "dense_layout_tflop_per_sample": round(ref_layout_fl / 1e12, 2),
                                    "note": "opt-in (MLA.readout_rows_only): the diffusion objective reads only the action read-out rows of the final hidden state "
                                            "(models/vlm/prismatic.py:1115-1126; lm_head + CE are unused and lazy, `output` is discarded by the trainer), so the "
                                            "last decoder layer runs o_proj + MLP, forward and backward, on those B' x T rows instead of B' x S. Same losses and "
                                            "gradients to rounding; the dense final hidden state / logits are produced on first access. model_tflop_per_sample / "
                                            "mfu in THIS line are the EXECUTED work"}} if readout_fl is not None else {}),
               **({"heads_encoders_tflop_per_sample_from_gemm_launches": round(heads_fl / 1e12, 2)} if heads_fl else {}),
               "model_tflops_per_gpu": round(tot_fl * B_PER_GPU / (ms * 1e-3) / 1e12, 1),
               "mfu_vs_2.5PF": round(tot_fl * B_PER_GPU / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
               "loss": {k: float(v) for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1},
               "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        if rehearsal:
            out["rehearsal"] = "MLA_BENCH_REHEARSAL=1: all ranks on ONE GPU over gloo, tiny model -- control flow only, not a measurement"
            out["config"]["parallelism"] = f"fsdp-gloo x{world} on one device (rehearsal)"
        if per_rank:
            out["per_rank"] = per_rank
            out["collective_knobs"] = {**coll_knobs, "inplace_reduce_scatter": bool(strat.sharded.inplace_reduce), "gemm_planned_cus": hip.gemm_cus() or "device"}
        if roof:
            out["roofline"] = roof
        if box:
            out["box"] = box
        if world == 1 and args.config == 1 and not args.tiny and not args.no_secondary and "RANK" not in os.environ:
            # configs[3] and configs[4] (the reference's policy: every layer checkpointed) become driver-observed --
            # each runs as its own process on the now empty GPU (1 warm-up + 3 timed steps), `value` / `config` above stay configs[1]
            del strat, mla, batch, losses
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["secondary"] = secondary_configs()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world > 1:
            assert out["rccl_ranks"] == world == args.gpus, (out["rccl_ranks"], world, args.gpus)   # N GPUs means N RCCL ranks, never fewer
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()
    # fd 1 is stdout again for whatever runs after main() in this process -- but only after the C library's buffered stdout (RCCL's
    # version banner sits there until exit when fd 1 is a pipe or a file) has been flushed to where fd 1 points NOW (stderr):
    # restored without the flush, the banner landed behind the JSON line on the real stdout
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    os.close(real_stdout)


if __name__ == "__main__":
    main()
