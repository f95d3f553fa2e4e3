"""Roofline table of BASELINE configs[1] (7B SFT, 32 x 548 tokens): every kernel >= 0.1 % of the step from a tools/step_breakdown.py file,
joined with its ALGORITHMIC work per step -- flops for the MFMA-bound kernels (priced against 2.5 PFLOP/s dense bf16), bytes = inputs read
+ outputs written once for the HBM-bound ones (priced against 8 TB/s spec; ~6.3 TB/s is what a float4 copy reaches, MI355X_MICROARCH.md).
Usage: python tools/roofline_table.py <step_breakdown_config1.txt>"""
import re, sys

T, H, I, L, V, S, BH, D = 17536, 4096, 11008, 32, 32064, 548, 32 * 32, 128
PARAMS = 6.76e9                                   # trainable in stage "finetune": the 7B decoder + lm_head/embeddings + projectors, heads
PF, TB = 2.5e15, 8.0e12
ATT = 4.0 * D * S * S / 2 * BH                    # causal forward flops of one layer; backward = 2.5 x
ROWS = [   # (substring of the kernel symbol, label, bound, work per STEP, note)
    # round 6: lm_head + cross entropy are lazy (not executed in a training step): their 2 T H V = 4.6 TF are no longer in this row
    # (348 launches per step instead of 349); pass --eager-lm-head as the second argument to price a step that runs them
    ("gemm256_kernelILi0ELi0ELi0", "gemm256<0,0,0> plain GEMM", "mfma", 2.0 * T * (12 * H * H + 6 * H * I) * L + (2.0 * T * H * V if "--eager-lm-head" in sys.argv else 0.0) + 3e12,
     "decoder fwd/dgrad/wgrad GEMMs that are not fused + heads/encoders (~3 T)" + (" + lm_head" if "--eager-lm-head" in sys.argv else "; lm_head is lazy since round 6")),
    ("gemm256_kernelILi0ELi0ELi1", "gemm256<0,0,1> gate|up + SwiGLU", "mfma", 2.0 * T * 2 * I * H * L, "GEMM flops only; the epilogue also moves 1.5 GB / launch"),
    ("gemm256_kernelILi0ELi0ELi2", "gemm256<0,0,2> d(act) + SwiGLU bwd", "mfma", 2.0 * T * I * H * L, "GEMM flops only; the epilogue also moves 2.3 GB / launch"),
    ("attn_fwd_kernel", "attention forward", "mfma", ATT * L, "causal flops 4 D S^2 / 2 per (b, h)"),
    ("attn_bwd_dq_kernel", "attention backward: dQ kernel", "mfma", 2.5 * ATT * L * 3 / 7, "3 of the 7 executed products (algorithmic bwd = 2.5 x fwd over both kernels)"),
    ("attn_bwd_dkv_kernel", "attention backward: dK/dV kernel", "mfma", 2.5 * ATT * L * 4 / 7, "4 of the 7 executed products"),
    ("attn_bwd_kernel", "attention backward (one pass)", "mfma", 2.5 * ATT * L, "5 products"),
    ("attn_bwd_merged_kernel", "attention backward (dQ + dK.dV, 1 launch)", "mfma", 2.5 * ATT * L, "algorithmic 5 products (7 executed), both block types in one launch (round 5)"),
    # round 5: at S = 548 the backward pair is nearer its HBM roof than its MFMA one -- algorithmic bytes of the TWO-kernel form, per layer:
    # dQ reads q, k, v, dO, o and writes dq, dq^T, o^T (8 x T x H x 2 B); dK dV reads q, k, v, dO and writes dk, dv, dk^T, dv^T (8 x ...)
    ("attn_bwd_dq_kernel#hbm", "  same kernel against the HBM roof", "hbm", 8.0 * T * H * 2 * L, "q k v dO o read once, dq dq^T o^T written (two-kernel form)"),
    ("attn_bwd_dkv_kernel#hbm", "  same kernel against the HBM roof", "hbm", 8.0 * T * H * 2 * L, "q k v dO read once, dk dv dk^T dv^T written"),
    ("attn_bwd_merged_kernel#hbm", "  same kernel against the HBM roof", "hbm", 12.0 * T * H * 2 * L, "q k v dO o read ONCE, dq dk dv + 4 transposed operands written (the two-launch form's own minimum is 16 x)"),
    ("adamw_vec4_kernel", "fused AdamW (+ bf16 copy)", "hbm", 30.0 * PARAMS, "30 B / parameter: p, m, v read+write, g read, bf16 write"),
    ("tile_transpose_kernelINS_6CopyOp", "tile transposes (W^T, dy^T)", "hbm", (4.0 * (3 * I * H + 4 * H * H) + 8.0 * T * H) * L, "read + write, 2 B each"),
    ("tile_transpose_kernelINS_10RmsApplyOp", "RMSNorm re-apply, transposed out", "hbm", 4.0 * T * H * 2 * L, "read h, write xn^T"),
    ("rmsnorm_bwd_kernel", "RMSNorm backward (+ residual add)", "hbm", 8.0 * T * H * (2 * L + 1), "dy, x, d_res read; dx written"),
    ("rmsnorm_fwd_kernel", "RMSNorm forward", "hbm", 4.0 * T * H * (2 * L + 1), "x read, xn written"),
    ("ce_fwd_kernel", "cross entropy forward (fp32 logits)", "hbm", 4.0 * T * V, "one pass over the fp32 logits"),
    ("sumsq_partial_kernel", "gradient sum of squares", "hbm", 4.0 * PARAMS, "only under FSDP / forced collectives: 4 B / parameter"),
    ("gemm256_fixup_kernel", "split-K fix-up", None, None, "sums <= 8 fp32 partial tiles of the last round, applies the epilogue"),
    ("gemm128_kernel", "gemm128 (small / odd shapes)", None, None, "encoder, projector and head GEMMs below the 256-tile contract"),
]
lines = open(sys.argv[1]).read().splitlines()
m = re.match(r"step wall ([\d.]+) ms, (\d+) launches, kernel time ([\d.]+) ms, idle gaps ([\d.]+) ms", lines[0])
wall = float(m.group(1))
print(f"# {lines[0]}")
print("# algorithmic work per step / measured kernel time per step (rocprofv3 kernel trace of one training step, tools/step_breakdown.py)")
print(f"{'kernel':38s} {'calls':>5s} {'ms/step':>8s} {'% step':>6s} {'bound':>5s} {'work/step':>12s} {'achieved':>13s} {'peak':>11s} {'frac':>6s}  note")
seen = 0.0
for ln in lines[1:]:
    mm = re.match(r"\s+([\d.]+) ms\s+(\d+) x\s+([\d.]+) us\s+(\S+)", ln)
    if not mm:
        continue
    ms, calls, sym = float(mm.group(1)), int(mm.group(2)), mm.group(4)
    if ms < 0.001 * wall:
        continue
    row = next((r for r in ROWS if r[0] in sym), None)
    extra = next((r for r in ROWS if r[0].endswith("#hbm") and r[0][:-4] in sym), None)
    seen += ms
    if row is None:
        print(f"{sym[-38:]:38s} {calls:5d} {ms:8.2f} {100 * ms / wall:6.2f} {'-':>5s} {'-':>12s} {'-':>13s} {'-':>11s} {'-':>6s}  (point / vision tower or framework kernel)")
        continue
    _, label, bound, work, note = row
    if bound == "mfma":
        ach = work / (ms * 1e-3)
        print(f"{label:38s} {calls:5d} {ms:8.2f} {100 * ms / wall:6.2f} {'mfma':>5s} {work / 1e12:9.2f} TF {ach / 1e15:8.3f} PF/s {PF / 1e15:6.1f} PF/s {ach / PF:6.3f}  {note}")
    elif bound == "hbm":
        ach = work / (ms * 1e-3)
        print(f"{label:38s} {calls:5d} {ms:8.2f} {100 * ms / wall:6.2f} {'hbm':>5s} {work / 1e9:9.1f} GB {ach / 1e12:8.2f} TB/s {TB / 1e12:6.1f} TB/s {ach / TB:6.3f}  {note}")
    else:
        print(f"{label:38s} {calls:5d} {ms:8.2f} {100 * ms / wall:6.2f} {'-':>5s} {'-':>12s} {'-':>13s} {'-':>11s} {'-':>6s}  {note}")
    if extra is not None:
        _, label, _, work, note = extra
        ach = work / (ms * 1e-3)
        print(f"{label:38s} {'':5s} {'':8s} {'':6s} {'hbm':>5s} {work / 1e9:9.1f} GB {ach / 1e12:8.2f} TB/s {TB / 1e12:6.1f} TB/s {ach / TB:6.3f}  {note}")
print(f"# kernels listed: {seen:.1f} ms of the {wall:.1f} ms step")
