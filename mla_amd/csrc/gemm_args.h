// Argument block shared by the GEMM kernels (gemm.hip, gemm256.hip, gemm_asm.hip).
#pragma once
#include "common.h"

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* R;     // optional residual, bf16 [M, ldr]
  const bf16_t* bias;  // optional bias, bf16 [N]
  int M, N, K;
  int lda, ldb, ldc, ldr;
  int out_fp32;    // 0: C is bf16, 1: C is fp32
  int accumulate;  // fp32 output only: C += result
  float alpha;
  int debug;  // experiments only (tools/)
  // batched launches (gemm128 / generic only): z = blockIdx.y = outer * n_inner + inner, element strides per operand
  int n_inner;
  long long sAo, sAi, sBo, sBi, sCo, sCi;
  // split-K tail of gemm256 (see gemm256.hip): tiles [0, sk_full) are computed whole; each of the remaining tiles is cut into
  // sk_split K-slices whose fp32 partial tiles go to sk_ws [unit][256][256]; a fix-up kernel sums them and applies the epilogue
  int sk_full, sk_split;
  float* sk_ws;
  // fp32-output launches (wgrad into the fp32 gradient buffer) can also leave sum(C^2) of the FINAL values (after accumulate) as one
  // partial per workgroup: slot = tile id for whole tiles, sk_full + 64 * tail_tile + block for the fix-up pass. The gradient norm of
  // the step is then a sum over these partials instead of a second pass over the 4-B/parameter gradient buffer.
  float* sq_out;
  // gemm256 bf16 fast epilogue only: rotary embedding applied to output columns [0, rope_cols) on the way out (heads of 128 columns,
  // pairs (d, d + 64), position = output row % rope_S; tables [rope_S, 64] fp32) -- the fused QKV projection (mla_gemm_qkv_rope)
  const float* rope_cos;
  const float* rope_sin;
  int rope_S, rope_cols;
  // gemm256 bf16 fast epilogue only: the product C is d(act) of a SwiGLU MLP and never leaves the chip -- the epilogue reads gate|up
  // (sw_gu [M, 2 sw_I]), applies the SwiGLU backward and writes d(gate|up) in both layouts: sw_dgu [M, 2 sw_I] and sw_dguT
  // [2 sw_I, sw_ldt] (mla_gemm_dact_swiglu_bwd). N == sw_I.
  // gemm256 bf16 fast epilogue only: fused gate|up projection + SwiGLU (mla_gemm_gateup_swiglu). N = 2 sf_I; workgroup column j
  // computes gate channels [128 j, 128 j + 128) AND up channels [sf_I + 128 j, ...) (its B half-tiles come from the two halves of
  // the packed weight), so every lane holds g and u of the same channel: C = gate|up [M, 2 sf_I] is stored as usual, and
  // act = silu(g) u goes to sf_act [M, sf_I] and (optional) sf_actT [sf_I, sf_ldt].
  int sf_I;
  bf16_t* sf_act;
  bf16_t* sf_actT;
  long long sf_ldt;
  const bf16_t* sw_gu;
  bf16_t* sw_dgu;
  bf16_t* sw_dguT;
  int sw_I;
  long long sw_ldt;
  // fused-epilogue launches only (EPI != 0), experiment switch MLA_GEMM_STAGGER=<mode>:<ticks>: part of the FIRST round of workgroups
  // starts `stagger_ticks` (10 ns units) late, so that the HBM-bound epilogue bursts of one half of the chip fall into the other
  // half's main loops instead of all 256 CUs storing at once. mode 1: odd XCDs late, mode 2: every other workgroup of each XCD late.
  int stagger_mode, stagger_ticks;
  // ---- RMSNorm folded into the projections (round 6; LlamaRMSNorm modeling_llama.py:76-90 feeding :351-353 / :240).
  // y = g * (x * rstd) followed by y W^T equals rstd (.) ((x * g) W^T): the row scale commutes with the product, the column scale does not.
  // PRODUCER side (gemm256 bf16 + residual whole-row epilogue and its split-K fix-up; N % 256 == 0): besides C = the new residual-stream
  // rows h, the launch leaves nrm_xg [M, ldc] = bf16(h * nrm_g) (h = the ROUNDED output, as the stand-alone norm would read it) and
  // nrm_ss [M, N / 256] = one fp32 partial of sum(h^2) per row and column tile (fixed lane order: deterministic).
  const bf16_t* nrm_g;
  bf16_t* nrm_xg;
  float* nrm_ss;
  // CONSUMER side (fused QKV + RoPE and fused gate|up + SwiGLU launches): output row m is multiplied by rstd[m] in fp32 before its single
  // rounding to bf16 (the tile image the fused epilogues read). rs_ss != null: rstd[m] = 1 / sqrt(sum_j rs_ss[m, j] / K + rs_eps) over
  // rs_parts partials, and the workgroups of column tile 0 store it to rs_rstd [M] (the backward needs it); rs_ss == null: rs_rstd is read.
  const float* rs_ss;
  float* rs_rstd;
  int rs_parts;
  float rs_eps;
};
