/* mla_hip.h -- C ABI of libmla_hip.so: the MI355X (gfx950) kernels behind the MLA-Llama2-7B training hot path.
 *
 * There is no FFI in the reference (ZhuoyangLiu2005/MLA is pure PyTorch + the flash_attn wheel); each entry point
 * replaces the PyTorch / flash-attn call sites cited next to it (paths relative to the reference root). The Python
 * binding a maintainer would add is shown in INTEGRATION.md and implemented in mla_amd/hip.py (ctypes).
 *
 * Contract
 *  - plain pointers + sizes only; every buffer (inputs, outputs, workspace) is allocated and owned by the caller
 *    (PyTorch); kernels never allocate, free or retain pointers past the call;
 *  - launch-only on `stream`, never synchronise, never touch the default stream; re-entrant, no global mutable state
 *    (forward runs on the main thread, backward on autograd worker threads);
 *  - return 0 on success, a negative value for argument / shape / alignment violations (checked on the host before
 *    launch), a positive hipError_t if the launch failed; mla_last_error() returns a thread-local message;
 *  - bf16 tensors are raw uint16 bit patterns, row-major, unit inner stride; statistics, losses and master gradients
 *    are fp32; indices are int64 as torch provides them (int32 where noted).
 */
#ifndef MLA_HIP_H
#define MLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mla_stream_t; /* == hipStream_t */

/* ---- library ---- */
const char* mla_last_error(void);
int mla_query(int what); /* 0: ABI version, 1: compiled gfx arch (950), 2: wavefront size (64), 3: 1 if the opt-in experiment kernels are compiled in */
/* sha256[:16] over the gemm256 kernel family's sources, stamped at build time (build.sh): bench.py compares it with the id recorded
 * next to the PMC traffic figures in profiles/ and reports roofline.traffic_stale when they differ */
const char* mla_gemm_source_id(void);
/* hardware-assumption self test (ds_read_b64_tr_b16 lane map, global_load_lds destination order) */
int mla_selftest(const void* src_1k, int* out_tr_256, void* out_glds_1k, mla_stream_t stream);
/* box calibration (bench.py's `box` block): `blocks` workgroups x 8 waves issue v_mfma_f32_16x16x32_bf16 back to back on the caller's
 * operands (512 x 24 x 8 bf16 = 196 608 bytes, 16-B aligned, e.g. N(0, 1)) for `iters` x 64 MFMAs per wave -- no LDS, no memory traffic:
 * what the matrix cores of this box sustain on random data under its power cap. out: blocks x 512 floats (sink);
 * *flops_per_launch (may be NULL) = blocks x 8 x iters x 64 x 16 384. Time it with events around the launches. */
int mla_calib_mfma(const void* operands, float* out, int blocks, int iters, double* flops_per_launch, mla_stream_t stream);
/* dispatch probe for the one-launch attention backward (mla_attn_bwd with head_sync): `blocks` workgroups of that kernel's shape each
 * take a start ticket from out[0] (caller zeroes `out`, 1 + 2 * blocks ints), stay resident ~hold_us, and record out[1 + 2 L] = ticket,
 * out[2 + 2 L] = HW_REG_XCC_ID of workgroup L. The caller decides: 8 XCDs, XCC_ID == L & 7, start order == id order per XCD. */
int mla_dispatch_probe(int* out, int blocks, int hold_us, mla_stream_t stream);

/* ---- GEMM: every nn.Linear forward / dgrad / wgrad on the path --------------------------------------------------
 * transformers/models/llama/modeling_llama.py:240 (gate/up/down), :351-353 (q/k/v), :390 (o_proj), :1254 (lm_head);
 * util/nn_utils.py:21-34; models/diffusion/models.py:112-123,173-189; models/mla/fuser/contrastive.py:173-183,208.
 * C[M,N] = alpha * sum_k Aop(m,k) Bop(n,k) (+ bias[n]) (+ R[m,n]) (+ C when accumulate).
 * a_mode/b_mode: 0 = operand stored [rows][K] (k contiguous), 1 = stored [K][rows] (reduction-major).
 * forward y = x W^T: (0,0); dgrad dx = dy W: (0,1); wgrad dW = dy^T x: (1,1). out_fp32: C is float (else bf16). */
int mla_gemm_bf16(const void* A, const void* B, void* C, const void* R, const void* bias, int M, int N, int K, int lda, int ldb,
                  int ldc, int ldr, int a_mode, int b_mode, int out_fp32, int accumulate, float alpha, int force_generic,
                  mla_stream_t stream);
/* same contract + a caller-owned scratch buffer (>= 64 MiB for full effect): the 256x256 kernel then cuts the tiles of its last,
 * partially filled round of workgroups into K-slices (fp32 partials in the workspace, fix-up pass); deterministic */
int mla_gemm_bf16_ws(const void* A, const void* B, void* C, const void* R, const void* bias, int M, int N, int K, int lda, int ldb,
                     int ldc, int ldr, int a_mode, int b_mode, int out_fp32, int accumulate, float alpha, int force_generic,
                     float* workspace, size_t workspace_bytes, mla_stream_t stream);
/* mla_gemm_bf16_ws with fp32 output (no bias / residual) that ALSO leaves sum(C^2) of the final values -- after `accumulate` -- as
 * *sq_slots partial sums in sq_out (capacity in floats >= mla_gemm_sq_slots(M, N, K, workspace_bytes)): the contribution of a
 * weight gradient to the clipping norm (training/strategies/fsdp.py:308-310) without a second pass over the fp32 gradient buffer.
 * Fixed partial order, deterministic. Shapes of the 256x256 kernel only (M, N >= 256, K % 64 == 0, N % 8 == 0): error otherwise. */
int mla_gemm_sq_slots(int M, int N, int K, size_t workspace_bytes);   /* partials the launch below writes; -1 = shape not supported */
/* Main loop of the 256x256 kernel's k-contiguous instantiations (all fused forms included): 1 = hand-scheduled assembly (default; needs
 * K % 128 == 0 and the whole-row epilogue, otherwise the launch uses the other one by itself), 0 = compiler-scheduled; any other value
 * only queries. Returns the mode in force. Results are bit-identical either way -- the switch exists for A/B measurements and tests.
 * Environment MLA_GEMM_KLOOP=0 sets the initial mode. */
int mla_gemm_kloop(int mode);
/* CUs the GEMM launches plan their rounds for (split-K tail of the last, partially filled round): n >= 8 and a multiple of 8 = plan
 * for n CUs (multi-GPU runs: the CUs RCCL's kernels leave free), 0 = the device's count (default; also MLA_GEMM_CUS in the
 * environment), n < 0 = query. Returns the setting in force, -1 for a bad argument. It only changes which tiles are split along K:
 * results agree to fp32 rounding across settings and are deterministic for a given one (K-slices are summed in a fixed order). */
int mla_gemm_cus(int n);
int mla_gemm_bf16_ws_sq(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int accumulate,
                        float alpha, float* workspace, size_t workspace_bytes, float* sq_out, int sq_capacity, int* sq_slots,
                        mla_stream_t stream);

/* fused q|k|v projection + rotary embedding (LlamaAttention.forward modeling_llama.py:351-361 + apply_rotary_pos_emb :184-208):
 * C[M, N] = A[M, K] B[N, K]^T in bf16 with columns [0, rope_cols) rotated per head of 128 (position = row % S, tables [S, 64]
 * fp32) in the GEMM epilogue; bit-identical to mla_gemm_bf16 followed by mla_rope_inplace. M, N >= 256, K % 64 == 0. */
int mla_gemm_qkv_rope(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, const float* rope_cos,
                      const float* rope_sin, int S, int rope_cols, mla_stream_t stream);

/* fused gate|up projection + SwiGLU (LlamaMLP.forward modeling_llama.py:240): gu[M, 2I] = x[M, K] wgu[2I, K]^T (wgu = gate rows then up
 * rows), act[M, I] = silu(gate) * up and, if actT != NULL, actT[I, ldt] = act^T, all from one GEMM launch; bit-identical to
 * mla_gemm_bf16 followed by mla_swiglu_fwd_dual. M >= 256, I % 128 == 0, K % 64 == 0. act may be NULL when actT is given (the
 * recomputation of a checkpointed layer needs the product in the wgrad layout only), gu may be NULL when act is given (the forward of a
 * checkpointed layer keeps nothing but its input). */
int mla_gemm_gateup_swiglu(const void* x, const void* wgu, void* gu, void* act, void* actT, int M, int I, int K, int lda, int ldb,
                           long long ldt, mla_stream_t stream);
/* fused down-projection dgrad + SwiGLU backward (autograd of LlamaMLP.forward modeling_llama.py:240): d(act) = dy[M, K] wT[I, K]^T is
 * consumed in the GEMM epilogue -- gate|up (gu [M, 2I]) is read there and d(gate|up) is written in both layouts, dgu [M, 2I] and
 * dguT [2I, ldt]; bit-identical to mla_gemm_bf16 followed by mla_swiglu_bwd_t. M, I >= 256, K % 64 == 0, M % 8 == 0. */
int mla_gemm_dact_swiglu_bwd(const void* dy, const void* wT, const void* gu, void* dgu, void* dguT, int M, int I, int K, int lda,
                             int ldb, long long ldt, mla_stream_t stream);

/* ---- RMSNorm folded into the projections (round 6): fused RMSNorm + QKV + RoPE and fused RMSNorm + gate|up + SwiGLU, one launch each.
 * LlamaRMSNorm (modeling_llama.py:76-90) feeds only projections (:351-353 q/k/v, :240 gate/up), and y W^T with y = g * (x * rstd)
 * equals rstd (.) ((x * g) W^T): the row scale commutes with the product. The GEMM that PRODUCES the residual-stream rows x
 * (LlamaDecoderLayer :744 / :750: o_proj / down_proj + residual) leaves, besides C = x, xg = bf16(x * g) with g = the NEXT norm's weight
 * and ss [M, N / 256] = one fp32 partial of sum(x^2) per row and 256-column tile (mla_gemm_res_norm; N % 256 == 0; deterministic; split-K
 * tail through `workspace` like mla_gemm_bf16_ws). The projection that CONSUMES them (the _rs forms; A = xg) multiplies its fp32
 * accumulator rows by rstd[m] = 1 / sqrt(sum_j ss[m, j] / K + eps) before the single rounding to bf16 the fused RoPE / SwiGLU epilogues
 * start from, and stores rstd [M] for the backward; with ss == NULL it reads rstd [M] instead. mla_rmsnorm_prep makes xg and rstd for rows
 * that do not come out of a GEMM (first layer, recomputation). The stand-alone norm pass disappears from the forward; the backward is
 * unchanged (it needs x and rstd only). Rounding points: x * g is rounded once where the reference rounds x * rstd and then g * that. */
int mla_gemm_res_norm(const void* A, const void* B, void* C, const void* R, const void* g, void* xg, float* ss, int M, int N, int K,
                      int lda, int ldb, int ldc, int ldr, float* workspace, size_t workspace_bytes, mla_stream_t stream);
int mla_gemm_qkv_rope_rs(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, const float* rope_cos,
                         const float* rope_sin, int S, int rope_cols, const float* ss, int parts, float eps, float* rstd,
                         mla_stream_t stream);
int mla_gemm_gateup_swiglu_rs(const void* x, const void* wgu, void* gu, void* act, void* actT, int M, int I, int K, int lda, int ldb,
                              long long ldt, const float* ss, int parts, float eps, float* rstd, mla_stream_t stream);
int mla_rmsnorm_prep(const void* x, const void* w, void* xg, float* rstd, int rows, int H, float eps, mla_stream_t stream);

/* ---- RMSNorm: LlamaRMSNorm.forward modeling_llama.py:76-90; timm RmsNorm in FinalLayer (models/diffusion/models.py:177) */
int mla_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int H, float eps, mla_stream_t stream);
int mla_rmsnorm_bwd_blocks(int rows);
/* dx = dres + d(rmsnorm)/dx ; dw (+)= sum_rows dy * x_hat ; workspace >= mla_rmsnorm_bwd_blocks(rows)*H floats */
int mla_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx, float* dw,
                    int dw_accumulate, int rows, int H, float* workspace, size_t workspace_bytes, mla_stream_t stream);
/* ---- timm==0.9.10 RmsNorm (FinalLayer.norm_final, models/diffusion/models.py:18,177; pin pyproject.toml:44): timm v0.9.10
 * timm/layers/fast_norm.py::rms_norm computes v = torch.var(x, dim=-1, keepdim=True) (unbiased, mean-subtracted) and returns
 * x * rsqrt(v + eps) * weight -- not the mean-of-squares norm above. mean/rstd [rows] fp32 are saved for the backward. */
int mla_timm_rmsnorm_fwd(const void* x, const void* w, void* y, float* mean, float* rstd, int rows, int H, float eps,
                         mla_stream_t stream);
/* dx = d(timm rms_norm)/dx ; dw (+)= sum_rows dy * x * rstd ; workspace >= mla_rmsnorm_bwd_blocks(rows)*H floats */
int mla_timm_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx, float* dw,
                         int dw_accumulate, int rows, int H, float* workspace, size_t workspace_bytes, mla_stream_t stream);
/* bias gradients of every nn.Linear with a bias on the path (autograd of util/nn_utils.py:21-34, models/diffusion/models.py:112-123):
 * out[n] (+)= sum_r dy[r][n]; workspace >= mla_colsum_blocks(rows)*N floats */
int mla_colsum_blocks(int rows);
int mla_colsum_bf16(const void* dy, float* out, int accumulate, int rows, int N, int ld, float* workspace, size_t workspace_bytes,
                    mla_stream_t stream);
/* nn.LayerNorm forward in the frozen vision tokenizer: models/mla/image/vision_tokenizer.py:21-24 */
int mla_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int rows, int H, float eps, mla_stream_t stream);

/* ---- RoPE in place on the q and k slices of a packed qkv buffer: apply_rotary_pos_emb modeling_llama.py:184-208 */
int mla_rope_inplace(void* buf, const float* cos_t, const float* sin_t, long long tokens, int S, int nheads, int D, int ld,
                     int q_off, int k_off, int backward, mla_stream_t stream);

/* ---- SwiGLU: LlamaMLP.forward modeling_llama.py:240; gu = [gate | up] per row */
int mla_swiglu_fwd(const void* gu, void* act, long long rows, int I, mla_stream_t stream);
int mla_swiglu_bwd(const void* dact, const void* gu, void* dgu, void* act_out, long long rows, int I, mla_stream_t stream);
/* ---- activations of the small heads: kind 0 GELU(erf) (MLPProjector util/nn_utils.py:21-34), 1 GELU(tanh) (timm Mlp in ActionEmbedder /
 * FinalLayer, models/diffusion/models.py:112-123,173-189), 2 ReLU (contrastive heads models/mla/fuser/contrastive.py:173-183; Point-PN
 * Point_PN.py:179,209,218), 3 SiLU (TimestepEmbedder models/diffusion/models.py:34-38) */
int mla_act_fwd(const void* x, void* y, long long n, int kind, mla_stream_t stream);
int mla_act_bwd(const void* dy, const void* x, void* dx, long long n, int kind, mla_stream_t stream);

/* ---- casts / adds / row gathers (sequence assembly of PrismaticVLM.forward models/vlm/prismatic.py:981-1038) */
int mla_cast_f32_to_bf16(const float* x, void* y, long long n, mla_stream_t stream);
int mla_cast_bf16_to_f32(const void* x, float* y, long long n, mla_stream_t stream);
int mla_add_bf16(const void* a, const void* b, void* y, long long n, mla_stream_t stream);
int mla_gather_rows_bf16(const void* src, const long long* idx, void* out, long long rows, int H, int scatter, mla_stream_t stream);

/* ---- tile transposes feeding the all-NT backward GEMMs (dx = dy W and dW = dy^T x of every Linear on the decoder path,
 * i.e. autograd of modeling_llama.py:240,351-353,390): dst[c][r] = src[r][c]; the two fused forms recompute the
 * normalised input (modeling_llama.py:85-90 with the saved rstd) / the SwiGLU product (:240) straight into [C, R]. */
int mla_transpose_bf16(const void* src, void* dst, long long R, int C, long long ld, long long ldt, mla_stream_t stream);
int mla_rmsnorm_apply_t(const void* x, const void* w, const float* rstd, void* dst, long long rows, int H, long long ldt,
                        mla_stream_t stream);
int mla_swiglu_fwd_t(const void* gu, void* dst, long long rows, int I, long long ldt, mla_stream_t stream);
/* SwiGLU forward (LlamaMLP.forward modeling_llama.py:240) writing the product in both layouts in one pass: act [rows, I] for the
 * down projection and actT [I, ldt] kept for the backward's wgrad GEMM (instead of recomputing it from gu there) */
int mla_swiglu_fwd_dual(const void* gu, void* act, void* actT, long long rows, int I, long long ldt, mla_stream_t stream);
/* SwiGLU backward writing both layouts in one pass: dgu [rows, 2I] for the dgrad GEMM and dguT [2I, ldt] for the wgrad GEMM
 * (modeling_llama.py:241 LlamaMLP backward; saves re-reading the 2I-wide gradient for a separate transpose) */
int mla_swiglu_bwd_t(const void* dact, const void* gu, void* dgu, void* dguT, long long rows, int I, long long ldt, mla_stream_t stream);

/* ---- embedding: LlamaModel.embed_tokens modeling_llama.py:975-976 (deterministic, atomics-free backward).
 * bwd: grad[ids[t]] += dy[t] in ascending token order. workspace: fp32 [tokens / 64, H] scratch or NULL; with it, every aligned
 * batch of 64 tokens that carries one id (padding) is summed first by its own workgroup and enters the walk as one row. */
int mla_embedding_fwd(const long long* ids, const void* table, void* out, long long tokens, int H, int vocab, mla_stream_t stream);
int mla_embedding_bwd(const long long* ids, const void* dy, float* grad, float* workspace, long long tokens, int H, int vocab,
                      mla_stream_t stream);

/* ---- optimizer: AdamW (training/strategies/fsdp.py:257) over the local fp32 shard + bf16 compute copy; clip (:308-310) */
int mla_adamw_step(float* p, const float* g, float* m, float* v, void* p16, long long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, const float* grad_scale, mla_stream_t stream);
/* the same update over a flat range laid out [weight-decayed | not decayed]: elements [0, n_decay) get weight_decay, the rest none --
 * both AdamW parameter groups of a sharding unit (training/strategies/fsdp.py:231-257) in one launch */
int mla_adamw_step_groups(float* p, const float* g, float* m, float* v, void* p16, long long n, long long n_decay, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int step, const float* grad_scale, mla_stream_t stream);
int mla_sumsq_f32(const float* x, long long n, float* out, int accumulate, float* workspace, size_t workspace_bytes,
                  mla_stream_t stream);
/* out[0] (+)= sum(partial[0 .. n)), fixed order: second stage of the gradient norm over mla_gemm_bf16_ws_sq partials */
int mla_sum_partials(const float* partial, int n, float* out, int accumulate, mla_stream_t stream);
int mla_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, mla_stream_t stream);

/* ---- diffusion: GaussianDiffusion.q_sample models/diffusion/gaussian_diffusion.py:214-229 */
int mla_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ac, const float* sqrt_1mac, float* out,
                 int batch, int per, int nsteps, mla_stream_t stream);

/* ---- causal attention (replaces flash_attn 2.5.5: modeling_llama.py:420-597; math :371-380). head_dim 128.
 * q/k/v: first element of each slice of the packed [B*S, ld_qkv] buffer; o / dout: [B*S, ld_o]; lse, delta: [B,H,S].
 * seqlens (int32 [B] or NULL): right-padded rows q >= seqlens[b] give zero output / zero gradients. */
int mla_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* seqlens, int B, int S, int H,
                 int head_dim, long long ld_qkv, long long ld_o, float scale, mla_stream_t stream);
/* Shared-prefix sequences (round 6, opt-in; models/mla/model_mla.py:148-180 tiles every sample R times although, without a point cloud,
 * the R copies differ only in their last rows [t, x, </s>]): one sequence [prefix | R suffix groups of grp_len rows]. Rows >= grp_start
 * belong to group (row - grp_start) / grp_len; a query attends to the prefix and, causally, to its OWN group. The RoPE positions of the
 * suffix rows come from the caller's tables (row s of rope_cos / rope_sin is the table row of position pos(s)). grp_len == 0: plain causal.
 * grp_starts (int32 [B] on the device, or NULL): a first suffix row per sample for ragged prompts (valid rows of sample b: [0, seqlens[b])).
 * mla_attn_bwd_g = mla_attn_bwd_t (transposed copies and head_sync optional) with the same grouping; rope_per_sample = 1: the tables are
 * [B * S, 64], one block of S rows per sample (per-sample positions), instead of one [S, 64] table. */
int mla_attn_fwd_g(const void* q, const void* k, const void* v, void* o, float* lse, const int* seqlens, int B, int S, int H,
                   int head_dim, long long ld_qkv, long long ld_o, float scale, int grp_start, int grp_len, const int* grp_starts,
                   mla_stream_t stream);
int mla_attn_bwd_g(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const int* seqlens,
                   void* dq, void* dk, void* dv, float* delta, int B, int S, int H, int head_dim, long long ld_qkv, long long ld_o,
                   float scale, const float* rope_cos, const float* rope_sin, void* dqT, void* dkT, void* dvT, void* oT,
                   long long ldt, int* head_sync, long long head_sync_ints, int grp_start, int grp_len, const int* grp_starts,
                   int rope_per_sample, mla_stream_t stream);
/* rope_cos / rope_sin ([S, 64] fp32, both or neither): when given, dq and dk are written with the backward of apply_rotary_pos_emb
 * (modeling_llama.py:184-208) already applied -- the same values mla_rope_inplace(backward = 1) would produce on them afterwards.
 * Launch form (mla_attn_bwd and mla_attn_bwd_t; results are bit-identical either way): with head_sync == NULL the dQ kernel and the
 * dK / dV kernel are two launches. With head_sync given, ONE launch runs both block types and the dQ blocks hand delta to the dK / dV
 * blocks of their head through two integer counters per head IN THE CALLER'S BUFFER: head_sync = mla_attn_bwd_sync_ints(B, H) ints
 * (4-B aligned), zero before the first call; every launch leaves it zero again (the last consumer of a head resets its pair), so one
 * buffer serves any sequence of shapes on a stream and a captured launch replays correctly. One buffer per stream in flight. The
 * library allocates nothing and keeps nothing. The one-launch form assumes workgroup id L runs on XCD L & 7 and that workgroups
 * start in id order per XCD (mla_dispatch_probe measures both; mla_amd/hip.py passes head_sync only on a device where they hold); a
 * wait that lasts 30 s of wall clock traps. MLA_ATTN_BWD_MERGED=0 forces two launches whatever is passed.
 * `delta` is written by the call and only meaningful after it has completed. */
long long mla_attn_bwd_sync_ints(int B, int H);
int mla_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const int* seqlens,
                 void* dq, void* dk, void* dv, float* delta, int B, int S, int H, int head_dim, long long ld_qkv, long long ld_o,
                 float scale, const float* rope_cos, const float* rope_sin, int* head_sync, long long head_sync_ints,
                 mla_stream_t stream);
/* mla_attn_bwd + token-contiguous copies dqT / dkT / dvT / oT [H * head_dim, ldt] (column = b * S + s; columns >= B * S untouched)
 * of dq / dk / dv / o: the k-contiguous operands of the q|k|v and o projection wgrad GEMMs (autograd of modeling_llama.py:371-380),
 * written from the registers that hold the rows instead of by four transpose passes. All four or none; S % 4 == 0, ldt % 4 == 0. */
int mla_attn_bwd_t(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const int* seqlens,
                   void* dq, void* dk, void* dv, float* delta, int B, int S, int H, int head_dim, long long ld_qkv, long long ld_o,
                   float scale, const float* rope_cos, const float* rope_sin, void* dqT, void* dkT, void* dvT, void* oT,
                   long long ldt, int* head_sync, long long head_sync_ints, mla_stream_t stream);
/* Five-product form of the backward -- an EXPERIMENT kernel (measured no faster than the seven-product pair on MI355X):
 * compiled only into the experiment build (build.sh MLA_EXPERIMENTAL=1, mla_query(3) == 1); the product library exports the symbol
 * and rejects the call with a negative code.
 * the dK / dV kernel hands dS^T to a one-product dQ kernel through the
 * caller-owned workspace `ws` (mla_attn_bwd_ws_bytes(B, S, H) bytes, 16-B aligned) instead of both kernels recomputing Q K^T and
 * dO V^T; delta = rowsum(O o dO) is its own pass. Same outputs and argument meaning as mla_attn_bwd_t (dqT / dkT / dvT / oT: all or
 * none); deterministic (no atomics). Replaces the flash-attn backward behind modeling_llama.py:531-553. */
long long mla_attn_bwd_ws_bytes(int B, int S, int H);
int mla_attn_bwd_ws(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const int* seqlens,
                    void* dq, void* dk, void* dv, float* delta, int B, int S, int H, int head_dim, long long ld_qkv, long long ld_o,
                    float scale, const float* rope_cos, const float* rope_sin, void* dqT, void* dkT, void* dvT, void* oT, long long ldt,
                    void* ws, long long ws_bytes, mla_stream_t stream);

/* ---- inference with a cached prefix (mla_amd/infer.py; replaces 7 of the 8 whole forwards of MLA.predict_action_diff,
 * models/mla/model_mla.py:742-775 + models/diffusion/gaussian_diffusion.py:608-688). HBM-bound: algorithmic bytes = the weight matrix /
 * the head's cached K and V rows.
 * mla_gemv_bf16: out row m = x[m] . W^T (+ residual[m]) for M <= 8 rows; W [N, K] k-contiguous is read exactly once (16 B per lane,
 *   non-temporal), x is held in LDS (M x K x 2 bytes <= 160 KiB), fp32 accumulation. Row m of `out` is written at
 *   out + (m / rows_per_batch) * out_batch_stride + (m % rows_per_batch) * ldo, so the q|k|v rows of the new tokens can land directly in
 *   the per-sample cache slots. Replaces the nn.Linear calls of modeling_llama.py:240, 351-353, 390 for the suffix rows.
 *   pre: what is applied to the input rows on their way into LDS -- 0 nothing; 1 LlamaRMSNorm with weight pre_w and eps
 *   (modeling_llama.py:76-90, the arithmetic of mla_rmsnorm_fwd); 2 SwiGLU: x rows are packed gate|up [2 K], the input is
 *   silu(gate) * up (modeling_llama.py:240, the arithmetic of mla_swiglu_fwd).
 *   rope_cos / rope_sin ([rows_per_batch, 64] fp32, both or neither; no residual then): columns [0, rope_cols) are rotated per head of
 *   128 in the epilogue (apply_rotary_pos_emb modeling_llama.py:184-208, the arithmetic of mla_rope_inplace on the bf16-rounded
 *   projection; row m uses table row m % rows_per_batch) -- with pre = 1 this is north_star's "fused RMSNorm + RoPE + QKV" as ONE kernel,
 *   on the inference path. */
int mla_gemv_bf16(const void* x, long long ldx, const void* W, long long ldw, void* out, long long ldo, long long out_batch_stride,
                  int rows_per_batch, const void* residual, long long ld_res, int M, int N, int K, int pre, const void* pre_w, float eps,
                  const float* rope_cos, const float* rope_sin, int rope_cols, mla_stream_t stream);
/* mla_attn_decode: R <= 8 new query rows per (sample, head) -- rows [S_kv - R, S_kv) of the packed q|k|v cache (row stride ld, sample stride
 *   batch_stride, q / k / v = the three slices' first elements) -- against keys / values [0, S_kv - R + r] (causal, scale applied to the
 *   scores, fp32 softmax, P rounded to bf16 before P V like the flash kernel). o: [B * R, H * 128] bf16. head_dim 128.
 *   Replaces the attention of modeling_llama.py:371-380 for the suffix rows. */
int mla_attn_decode(const void* q, const void* k, const void* v, void* o, int B, int H, int head_dim, int S_kv, int R, long long ld,
                    long long batch_stride, long long ld_o, float scale, mla_stream_t stream);

/* ---- losses: CrossEntropyLoss modeling_llama.py:1258-1269; InfoNCE models/mla/fuser/contrastive.py:208-215 */
int mla_ce_fwd(const void* logits, int logits_fp32, long long ld, const long long* labels, float* loss, float* lse, int rows,
               int ncols, long long ignore_index, mla_stream_t stream);
int mla_ce_bwd(const void* logits, int logits_fp32, long long ld, const long long* labels, const float* lse, const float* gscale,
               float inv_count, void* dlogits, long long ldd, int rows, int ncols, long long ignore_index, mla_stream_t stream);
int mla_infonce_bwd(const float* L, const float* rlse, const float* clse, const float* gscale, void* dL, int M, int Mp,
                    mla_stream_t stream);
int mla_l2norm_fwd(const void* x, void* y, float* norms, int rows, int ncols, float eps, mla_stream_t stream);
int mla_l2norm_bwd(const void* dy, const void* y, const float* norms, void* dx, int rows, int ncols, mla_stream_t stream);

/* ---- point-cloud tokenizer (forward): models/mla/pointcloud/backbone/Point_PN.py
 * furthest_point_sample :6-21 (start index = explicit input), knn_point :62-73, LGA :115-158 + PosE_Geo :231-249,
 * BatchNorm in train mode :179,209,215, Pooling :166-169; projection models/mla/fuser/contrastive.py:5-45 */
int mla_project_points(const float* xyz, const float* consts21, long long* idx, unsigned char* valid, int n, float W, float Hh,
                       float stride, int ph, int pw, mla_stream_t stream);
int mla_fps(const float* xyz, const long long* start, long long* out, int B, int N, int npoint, mla_stream_t stream);
int mla_knn(const float* xyz, const float* centers, int* out, int B, int N, int G, int k, mla_stream_t stream);
int mla_gather_rows_f32(const float* src, const long long* idx, float* out, int B, int N, int G, int W, mla_stream_t stream);
int mla_lga_prep(const float* xyz, const void* feats, const long long* fps_idx, const int* knn, void* rows, float* lc_xyz, int B,
                 int N, int G, int K, int C, float alpha, float beta, mla_stream_t stream);
int mla_colstats_blocks(long long rows);
int mla_colstats_bf16(const void* x, float* mean, float* var, long long rows, int C, int ld, float* workspace, size_t workspace_bytes,
                      mla_stream_t stream);
int mla_bn_apply(const void* x, const float* mean, const float* var, const void* w, const void* b, const void* res, void* y,
                 long long rows, int C, float eps, int relu, mla_stream_t stream);
int mla_maxpool_k(const void* x, void* out, long long groups, int K, int C, mla_stream_t stream);
/* backward of the two (trainable point tokenizer, stage "pretrain" with use_pointcloud): feature gradient of lga_prep -- a gather per
 * target point with every sum in a fixed order (deterministic, no atomics; every element of dfeats [B, N, C] fp32 is written) -- and
 * of the max over the K neighbours (first maximum). Reference: models/mla/pointcloud/backbone/Point_PN.py:115-158 (autograd of LGA). */
int mla_lga_prep_bwd(const void* drows, const long long* fps_idx, const int* knn, float* dfeats, int B, int N, int G, int K, int C,
                     mla_stream_t stream);
int mla_maxpool_k_bwd(const void* x, const void* dy, void* dx, long long groups, int K, int C, mla_stream_t stream);

/* ---- vision tokenizer (forward): models/mla/image/vision_tokenizer.py  Conv2d patchify :110,122 (im2col + GEMM),
 * LocalAttention :26-47 (avg-pool, 3x3 window attention) */
int mla_im2col_patch(const void* pix, int pix_fp32, void* rows, int B, int CT, int Himg, int Wimg, int P, int Kpad,
                     mla_stream_t stream);
int mla_avgpool_tokens(const void* x, void* y, int B, int gh, int gw, int C, int cs, mla_stream_t stream);
int mla_local_attn(const void* q, const void* kv, void* out, int B, int gh, int gw, int C, int cs, int heads, float scale,
                   mla_stream_t stream);
/* backward of the two (stage "pretrain": models/vlm/prismatic.py:415-447 trains vision_tower_2d): window-attention gradients
 * (dq [windows, C], dkv [tokens, 2C]; every k/v row belongs to one window) and avg-pool backward fused with a second addend */
int mla_local_attn_bwd(const void* q, const void* kv, const void* dout, void* dq, void* dkv, int B, int gh, int gw, int C, int cs,
                       int heads, float scale, mla_stream_t stream);
/* CLIP-style preprocessing of uint8 HWC frames on the GPU (the reference runs CLIPImageProcessor on the CPU per frame,
 * vla/datasets/datasets.py:52-69): PIL-exact 8-bit separable bicubic resize (host-built 2^22 fixed-point taps), 1/255 rescale,
 * mean/std normalisation, optional all-ones mask channel; mean3 / std3 are HOST pointers */
int mla_clip_preprocess(const unsigned char* img, int B, int Hin, int Win, const int* bounds_h, const int* coef_h, int ks_h,
                        const int* bounds_v, const int* coef_v, int ks_v, void* out, int out_fp32, int OH, int OW, const float* mean3,
                        const float* std3, int mask_channel, mla_stream_t stream);
int mla_avgpool_tokens_bwd(const void* dy, const void* other, void* dx, int B, int gh, int gw, int C, int cs, mla_stream_t stream);

/* ---- batched GEMM (two-level batch: outer = sample, inner = head) for the generation heads' nn.MultiheadAttention products
 * (models/mla/generation/models.py:44,103-122: QK^T, PV and their gradients). Same operand modes as mla_gemm_bf16; batch
 * (o, i) adds o*s?o + i*s?i ELEMENTS to each base pointer; n_inner >= 1. No bias / residual. */
int mla_gemm_batched_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int a_mode, int b_mode,
                          int out_fp32, float alpha, int n_outer, int n_inner, long long sAo, long long sAi, long long sBo,
                          long long sBi, long long sCo, long long sCi, mla_stream_t stream);

/* ---- post-training generation heads (BASELINE config[3]): models/mla/generation/models.py
 * nn.TransformerDecoderLayer :103-122 / TransformerBlock :39-65 pieces: softmax(+attention dropout) over the first nvalid of
 * ncols key columns, Dropout (+ residual add), DropPath (per-sample scale), LayerNorm with backward; PointCloudGenerationModule
 * :351-386: sequence mean :359, BatchNorm1d(train) backward :335; chamfer_distance_l2 gen_loss.py:12-18; image loss
 * models/vlm/prismatic.py:780-816 with ImageGenerationModule._generate_generated_patches models.py:226-286 evaluated for the
 * all-true ROI mask (use_roi = False, scripts/post_rlbench.sh:26) and images_to_patches utils.py:7-18 addressing.
 * Dropout decisions are a counter-based hash of (seed, element index): backward regenerates the forward mask. */
int mla_softmax_rows_fwd(const float* scores, void* P, void* Pd, long long rows, int ncols, int nvalid, float p,
                         unsigned long long seed, mla_stream_t stream);
/* dS = P' o (g - sum(g o P')), g = dPd o mask / (1 - p), P' = P / sum(P) (the bf16 probabilities renormalised in fp32).
 * dPd: fp32 (dpd_fp32 = 1, the product path: dP is never rounded, like torch's fused attention) or bf16. */
int mla_softmax_rows_bwd(const void* dPd, int dpd_fp32, const void* P, void* dS, long long rows, int ncols, int nvalid, float p,
                         unsigned long long seed, mla_stream_t stream);
int mla_dropout_fwd(const void* x, const void* residual, void* y, long long n, float p, unsigned long long seed, mla_stream_t stream);
int mla_dropout_bwd(const void* dy, void* dx, long long n, float p, unsigned long long seed, mla_stream_t stream);
int mla_scale_batch(const void* x, const float* scale, void* y, long long batch, long long per, mla_stream_t stream);
int mla_layernorm_stats_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, long long rows, int H,
                            float eps, mla_stream_t stream);
int mla_layernorm_bwd_blocks(long long rows);
int mla_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx, float* dw,
                      float* db, int accumulate, long long rows, int H, float* workspace, size_t workspace_bytes,
                      mla_stream_t stream);
int mla_seqmean_fwd(const void* x, void* y, int B, int S, int C, mla_stream_t stream);
int mla_seqmean_bwd(const void* dy, void* dx, int B, int S, int C, mla_stream_t stream);
int mla_bn_bwd_blocks(long long rows);
int mla_bn_bwd(const void* dy, const void* x, const float* mean, const float* var, const void* w, void* dx, float* dw, float* db,
               int accumulate, long long rows, int C, float eps, float* workspace, size_t workspace_bytes, mla_stream_t stream);
int mla_chamfer_fwd(const float* pred, const float* gt, float* d1, int* i1, float* d2, int* i2, float* loss, int B, int N, int M,
                    float* workspace, size_t workspace_bytes, mla_stream_t stream);
int mla_chamfer_bwd(const float* pred, const float* gt, const float* d1, const int* i1, const float* d2, const int* i2,
                    const float* gscale, float* dpred, int B, int N, int M, mla_stream_t stream);
int mla_imgloss_fwd(const void* delta_raw, int ld, const void* curr, const void* next, int img_fp32, float* sums, int B, int CT_curr,
                    int CT_next, int HW, int ps, float clip, float* workspace, size_t workspace_bytes, mla_stream_t stream);
/* partial ROI (use_roi = True): per patch either the ROI rule or warp (bilinear translation, border clamp) + alpha blend
 * (models.py:243-283), with the ROI / background / delta-reward sums of prismatic.py:786-816; backward also yields the alpha / offset
 * head gradients (one block per patch) */
int mla_imgroi_fwd(const void* delta_raw, int ld, const void* a_raw, int lda, const void* o_raw, int ldo, const unsigned char* roi,
                   const void* curr, const void* next, int img_fp32, float* sums, int B, int CT_curr, int CT_next, int HW, int ps,
                   float clip, float shift, float* workspace, size_t workspace_bytes, mla_stream_t stream);
int mla_imgroi_bwd(const void* delta_raw, int ld, const void* a_raw, int lda, const void* o_raw, int ldo, const unsigned char* roi,
                   const void* curr, const void* next, int img_fp32, const float* coef, void* ddelta_raw, float* dalpha_raw,
                   float* doff_raw, int B, int CT_curr, int CT_next, int HW, int ps, float clip, float shift, mla_stream_t stream);
int mla_imgloss_bwd(const void* delta_raw, int ld, const void* curr, const void* next, int img_fp32, const float* gscale,
                    void* ddelta_raw, int B, int CT_curr, int CT_next, int HW, int ps, float clip, mla_stream_t stream);

/* ---- multi-GPU rehearsal (no reference counterpart): stand-in for a collective's kernel on a side stream -- `blocks` resident
 * 1024-thread workgroups stream out = a + b over n fp32 (n % 4 == 0), sleeping `sleep_ticks` x 8128 clocks after every 64 KiB chunk; lds_bytes =
 * 163840 makes each workgroup the only resident of its CU (the CU is taken from a gemm256 workgroup), 0 lets it share the CU.
 * tools/contention_rehearsal.py measures the training step's sensitivity to it (profiles/r4_contention.txt). */
int mla_side_traffic(const float* a, const float* b, float* out, long long n, int blocks, int lds_bytes, int sleep_ticks, mla_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MLA_HIP_H */
