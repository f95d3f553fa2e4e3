// Encoder-free vision tokenizer kernels: forward (vision_tower_2d is frozen on the SFT / post-training path) and the two
// backward kernels that stage 'pretrain' needs (models/vlm/prismatic.py:415-447 trains the tower there).
// Reference: models/mla/image/vision_tokenizer.py -- Conv2d(3->C, k = s = 14) patchify :110,122 (here: im2col +
// the MFMA GEMM), LocalAttention :14-47 (3x3 window attention, scale = C^-0.5).  The reference loops over samples in
// Python with a host sync each; here the whole batch is one launch per stage.
#include "common.h"
#include <math.h>

namespace {

template <typename T> __device__ __forceinline__ float ldp(const T* p);
template <> __device__ __forceinline__ float ldp<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldp<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// rows[(b, py, px)][k = c*P*P + ky*P + kx] = pix[b][c][py*P + ky][px*P + kx], zero padded to Kpad columns.
// pix has CT channels per image (RGB + mask) of which the first 3 are used.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ pix, bf16_t* __restrict__ rows, int B, int CT, int Himg,
                                                     int Wimg, int P, int Kpad) {
  const int gh = Himg / P, gw = Wimg / P;
  const long long total = (long long)B * gh * gw * Kpad;
  const int Kreal = 3 * P * P;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int k = (int)(e % Kpad);
    const long long r = e / Kpad;
    float v = 0.f;
    if (k < Kreal) {
      const int px = (int)(r % gw), py = (int)((r / gw) % gh), b = (int)(r / ((long long)gw * gh));
      const int c = k / (P * P), rem = k % (P * P), ky = rem / P, kx = rem % P;
      v = ldp<T>(pix + (((size_t)b * CT + c) * Himg + (py * P + ky)) * Wimg + (px * P + kx));
    }
    rows[e] = f2bf(v);
  }
}

// tokens [B, gh, gw, C] (channel-last rows) -> pooled [B, gh/cs, gw/cs, C], mean over the cs x cs window
__global__ __launch_bounds__(256) void avgpool_tokens_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int gh,
                                                             int gw, int C, int cs) {
  const int oh = gh / cs, ow = gw / cs;
  const long long total = (long long)B * oh * ow * C;
  const float inv = 1.f / (float)(cs * cs);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    long long r = e / C;
    const int j = (int)(r % ow), i = (int)((r / ow) % oh), b = (int)(r / ((long long)ow * oh));
    float s = 0.f;
    for (int dy = 0; dy < cs; ++dy)
      for (int dx = 0; dx < cs; ++dx) s += bf2f(x[(((size_t)b * gh + i * cs + dy) * gw + j * cs + dx) * C + c]);
    y[e] = f2bf(s * inv);
  }
}

// One block per window. q [B*oh*ow, C]; kv [B*gh*gw, 2C] (k = [:, :C], v = [:, C:]); heads x 128... generic head dim
// hd = C / heads (hd % 32 == 0 handled: 32 lanes per head, hd/32 channels per lane; heads*32 == blockDim.x = 256).
__global__ __launch_bounds__(256) void local_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv,
                                                         bf16_t* __restrict__ out, int B, int gh, int gw, int C, int cs, float scale) {
  const int oh = gh / cs, ow = gw / cs;
  const int win = blockIdx.x;
  const int j = win % ow, i = (win / ow) % oh, b = win / (ow * oh);
  const int head = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int hd = C / 8, per = hd / 32;
  const int c0 = head * hd + l * per;
  float qv[8];
  for (int t = 0; t < per; ++t) qv[t] = bf2f(q[(size_t)win * C + c0 + t]) * scale;
  const int N = cs * cs;
  float sc[16];
  float m = -INFINITY;
  for (int n = 0; n < N; ++n) {
    const int dy = n / cs, dx = n % cs;
    const size_t row = ((size_t)b * gh + i * cs + dy) * gw + j * cs + dx;
    float s = 0.f;
    for (int t = 0; t < per; ++t) s += qv[t] * bf2f(kv[row * 2 * C + c0 + t]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    sc[n] = s;
    m = fmaxf(m, s);
  }
  float den = 0.f;
  for (int n = 0; n < N; ++n) { sc[n] = __expf(sc[n] - m); den += sc[n]; }
  const float inv = 1.f / den;
  float acc[8];
  for (int t = 0; t < per; ++t) acc[t] = 0.f;
  for (int n = 0; n < N; ++n) {
    const int dy = n / cs, dx = n % cs;
    const size_t row = ((size_t)b * gh + i * cs + dy) * gw + j * cs + dx;
    const float p = sc[n] * inv;
    for (int t = 0; t < per; ++t) acc[t] += p * bf2f(kv[row * 2 * C + C + c0 + t]);
  }
  for (int t = 0; t < per; ++t) out[(size_t)win * C + c0 + t] = f2bf(acc[t]);
}

// backward of the window attention: one block per window (each k/v row belongs to exactly one window -> plain stores).
//   p = softmax_n(scale * q.k_n);  dP_n = dout.v_n;  dS_n = p_n (dP_n - sum_m p_m dP_m)
//   dq = scale * sum_n dS_n k_n;   dk_n = scale * dS_n q;   dv_n = p_n dout
__global__ __launch_bounds__(256) void local_attn_bwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv,
                                                             const bf16_t* __restrict__ dout, bf16_t* __restrict__ dq,
                                                             bf16_t* __restrict__ dkv, int B, int gh, int gw, int C, int cs, float scale) {
  const int oh = gh / cs, ow = gw / cs;
  const int win = blockIdx.x;
  const int j = win % ow, i = (win / ow) % oh, b = win / (ow * oh);
  const int head = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int hd = C / 8, per = hd / 32;
  const int c0 = head * hd + l * per;
  const int N = cs * cs;
  float qv[8], go[8];
  for (int t = 0; t < per; ++t) {
    qv[t] = bf2f(q[(size_t)win * C + c0 + t]);
    go[t] = bf2f(dout[(size_t)win * C + c0 + t]);
  }
  float sc[16], dp[16];
  float m = -INFINITY;
  for (int n = 0; n < N; ++n) {
    const size_t row = ((size_t)b * gh + i * cs + n / cs) * gw + j * cs + n % cs;
    float s = 0.f, d = 0.f;
    for (int t = 0; t < per; ++t) {
      s += qv[t] * scale * bf2f(kv[row * 2 * C + c0 + t]);
      d += go[t] * bf2f(kv[row * 2 * C + C + c0 + t]);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); d += __shfl_xor(d, o, 64); }
    sc[n] = s; dp[n] = d;
    m = fmaxf(m, s);
  }
  float den = 0.f;
  for (int n = 0; n < N; ++n) { sc[n] = __expf(sc[n] - m); den += sc[n]; }
  const float inv = 1.f / den;
  float dot = 0.f;
  for (int n = 0; n < N; ++n) { sc[n] *= inv; dot += sc[n] * dp[n]; }
  float dqa[8];
  for (int t = 0; t < per; ++t) dqa[t] = 0.f;
  for (int n = 0; n < N; ++n) {
    const size_t row = ((size_t)b * gh + i * cs + n / cs) * gw + j * cs + n % cs;
    const float ds = sc[n] * (dp[n] - dot) * scale;
    for (int t = 0; t < per; ++t) {
      dqa[t] += ds * bf2f(kv[row * 2 * C + c0 + t]);
      dkv[row * 2 * C + c0 + t] = f2bf(ds * qv[t]);
      dkv[row * 2 * C + C + c0 + t] = f2bf(sc[n] * go[t]);
    }
  }
  for (int t = 0; t < per; ++t) dq[(size_t)win * C + c0 + t] = f2bf(dqa[t]);
}

// dx[b, y, x, c] = dpooled[b, y/cs, x/cs, c] / cs^2 (+ other[b, y, x, c]): backward of avgpool_tokens fused with the sum of the
// token gradient that arrives through the k/v LayerNorm
__global__ __launch_bounds__(256) void avgpool_tokens_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ other,
                                                                 bf16_t* __restrict__ dx, int B, int gh, int gw, int C, int cs) {
  const int oh = gh / cs, ow = gw / cs;
  const long long total = (long long)B * gh * gw * C;
  const float inv = 1.f / (float)(cs * cs);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    long long r = e / C;
    const int x = (int)(r % gw), y = (int)((r / gw) % gh), b = (int)(r / ((long long)gw * gh));
    float v = bf2f(dy[(((size_t)b * oh + y / cs) * ow + x / cs) * C + c]) * inv;
    if (other) v += bf2f(other[e]);
    dx[e] = f2bf(v);
  }
}

// CLIP-style preprocessing of uint8 HWC frames (vision_tokenizer.py:98-105 = CLIPImageProcessor(size 672, crop 672, rescale 1/255,
// normalise); called per frame on the CPU by the reference's dataset, vla/datasets/datasets.py:52-69): PIL's separable bicubic resize in
// 8-bit fixed point -- horizontal pass, round to uint8, vertical pass, round to uint8 (libImaging/Resample.c; the coefficient tables
// are built on the host exactly like precompute_coeffs / normalize_coeffs_8bpc) -- then float32((double)u8 / 255 ... ) - mean) / std and
// an all-ones mask channel. One thread per output pixel; <= 5 x 5 taps.
__device__ __forceinline__ int clip8_fx(int v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : v); }
template <typename TO>
__global__ __launch_bounds__(256) void clip_preprocess_kernel(const unsigned char* __restrict__ img, const int* __restrict__ bh,
                                                              const int* __restrict__ kh, const int* __restrict__ bv,
                                                              const int* __restrict__ kv, TO* __restrict__ out, int B, int Hin, int Win,
                                                              int OH, int OW, int ksh, int ksv, float m0, float m1, float m2, float s0,
                                                              float s1, float s2, int mask_channel) {
  const long long total = (long long)B * OH * OW;
  const int CT = 3 + (mask_channel ? 1 : 0);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int xx = (int)(e % OW), yy = (int)((e / OW) % OH), b = (int)(e / ((long long)OW * OH));
    const int xmin = bh[xx * 2], xn = bh[xx * 2 + 1], ymin = bv[yy * 2], yn = bv[yy * 2 + 1];
    int acc[3] = {1 << 21, 1 << 21, 1 << 21};
    for (int y = 0; y < yn; ++y) {
      const unsigned char* row = img + ((size_t)b * Hin + (ymin + y)) * Win * 3;
      int h[3] = {1 << 21, 1 << 21, 1 << 21};
      for (int x = 0; x < xn; ++x) {
        const int k = kh[xx * ksh + x];
        const unsigned char* px = row + (size_t)(xmin + x) * 3;
        h[0] += px[0] * k; h[1] += px[1] * k; h[2] += px[2] * k;
      }
      const int k = kv[yy * ksv + y];
      acc[0] += clip8_fx(h[0]) * k; acc[1] += clip8_fx(h[1]) * k; acc[2] += clip8_fx(h[2]) * k;
    }
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (float)((double)clip8_fx(acc[c]) * 0.00392156862745098);          // uint8 * (1/255) in double, then float32
      const float n = __fdiv_rn(__fsub_rn(v, mean[c]), sd[c]);
      const size_t o = (((size_t)b * CT + c) * OH + yy) * OW + xx;
      if (sizeof(TO) == 4) ((float*)out)[o] = n; else ((bf16_t*)out)[o] = f2bf(n);
    }
    if (mask_channel) {
      const size_t o = (((size_t)b * CT + 3) * OH + yy) * OW + xx;
      if (sizeof(TO) == 4) ((float*)out)[o] = 1.0f; else ((bf16_t*)out)[o] = f2bf(1.0f);
    }
  }
}

inline int gridn(long long items, int cap = 16384) {
  long long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int mla_im2col_patch(const void* pix, int pix_fp32, void* rows, int B, int CT, int Himg, int Wimg, int P, int Kpad,
                                hipStream_t stream) {
  MLA_CHECK_ARG(pix && rows && CT >= 3 && Himg % P == 0 && Wimg % P == 0 && Kpad >= 3 * P * P, "mla_im2col_patch: bad args");
  const long long total = (long long)B * (Himg / P) * (Wimg / P) * Kpad;
  if (pix_fp32)
    hipLaunchKernelGGL(im2col_kernel<float>, dim3(gridn(total)), dim3(256), 0, stream, (const float*)pix, (bf16_t*)rows, B, CT, Himg, Wimg, P, Kpad);
  else
    hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(gridn(total)), dim3(256), 0, stream, (const bf16_t*)pix, (bf16_t*)rows, B, CT, Himg, Wimg, P, Kpad);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_avgpool_tokens(const void* x, void* y, int B, int gh, int gw, int C, int cs, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && gh % cs == 0 && gw % cs == 0, "mla_avgpool_tokens: bad args");
  hipLaunchKernelGGL(avgpool_tokens_kernel, dim3(gridn((long long)B * (gh / cs) * (gw / cs) * C)), dim3(256), 0, stream,
                     (const bf16_t*)x, (bf16_t*)y, B, gh, gw, C, cs);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_local_attn(const void* q, const void* kv, void* out, int B, int gh, int gw, int C, int cs, int heads,
                              float scale, hipStream_t stream) {
  MLA_CHECK_ARG(q && kv && out, "mla_local_attn: null pointer");
  MLA_CHECK_ARG(heads == 8 && C % 256 == 0 && C / 8 / 32 <= 8 && cs * cs <= 16 && gh % cs == 0 && gw % cs == 0,
                "mla_local_attn: need 8 heads, C %% 256 == 0, window <= 16");
  hipLaunchKernelGGL(local_attn_kernel, dim3(B * (gh / cs) * (gw / cs)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)kv,
                     (bf16_t*)out, B, gh, gw, C, cs, scale);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_local_attn_bwd(const void* q, const void* kv, const void* dout, void* dq, void* dkv, int B, int gh, int gw, int C,
                                  int cs, int heads, float scale, hipStream_t stream) {
  MLA_CHECK_ARG(q && kv && dout && dq && dkv, "mla_local_attn_bwd: null pointer");
  MLA_CHECK_ARG(heads == 8 && C % 256 == 0 && C / 8 / 32 <= 8 && cs * cs <= 16 && gh % cs == 0 && gw % cs == 0,
                "mla_local_attn_bwd: need 8 heads, C %% 256 == 0, window <= 16");
  hipLaunchKernelGGL(local_attn_bwd_kernel, dim3(B * (gh / cs) * (gw / cs)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)kv,
                     (const bf16_t*)dout, (bf16_t*)dq, (bf16_t*)dkv, B, gh, gw, C, cs, scale);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_avgpool_tokens_bwd(const void* dy, const void* other, void* dx, int B, int gh, int gw, int C, int cs, hipStream_t stream) {
  MLA_CHECK_ARG(dy && dx && gh % cs == 0 && gw % cs == 0, "mla_avgpool_tokens_bwd: bad args");
  hipLaunchKernelGGL(avgpool_tokens_bwd_kernel, dim3(gridn((long long)B * gh * gw * C)), dim3(256), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)other, (bf16_t*)dx, B, gh, gw, C, cs);
  MLA_LAUNCH_CHECK();
}

// img: uint8 [B, Hin, Win, 3]; bounds_* int32 [O, 2] = (first tap, tap count), coef_* int32 [O, ks] fixed-point (2^22) taps of the
// horizontal / vertical pass; out [B, 3 (+1 mask), OH, OW] float32 or bf16
extern "C" int mla_clip_preprocess(const unsigned char* img, int B, int Hin, int Win, const int* bounds_h, const int* coef_h, int ks_h,
                                   const int* bounds_v, const int* coef_v, int ks_v, void* out, int out_fp32, int OH, int OW,
                                   const float* mean3, const float* std3, int mask_channel, hipStream_t stream) {
  MLA_CHECK_ARG(img && bounds_h && coef_h && bounds_v && coef_v && out && mean3 && std3 && B > 0, "mla_clip_preprocess: bad args");
  const long long total = (long long)B * OH * OW;
  if (out_fp32)
    hipLaunchKernelGGL(clip_preprocess_kernel<float>, dim3(gridn(total)), dim3(256), 0, stream, img, bounds_h, coef_h, bounds_v, coef_v,
                       (float*)out, B, Hin, Win, OH, OW, ks_h, ks_v, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], mask_channel);
  else
    hipLaunchKernelGGL(clip_preprocess_kernel<bf16_t>, dim3(gridn(total)), dim3(256), 0, stream, img, bounds_h, coef_h, bounds_v, coef_v,
                       (bf16_t*)out, B, Hin, Win, OH, OW, ks_h, ks_v, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], mask_channel);
  MLA_LAUNCH_CHECK();
}
