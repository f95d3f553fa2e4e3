// Point-cloud tokenizer kernels (forward only: the 3-D tower is frozen on the SFT / post-training path,
// models/vlm/prismatic.py:463-467) + camera projection of the point centres.
// Reference: models/mla/pointcloud/backbone/Point_PN.py (furthest_point_sample :6-21, knn_point :62-73, LGA :115-158,
// PosE_Geo :231-249, Linear1Layer/Linear2Layer :173-219, Pooling :166-169), models/mla/fuser/contrastive.py:5-45.
// The reference's FPS is a 512/256-iteration Python loop with a host sync per iteration; here one workgroup per cloud
// keeps the cloud and the running distances on chip for the whole selection.
#include "common.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------ projection
__global__ __launch_bounds__(256) void project_points_kernel(const float* __restrict__ xyz, const float* __restrict__ c,
                                                             long long* __restrict__ idx, unsigned char* __restrict__ valid,
                                                             int n, float W, float Hh, float stride, int ph, int pw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
  float cam[3], uvw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float a = __fmul_rn(x, c[r * 3]);
    a = fmaf(y, c[r * 3 + 1], a);
    a = fmaf(z, c[r * 3 + 2], a);
    cam[r] = __fadd_rn(a, c[9 + r]);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float a = __fmul_rn(cam[0], c[12 + r * 3]);
    a = fmaf(cam[1], c[12 + r * 3 + 1], a);
    uvw[r] = fmaf(cam[2], c[12 + r * 3 + 2], a);
  }
  const float zz = uvw[2];
  const float den = __fadd_rn(zz, 1e-6f);
  const float px = __fdiv_rn(uvw[0], den), py = __fdiv_rn(uvw[1], den);
  long long row = (long long)floorf(__fdiv_rn(py, stride));
  long long col = (long long)floorf(__fdiv_rn(px, stride));
  const bool v = (zz > 0.f) && (px >= 0.f) && (px < W) && (py >= 0.f) && (py < Hh);
  row = row < 0 ? 0 : (row > ph - 1 ? ph - 1 : row);
  col = col < 0 ? 0 : (col > pw - 1 ? pw - 1 : col);
  idx[i * 2] = row;
  idx[i * 2 + 1] = col;
  valid[i] = v ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ FPS
constexpr int FPS_MAXP = 8;  // points per thread (N <= 2048)
// Farthest-point sampling (Point_PN.py:10-21 index arithmetic; the start index is drawn by the caller): one workgroup per cloud, the whole
// loop on chip. A step is a latency chain -- new centre -> distance update -> arg-max over the cloud -> next centre -- so what counts is
// the length of that chain, not throughput (32 workgroups on 256 CUs). Round 6: the lane's points live in registers (no LDS reads in the
// update), the (distance, index) pair is ONE 64-bit key (distance bits << 32 | ~index: larger distance wins, then the LOWER index, exactly
// the old two-field comparison), reduced inside a wave on the VALU's DPP path (row_shr 1/2/4/8, row_bcast 15/31: ~10 short-latency
// moves instead of twelve dependent ds_bpermute round trips) and across the four waves through double-buffered LDS slots with ONE
// barrier per step (two before). Same comparisons on the same values: indices identical to the previous kernel and to the reference.
__device__ __forceinline__ unsigned long long fps_dpp_max(unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define FPS_STEP(CTRL, ROWMASK)                                                                                          \
  {                                                                                                                      \
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROWMASK, 0xf, true);            \
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROWMASK, 0xf, true);    \
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                                    \
    v = o > v ? o : v;                                                                                                   \
  }
  FPS_STEP(0x111, 0xf) FPS_STEP(0x112, 0xf) FPS_STEP(0x114, 0xf) FPS_STEP(0x118, 0xf)      // row_shr 1, 2, 4, 8: lane 15 of a row = its max
  FPS_STEP(0x142, 0xa)                                                                    // row_bcast:15 -> lanes 31 / 63
  FPS_STEP(0x143, 0xc)                                                                    // row_bcast:31 -> lane 63 = the wave's max
#undef FPS_STEP
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
#else
  return v;
#endif
}

__global__ __launch_bounds__(256) void fps_kernel(const float* __restrict__ xyz, const long long* __restrict__ start,
                                                  long long* __restrict__ out, int N, int npoint) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pts = (float*)smem;                                            // [N][3]: the centre of a step is fetched from here
  unsigned long long* slot = (unsigned long long*)(pts + ((N * 3 + 3) & ~3));   // [2][4]: per-wave best key, double-buffered
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* src = xyz + (size_t)b * N * 3;
  for (int e = tid; e < N * 3; e += 256) pts[e] = src[e];
  float dist[FPS_MAXP], px[FPS_MAXP], py[FPS_MAXP], pz[FPS_MAXP];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < FPS_MAXP; ++i) {
    const int p = tid + i * 256;
    dist[i] = 1e10f;
    px[i] = py[i] = pz[i] = 0.f;
    if (p < N) { px[i] = pts[p * 3]; py[i] = pts[p * 3 + 1]; pz[i] = pts[p * 3 + 2]; }
  }
  int far = (int)start[b];
  for (int it = 0; it < npoint; ++it) {
    if (tid == 0) out[(size_t)b * npoint + it] = far;
    const float cx = pts[far * 3], cy = pts[far * 3 + 1], cz = pts[far * 3 + 2];
    unsigned long long key = 0ull;               // below every real key (distances are >= 0, ~index < 2^32 - 1 only for index > 0 ... and
                                                 // a real key with distance 0 and index 0 is 0x00000000ffffffff > 0)
#pragma unroll
    for (int i = 0; i < FPS_MAXP; ++i) {
      const int p = tid + i * 256;
      if (p < N) {
        const float dx = __fsub_rn(px[i], cx), dy = __fsub_rn(py[i], cy), dz = __fsub_rn(pz[i], cz);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (d < dist[i]) dist[i] = d;
        // non-negative floats order like their bit patterns; ~p makes the LOWER index the larger key among equal distances
        const unsigned long long k = ((unsigned long long)__float_as_uint(dist[i]) << 32) | (unsigned)(~(unsigned)p);
        key = k > key ? k : key;
      }
    }
    key = fps_dpp_max(key);
    if (lane == 0) slot[(it & 1) * 4 + wid] = key;
    __syncthreads();      // one barrier per step: the other buffer is rewritten only after everybody has passed THIS barrier
    const unsigned long long* sl = slot + (it & 1) * 4;
    unsigned long long best = sl[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) best = sl[w] > best ? sl[w] : best;
    far = (int)(~(unsigned)best);
  }
}

// ------------------------------------------------------------------------------------------------ kNN
// One block per centre: distances to all N (<= 1024) points with the reference's expansion
// (-2 <c,p> + |c|^2 + |p|^2, square_distance Point_PN.py:23-42), bitonic sort of (dist, index), first k indices.
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz, const float* __restrict__ centers,
                                                  int* __restrict__ out, int N, int G, int k) {
  __shared__ float key[1024];
  __shared__ int val[1024];
  const int bg = blockIdx.x, b = bg / G, tid = threadIdx.x;
  const float* c = centers + (size_t)bg * 3;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float cs = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
  const float* pts = xyz + (size_t)b * N * 3;
  for (int p = tid; p < 1024; p += 256) {
    float d = INFINITY;
    if (p < N) {
      const float x = pts[p * 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
      float dot = __fmul_rn(cx, x);
      dot = fmaf(cy, y, dot);
      dot = fmaf(cz, z, dot);
      const float ps = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
      d = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), cs), ps);
    }
    key[p] = d;
    val[p] = p;
  }
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < 512; t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool asc = ((lo & size) == 0);
        const float a = key[lo], bb = key[hi];
        const int ia = val[lo], ib = val[hi];
        const bool gt = (a > bb) || (a == bb && ia > ib);
        if (gt == asc) { key[lo] = bb; key[hi] = a; val[lo] = ib; val[hi] = ia; }
      }
      __syncthreads();
    }
  }
  for (int j = tid; j < k; j += 256) out[(size_t)bg * k + j] = val[j];
}

// ------------------------------------------------------------------------------------------------ LGA prep
// rows[(b,g,k)][0:C] = feats[knn] ; rows[..][C:2C] = feats[centre] ; + sin/cos positional embedding of the
// centre-subtracted, per-group max-abs normalised neighbour offset (type 'scan', alpha 1000, beta 100).
__global__ __launch_bounds__(256) void lga_prep_kernel(const float* __restrict__ xyz, const bf16_t* __restrict__ feats,
                                                       const long long* __restrict__ fps_idx, const int* __restrict__ knn,
                                                       bf16_t* __restrict__ rows, float* __restrict__ lc_xyz, int N, int G,
                                                       int K, int C, float alpha, float beta) {
  __shared__ float rel[128][3];
  __shared__ int nb[128];
  __shared__ float mx[3];
  const int bg = blockIdx.x, b = bg / G, tid = threadIdx.x;
  const int ci = (int)fps_idx[bg];
  const float* pts = xyz + (size_t)b * N * 3;
  const float cx = pts[ci * 3], cy = pts[ci * 3 + 1], cz = pts[ci * 3 + 2];
  if (tid < 3) lc_xyz[(size_t)bg * 3 + tid] = pts[ci * 3 + tid];
  for (int kk = tid; kk < K; kk += 256) {
    const int j = knn[(size_t)bg * K + kk];
    nb[kk] = j;
    rel[kk][0] = pts[j * 3] - cx;
    rel[kk][1] = pts[j * 3 + 1] - cy;
    rel[kk][2] = pts[j * 3 + 2] - cz;
  }
  __syncthreads();
  if (tid < 3) {
    float m = 0.f;
    for (int kk = 0; kk < K; ++kk) m = fmaxf(m, fabsf(rel[kk][tid]));
    mx[tid] = fmaxf(m, 1e-6f);
  }
  __syncthreads();
  const int OD = 2 * C, fd = OD / 6;
  const bf16_t* fb = feats + (size_t)b * N * C;
  for (int e = tid; e < K * OD; e += 256) {
    const int kk = e / OD, ch = e % OD;
    const float f = ch < C ? bf2f(fb[(size_t)nb[kk] * C + ch]) : bf2f(fb[(size_t)ci * C + (ch - C)]);
    const int coord = ch / (2 * fd), r = ch % (2 * fd);
    const int fi = r < fd ? r : r - fd;
    const float de = powf(alpha, (float)fi / (float)fd);
    const float arg = (beta * (rel[kk][coord] / mx[coord])) / de;
    const float pe = r < fd ? sinf(arg) : cosf(arg);
    rows[((size_t)bg * K + kk) * OD + ch] = f2bf(f + pe);
  }
}

// Same result, 8 channels per lane: the sin and the cos half of a (coordinate, frequency) pair share one sincosf, the
// 1 / alpha^(i/fd) table is computed once per block, features move as 16-B loads / stores. Needs fd = 2C/6 a multiple of 8.
__global__ __launch_bounds__(256) void lga_prep_vec_kernel(const float* __restrict__ xyz, const bf16_t* __restrict__ feats,
                                                           const long long* __restrict__ fps_idx, const int* __restrict__ knn,
                                                           bf16_t* __restrict__ rows, float* __restrict__ lc_xyz, int N, int G,
                                                           int K, int C, float alpha, float beta) {
  __shared__ float rel[128][3];
  __shared__ int nb[128];
  __shared__ float mx[3];
  __shared__ float de[256];
  const int bg = blockIdx.x, b = bg / G, tid = threadIdx.x;
  const int ci = (int)fps_idx[bg];
  const float* pts = xyz + (size_t)b * N * 3;
  const float cx = pts[ci * 3], cy = pts[ci * 3 + 1], cz = pts[ci * 3 + 2];
  const int OD = 2 * C, fd = OD / 6;
  if (tid < 3) lc_xyz[(size_t)bg * 3 + tid] = pts[ci * 3 + tid];
  if (tid < fd) de[tid] = powf(alpha, (float)tid / (float)fd);
  for (int kk = tid; kk < K; kk += 256) {
    const int j = knn[(size_t)bg * K + kk];
    nb[kk] = j;
    rel[kk][0] = pts[j * 3] - cx;
    rel[kk][1] = pts[j * 3 + 1] - cy;
    rel[kk][2] = pts[j * 3 + 2] - cz;
  }
  __syncthreads();
  if (tid < 3) {
    float m = 0.f;
    for (int kk = 0; kk < K; ++kk) m = fmaxf(m, fabsf(rel[kk][tid]));
    mx[tid] = fmaxf(m, 1e-6f);
  }
  __syncthreads();
  const bf16_t* fb = feats + (size_t)b * N * C;
  const int nch = fd >> 3, per_k = 3 * nch;
  for (int e = tid; e < K * per_k; e += 256) {
    const int kk = e / per_k, q = e % per_k;
    const int coord = q / nch, f0 = (q % nch) * 8;
    const float base = beta * (rel[kk][coord] / mx[coord]);
    const int ch_s = coord * 2 * fd + f0, ch_c = ch_s + fd;          // sin half, cos half
    float fs[8], fc[8];
    unpack8(*(const u32x4_t*)(ch_s < C ? fb + (size_t)nb[kk] * C + ch_s : fb + (size_t)ci * C + (ch_s - C)), fs);
    unpack8(*(const u32x4_t*)(ch_c < C ? fb + (size_t)nb[kk] * C + ch_c : fb + (size_t)ci * C + (ch_c - C)), fc);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float arg = base / de[f0 + j];
      fs[j] += sinf(arg);
      fc[j] += cosf(arg);
    }
    bf16_t* o = rows + ((size_t)bg * K + kk) * OD;
    *(u32x4_t*)(o + ch_s) = pack8(fs);
    *(u32x4_t*)(o + ch_c) = pack8(fc);
  }
}

// backward of lga_prep w.r.t. the point features (coordinates carry no gradient: the positional embedding is a constant):
//   dfeats[b][n][c] = sum over the groups g whose kNN list holds n (at position k)  drows[(b,g,k)][c]
//                   + sum_k drows[(b,g,k)][C + c]   for the group g whose centre is n (FPS indices are distinct: at most one)
// GATHER form, one workgroup per target point (round 4): every sum runs in a fixed order (ascending g, then ascending k), so the
// gradient of the trainable point tower is bit-reproducible -- the round-1..3 kernel scattered with fp32 atomicAdd in arrival order.
// All threads scan the batch's G*K indices coalesced (L2-resident, 166 KB at G = 512, K = 81) and set bit k of group g's 128-bit
// hit mask (integer LDS atomics: order-independent); wave 0 compacts the groups with a hit, wave 1 the groups whose centre is n, in
// ascending order; then every thread owns channels and adds the listed rows. Writes every element of dfeats (no zero-initialised
// buffer needed). A true kNN list holds a point at most once per group; repeated indices are handled all the same. K <= 128.
__global__ __launch_bounds__(256) void lga_prep_bwd_kernel(const bf16_t* __restrict__ drows, const long long* __restrict__ fps_idx,
                                                           const int* __restrict__ knn, float* __restrict__ dfeats, int N, int G, int K,
                                                           int C) {
  extern __shared__ int lga_sh[];               // mask[4 G] | list[G] | clist[G] | counts[2]
  unsigned* mask = (unsigned*)lga_sh;
  int* list = lga_sh + 4 * G;
  int* clist = lga_sh + 5 * G;
  int* misc = lga_sh + 6 * G;
  const int bn = blockIdx.x, b = bn / N, n = bn % N, tid = threadIdx.x;
  const int OD = 2 * C;
  for (int i = tid; i < 4 * G; i += 256) mask[i] = 0u;
  __syncthreads();
  const int* kb = knn + (size_t)b * G * K;
  const int total = G * K;
  for (int e = tid; e < total; e += 256)
    if (kb[e] == n) {
      const int g = e / K, k = e - g * K;
      atomicOr(&mask[g * 4 + (k >> 5)], 1u << (k & 31));
    }
  __syncthreads();
  if (tid < 128) {                              // ordered compaction, one wave each: groups with a neighbour hit, groups centred on n
    const bool nb = tid < 64;
    const int lane = tid & 63;
    int cnt = 0;
    for (int g0 = 0; g0 < G; g0 += 64) {
      const int g = g0 + lane;
      bool on = false;
      if (g < G) on = nb ? ((mask[g * 4] | mask[g * 4 + 1] | mask[g * 4 + 2] | mask[g * 4 + 3]) != 0u) : ((int)fps_idx[(size_t)b * G + g] == n);
      const unsigned long long m = __ballot(on);
      if (on) (nb ? list : clist)[cnt + __popcll(m & ((1ull << lane) - 1ull))] = g;
      cnt += __popcll(m);
    }
    if (lane == 0) misc[nb ? 0 : 1] = cnt;
  }
  __syncthreads();
  const int cnt = misc[0], ccnt = misc[1];
  const bf16_t* db = drows + (size_t)b * G * K * OD;
  for (int ch = tid; ch < C; ch += 256) {
    float acc = 0.f;
    for (int i = 0; i < cnt; ++i) {
      const int g = list[i];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        unsigned m = mask[g * 4 + w];
        while (m) {                              // ascending k; the order is part of the result
          const int k = w * 32 + __ffs((int)m) - 1;
          m &= m - 1;
          acc += bf2f(db[((size_t)g * K + k) * OD + ch]);
        }
      }
    }
    for (int j = 0; j < ccnt; ++j) {
      float s = 0.f;
      for (int kk = 0; kk < K; ++kk) s += bf2f(db[((size_t)clist[j] * K + kk) * OD + C + ch]);
      acc += s;
    }
    dfeats[((size_t)b * N + n) * C + ch] = acc;
  }
}

// backward of the max over the K neighbours: the gradient goes to the first neighbour that attains the maximum
__global__ __launch_bounds__(256) void maxpool_k_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                            bf16_t* __restrict__ dx, long long groups, int K, int C) {
  const long long total = groups * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long g = e / C;
    const int c = (int)(e % C);
    const bf16_t* col = x + (size_t)g * K * C + c;
    float best = bf2f(col[0]);
    int bi = 0;
    for (int k = 1; k < K; ++k) {
      const float v = bf2f(col[(size_t)k * C]);
      if (v > best) { best = v; bi = k; }
    }
    const bf16_t gy = dy[e];
    for (int k = 0; k < K; ++k) dx[((size_t)g * K + k) * C + c] = (k == bi) ? gy : (bf16_t)0;
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm (train)
// partial[(blockIdx.y*2 + {0,1}) * C + c] = sum / sum of squares over this block's row slice
__global__ __launch_bounds__(256) void colstats_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                               long long rows, int C, int ld) {
  __shared__ float s1[4][64], s2[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + c;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float a = 0.f, q = 0.f;
  if (ch < C)
    for (long long r = r0 + rl; r < r1; r += 4) { const float v = bf2f(x[r * ld + ch]); a += v; q += v * v; }
  s1[rl][c] = a; s2[rl][c] = q;
  __syncthreads();
  if (rl == 0 && ch < C) {
    partial[((size_t)blockIdx.y * 2) * C + ch] = s1[0][c] + s1[1][c] + s1[2][c] + s1[3][c];
    partial[((size_t)blockIdx.y * 2 + 1) * C + ch] = s2[0][c] + s2[1][c] + s2[2][c] + s2[3][c];
  }
}
// same partials from 16-B loads: lane = (row lane rl, 8-channel chunk cg) with RL = 256 / (C/8) row lanes (blockDim = RL * C/8);
// the RL per-lane sums of a column are combined in a fixed order (deterministic). Needs C % 8 == 0, C <= 2048, ld % 8 == 0.
__global__ __launch_bounds__(256) void colstats_partial_vec_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                                   long long rows, int C, int ld) {
  extern __shared__ float sh[];                      // [2][RL][C]
  const int C8 = C >> 3, RL = blockDim.x / C8;
  const int cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float a[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = q[j] = 0.f;
  long long r = r0 + rl;
  for (; r + 3LL * RL < r1; r += 4LL * RL) {         // four rows in flight per lane (round 6: one was latency-bound); same order of adds
    u32x4_t w4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w4[u] = *(const u32x4_t*)(x + (r + (long long)u * RL) * ld + cg * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      unpack8(w4[u], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] += v[j]; q[j] += v[j] * v[j]; }
    }
  }
  for (; r < r1; r += RL) {
    float v[8];
    unpack8(*(const u32x4_t*)(x + r * ld + cg * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] += v[j]; q[j] += v[j] * v[j]; }
  }
  float* s1 = sh + (size_t)rl * C + cg * 8;
  float* s2 = sh + (size_t)(RL + rl) * C + cg * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = a[j]; s2[j] = q[j]; }
  __syncthreads();
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    float ta = 0.f, tq = 0.f;
    for (int l = 0; l < RL; ++l) { ta += sh[(size_t)l * C + ch]; tq += sh[(size_t)(RL + l) * C + ch]; }
    partial[((size_t)blockIdx.x * 2) * C + ch] = ta;
    partial[((size_t)blockIdx.x * 2 + 1) * C + ch] = tq;
  }
}
// 16 columns per workgroup, 16 lanes per column striding over the P partial rows (four partial rows in flight per lane), then a
// fixed-order tree over the 16 lanes: round 1 summed every column in ONE thread (57 us at P = 256 -- a third of the statistics pass)
__global__ __launch_bounds__(256) void colstats_final_kernel(const float* __restrict__ partial, float* __restrict__ mean,
                                                             float* __restrict__ var, int P, int C, long long rows) {
  __shared__ double sa[16][17], sq[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double a = 0.0, q = 0.0;
  if (ch < C)
    for (int p = pl; p < P; p += 16) {
      a += (double)partial[((size_t)p * 2) * C + ch];
      q += (double)partial[((size_t)p * 2 + 1) * C + ch];
    }
  sa[pl][cl] = a;
  sq[pl][cl] = q;
  __syncthreads();
  if (pl == 0 && ch < C) {
    double ta = 0.0, tq = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ta += sa[i][cl]; tq += sq[i][cl]; }
    const double m = ta / (double)rows;
    double v = tq / (double)rows - m * m;
    if (v < 0.0) v = 0.0;
    mean[ch] = (float)m;
    var[ch] = (float)v;  // biased (what BatchNorm normalises with)
  }
}
// y = (x - mean) * rsqrt(var + eps) * w + b (+ residual) (relu)
// Round 6: lane = (row lane, 8-channel chunk) with blockDim = RL * C/8 like colstats_partial_vec_kernel, so a lane keeps ITS eight
// columns' mean / 1/sqrt(var + eps) / w / b in registers for all its rows (the flat grid-stride form recomputed a divide and a square root
// and issued four scalar loads per ELEMENT: 185 us per launch at 2.2 TB/s); two rows in flight per lane. Same expression per element.
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const bf16_t* __restrict__ w,
                                                       const bf16_t* __restrict__ b, const bf16_t* __restrict__ res,
                                                       bf16_t* __restrict__ y, long long rows, int C, float eps, int relu) {
  const int cpr = C >> 3, RL = blockDim.x / cpr;
  const int cg = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  const int c0 = cg * 8;
  float mu[8], rs[8], wv[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = mean[c0 + j];
    rs[j] = 1.0f / sqrtf(var[c0 + j] + eps);
    wv[j] = bf2f(w[c0 + j]);
    bv[j] = bf2f(b[c0 + j]);
  }
  auto one = [&](long long r, const u32x4_t xv, const u32x4_t rv) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t wd = xv[j >> 1], rd = rv[j >> 1];
      const float v = (j & 1) ? bfhi(wd) : bflo(wd);
      const float rr = (j & 1) ? bfhi(rd) : bflo(rd);
      float t = (v - mu[j]) * rs[j] * wv[j] + bv[j] + rr;
      if (relu) t = t > 0.f ? t : 0.f;
      o[j] = t;
    }
    u32x4_t ov;
    ov[0] = pack2bf(o[0], o[1]); ov[1] = pack2bf(o[2], o[3]); ov[2] = pack2bf(o[4], o[5]); ov[3] = pack2bf(o[6], o[7]);
    *(u32x4_t*)(y + r * C + c0) = ov;
  };
  const long long step = (long long)gridDim.x * RL;
  long long r = (long long)blockIdx.x * RL + rl;
  const u32x4_t z = {0u, 0u, 0u, 0u};
  for (; r + step < rows; r += 2 * step) {
    const u32x4_t x0 = *(const u32x4_t*)(x + r * C + c0), x1 = *(const u32x4_t*)(x + (r + step) * C + c0);
    const u32x4_t r0 = res ? *(const u32x4_t*)(res + r * C + c0) : z, r1 = res ? *(const u32x4_t*)(res + (r + step) * C + c0) : z;
    one(r, x0, r0);
    one(r + step, x1, r1);
  }
  if (r < rows) one(r, *(const u32x4_t*)(x + r * C + c0), res ? *(const u32x4_t*)(res + r * C + c0) : z);
}
// generic form (C / 8 > 256): flat grid-stride over 8-channel pieces
__global__ __launch_bounds__(256) void bn_apply_flat_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ y, long long rows, int C, float eps, int relu) {
  const int cpr = C >> 3;
  const long long total = rows * cpr;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / cpr;
    const int c0 = (int)(idx % cpr) * 8;
    const u32x4_t xv = *(const u32x4_t*)(x + r * C + c0);
    u32x4_t rv = {0u, 0u, 0u, 0u};
    if (res) rv = *(const u32x4_t*)(res + r * C + c0);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t wd = xv[j >> 1], rd = rv[j >> 1];
      const float v = (j & 1) ? bfhi(wd) : bflo(wd);
      const float rr = (j & 1) ? bfhi(rd) : bflo(rd);
      const int c = c0 + j;
      float t = (v - mean[c]) * (1.0f / sqrtf(var[c] + eps)) * bf2f(w[c]) + bf2f(b[c]) + rr;
      if (relu) t = t > 0.f ? t : 0.f;
      o[j] = t;
    }
    u32x4_t ov;
    ov[0] = pack2bf(o[0], o[1]); ov[1] = pack2bf(o[2], o[3]); ov[2] = pack2bf(o[4], o[5]); ov[3] = pack2bf(o[6], o[7]);
    *(u32x4_t*)(y + r * C + c0) = ov;
  }
}
// out[g][c] = max_k x[g][k][c]
__global__ __launch_bounds__(256) void maxpool_k_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long groups,
                                                        int K, int C) {
  const long long total = groups * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long g = idx / C;
    const int c = (int)(idx % C);
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, bf2f(x[(g * K + k) * C + c]));
    out[idx] = f2bf(m);
  }
}
// same, 8 channels per lane from 16-B loads with four neighbours in flight (round 6: the scalar form moved 2 B per lane per load,
// 250 us per launch); max is exact in any order, so the result is the same bf16 value. C % 8 == 0, 16-B aligned.
__global__ __launch_bounds__(256) void maxpool_k_vec_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long groups,
                                                            int K, int C) {
  const int cpr = C >> 3;
  const long long total = groups * cpr;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long g = idx / cpr;
    const int c0 = (int)(idx % cpr) * 8;
    const bf16_t* base = x + g * K * C + c0;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    int k = 0;
    for (; k + 3 < K; k += 4) {
      u32x4_t w4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w4[u] = *(const u32x4_t*)(base + (size_t)(k + u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8(w4[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    }
    for (; k < K; ++k) {
      float v[8];
      unpack8(*(const u32x4_t*)(base + (size_t)k * C), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
    }
    *(u32x4_t*)(out + g * C + c0) = pack8(m);
  }
}
// gather rows: out[i][:] = src[idx[i]][:] (fp32, 3 wide or any width)
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ src, const long long* __restrict__ idx,
                                                              float* __restrict__ out, int B, int N, int G, int W) {
  const long long total = (long long)B * G * W;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int w = (int)(e % W);
    const long long bg = e / W;
    const int b = (int)(bg / G);
    out[e] = src[((size_t)b * N + idx[bg]) * W + w];
  }
}

inline int gridn(long long items, int cap = 8192) {
  long long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int mla_project_points(const float* xyz, const float* consts21, long long* idx, unsigned char* valid, int n,
                                  float W, float Hh, float stride, int ph, int pw, hipStream_t stream) {
  MLA_CHECK_ARG(xyz && consts21 && idx && valid && n > 0, "mla_project_points: bad args");
  hipLaunchKernelGGL(project_points_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xyz, consts21, idx, valid, n, W, Hh,
                     stride, ph, pw);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_fps(const float* xyz, const long long* start, long long* out, int B, int N, int npoint, hipStream_t stream) {
  MLA_CHECK_ARG(xyz && start && out && B > 0 && N > 0 && N <= 2048 && npoint > 0 && npoint <= N, "mla_fps: need N <= 2048");
  hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(256), ((N * 3 + 3) & ~3) * sizeof(float) + 64, stream, xyz, start, out, N, npoint);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_knn(const float* xyz, const float* centers, int* out, int B, int N, int G, int k, hipStream_t stream) {
  MLA_CHECK_ARG(xyz && centers && out && N > 0 && N <= 1024 && k > 0 && k <= N, "mla_knn: need N <= 1024, k <= N");
  hipLaunchKernelGGL(knn_kernel, dim3(B * G), dim3(256), 0, stream, xyz, centers, out, N, G, k);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_lga_prep(const float* xyz, const void* feats, const long long* fps_idx, const int* knn, void* rows,
                            float* lc_xyz, int B, int N, int G, int K, int C, float alpha, float beta, hipStream_t stream) {
  MLA_CHECK_ARG(xyz && feats && fps_idx && knn && rows && lc_xyz, "mla_lga_prep: null pointer");
  MLA_CHECK_ARG(K <= 128 && (2 * C) % 6 == 0, "mla_lga_prep: need K <= 128 and 2C divisible by 6");
  const int fd = 2 * C / 6;
  if (fd % 8 == 0 && fd <= 256 && C % 8 == 0 && (((uintptr_t)feats | (uintptr_t)rows) & 15) == 0)
    hipLaunchKernelGGL(lga_prep_vec_kernel, dim3(B * G), dim3(256), 0, stream, xyz, (const bf16_t*)feats, fps_idx, knn, (bf16_t*)rows,
                       lc_xyz, N, G, K, C, alpha, beta);
  else
    hipLaunchKernelGGL(lga_prep_kernel, dim3(B * G), dim3(256), 0, stream, xyz, (const bf16_t*)feats, fps_idx, knn, (bf16_t*)rows,
                       lc_xyz, N, G, K, C, alpha, beta);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_colstats_blocks(long long rows) {
  // 1024: the partial pass is latency-bound with one workgroup per CU (176 us at 256 workgroups, 123 at 512, 115 at 1024)
  static const long long cap = getenv("MLA_COLSTATS_BLOCKS") ? atoll(getenv("MLA_COLSTATS_BLOCKS")) : 1024;
  long long r = rows / 2048;
  return (int)(r < 1 ? 1 : (r > cap ? cap : r));
}
// workspace: mla_colstats_blocks(rows) * 2 * C floats
extern "C" int mla_colstats_bf16(const void* x, float* mean, float* var, long long rows, int C, int ld, float* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  MLA_CHECK_ARG(x && mean && var && workspace && rows > 0, "mla_colstats_bf16: bad args");
  const int P = mla_colstats_blocks(rows);
  MLA_CHECK_ARG(workspace_bytes >= (size_t)P * 2 * C * sizeof(float), "mla_colstats_bf16: workspace too small");
  const int C8 = C / 8;
  if (C % 8 == 0 && C8 <= 256 && ld % 8 == 0 && (((uintptr_t)x) & 15) == 0) {
    const int RL = 256 / C8;
    hipLaunchKernelGGL(colstats_partial_vec_kernel, dim3(P), dim3(RL * C8), (size_t)2 * RL * C * sizeof(float), stream, (const bf16_t*)x,
                       workspace, rows, C, ld);
  } else {
    hipLaunchKernelGGL(colstats_partial_kernel, dim3((C + 63) / 64, P), dim3(256), 0, stream, (const bf16_t*)x, workspace, rows, C, ld);
  }
  hipLaunchKernelGGL(colstats_final_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, workspace, mean, var, P, C, rows);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_bn_apply(const void* x, const float* mean, const float* var, const void* w, const void* b, const void* res,
                            void* y, long long rows, int C, float eps, int relu, hipStream_t stream) {
  MLA_CHECK_ARG(x && mean && var && w && b && y && C % 8 == 0, "mla_bn_apply: bad args (C %% 8)");
  const int cpr = C / 8;
  if (cpr <= 256 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) == 0) {
    const int RL = 256 / cpr;
    long long nb = (rows + RL - 1) / RL;
    if (nb > 4096) nb = 4096;          // 16 workgroups per CU's worth of row walkers; every lane keeps its columns' constants
    hipLaunchKernelGGL(bn_apply_kernel, dim3((int)nb), dim3(RL * cpr), 0, stream, (const bf16_t*)x, mean, var, (const bf16_t*)w,
                       (const bf16_t*)b, (const bf16_t*)res, (bf16_t*)y, rows, C, eps, relu);
  } else {
    hipLaunchKernelGGL(bn_apply_flat_kernel, dim3(gridn(rows * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, mean, var,
                       (const bf16_t*)w, (const bf16_t*)b, (const bf16_t*)res, (bf16_t*)y, rows, C, eps, relu);
  }
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_maxpool_k(const void* x, void* out, long long groups, int K, int C, hipStream_t stream) {
  MLA_CHECK_ARG(x && out, "mla_maxpool_k: null pointer");
  if (C % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0)
    hipLaunchKernelGGL(maxpool_k_vec_kernel, dim3(gridn(groups * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, groups, K, C);
  else
    hipLaunchKernelGGL(maxpool_k_kernel, dim3(gridn(groups * C)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, groups, K, C);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_gather_rows_f32(const float* src, const long long* idx, float* out, int B, int N, int G, int W,
                                   hipStream_t stream) {
  MLA_CHECK_ARG(src && idx && out, "mla_gather_rows_f32: null pointer");
  hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(gridn((long long)B * G * W)), dim3(256), 0, stream, src, idx, out, B, N, G, W);
  MLA_LAUNCH_CHECK();
}

// backward kernels of the point tokenizer (stage "pretrain" with use_pointcloud: Point_PN.py:115-158 LGA, :166-169 Pooling)
extern "C" int mla_lga_prep_bwd(const void* drows, const long long* fps_idx, const int* knn, float* dfeats, int B, int N, int G,
                                int K, int C, hipStream_t stream) {
  MLA_CHECK_ARG(drows && fps_idx && knn && dfeats, "mla_lga_prep_bwd: null pointer");
  MLA_CHECK_ARG(B > 0 && N > 0 && G > 0 && K > 0 && C > 0 && K <= 128 && (size_t)(6 * G + 2) * sizeof(int) <= 48 * 1024, "mla_lga_prep_bwd: bad shape (K <= 128, G <= 2047)");
  hipLaunchKernelGGL(lga_prep_bwd_kernel, dim3(B * N), dim3(256), (6 * G + 2) * sizeof(int), stream, (const bf16_t*)drows, fps_idx, knn,
                     dfeats, N, G, K, C);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_maxpool_k_bwd(const void* x, const void* dy, void* dx, long long groups, int K, int C, hipStream_t stream) {
  MLA_CHECK_ARG(x && dy && dx, "mla_maxpool_k_bwd: null pointer");
  hipLaunchKernelGGL(maxpool_k_bwd_kernel, dim3(gridn(groups * C)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx,
                     groups, K, C);
  MLA_LAUNCH_CHECK();
}
