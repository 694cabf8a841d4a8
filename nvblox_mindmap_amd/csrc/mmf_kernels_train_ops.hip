// mmf_kernels_train_ops.hip -- element-wise / normalisation passes of the policy's TRAINING step, forward and backward.
//
// LayerNorm(a + b) over rows of D <= 128 channels (the diffusion head's post-norm blocks: mindmap/diffuser_actor/layers.py, D = 120):
// torch's native_layer_norm spends 67 us forward + 53 us backward on a [19 712, 120] input (a block per 120-float row); here half a
// wave takes a row (32 lanes x float4), the residual add rides in the forward pass, and the backward pass produces dx and the
// per-workgroup column partials of dgamma / dbeta in one sweep (a second small kernel adds the partials in a fixed order:
// deterministic).  float32 throughout; mean / variance by the two-pass formula on registers.
#include <hip/hip_runtime.h>

#include "mmf_launch.h"

namespace mmf {
namespace {

constexpr int kLnMaxWgs = 256;  // one per CU: the sweep is short, the partials' sum is what must stay small

__device__ __forceinline__ float half_wave_sum(float x) {  // over the 32 lanes of a half wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 32);
  return x;
}

// y = LN(a + b) gamma + beta; writes s = a + b (when b), mean and rstd per row for the backward pass.
__global__ __launch_bounds__(256) void k_ln_train_fwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, long long rows, int D, float* __restrict__ s_out,
                                                     float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int l = threadIdx.x & 31, c = 4 * l;
  if (row >= rows) return;
  const bool on = c < D;
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (on) {
    v = *reinterpret_cast<const float4*>(a + row * D + c);
    if (b) {
      const float4 w = *reinterpret_cast<const float4*>(b + row * D + c);
      v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
      *reinterpret_cast<float4*>(s_out + row * D + c) = v;
    }
  }
  const float inv_d = 1.0f / (float)D;
  const float mean = half_wave_sum((v.x + v.y) + (v.z + v.w)) * inv_d;
  float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (on) d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  const float var = half_wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * inv_d;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (on) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = d.x * rstd * g.x + be.x;
    o.y = d.y * rstd * g.y + be.y;
    o.z = d.z * rstd * g.z + be.z;
    o.w = d.w * rstd * g.w + be.w;
    *reinterpret_cast<float4*>(y + row * D + c) = o;
  }
  if (l == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// dx = (g gamma - mean_D(g gamma) - xhat mean_D(g gamma xhat)) rstd; column partials of dgamma = sum g xhat, dbeta = sum g.
__global__ __launch_bounds__(256) void k_ln_train_bwd(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, long long rows, int D,
                                                     float* __restrict__ dx, float* __restrict__ partials) {
  __shared__ float s_part[8][2][128];
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31, c = 4 * l;
  const bool on = c < D;
  const float inv_d = 1.0f / (float)D;
  float4 gm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (on) gm = *reinterpret_cast<const float4*>(gamma + c);
  float4 ag = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ab = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (long long row = (long long)blockIdx.x * 8 + hw; row < rows; row += (long long)gridDim.x * 8) {
    float4 gv = make_float4(0.0f, 0.0f, 0.0f, 0.0f), xv = gv;
    const float mu = mean[row], rs = rstd[row];
    if (on) {
      gv = *reinterpret_cast<const float4*>(g + row * D + c);
      xv = *reinterpret_cast<const float4*>(x + row * D + c);
      xv = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
    }
    const float4 gg = make_float4(gv.x * gm.x, gv.y * gm.y, gv.z * gm.z, gv.w * gm.w);
    const float c1 = half_wave_sum((gg.x + gg.y) + (gg.z + gg.w)) * inv_d;
    const float c2 = half_wave_sum((gg.x * xv.x + gg.y * xv.y) + (gg.z * xv.z + gg.w * xv.w)) * inv_d;
    if (on) {
      float4 o;
      o.x = (gg.x - c1 - xv.x * c2) * rs;
      o.y = (gg.y - c1 - xv.y * c2) * rs;
      o.z = (gg.z - c1 - xv.z * c2) * rs;
      o.w = (gg.w - c1 - xv.w * c2) * rs;
      *reinterpret_cast<float4*>(dx + row * D + c) = o;
    }
    ag.x += gv.x * xv.x, ag.y += gv.y * xv.y, ag.z += gv.z * xv.z, ag.w += gv.w * xv.w;
    ab.x += gv.x, ab.y += gv.y, ab.z += gv.z, ab.w += gv.w;
  }
  *reinterpret_cast<float4*>(&s_part[hw][0][c]) = ag;
  *reinterpret_cast<float4*>(&s_part[hw][1][c]) = ab;
  __syncthreads();
  {  // 256 threads = 2 x 128 columns: add the eight half waves in a fixed order
    const int which = threadIdx.x >> 7, col = threadIdx.x & 127;
    float t = 0.0f;
#pragma unroll
    for (int h = 0; h < 8; ++h) t += s_part[h][which][col];
    partials[((long long)blockIdx.x * 2 + which) * 128 + col] = t;
  }
}

// 1024 threads = 4 parts x (2 x 128 columns): a part adds its quarter of the workgroups' partials (independent loads, unrolled),
// the four parts are added in a fixed order through LDS: deterministic, a few dependent rounds instead of n_wgs.
__global__ __launch_bounds__(1024) void k_ln_train_bwd_reduce(const float* __restrict__ partials, int n_wgs, int D, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
  __shared__ float s_sum[4][256];
  const int part = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int per = (n_wgs + 3) >> 2, w0 = part * per, w1 = min(n_wgs, w0 + per);
  float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int w = w0;
  for (; w + 8 <= w1; w += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += partials[(long long)(w + u) * 256 + t];
  }
  for (; w < w1; ++w) acc[0] += partials[(long long)w * 256 + t];
  s_sum[part][t] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (part == 0) {
    const int which = t >> 7, col = t & 127;
    if (col < D) (which ? dbeta : dgamma)[col] = (s_sum[0][t] + s_sum[1][t]) + (s_sum[2][t] + s_sum[3][t]);
  }
}

// ---- AdaLN modulation, backward: y = x (1 + scale_b) + shift_b with (scale | shift) [B, 2 D] broadcast over the L rows of a batch element.
// dx = g (1 + scale); dscale_b = sum_L g x; dshift_b = sum_L g -- one sweep, column partials per (batch element, row chunk), added in
// a fixed order by the second kernel.  (The forward pass is mmf_adaln_modulate.)
constexpr int kAdaChunks = 8;

__global__ __launch_bounds__(256) void k_adaln_train_bwd(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ ss, int L,
                                                        int D, float* __restrict__ dx, float* __restrict__ partials) {
  __shared__ float s_part[8][2][128];
  const int b = blockIdx.x / kAdaChunks, ch = blockIdx.x % kAdaChunks;
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31, c = 4 * l;
  const bool on = c < D;
  float4 sc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (on) sc = *reinterpret_cast<const float4*>(ss + (long long)b * 2 * D + c);
  sc = make_float4(1.0f + sc.x, 1.0f + sc.y, 1.0f + sc.z, 1.0f + sc.w);
  float4 as = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ah = as;
  if (on)
    for (int row = ch * 8 + hw; row < L; row += kAdaChunks * 8) {
      const long long o = ((long long)b * L + row) * D + c;
      const float4 gv = *reinterpret_cast<const float4*>(g + o), xv = *reinterpret_cast<const float4*>(x + o);
      *reinterpret_cast<float4*>(dx + o) = make_float4(gv.x * sc.x, gv.y * sc.y, gv.z * sc.z, gv.w * sc.w);
      as.x += gv.x * xv.x, as.y += gv.y * xv.y, as.z += gv.z * xv.z, as.w += gv.w * xv.w;
      ah.x += gv.x, ah.y += gv.y, ah.z += gv.z, ah.w += gv.w;
    }
  *reinterpret_cast<float4*>(&s_part[hw][0][c]) = as;
  *reinterpret_cast<float4*>(&s_part[hw][1][c]) = ah;
  __syncthreads();
  const int which = threadIdx.x >> 7, col = threadIdx.x & 127;
  float t = 0.0f;
#pragma unroll
  for (int h = 0; h < 8; ++h) t += s_part[h][which][col];
  partials[((long long)blockIdx.x * 2 + which) * 128 + col] = t;
}

__global__ __launch_bounds__(256) void k_adaln_train_bwd_reduce(const float* __restrict__ partials, int D, float* __restrict__ dss) {
  const int b = blockIdx.x, which = threadIdx.x >> 7, col = threadIdx.x & 127;
  if (col >= D) return;
  float t = 0.0f;
#pragma unroll
  for (int ch = 0; ch < kAdaChunks; ++ch) t += partials[(((long long)b * kAdaChunks + ch) * 2 + which) * 128 + col];
  dss[(long long)b * 2 * D + which * D + col] = t;
}

// ---- Linear layer, backward with respect to its parameters: dW [N, K] = g^T x, db [N] = column sums of g, for g [R, N], x [R, K] with
// R in the tens of thousands and N, K <= 256 (the trainable stacks: 120 -> 120 / 240 over 19 712 or 98 304 token rows).  The BLAS
// libraries' best solution for these "short and very deep" products runs at 20 - 36 TFLOP/s (28 us / 157 us); here the rows are split
// over the chip, every split multiplies on the f32 matrix cores (M = 16 outputs, N = 16 inputs, K = 4 rows per instruction, operands
// straight from global memory: a lane's A value is g[row][o], its B value x[row][i]) and writes its partial [N K + N] once; a second
// kernel adds the splits in a fixed order (deterministic).  The bias gradient costs nothing extra: the A operands are the g values.
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kWgSplitsMax = 128;

template <int NIT>  // 16-wide input tiles (ceil(K / 16))
__global__ __launch_bounds__(512) void k_linear_wgrad(const float* __restrict__ g, const float* __restrict__ x, long long R, int N, int K,
                                                     int rows_per_split, float* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, rr = lane >> 4;
  const int otile = blockIdx.y * 8 + wave;
  if (otile * 16 >= N) return;
  const int o = otile * 16 + col;
  const bool o_ok = o < N;
  const long long r_begin = (long long)blockIdx.x * rows_per_split;
  const long long r_end = r_begin + rows_per_split < R ? r_begin + rows_per_split : R;
  f4v acc[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) acc[j] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
  float bsum = 0.0f;
  for (long long r = r_begin; r < r_end; r += 16) {  // 16 rows = four matrix instructions per tile: all loads first
    float a[4], bv[4][NIT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row = r + 4 * u + rr;
      const bool ok = row < r_end;
      a[u] = (ok && o_ok) ? g[row * N + o] : 0.0f;
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int i = 16 * j + col;
        bv[u][j] = (ok && i < K) ? x[row * K + i] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bsum += a[u];
#pragma unroll
      for (int j = 0; j < NIT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u][j], acc[j], 0, 0, 0);
    }
  }
  float* part = partials + (long long)blockIdx.x * ((long long)N * K + N);
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int i = 16 * j + col;
    if (i < K) {
      const float v[4] = {acc[j].x, acc[j].y, acc[j].z, acc[j].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int oo = otile * 16 + 4 * rr + t;
        if (oo < N) part[(long long)oo * K + i] = v[t];
      }
    }
  }
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (rr == 0 && o_ok) part[(long long)N * K + o] = bsum;
}

// dW | db = sum over the splits: 1024 threads = 4 parts x 256 elements, the parts added in a fixed order through LDS
__global__ __launch_bounds__(1024) void k_linear_wgrad_reduce(const float* __restrict__ partials, int n_splits, long long E, long long NK,
                                                             float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float s_sum[4][256];
  const int part = threadIdx.x >> 8, t = threadIdx.x & 255;
  const long long e = (long long)blockIdx.x * 256 + t;
  const int per = (n_splits + 3) >> 2, w0 = part * per, w1 = min(n_splits, w0 + per);
  float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (e < E) {
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += partials[(long long)(w + u) * E + e];
    }
    for (; w < w1; ++w) acc[0] += partials[(long long)w * E + e];
  }
  s_sum[part][t] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (part == 0 && e < E) {
    const float v = (s_sum[0][t] + s_sum[1][t]) + (s_sum[2][t] + s_sum[3][t]);
    if (e < NK)
      dW[e] = v;
    else if (db)
      db[e - NK] = v;
  }
}

}  // namespace

size_t ln_train_partials_bytes() { return sizeof(float) * (size_t)kLnMaxWgs * 2 * 128; }

// 0 = launched, 1 = unsupported shape
int launch_ln_train_fwd(const float* a, const float* b, const float* gamma, const float* beta, float eps, long long rows, int D, float* s_out,
                        float* y, float* mean, float* rstd, hipStream_t s) {
  if (D <= 0 || D > 128 || (D & 3) || rows <= 0 || (b && !s_out)) return 1;
  hipLaunchKernelGGL(k_ln_train_fwd, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, s, a, b, gamma, beta, eps, rows, D, s_out, y, mean, rstd);
  return 0;
}

int launch_ln_train_bwd(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, long long rows, int D, float* dx,
                        float* dgamma, float* dbeta, float* partials, hipStream_t s) {
  if (D <= 0 || D > 128 || (D & 3) || rows <= 0) return 1;
  long long want = (rows + 31) / 32;  // >= 4 rows per half wave
  const int n_wgs = (int)(want < 1 ? 1 : (want > kLnMaxWgs ? kLnMaxWgs : want));
  hipLaunchKernelGGL(k_ln_train_bwd, dim3(n_wgs), dim3(256), 0, s, g, x, gamma, mean, rstd, rows, D, dx, partials);
  hipLaunchKernelGGL(k_ln_train_bwd_reduce, dim3(1), dim3(1024), 0, s, partials, n_wgs, D, dgamma, dbeta);
  return 0;
}

}  // namespace mmf

namespace mmf {
size_t adaln_train_scratch_bytes(int B) { return sizeof(float) * (size_t)B * kAdaChunks * 2 * 128; }

int launch_adaln_train_bwd(const float* g, const float* x, const float* ss, int B, int L, int D, float* dx, float* dss, float* partials,
                           hipStream_t s) {
  if (D <= 0 || D > 128 || (D & 3) || B <= 0 || L <= 0) return 1;
  hipLaunchKernelGGL(k_adaln_train_bwd, dim3((unsigned)(B * kAdaChunks)), dim3(256), 0, s, g, x, ss, L, D, dx, partials);
  hipLaunchKernelGGL(k_adaln_train_bwd_reduce, dim3((unsigned)B), dim3(256), 0, s, partials, D, dss);
  return 0;
}
}  // namespace mmf

namespace mmf {
static int wgrad_splits(long long R, int N) {
  const int gy = (N + 127) / 128;
  long long s = 256 / gy;
  if (s > kWgSplitsMax) s = kWgSplitsMax;
  const long long by_rows = (R + 63) / 64;  // at least 64 rows per split
  if (s > by_rows) s = by_rows;
  return (int)(s < 1 ? 1 : s);
}
size_t linear_wgrad_scratch_bytes(long long R, int N, int K) { return sizeof(float) * (size_t)wgrad_splits(R, N) * ((size_t)N * K + N); }

int launch_linear_wgrad(const float* g, const float* x, long long R, int N, int K, float* dW, float* db, float* partials, hipStream_t s) {
  if (R <= 0 || N <= 0 || K <= 0 || N > 256 || K > 128) return 1;
  const int S = wgrad_splits(R, N);
  int rps = (int)((R + S - 1) / S);
  rps = (rps + 15) & ~15;
  const dim3 grid((unsigned)S, (unsigned)((N + 127) / 128));
  const int nit = (K + 15) / 16;
  switch (nit) {
    case 1: hipLaunchKernelGGL(k_linear_wgrad<1>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    case 2: hipLaunchKernelGGL(k_linear_wgrad<2>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    case 3: hipLaunchKernelGGL(k_linear_wgrad<3>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    case 4: hipLaunchKernelGGL(k_linear_wgrad<4>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    case 5: hipLaunchKernelGGL(k_linear_wgrad<5>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    case 6: hipLaunchKernelGGL(k_linear_wgrad<6>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    case 7: hipLaunchKernelGGL(k_linear_wgrad<7>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
    default: hipLaunchKernelGGL(k_linear_wgrad<8>, grid, dim3(512), 0, s, g, x, R, N, K, rps, partials); break;
  }
  const long long NK = (long long)N * K, E = NK + N;
  hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((unsigned)((E + 255) / 256)), dim3(1024), 0, s, partials, S, E, NK, dW, db);
  return 0;
}
}  // namespace mmf
