// mmf_kernels_policy.hip -- inference-side fused ops of the diffusion head (policy serving path).  gfx950 / wave64.
//
// At batch 1 one denoising step of the head is ~700 kernels of a few microseconds on [616 x 120] tensors: launch-bound
// glue around ~60 small GEMMs.  These kernels replace the glue of mindmap/diffuser_actor/{layers,position_encodings,
// multihead_custom_attention}.py at inference:
//   k_rotary_apply    x * cos + rotate_pairs(x) * sin        (7 elementwise kernels -> 1; same float operations)
//   k_adaln_modulate  x * (1 + scale) + shift                (4 -> 1; same float operations)
//   k_attention_rows / k_attention_few   softmax(q k^T / sqrt(d) + padding mask) v per head, fp32, online softmax
//                     (the SDPA math fallback: ~10 kernels -> 1; agrees with it to float rounding, not bit for bit)
#include "mmf_launch.h"

namespace mmf {

// x, out: [rows, D] with row strides (x may be a column slice of a wider matrix); cos, sin: [rows, D] contiguous.
__global__ __launch_bounds__(256) void k_rotary_apply(const float* __restrict__ x, long long x_stride, const float* __restrict__ cs,
                                                     const float* __restrict__ sn, float* __restrict__ out, long long rows, int D) {
  const int half = D >> 1;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * half) return;
  const long long r = i / half;
  const int k = (int)(i - r * half) * 2;
  const float x0 = x[r * x_stride + k], x1 = x[r * x_stride + k + 1];
  const float c0 = cs[r * D + k], c1 = cs[r * D + k + 1], s0 = sn[r * D + k], s1 = sn[r * D + k + 1];
  // apply_rotary: x * cos + x_rot * sin with x_rot = (-x1, x0) per pair
  out[r * D + k] = x0 * c0 + (-x1) * s0;
  out[r * D + k + 1] = x1 * c1 + x0 * s1;
}

// The gradient of k_rotary_apply with respect to x (training): dx_{2k} = g_{2k} cos_{2k} + g_{2k+1} sin_{2k+1},
// dx_{2k+1} = g_{2k+1} cos_{2k+1} + (-g_{2k}) sin_{2k} -- the floats autograd computes for x * cos + rotate_pairs(x) * sin.
__global__ __launch_bounds__(256) void k_rotary_apply_grad(const float* __restrict__ g, const float* __restrict__ cs, const float* __restrict__ sn,
                                                          float* __restrict__ dx, long long n_pairs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs) return;
  const float2 gg = reinterpret_cast<const float2*>(g)[i], c = reinterpret_cast<const float2*>(cs)[i], sv = reinterpret_cast<const float2*>(sn)[i];
  float2 o;
  o.x = gg.x * c.x + gg.y * sv.y;
  o.y = gg.y * c.y + (-gg.x) * sv.x;
  reinterpret_cast<float2*>(dx)[i] = o;
}

// x, out: [B, L, D]; ss: [B, 2D] = (scale | shift)
__global__ __launch_bounds__(256) void k_adaln_modulate(const float* __restrict__ x, const float* __restrict__ ss,
                                                       float* __restrict__ out, int L, int D, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  const long long b = i / ((long long)L * D);
  const float scale = ss[b * 2 * D + d], shift = ss[b * 2 * D + D + d];
  out[i] = x[i] * (1.0f + scale) + shift;
}

// One reverse-diffusion update of the trajectory for both schedulers at once (channels [0, split) use coefficient set A,
// [split, C) set B): x0 = (x - s1 * eps) * inv_s2 [clamped]; prev = c0 * x0 + c1 * x [+ sigma * noise] -- the arithmetic of
// DDPMScheduler.step (diffuser_actor/scheduler.py) in one launch instead of ~20.  coef = {s1, inv_s2, c0, c1, sigma, clip}.
struct DdpmCoef {
  float s1, inv_s2, c0, c1, sigma, clip;
};
__global__ __launch_bounds__(256) void k_ddpm_step(const float* __restrict__ x, const float* __restrict__ eps, long long eps_stride,
                                                  const float* __restrict__ noise, float* __restrict__ out, long long rows, int C,
                                                  int split, DdpmCoef A, DdpmCoef B) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const long long r = i / C;
  const int c = (int)(i - r * C);
  const DdpmCoef K = c < split ? A : B;
  const float xv = x[i];
  float x0 = (xv - K.s1 * eps[r * eps_stride + c]) * K.inv_s2;
  if (K.clip > 0.0f) x0 = fminf(fmaxf(x0, -K.clip), K.clip);
  float prev = K.c0 * x0 + K.c1 * xv;
  if (K.sigma > 0.0f) prev = prev + K.sigma * noise[i];
  out[i] = prev;
}

void launch_ddpm_step(const float* x, const float* eps, long long eps_stride, const float* noise, float* out, long long rows, int C,
                      int split, const float* coefA, const float* coefB, hipStream_t s) {
  DdpmCoef A{coefA[0], coefA[1], coefA[2], coefA[3], coefA[4], coefA[5]}, B{coefB[0], coefB[1], coefB[2], coefB[3], coefB[4], coefB[5]};
  const long long n = rows * C;
  if (n > 0) hipLaunchKernelGGL(k_ddpm_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, eps, eps_stride, noise, out, rows, C, split, A, B);
}

// ---- small attention -----------------------------------------------------------------------------------------------------
// Head dim DH is a compile-time constant (the policy uses 120 / 8 = 15).  Two shapes:
//  * k_attention_rows<DH>: many query rows.  Workgroup = 8 query rows x 32 key lanes; keys / values stream through LDS in
//    tiles of 320; lane j of a row scores keys j, j+32, ... of the tile, the row maximum / sum are combined with 32-lane
//    shuffles once per tile (one rescale per tile, not per key), each lane keeps a partial output that is summed over the
//    32 lanes at the end.
//  * k_attention_few<DH>: a handful of query rows over thousands of keys (cross-attention of the trajectory tokens).
//    Workgroup = one query row, 256 threads stride over the keys straight from global memory (no reuse to stage for),
//    then a workgroup reduction of (max, sum, partial output).
constexpr int kAttTile = 320;
constexpr int kAttRows = 8;   // query rows per workgroup of k_attention_rows (each workgroup stages its head's K/V once)

template <int DH>
__global__ __launch_bounds__(256) void k_attention_rows(const float* __restrict__ q, const float* __restrict__ k, long long k_stride,
                                                       const float* __restrict__ v, long long v_stride,
                                                       const uint8_t* __restrict__ pad, float* __restrict__ out, int Lq, int Lk,
                                                       int H, float scale) {
  // kAttRows query rows x (256 / kAttRows) key lanes per workgroup; lane j of a row scores keys j, j + KL, ... of the tile.
  // Every workgroup stages its head's K and V: more rows per workgroup = less L2 -> LDS traffic.  The tile is large
  // (kAttTile keys, ~40 KB of LDS for K and V): a tile costs one global-load latency + two barriers, so a 616-key head is
  // two such rounds (64-key tiles made it ten, and the kernel was a chain of load latencies).
  // odd row stride: the 32 key lanes of a query row read consecutive tile rows at the same channel -> distinct banks
  constexpr int RS = DH | 1, QR = kAttRows, KL = 256 / kAttRows, KPL = kAttTile / KL;
  __shared__ float sK[kAttTile][RS];
  __shared__ float sV[kAttTile][RS];
  __shared__ uint8_t sP[kAttTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = threadIdx.x / KL, lane = threadIdx.x % KL;
  const int row = blockIdx.x * QR + qi;
  const int D = H * DH;
  const bool live = row < Lq;
  float qv[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) qv[c] = live ? q[((size_t)b * Lq + row) * D + h * DH + c] * scale : 0.0f;
  float m = -INFINITY, ssum = 0.0f;
  float acc[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = 0.0f;
  const size_t kb = (size_t)b * Lk;
  for (int k0 = 0; k0 < Lk; k0 += kAttTile) {
    const int nk = Lk - k0 < kAttTile ? Lk - k0 : kAttTile;
    __syncthreads();
    for (int e = threadIdx.x; e < nk * DH; e += 256) {
      const int kk = e / DH, c = e - kk * DH;
      sK[kk][c] = k[(kb + k0 + kk) * k_stride + h * DH + c];
      sV[kk][c] = v[(kb + k0 + kk) * v_stride + h * DH + c];
    }
    for (int e = threadIdx.x; e < kAttTile; e += 256) sP[e] = (e < nk) ? (pad ? pad[kb + k0 + e] : 0) : 1;
    __syncthreads();
    float sc[KPL];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int kk = lane + KL * j;
      float dot = -INFINITY;
      if (!sP[kk]) {
        dot = 0.0f;
#pragma unroll
        for (int c = 0; c < DH; ++c) dot += qv[c] * sK[kk][c];
      }
      sc[j] = dot;
      tmax = fmaxf(tmax, dot);
    }
#pragma unroll
    for (int off = KL / 2; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off, KL));
    const float mn = fmaxf(m, tmax);
    if (mn == -INFINITY) continue;  // every key so far is padding (row-uniform: all key lanes of the row agree)
    const float corr = __expf(m - mn);  // m = -inf: 0
    ssum *= corr;
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] *= corr;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      if (sc[j] == -INFINITY) continue;
      const int kk = lane + KL * j;
      const float p = __expf(sc[j] - mn);
      ssum += p;
#pragma unroll
      for (int c = 0; c < DH; ++c) acc[c] += p * sV[kk][c];
    }
    m = mn;
  }
#pragma unroll
  for (int off = KL / 2; off > 0; off >>= 1) {
    ssum += __shfl_xor(ssum, off, KL);
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] += __shfl_xor(acc[c], off, KL);
  }
  if (live && lane == 0) {
    float* o = out + ((size_t)b * Lq + row) * D + h * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = acc[c] / ssum;
  }
}

template <int DH>
__global__ __launch_bounds__(256) void k_attention_few(const float* __restrict__ q, const float* __restrict__ k, long long k_stride,
                                                      const float* __restrict__ v, long long v_stride,
                                                      const uint8_t* __restrict__ pad, float* __restrict__ out, int Lq, int Lk, int H,
                                                      float scale) {
  __shared__ float sRed[4][DH + 2];
  const int b = blockIdx.z, h = blockIdx.y, row = blockIdx.x;
  const int D = H * DH;
  float qv[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) qv[c] = q[((size_t)b * Lq + row) * D + h * DH + c] * scale;
  float m = -INFINITY, ssum = 0.0f;
  float acc[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = 0.0f;
  const size_t kb = (size_t)b * Lk;
  for (int kk = threadIdx.x; kk < Lk; kk += 256) {
    if (pad && pad[kb + kk]) continue;
    const float* kp = k + (kb + kk) * k_stride + h * DH;
    const float* vp = v + (kb + kk) * v_stride + h * DH;
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < DH; ++c) dot += qv[c] * kp[c];
    const float mn = fmaxf(m, dot);
    const float corr = __expf(m - mn), p = __expf(dot - mn);
    ssum = ssum * corr + p;
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = acc[c] * corr + p * vp[c];
    m = mn;
  }
  // wave reduction, then the four waves through LDS
  float mg = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mg = fmaxf(mg, __shfl_xor(mg, off, 64));
  __shared__ float sMax[4];
  if ((threadIdx.x & 63) == 0) sMax[threadIdx.x >> 6] = mg;
  __syncthreads();
  mg = fmaxf(fmaxf(sMax[0], sMax[1]), fmaxf(sMax[2], sMax[3]));
  const float w = (m == -INFINITY) ? 0.0f : __expf(m - mg);
  ssum *= w;
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] *= w;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ssum += __shfl_xor(ssum, off, 64);
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < DH; ++c) sRed[wv][c] = acc[c];
    sRed[wv][DH] = ssum;
  }
  __syncthreads();
  if (threadIdx.x < DH) {
    const float tot = sRed[0][DH] + sRed[1][DH] + sRed[2][DH] + sRed[3][DH];
    const float a = sRed[0][threadIdx.x] + sRed[1][threadIdx.x] + sRed[2][threadIdx.x] + sRed[3][threadIdx.x];
    out[((size_t)b * Lq + row) * D + h * DH + threadIdx.x] = a / tot;
  }
}

template <int DH>
static void attention_dispatch(const float* q, const float* k, long long k_stride, const float* v, long long v_stride, const uint8_t* pad,
                               float* out, int B, int Lq, int Lk, int H, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)DH);
  if (Lq >= 8)
    hipLaunchKernelGGL(k_attention_rows<DH>, dim3((Lq + kAttRows - 1) / kAttRows, H, B), dim3(256), 0, s, q, k, k_stride, v, v_stride, pad, out, Lq, Lk, H,
                       scale);
  else
    hipLaunchKernelGGL(k_attention_few<DH>, dim3(Lq, H, B), dim3(256), 0, s, q, k, k_stride, v, v_stride, pad, out, Lq, Lk, H, scale);
}

// ---- whole-block kernels for D-channel token streams (D = embedding dim, 120 in the policy) -------------------------------
// One thread per output channel keeps its weight row(s) in registers (fetched from the TRANSPOSED matrix [in, out], so the
// fetch is coalesced across threads); a workgroup walks a contiguous range of tokens, the token's activation vector is
// exchanged through LDS (broadcast reads).  [616 x 120] x [120 x 120] is 18 MFLOP: far too
// small for a library GEMM launch (6-8 us each) per Linear; here a block of the network is one launch.
template <int D>
struct TokLds {
  float a[D + 8], b[2 * D + 8];
  float red[8];
};

template <int D>
__device__ __forceinline__ float dot_row(const float (&w)[D], const float* __restrict__ s) {
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(s + c);
    acc += w[c] * v.x;
    acc += w[c + 1] * v.y;
    acc += w[c + 2] * v.z;
    acc += w[c + 3] * v.w;
  }
  return acc;
}

// sum over the D active threads (threads >= D contribute 0): wave shuffles + one LDS exchange; result to all threads
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += red[w];
  return t;
}

template <int D, int NT>
__device__ __forceinline__ float layer_norm_channel(float r, bool active, float gamma, float beta, float eps, float* red) {
  const float mean = block_sum<NT>(active ? r : 0.0f, red) / (float)D;
  const float dlt = active ? r - mean : 0.0f;
  const float var = block_sum<NT>(dlt * dlt, red) / (float)D;
  return dlt * rsqrtf(var + eps) * gamma + beta;
}

// FeedForwardBlock at inference: h = x*(1+scale)+shift; out = LayerNorm(h + fc2(relu(fc1(h))))      (hidden width = D)
template <int D>
__global__ __launch_bounds__(128) void k_ffn_block(const float* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ W1,
                                                  const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                  float* __restrict__ out, int L, long long tokens, int tok_per_wg) {
  __shared__ __attribute__((aligned(16))) TokLds<D> S;
  const int j = threadIdx.x;
  const bool act = j < D;
  float w1[D], w2[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    w1[c] = act ? W1[(size_t)c * D + j] : 0.0f;  // weights arrive transposed ([in, out]): coalesced over threads
    w2[c] = act ? W2[(size_t)c * D + j] : 0.0f;
  }
  const float bb1 = act ? b1[j] : 0.0f, bb2 = act ? b2[j] : 0.0f, gg = act ? gamma[j] : 0.0f, be = act ? beta[j] : 0.0f;
  const long long t0 = (long long)blockIdx.x * tok_per_wg;
  for (long long t = t0; t < t0 + tok_per_wg && t < tokens; ++t) {
    const long long b = t / L;
    float h = 0.0f;
    if (act) {
      h = x[t * D + j];
      if (ss) h = h * (1.0f + ss[b * 2 * D + j]) + ss[b * 2 * D + D + j];
      S.a[j] = h;
    }
    __syncthreads();
    const float u = fmaxf(dot_row<D>(w1, S.a) + bb1, 0.0f);
    if (act) S.b[j] = u;
    __syncthreads();
    const float r = h + (dot_row<D>(w2, S.b) + bb2);
    const float o = layer_norm_channel<D, 128>(r, act, gg, be, eps, S.red);
    if (act) out[t * D + j] = o;
  }
}

// Query side of an AttentionBlock: q = rotary(q_proj(x*(1+scale)+shift))            (ss / cos may be null)
template <int D>
__global__ __launch_bounds__(128) void k_q_block(const float* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ Wq,
                                                const float* __restrict__ bq, const float* __restrict__ cs, const float* __restrict__ sn,
                                                float* __restrict__ out, int L, long long tokens, int tok_per_wg) {
  __shared__ __attribute__((aligned(16))) TokLds<D> S;
  const int j = threadIdx.x;
  const bool act = j < D;
  float w[D];
#pragma unroll
  for (int c = 0; c < D; ++c) w[c] = act ? Wq[(size_t)c * D + j] : 0.0f;  // transposed weights
  const float bb = act ? bq[j] : 0.0f;
  const long long t0 = (long long)blockIdx.x * tok_per_wg;
  for (long long t = t0; t < t0 + tok_per_wg && t < tokens; ++t) {
    const long long b = t / L;
    __syncthreads();
    if (act) {
      float h = x[t * D + j];
      if (ss) h = h * (1.0f + ss[b * 2 * D + j]) + ss[b * 2 * D + D + j];
      S.a[j] = h;
    }
    __syncthreads();
    const float qv = dot_row<D>(w, S.a) + bb;
    if (!cs) {
      if (act) out[t * D + j] = qv;
      continue;
    }
    if (act) S.b[j] = qv;
    __syncthreads();
    if (act) {
      const float partner = (j & 1) ? S.b[j - 1] : -S.b[j + 1];  // x_rot = (-x1, x0) per channel pair
      out[t * D + j] = qv * cs[t * D + j] + partner * sn[t * D + j];
    }
  }
}

// Key / value side: k = rotary(kv_proj(m)[:D]), v = kv_proj(m)[D:]          (2D threads: thread j owns output channel j)
template <int D>
__global__ __launch_bounds__(256) void k_kv_block(const float* __restrict__ m, const float* __restrict__ Wkv, const float* __restrict__ bkv,
                                                 const float* __restrict__ cs, const float* __restrict__ sn, float* __restrict__ kout,
                                                 float* __restrict__ vout, long long tokens, int tok_per_wg) {
  __shared__ __attribute__((aligned(16))) TokLds<D> S;
  const int j = threadIdx.x;
  const bool act = j < 2 * D;
  float w[D];
#pragma unroll
  for (int c = 0; c < D; ++c) w[c] = act ? Wkv[(size_t)c * (2 * D) + j] : 0.0f;  // transposed weights [D, 2D]
  const float bb = act ? bkv[j] : 0.0f;
  const long long t0 = (long long)blockIdx.x * tok_per_wg;
  for (long long t = t0; t < t0 + tok_per_wg && t < tokens; ++t) {
    __syncthreads();
    if (j < D) S.a[j] = m[t * D + j];
    __syncthreads();
    const float y = dot_row<D>(w, S.a) + bb;
    if (act) S.b[j] = y;
    __syncthreads();
    if (j < D) {
      float kv = y;
      if (cs) {
        const float partner = (j & 1) ? S.b[j - 1] : -S.b[j + 1];
        kv = y * cs[t * D + j] + partner * sn[t * D + j];
      }
      kout[t * D + j] = kv;
    } else if (act) {
      vout[t * D + (j - D)] = y;
    }
  }
}

// Output side of an AttentionBlock: out = LayerNorm(res + out_proj(att))
template <int D>
__global__ __launch_bounds__(128) void k_attn_out_block(const float* __restrict__ att, const float* __restrict__ res,
                                                       const float* __restrict__ Wo, const float* __restrict__ bo,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float* __restrict__ out, long long tokens, int tok_per_wg) {
  __shared__ __attribute__((aligned(16))) TokLds<D> S;
  const int j = threadIdx.x;
  const bool act = j < D;
  float w[D];
#pragma unroll
  for (int c = 0; c < D; ++c) w[c] = act ? Wo[(size_t)c * D + j] : 0.0f;  // transposed weights
  const float bb = act ? bo[j] : 0.0f, gg = act ? gamma[j] : 0.0f, be = act ? beta[j] : 0.0f;
  const long long t0 = (long long)blockIdx.x * tok_per_wg;
  for (long long t = t0; t < t0 + tok_per_wg && t < tokens; ++t) {
    __syncthreads();
    if (act) S.a[j] = att[t * D + j];
    __syncthreads();
    const float r = (act ? res[t * D + j] : 0.0f) + (dot_row<D>(w, S.a) + bb);
    const float o = layer_norm_channel<D, 128>(r, act, gg, be, eps, S.red);
    if (act) out[t * D + j] = o;
  }
}

// Self-attention input side in one launch: q = rotary(q_proj(x*(1+scale)+shift)), k = rotary(kv_proj(x)[:D]), v = kv_proj(x)[D:].
// 3D threads: thread j < D owns query channel j (modulated input), thread D + j owns key/value channel j (raw input).
template <int D>
__global__ __launch_bounds__(384) void k_qkv_block(const float* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ Wq,
                                                  const float* __restrict__ bq, const float* __restrict__ Wkv, const float* __restrict__ bkv,
                                                  const float* __restrict__ cs, const float* __restrict__ sn, float* __restrict__ qout,
                                                  float* __restrict__ kout, float* __restrict__ vout, int L, long long tokens, int tok_per_wg) {
  __shared__ __attribute__((aligned(16))) float s_raw[D + 8], s_mod[D + 8], s_y[3 * D + 8];
  const int j = threadIdx.x;
  const bool act = j < 3 * D, is_q = j < D;
  float w[D];
#pragma unroll
  for (int c = 0; c < D; ++c) w[c] = !act ? 0.0f : (is_q ? Wq[(size_t)c * D + j] : Wkv[(size_t)c * (2 * D) + (j - D)]);
  const float bb = !act ? 0.0f : (is_q ? bq[j] : bkv[j - D]);
  const long long t0 = (long long)blockIdx.x * tok_per_wg;
  for (long long t = t0; t < t0 + tok_per_wg && t < tokens; ++t) {
    const long long b = t / L;
    __syncthreads();
    if (is_q) {
      const float xv = x[t * D + j];
      s_raw[j] = xv;
      s_mod[j] = ss ? xv * (1.0f + ss[b * 2 * D + j]) + ss[b * 2 * D + D + j] : xv;
    }
    __syncthreads();
    const float y = dot_row<D>(w, is_q ? s_mod : s_raw) + bb;
    if (act) s_y[j] = y;
    __syncthreads();
    if (j < 2 * D) {  // q (j < D) and k (D <= j < 2D): rotary; the pair partner sits next to it in s_y
      const int c = is_q ? j : j - D;
      float r = y;
      if (cs) {
        const float partner = (c & 1) ? s_y[j - 1] : -s_y[j + 1];
        r = y * cs[t * D + c] + partner * sn[t * D + c];
      }
      (is_q ? qout : kout)[t * D + c] = r;
    } else if (act) {
      vout[t * D + (j - 2 * D)] = y;
    }
  }
}

// Output side of a layer in one launch: x1 = LayerNorm(res + out_proj(att)); h = x1*(1+scale)+shift;
// out = LayerNorm(h + fc2(relu(fc1(h))))            (AttentionBlock tail + the following FeedForwardBlock)
template <int D>
__global__ __launch_bounds__(128) void k_out_ffn_block(const float* __restrict__ att, const float* __restrict__ res, const float* __restrict__ Wo,
                                                      const float* __restrict__ bo, const float* __restrict__ g1, const float* __restrict__ be1,
                                                      float eps1, const float* __restrict__ ss, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                                                      const float* __restrict__ g2, const float* __restrict__ be2, float eps2,
                                                      float* __restrict__ out, int L, long long tokens, int tok_per_wg) {
  __shared__ __attribute__((aligned(16))) TokLds<D> S;
  const int j = threadIdx.x;
  const bool act = j < D;
  float wo[D], w1[D], w2[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    wo[c] = act ? Wo[(size_t)c * D + j] : 0.0f;
    w1[c] = act ? W1[(size_t)c * D + j] : 0.0f;
    w2[c] = act ? W2[(size_t)c * D + j] : 0.0f;
  }
  const float bbo = act ? bo[j] : 0.0f, bb1 = act ? b1[j] : 0.0f, bb2 = act ? b2[j] : 0.0f;
  const float gg1 = act ? g1[j] : 0.0f, bt1 = act ? be1[j] : 0.0f, gg2 = act ? g2[j] : 0.0f, bt2 = act ? be2[j] : 0.0f;
  const long long t0 = (long long)blockIdx.x * tok_per_wg;
  for (long long t = t0; t < t0 + tok_per_wg && t < tokens; ++t) {
    const long long b = t / L;
    __syncthreads();
    if (act) S.a[j] = att[t * D + j];
    __syncthreads();
    const float r1 = (act ? res[t * D + j] : 0.0f) + (dot_row<D>(wo, S.a) + bbo);
    const float x1 = layer_norm_channel<D, 128>(r1, act, gg1, bt1, eps1, S.red);
    float h = x1;
    if (act && ss) h = x1 * (1.0f + ss[b * 2 * D + j]) + ss[b * 2 * D + D + j];
    __syncthreads();
    if (act) S.a[j] = h;
    __syncthreads();
    const float u = fmaxf(dot_row<D>(w1, S.a) + bb1, 0.0f);
    if (act) S.b[j] = u;
    __syncthreads();
    const float r2 = h + (dot_row<D>(w2, S.b) + bb2);
    const float o = layer_norm_channel<D, 128>(r2, act, gg2, bt2, eps2, S.red);
    if (act) out[t * D + j] = o;
  }
}

static inline int tok_split(long long tokens, int* tok_per_wg) {
  // enough workgroups to fill the chip, few enough that the per-workgroup weight fetch (58-115 KB) stays small
  int per = (int)((tokens + 255) / 256);
  if (per < 1) per = 1;
  *tok_per_wg = per;
  return (int)((tokens + per - 1) / per);
}

int launch_qkv_block(const float* x, const float* ss, const float* Wq, const float* bq, const float* Wkv, const float* bkv, const float* cs,
                     const float* sn, float* qout, float* kout, float* vout, int B, int L, int D, hipStream_t s) {
  if (D != 120) return 1;
  int per;
  const long long tokens = (long long)B * L;
  const int g = tok_split(tokens, &per);
  hipLaunchKernelGGL(k_qkv_block<120>, dim3(g), dim3(384), 0, s, x, ss, Wq, bq, Wkv, bkv, cs, sn, qout, kout, vout, L, tokens, per);
  return 0;
}
int launch_out_ffn_block(const float* att, const float* res, const float* Wo, const float* bo, const float* g1, const float* be1, float eps1,
                         const float* ss, const float* W1, const float* b1, const float* W2, const float* b2, const float* g2,
                         const float* be2, float eps2, float* out, int B, int L, int D, hipStream_t s) {
  if (D != 120) return 1;
  int per;
  const long long tokens = (long long)B * L;
  const int g = tok_split(tokens, &per);
  hipLaunchKernelGGL(k_out_ffn_block<120>, dim3(g), dim3(128), 0, s, att, res, Wo, bo, g1, be1, eps1, ss, W1, b1, W2, b2, g2, be2, eps2, out, L,
                     tokens, per);
  return 0;
}

int launch_ffn_block(const float* x, const float* ss, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* gamma, const float* beta, float eps, float* out, int B, int L, int D, hipStream_t s) {
  if (D != 120) return 1;
  int per;
  const long long tokens = (long long)B * L;
  const int g = tok_split(tokens, &per);
  hipLaunchKernelGGL(k_ffn_block<120>, dim3(g), dim3(128), 0, s, x, ss, W1, b1, W2, b2, gamma, beta, eps, out, L, tokens, per);
  return 0;
}
int launch_q_block(const float* x, const float* ss, const float* Wq, const float* bq, const float* cs, const float* sn, float* out,
                   int B, int L, int D, hipStream_t s) {
  if (D != 120) return 1;
  int per;
  const long long tokens = (long long)B * L;
  const int g = tok_split(tokens, &per);
  hipLaunchKernelGGL(k_q_block<120>, dim3(g), dim3(128), 0, s, x, ss, Wq, bq, cs, sn, out, L, tokens, per);
  return 0;
}
int launch_kv_block(const float* m, const float* Wkv, const float* bkv, const float* cs, const float* sn, float* kout, float* vout,
                    long long tokens, int D, hipStream_t s) {
  if (D != 120) return 1;
  int per;
  const int g = tok_split(tokens, &per);
  hipLaunchKernelGGL(k_kv_block<120>, dim3(g), dim3(256), 0, s, m, Wkv, bkv, cs, sn, kout, vout, tokens, per);
  return 0;
}
int launch_attn_out_block(const float* att, const float* res, const float* Wo, const float* bo, const float* gamma, const float* beta,
                          float eps, float* out, long long tokens, int D, hipStream_t s) {
  if (D != 120) return 1;
  int per;
  const int g = tok_split(tokens, &per);
  hipLaunchKernelGGL(k_attn_out_block<120>, dim3(g), dim3(128), 0, s, att, res, Wo, bo, gamma, beta, eps, out, tokens, per);
  return 0;
}

void launch_rotary_apply(const float* x, long long x_stride, const float* cs, const float* sn, float* out, long long rows, int D,
                         hipStream_t s) {
  const long long n = rows * (D / 2);
  if (n > 0) hipLaunchKernelGGL(k_rotary_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, x_stride, cs, sn, out, rows, D);
}

void launch_rotary_apply_grad(const float* g, const float* cs, const float* sn, float* dx, long long rows, int D, hipStream_t s) {
  const long long n = rows * (D / 2);
  if (n > 0) hipLaunchKernelGGL(k_rotary_apply_grad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, cs, sn, dx, n);
}

void launch_adaln_modulate(const float* x, const float* ss, float* out, int B, int L, int D, hipStream_t s) {
  const long long n = (long long)B * L * D;
  if (n > 0) hipLaunchKernelGGL(k_adaln_modulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ss, out, L, D, n);
}

int launch_attention_small(const float* q, const float* k, long long k_stride, const float* v, long long v_stride, const uint8_t* pad,
                           float* out, int B, int Lq, int Lk, int H, int d, hipStream_t s) {
  if (B <= 0 || Lq <= 0 || Lk <= 0 || H <= 0) return 1;
  switch (d) {
    case 8: attention_dispatch<8>(q, k, k_stride, v, v_stride, pad, out, B, Lq, Lk, H, s); return 0;
    case 15: attention_dispatch<15>(q, k, k_stride, v, v_stride, pad, out, B, Lq, Lk, H, s); return 0;
    case 16: attention_dispatch<16>(q, k, k_stride, v, v_stride, pad, out, B, Lq, Lk, H, s); return 0;
    case 20: attention_dispatch<20>(q, k, k_stride, v, v_stride, pad, out, B, Lq, Lk, H, s); return 0;
    case 24: attention_dispatch<24>(q, k, k_stride, v, v_stride, pad, out, B, Lq, Lk, H, s); return 0;
    case 32: attention_dispatch<32>(q, k, k_stride, v, v_stride, pad, out, B, Lq, Lk, H, s); return 0;
    default: return 1;
  }
}

}  // namespace mmf
