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
//  * k_attention_rows<DH>: many query rows.  Workgroup = 16 query rows x 16 key lanes; keys / values stream through LDS in
//    tiles of 64; lane j of a row scores keys j, j+16, j+32, j+48 of the tile, the row maximum / sum are combined with
//    16-lane shuffles once per tile (one rescale per tile, not per key), each lane keeps a partial output that is summed
//    over the 16 lanes at the end.
//  * k_attention_few<DH>: a handful of query rows over thousands of keys (cross-attention of the trajectory tokens).
//    Workgroup = one query row, 256 threads stride over the keys straight from global memory (no reuse to stage for),
//    then a workgroup reduction of (max, sum, partial output).
constexpr int kAttTile = 64;

template <int DH>
__global__ __launch_bounds__(256) void k_attention_rows(const float* __restrict__ q, const float* __restrict__ k, long long k_stride,
                                                       const float* __restrict__ v, long long v_stride,
                                                       const uint8_t* __restrict__ pad, float* __restrict__ out, int Lq, int Lk,
                                                       int H, float scale) {
  __shared__ float sK[kAttTile][DH + 1];
  __shared__ float sV[kAttTile][DH + 1];
  __shared__ uint8_t sP[kAttTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = threadIdx.x >> 4, lane = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + qi;
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
    // stage: thread t -> key t/4, channels (t%4)*4 .. +3
    {
      const int kk = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 4;
      if (kk < nk) {
        const float* kp = k + (kb + k0 + kk) * k_stride + h * DH;
        const float* vp = v + (kb + k0 + kk) * v_stride + h * DH;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c0 + c < DH) {
            sK[kk][c0 + c] = kp[c0 + c];
            sV[kk][c0 + c] = vp[c0 + c];
          }
        if (DH > 16) {
#pragma unroll
          for (int c = 16 + c0; c < DH; c += 16)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c + e < DH) {
                sK[kk][c + e] = kp[c + e];
                sV[kk][c + e] = vp[c + e];
              }
        }
      }
      if (threadIdx.x < kAttTile) sP[threadIdx.x] = (threadIdx.x < nk && pad) ? pad[kb + k0 + threadIdx.x] : (threadIdx.x < nk ? 0 : 1);
    }
    __syncthreads();
    float sc[4];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = lane + 16 * j;
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
    for (int off = 8; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off, 16));
    const float mn = fmaxf(m, tmax);
    if (mn == -INFINITY) continue;  // every key so far is padding (row-uniform: all 16 lanes agree)
    const float corr = __expf(m - mn);  // m = -inf: 0
    ssum *= corr;
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] *= corr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = lane + 16 * j;
      const float p = sc[j] == -INFINITY ? 0.0f : __expf(sc[j] - mn);
      ssum += p;
#pragma unroll
      for (int c = 0; c < DH; ++c) acc[c] += p * sV[kk][c];
    }
    m = mn;
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    ssum += __shfl_xor(ssum, off, 16);
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] += __shfl_xor(acc[c], off, 16);
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
  if (Lq >= 16)
    hipLaunchKernelGGL(k_attention_rows<DH>, dim3((Lq + 15) / 16, H, B), dim3(256), 0, s, q, k, k_stride, v, v_stride, pad, out, Lq, Lk, H,
                       scale);
  else
    hipLaunchKernelGGL(k_attention_few<DH>, dim3(Lq, H, B), dim3(256), 0, s, q, k, k_stride, v, v_stride, pad, out, Lq, Lk, H, scale);
}

void launch_rotary_apply(const float* x, long long x_stride, const float* cs, const float* sn, float* out, long long rows, int D,
                         hipStream_t s) {
  const long long n = rows * (D / 2);
  if (n > 0) hipLaunchKernelGGL(k_rotary_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, x_stride, cs, sn, out, rows, D);
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
