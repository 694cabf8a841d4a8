// mmf_kernels_fps.hip -- farthest-point sampling in feature space for the policy encoder.  gfx950 / wave64.
//
// Replaces dgl.geometry.farthest_point_sampler(x[B,N,C], npoints, start_idx=0) (a dgl CUDA op without a ROCm
// build) called at mindmap/diffuser_actor/encoder.py:366-370 with B = 32, N = 3072 tokens, C = 120 channels,
// npoints = N/5 = 614.
//
// Algorithm (classic FPS): dist[i] = +inf; pick start; repeat npoints-1 times: dist[i] = min(dist[i], |x_i - x_sel|^2),
// select argmax_i dist[i] (first index on ties).  Iterations are sequential, so one batch element is one workgroup
// (1024 threads): the selected row is broadcast through LDS, every thread keeps the running distance of its points in
// registers, the argmax is a wave-shuffle + LDS reduction.  The point matrix of one batch element (1.5 MB) is
// re-read from L2 every iteration: the kernel is L2-bandwidth / latency bound, ~3 us per iteration.
#include "mmf_launch.h"

namespace mmf {

constexpr int kFpsThreads = 1024;
constexpr int kFpsMaxPerThread = 8;  // N <= 8192
constexpr int kFpsMaxC = 1024;

__global__ __launch_bounds__(kFpsThreads) void k_fps(const float* __restrict__ x, int N, int C, int npoints, int start,
                                                    long long* __restrict__ out_idx) {
  __shared__ float s_sel[kFpsMaxC];
  __shared__ float s_best_d[kFpsThreads / 64];
  __shared__ int s_best_i[kFpsThreads / 64];
  __shared__ int s_cur;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * N * C;
  long long* ob = out_idx + (size_t)b * npoints;
  float dist[kFpsMaxPerThread];
#pragma unroll
  for (int k = 0; k < kFpsMaxPerThread; ++k) dist[k] = 3.0e38f;
  int cur = start;
  if (tid == 0) ob[0] = cur;
  for (int it = 1; it < npoints; ++it) {
    for (int c = tid; c < C; c += kFpsThreads) s_sel[c] = xb[(size_t)cur * C + c];
    __syncthreads();
    float best_d = -1.0f;
    int best_i = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < kFpsMaxPerThread; ++k) {
      const int i = tid + k * kFpsThreads;
      if (i < N) {
        const float* xi = xb + (size_t)i * C;
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
          const float d = xi[c] - s_sel[c];
          acc += d * d;
        }
        const float dm = fminf(dist[k], acc);
        dist[k] = dm;
        if (dm > best_d) {  // ascending i within a thread: keeps the first index on ties
          best_d = dm;
          best_i = i;
        }
      }
    }
    // argmax over the workgroup, ties -> smallest index
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_down(best_d, off, 64);
      const int oi = __shfl_down(best_i, off, 64);
      if (od > best_d || (od == best_d && oi < best_i)) {
        best_d = od;
        best_i = oi;
      }
    }
    if (lane == 0) {
      s_best_d[wave] = best_d;
      s_best_i[wave] = best_i;
    }
    __syncthreads();
    if (tid == 0) {
      float bd = s_best_d[0];
      int bi = s_best_i[0];
      for (int w = 1; w < kFpsThreads / 64; ++w)
        if (s_best_d[w] > bd || (s_best_d[w] == bd && s_best_i[w] < bi)) {
          bd = s_best_d[w];
          bi = s_best_i[w];
        }
      s_cur = bi;
      ob[it] = bi;
    }
    __syncthreads();
    cur = s_cur;
  }
}

int launch_fps(const float* x, int B, int N, int C, int npoints, int start, long long* out_idx, hipStream_t s) {
  if (N > kFpsThreads * kFpsMaxPerThread || C > kFpsMaxC) return 1;
  hipLaunchKernelGGL(k_fps, dim3(B), dim3(kFpsThreads), 0, s, x, N, C, npoints, start, out_idx);
  return 0;
}

}  // namespace mmf
