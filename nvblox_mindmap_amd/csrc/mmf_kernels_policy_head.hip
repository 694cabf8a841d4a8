// The per-step head and tail of the diffusion head at inference (batch 1-2, a handful of trajectory tokens): everything that
// happens before the first attention layer and after the last one is ~35 library launches on tensors of a few hundred floats.
//
//   k_step_prologue  tokens = traj_encoder(trajectory) + position code; cond = silu(time embedding + history);
//                    ALL AdaLN scale/shift projections of the step (one GEMV, [120] x [120, NA]); 3-D rotary cos / sin of the
//                    trajectory positions, written straight into the first rows of the sequence-wide rotary tables
//   k_head_outputs   rotation_proj / position_proj of the trajectory rows of the two output stacks, the four small MLPs
//                    (position 3, rotation 6, openness 1, head yaw 1), concatenated into pred [B, L, G, 10]
#include "mmf_device.h"
#include "mmf_launch.h"

namespace mmf {

constexpr int kHD = 120;  // embedding dim

// grid (ceil(NA / 256) + 1, B), 256 threads.  Blocks x < nA: AdaLN outputs; block x == nA: tokens + rotary of batch element b.
__global__ __launch_bounds__(256) void k_step_prologue(const float* __restrict__ traj, int nt, const float* __restrict__ WeT,
                                                      const float* __restrict__ be, const float* __restrict__ pos_table,
                                                      const float* __restrict__ time_row, const float* __restrict__ history,
                                                      const float* __restrict__ freq, const float* __restrict__ AwT,
                                                      const float* __restrict__ Ab, int NA, float* __restrict__ tokens,
                                                      float* __restrict__ adaln, float* __restrict__ cos_out,
                                                      float* __restrict__ sin_out, long long rot_batch_stride) {
  const int b = blockIdx.y, nA = (NA + 255) / 256;
  if ((int)blockIdx.x < nA) {
    __shared__ float s_c[kHD];
    if (threadIdx.x < kHD) {
      const float c = time_row[threadIdx.x] + history[(size_t)b * kHD + threadIdx.x];
      s_c[threadIdx.x] = c / (1.0f + expf(-c));  // silu
    }
    __syncthreads();
    const int o = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (o < NA) {
      float acc = 0.0f;
#pragma unroll 8
      for (int c = 0; c < kHD; ++c) acc += s_c[c] * AwT[(size_t)c * NA + o];
      adaln[(size_t)b * NA + o] = acc + Ab[o];
    }
    return;
  }
  const int third = kHD / 3;
  for (int e = threadIdx.x; e < nt * kHD; e += 256) {
    const int i = e / kHD, j = e - i * kHD;
    const float* t = traj + ((size_t)b * nt + i) * 9;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; ++c) acc += t[c] * WeT[c * kHD + j];
    tokens[((size_t)b * nt + i) * kHD + j] = (acc + be[j]) + pos_table[i * kHD + j];
    const int a = j / third, k = (j - a * third) >> 1;
    const float ang = t[a] * freq[k];
    cos_out[(size_t)b * rot_batch_stride + (size_t)i * kHD + j] = cosf(ang);
    sin_out[(size_t)b * rot_batch_stride + (size_t)i * kHD + j] = sinf(ang);
  }
}

void launch_step_prologue(const float* traj, int B, int nt, const float* WeT, const float* be, const float* pos_table, const float* time_row,
                          const float* history, const float* freq, const float* AwT, const float* Ab, int NA, float* tokens, float* adaln,
                          float* cos_out, float* sin_out, long long rot_batch_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_step_prologue, dim3((NA + 255) / 256 + 1, B), dim3(256), 0, s, traj, nt, WeT, be, pos_table, time_row, history, freq, AwT,
                     Ab, NA, tokens, adaln, cos_out, sin_out, rot_batch_stride);
}

// ---- output heads ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxG = 4;

// out[g][j] = act(bias[j] + sum_c in[g][c] Wt[c * nout + j]) for g < G, j < nout <= 128; the reduction is split over the four
// 128-thread parts of the workgroup and summed through LDS.  `concat`: one input vector made of the G rows of `in` (head yaw).
__device__ __forceinline__ void dense(const float* __restrict__ Wt, const float* __restrict__ bias, int nin, int nout, const float (*in)[128],
                                      int G, bool concat, bool relu, float (*out)[128], float (*red)[kMaxG][128]) {
  const int j = threadIdx.x & 127, part = threadIdx.x >> 7;
  const int per = nin / 4, c0 = part * per;
  const int rows = concat ? 1 : G;
  float acc[kMaxG] = {0.f, 0.f, 0.f, 0.f};
  if (j < nout)
    for (int c = c0; c < c0 + per; ++c) {
      const float w = Wt[(size_t)c * nout + j];
      if (concat) {
        acc[0] += w * in[c / kHD][c % kHD];
      } else {
#pragma unroll
        for (int g = 0; g < kMaxG; ++g)
          if (g < G) acc[g] += w * in[g][c];
      }
    }
#pragma unroll
  for (int g = 0; g < kMaxG; ++g) red[part][g][j] = acc[g];
  __syncthreads();
  if (part == 0 && j < nout) {
    const float bb = bias[j];
    for (int g = 0; g < rows; ++g) {
      float v = ((red[0][g][j] + red[1][g][j]) + (red[2][g][j] + red[3][g][j])) + bb;
      out[g][j] = relu ? fmaxf(v, 0.0f) : v;
    }
  }
  __syncthreads();
}

struct HeadWeights {  // transposed ([in, out]) weights and biases
  const float *rp, *rpb, *pp, *ppb;        // rotation_proj, position_proj
  const float *r1, *r1b, *r2, *r2b;        // rotation_out
  const float *p1, *p1b, *p2, *p2b;        // position_out
  const float *o1, *o1b, *o2, *o2b;        // openness_out
  const float *y1, *y1b, *y2, *y2b;        // head_yaw_out (null: no head yaw)
};

// grid (B * L), 512 threads: one (batch element, horizon step) = G trajectory tokens
__global__ __launch_bounds__(512) void k_head_outputs(const float* __restrict__ rot_seq, const float* __restrict__ pos_seq,
                                                     long long seq_batch_stride, int L, int G, HeadWeights W, float* __restrict__ pred,
                                                     float* __restrict__ head_yaw) {
  __shared__ float s_in[2][kMaxG][128], s_feat[2][kMaxG][128], s_h[4][kMaxG][128], s_o[4][kMaxG][128];
  __shared__ float s_red[4][kMaxG][128];
  const int b = (int)blockIdx.x / L, l = (int)blockIdx.x % L;
  for (int e = threadIdx.x; e < 2 * G * kHD; e += 512) {
    const int which = e / (G * kHD), r = e - which * G * kHD, g = r / kHD, c = r - g * kHD;
    s_in[which][g][c] = (which ? pos_seq : rot_seq)[(size_t)b * seq_batch_stride + (size_t)(l * G + g) * kHD + c];
  }
  __syncthreads();
  dense(W.rp, W.rpb, kHD, kHD, s_in[0], G, false, false, s_feat[0], s_red);
  dense(W.pp, W.ppb, kHD, kHD, s_in[1], G, false, false, s_feat[1], s_red);
  dense(W.r1, W.r1b, kHD, kHD, s_feat[0], G, false, true, s_h[0], s_red);
  dense(W.p1, W.p1b, kHD, kHD, s_feat[1], G, false, true, s_h[1], s_red);
  dense(W.o1, W.o1b, kHD, kHD, s_feat[1], G, false, true, s_h[2], s_red);
  if (W.y1) dense(W.y1, W.y1b, kHD * G, kHD, s_feat[1], G, true, true, s_h[3], s_red);
  dense(W.p2, W.p2b, kHD, 3, s_h[1], G, false, false, s_o[0], s_red);
  dense(W.r2, W.r2b, kHD, 6, s_h[0], G, false, false, s_o[1], s_red);
  dense(W.o2, W.o2b, kHD, 1, s_h[2], G, false, false, s_o[2], s_red);
  if (W.y1) dense(W.y2, W.y2b, kHD, 1, s_h[3], 1, false, false, s_o[3], s_red);
  if ((int)threadIdx.x < G * 10) {
    const int g = threadIdx.x / 10, k = threadIdx.x % 10;
    const float v = k < 3 ? s_o[0][g][k] : (k < 9 ? s_o[1][g][k - 3] : s_o[2][g][0]);
    pred[(((size_t)b * L + l) * G + g) * 10 + k] = v;
  }
  if (W.y1 && threadIdx.x == 0) head_yaw[(size_t)b * L + l] = s_o[3][0][0];
}

int launch_head_outputs(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G, const float* const* w,
                        float* pred, float* head_yaw, hipStream_t s) {
  if (G < 1 || G > kMaxG) return 1;
  HeadWeights W{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15], w[16], w[17], w[18], w[19]};
  hipLaunchKernelGGL(k_head_outputs, dim3(B * L), dim3(512), 0, s, rot_seq, pos_seq, seq_batch_stride, L, G, W, pred, head_yaw);
  return 0;
}

}  // namespace mmf
