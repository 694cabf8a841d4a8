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

struct HeadWeights {  // transposed ([in, out]) weights and biases
  const float *rp, *rpb, *pp, *ppb;        // rotation_proj, position_proj
  const float *r1, *r1b, *r2, *r2b;        // rotation_out
  const float *p1, *p1b, *p2, *p2b;        // position_out
  const float *o1, *o1b, *o2, *o2b;        // openness_out
  const float *y1, *y1b, *y2, *y2b;        // head_yaw_out (null: no head yaw)
};

// acc[g] += sum_{c in [c0, c0 + N)} in[g][c] * Wt[c * kHD + j]: N independent, coalesced weight loads in flight at once
template <int N>
__device__ __forceinline__ void dot_cols(const float* __restrict__ Wt, int j, int c0, const float (*in)[128], int G, float (&acc)[kMaxG]) {
  float w[N];
#pragma unroll
  for (int c = 0; c < N; ++c) w[c] = Wt[(size_t)(c0 + c) * kHD + j];
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int g = 0; g < kMaxG; ++g)
      if (g < G) acc[g] += w[c] * in[g][c0 + c];
}

// grid (B * L), 512 threads = 4 parts x 128 channel threads: one (batch element, horizon step) = G trajectory tokens.
//   stage A  rotation_proj (parts 0, 1: halves of the reduction) | position_proj (parts 2, 3)
//   stage B  first layers of the four MLPs, one per part (rotation | position | openness | head yaw), ReLU
//   stage C  their 10 G + 1 scalar outputs, one 8-lane group each
__global__ __launch_bounds__(512) void k_head_outputs(const float* __restrict__ rot_seq, const float* __restrict__ pos_seq,
                                                     long long seq_batch_stride, int L, int G, HeadWeights W, float* __restrict__ pred,
                                                     float* __restrict__ head_yaw) {
  __shared__ float s_in[2][kMaxG][128], s_feat[2][kMaxG][128], s_h[4][kMaxG][128];
  __shared__ float s_red[4][kMaxG][128];
  const int b = (int)blockIdx.x / L, l = (int)blockIdx.x % L;
  const int j = threadIdx.x & 127, part = threadIdx.x >> 7;
  const bool act = j < kHD;
  for (int e = threadIdx.x; e < 2 * G * kHD; e += 512) {
    const int which = e / (G * kHD), r = e - which * G * kHD, g = r / kHD, c = r - g * kHD;
    s_in[which][g][c] = (which ? pos_seq : rot_seq)[(size_t)b * seq_batch_stride + (size_t)(l * G + g) * kHD + c];
  }
  __syncthreads();
  {  // stage A
    float acc[kMaxG] = {0.f, 0.f, 0.f, 0.f};
    if (act) dot_cols<60>(part < 2 ? W.rp : W.pp, j, (part & 1) * 60, s_in[part >> 1], G, acc);
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) s_red[part][g][j] = acc[g];
    __syncthreads();
    if (act && !(part & 1)) {
      const float bb = (part ? W.ppb : W.rpb)[j];
      for (int g = 0; g < G; ++g) s_feat[part >> 1][g][j] = (s_red[part][g][j] + s_red[part + 1][g][j]) + bb;
    }
    __syncthreads();
  }
  {  // stage B
    float acc[kMaxG] = {0.f, 0.f, 0.f, 0.f};
    if (part < 3) {
      const float* Wt = part == 0 ? W.r1 : (part == 1 ? W.p1 : W.o1);
      const float* bs = part == 0 ? W.r1b : (part == 1 ? W.p1b : W.o1b);
      if (act) {
        dot_cols<60>(Wt, j, 0, s_feat[part ? 1 : 0], G, acc);
        dot_cols<60>(Wt, j, 60, s_feat[part ? 1 : 0], G, acc);
        const float bb = bs[j];
        for (int g = 0; g < G; ++g) s_h[part][g][j] = fmaxf(acc[g] + bb, 0.0f);
      }
    } else if (W.y1 && act) {  // head yaw: one input vector made of the G position features
      float a1[kMaxG] = {0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < G; ++g) {
        dot_cols<60>(W.y1 + (size_t)g * kHD * kHD, j, 0, &s_feat[1][g], 1, a1);
        dot_cols<60>(W.y1 + (size_t)g * kHD * kHD, j, 60, &s_feat[1][g], 1, a1);
      }
      s_h[3][0][j] = fmaxf(a1[0] + W.y1b[j], 0.0f);
    }
    __syncthreads();
  }
  {  // stage C: output o of token g = one 8-lane group (15 channels per lane)
    const int grp = threadIdx.x >> 3, q = threadIdx.x & 7;
    const int n_out = 10 * G + (W.y1 ? 1 : 0);
    const bool live = grp < n_out;
    const int g = live ? (grp < 10 * G ? grp / 10 : 0) : 0, k = grp - 10 * g;  // k: 0-2 position, 3-8 rotation, 9 openness, 10 yaw
    const bool yaw = grp == 10 * G;
    const float* Wt = yaw ? W.y2 : (k < 3 ? W.p2 : (k < 9 ? W.r2 : W.o2));
    const int nout = yaw ? 1 : (k < 3 ? 3 : (k < 9 ? 6 : 1)), col = yaw ? 0 : (k < 3 ? k : (k < 9 ? k - 3 : 0));
    const float(*h)[128] = yaw ? s_h[3] : (k < 3 ? s_h[1] : (k < 9 ? s_h[0] : s_h[2]));
    float acc = 0.0f;
    if (live) {
#pragma unroll
      for (int i = 0; i < 15; ++i) {
        const int c = q * 15 + i;
        acc += h[yaw ? 0 : g][c] * Wt[c * nout + col];
      }
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (live && q == 0) {
      const float bb = (yaw ? W.y2b : (k < 3 ? W.p2b : (k < 9 ? W.r2b : W.o2b)))[col];
      if (yaw)
        head_yaw[(size_t)b * L + l] = acc + bb;
      else
        pred[(((size_t)b * L + l) * G + g) * 10 + k] = acc + bb;
    }
  }
}

int launch_head_outputs(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G, const float* const* w,
                        float* pred, float* head_yaw, hipStream_t s) {
  if (G < 1 || G > kMaxG) return 1;
  HeadWeights W{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15], w[16], w[17], w[18], w[19]};
  hipLaunchKernelGGL(k_head_outputs, dim3(B * L), dim3(512), 0, s, rot_seq, pos_seq, seq_batch_stride, L, G, W, pred, head_yaw);
  return 0;
}

}  // namespace mmf
