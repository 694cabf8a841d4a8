// The per-step head and tail of the diffusion head at inference (batch 1-2, a handful of trajectory tokens): everything that
// happens before the first attention layer and after the last one is ~35 library launches on tensors of a few hundred floats.
//
//   k_step_prologue  tokens = traj_encoder(trajectory) + position code; cond = silu(time embedding + history);
//                    ALL AdaLN scale/shift projections of the step (one GEMV, [120] x [120, NA]); 3-D rotary cos / sin of the
//                    trajectory positions, written straight into the first rows of the sequence-wide rotary tables
//   k_head_outputs   rotation_proj / position_proj of the trajectory rows of the two output stacks, the four small MLPs
//                    (position 3, rotation 6, openness 1, head yaw 1), concatenated into pred [B, L, G, 10]; as the step's TAIL
//                    also the reverse-diffusion update of the trajectory and the next step's tokens / rotary codes
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

// N independent, coalesced weight loads of output column j (rows c0 .. c0 + N of the transposed matrix) ...
template <int N>
__device__ __forceinline__ void load_cols(const float* __restrict__ Wt, int j, int c0, float (&w)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c) w[c] = Wt[(size_t)(c0 + c) * kHD + j];
}
// ... and acc[g] += sum_c in[g][c0 + c] * w[c] for g < G (compile time: no branches in the unrolled body; 16-byte LDS reads)
template <int N, int G>
__device__ __forceinline__ void fma_cols(const float (&w)[N], int c0, const float (*in)[128], float (&acc)[kMaxG]) {
#pragma unroll
  for (int c = 0; c < N; c += 4)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(&in[g][c0 + c]);
      acc[g] += w[c] * v.x;
      acc[g] += w[c + 1] * v.y;
      acc[g] += w[c + 2] * v.z;
      acc[g] += w[c + 3] * v.w;
    }
}

struct StepCoef {  // DDPMScheduler.step_coefficients
  float s1, inv_s2, c0, c1, sigma, clip;
};
struct StepTail {  // the reverse-diffusion update and the next step's trajectory tokens (null traj: outputs only)
  const float *traj, *noise;            // [B, L, G, 9]
  StepCoef pos, rot;                    // channels [0, 3) / [3, 9)
  float* traj_out;                      // [B, L, G, 9]
  const float *WeT, *be, *pos_table, *freq;
  float *tokens_out, *cos_out, *sin_out;  // tokens_out null: last step, no next tokens
  long long rot_batch_stride;
};

// grid (B * L, 2 roles), 512 threads = 4 parts x 128 channel threads: one (batch element, horizon step) = G trajectory tokens.
// Role 0 = rotation and position branches (+ the step tail, which needs exactly their outputs), role 1 = openness and head-yaw
// branches: two compute units share the weight traffic (6 matrices of 57 KB), each thread has 120 weight loads in flight.
//   stage A  rotation_proj (parts 0, 1: halves of the reduction; role 0 only) | position_proj (parts 2, 3)
//   stage B  first MLP layers, ReLU: role 0 rotation (parts 0, 1) | position (parts 2, 3); role 1 openness | head yaw
//   stage C  the scalar outputs, one 8-lane group each: role 0 position 3 + rotation 6 per token; role 1 openness per token + yaw
//   tail     (role 0) x_{t-1} from (x_t, predicted noise, pre-drawn noise); next step's tokens and rotary codes
template <int G>
__global__ __launch_bounds__(512) void k_head_outputs(const float* __restrict__ rot_seq, const float* __restrict__ pos_seq,
                                                     long long seq_batch_stride, int L, HeadWeights W, float* __restrict__ pred,
                                                     float* __restrict__ head_yaw, StepTail T) {
  __shared__ __attribute__((aligned(16))) float s_in[2][kMaxG][128], s_feat[2][kMaxG][128], s_h[2][kMaxG][128];
  __shared__ float s_red[4][kMaxG][128];
  __shared__ float s_pred[kMaxG][10], s_traj[kMaxG][9];
  const int b = (int)blockIdx.x / L, l = (int)blockIdx.x % L, role = blockIdx.y;
  const int j = threadIdx.x & 127, part = threadIdx.x >> 7, half = part & 1, pair = part >> 1;
  const bool act = j < kHD;
  const int jc = act ? j : 0;
  const bool yaw_on = W.y1 != nullptr;
  // stage B matrix of this thread's pair: role 0: rotation_out.0 | position_out.0; role 1: openness_out.0 | head_yaw_out.0
  const float* WB = role == 0 ? (pair == 0 ? W.r1 : W.p1) : (pair == 0 ? W.o1 : W.y1);
  const bool doA = role == 0 || pair == 1, doB = role == 0 || pair == 0 || yaw_on, yawB = role == 1 && pair == 1;
  float wA[60], wB[60];
  if (doA) load_cols<60>(pair == 0 ? W.rp : W.pp, jc, half * 60, wA);
  if (doB) load_cols<60>(WB, jc, half * 60, wB);  // head yaw: rows of token 0 (the other tokens' rows follow in the loop below)
  for (int e = threadIdx.x; e < 2 * G * kHD; e += 512) {
    const int which = e / (G * kHD), r = e - which * G * kHD, g = r / kHD, c = r - g * kHD;
    s_in[which][g][c] = (which ? pos_seq : rot_seq)[(size_t)b * seq_batch_stride + (size_t)(l * G + g) * kHD + c];
  }
  __syncthreads();
  {  // stage A
    float acc[kMaxG] = {0.f, 0.f, 0.f, 0.f};
    if (doA) fma_cols<60, G>(wA, half * 60, s_in[pair], acc);
#pragma unroll
    for (int g = 0; g < G; ++g) s_red[part][g][j] = acc[g];
    __syncthreads();
    if (act && half == 0 && doA) {
      const float bb = (pair ? W.ppb : W.rpb)[j];
#pragma unroll
      for (int g = 0; g < G; ++g) s_feat[pair][g][j] = (s_red[part][g][j] + s_red[part + 1][g][j]) + bb;
    }
    __syncthreads();
  }
  {  // stage B
    float acc[kMaxG] = {0.f, 0.f, 0.f, 0.f};
    if (yawB) {  // one input vector made of the G position features: [120 G] x [120 G, 120]
      if (yaw_on) {
        fma_cols<60, 1>(wB, half * 60, &s_feat[1][0], acc);
#pragma unroll
        for (int g = 1; g < G; ++g) {
          load_cols<60>(W.y1 + (size_t)g * kHD * kHD, jc, half * 60, wB);
          fma_cols<60, 1>(wB, half * 60, &s_feat[1][g], acc);
        }
      }
    } else {
      fma_cols<60, G>(wB, half * 60, s_feat[role == 0 ? pair : 1], acc);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) s_red[part][g][j] = acc[g];
    __syncthreads();
    if (act && half == 0 && doB) {
      const float bb = (role == 0 ? (pair == 0 ? W.r1b : W.p1b) : (pair == 0 ? W.o1b : W.y1b))[j];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (!yawB || g == 0) s_h[pair][g][j] = fmaxf((s_red[part][g][j] + s_red[part + 1][g][j]) + bb, 0.0f);
    }
    __syncthreads();
  }
  {  // stage C: one scalar output = one 8-lane group (15 channels per lane)
    const int grp = threadIdx.x >> 3, q = threadIdx.x & 7;
    // role 0: group = 9 g + k, k in 0..2 position (hidden s_h[1]), 3..8 rotation (s_h[0]); role 1: g < G openness (s_h[0]), G: yaw (s_h[1][0])
    const int n_out = role == 0 ? 9 * G : G + (yaw_on ? 1 : 0);
    const bool live = grp < n_out;
    const bool yaw = role == 1 && grp == G;
    const int g = (!live || yaw) ? 0 : (role == 0 ? grp / 9 : grp), k = role == 0 ? grp - 9 * g : 9;
    const float* Wt = role == 0 ? (k < 3 ? W.p2 : W.r2) : (yaw ? W.y2 : W.o2);
    const float* bs = role == 0 ? (k < 3 ? W.p2b : W.r2b) : (yaw ? W.y2b : W.o2b);
    const int nout = role == 0 ? (k < 3 ? 3 : 6) : 1, col = role == 0 ? (k < 3 ? k : k - 3) : 0;
    const float* h = role == 0 ? s_h[k < 3 ? 1 : 0][g] : s_h[yaw ? 1 : 0][g];
    float acc = 0.0f;
    if (live) {
#pragma unroll
      for (int i = 0; i < 15; ++i) {
        const int c = q * 15 + i;
        acc += h[c] * Wt[c * nout + col];
      }
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (live && q == 0) {
      const float v = acc + bs[col];
      if (yaw) {
        head_yaw[(size_t)b * L + l] = v;
      } else {
        pred[(((size_t)b * L + l) * G + g) * 10 + k] = v;
        s_pred[g][k] = v;
      }
    }
  }
  if (role != 0 || T.traj == nullptr) return;
  __syncthreads();
  if ((int)threadIdx.x < G * 9) {  // x_{t-1}: the arithmetic of k_ddpm_step
    const int g = threadIdx.x / 9, c = threadIdx.x % 9;
    const size_t e = (((size_t)b * L + l) * G + g) * 9 + c;
    const StepCoef K = c < 3 ? T.pos : T.rot;
    const float xv = T.traj[e];
    float x0 = (xv - K.s1 * s_pred[g][c]) * K.inv_s2;
    if (K.clip > 0.0f) x0 = fminf(fmaxf(x0, -K.clip), K.clip);
    float prev = K.c0 * x0 + K.c1 * xv;
    if (K.sigma > 0.0f) prev = prev + K.sigma * T.noise[e];
    T.traj_out[e] = prev;
    s_traj[g][c] = prev;
  }
  if (T.tokens_out == nullptr) return;
  __syncthreads();
  const int third = kHD / 3, nt = L * G;
  for (int e = threadIdx.x; e < G * kHD; e += 512) {  // the next step's trajectory tokens and rotary codes (k_step_prologue)
    const int g = e / kHD, c = e - g * kHD, i = l * G + g;
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < 9; ++u) acc += s_traj[g][u] * T.WeT[u * kHD + c];
    T.tokens_out[((size_t)b * nt + i) * kHD + c] = (acc + T.be[c]) + T.pos_table[i * kHD + c];
    const int a = c / third, k = (c - a * third) >> 1;
    const float ang = s_traj[g][a] * T.freq[k];
    T.cos_out[(size_t)b * T.rot_batch_stride + (size_t)i * kHD + c] = cosf(ang);
    T.sin_out[(size_t)b * T.rot_batch_stride + (size_t)i * kHD + c] = sinf(ang);
  }
}

static HeadWeights head_weights(const float* const* w) {
  return HeadWeights{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15], w[16], w[17], w[18], w[19]};
}

template <int G>
static void head_outputs_g(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, const HeadWeights& W, float* pred,
                           float* head_yaw, const StepTail& T, hipStream_t s) {
  hipLaunchKernelGGL(k_head_outputs<G>, dim3(B * L, 2), dim3(512), 0, s, rot_seq, pos_seq, seq_batch_stride, L, W, pred, head_yaw, T);
}
static int head_outputs_dispatch(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G,
                                 const HeadWeights& W, float* pred, float* head_yaw, const StepTail& T, hipStream_t s) {
  switch (G) {
    case 1: head_outputs_g<1>(rot_seq, pos_seq, seq_batch_stride, B, L, W, pred, head_yaw, T, s); return 0;
    case 2: head_outputs_g<2>(rot_seq, pos_seq, seq_batch_stride, B, L, W, pred, head_yaw, T, s); return 0;
    case 3: head_outputs_g<3>(rot_seq, pos_seq, seq_batch_stride, B, L, W, pred, head_yaw, T, s); return 0;
    case 4: head_outputs_g<4>(rot_seq, pos_seq, seq_batch_stride, B, L, W, pred, head_yaw, T, s); return 0;
    default: return 1;
  }
}

int launch_head_outputs(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G, const float* const* w,
                        float* pred, float* head_yaw, hipStream_t s) {
  StepTail T{};
  return head_outputs_dispatch(rot_seq, pos_seq, seq_batch_stride, B, L, G, head_weights(w), pred, head_yaw, T, s);
}

int launch_step_tail(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G, const float* const* w,
                     float* pred, float* head_yaw, const float* traj, const float* noise, const float* coef_pos, const float* coef_rot,
                     float* traj_out, const float* WeT, const float* be, const float* pos_table, const float* freq, float* tokens_out,
                     float* cos_out, float* sin_out, long long rot_batch_stride, hipStream_t s) {
  StepTail T{traj, noise, StepCoef{coef_pos[0], coef_pos[1], coef_pos[2], coef_pos[3], coef_pos[4], coef_pos[5]},
             StepCoef{coef_rot[0], coef_rot[1], coef_rot[2], coef_rot[3], coef_rot[4], coef_rot[5]}, traj_out, WeT, be, pos_table, freq,
             tokens_out, cos_out, sin_out, rot_batch_stride};
  return head_outputs_dispatch(rot_seq, pos_seq, seq_batch_stride, B, L, G, head_weights(w), pred, head_yaw, T, s);
}

}  // namespace mmf
