// The attention kernel of the diffusion head's inference path on the matrix cores (the projections and the out_proj +
// LayerNorm + feed-forward block around it: mmf_kernels_policy_layer.hip; thread-per-channel forms: mmf_kernels_policy.hip).
//
// A denoising step works on ~616 tokens x 120 channels, 8 heads of 15 channels.  Tiles of 16 queries x 16 keys on
// v_mfma_f32_16x16x16_f16; every f32 product is three fp16 products of operands split x = hi + lo / 2048 (22-bit mantissas, f32
// accumulation, the two cross terms in their own accumulator: see mmf_kernels_policy_layer.hip) -- q, k, v arrive split from the
// projection kernels, the probabilities are split in registers.
//
//   operand maps (lane l, j = l & 15, s = l >> 4):   A[i = j][k = 4 s + t]   B[k = 4 s + t][col = j]   D[row = 4 s + r][col = j]
//
//   k_attention_heads softmax(q k^T / sqrt(dh) + padding) v over HEAD-MAJOR operands padded to 16 channels (Qp, Kp
//                     [B, H, L16, 16], Vt [B, H, 16, L16], every aligned group of four values = {4 hi | 4 lo} halves in 16
//                     bytes: an operand of a tile is one aligned 16-byte access per lane).  Scores are produced TRANSPOSED
//                     (S^T = K Q^T), which leaves each lane holding, for its query row, exactly the four probabilities the B
//                     operand of O^T = V^T P^T needs, and makes the softmax statistics of a query row lane-aligned with the
//                     columns of O^T: no transposes, no LDS, two shuffles per reduction.  The keys of a (query tile, head)
//                     are split over the waves of the workgroup and merged once through LDS
#include <cstdlib>

#include "mmf_device.h"
#include "mmf_launch.h"
#include "mmf_trace_device.h"

namespace mmf {
// Phase marks of the instrumented build (`make WG_TRACE=1`, tools/policy_phase_trace.py): thread 0 of a workgroup stores the
// 100 MHz wall clock at up to 8 points, behind the 6 x 8192 frame records of the trace buffer, at record (8 u64) `base` +
// linear workgroup index.  `dep` pins the mark behind the value's producer.  Compiled out of the default build.
#ifdef MMF_WG_TRACE
__device__ __forceinline__ void pt_mark(int base, int i, float dep) {
  asm volatile("" ::"v"(dep));
  if (g_wg_trace && threadIdx.x == 0) {
    const long long blk = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
    const long long off = 3ll * 6 * 8192 + 8 * (base + blk) + i;
    if (off < 3ll * g_wg_trace_cap) g_wg_trace[off] = (unsigned long long)wall_clock64();
  }
}
#define MMF_PT(base, i, dep) ::mmf::pt_mark(base, i, dep)
#endif
}  // namespace mmf

#include "mmf_policy_attention.h"

namespace mmf {
using namespace att;

// grid (query tiles | key splits, H, B), 64 NW threads
template <int NW, int CH, int SPLIT>
__global__ __launch_bounds__(64 * NW) void k_attention_heads(const float* __restrict__ Qp, const float* __restrict__ Kp,
                                                            const float* __restrict__ Vt, const uint8_t* __restrict__ pad,
                                                            float* __restrict__ out, int Lq, int Lq16, int Lk, int Lk16, float scale) {
  __shared__ AttLds LD;
  const int split = SPLIT > 1 ? (int)blockIdx.x : 0;
  const int q0 = SPLIT > 1 ? 0 : (int)blockIdx.x * 16;
  attention_body<NW, CH, SPLIT>(Qp, Kp, Vt, pad, out, nullptr, 0u, Lq, Lq16, Lk, Lk16, scale, split, q0, (int)blockIdx.y, (int)blockIdx.z, LD);
}

MMF_DEFINE_WG_TRACE_SETTER(set_wg_trace_policy)

int launch_attention_heads(const float* Qp, const float* Kp, const float* Vt, const uint8_t* pad, float* out, int B, int Lq, int Lk,
                           int H, int dh, hipStream_t s) {
  if (H != kH || dh != kDH) return 1;
  const int Lq16 = (Lq + 15) / 16 * 16, Lk16 = (Lk + 15) / 16 * 16;
  const float scale = 1.0f / sqrtf((float)dh);
  if (Lk16 / 16 > 80 && Lq16 / 16 * H * B < 128)  // long key axis, few workgroups: spread the keys over 16 waves
    hipLaunchKernelGGL((k_attention_heads<16, 6, 1>), dim3(Lq16 / 16, H, B), dim3(1024), 0, s, Qp, Kp, Vt, pad, out, Lq, Lq16, Lk, Lk16, scale);
  else {
    static const int nw = getenv("MMF_DEBUG_ATT_NW") ? atoi(getenv("MMF_DEBUG_ATT_NW")) : 8;  // 8 waves: 7.9 us at the policy shape (4: 8.6, 16: 10.1)
    if (nw == 8)
      hipLaunchKernelGGL((k_attention_heads<8, 5, 1>), dim3(Lq16 / 16, H, B), dim3(512), 0, s, Qp, Kp, Vt, pad, out, Lq, Lq16, Lk, Lk16, scale);
    else if (nw == 16)
      hipLaunchKernelGGL((k_attention_heads<16, 3, 1>), dim3(Lq16 / 16, H, B), dim3(1024), 0, s, Qp, Kp, Vt, pad, out, Lq, Lq16, Lk, Lk16, scale);
    else
      hipLaunchKernelGGL((k_attention_heads<4, 10, 1>), dim3(Lq16 / 16, H, B), dim3(256), 0, s, Qp, Kp, Vt, pad, out, Lq, Lq16, Lk, Lk16, scale);
  }
  return 0;
}

// Lq <= 16 query rows over a long key axis: kAttSplit workgroups of 16 waves per (batch element, head), partial results out
constexpr int kAttSplit = 4;
int launch_attention_heads_split(const float* Qp, const float* Kp, const float* Vt, const uint8_t* pad, float* partials, int B, int Lq, int Lk,
                                 int H, int dh, hipStream_t s) {
  if (H != kH || dh != kDH || Lq > 16) return 1;
  const int Lk16 = (Lk + 15) / 16 * 16;
  hipLaunchKernelGGL((k_attention_heads<16, 3, kAttSplit>), dim3(kAttSplit, H, B), dim3(1024), 0, s, Qp, Kp, Vt, pad, partials, Lq, 16, Lk, Lk16,
                     1.0f / sqrtf((float)dh));
  return kAttSplit << 8;  // (number of splits << 8): the caller sizes / passes on the partial buffer with it
}

}  // namespace mmf
