// mmf_mask_device.h -- bit-packed mask algebra as device functions, so that the stand-alone mask kernels
// (mmf_kernels_image.hip) and the horizontally fused frame kernels (mmf_kernels_map.hip) share one body.
//
// integrate_frame's masks (mindmap/mapping/helpers/nvblox_mapping_helpers.py:201-253):
//   row pass   : one workgroup per image row; every wave ballots 64 pixels into a 64-bit word for each of the two
//                "bad pixel" predicates (input mask == 0, !(depth > min_d)); the words of the row are dilated
//                horizontally by k with shifts across word boundaries and stored as [H][nw] u64 bit-rows.  Also emits
//                depth_mask = input_mask & (depth > min_d) on the way.
//   column pass: one workgroup per OUTPUT row; all threads OR the bit-rows of the 2k+1 source rows, then expand bits
//                to bytes with the nearest-neighbour upsample and the border mask.
#pragma once
#include "mmf_device.h"

namespace mmf {

constexpr int kMaxMaskWords = 64;  // rows up to 4096 pixels

struct MaskJob {
  const uint8_t* mask;     // [H,W] input mask (0/1 bytes)
  int invert;              // != 0: the mask is used inverted (valid where the byte is 0): static = ~dynamic without a pass
  const float* depth;      // [H,W]
  float min_d;
  int H, W, nw, k0, k1;    // nw = ceil(W/64); k0 / k1 = erosion radius of the input / valid-depth mask
  u64* bits_in;            // [H][nw] row-dilated bad-pixel bit rows
  u64* bits_d;
  uint8_t* depth_mask_out; // [H,W] or nullptr
  float* masked_depth_out; // [H,W] or nullptr: depth where (mask != 0 and depth > min_d), else 0
  int Hf, Wf;
  float sh, sw;            // H/Hf, W/Wf
  int bh, bw;              // border in pixels
  uint8_t* out;            // [Hf,Wf] feature mask
  // optional: which 16x16-pixel patches hold a pixel of the depth mask.  The row pass stores `patch_tag` into
  // patch_flags[(y >> 4) * patches_x + (x >> 4)] for every such patch (nobody clears: a patch is marked iff its byte equals
  // this frame's tag).  Consumer: the sphere tracer of a paired frame skips ray patches no gate can read (SphereArgs).
  uint8_t* patch_flags = nullptr;
  int patches_x = 0, patches_y = 0, patch_tag = 0;
};

__device__ inline u64 row_word(const u64* w, int j, int nw) { return (j >= 0 && j < nw) ? w[j] : 0ull; }

// Row pass of image row y by the calling workgroup (any blockDim that is a multiple of 64).
__device__ inline void mask_rowbits_row(const MaskJob& J, int y, u64* s_in, u64* s_d) {
  const int lane = threadIdx.x & 63;
  const int nt = blockDim.x;
  for (int x0 = 0; x0 < J.nw * 64; x0 += nt) {
    const int x = x0 + threadIdx.x;
    bool bad_in = false, bad_d = false;
    if (x < J.W) {
      const size_t i = (size_t)y * J.W + x;
      bad_in = J.mask ? ((J.mask[i] == 0) != (J.invert != 0)) : false;
      bad_d = J.depth ? !(J.depth[i] > J.min_d) : false;
      if (J.depth_mask_out) J.depth_mask_out[i] = (!bad_in && !bad_d) ? 1 : 0;
      if (J.masked_depth_out) J.masked_depth_out[i] = (!bad_in && !bad_d) ? J.depth[i] : 0.0f;
    }
    const u64 b_in = __ballot(bad_in), b_d = __ballot(bad_d);
    if (lane == 0 && (x >> 6) < J.nw) {
      s_in[x >> 6] = b_in;
      s_d[x >> 6] = b_d;
    }
    if (J.patch_flags && (lane & 15) == 0 && x < J.W) {  // one lane per 16-pixel group: any pixel of the depth mask in it?
      const int nb = J.W - x < 16 ? J.W - x : 16;
      const u64 valid = (~(b_in | b_d) >> lane) & ((1ull << nb) - 1ull);
      const int pxx = x >> 4, pyy = y >> 4;
      if (valid && pxx < J.patches_x && pyy < J.patches_y) J.patch_flags[pyy * J.patches_x + pxx] = (uint8_t)J.patch_tag;
    }
  }
  __syncthreads();
  // horizontal dilation: bit x of the result = OR of bits [x-k, x+k]
  if (2 * J.nw <= 64 && J.k0 < 64 && J.k1 < 64) {
    // Rows up to 2048 pixels: one lane of the first wave per word (both masks), neighbour words by wave shuffles, and the
    // run [0, k] covered by DOUBLING -- r |= r << 1, << 2, << 4 ... then one last shift of k + 1 - covered -- in each
    // direction: ~2 log2(k) multiword shifts instead of 2 k.  (The linear loop below was ~1 500 instructions on ONE wave of
    // every row's workgroup: a quarter of the instruction issue of the whole k_front launch.)  Same bit set, exactly.
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const bool act = l < 2 * J.nw, second = l >= J.nw;
      const int w = second ? l - J.nw : l;
      const int k = second ? J.k1 : J.k0, kmax = J.k0 > J.k1 ? J.k0 : J.k1;
      const u64 x = act ? (second ? s_d : s_in)[w] : 0ull;
      const bool has_lo = act && w > 0, has_hi = act && w < J.nw - 1;
      u64 up = x, dn = x;  // dilated towards higher / lower pixel indices
      auto shift_up = [&](u64 v, int sft) {  // multiword v << sft, 0 < sft < 64
        const u64 nb = __shfl_up(v, 1, 64);
        return (v << sft) | ((has_lo ? nb : 0ull) >> (64 - sft));
      };
      auto shift_dn = [&](u64 v, int sft) {
        const u64 nb = __shfl_down(v, 1, 64);
        return (v >> sft) | ((has_hi ? nb : 0ull) << (64 - sft));
      };
      int cov = 1;  // shifts 0 .. cov - 1 are covered
      for (int c = 1; 2 * c <= kmax + 1; c *= 2) {  // uniform trip count; lanes whose k is smaller sit steps out
        const u64 a = shift_up(up, c), b = shift_dn(dn, c);
        if (2 * c <= k + 1) {
          up |= a;
          dn |= b;
          cov = 2 * c;
        }
      }
      const int rest = k + 1 - cov;  // 0 <= rest <= cov
      const u64 a = shift_up(up, rest > 0 ? rest : 1), b = shift_dn(dn, rest > 0 ? rest : 1);
      if (rest > 0) {
        up |= a;
        dn |= b;
      }
      if (act) (second ? J.bits_d : J.bits_in)[(size_t)y * J.nw + w] = up | dn;
    }
    return;
  }
  for (int j = threadIdx.x; j < 2 * J.nw; j += nt) {
    const bool second = j >= J.nw;
    const int w = second ? j - J.nw : j;
    const u64* src = second ? s_d : s_in;
    const int k = second ? J.k1 : J.k0;
    u64 r = src[w];
    for (int s = 1; s <= k; ++s) {
      const int q = s >> 6, sh = s & 63;  // shift by s = q words + sh bits
      u64 left = row_word(src, w - q, J.nw) << sh;
      if (sh) left |= row_word(src, w - q - 1, J.nw) >> (64 - sh);
      u64 right = row_word(src, w + q, J.nw) >> sh;
      if (sh) right |= row_word(src, w + q + 1, J.nw) << (64 - sh);
      r |= left | right;
    }
    (second ? J.bits_d : J.bits_in)[(size_t)y * J.nw + w] = r;
  }
}

// Column pass + nearest upsample + border of output row yf by the calling workgroup.
__device__ inline void mask_colemit_row(const MaskJob& J, int yf, u64* s_bad) {
  const int nt = blockDim.x;
  const bool row_ok = (J.bh <= 0 || J.bw <= 0) || (yf >= J.bh && yf < J.Hf - J.bh);
  if (row_ok) {  // block-uniform
    int ys = (int)floorf((float)yf * J.sh);
    ys = ys > J.H - 1 ? J.H - 1 : ys;
    for (int j = threadIdx.x; j < J.nw; j += nt) s_bad[j] = 0ull;
    __syncthreads();
    // vertical OR of the 2k+1 source bit-rows, spread over all threads: thread -> (word, row phase)
    const int groups = nt / J.nw > 0 ? nt / J.nw : 1;
    const int j = threadIdx.x % J.nw, g = threadIdx.x / J.nw;
    if (g < groups) {
      u64 r = 0;
      {
        const int lo = ys - J.k0 < 0 ? 0 : ys - J.k0, hi = ys + J.k0 > J.H - 1 ? J.H - 1 : ys + J.k0;
        for (int yy = lo + g; yy <= hi; yy += groups) r |= J.bits_in[(size_t)yy * J.nw + j];
      }
      {
        const int lo = ys - J.k1 < 0 ? 0 : ys - J.k1, hi = ys + J.k1 > J.H - 1 ? J.H - 1 : ys + J.k1;
        for (int yy = lo + g; yy <= hi; yy += groups) r |= J.bits_d[(size_t)yy * J.nw + j];
      }
      if (r) atomicOr(&s_bad[j], r);
    }
    __syncthreads();
  }
  for (int xf = threadIdx.x; xf < J.Wf; xf += nt) {
    uint8_t res = 0;
    if (row_ok && ((J.bh <= 0 || J.bw <= 0) || (xf >= J.bw && xf < J.Wf - J.bw))) {
      int xs = (int)floorf((float)xf * J.sw);
      xs = xs > J.W - 1 ? J.W - 1 : xs;
      res = ((s_bad[xs >> 6] >> (xs & 63)) & 1ull) ? 0 : 1;
    }
    J.out[(size_t)yf * J.Wf + xf] = res;
  }
}

}  // namespace mmf
