// mmf_kernels_image.hip -- image-side kernels of the fusion path (mindmap/image_processing):
// depth back-projection, mask erosion / feature-mask algebra, feature-map upsample+pad+cast.
// gfx950 / wave64.  All of these are pure streaming kernels bound by HBM bandwidth.
#include "mmf_launch.h"
#include "mmf_mask_device.h"

namespace mmf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// Back-projection (image_processing/backprojection.py:51-146):
//   p_C = depth * (K^-1 (u, v, 1)^T), u = column, v = row (integer pixel coordinates, no +0.5),
//   p_W = T (p_C, 1)^T, non-finite -> 0, output channel-first [B,3,H,W].
// 4 B read + 12 B written per pixel; VEC pixels per thread as 16-byte accesses.
// ------------------------------------------------------------------------------------------------
struct Mat3 {
  float m[9];
};

__device__ inline Mat3 inverse3(const float* K) {
  // cofactor inverse in float64, rounded once to float32
  const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double id = 1.0 / det;
  Mat3 r;
  r.m[0] = (float)(A * id);
  r.m[1] = (float)(-(b * i - c * h) * id);
  r.m[2] = (float)((b * f - c * e) * id);
  r.m[3] = (float)(B * id);
  r.m[4] = (float)((a * i - c * g) * id);
  r.m[5] = (float)(-(a * f - c * d) * id);
  r.m[6] = (float)(C * id);
  r.m[7] = (float)(-(a * h - b * g) * id);
  r.m[8] = (float)((a * e - b * d) * id);
  return r;
}

__device__ inline float finite_or_zero(float x) { return isfinite(x) ? x : 0.0f; }

template <int VEC>
__global__ __launch_bounds__(256) void k_backproject(const float* __restrict__ depth, const float* __restrict__ K,
                                                    const float* __restrict__ T, int H, int W, float* __restrict__ out) {
  const int b = blockIdx.y;
  const size_t HW = (size_t)H * W;
  const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (p0 >= HW) return;
  const Mat3 Ki = inverse3(K + 9 * b);
  const float* Tb = T + 16 * b;
  float d[VEC], ox[VEC], oy[VEC], oz[VEC];
  if (VEC == 4) {
    const float4 dv = *reinterpret_cast<const float4*>(depth + b * HW + p0);
    d[0] = dv.x;
    d[VEC > 1 ? 1 : 0] = dv.y;
    d[VEC > 2 ? 2 : 0] = dv.z;
    d[VEC > 3 ? 3 : 0] = dv.w;
  } else {
    d[0] = depth[b * HW + p0];
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const size_t p = p0 + k;
    const float u = (float)(int)(p % W), v = (float)(int)(p / W);
    const float rx = (u * Ki.m[0] + v * Ki.m[1]) + Ki.m[2];
    const float ry = (u * Ki.m[3] + v * Ki.m[4]) + Ki.m[5];
    const float rz = (u * Ki.m[6] + v * Ki.m[7]) + Ki.m[8];
    const float xc = d[k] * rx, yc = d[k] * ry, zc = d[k] * rz;
    ox[k] = finite_or_zero(((xc * Tb[0] + yc * Tb[1]) + zc * Tb[2]) + Tb[3]);
    oy[k] = finite_or_zero(((xc * Tb[4] + yc * Tb[5]) + zc * Tb[6]) + Tb[7]);
    oz[k] = finite_or_zero(((xc * Tb[8] + yc * Tb[9]) + zc * Tb[10]) + Tb[11]);
  }
  float* o = out + (size_t)b * 3 * HW + p0;
  if (VEC == 4) {
    *reinterpret_cast<float4*>(o) = make_float4(ox[0], ox[VEC > 1 ? 1 : 0], ox[VEC > 2 ? 2 : 0], ox[VEC > 3 ? 3 : 0]);
    *reinterpret_cast<float4*>(o + HW) = make_float4(oy[0], oy[VEC > 1 ? 1 : 0], oy[VEC > 2 ? 2 : 0], oy[VEC > 3 ? 3 : 0]);
    *reinterpret_cast<float4*>(o + 2 * HW) = make_float4(oz[0], oz[VEC > 1 ? 1 : 0], oz[VEC > 2 ? 2 : 0], oz[VEC > 3 ? 3 : 0]);
  } else {
    o[0] = ox[0];
    o[HW] = oy[0];
    o[2 * HW] = oz[0];
  }
}

// ------------------------------------------------------------------------------------------------
// Loader sample -> frame inputs (mapping/helpers/nvblox_input_helpers.py:57-69): rgb (3,H,W) float in [0,1] ->
// (H,W,3) uint8 as `(rgb * 255).to(torch.uint8)` (truncation), plus what the helper needs ON THE HOST, gathered into one
// 20-float record so that a single device->host copy (one synchronisation) serves it: the operands of its two range
// assertions (min, max, "a NaN was seen"), the camera pose 7-vector and the 3x3 intrinsics.
// 12 B read + 3 B written per pixel.  k_rgb_u8: 4 pixels per thread (three 16 B loads, three 4 B stores) when H*W % 4 == 0.
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void k_rgb_u8(const float* __restrict__ rgb, size_t HW, uint8_t* __restrict__ out,
                                               float* __restrict__ partial) {
  __shared__ float s_lo[4], s_hi[4];
  __shared__ int s_nan[4];
  float lo = INFINITY, hi = -INFINITY;
  int nan = 0;
  for (size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; p0 < HW; p0 += (size_t)gridDim.x * 256 * VEC) {
    float c[3][VEC];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4*>(rgb + ch * HW + p0);
        c[ch][0] = v.x;
        c[ch][VEC > 1 ? 1 : 0] = v.y;
        c[ch][VEC > 2 ? 2 : 0] = v.z;
        c[ch][VEC > 3 ? 3 : 0] = v.w;
      } else {
        c[ch][0] = rgb[ch * HW + p0];
      }
    }
    uint8_t b[3 * VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float v = c[ch][k];
        nan |= (v != v) ? 1 : 0;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        const float s = v * 255.0f;
        b[3 * k + ch] = (uint8_t)(int)(s < 0.0f ? 0.0f : (s > 255.0f ? 255.0f : s));  // (out of range only if the assertion fails anyway)
      }
    if (VEC == 4) {
      unsigned w[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        w[q] = (unsigned)b[(4 * q) % (3 * VEC)] | ((unsigned)b[(4 * q + 1) % (3 * VEC)] << 8) | ((unsigned)b[(4 * q + 2) % (3 * VEC)] << 16) |
               ((unsigned)b[(4 * q + 3) % (3 * VEC)] << 24);
      unsigned* o = reinterpret_cast<unsigned*>(out + 3 * p0);
      o[0] = w[0];
      o[1] = w[1];
      o[2] = w[2];
    } else {
      out[3 * p0] = b[0];
      out[3 * p0 + 1] = b[1 % (3 * VEC)];
      out[3 * p0 + 2] = b[2 % (3 * VEC)];
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d, 64));
    hi = fmaxf(hi, __shfl_xor(hi, d, 64));
    nan |= __shfl_xor(nan, d, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
    s_nan[wave] = nan;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x] = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    partial[3 * blockIdx.x + 1] = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    partial[3 * blockIdx.x + 2] = (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) ? 1.0f : 0.0f;
  }
}

// one workgroup: small[0..3] = {min, max, saw NaN, 0}, small[4..10] = pose, small[11..19] = K.
// HOST: `small` is a record in coherent host memory the caller polls -- every writer fences at system scope, then ONE release store of
// the call's sequence number into `flag` tells the host the 20 floats are there (no copy engine, no stream synchronisation).
template <bool HOST>
__global__ __launch_bounds__(256) void k_sample_small(const float* __restrict__ partial, int n_partial, const float* __restrict__ pose7,
                                                     const float* __restrict__ K9, float* __restrict__ small, unsigned* flag, unsigned seq) {
  __shared__ float s_lo[4], s_hi[4], s_nan[4];
  float lo = INFINITY, hi = -INFINITY, nan = 0.0f;
  for (int i = threadIdx.x; i < n_partial; i += 256) {
    lo = fminf(lo, partial[3 * i]);
    hi = fmaxf(hi, partial[3 * i + 1]);
    nan = fmaxf(nan, partial[3 * i + 2]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d, 64));
    hi = fmaxf(hi, __shfl_xor(hi, d, 64));
    nan = fmaxf(nan, __shfl_xor(nan, d, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
    s_nan[wave] = nan;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    small[0] = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    small[1] = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    small[2] = fmaxf(fmaxf(s_nan[0], s_nan[1]), fmaxf(s_nan[2], s_nan[3]));
    small[3] = 0.0f;
  }
  if (threadIdx.x < 7) small[4 + threadIdx.x] = pose7[threadIdx.x];
  if (threadIdx.x >= 64 && threadIdx.x < 73) small[11 + threadIdx.x - 64] = K9[threadIdx.x - 64];
  if (HOST) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int sample_inputs_scratch_floats() { return 3 * 1024; }

void launch_sample_inputs(const float* rgb_chw, int H, int W, const float* pose7, const float* K9, uint8_t* rgb_out, float* small,
                          float* scratch, hipStream_t s, unsigned* host_flag, unsigned seq) {
  const size_t HW = (size_t)H * W;
  const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(rgb_chw) & 15) == 0 && (reinterpret_cast<uintptr_t>(rgb_out) & 3) == 0;
  const size_t per = vec ? 1024 : 256;
  int g = (int)((HW + per - 1) / per);
  g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
  if (vec)
    hipLaunchKernelGGL(k_rgb_u8<4>, dim3(g), dim3(256), 0, s, rgb_chw, HW, rgb_out, scratch);
  else
    hipLaunchKernelGGL(k_rgb_u8<1>, dim3(g), dim3(256), 0, s, rgb_chw, HW, rgb_out, scratch);
  if (host_flag)  // `small` is the host-visible record (mmf_sample_frame_inputs_host)
    hipLaunchKernelGGL(k_sample_small<true>, dim3(1), dim3(256), 0, s, (const float*)scratch, g, pose7, K9, small, host_flag, seq);
  else
    hipLaunchKernelGGL(k_sample_small<false>, dim3(1), dim3(256), 0, s, (const float*)scratch, g, pose7, K9, small, (unsigned*)nullptr, 0u);
}

// ------------------------------------------------------------------------------------------------
// erode_mask (image_processing/image_mask_operations.py:16-41): k iterations of a zero-padded 3x3
// max-pool on the inverted mask == one (2k+1)x(2k+1) square dilation of the inverted mask, done
// separably: row pass into tmp, column pass out.  Pixels outside the image never contribute.
// tmp bit0: row-dilated (mask == 0);  bit1 (feature-mask variant): row-dilated !(depth > min_d).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rowpass(const uint8_t* __restrict__ mask, const float* __restrict__ depth, float min_d,
                                                int H, int W, int k0, int k1, uint8_t* __restrict__ tmp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const uint8_t* mrow = mask ? mask + (size_t)y * W : nullptr;
  unsigned r = 0;
  if (mrow) {
    const int lo = x - k0 < 0 ? 0 : x - k0, hi = x + k0 > W - 1 ? W - 1 : x + k0;
    for (int xx = lo; xx <= hi; ++xx) r |= (mrow[xx] == 0) ? 1u : 0u;
  }
  if (depth) {
    const float* drow = depth + (size_t)y * W;
    const int lo = x - k1 < 0 ? 0 : x - k1, hi = x + k1 > W - 1 ? W - 1 : x + k1;
    for (int xx = lo; xx <= hi; ++xx) r |= (!(drow[xx] > min_d)) ? 2u : 0u;
  }
  tmp[(size_t)y * W + x] = (uint8_t)r;
}

__global__ __launch_bounds__(256) void k_colpass_erode(const uint8_t* __restrict__ tmp, int H, int W, int k,
                                                      uint8_t* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int lo = y - k < 0 ? 0 : y - k, hi = y + k > H - 1 ? H - 1 : y + k;
  unsigned r = 0;
  for (int yy = lo; yy <= hi; ++yy) r |= tmp[(size_t)yy * W + x];
  out[(size_t)y * W + x] = (r & 1u) ? 0 : 1;
}

// Column pass + nearest upsample to (Hf,Wf) + border mask (nvblox_mapping_helpers.py:222-253,
// image_mask_operations.py:44-68).  Nearest source index = min(floor(dst * (in/out)), in-1).
__global__ __launch_bounds__(256) void k_colpass_feature_mask(const uint8_t* __restrict__ tmp, int H, int W, int k0, int k1,
                                                             int Hf, int Wf, float sh, float sw, int bh, int bw,
                                                             uint8_t* __restrict__ out) {
  const int xf = blockIdx.x * blockDim.x + threadIdx.x, yf = blockIdx.y;
  if (xf >= Wf) return;
  uint8_t res = 0;
  const bool border_ok = (bh <= 0 || bw <= 0) || (yf >= bh && yf < Hf - bh && xf >= bw && xf < Wf - bw);
  if (border_ok) {
    int ys = (int)floorf((float)yf * sh), xs = (int)floorf((float)xf * sw);
    ys = ys > H - 1 ? H - 1 : ys;
    xs = xs > W - 1 ? W - 1 : xs;
    unsigned r = 0;
    {
      const int lo = ys - k0 < 0 ? 0 : ys - k0, hi = ys + k0 > H - 1 ? H - 1 : ys + k0;
      for (int yy = lo; yy <= hi; ++yy) r |= tmp[(size_t)yy * W + xs] & 1u;
    }
    {
      const int lo = ys - k1 < 0 ? 0 : ys - k1, hi = ys + k1 > H - 1 ? H - 1 : ys + k1;
      for (int yy = lo; yy <= hi; ++yy) r |= tmp[(size_t)yy * W + xs] & 2u;
    }
    res = r ? 0 : 1;
  }
  out[(size_t)yf * Wf + xf] = res;
}

__global__ __launch_bounds__(256) void k_depth_mask(const uint8_t* __restrict__ mask, const float* __restrict__ depth, size_t n,
                                                   float min_d, uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool m = mask ? mask[i] != 0 : true;
  out[i] = (m && depth[i] > min_d) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Bit-packed mask algebra (fast path of integrate_frame's masks): bodies in mmf_mask_device.h.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_mask_rowbits(MaskJob J) {
  __shared__ u64 s_in[kMaxMaskWords], s_d[kMaxMaskWords];
  mask_rowbits_row(J, blockIdx.x, s_in, s_d);
}

__global__ __launch_bounds__(256) void k_mask_colemit(MaskJob J) {
  __shared__ u64 s_bad[kMaxMaskWords];
  mask_colemit_row(J, blockIdx.x, s_bad);
}

// ------------------------------------------------------------------------------------------------
// Feature-map upsample (feature_extraction.py:126-128,188-191,198-210 + nvblox_mapping_helpers.py:256):
// bilinear align_corners=False of a channels-last low-res map [h,w,Cin] f32 to [Hf,Wf,Cpad] f16 with
// channels >= Cin zero.  Write-bound (the low-res map is <= 1 MB and stays in L2): one thread = 8 channels x a run of
// kUpRun consecutive output pixels of one row.  The 4 corner vectors (4 x 32 B) stay in registers along the run and are
// re-fetched only when the run crosses into the next low-res cell (every Wf/w pixels), so L2 reads are ~1/8 of the
// bytes stored instead of 8x (first version: 290 GB/s).  Lanes of a wave cover consecutive channel chunks of the same
// pixel: every store instruction writes one contiguous piece of an output pixel's channel vector.
// ------------------------------------------------------------------------------------------------
constexpr int kUpRun = 16;

struct Up8 {
  float v[8];
};
template <bool VEC>
__device__ __forceinline__ Up8 up_load8(const float* __restrict__ low, int w, int Cin, int y, int x, int c0) {
  Up8 r;
  const float* p = low + ((size_t)y * w + x) * Cin + c0;
  if (VEC) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w, r.v[4] = b.x, r.v[5] = b.y, r.v[6] = b.z, r.v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = (c0 + k < Cin) ? p[k] : 0.0f;
  }
  return r;
}

// FMA: the spec switch mmf_params.fma_contraction applied to this op (mmf_upsample_features_spec; see mmf_device.h)
template <bool VEC, int FMA>
__global__ __launch_bounds__(256) void k_upsample_features(const float* __restrict__ low, int h, int w, int Cin,
                                                          __half* __restrict__ out, int Hf, int Wf, int Cpad, float sh,
                                                          float sw) {
  const int nch = Cpad >> 3;
  const int runs = (Wf + kUpRun - 1) / kUpRun;
  const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Hf * runs * nch;
  if (item >= total) return;
  const int ch = (int)(item % nch);
  const size_t run = item / nch;
  const int xr = (int)(run % runs), yf = (int)(run / runs);
  const int c0 = ch * 8;
  const int xbeg = xr * kUpRun, xend = xbeg + kUpRun < Wf ? xbeg + kUpRun : Wf;
  __half* orow = out + ((size_t)yf * Wf) * Cpad + c0;
  if (c0 >= Cin) {  // zero padding of the feature array
    half8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
    for (int xf = xbeg; xf < xend; ++xf) *reinterpret_cast<half8*>(orow + (size_t)xf * Cpad) = z;
    return;
  }
  float sy = madd<FMA>(sh, (float)yf + 0.5f, -0.5f);
  sy = sy < 0.0f ? 0.0f : sy;
  const int y0 = (int)sy < h - 1 ? (int)sy : h - 1;
  const int y1 = y0 < h - 1 ? y0 + 1 : y0;
  const float ly1 = sy - (float)y0;
  const float ly0 = 1.0f - ly1;
  int cx0 = -1, cx1 = -1;
  Up8 a00, a01, a10, a11;
  for (int xf = xbeg; xf < xend; ++xf) {
    float sx = madd<FMA>(sw, (float)xf + 0.5f, -0.5f);
    sx = sx < 0.0f ? 0.0f : sx;
    const int x0 = (int)sx < w - 1 ? (int)sx : w - 1;
    const int x1 = x0 < w - 1 ? x0 + 1 : x0;
    const float lx1 = sx - (float)x0;
    const float lx0 = 1.0f - lx1;
    if (x0 != cx0 || x1 != cx1) {
      a00 = up_load8<VEC>(low, w, Cin, y0, x0, c0);
      a01 = up_load8<VEC>(low, w, Cin, y0, x1, c0);
      a10 = up_load8<VEC>(low, w, Cin, y1, x0, c0);
      a11 = up_load8<VEC>(low, w, Cin, y1, x1, c0);
      cx0 = x0;
      cx1 = x1;
    }
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float val = madd2<FMA>(ly0, madd2<FMA>(lx0, a00.v[k], lx1, a01.v[k]), ly1, madd2<FMA>(lx0, a10.v[k], lx1, a11.v[k]));
      o[k] = (_Float16)val;
    }
    *reinterpret_cast<half8*>(orow + (size_t)xf * Cpad) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
void launch_backproject(const float* depth, const float* K, const float* T, int B, int H, int W, float* out, hipStream_t s) {
  const size_t HW = (size_t)H * W;
  if (B <= 0 || HW == 0) return;
  const bool vec = (HW % 4 == 0) && ((uintptr_t)depth % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (vec) {
    const unsigned gx = (unsigned)((HW / 4 + 255) / 256);
    hipLaunchKernelGGL(k_backproject<4>, dim3(gx, B), dim3(256), 0, s, depth, K, T, H, W, out);
  } else {
    const unsigned gx = (unsigned)((HW + 255) / 256);
    hipLaunchKernelGGL(k_backproject<1>, dim3(gx, B), dim3(256), 0, s, depth, K, T, H, W, out);
  }
}

void launch_erode(const uint8_t* mask, uint8_t* out, uint8_t* tmp, int H, int W, int k, hipStream_t s) {
  dim3 g((W + 255) / 256, H);
  hipLaunchKernelGGL(k_rowpass, g, dim3(256), 0, s, mask, (const float*)nullptr, 0.0f, H, W, k, 0, tmp);
  hipLaunchKernelGGL(k_colpass_erode, g, dim3(256), 0, s, (const uint8_t*)tmp, H, W, k, out);
}

void launch_feature_mask(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, int k_in, int k_depth,
                         int border_percent, int Hf, int Wf, uint8_t* out, uint8_t* tmp, hipStream_t s) {
  launch_frame_masks(input_mask, depth, H, W, min_d, k_in, k_depth, border_percent, Hf, Wf, nullptr, out, tmp, s);
}

// Fill the job descriptor of the bit-packed path; returns false if the image does not fit it (then the byte
// kernels are used).  `tmp` holds H*W bytes.
bool make_mask_job(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, int k_in, int k_depth,
                   int border_percent, int Hf, int Wf, uint8_t* depth_mask_out, uint8_t* feature_mask_out, uint8_t* tmp,
                   MaskJob& J) {
  const int nw = (W + 63) / 64;
  if (!(nw <= kMaxMaskWords && (size_t)2 * H * nw * sizeof(u64) <= (size_t)H * W && ((uintptr_t)tmp % 8 == 0))) return false;
  J.mask = input_mask;
  J.invert = 0;
  J.depth = depth;
  J.min_d = min_d;
  J.H = H;
  J.W = W;
  J.nw = nw;
  J.k0 = k_in;
  J.k1 = k_depth;
  J.bits_in = reinterpret_cast<u64*>(tmp);
  J.bits_d = J.bits_in + (size_t)H * nw;
  J.depth_mask_out = depth_mask_out;
  J.masked_depth_out = nullptr;
  J.Hf = Hf;
  J.Wf = Wf;
  J.sh = (float)H / (float)Hf;
  J.sw = (float)W / (float)Wf;
  // int(mask_border_percent * 0.01 * height) in Python double arithmetic (image_mask_operations.py:62-63)
  J.bh = (int)((double)border_percent * 0.01 * (double)Hf);
  J.bw = (int)((double)border_percent * 0.01 * (double)Wf);
  J.out = feature_mask_out;
  return true;
}

// depth_mask_out (optional) and the feature mask in two launches.  `tmp` holds H*W bytes.
void launch_frame_masks(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, int k_in, int k_depth,
                        int border_percent, int Hf, int Wf, uint8_t* depth_mask_out, uint8_t* feature_mask_out, uint8_t* tmp,
                        hipStream_t s) {
  MaskJob J;
  if (make_mask_job(input_mask, depth, H, W, min_d, k_in, k_depth, border_percent, Hf, Wf, depth_mask_out, feature_mask_out, tmp,
                    J)) {
    const int rb_threads = J.nw * 64 < 1024 ? J.nw * 64 : 1024;
    hipLaunchKernelGGL(k_mask_rowbits, dim3(H), dim3(rb_threads), 0, s, J);
    hipLaunchKernelGGL(k_mask_colemit, dim3(Hf), dim3(256), 0, s, J);
    return;
  }
  // generic byte path (very narrow or very wide images)
  const int bh = (int)((double)border_percent * 0.01 * (double)Hf);
  const int bw = (int)((double)border_percent * 0.01 * (double)Wf);
  const float sh = (float)H / (float)Hf, sw = (float)W / (float)Wf;
  if (depth_mask_out) launch_depth_mask(input_mask, depth, H, W, min_d, depth_mask_out, s);
  dim3 g((W + 255) / 256, H);
  hipLaunchKernelGGL(k_rowpass, g, dim3(256), 0, s, input_mask, depth, min_d, H, W, k_in, k_depth, tmp);
  dim3 gf((Wf + 255) / 256, Hf);
  hipLaunchKernelGGL(k_colpass_feature_mask, gf, dim3(256), 0, s, (const uint8_t*)tmp, H, W, k_in, k_depth, Hf, Wf, sh, sw, bh,
                     bw, feature_mask_out);
}

void launch_depth_mask(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, uint8_t* out, hipStream_t s) {
  const size_t n = (size_t)H * W;
  if (!n) return;
  hipLaunchKernelGGL(k_depth_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, input_mask, depth, n, min_d, out);
}

void launch_upsample_features(const float* lowres, int h, int w, int Cin, __half* out, int Hf, int Wf, int Cpad, hipStream_t s, bool fma) {
  const size_t total = (size_t)Hf * ((Wf + kUpRun - 1) / kUpRun) * (Cpad / 8);
  if (!total) return;
  const float sh = (float)h / (float)Hf, sw = (float)w / (float)Wf;
  const dim3 grid((unsigned)((total + 255) / 256));
  const bool vec = Cin % 8 == 0 && ((uintptr_t)lowres & 15) == 0;
  if (vec && fma)
    hipLaunchKernelGGL((k_upsample_features<true, true>), grid, dim3(256), 0, s, lowres, h, w, Cin, out, Hf, Wf, Cpad, sh, sw);
  else if (vec)
    hipLaunchKernelGGL((k_upsample_features<true, false>), grid, dim3(256), 0, s, lowres, h, w, Cin, out, Hf, Wf, Cpad, sh, sw);
  else if (fma)
    hipLaunchKernelGGL((k_upsample_features<false, true>), grid, dim3(256), 0, s, lowres, h, w, Cin, out, Hf, Wf, Cpad, sh, sw);
  else
    hipLaunchKernelGGL((k_upsample_features<false, false>), grid, dim3(256), 0, s, lowres, h, w, Cin, out, Hf, Wf, Cpad, sh, sw);
}

}  // namespace mmf
