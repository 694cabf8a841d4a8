// mmf_app_device.h -- device code of the appearance update (gate, colour blend, feature gating, survivor list, balanced row
// update) as inline functions: the roles of the kernels of mmf_kernels_app.hip AND of the launches of mmf_kernels_map.hip that host
// the previous frame's appearance tail (deferred mode, mmf_set_deferred_feature_rows).
#pragma once
#include <hip/hip_ext.h>

#include "mmf_launch.h"
#include "mmf_trace_device.h"
#include "mmf_device.h"

namespace mmf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load with dword alignment
typedef unsigned short ushort_u __attribute__((aligned(1)));             // 2-byte load with byte alignment
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));                 // a channel pair: v_pk_mul_f32 / v_pk_add_f32

// ------------------------------------------------------------------------------------------------
// Shared per-voxel gate (projection, occlusion test against the synthetic depth, bilinear footprint,
// mask).  Same operation order as oracle/mmf_oracle.c:app_gate.
// ------------------------------------------------------------------------------------------------
// geometry + occlusion part: everything that does not depend on the integration mask
template <int FMA = 0>
__device__ inline bool app_gate_geo(const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* __restrict__ synth, int Ws,
                                    int Hs, int bx, int by, int bz, int lin, int& x0, int& y0, float& wx, float& wy) {
  float c[3], p[3], u, v;
  voxel_centre<FMA>(mc, bx, by, bz, lin, c);
  xform<FMA>(T_C_L, c, p);
  if (!project<FMA>(cam, p, u, v)) return false;
  if (mc.max_dist > 0.0f && p[2] > mc.max_dist) return false;
  const float sf = (float)mc.st_sf;
  int sx, sy;
  float swx, swy;
  if (!bilin_setup(u / sf, v / sf, Ws, Hs, sx, sy, swx, swy)) return false;
  if (!bilin_setup(u, v, cam.W, cam.H, x0, y0, wx, wy)) return false;
  const size_t si = (size_t)sy * Ws + sx;
  const float2_u s0 = *reinterpret_cast<const float2_u*>(synth + si);
  const float2_u s1 = *reinterpret_cast<const float2_u*>(synth + si + Ws);
  if (!(s0.x > 0.0f) || !(s0.y > 0.0f) || !(s1.x > 0.0f) || !(s1.y > 0.0f)) return false;
  const float sd = bilin<FMA>(s0.x, s0.y, s1.x, s1.y, swx, swy);
  if (fabsf(sd - p[2]) > mc.trunc) return false;
  return true;
}

// mask part: all four taps of the bilinear footprint must be inside the mask
__device__ inline bool app_gate_mask(const uint8_t* __restrict__ mask, int W, int x0, int y0) {
  if (!mask) return true;
  const size_t i = (size_t)y0 * W + x0;
  const unsigned m0 = *reinterpret_cast<const ushort_u*>(mask + i);
  const unsigned m1 = *reinterpret_cast<const ushort_u*>(mask + i + W);
  return (m0 & 0xffu) && (m0 & 0xff00u) && (m1 & 0xffu) && (m1 & 0xff00u);
}

template <int FMA = 0>
__device__ inline bool app_gate(const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const uint8_t* __restrict__ mask,
                                const float* __restrict__ synth, int Ws, int Hs, int bx, int by, int bz, int lin, int& x0,
                                int& y0, float& wx, float& wy) {
  float c[3], p[3], u, v;
  voxel_centre<FMA>(mc, bx, by, bz, lin, c);
  xform<FMA>(T_C_L, c, p);
  if (!project<FMA>(cam, p, u, v)) return false;
  if (mc.max_dist > 0.0f && p[2] > mc.max_dist) return false;
  const float sf = (float)mc.st_sf;
  int sx, sy;
  float swx, swy;
  // both footprints first, then every load of the gate in one batch (row pairs: 8-byte / 2-byte loads); the
  // accept/reject result is the same as testing them one after the other
  if (!bilin_setup(u / sf, v / sf, Ws, Hs, sx, sy, swx, swy)) return false;
  if (!bilin_setup(u, v, cam.W, cam.H, x0, y0, wx, wy)) return false;
  const size_t si = (size_t)sy * Ws + sx;
  const float2_u s0 = *reinterpret_cast<const float2_u*>(synth + si);
  const float2_u s1 = *reinterpret_cast<const float2_u*>(synth + si + Ws);
  unsigned m0 = 0x0101u, m1 = 0x0101u;
  if (mask) {
    const size_t i = (size_t)y0 * cam.W + x0;
    m0 = *reinterpret_cast<const ushort_u*>(mask + i);
    m1 = *reinterpret_cast<const ushort_u*>(mask + i + cam.W);
  }
  if (!(s0.x > 0.0f) || !(s0.y > 0.0f) || !(s1.x > 0.0f) || !(s1.y > 0.0f)) return false;
  const float sd = bilin<FMA>(s0.x, s0.y, s1.x, s1.y, swx, swy);
  if (fabsf(sd - p[2]) > mc.trunc) return false;
  if (!(m0 & 0xffu) || !(m0 & 0xff00u) || !(m1 & 0xffu) || !(m1 & 0xff00u)) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Colour: voxel = {uchar4 rgb_, float w} (8 B); one workgroup of 256 threads per block, 2 voxels (16 B) per thread.
// ------------------------------------------------------------------------------------------------

// blend one colour voxel {rgb_, w} with the bilinear sample at footprint (x0,y0,wx,wy)
// DIV: the spec switch mmf_params.appearance_blend_division (stand-alone kernels only; every fused launch is built with the default)
template <bool DIV = false, int FMA = 0>
__device__ inline void color_update(const uint8_t* __restrict__ rgb, int W, const MapConsts& mc, int x0, int y0, float wx, float wy,
                                    unsigned& ex, unsigned& ey) {
  // the two pixels of a footprint row are 6 consecutive bytes: one 4-byte + one 2-byte load (byte-aligned) instead of six
  typedef unsigned u32_b __attribute__((aligned(1)));
  typedef unsigned short u16_b __attribute__((aligned(1)));
  const uint8_t* r0 = rgb + ((size_t)y0 * W + x0) * 3;
  const uint8_t* r1 = r0 + (size_t)W * 3;
  const u64 top = (u64)*reinterpret_cast<const u32_b*>(r0) | ((u64)*reinterpret_cast<const u16_b*>(r0 + 4) << 32);
  const u64 bot = (u64)*reinterpret_cast<const u32_b*>(r1) | ((u64)*reinterpret_cast<const u16_b*>(r1 + 4) << 32);
  const float Wv = __uint_as_float(ey);
  const float wm = mc.app_wm;
  const float inv = 1.0f / (Wv + wm);
  unsigned out = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float a = bilin<FMA>((float)((top >> (8 * k)) & 0xffu), (float)((top >> (8 * (k + 3))) & 0xffu), (float)((bot >> (8 * k)) & 0xffu),
                          (float)((bot >> (8 * (k + 3))) & 0xffu), wx, wy);
    const float Aold = (float)((ex >> (8 * k)) & 0xffu);
    const float num = madd2<FMA>(Aold, Wv, a, wm);
    const float An = DIV ? num / (Wv + wm) : num * inv;
    out |= ((unsigned)floorf(An + 0.5f) & 0xffu) << (8 * k);
  }
  ex = out;
  ey = __float_as_uint(fminf(Wv + wm, mc.app_max_w));
}

template <bool DIV = false, int FMA = 0>
__device__ inline void color_body(const AppArgs& A, const MapConsts& mc, const float* __restrict__ synth, int Ws, int Hs, int bid,
                                  int nb) {
  const LayerDev& L = A.L;
  const Cam& cam = A.cam;
  const uint8_t* __restrict__ rgb = reinterpret_cast<const uint8_t*>(A.image);
  const int n = *A.sc.cand_count;
  const int chunk = (n + 7) >> 3;
  for (int j = bid; j < chunk * 8; j += nb) {
    const int i = xcd_candidate(j, chunk);
    if (i >= n) continue;
    const int slot = A.sc.cand_slot[i];
    if (slot < 0) continue;
    const bool is_new = A.sc.cand_new[i] != 0;
    int bx, by, bz;
    unpack_key(A.sc.cand_key[i], bx, by, bz);
    uint4* vox2 = reinterpret_cast<uint4*>(L.pool) + (size_t)slot * (kVPB / 2) + threadIdx.x;
    uint4 e2 = is_new ? make_uint4(0u, 0u, 0u, 0u) : *vox2;
    bool upd = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int lin = threadIdx.x * 2 + r;
      unsigned ex = r ? e2.z : e2.x, ey = r ? e2.w : e2.y;
      int x0, y0;
      float wx, wy;
      if (app_gate<FMA>(mc, cam, A.T_C_L, A.mask, synth, Ws, Hs, bx, by, bz, lin, x0, y0, wx, wy)) {
        color_update<DIV, FMA>(rgb, cam.W, mc, x0, y0, wx, wy, ex, ey);
        upd = true;
      }
      if (r) {
        e2.z = ex;
        e2.w = ey;
      } else {
        e2.x = ex;
        e2.y = ey;
      }
    }
    if (upd || is_new) *vox2 = e2;
  }
}

// ------------------------------------------------------------------------------------------------
// Features: block payload = half[512][C] (+ float w[512] in poolw).  One workgroup (256 threads) per
// candidate block.
//   phase 1: every voxel is gated once (2 voxels per thread); survivors are compacted into LDS
//            (voxel id, top-left tap pixel, bilinear weights, old weight) with a wave ballot;
//            the new weight is written immediately.
//   phase 2: 32 groups of 8 lanes walk the compacted list; a group moves one voxel's channels in
//            128-byte pieces (8 lanes x 16 B): 4 image taps + the voxel row in, blended row out.
//            Image taps are HWC f16, so each tap piece is one contiguous 128 B line.
// HBM-bound: algorithmic traffic = feature image once + (2C+4) B read and written per touched voxel.
// ------------------------------------------------------------------------------------------------
struct FeatLds {
  uint16_t lin[kVPB];
  uint32_t pix[kVPB];
  float wx[kVPB], wy[kVPB], W[kVPB];
  uint8_t valid[kVPB];
  int n, base;
};

// One tap of the virtual up-sampled image: f16( bilinear align_corners=False of the low-res map at pixel (xf,yf) ),
// the arithmetic of k_upsample_features (mmf_kernels_image.hip), 8 channels starting at c0.
struct LowAxis {
  int i0, i1;
  float l0, l1;
};
template <int FMA = 0>
__device__ __forceinline__ LowAxis low_axis(float scale, int out_idx, int n_in) {
  float sv = madd<FMA>(scale, (float)out_idx + 0.5f, -0.5f);
  sv = sv < 0.0f ? 0.0f : sv;
  LowAxis a;
  a.i0 = (int)sv < n_in - 1 ? (int)sv : n_in - 1;
  a.i1 = a.i0 < n_in - 1 ? a.i0 + 1 : a.i0;
  a.l1 = sv - (float)a.i0;
  a.l0 = 1.0f - a.l1;
  return a;
}
struct Low8 {
  float4 lo, hi;
};
__device__ __forceinline__ Low8 low_load8(const float* __restrict__ low, int w, int Cin, int y, int x, int c0) {
  const float4* p = reinterpret_cast<const float4*>(low + ((size_t)y * w + x) * Cin + c0);
  Low8 r;
  r.lo = p[0];
  r.hi = p[1];
  return r;
}
__device__ __forceinline__ Low8 low_ptr8(const float* __restrict__ p) {
  Low8 r;
  r.lo = reinterpret_cast<const float4*>(p)[0];
  r.hi = reinterpret_cast<const float4*>(p)[1];
  return r;
}
__device__ __forceinline__ float f4_at(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ float low_at(const Low8& v, int k) {
  return k == 0 ? v.lo.x : k == 1 ? v.lo.y : k == 2 ? v.lo.z : k == 3 ? v.lo.w : k == 4 ? v.hi.x : k == 5 ? v.hi.y : k == 6 ? v.hi.z : v.hi.w;
}
template <int FMA = 0>
__device__ __forceinline__ half8 low_tap(const Low8& a00, const Low8& a01, const Low8& a10, const Low8& a11, const LowAxis& X,
                                         const LowAxis& Y) {
  half8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float val = madd2<FMA>(Y.l0, madd2<FMA>(X.l0, low_at(a00, k), X.l1, low_at(a01, k)), Y.l1, madd2<FMA>(X.l0, low_at(a10, k), X.l1, low_at(a11, k)));
    o[k] = (_Float16)val;
  }
  return o;
}

// One tap from its own four texels: the rare case of a footprint that straddles a low-res cell border.  A real call, not
// inlined: inlined four times it doubles the register count of every kernel that updates rows from a low-res map (and halves
// the rows in flight per SIMD) for the sake of 1 voxel in ~8.
template <int FMA = 0>
__device__ __noinline__ half8 low_tap_at(const float* __restrict__ low, int w, int Cin, int c0, LowAxis X, LowAxis Y) {
  return low_tap<FMA>(low_load8(low, w, Cin, Y.i0, X.i0, c0), low_load8(low, w, Cin, Y.i0, X.i1, c0), low_load8(low, w, Cin, Y.i1, X.i0, c0),
                 low_load8(low, w, Cin, Y.i1, X.i1, c0), X, Y);
}

// One surviving voxel: blend its channel row with the bilinear sample of the feature image (or of the virtual up-sampled
// low-res map).  `lanes` lanes (gl = 0..lanes-1) share the row in 16-byte pieces.
// packed forms of madd2 (two channels per instruction: v_pk_fma_f32 / v_pk_mul_f32 + v_pk_add_f32)
template <int FMA>
__device__ __forceinline__ f32x2 pk_madd2(float a, f32x2 x, float b, f32x2 y) {
  if constexpr ((FMA & 1) != 0) {
    const f32x2 av = {a, a};
    return __builtin_elementwise_fma(av, x, b * y);
  } else {
    return a * x + b * y;
  }
}
template <int FMA>
__device__ __forceinline__ f32x2 pk_madd2(f32x2 a, float x, f32x2 b, float y) {
  if constexpr ((FMA & 1) != 0) {
    const f32x2 xv = {x, x};
    return __builtin_elementwise_fma(a, xv, b * y);
  } else {
    return a * x + b * y;
  }
}

template <bool LOW, bool DIV = false, int FMA = 0>
__device__ __forceinline__ void feature_voxel(const AppArgs& Aa, const MapConsts& mc, __half* __restrict__ A, bool is_new, size_t pix,
                                              float wx, float wy, float Wv, int gl, int lanes) {
  const Cam& cam = Aa.cam;
  const int C = mc.C, nch = C >> 3;
  const float wm = mc.app_wm;
  const float inv = 1.0f / (Wv + wm);
  // the voxel's piece `ch` (8 channels) blended with the four taps of the footprint
  auto blend = [&](int ch, const half8& a00, const half8& a10, const half8& a01, const half8& a11) {
    half8 av;
    if (is_new) {
#pragma unroll
      for (int k = 0; k < 8; ++k) av[k] = (_Float16)0.0f;
    } else {
      av = *reinterpret_cast<const half8*>(A + ch * 8);
    }
    // two channels per instruction (v_pk_mul_f32 / v_pk_add_f32: the same multiplications and additions in the same order, element
    // by element -- no contraction): this loop is the instruction count of the row update, which is issue-bound as a guest of the
    // sphere-trace launch and, from a low-res map, on its own
    half8 o;
    const float ux = 1.0f - wx, uy = 1.0f - wy;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const f32x2 v00 = {(float)a00[k], (float)a00[k + 1]}, v10 = {(float)a10[k], (float)a10[k + 1]};
      const f32x2 v01 = {(float)a01[k], (float)a01[k + 1]}, v11 = {(float)a11[k], (float)a11[k + 1]};
      f32x2 a;
      if constexpr ((FMA & 2) != 0) {  // mmf_params.bilinear_four_weight_sum: the tree of bilin<FMA> (mmf_device.h), two channels at a time
        const float w00 = ux * uy, w01 = ux * wy, w10 = wx * uy, w11 = wx * wy;
        f32x2 t = pk_madd2<FMA>(w00, v00, w01, v01);
        if constexpr ((FMA & 1) != 0) {
          const f32x2 w10v = {w10, w10}, w11v = {w11, w11};
          t = __builtin_elementwise_fma(w10v, v10, t);
          a = __builtin_elementwise_fma(w11v, v11, t);
        } else {
          t = w10 * v10 + t;
          a = w11 * v11 + t;
        }
      } else {
        const f32x2 top = pk_madd2<FMA>(ux, v00, wx, v10);
        const f32x2 bot = pk_madd2<FMA>(ux, v01, wx, v11);
        a = pk_madd2<FMA>(uy, top, wy, bot);
      }
      const f32x2 old = {(float)av[k], (float)av[k + 1]};
      const f32x2 num = pk_madd2<FMA>(old, Wv, a, wm);
      const f32x2 An = DIV ? num / (Wv + wm) : num * inv;
      o[k] = (_Float16)An.x;
      o[k + 1] = (_Float16)An.y;
    }
    *reinterpret_cast<half8*>(A + ch * 8) = o;
  };
  if constexpr (!LOW) {
    const __half* __restrict__ feat = reinterpret_cast<const __half*>(Aa.image);
    const __half* t00 = feat + pix * C;
    const __half* t10 = t00 + C;
    const __half* t01 = t00 + (size_t)cam.W * C;
    const __half* t11 = t01 + C;
    for (int ch = gl; ch < nch; ch += lanes)
      blend(ch, *reinterpret_cast<const half8*>(t00 + ch * 8), *reinterpret_cast<const half8*>(t10 + ch * 8),
            *reinterpret_cast<const half8*>(t01 + ch * 8), *reinterpret_cast<const half8*>(t11 + ch * 8));
  } else {
    // LOW: the four taps are pixels (px,py) (px+1,py) (px,py+1) (px+1,py+1) of the virtual up-sampled image.  Three loops,
    // none unrolled: the register count of this kernel decides how many rows are in flight per SIMD (it is latency-bound on
    // the L2-resident low-res map), and one loop with every case inside cost 170 VGPRs = 2 waves per SIMD.
    const LowRes LR = Aa.low;
    const int py = (int)(pix / (size_t)cam.W), px = (int)(pix - (size_t)py * cam.W);
    const LowAxis X0 = low_axis<FMA>(LR.sw, px, LR.w), X1 = low_axis<FMA>(LR.sw, px + 1, LR.w);
    const LowAxis Y0 = low_axis<FMA>(LR.sh, py, LR.h), Y1 = low_axis<FMA>(LR.sh, py + 1, LR.h);
    const int nin = LR.cin >> 3 < nch ? LR.cin >> 3 : nch;  // pieces that exist in the map; the rest are the zero pad channels
    const bool one_cell = X0.i0 == X1.i0 && X0.i1 == X1.i1 && Y0.i0 == Y1.i0 && Y0.i1 == Y1.i1;
    if (one_cell) {  // usual case: the footprint lies inside one low-res cell, 4 texels serve 4 taps
      const float* r0 = LR.data + ((size_t)Y0.i0 * LR.w) * LR.cin;
      const float* r1 = LR.data + ((size_t)Y0.i1 * LR.w) * LR.cin;
      const float *p00 = r0 + (size_t)X0.i0 * LR.cin, *p01 = r0 + (size_t)X0.i1 * LR.cin;
      const float *p10 = r1 + (size_t)X0.i0 * LR.cin, *p11 = r1 + (size_t)X0.i1 * LR.cin;
#pragma unroll 1
      for (int ch = gl; ch < nin; ch += lanes) {
        const int c0 = ch * 8;
        half8 t00, t10, t01, t11;
        // four channels at a time: 4 x 16 B of texels live instead of 4 x 32 B (same arithmetic as low_tap, channel by channel)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 q00 = reinterpret_cast<const float4*>(p00 + c0)[h], q01 = reinterpret_cast<const float4*>(p01 + c0)[h];
          const float4 q10 = reinterpret_cast<const float4*>(p10 + c0)[h], q11 = reinterpret_cast<const float4*>(p11 + c0)[h];
#pragma unroll
          for (int k = 0; k < 4; k += 2) {  // channel pairs: packed f32 multiplies / adds, the arithmetic of low_tap element by element
            const f32x2 a = {f4_at(q00, k), f4_at(q00, k + 1)}, b = {f4_at(q01, k), f4_at(q01, k + 1)};
            const f32x2 c = {f4_at(q10, k), f4_at(q10, k + 1)}, d = {f4_at(q11, k), f4_at(q11, k + 1)};
            const f32x2 r0x0 = pk_madd2<FMA>(X0.l0, a, X0.l1, b), r0x1 = pk_madd2<FMA>(X1.l0, a, X1.l1, b);  // upper texel row at the two tap columns
            const f32x2 r1x0 = pk_madd2<FMA>(X0.l0, c, X0.l1, d), r1x1 = pk_madd2<FMA>(X1.l0, c, X1.l1, d);  // lower texel row
            const f32x2 v00 = pk_madd2<FMA>(Y0.l0, r0x0, Y0.l1, r1x0), v10 = pk_madd2<FMA>(Y0.l0, r0x1, Y0.l1, r1x1);
            const f32x2 v01 = pk_madd2<FMA>(Y1.l0, r0x0, Y1.l1, r1x0), v11 = pk_madd2<FMA>(Y1.l0, r0x1, Y1.l1, r1x1);
            // (two channels per conversion: v_cvt_pk_f16_f32, round to nearest even like the scalar cast)
            const half2v c00 = __builtin_convertvector(v00, half2v), c10 = __builtin_convertvector(v10, half2v);
            const half2v c01 = __builtin_convertvector(v01, half2v), c11 = __builtin_convertvector(v11, half2v);
            t00[4 * h + k] = c00.x, t00[4 * h + k + 1] = c00.y;
            t10[4 * h + k] = c10.x, t10[4 * h + k + 1] = c10.y;
            t01[4 * h + k] = c01.x, t01[4 * h + k + 1] = c01.y;
            t11[4 * h + k] = c11.x, t11[4 * h + k + 1] = c11.y;
          }
          if (h == 0) __builtin_amdgcn_sched_barrier(0);
        }
        blend(ch, t00, t10, t01, t11);
      }
    } else {  // the footprint straddles a cell border (1 in ~8 at 16x up-sampling): one tap at a time, each a call
#pragma unroll 1
      for (int ch = gl; ch < nin; ch += lanes) {
        const int c0 = ch * 8;
        const half8 a00 = low_tap_at<FMA>(LR.data, LR.w, LR.cin, c0, X0, Y0), a10 = low_tap_at<FMA>(LR.data, LR.w, LR.cin, c0, X1, Y0),
                    a01 = low_tap_at<FMA>(LR.data, LR.w, LR.cin, c0, X0, Y1), a11 = low_tap_at<FMA>(LR.data, LR.w, LR.cin, c0, X1, Y1);
        blend(ch, a00, a10, a01, a11);
      }
    }
    half8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
#pragma unroll 1
    for (int ch = nin + gl; ch < nch; ch += lanes) blend(ch, z, z, z, z);
  }
}

// Phase 2 inside the workgroup that gated the block: 32 groups of 8 lanes walk the survivor list in LDS.
template <bool LOW, bool DIV = false, int FMA = 0>
__device__ inline void feature_apply(const AppArgs& A, const MapConsts& mc, FeatLds& S, int slot, bool is_new) {
  const int C = mc.C;
  const int group = threadIdx.x >> 3, gl = threadIdx.x & 7;
  const int nv = S.n;
  __half* blk = reinterpret_cast<__half*>(A.L.pool) + (size_t)slot * kVPB * C;
  for (int vi = group; vi < nv; vi += 32)
    feature_voxel<LOW, DIV, FMA>(A, mc, blk + (size_t)S.lin[vi] * C, is_new, S.pix[vi], S.wx[vi], S.wy[vi], S.W[vi], gl, 8);
}

// Phase 2 deferred: append the survivor list to the frame's global list (FlatList); k_feature_flat then spreads the
// rows evenly over the whole chip.  Blocks on a surface have up to 512 survivors, the average is ~50: done inside the
// gating workgroup, the heavy blocks set the kernel time (16 serial rounds vs 1.6 on average).
// Returns false (workgroup-uniform) when the list is not in use; contains a barrier.
__device__ inline bool feature_publish(const AppArgs& A, FeatLds& S, int slot, bool is_new, int cand) {
  if (!A.flat.rec) return false;
  const int nv = S.n;
  // by pool slot, not by candidate position: slots are unique, so a sub-list can never hold more than ceil(cap / 64) blocks
  // (its region's size) whatever the candidate list looks like
  (void)cand;
  const int sub = slot & (kFlatSubLists - 1);
  if (threadIdx.x == 0) S.base = nv ? atomicAdd(A.flat.count + sub * kFlatCountStride, nv) : 0;
  __syncthreads();
  const int base = sub * A.flat.seg_cap + S.base;
  const unsigned hi = ((unsigned)slot << 9) | (is_new ? 0x80000000u : 0u);
  for (int v = threadIdx.x; v < nv; v += 256) {
    A.flat.rec[base + v] = make_uint4(hi | S.lin[v], S.pix[v], __float_as_uint(S.wx[v]), __float_as_uint(S.wy[v]));
    A.flat.w[base + v] = S.W[v];
  }
  return true;
}

// rows of a new block that were not updated must read as zero
__device__ inline void feature_zero_fill(const AppArgs& A, const MapConsts& mc, const FeatLds& S, int slot) {
  const int C = mc.C, nch = C >> 3;
  const int group = threadIdx.x >> 3, gl = threadIdx.x & 7;
  __half* blk = reinterpret_cast<__half*>(A.L.pool) + (size_t)slot * kVPB * C;
  half8 z;
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
  for (int lin = group; lin < kVPB; lin += 32) {
    if (S.valid[lin]) continue;
    __half* A2 = blk + (size_t)lin * C;
    for (int ch = gl; ch < nch; ch += 8) *reinterpret_cast<half8*>(A2 + ch * 8) = z;
  }
}

// tail of both gating bodies once the survivor list of the block is complete in LDS (callers synchronised before)
// PUBLISH_ONLY: the caller guarantees a survivor list (A.flat.rec != nullptr), so the in-workgroup row update is not even
// compiled in -- it is the register-hungriest code of the gating kernels (LOW: 175 VGPRs = 2 waves per SIMD, against 5).
template <bool LOW, bool PUBLISH_ONLY = false, bool DIV = false, int FMA = 0>
__device__ inline void feature_finish(const AppArgs& A, const MapConsts& mc, FeatLds& S, int slot, bool is_new, int cand) {
  if constexpr (PUBLISH_ONLY) {
    feature_publish(A, S, slot, is_new, cand);
  } else {
    // statistics: with the survivor list the frame's total is added once by k_feature_flat (one more same-address atomic per
    // gating workgroup otherwise)
    if (threadIdx.x == 0 && A.stats && S.n && !A.flat.rec) atomicAdd(reinterpret_cast<unsigned long long*>(A.stats + 8), (unsigned long long)S.n);
    if (!feature_publish(A, S, slot, is_new, cand)) feature_apply<LOW, DIV, FMA>(A, mc, S, slot, is_new);
  }
  if (is_new) feature_zero_fill(A, mc, S, slot);
}

// Balanced phase 2: the frame's survivor list, `lpv` lanes per voxel row, any grid size (workgroup bid of nb).
template <bool LOW, int FMA = 0>
__device__ inline void feature_flat_role(const AppArgs& A, const MapConsts& mc, int lpv, int bid, int nb, int* s_prefix) {
  const long long tr0 = wg_trace_begin();
  // prefix sums of the sub-list counters (one wave, shuffles): flat position v lives in sub-list k with prefix[k] <= v < prefix[k+1]
  if (threadIdx.x < 64) {
    const int c = threadIdx.x < kFlatSubLists ? A.flat.count[threadIdx.x * kFlatCountStride] : 0;
    const int incl = wave_incl_scan(c);
    s_prefix[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) s_prefix[0] = 0;
  }
  __syncthreads();
  const int total = s_prefix[kFlatSubLists];
  if (bid == 0 && threadIdx.x == 0) {
    if (A.flat.hint) *A.flat.hint = total;
    if (A.stats && total) atomicAdd(reinterpret_cast<unsigned long long*>(A.stats + 8), (unsigned long long)total);
  }
  const int vpw = 256 / lpv;
  const int group = threadIdx.x / lpv, gl = threadIdx.x % lpv;
  const int C = mc.C;
  __half* pool = reinterpret_cast<__half*>(A.L.pool);
  for (int v = bid * vpw + group; v < total; v += nb * vpw) {
    int k = 0;  // binary search over 64 sub-lists: largest k with prefix[k] <= v
#pragma unroll
    for (int step = kFlatSubLists / 2; step > 0; step >>= 1)
      if (s_prefix[k + step] <= v) k += step;
    const int at = k * A.flat.seg_cap + (v - s_prefix[k]);
    const uint4 r = A.flat.rec[at];
    const float Wv = A.flat.w[at];
    const size_t row = (size_t)(r.x & 0x7fffffffu);  // slot * 512 + lin
    feature_voxel<LOW, false, FMA>(A, mc, pool + row * C, (r.x >> 31) != 0u, r.y, __uint_as_float(r.z), __uint_as_float(r.w), Wv, gl, lpv);
  }
  wg_trace_end(tr0, kTrFeatureFlat);
}

template <bool LOW, bool DIV = false, int FMA = 0>
__device__ inline void feature_body(const AppArgs& A, const MapConsts& mc, const float* __restrict__ synth, int Ws, int Hs,
                                    int bid, int nb, FeatLds& S) {
  const LayerDev& L = A.L;
  const Cam& cam = A.cam;
  const Rigid& T_C_L = A.T_C_L;
  const uint8_t* __restrict__ mask = A.mask;
  const Scratch& sc = A.sc;
  uint16_t* s_lin = S.lin;
  uint32_t* s_pix = S.pix;
  float *s_wx = S.wx, *s_wy = S.wy, *s_W = S.W;
  uint8_t* s_valid = S.valid;
  int& s_n = S.n;

  const int n = *sc.cand_count;
  const int chunk = (n + 7) >> 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const float wm = mc.app_wm;

  for (int j = bid; j < chunk * 8; j += nb) {
    const int i = xcd_candidate(j, chunk);
    if (i >= n) continue;
    const int slot = sc.cand_slot[i];
    if (slot < 0) continue;
    const bool is_new = sc.cand_new[i] != 0;
    int bx, by, bz;
    unpack_key(sc.cand_key[i], bx, by, bz);
    if (tid == 0) s_n = 0;
    __syncthreads();

    float* wts = L.poolw + (size_t)slot * kVPB;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int lin = tid + 256 * r;
      int x0, y0;
      float wx, wy;
      const bool valid = app_gate<FMA>(mc, cam, T_C_L, mask, synth, Ws, Hs, bx, by, bz, lin, x0, y0, wx, wy);
      const u64 bal = __ballot(valid);
      int base = 0;
      if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
      base = __shfl(base, 0, 64);
      if (valid) {
        const float Wold = is_new ? 0.0f : wts[lin];
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        s_lin[pos] = (uint16_t)lin;
        s_pix[pos] = (uint32_t)(y0 * cam.W + x0);
        s_wx[pos] = wx;
        s_wy[pos] = wy;
        s_W[pos] = Wold;
        wts[lin] = fminf(Wold + wm, mc.app_max_w);
      } else if (is_new) {
        wts[lin] = 0.0f;
      }
      if (is_new) s_valid[lin] = valid ? 1 : 0;
    }
    __syncthreads();

    feature_finish<LOW, false, DIV, FMA>(A, mc, S, slot, is_new, i);
    __syncthreads();
  }
}

// Colour and feature update of the candidate blocks of a fused frame with ONE geometric gate per voxel.  In a fused
// frame both layers see the same camera and the same candidate list (the allocation jobs read the same flags), so the
// projection, the two bilinear footprints and the occlusion test against the synthetic depth are evaluated once; only
// the masks (depth mask for colour, eroded feature mask for features) differ.  Voxel order: thread t owns voxels 2t, 2t+1.
template <bool LOW, bool PUB = false, int FMA = 0>
__device__ inline void app_frame_body(const AppArgs& Ac, const AppArgs& Af, const MapConsts& mc, const float* __restrict__ synth,
                                      int Ws, int Hs, int bid, int nb, FeatLds& S) {
  const Cam& cam = Ac.cam;
  const uint8_t* __restrict__ rgb = reinterpret_cast<const uint8_t*>(Ac.image);
  const int n = *Ac.sc.cand_count;
  const int chunk = (n + 7) >> 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const float wm = mc.app_wm;
  for (int j = bid; j < chunk * 8; j += nb) {
    const int i = xcd_candidate(j, chunk);
    if (i >= n) continue;
    const int cslot = Ac.sc.cand_slot[i], fslot = Af.sc.cand_slot[i];
    if (cslot < 0 && fslot < 0) continue;
    const bool c_new = Ac.sc.cand_new[i] != 0, f_new = Af.sc.cand_new[i] != 0;
    int bx, by, bz;
    unpack_key(Ac.sc.cand_key[i], bx, by, bz);
    if (tid == 0) S.n = 0;
    __syncthreads();
    uint4* vox2 = reinterpret_cast<uint4*>(Ac.L.pool) + (size_t)(cslot < 0 ? 0 : cslot) * (kVPB / 2) + tid;
    uint4 e2 = make_uint4(0u, 0u, 0u, 0u);
    if (cslot >= 0 && !c_new) e2 = *vox2;
    float* wts = Af.L.poolw + (size_t)(fslot < 0 ? 0 : fslot) * kVPB;
    float2 w2 = make_float2(0.0f, 0.0f);
    if (fslot >= 0 && !f_new) w2 = *reinterpret_cast<const float2*>(wts + 2 * tid);
    bool c_upd = false, f_upd = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int lin = tid * 2 + r;
      int x0 = 0, y0 = 0;
      float wx = 0.0f, wy = 0.0f;
      const bool geo = app_gate_geo<FMA>(mc, cam, Ac.T_C_L, synth, Ws, Hs, bx, by, bz, lin, x0, y0, wx, wy);
      const bool c_ok = geo && cslot >= 0 && app_gate_mask(Ac.mask, cam.W, x0, y0);
      const bool f_ok = geo && fslot >= 0 && app_gate_mask(Af.mask, cam.W, x0, y0);
      if (c_ok) {
        unsigned ex = r ? e2.z : e2.x, ey = r ? e2.w : e2.y;
        color_update<false, FMA>(rgb, cam.W, mc, x0, y0, wx, wy, ex, ey);
        if (r) {
          e2.z = ex;
          e2.w = ey;
        } else {
          e2.x = ex;
          e2.y = ey;
        }
        c_upd = true;
      }
      const u64 bal = __ballot(f_ok);
      int base = 0;
      if (lane == 0 && bal) base = atomicAdd(&S.n, __popcll(bal));
      base = __shfl(base, 0, 64);
      if (f_ok) {
        const float Wold = r ? w2.y : w2.x;
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        S.lin[pos] = (uint16_t)lin;
        S.pix[pos] = (uint32_t)(y0 * cam.W + x0);
        S.wx[pos] = wx;
        S.wy[pos] = wy;
        S.W[pos] = Wold;
        const float Wn = fminf(Wold + wm, mc.app_max_w);
        if (r) w2.y = Wn; else w2.x = Wn;
        f_upd = true;
      }
      if (f_new) S.valid[lin] = f_ok ? 1 : 0;
    }
    if (cslot >= 0 && (c_upd || c_new)) *vox2 = e2;
    if (fslot >= 0 && (f_upd || f_new)) *reinterpret_cast<float2*>(wts + 2 * tid) = w2;
    __syncthreads();
    if (fslot >= 0) feature_finish<LOW, PUB, false, FMA>(Af, mc, S, fslot, f_new, i);
    __syncthreads();
  }
}

}  // namespace mmf
