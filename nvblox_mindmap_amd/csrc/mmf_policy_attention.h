// mmf_policy_attention.h -- the attention tile of the diffusion head's inference path as a device function (kernels:
// mmf_kernels_policy_mfma.hip k_attention_heads; mmf_kernels_policy_layer.hip k_cross_layer).  See the notes there.
#pragma once
#include "mmf_device.h"
#include "mmf_trace_device.h"

namespace mmf {
namespace att {

constexpr int kH = 8, kDH = 15, kD = 120;  // the policy's heads / head dim / embedding dim (the kernel is built for these)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

__device__ __forceinline__ f32x4 mfma16(h4 a, h4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

struct HiLo {  // four values as {4 hi | 4 lo} halves: value = hi + lo / 2048
  h4 hi, lo;
};
struct U2 {
  uint32_t a, b;
};
__device__ __forceinline__ HiLo load_hilo(const float* __restrict__ p) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  HiLo v;
  v.hi = __builtin_bit_cast(h4, U2{r.x, r.y});
  v.lo = __builtin_bit_cast(h4, U2{r.z, r.w});
  return v;
}
__device__ __forceinline__ HiLo split4(const float (&p)[4]) {
  const hp2 h01 = __builtin_amdgcn_cvt_pkrtz(p[0], p[1]), h23 = __builtin_amdgcn_cvt_pkrtz(p[2], p[3]);
  const hp2 l01 = __builtin_amdgcn_cvt_pkrtz((p[0] - (float)h01[0]) * kLoScale, (p[1] - (float)h01[1]) * kLoScale);
  const hp2 l23 = __builtin_amdgcn_cvt_pkrtz((p[2] - (float)h23[0]) * kLoScale, (p[3] - (float)h23[1]) * kLoScale);
  HiLo v;
  v.hi = __builtin_bit_cast(h4, U2{__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23)});
  v.lo = __builtin_bit_cast(h4, U2{__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23)});
  return v;
}


#ifndef MMF_PT
#define MMF_PT(base, i, dep)
#define MMF_ATT_OWN_PT
#endif
constexpr int kPtAtt = 256;

// ---- attention over head-major operands -------------------------------------------------------------------------------------
// CH = key tiles a wave scores before it runs the softmax update (4 CH score registers + 8 CH operand registers)

// NW waves per workgroup share the keys of one (query tile, head): 4 for self-attention over a few hundred keys, 16 when a
// handful of query rows attend to thousands of keys (the trajectory tokens over the full context).
// SPLIT > 1 (one query tile per batch element, i.e. Lq <= 16): the keys are additionally divided among SPLIT WORKGROUPS, each
// of which leaves an un-normalised partial result {O^T [16 channels][16 rows], running maximum [16], sum [16]} in `out`
// ([B, H, SPLIT, 18, 16] floats) -- k_out_ffn_mfma merges them while it loads its input tile.  A cross-workgroup merge inside
// this kernel would need a device-scope release per workgroup; the kernel boundary that follows anyway is free.
constexpr int kPartRows = 18;  // rows of one partial: 16 channels of O^T, then the maxima, then the sums

struct AttLds {
  float sM[16][16], sL[16][16];
  float sO[16][16][17];
};

// One (query tile, head[, key split]) of a batch element by NW waves of a workgroup (its waves wave_base .. wave_base + NW - 1;
// every wave of the workgroup must run a unit: the merge has a workgroup barrier).  SPLIT > 1: partial {O^T, maxima, sums} to
// `out` ([B, H, SPLIT, 18, 16] floats); SPLIT == 1: the output rows to `out` [B, Lq, D].  `tagged` != null: to the same positions
// of a buffer of self-validating 64-bit words {tag | f32 bits} instead, which a workgroup of the SAME launch polls (relaxed
// agent-scope stores, no fence: k_cross_layer, k_self_layer).
template <int NW, int CH, int SPLIT>
__device__ __forceinline__ void attention_body(const float* __restrict__ Qp, const float* __restrict__ Kp, const float* __restrict__ Vt,
                                               const uint8_t* __restrict__ pad, float* __restrict__ out, unsigned long long* tagged,
                                               unsigned tag, int Lq, int Lq16, int Lk, int Lk16, float scale, int split, int q0, int h,
                                               int b, AttLds& LD, int wave_base = 0) {
  float (&sM)[16][16] = LD.sM;
  float (&sL)[16][16] = LD.sL;
  float (&sO)[16][16][17] = LD.sO;
  const int lane = threadIdx.x & 63, w = (int)(threadIdx.x >> 6) - wave_base, j = lane & 15, s = lane >> 4;  // (wave_base: the unit's first wave)
  const size_t bh = (size_t)b * kH + h;

  MMF_PT(kPtAtt, 0, 0.0f);
  const HiLo q = load_hilo(Qp + (bh * Lq16 + q0 + j) * 16 + 4 * s);
  const float* Kb = Kp + bh * Lk16 * 16;
  const float* Vb = Vt + (bh * 16 + j) * Lk16;
  // key padding: [B, Lk16] bytes (1 = ignore; keys >= Lk are marked too), one aligned 32-bit word per (tile, lane)
  const uint8_t* pb = pad ? pad + (size_t)b * Lk16 : nullptr;

  float m_run = -INFINITY, l_run = 0.0f;
  // O^T: rows = channel 4 s + r, column = query row j; main and cross-term accumulators, two sets alternating between tiles
  f32x4 Om0 = {0.f, 0.f, 0.f, 0.f}, Ox0 = Om0, Om1 = Om0, Ox1 = Om0;
  const int all_tiles = Lk16 / 16, per_split = (all_tiles + SPLIT - 1) / SPLIT;
  const int t_begin = split * per_split, ntiles = min(t_begin + per_split, all_tiles);  // this workgroup's key tiles
  for (int tb = t_begin + w; tb < ntiles; tb += NW * CH) {  // this wave's tiles: tb, tb + NW, ...
    HiLo kv[CH], vv[CH];
    uint32_t pw[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int t = tb + NW * i, tc = min(t, all_tiles - 1);  // tiles beyond the end: a valid tile's data, every key marked dead
      kv[i] = load_hilo(Kb + ((size_t)tc * 16 + j) * 16 + 4 * s);
      vv[i] = load_hilo(Vb + tc * 16 + 4 * s);
      uint32_t word = 0u;
      if (pb) word = *reinterpret_cast<const uint32_t*>(pb + tc * 16 + 4 * s);
      pw[i] = t < ntiles ? word : 0xffffffffu;
    }
    MMF_PT(kPtAtt, 1, (float)kv[CH - 1].hi[0] + (float)vv[CH - 1].hi[0]);
    f32x4 S[CH];
    float cmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 cm = mfma16(kv[i].hi, q.hi, z);
      f32x4 cx = mfma16(kv[i].hi, q.lo, z);
      cx = mfma16(kv[i].lo, q.hi, cx);
      f32x4 c;
      const int t = tb + NW * i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c[r] = (cm[r] + cx[r] * kLoInv) * scale;
        const int key = t * 16 + 4 * s + r;
        const bool dead = key >= Lk || ((pw[i] >> (8 * r)) & 0xffu) != 0u;
        c[r] = dead ? -INFINITY : c[r];
        cmax = fmaxf(cmax, c[r]);
      }
      S[i] = c;
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
    MMF_PT(kPtAtt, 2, cmax);
    const float m_new = fmaxf(m_run, cmax);
    const float corr = (m_run == -INFINITY) ? 0.0f : __expf(m_run - m_new);
    l_run *= corr;
    Om0 *= corr;
    Ox0 *= corr;
    Om1 *= corr;
    Ox1 *= corr;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = (S[i][r] == -INFINITY) ? 0.0f : __expf(S[i][r] - m_new);
        l_run += p[r];
      }
      const HiLo P = split4(p);
      if (i & 1) {
        Om1 = mfma16(vv[i].hi, P.hi, Om1);
        Ox1 = mfma16(vv[i].hi, P.lo, Ox1);
        Ox1 = mfma16(vv[i].lo, P.hi, Ox1);
      } else {
        Om0 = mfma16(vv[i].hi, P.hi, Om0);
        Ox0 = mfma16(vv[i].hi, P.lo, Ox0);
        Ox0 = mfma16(vv[i].lo, P.hi, Ox0);
      }
    }
    m_run = m_new;
  }
  f32x4 O0, O1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    O0[r] = Om0[r] + Ox0[r] * kLoInv;
    O1[r] = Om1[r] + Ox1[r] * kLoInv;
  }
  MMF_PT(kPtAtt, 3, O0[0] + O1[0]);
  // merge the four key ranges
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const f32x4 O = O0 + O1;
  if (s == 0) {
    sM[w][j] = m_run;
    sL[w][j] = l_run;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sO[w][4 * s + r][j] = O[r];
  __syncthreads();
  MMF_PT(kPtAtt, 4, 0.0f);
  if (w == 0) {
    float M = sM[0][j];
#pragma unroll
    for (int u = 1; u < NW; ++u) M = fmaxf(M, sM[u][j]);
    float f[NW], l = 0.0f;
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      f[u] = (sM[u][j] == -INFINITY) ? 0.0f : __expf(sM[u][j] - M);
      l += f[u] * sL[u][j];
    }
    const int row = q0 + j;
    float* part = out + (((size_t)b * kH + h) * SPLIT + split) * kPartRows * 16;  // SPLIT > 1 only
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = 4 * s + r;
      float o = 0.0f;
#pragma unroll
      for (int u = 0; u < NW; ++u) o += f[u] * sO[u][ch][j];
      if (SPLIT > 1) {
        if (tagged) {
          if (j < Lq)  // (the live query rows only: the consumer polls exactly these)
            __hip_atomic_store(tagged + ((((size_t)b * kH + h) * SPLIT + split) * kPartRows + ch) * 16 + j,
                               ((unsigned long long)tag << 32) | __float_as_uint(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
          part[ch * 16 + j] = o;
      } else if (ch < kDH && row < Lq) {
        const size_t at = ((size_t)b * Lq + row) * kD + h * kDH + ch;
        if (tagged)  // (SPLIT == 1: the attention output itself, handed to a block workgroup of the same launch)
          __hip_atomic_store(tagged + at, ((unsigned long long)tag << 32) | __float_as_uint(o / l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          out[at] = o / l;
      }
    }
    if (SPLIT > 1 && s == 0) {
      if (tagged && j >= Lq) {
      } else if (tagged) {
        unsigned long long* tp = tagged + ((((size_t)b * kH + h) * SPLIT + split) * kPartRows + 16) * 16 + j;
        __hip_atomic_store(tp, ((unsigned long long)tag << 32) | __float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tp + 16, ((unsigned long long)tag << 32) | __float_as_uint(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        part[16 * 16 + j] = M;
        part[17 * 16 + j] = l;
      }
    }
  }
  MMF_PT(kPtAtt, 5, 0.0f);
}


#ifdef MMF_ATT_OWN_PT
#undef MMF_PT
#undef MMF_ATT_OWN_PT
#endif

}  // namespace att
}  // namespace mmf
