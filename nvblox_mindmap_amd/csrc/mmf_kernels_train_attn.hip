// mmf_kernels_train_attn.hip -- forward AND backward of the policy's small-head attention for the TRAINING step.  gfx950 / wave64.
//
// The trainable transformer stacks of the diffusion head (mindmap/diffuser_actor/diffusion_head.py: self-attention over the 616
// trajectory + context tokens, 8 heads of 15 channels, key-padding mask) run 8 such attentions per step.  torch's SDPA has no
// kernel shaped for a 15-channel head: padded to 16 it takes the memory-efficient path, 0.21 ms forward + 0.48 ms backward per
// layer at batch 32 (5.5 ms of a 51 ms step) for 0.4 GFLOP-scale products, plus the pad / transpose copies around it.
//
// Here: float32 throughout on v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation: the accuracy of the f32 reference),
// flash style, operands read straight from the [B, L, heads * head_dim] projections (no head transpose, no channel padding in
// memory: channel 15 is a zero in registers / LDS).
//   * a wave owns 16 rows of the accumulator's COLUMN side: 16 queries (forward, dQ) or 16 keys (dK / dV); the other side is
//     streamed through LDS in chunks of 128 rows.  Every product is arranged so that a lane's accumulator column is "its" query
//     (key): S^T = K Q^T, dP^T = V dO^T, O^T = V^T P^T, dQ^T = K^T dS^T  |  S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS.
//     The accumulator of the first product of a chain (lane (col, grp) holds rows 4 grp + r) IS the B operand of the next one
//     (k index = grp): no data movement between the products, softmax statistics are lane-local + two cross-lane steps.
//   * the MFMA's k index is free to mean any channel as long as A and B agree: k = grp at step s means channel 4 grp + s, so an
//     operand row is ONE 16-byte LDS read (or four registers loaded once).
//   * backward recomputes P from the saved log-sum-exp (base 2: the logits carry log2 e); D = rowsum(dO o O) is a by-product of
//     the dQ kernel, read by the dK / dV kernel behind it.
// MFMA instructions per 16 x 16 tile of (query, key) pairs: forward 8, dQ 12, dK / dV 16.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mmf_launch.h"

namespace mmf {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef MMF_TA_CHUNK
#define MMF_TA_CHUNK 128
#endif
#ifndef MMF_TA_WAVES
#define MMF_TA_WAVES 8
#endif
constexpr int kC = MMF_TA_CHUNK;   // rows per LDS chunk (tuning: -DMMF_TA_CHUNK=...)
constexpr int kNW = MMF_TA_WAVES;  // waves per workgroup: kNW x 16 accumulator columns
constexpr int kNT = 64 * kNW;
constexpr int kRS = 20;       // floats per row of a row-major chunk (16 + 4: rows stay 16-byte aligned, bank spread)
constexpr int kTS = kC + 4;   // floats per row of a transposed chunk
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

struct AttnArgs {
  const float *q, *k, *v;          // rows of head h at base + b * batch_stride + l * row_stride + h * hd
  long long q_rs, q_bs, k_rs, k_bs, v_rs, v_bs;
  const uint8_t* pad;              // [B, Lk], != 0: the key is ignored (may be null)
  int B, H, Lq, Lk, hd;
  float scale;
};

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// The four lanes that share an accumulator column are l, l ^ 16, l ^ 32, l ^ 48: v_permlane16_swap exchanges the odd row of a
// row pair with the even row of the other operand, v_permlane32_swap the wave's halves (gfx950; vector unit, no LDS round trip).
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xsum(float x) {
  u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float xmax(float x) {
  u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
// v_exp_f32 as it is (no denormal-range rescue: a probability below 2^-126 is a zero here)
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// A chunk = rows [r0, r0 + kC) of a [*, L, H hd] tensor (head h), brought into LDS in two steps so that the global loads of
// chunk i + 1 are in flight while chunk i is being multiplied: load_rows (registers; rows beyond n_rows and channels beyond hd
// are zeros, `mul` scales), store_rows (row-major and / or transposed).  Sixteen lanes take one row's channels (consecutive
// addresses).
constexpr int kPer = kC * 16 / kNT;  // elements per thread and chunk
struct RowRegs {
  float x[kPer];
};
__device__ __forceinline__ void load_rows(RowRegs& R, const float* __restrict__ base, long long rs, int r0, int n_rows, int hd, float mul) {
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int e = (int)threadIdx.x + i * kNT, row = e >> 4, c = e & 15;
    R.x[i] = (r0 + row < n_rows && c < hd) ? base[(long long)(r0 + row) * rs + c] * mul : 0.0f;
  }
}
__device__ __forceinline__ void store_rows(const RowRegs& R, float (*rm)[kRS], float (*tr)[kTS]) {
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int e = (int)threadIdx.x + i * kNT, row = e >> 4, c = e & 15;
    if (rm) rm[row][c] = R.x[i];
    if (tr) tr[c][row] = R.x[i];
  }
}
// one value per chunk row (bias / lse / D): threads 0 .. kC - 1
__device__ __forceinline__ float key_bias(const AttnArgs& A, int b, int kk) {
  return (kk < A.Lk && !(A.pad && A.pad[(long long)b * A.Lk + kk])) ? 0.0f : -INFINITY;
}

// ---- forward --------------------------------------------------------------------------------------------------------------------
// grid: B H ceil(Lq / 128) workgroups of 8 waves; wave w of query tile t owns queries 128 t + 16 w + (0..15).
// FEW (Lq <= 16: the cross-attention of a few query tokens over a long memory): one workgroup per (batch, head), every wave holds
// the SAME 16 query columns and takes every kNW-th 16-key tile of a chunk; the waves' (maximum, sum, accumulator) are merged through
// LDS at the end.
template <bool FEW>
__global__ __launch_bounds__(kNT) void k_tattn_fwd(AttnArgs A, float* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float sK[kC][kRS];
  __shared__ __attribute__((aligned(16))) float sVt[16][kTS];
  __shared__ __attribute__((aligned(16))) float sBias[kC];
  const int nqt = FEW ? 1 : (A.Lq + 16 * kNW - 1) / (16 * kNW);
  const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, b = bh / A.H, h = bh % A.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, grp = lane >> 4;
  const int qi = FEW ? col : qt * 16 * kNW + wave * 16 + col;
  const bool qok = qi < A.Lq;
  const int D = A.H * A.hd;
  float qreg[4];
  {
    const float* qp = A.q + (long long)b * A.q_bs + (long long)(qok ? qi : 0) * A.q_rs + h * A.hd;
    const float m = A.scale * kLog2e;
#pragma unroll
    for (int s = 0; s < 4; ++s) qreg[s] = (qok && 4 * grp + s < A.hd) ? qp[4 * grp + s] * m : 0.0f;
  }
  const float* kb = A.k + (long long)b * A.k_bs + h * A.hd;
  const float* vb = A.v + (long long)b * A.v_bs + h * A.hd;
  float m_run = -INFINITY, l_run = 0.0f;
  f4 o = {0.0f, 0.0f, 0.0f, 0.0f};
  RowRegs rk, rv;
  float rbias = 0.0f;
  load_rows(rk, kb, A.k_rs, 0, A.Lk, A.hd, 1.0f);
  load_rows(rv, vb, A.v_rs, 0, A.Lk, A.hd, 1.0f);
  if (threadIdx.x < kC) rbias = key_bias(A, b, (int)threadIdx.x);
  for (int k0 = 0; k0 < A.Lk; k0 += kC) {
    __syncthreads();
    store_rows(rk, sK, nullptr);
    store_rows(rv, nullptr, sVt);
    if (threadIdx.x < kC) sBias[threadIdx.x] = rbias;
    __syncthreads();
    if (k0 + kC < A.Lk) {  // the next chunk's loads fly during this chunk's products
      load_rows(rk, kb, A.k_rs, k0 + kC, A.Lk, A.hd, 1.0f);
      load_rows(rv, vb, A.v_rs, k0 + kC, A.Lk, A.hd, 1.0f);
      if (threadIdx.x < kC) rbias = key_bias(A, b, k0 + kC + (int)threadIdx.x);
    }
    const int nk = min(kC, A.Lk - k0);
    constexpr int NTL = FEW ? 1 : 4;  // tiles of 16 keys per softmax update
    for (int t0 = FEW ? 16 * wave : 0; t0 < nk; t0 += FEW ? 16 * kNW : 64) {
      f4 s[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const f4 ka = *reinterpret_cast<const f4*>(&sK[t0 + 16 * t + col][4 * grp]);
        f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        acc = mfma4(ka.x, qreg[0], acc);
        acc = mfma4(ka.y, qreg[1], acc);
        acc = mfma4(ka.z, qreg[2], acc);
        acc = mfma4(ka.w, qreg[3], acc);
        s[t] = acc + *reinterpret_cast<const f4*>(&sBias[t0 + 16 * t + 4 * grp]);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < NTL; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t].x, s[t].y), fmaxf(s[t].z, s[t].w)));
      mx = xmax(mx);
      const float m_new = fmaxf(m_run, mx);
      const float m_use = m_new == -INFINITY ? 0.0f : m_new;  // every key so far is masked: all p are exp2(-inf) = 0
      const float alpha = ex2(m_run - m_use);
      m_run = m_new;
      float ps = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        s[t].x = ex2(s[t].x - m_use);
        s[t].y = ex2(s[t].y - m_use);
        s[t].z = ex2(s[t].z - m_use);
        s[t].w = ex2(s[t].w - m_use);
        ps += (s[t].x + s[t].y) + (s[t].z + s[t].w);
      }
      l_run = l_run * alpha + ps;
      o *= alpha;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const f4 va = *reinterpret_cast<const f4*>(&sVt[col][t0 + 16 * t + 4 * grp]);
        o = mfma4(va.x, s[t].x, o);
        o = mfma4(va.y, s[t].y, o);
        o = mfma4(va.z, s[t].z, o);
        o = mfma4(va.w, s[t].w, o);
      }
    }
  }
  float l_tot = xsum(l_run);
  if (FEW) {  // merge the waves' partial softmax states (all over the same queries, disjoint keys)
    __shared__ float s_m[kNW][16], s_l[kNW][16];
    __shared__ __attribute__((aligned(16))) float s_o[kNW][64][4];
    if (grp == 0) s_m[wave][col] = m_run, s_l[wave][col] = l_tot;
    *reinterpret_cast<f4*>(&s_o[wave][lane][0]) = o;
    __syncthreads();
    if (wave != 0) return;
    float m_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < kNW; ++w) m_all = fmaxf(m_all, s_m[w][col]);
    const float m_use = m_all == -INFINITY ? 0.0f : m_all;
    l_tot = 0.0f;
    o = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < kNW; ++w) {
      const float f = ex2(s_m[w][col] - m_use);
      l_tot += s_l[w][col] * f;
      o += *reinterpret_cast<const f4*>(&s_o[w][lane][0]) * f;
    }
    m_run = m_all;
  }
  const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (qok) {
    float* op = out + ((long long)b * A.Lq + qi) * D + h * A.hd;
    const float ov[4] = {o.x * inv, o.y * inv, o.z * inv, o.w * inv};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * grp + r < A.hd) op[4 * grp + r] = ov[r];
    if (grp == 0) lse[((long long)b * A.H + h) * A.Lq + qi] = l_tot > 0.0f ? m_run + log2f(l_tot) : INFINITY;  // (base 2)
  }
}

// ---- backward: dQ ---------------------------------------------------------------------------------------------------------------
// Also writes D[b, h, q] = sum_c dO[q, c] O[q, c] (dsum) for the dK / dV kernel that follows it on the stream.
template <bool FEW>
__global__ __launch_bounds__(kNT) void k_tattn_dq(AttnArgs A, const float* __restrict__ out, const float* __restrict__ dout,
                                                  const float* __restrict__ lse, float* __restrict__ dq, float* __restrict__ dsum_out) {
  __shared__ __attribute__((aligned(16))) float sK[kC][kRS];
  __shared__ __attribute__((aligned(16))) float sKt[16][kTS];
  __shared__ __attribute__((aligned(16))) float sV[kC][kRS];
  __shared__ __attribute__((aligned(16))) float sBias[kC];
  const int nqt = FEW ? 1 : (A.Lq + 16 * kNW - 1) / (16 * kNW);
  const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, b = bh / A.H, h = bh % A.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, grp = lane >> 4;
  const int qi = FEW ? col : qt * 16 * kNW + wave * 16 + col;
  const bool qok = qi < A.Lq;
  const int D = A.H * A.hd;
  float qreg[4], doreg[4], dsum = 0.0f;
  {
    const float* qp = A.q + (long long)b * A.q_bs + (long long)(qok ? qi : 0) * A.q_rs + h * A.hd;
    const long long orow = ((long long)b * A.Lq + (qok ? qi : 0)) * D + h * A.hd;
    const float m = A.scale * kLog2e;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = qok && 4 * grp + s < A.hd;
      qreg[s] = ok ? qp[4 * grp + s] * m : 0.0f;
      doreg[s] = ok ? dout[orow + 4 * grp + s] : 0.0f;
      dsum += ok ? doreg[s] * out[orow + 4 * grp + s] : 0.0f;
    }
  }
  dsum = xsum(dsum);  // D[q] = sum_c dO[q, c] O[q, c]
  if (qok && grp == 0 && (!FEW || wave == 0)) dsum_out[((long long)b * A.H + h) * A.Lq + qi] = dsum;
  const float my_lse = qok ? lse[((long long)b * A.H + h) * A.Lq + qi] : INFINITY;
  const float* kb = A.k + (long long)b * A.k_bs + h * A.hd;
  const float* vb = A.v + (long long)b * A.v_bs + h * A.hd;
  f4 acc_q = {0.0f, 0.0f, 0.0f, 0.0f};
  RowRegs rk, rv;
  float rbias = 0.0f;
  load_rows(rk, kb, A.k_rs, 0, A.Lk, A.hd, 1.0f);
  load_rows(rv, vb, A.v_rs, 0, A.Lk, A.hd, 1.0f);
  if (threadIdx.x < kC) rbias = key_bias(A, b, (int)threadIdx.x);
  for (int k0 = 0; k0 < A.Lk; k0 += kC) {
    __syncthreads();
    store_rows(rk, sK, sKt);
    store_rows(rv, sV, nullptr);
    if (threadIdx.x < kC) sBias[threadIdx.x] = rbias;
    __syncthreads();
    if (k0 + kC < A.Lk) {
      load_rows(rk, kb, A.k_rs, k0 + kC, A.Lk, A.hd, 1.0f);
      load_rows(rv, vb, A.v_rs, k0 + kC, A.Lk, A.hd, 1.0f);
      if (threadIdx.x < kC) rbias = key_bias(A, b, k0 + kC + (int)threadIdx.x);
    }
    const int nk = min(kC, A.Lk - k0);
    for (int t0 = FEW ? 16 * wave : 0; t0 < nk; t0 += FEW ? 16 * kNW : 16) {
      const f4 ka = *reinterpret_cast<const f4*>(&sK[t0 + col][4 * grp]);
      const f4 va = *reinterpret_cast<const f4*>(&sV[t0 + col][4 * grp]);
      f4 s = {0.0f, 0.0f, 0.0f, 0.0f}, dp = {0.0f, 0.0f, 0.0f, 0.0f};
      s = mfma4(ka.x, qreg[0], s);
      dp = mfma4(va.x, doreg[0], dp);
      s = mfma4(ka.y, qreg[1], s);
      dp = mfma4(va.y, doreg[1], dp);
      s = mfma4(ka.z, qreg[2], s);
      dp = mfma4(va.z, doreg[2], dp);
      s = mfma4(ka.w, qreg[3], s);
      dp = mfma4(va.w, doreg[3], dp);
      s += *reinterpret_cast<const f4*>(&sBias[t0 + 4 * grp]);
      f4 ds;
      ds.x = ex2(s.x - my_lse) * (dp.x - dsum);
      ds.y = ex2(s.y - my_lse) * (dp.y - dsum);
      ds.z = ex2(s.z - my_lse) * (dp.z - dsum);
      ds.w = ex2(s.w - my_lse) * (dp.w - dsum);
      const f4 kt = *reinterpret_cast<const f4*>(&sKt[col][t0 + 4 * grp]);
      acc_q = mfma4(kt.x, ds.x, acc_q);
      acc_q = mfma4(kt.y, ds.y, acc_q);
      acc_q = mfma4(kt.z, ds.z, acc_q);
      acc_q = mfma4(kt.w, ds.w, acc_q);
    }
  }
  if (FEW) {  // the waves hold partial sums over disjoint keys: add them in a fixed order
    __shared__ __attribute__((aligned(16))) float s_q[kNW][64][4];
    *reinterpret_cast<f4*>(&s_q[wave][lane][0]) = acc_q;
    __syncthreads();
    if (wave != 0) return;
    acc_q = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < kNW; ++w) acc_q += *reinterpret_cast<const f4*>(&s_q[w][lane][0]);
  }
  if (qok) {
    float* dp_ = dq + ((long long)b * A.Lq + qi) * D + h * A.hd;
    const float dv[4] = {acc_q.x * A.scale, acc_q.y * A.scale, acc_q.z * A.scale, acc_q.w * A.scale};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * grp + r < A.hd) dp_[4 * grp + r] = dv[r];
  }
}

// ---- backward: dK, dV -----------------------------------------------------------------------------------------------------------
// grid: B H ceil(Lk / 128); a wave owns 16 keys, the queries stream through LDS.
__global__ __launch_bounds__(kNT) void k_tattn_dkv(AttnArgs A, const float* __restrict__ dout, const float* __restrict__ lse,
                                                   const float* __restrict__ dsum, float* __restrict__ dk, float* __restrict__ dv) {
  __shared__ __attribute__((aligned(16))) float sQ[kC][kRS];
  __shared__ __attribute__((aligned(16))) float sQt[16][kTS];
  __shared__ __attribute__((aligned(16))) float sDo[kC][kRS];
  __shared__ __attribute__((aligned(16))) float sDot[16][kTS];
  __shared__ __attribute__((aligned(16))) float sLse[kC];
  __shared__ __attribute__((aligned(16))) float sD[kC];
  const int nkt = (A.Lk + 16 * kNW - 1) / (16 * kNW);
  const int kt = blockIdx.x % nkt, bh = blockIdx.x / nkt, b = bh / A.H, h = bh % A.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, grp = lane >> 4;
  const int ki = kt * 16 * kNW + wave * 16 + col;
  const bool kok = ki < A.Lk;
  const int D = A.H * A.hd;
  float kreg[4], vreg[4];
  {
    const float* kp = A.k + (long long)b * A.k_bs + (long long)(kok ? ki : 0) * A.k_rs + h * A.hd;
    const float* vp = A.v + (long long)b * A.v_bs + (long long)(kok ? ki : 0) * A.v_rs + h * A.hd;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = kok && 4 * grp + s < A.hd;
      kreg[s] = ok ? kp[4 * grp + s] : 0.0f;
      vreg[s] = ok ? vp[4 * grp + s] : 0.0f;
    }
  }
  const float kbias = (kok && !(A.pad && A.pad[(long long)b * A.Lk + ki])) ? 0.0f : -INFINITY;
  const float* qb = A.q + (long long)b * A.q_bs + h * A.hd;
  const float* dob = dout + (long long)b * A.Lq * D + h * A.hd;
  const float* lb = lse + ((long long)b * A.H + h) * A.Lq;
  const float* db = dsum + ((long long)b * A.H + h) * A.Lq;
  f4 acc_k = {0.0f, 0.0f, 0.0f, 0.0f}, acc_v = {0.0f, 0.0f, 0.0f, 0.0f};
  RowRegs rq, rg;
  float rl = INFINITY, rd = 0.0f;
  const float qmul = A.scale * kLog2e;
  load_rows(rq, qb, A.q_rs, 0, A.Lq, A.hd, qmul);
  load_rows(rg, dob, (long long)D, 0, A.Lq, A.hd, 1.0f);
  if (threadIdx.x < kC && (int)threadIdx.x < A.Lq) rl = lb[threadIdx.x], rd = db[threadIdx.x];
  for (int q0 = 0; q0 < A.Lq; q0 += kC) {
    __syncthreads();
    store_rows(rq, sQ, sQt);
    store_rows(rg, sDo, sDot);
    if (threadIdx.x < kC) sLse[threadIdx.x] = rl, sD[threadIdx.x] = rd;  // rows beyond Lq: lse = inf, p = exp2(s - inf) = 0
    __syncthreads();
    if (q0 + kC < A.Lq) {
      load_rows(rq, qb, A.q_rs, q0 + kC, A.Lq, A.hd, qmul);
      load_rows(rg, dob, (long long)D, q0 + kC, A.Lq, A.hd, 1.0f);
      rl = INFINITY, rd = 0.0f;
      const int qq = q0 + kC + (int)threadIdx.x;
      if (threadIdx.x < kC && qq < A.Lq) rl = lb[qq], rd = db[qq];
    }
    const int nq = min(kC, A.Lq - q0);
    for (int t0 = 0; t0 < nq; t0 += 16) {
      const f4 qa = *reinterpret_cast<const f4*>(&sQ[t0 + col][4 * grp]);
      const f4 ga = *reinterpret_cast<const f4*>(&sDo[t0 + col][4 * grp]);
      f4 s = {0.0f, 0.0f, 0.0f, 0.0f}, dp = {0.0f, 0.0f, 0.0f, 0.0f};
      s = mfma4(qa.x, kreg[0], s);
      dp = mfma4(ga.x, vreg[0], dp);
      s = mfma4(qa.y, kreg[1], s);
      dp = mfma4(ga.y, vreg[1], dp);
      s = mfma4(qa.z, kreg[2], s);
      dp = mfma4(ga.z, vreg[2], dp);
      s = mfma4(qa.w, kreg[3], s);
      dp = mfma4(ga.w, vreg[3], dp);
      const f4 l4 = *reinterpret_cast<const f4*>(&sLse[t0 + 4 * grp]);
      const f4 d4 = *reinterpret_cast<const f4*>(&sD[t0 + 4 * grp]);
      f4 p, ds;
      p.x = ex2((s.x + kbias) - l4.x);
      p.y = ex2((s.y + kbias) - l4.y);
      p.z = ex2((s.z + kbias) - l4.z);
      p.w = ex2((s.w + kbias) - l4.w);
      ds.x = p.x * (dp.x - d4.x);
      ds.y = p.y * (dp.y - d4.y);
      ds.z = p.z * (dp.z - d4.z);
      ds.w = p.w * (dp.w - d4.w);
      const f4 gt = *reinterpret_cast<const f4*>(&sDot[col][t0 + 4 * grp]);
      const f4 qt4 = *reinterpret_cast<const f4*>(&sQt[col][t0 + 4 * grp]);
      acc_v = mfma4(gt.x, p.x, acc_v);
      acc_k = mfma4(qt4.x, ds.x, acc_k);
      acc_v = mfma4(gt.y, p.y, acc_v);
      acc_k = mfma4(qt4.y, ds.y, acc_k);
      acc_v = mfma4(gt.z, p.z, acc_v);
      acc_k = mfma4(qt4.z, ds.z, acc_k);
      acc_v = mfma4(gt.w, p.w, acc_v);
      acc_k = mfma4(qt4.w, ds.w, acc_k);
    }
  }
  if (kok) {
    float* dkp = dk + ((long long)b * A.Lk + ki) * D + h * A.hd;
    float* dvp = dv + ((long long)b * A.Lk + ki) * D + h * A.hd;
    // sQt carries scale * log2 e: dK = scale * dS^T Q = (dS^T Q_scaled) * ln 2
    const float kk[4] = {acc_k.x * kLn2, acc_k.y * kLn2, acc_k.z * kLn2, acc_k.w * kLn2};
    const float vv[4] = {acc_v.x, acc_v.y, acc_v.z, acc_v.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * grp + r < A.hd) {
        dkp[4 * grp + r] = kk[r];
        dvp[4 * grp + r] = vv[r];
      }
  }
}

bool args_ok(const AttnArgs& A) { return A.B > 0 && A.H > 0 && A.Lq > 0 && A.Lk > 0 && A.hd >= 1 && A.hd <= 16; }

}  // namespace

// 0 = launched, 1 = unsupported shape (head_dim > 16)
int launch_train_attention_fwd(const float* q, const float* k, const float* v, const long long* strides6, const uint8_t* pad, int B, int H,
                               int Lq, int Lk, int hd, float scale, float* out, float* lse, hipStream_t s) {
  AttnArgs A{q, k, v, strides6[0], strides6[1], strides6[2], strides6[3], strides6[4], strides6[5], pad, B, H, Lq, Lk, hd, scale};
  if (!args_ok(A)) return 1;
  if (Lq <= 16)
    hipLaunchKernelGGL(k_tattn_fwd<true>, dim3((unsigned)(B * H)), dim3(kNT), 0, s, A, out, lse);
  else
    hipLaunchKernelGGL(k_tattn_fwd<false>, dim3((unsigned)(B * H * ((Lq + 16 * kNW - 1) / (16 * kNW)))), dim3(kNT), 0, s, A, out, lse);
  return 0;
}

int launch_train_attention_bwd(const float* q, const float* k, const float* v, const long long* strides6, const uint8_t* pad, int B, int H,
                               int Lq, int Lk, int hd, float scale, const float* out, const float* dout, const float* lse, float* dsum,
                               float* dq, float* dk, float* dv, hipStream_t s) {
  AttnArgs A{q, k, v, strides6[0], strides6[1], strides6[2], strides6[3], strides6[4], strides6[5], pad, B, H, Lq, Lk, hd, scale};
  if (!args_ok(A)) return 1;
  if (Lq <= 16)
    hipLaunchKernelGGL(k_tattn_dq<true>, dim3((unsigned)(B * H)), dim3(kNT), 0, s, A, out, dout, lse, dq, dsum);
  else
    hipLaunchKernelGGL(k_tattn_dq<false>, dim3((unsigned)(B * H * ((Lq + 16 * kNW - 1) / (16 * kNW)))), dim3(kNT), 0, s, A, out, dout, lse, dq, dsum);
  hipLaunchKernelGGL(k_tattn_dkv, dim3((unsigned)(B * H * ((Lk + 16 * kNW - 1) / (16 * kNW)))), dim3(kNT), 0, s, A, dout, lse, dsum, dk, dv);
  return 0;
}

}  // namespace mmf
