// mmf_kernels_backbone.hip -- self-attention of the FROZEN image backbone (ViT-B/16 shape: head dim 64) at float32 accuracy on the
// fp16 matrix cores of gfx950.
//
// The training step's frozen backbone runs its Linears as one fp16 GEMM of split operands each (diffuser_actor/split_linear.py);
// what was left on f32 arithmetic was the attention: torch's f32 SDPA kernel reaches 81 TFLOP/s (the f32 matrix rate of this part
// is 1/16 of its fp16 rate), 15 ms of a 59 ms step.  Here every f32 operand is split x = hi + lo / 2048 (hi = fp16(x),
// lo = fp16((x - hi) 2048): 22 bits of mantissa, lo kept a normal number) and every product a b is
//     a_hi b_hi  +  (a_hi b_lo + a_lo b_hi) / 2048                          (the lo lo term is 2^-22 of the product)
// on v_mfma_f32_32x32x16_f16 with f32 accumulation: two accumulators (main, cross), combined once per tile.  Softmax statistics,
// exponentials and the rescaling run in f32 on the vector unit; P is split like the other operands before P V.
//
// Flash-style: a workgroup = 4 waves x 32 query rows of one (batch, head); K / V tiles of 64 keys are split ONCE per workgroup
// into LDS (K row-major, V transposed with the key order of the MFMA's accumulator layout, both XOR-swizzled in 16-byte chunks:
// conflict-free ds_read_b128 for the operand reads), double-buffered against the global loads of the next tile.  S is computed
// TRANSPOSED (A = K tile, B = Q^T): a lane then owns one query column of the accumulator -- row maximum / sum are in-lane
// reductions + one exchange with lane ^ 32 -- and its 16 accumulator values of a 32-key sub-tile are exactly the B operand
// (P^T) of the second product O^T = V^T P^T, with no data movement in between.
#include <hip/hip_runtime.h>

#include "mmf_launch.h"

namespace mmf {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr float kLo = 2048.0f, kLoInv = 1.0f / 2048.0f;
constexpr int kD = 64;     // head dimension
constexpr int kTK = 64;    // keys per tile
constexpr int kQW = 32;    // query rows per wave
constexpr int kQB = 128;   // query rows per workgroup

__device__ __forceinline__ f16v mfma(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    hi[t] = (_Float16)x[t];
    lo[t] = (_Float16)((x[t] - (float)hi[t]) * kLo);
  }
}

// LDS addressing, in halves.  K tile: [key 64][d 64], the 16-byte chunk index XORed with (key >> 1) & 7.  V tile: [d 64][slot 64]
// (slot = the position of a key in the MFMA operand order, key_slot below), chunk index XORed with (d >> 1) & 7.
// (a row is 128 bytes = half the 64 banks: two consecutive rows tile the banks, so the swizzle runs over row >> 1 -- with row & 7 the
// rows 2 apart met in the same banks and SQ_LDS_BANK_CONFLICT was 64 % of the LDS cycles)
__device__ __forceinline__ int k_addr(int key, int chunk) { return key * kD + ((chunk ^ ((key >> 1) & 7)) << 3); }
__device__ __forceinline__ int v_addr(int d, int chunk) { return d * kTK + ((chunk ^ ((d >> 1) & 7)) << 3); }
// A 32 x 32 accumulator holds, in lane (col, half) register r, row (r & 3) + 8 (r >> 2) + 4 half.  The second product consumes a
// sub-tile's 16 registers as two operands of 8 (registers 8 u .. 8 u + 7): step s = 2 sub + u covers keys 16 s .. 16 s + 15, and
// operand element (half, j) is key 16 s + 8 (j >> 2) + 4 half + (j & 3).  slot = 16 s + 8 half + j is where V^T keeps that key.
__device__ __forceinline__ int key_slot(int key) {
  return (key & ~15) + ((key >> 2) & 1) * 8 + ((key >> 3) & 1) * 4 + (key & 3);
}

struct Tile {  // one thread's share of a K / V tile on its way from global memory to LDS
  float4 k[4];  // K[key = tid >> 2][16 (tid & 3) ..]
  float4 v[4];  // V[keys 2 p, 2 p + 1 (p = tid & 31)][8 (tid >> 5) ..]: v[0..1] the even key, v[2..3] the odd one (a wave's lanes then
                // write 32 different words of two rows of V^T per store: 2-way bank conflicts instead of 8-way)
};

__device__ __forceinline__ void tile_load(Tile& T, const float* __restrict__ K, const float* __restrict__ V, long long rs, int k0, int tid) {
  const float* kp = K + (long long)(k0 + (tid >> 2)) * rs + (tid & 3) * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) T.k[i] = reinterpret_cast<const float4*>(kp)[i];
  const float* vp = V + (long long)(k0 + 2 * (tid & 31)) * rs + (tid >> 5) * 8;
  T.v[0] = reinterpret_cast<const float4*>(vp)[0];
  T.v[1] = reinterpret_cast<const float4*>(vp)[1];
  T.v[2] = reinterpret_cast<const float4*>(vp + rs)[0];
  T.v[3] = reinterpret_cast<const float4*>(vp + rs)[1];
}

__device__ __forceinline__ void tile_store(const Tile& T, _Float16* __restrict__ sKh, _Float16* __restrict__ sKl, _Float16* __restrict__ sVh,
                                           _Float16* __restrict__ sVl, int tid) {
  {
    const int key = tid >> 2, c0 = (tid & 3) * 2;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float x[8] = {T.k[2 * c].x, T.k[2 * c].y, T.k[2 * c].z, T.k[2 * c].w, T.k[2 * c + 1].x, T.k[2 * c + 1].y, T.k[2 * c + 1].z, T.k[2 * c + 1].w};
      h8 hi, lo;
      split8(x, hi, lo);
      *reinterpret_cast<h8*>(sKh + k_addr(key, c0 + c)) = hi;
      *reinterpret_cast<h8*>(sKl + k_addr(key, c0 + c)) = lo;
    }
  }
  {
    const int slot = key_slot(2 * (tid & 31));  // even: the odd key of the pair sits in slot + 1
    const int d0 = (tid >> 5) * 8;
    const float a[8] = {T.v[0].x, T.v[0].y, T.v[0].z, T.v[0].w, T.v[1].x, T.v[1].y, T.v[1].z, T.v[1].w};
    const float b[8] = {T.v[2].x, T.v[2].y, T.v[2].z, T.v[2].w, T.v[3].x, T.v[3].y, T.v[3].z, T.v[3].w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const _Float16 ah = (_Float16)a[t], bh = (_Float16)b[t];
      const h2 hi = {ah, bh};
      const h2 lo = {(_Float16)((a[t] - (float)ah) * kLo), (_Float16)((b[t] - (float)bh) * kLo)};
      const int at = v_addr(d0 + t, slot >> 3) + (slot & 7);
      *reinterpret_cast<h2*>(sVh + at) = hi;
      *reinterpret_cast<h2*>(sVl + at) = lo;
    }
  }
}

// q / k / v: [B, L, ...] rows of one head at stride `rs` floats between consecutive l, `bs` between batches; head h at + 64 h.
// out: [B, L, H 64] (the layout the next Linear reads: no transpose copy).  L % 128 == 0.
// SPLIT_OUT: the output is written as the split operand of the next Linear's GEMM (mmf_split_activations3's layout: rows of
// [hi | hi / 2048 | lo | 1, 1 / 2048, 0 x 62] fp16 over K = 64 H channels) instead of float32 -- one pass over the activation less.
template <bool SPLIT_OUT>
__global__ __launch_bounds__(256, 2) void k_attn_split64(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                        long long rs, long long bs, int H, int L, float scale, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) _Float16 sK[2][2][kTK * kD];
  __shared__ __attribute__((aligned(16))) _Float16 sV[2][2][kD * kTK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int b = (int)blockIdx.y / H, h = (int)blockIdx.y % H;
  const float* Qb = Q + (long long)b * bs + h * kD;
  const float* Kb = K + (long long)b * bs + h * kD;
  const float* Vb = V + (long long)b * bs + h * kD;
  const int q = (int)blockIdx.x * kQB + wave * kQW + l31;

  // the wave's Q^T operand (B of the first product): lane (q, half) holds d = 16 kk + 8 half + j; the softmax scale (a power of two
  // for d = 64: exact) is applied before the split
  h8 qh[4], ql[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const float4* p = reinterpret_cast<const float4*>(Qb + (long long)q * rs + kk * 16 + half * 8);
    const float4 a = p[0], c = p[1];
    // (log2 e rides in the scale: the exponentials below are bare v_exp_f32; one more rounding of q at 2^-24)
    const float sc = scale * 1.44269504088896340736f;
    const float x[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, c.x * sc, c.y * sc, c.z * sc, c.w * sc};
    split8(x, qh[kk], ql[kk]);
  }

  f16v om[2], ox[2];  // O^T accumulators (main, cross) of the two 32-row halves of d
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) om[db][r] = ox[db][r] = 0.0f;
  float m_run = -3.0e38f, l_run = 0.0f;

  Tile T;
  tile_load(T, Kb, Vb, rs, 0, tid);
  tile_store(T, sK[0][0], sK[0][1], sV[0][0], sV[0][1], tid);
  __syncthreads();
  const int nt = L / kTK;
  for (int t = 0; t < nt; ++t) {
    const int st = t & 1;
    if (t + 1 < nt) tile_load(T, Kb, Vb, rs, (t + 1) * kTK, tid);  // in flight during the products below
    const _Float16* Kh = sK[st][0];
    const _Float16* Kl = sK[st][1];
    const _Float16* Vh = sV[st][0];
    const _Float16* Vl = sV[st][1];
    // S^T = K Q^T for the tile's two 32-key sub-tiles
    float s[2][16];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f16v sm, sx;
#pragma unroll
      for (int r = 0; r < 16; ++r) sm[r] = sx[r] = 0.0f;
      const int key = 32 * sub + l31;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const h8 kh = *reinterpret_cast<const h8*>(Kh + k_addr(key, 2 * kk + half));
        const h8 kl = *reinterpret_cast<const h8*>(Kl + k_addr(key, 2 * kk + half));
        sm = mfma(kh, qh[kk], sm);
        sx = mfma(kh, ql[kk], sx);
        sx = mfma(kl, qh[kk], sx);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[sub][r] = sm[r] + sx[r] * kLoInv;
    }
    // online softmax of the lane's query column (its 32 values + the partner lane's 32)
    float mx = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[sub][r] = __builtin_amdgcn_exp2f(s[sub][r] - m_new);
        psum += s[sub][r];
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        om[db][r] *= alpha;
        ox[db][r] *= alpha;
      }
    m_run = m_new;
    // O^T += V^T P^T: four steps of 16 keys
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float x[8] = {s[sub][8 * u], s[sub][8 * u + 1], s[sub][8 * u + 2], s[sub][8 * u + 3],
                            s[sub][8 * u + 4], s[sub][8 * u + 5], s[sub][8 * u + 6], s[sub][8 * u + 7]};
        h8 ph, pl;
        split8(x, ph, pl);
        const int step = 2 * sub + u;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int d = 32 * db + l31;
          const h8 vh = *reinterpret_cast<const h8*>(Vh + v_addr(d, 2 * step + half));
          const h8 vl = *reinterpret_cast<const h8*>(Vl + v_addr(d, 2 * step + half));
          om[db] = mfma(vh, ph, om[db]);
          ox[db] = mfma(vh, pl, ox[db]);
          ox[db] = mfma(vl, ph, ox[db]);
        }
      }
    if (t + 1 < nt) tile_store(T, sK[st ^ 1][0], sK[st ^ 1][1], sV[st ^ 1][0], sV[st ^ 1][1], tid);
    __syncthreads();
  }
  // O^T[d][q] -> out[b, q, 64 h + d]; the lane holds d = 32 db + (r & 3) + 8 (r >> 2) + 4 half: four consecutive d per register quad
  const float inv_l = 1.0f / l_run;
  const int KC = H * kD;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  float* op = out + ((long long)b * L + q) * (long long)KC + h * kD;
  _Float16* sp = reinterpret_cast<_Float16*>(out) + ((long long)b * L + q) * (long long)(3 * KC + 64) + h * kD;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (om[db][4 * g + e] + ox[db][4 * g + e] * kLoInv) * inv_l;
        asm volatile("" : "+v"(o[e]));  // (one value for the hi and the lo part: see k_split_act3_src)
      }
      const int dd = 32 * db + 8 * g + 4 * half;
      if (SPLIT_OUT) {
        h4 hi, hs, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = (_Float16)o[e];
          const _Float16 l = (_Float16)((o[e] - (float)hi[e]) * kLo);
          hs[e] = (_Float16)((float)hi[e] * kLoInv);
          lo[e] = (_Float16)((float)l * kLoInv);
        }
        *reinterpret_cast<h4*>(sp + dd) = hi;
        *reinterpret_cast<h4*>(sp + KC + dd) = hs;
        *reinterpret_cast<h4*>(sp + 2 * KC + dd) = lo;
      } else {
        *reinterpret_cast<float4*>(op + dd) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  if (SPLIT_OUT && h == 0) {  // the row's bias columns: 1, 1 / 2048, zeros (lane half 0: the first 32, half 1: the rest)
    _Float16* tp = sp + 3 * KC + 32 * half;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      if (c == 0 && half == 0) {
        z[0] = (_Float16)1.0f;
        z[1] = (_Float16)kLoInv;
      }
      *reinterpret_cast<h8*>(tp + 8 * c) = z;
    }
  }
}

}  // namespace

// 0 = launched, 1 = unsupported shape
int launch_attention_split(const float* q, const float* k, const float* v, long long row_stride, long long batch_stride, int B, int H, int L,
                           int head_dim, float scale, void* out, int split_out, hipStream_t s) {
  if (head_dim != kD || L <= 0 || L % kQB != 0 || B <= 0 || H <= 0 || (row_stride & 3) || (batch_stride & 3)) return 1;
  const dim3 grid((unsigned)(L / kQB), (unsigned)(B * H));
  if (split_out)
    hipLaunchKernelGGL(k_attn_split64<true>, grid, dim3(256), 0, s, q, k, v, row_stride, batch_stride, H, L, scale, reinterpret_cast<float*>(out));
  else
    hipLaunchKernelGGL(k_attn_split64<false>, grid, dim3(256), 0, s, q, k, v, row_stride, batch_stride, H, L, scale, reinterpret_cast<float*>(out));
  return 0;
}

}  // namespace mmf
