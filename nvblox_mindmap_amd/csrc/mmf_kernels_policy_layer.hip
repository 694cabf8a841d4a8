// Per-layer inference kernels of the diffusion head around the attention kernel (mmf_kernels_policy_mfma.hip): the q | k | v
// projections and the out_proj + LayerNorm + feed-forward block, on 16-token tiles, D = 120 channels, 8 heads of 15.
//
// GEMMs: float32 results from THREE half-precision matrix-core products.  Every f32 operand is split x = hi + lo / 2048 with
// hi = fp16(x) and lo = fp16((x - hi) * 2048) -- 22 bits of mantissa; the scaling keeps lo a normal number for |x| > 6e-5 / 2048 --
// and a product accumulates  hi.hi  in one f32 accumulator and  hi.lo + lo.hi  in a second one that is scaled by 1/2048 at the
// end (the lo.lo term is 2^-22 of the product).  v_mfma_f32_16x16x32_f16 runs at 16x the rate of v_mfma_f32_16x16x4_f32, so a
// 16 x 128 x 16 tile product costs 12 MFMAs of 16 cycles instead of 32 of 32 cycles: these kernels were bound by the f32 matrix
// rate of the few CUs a 616-token step can occupy (39 tiles: 1.35 us per GEMM, three per block).  Against the reference's
// DiffusionHead outputs the split form deviates by 1.1e-6, the f32 MFMA form by 0.8e-6 (tests/test_gpu_policy_golden.py: 1e-4).
// Operands must be finite and below 65504 in magnitude (activations behind a LayerNorm and trained weights are).
//
//   weights   pre-split once per model by k_split_weight (mmf_split_linear_weight): [out][chunk 4][s 4][hi 8 | lo 8] halves,
//             zero beyond D: a lane's B operand of a 32-wide reduction chunk is two adjacent 16-byte pieces of one 128-byte line
//   A tiles   live in LDS as two fp16 planes [16 tokens][136]: written by the lanes that produce them (LayerNorm lanes own 8
//             adjacent channels of a token = one operand piece; GEMM epilogues write their D-layout values as halves), read as
//             16-byte pieces.  Reduction chunk c, lane (i = l & 15, s = l >> 4), element t  <->  channel 32 c + 8 s + t for A
//             and B alike
//   D tiles   row = 4 s + r, col = l & 15 (as every gfx950 MFMA)
//
//   k_qkv_heads      q = rotary(q_proj(modulated x)) | k = rotary(k_proj(x)) | v = v_proj(x), written HEAD-MAJOR and padded to
//                    16 channels: Qp, Kp [B, H, L16, 16], Vt [B, H, 16, L16] -- the layouts k_attention_heads loads from
//   k_out_ffn_mfma   x1 = LN(res + out_proj(att)); h = modulate(x1); out = LN(h + fc2(relu(fc1(h))))
//   k_out_ffn_qkv    the same, then the NEXT layer's q | k | v on the tile, which never leaves the workgroup
//   ...2             two independent stacks of identical shape (the rotation and the position stack) in one launch
//
// Loads are written REQUEST FIRST, USE LATER and pinned with scheduling barriers: the compiler keeps the program order of loads,
// puts a wait in front of the first use, and otherwise sinks every load to its use (one memory round trip per MFMA group).
#include <cstdlib>

#include "mmf_device.h"
#include "mmf_launch.h"
#include "mmf_trace_device.h"

namespace mmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));

constexpr int kD = 120, kH = 8, kDH = 15;  // the policy's embedding dim / heads / head dim (the kernels are built for these)
constexpr int kRS = 132;                    // LDS row stride of an f32 tile (floats)
constexpr int kPS = 136;                    // LDS row stride of an fp16 plane (halves): 272 B, rows 4 banks apart
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;
constexpr int kWRow = 256;                  // halves per output row of a split weight matrix: 4 chunks x 4 s x (8 hi | 8 lo)

#ifdef MMF_WG_TRACE
constexpr int kPtQkv = 0, kPtOutFfn = 1024;
__device__ __forceinline__ void pt_mark(int base, int i, float dep) {
  asm volatile("" ::"v"(dep));
  if (g_wg_trace && threadIdx.x == 0) {
    const long long blk = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
    const long long off = 3ll * 6 * 8192 + 8 * (base + blk) + i;
    if (off < 3ll * g_wg_trace_cap) g_wg_trace[off] = (unsigned long long)wall_clock64();
  }
}
#define MMF_PT(base, i, dep) pt_mark(base, i, dep)
#else
#define MMF_PT(base, i, dep)
#endif

// ---- operand splitting ---------------------------------------------------------------------------------------------------------
struct Split8 {
  h8 hi, lo;
};
struct U4 {
  uint32_t u[4];
};
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const hp2 ph = __builtin_amdgcn_cvt_pkrtz(a, b);  // toward zero: x - hi is exact and has the sign of x
  const float ra = (a - (float)ph[0]) * kLoScale, rb = (b - (float)ph[1]) * kLoScale;
  const hp2 pl = __builtin_amdgcn_cvt_pkrtz(ra, rb);
  hi = __builtin_bit_cast(uint32_t, ph);
  lo = __builtin_bit_cast(uint32_t, pl);
}
__device__ __forceinline__ Split8 split8(const float (&v)[8]) {
  U4 H, Lo;
#pragma unroll
  for (int t = 0; t < 4; ++t) split2(v[2 * t], v[2 * t + 1], H.u[t], Lo.u[t]);
  Split8 S;
  S.hi = __builtin_bit_cast(h8, H);
  S.lo = __builtin_bit_cast(h8, Lo);
  return S;
}

struct Planes {  // an A tile: 16 tokens x 128 channels (120 .. 127 finite, their weights are zero)
  _Float16 hi[16][kPS];
  _Float16 lo[16][kPS];
};
__device__ __forceinline__ void store_piece(Planes& P, int tl, int c0, const float (&v)[8]) {
  const Split8 S = split8(v);
  *reinterpret_cast<h8*>(&P.hi[tl][c0]) = S.hi;
  *reinterpret_cast<h8*>(&P.lo[tl][c0]) = S.lo;
}

struct BTile {  // B operand of one 16-column tile: 4 reduction chunks
  h8 hi[4], lo[4];
};
__device__ __forceinline__ void load_btile(const _Float16* __restrict__ W, int row, int s, BTile& B) {
  const _Float16* p = W + (size_t)row * kWRow + s * 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    B.hi[c] = *reinterpret_cast<const h8*>(p + c * 64);
    B.lo[c] = *reinterpret_cast<const h8*>(p + c * 64 + 8);
  }
}

__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// two 16 x 16 output tiles of the wave from one pass over the A tile: 24 MFMAs on four accumulators
__device__ __forceinline__ void gemm2(const Planes& A, int i, int s, const BTile& B0, const BTile& B1, f32x4& y0, f32x4& y1) {
  f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, x0 = m0, m1 = m0, x1 = m0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const h8 ah = *reinterpret_cast<const h8*>(&A.hi[i][32 * c + 8 * s]);
    const h8 al = *reinterpret_cast<const h8*>(&A.lo[i][32 * c + 8 * s]);
    m0 = mfma_h(ah, B0.hi[c], m0);
    m1 = mfma_h(ah, B1.hi[c], m1);
    x0 = mfma_h(ah, B0.lo[c], x0);
    x1 = mfma_h(ah, B1.lo[c], x1);
    x0 = mfma_h(al, B0.hi[c], x0);
    x1 = mfma_h(al, B1.hi[c], x1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    y0[r] = m0[r] + x0[r] * kLoInv;
    y1[r] = m1[r] + x1[r] * kLoInv;
  }
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
  v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
}

// weights [out, in = D] f32 (torch.nn.Linear's own layout) -> the split layout; one thread per (row, chunk, s) piece
__global__ __launch_bounds__(256) void k_split_weight(const float* __restrict__ W, int rows, _Float16* __restrict__ dst) {
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (e >= rows * 16) return;
  const int row = e >> 4, piece = e & 15, c0 = 8 * piece;  // piece = 4 c + s: channels 32 c + 8 s + t = 8 piece + t
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = c0 + t < kD ? W[(size_t)row * kD + c0 + t] : 0.0f;
  // round to nearest for the constant operand (either rounding gives a 22-bit representation)
  h8 hi, lo;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    hi[t] = (_Float16)v[t];
    lo[t] = (_Float16)((v[t] - (float)hi[t]) * kLoScale);
  }
  _Float16* p = dst + (size_t)row * kWRow + piece * 16;
  *reinterpret_cast<h8*>(p) = hi;
  *reinterpret_cast<h8*>(p + 8) = lo;
}

int launch_split_weight(const float* W, int rows, int cols, void* dst, hipStream_t s) {
  if (cols != kD || rows <= 0) return 1;
  hipLaunchKernelGGL(k_split_weight, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, W, rows, reinterpret_cast<_Float16*>(dst));
  return 0;
}

// ---- q | k | v projections, rotary, head-major outputs ---------------------------------------------------------------------
struct QkvArgs {
  const float* ss;       // AdaLN (scale | shift) [B, 2 D] of the query input or null
  const _Float16* Wq;    // split [D] rows
  const float* bq;
  const _Float16* Wkv;   // split [2 D] rows: keys then values
  const float *bkv, *cs, *sn;  // cs / sn [B, L, D] or null
  float *Qp, *Kp, *Vt;
};

// One role (0 = q, 1 = k, 2 = v) of a 16-token tile (batch element b, first token l0) for the wave's two heads 2 w, 2 w + 1
// (30 channels: rotary pairs never leave the wave).  Lanes of the padding column (j = 15) and of rows beyond L compute on
// clamped addresses; their results are zeroed at the store.  ROLE / ROT (rotary tables present) are TEMPLATE parameters and the
// kernels branch once, at the top, into a straight-line body: a run-time condition around a group of loads makes the compiler
// split the request sequence at the branch, wait there, and start a second round trip behind it.
struct QkvOps {
  BTile B0, B1;
  float bb[2], cv[2][4], sv[2][4];
};
template <int ROLE, bool ROT>
__device__ __forceinline__ void qkv_role_loads(const QkvArgs& Q, int b, int l0, int L, int w, int j, int s, QkvOps& O) {
  const _Float16* W = ROLE == 0 ? Q.Wq : Q.Wkv + (ROLE == 2 ? (size_t)kD * kWRow : 0);  // values: rows D .. 2 D - 1 of Wkv
  const float* bias = ROLE == 0 ? Q.bq : Q.bkv + (ROLE == 2 ? kD : 0);
  const int jc = min(j, kDH - 1);
  load_btile(W, kDH * (2 * w) + jc, s, O.B0);
  load_btile(W, kDH * (2 * w + 1) + jc, s, O.B1);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = kDH * (2 * w + n) + jc;
    O.bb[n] = bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ROLE < 2 && ROT) {
        const size_t e = ((size_t)b * L + min(l0 + 4 * s + r, L - 1)) * kD + col;
        O.cv[n][r] = Q.cs[e];
        O.sv[n][r] = Q.sn[e];
      } else {
        O.cv[n][r] = 1.0f;
        O.sv[n][r] = 0.0f;
      }
    }
  }
}

template <int ROLE, bool ROT>
__device__ __forceinline__ void qkv_role_compute(const Planes& A, const QkvOps& O, const QkvArgs& Q, int b, int l0, int L, int L16, int w,
                                                 int j, int s) {
  f32x4 y[2];
  gemm2(A, j, s, O.B0, O.B1, y[0], y[1]);
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) y[n][r] += O.bb[n];
  MMF_PT(kPtQkv, 3, y[1][0]);

  if (ROLE < 2 && ROT) {
    // rotary over the 120-vector: out[c] = y[c] cos[c] + (c odd ? y[c-1] : -y[c+1]) sin[c].  Within the wave's 30 channels
    // p = 15 n + j the partner is p ^ 1, held by lane (s, j') of tile n'.
    f32x4 part[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int p = kDH * n + (j < kDH ? j : 0), pp = p ^ 1;
      const int np = pp >= kDH ? 1 : 0, jp = pp - kDH * np;
      const int src = s * 16 + jp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = __shfl(y[0][r], src, 64), v1 = __shfl(y[1][r], src, 64);
        const float v = np ? v1 : v0;
        part[n][r] = (p & 1) ? v : -v;
      }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) y[n][r] = y[n][r] * O.cv[n][r] + part[n][r] * O.sv[n][r];
  }

#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int h = 2 * w + n;
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (j < kDH && l0 + 4 * s + r < L) ? y[n][r] : 0.0f;  // padding stays finite (zero)
    if (ROLE == 2) {
      *reinterpret_cast<f32x4*>(Q.Vt + (((size_t)b * kH + h) * 16 + j) * L16 + l0 + 4 * s) = o;
    } else {
      float* P = (ROLE == 0 ? Q.Qp : Q.Kp) + (((size_t)b * kH + h) * L16 + l0 + 4 * s) * 16 + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) P[r * 16] = o[r];
    }
  }
}

// One (tile, role) of k_qkv_heads / k_qkv_heads2.  The 256 lanes convert the tile together: lane (tl = tid >> 4, q = tid & 15)
// owns channels 8 q .. 8 q + 7 of token tl -- one operand piece (lane 15: the zero padding)
template <int ROLE, bool ROT, bool MOD>
__device__ __forceinline__ void qkv_heads_tile(const float* __restrict__ x, const QkvArgs& Q, int L, int L16, Planes& P) {
  const int tpb = L16 / 16;
  const int b = (int)blockIdx.x / tpb, l0 = ((int)blockIdx.x % tpb) * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, s = lane >> 4;
  const int tl = tid >> 4, q = tid & 15, c0 = 8 * min(q, 14);
  const int tok = l0 + tl;
  MMF_PT(kPtQkv, 0, 0.0f);
  float v[8], g[8], h[8];
  load8(x + ((size_t)b * L + min(tok, L - 1)) * kD + c0, v);
  if (MOD) {  // AdaLN modulation of the query input
    load8(Q.ss + (size_t)b * 2 * kD + c0, g);
    load8(Q.ss + (size_t)b * 2 * kD + kD + c0, h);
  }
  QkvOps O;
  qkv_role_loads<ROLE, ROT>(Q, b, l0, L, w, j, s, O);
  __builtin_amdgcn_sched_barrier(0);
  const bool live = tok < L && q < 15;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (MOD) v[t] = v[t] * (1.0f + g[t]) + h[t];
    v[t] = live ? v[t] : 0.0f;
  }
  store_piece(P, tl, 8 * q, v);
  __syncthreads();
  MMF_PT(kPtQkv, 1, v[0]);
  qkv_role_compute<ROLE, ROT>(P, O, Q, b, l0, L, L16, w, j, s);
  MMF_PT(kPtQkv, 4, 0.0f);
}

__device__ __forceinline__ void qkv_heads_body(const float* __restrict__ x, const QkvArgs& Q, int L, int L16, int role, Planes& P) {
  const bool rot = Q.cs != nullptr, mod = Q.ss != nullptr;
  if (role == 0) {
    if (rot) {
      if (mod)
        qkv_heads_tile<0, true, true>(x, Q, L, L16, P);
      else
        qkv_heads_tile<0, true, false>(x, Q, L, L16, P);
    } else {
      if (mod)
        qkv_heads_tile<0, false, true>(x, Q, L, L16, P);
      else
        qkv_heads_tile<0, false, false>(x, Q, L, L16, P);
    }
  } else if (role == 1) {
    if (rot)
      qkv_heads_tile<1, true, false>(x, Q, L, L16, P);
    else
      qkv_heads_tile<1, false, false>(x, Q, L, L16, P);
  } else {
    qkv_heads_tile<2, false, false>(x, Q, L, L16, P);
  }
}

// grid (B * L16 / 16, roles), 256 threads
__global__ __launch_bounds__(256) void k_qkv_heads(const float* __restrict__ x, QkvArgs Q, int L, int L16, int role0) {
  __shared__ __attribute__((aligned(16))) Planes P;
  qkv_heads_body(x, Q, L, L16, role0 + (int)blockIdx.y, P);  // 0 = q, 1 = k, 2 = v
}

// (The stack's argument block is chosen by a BRANCH around two calls, not by `second ? Q1 : Q0`: a reference picked at run
// time makes the compiler copy the block to scratch memory and read every pointer back from there.)
__global__ __launch_bounds__(256) void k_qkv_heads2(const float* __restrict__ x0, const float* __restrict__ x1, QkvArgs Q0, QkvArgs Q1,
                                                   int L, int L16) {
  __shared__ __attribute__((aligned(16))) Planes P;
  const int role = (int)blockIdx.y;  // 0 = q, 1 = k, 2 = v
  if (blockIdx.z != 0)
    qkv_heads_body(x1, Q1, L, L16, role, P);
  else
    qkv_heads_body(x0, Q0, L, L16, role, P);
}

// ---- out_proj + LayerNorm + feed-forward block (+ the next layer's projections) -----------------------------------------------
constexpr int kPartRows = 18;  // rows of one key-split attention partial: 16 channels of O^T, then the maxima, then the sums
struct AttPartials {            // k_attention_heads<.., SPLIT > 1> output to merge instead of reading `att` (null: read att)
  const float* part;            // [B, H, n_split, 18, 16]
  int n_split, Lq;              // Lq <= 16 query rows per batch element
};

struct OutFfnArgs {
  const float *att, *res;
  const _Float16* Wo;
  const float *bo, *g1, *be1, *ss;  // ss: AdaLN (scale | shift) of the FFN or null
  const _Float16* W1;
  const float* b1;
  const _Float16* W2;
  const float *b2, *g2, *be2;
  float eps1, eps2;
  float* out;
};

struct TileLds {
  float sY[16][kRS];  // pre-LayerNorm sums: written in D layout by a GEMM epilogue, read by the LayerNorm lanes
  float sH[16][kRS];  // h = modulate(LN1(..)) in f32: the residual of fc2
  Planes P0, P1;      // A tiles: att -> u = relu(fc1 h) -> modulated x2 | h -> x2
};

struct LnShare {
  float g[8], b[8];
};
__device__ __forceinline__ LnShare load_ln_share(const float* __restrict__ gamma, const float* __restrict__ beta, int c0) {
  LnShare P;
  load8(gamma + c0, P.g);
  load8(beta + c0, P.b);
  return P;
}

// LayerNorm of token tl = tid >> 4 by its 16 lanes (lane q < 15 owns channels 8 q .. 8 q + 7, lane 15 idles): o = LN(src + add)
// [* (1 + sc) + sh]
__device__ __forceinline__ void ln_row(const float (*src)[kRS], int tl, int q, const float (&add)[8], const LnShare& P, float eps,
                                       const float (&sc)[8], const float (&sh)[8], float (&o)[8]) {
  const bool own = q < 15;
  const int c0 = 8 * min(q, 14);
  float v[8];
  load8(&src[tl][c0], v);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = own ? v[i] + add[i] : 0.0f;
    sum += v[i];
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  const float mean = sum / (float)kD;
  float var = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float d = own ? v[i] - mean : 0.0f;
    v[i] = d;
    var += d * d;
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
  const float inv = rsqrtf(var / (float)kD + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float n = v[i] * inv * P.g[i] + P.b[i];
    o[i] = own ? n * (1.0f + sc[i]) + sh[i] : 0.0f;
  }
}

// x1 = LN1(res + out_proj(att)); h = modulate(x1); out = LN2(h + fc2(relu(fc1(h)))) for the 16 tokens t0 .. t0 + 15 of the
// flattened [B L] token axis (`tokens` = end of the tile's token range; rows beyond it are inert); wave w owns output columns
// [32 w, 32 w + 32).  QKV: then the next layer's projections `roles` (bit 0 q, 1 k, 2 v) of the tile (batch element b, first
// token l0 = t0 - b L); QROT / QMOD: that layer has rotary tables / modulates its query input.
template <bool QKV, bool QROT, bool QMOD>
__device__ __forceinline__ void out_ffn_tile(const OutFfnArgs& A, long long t0, long long tokens, int L, TileLds& S, const AttPartials& AP,
                                             const QkvArgs& Q, int roles, int b, int l0, int L16) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, s = lane >> 4;
  const int tl = tid >> 4, q = tid & 15, c0 = 8 * min(q, 14);
  const long long ltok = t0 + tl, ltokc = min(ltok, tokens - 1);
  const bool live = ltok < tokens && q < 15;
  MMF_PT(kPtOutFfn, 0, 0.0f);
  // requests in the order of first use: the input piece, the out_proj weights, LayerNorm 1 operands
  // (output columns 120 .. 127 -- the last 8 lanes of wave 3's second tile -- compute on row 119's weights; their results are
  // finite and meet zero weights wherever they are reduced over)
  const int colc[2] = {min(32 * w + j, kD - 1), min(32 * w + 16 + j, kD - 1)};
  float av[8];
  if (AP.part == nullptr) load8(A.att + ltokc * kD + c0, av);
  BTile B0, B1, C0, C1;
  load_btile(A.Wo, colc[0], s, B0);
  load_btile(A.Wo, colc[1], s, B1);
  float bbo[2], bb1[2], bb2[2], rs[8], sc[8], sh[8];
#pragma unroll
  for (int n = 0; n < 2; ++n) bbo[n] = A.bo[colc[n]];
  const LnShare P1 = load_ln_share(A.g1, A.be1, c0);
  load8(A.res + ltokc * kD + c0, rs);  // the residual row of this lane's LayerNorm token
  {
    const bool has = A.ss != nullptr;
    const float* sp = has ? A.ss + (size_t)((int)ltokc / L) * 2 * kD : A.g1;  // (no AdaLN: any readable floats, zeroed below)
    load8(sp + c0, sc);
    load8(sp + (has ? kD : 0) + c0, sh);
  }
  __builtin_amdgcn_sched_barrier(0);

  if (AP.part != nullptr) {
    // the attention output of this tile, merged from the key splits: element (token, channel c = 15 h + ch) =
    // sum_sp e^(m_sp - M) O_sp[ch][row] / sum_sp e^(m_sp - M) l_sp, M = max_sp m_sp
    for (int e = tid; e < 16 * 128; e += 256) {
      const int et = e >> 7, c = e & 127;
      const long long tok = t0 + et;
      float v = 0.0f;
      if (c < kD && tok < tokens) {
        const int eb = (int)(tok / AP.Lq), row = (int)(tok - (long long)eb * AP.Lq), h = c / kDH, ch = c - h * kDH;
        const float* P = AP.part + ((size_t)eb * kH + h) * AP.n_split * kPartRows * 16;
        float M = -INFINITY;
        for (int sp = 0; sp < AP.n_split; ++sp) M = fmaxf(M, P[(sp * kPartRows + 16) * 16 + row]);
        float num = 0.0f, den = 0.0f;
        for (int sp = 0; sp < AP.n_split; ++sp) {
          const float m = P[(sp * kPartRows + 16) * 16 + row];
          const float f = (m == -INFINITY) ? 0.0f : __expf(m - M);
          num += f * P[(sp * kPartRows + ch) * 16 + row];
          den += f * P[(sp * kPartRows + 17) * 16 + row];
        }
        v = num / den;
      }
      S.sY[et][c] = v;
    }
    __syncthreads();
    load8(&S.sY[tl][c0], av);
    __syncthreads();  // sY is written again below
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    av[t] = live ? av[t] : 0.0f;
    if (ltok >= tokens) rs[t] = 0.0f;
    if (A.ss == nullptr) sc[t] = 0.0f, sh[t] = 0.0f;  // x (1 + 0) + 0 = x exactly
  }
  store_piece(S.P0, tl, 8 * q, av);
  __syncthreads();

  // ---- x1 = LN1(res + out_proj(att)), h = modulate(x1);  fc1's weights are requested ahead of the MFMAs
  load_btile(A.W1, colc[0], s, C0);
  load_btile(A.W1, colc[1], s, C1);
#pragma unroll
  for (int n = 0; n < 2; ++n) bb1[n] = A.b1[colc[n]];
  __builtin_amdgcn_sched_barrier(0);
  MMF_PT(kPtOutFfn, 1, bbo[0]);
  {
    f32x4 y[2];
    gemm2(S.P0, j, s, B0, B1, y[0], y[1]);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) S.sY[4 * s + r][32 * w + 16 * n + j] = y[n][r] + bbo[n];
    MMF_PT(kPtOutFfn, 2, y[1][0]);
  }
  __syncthreads();
  {
    float o[8];
    ln_row(S.sY, tl, q, rs, P1, A.eps1, sc, sh, o);
    *reinterpret_cast<float4*>(&S.sH[tl][8 * q]) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(&S.sH[tl][8 * q + 4]) = make_float4(o[4], o[5], o[6], o[7]);
    store_piece(S.P1, tl, 8 * q, o);
  }
  __syncthreads();

  // ---- u = relu(fc1(h));  fc2's weights and the LayerNorm 2 operands are requested ahead of the MFMAs
  load_btile(A.W2, colc[0], s, B0);
  load_btile(A.W2, colc[1], s, B1);
#pragma unroll
  for (int n = 0; n < 2; ++n) bb2[n] = A.b2[colc[n]];
  const LnShare P2 = load_ln_share(A.g2, A.be2, c0);
  float qg[8], qh[8];
  if (QKV && QMOD) {
    load8(Q.ss + (size_t)b * 2 * kD + c0, qg);
    load8(Q.ss + (size_t)b * 2 * kD + kD + c0, qh);
  }
  __builtin_amdgcn_sched_barrier(0);
  MMF_PT(kPtOutFfn, 3, bb1[0]);
  {
    f32x4 y[2];
    gemm2(S.P1, j, s, C0, C1, y[0], y[1]);
    // the D-layout values go straight into the next A tile as halves
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = 32 * w + 16 * n + j;
      uint32_t hi01, lo01, hi23, lo23;
      split2(fmaxf(y[n][0] + bb1[n], 0.0f), fmaxf(y[n][1] + bb1[n], 0.0f), hi01, lo01);
      split2(fmaxf(y[n][2] + bb1[n], 0.0f), fmaxf(y[n][3] + bb1[n], 0.0f), hi23, lo23);
      const hp2 h01 = __builtin_bit_cast(hp2, hi01), l01 = __builtin_bit_cast(hp2, lo01);
      const hp2 h23 = __builtin_bit_cast(hp2, hi23), l23 = __builtin_bit_cast(hp2, lo23);
      S.P0.hi[4 * s][col] = (_Float16)h01[0], S.P0.hi[4 * s + 1][col] = (_Float16)h01[1];
      S.P0.hi[4 * s + 2][col] = (_Float16)h23[0], S.P0.hi[4 * s + 3][col] = (_Float16)h23[1];
      S.P0.lo[4 * s][col] = (_Float16)l01[0], S.P0.lo[4 * s + 1][col] = (_Float16)l01[1];
      S.P0.lo[4 * s + 2][col] = (_Float16)l23[0], S.P0.lo[4 * s + 3][col] = (_Float16)l23[1];
    }
    MMF_PT(kPtOutFfn, 4, y[1][0]);
  }
  __syncthreads();

  // ---- out = LN2(h + fc2(u));  the first role's operands of the next layer are requested ahead of the MFMAs
  QkvOps O;
  if (QKV) {
    if (roles & 2)
      qkv_role_loads<1, QROT>(Q, b, l0, L, w, j, s, O);
    else
      qkv_role_loads<0, QROT>(Q, b, l0, L, w, j, s, O);
    __builtin_amdgcn_sched_barrier(0);
  }
  MMF_PT(kPtOutFfn, 5, bb2[0]);
  {
    f32x4 y[2];
    gemm2(S.P0, j, s, B0, B1, y[0], y[1]);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 32 * w + 16 * n + j;
        S.sY[4 * s + r][col] = S.sH[4 * s + r][col] + (y[n][r] + bb2[n]);
      }
    MMF_PT(kPtOutFfn, 6, y[1][0]);
  }
  __syncthreads();
  {
    const float zero8[8] = {};
    float o[8];
    ln_row(S.sY, tl, q, zero8, P2, A.eps2, zero8, zero8, o);
    if (live) {
      *reinterpret_cast<float4*>(A.out + ltok * kD + 8 * q) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(A.out + ltok * kD + 8 * q + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (QKV) {  // the next layer's A tiles: the block's output (rows beyond the batch element's tokens zero), and its modulated form
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = live ? o[t] : 0.0f;
      store_piece(S.P1, tl, 8 * q, o);
      if (QMOD) {
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = live ? o[t] * (1.0f + qg[t]) + qh[t] : 0.0f;
        store_piece(S.P0, tl, 8 * q, o);
      }
    }
  }
  MMF_PT(kPtOutFfn, 7, 0.0f);
  if (!QKV) return;
  __syncthreads();
  // keys, values, queries (roles 7) or the queries alone (roles 1): each role's operands are requested before the previous
  // role's MFMAs
  const Planes& Aq = QMOD ? S.P0 : S.P1;
  if (roles & 2) {
    QkvOps O2;
    qkv_role_loads<2, false>(Q, b, l0, L, w, j, s, O2);
    __builtin_amdgcn_sched_barrier(0);
    qkv_role_compute<1, QROT>(S.P1, O, Q, b, l0, L, L16, w, j, s);
    qkv_role_loads<0, QROT>(Q, b, l0, L, w, j, s, O);
    __builtin_amdgcn_sched_barrier(0);
    qkv_role_compute<2, false>(S.P1, O2, Q, b, l0, L, L16, w, j, s);
    qkv_role_compute<0, QROT>(Aq, O, Q, b, l0, L, L16, w, j, s);
  } else {
    qkv_role_compute<0, QROT>(Aq, O, Q, b, l0, L, L16, w, j, s);
  }
}

// Helper workgroups (the grid beyond the tiles): every launch starts behind an L2 invalidate, so its weights arrive at the
// latency of the memory side -- and a CU keeps only so many misses in flight: ~60 KB of weights per GEMM stream into a tile's CU
// at 20 - 35 GB/s (2 - 3 us per matrix).  Idle CUs of the same XCD pull the matrices into the shared L2 meanwhile, a slice each;
// the tile workgroups' later requests are L2 hits.  `rows` split-weight rows of 512 bytes per matrix.
__device__ __forceinline__ void warm_weights(const _Float16* W0, const _Float16* W1, const _Float16* W2, int rows, int helper, int nhelpers) {
  const int per = 3 * rows * 2;  // 256-byte half rows: one 16-byte piece per lane of a 16-lane group
  float acc = 0.0f;
  for (int u = helper * 16 + ((int)threadIdx.x >> 4); u < per; u += nhelpers * 16) {
    const int m = u / (rows * 2), r = u - m * rows * 2;
    const _Float16* W = m == 0 ? W0 : (m == 1 ? W1 : W2);
    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(W) + (size_t)r * 256 + ((int)threadIdx.x & 15) * 16);
    acc += v.x;
  }
  if (acc == 1.2345e-38f) asm volatile("s_nop 0");  // keep the loads
}

// grid = 16-token tiles of the flattened [B L] token axis (+ helper workgroups)
__global__ __launch_bounds__(256) void k_out_ffn_mfma(OutFfnArgs A, int L, long long tokens, AttPartials AP, int tiles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  if ((int)blockIdx.x >= tiles) {
    warm_weights(A.Wo, A.W1, A.W2, kD, (int)blockIdx.x - tiles, (int)gridDim.x - tiles);
    return;
  }
  out_ffn_tile<false, false, false>(A, (long long)blockIdx.x * 16, tokens, L, S, AP, QkvArgs{}, 0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_out_ffn_mfma2(OutFfnArgs A0, OutFfnArgs A1, int L, long long tokens, int tiles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  const AttPartials none{nullptr, 0, 0};
  if ((int)blockIdx.x >= tiles)
    out_ffn_tile<false, false, false>(A1, (long long)((int)blockIdx.x - tiles) * 16, tokens, L, S, none, QkvArgs{}, 0, 0, 0, 0);
  else
    out_ffn_tile<false, false, false>(A0, (long long)blockIdx.x * 16, tokens, L, S, none, QkvArgs{}, 0, 0, 0, 0);
}

// The tail of layer i and the head of layer i + 1 in one launch: the block on a 16-token tile of ONE batch element, then the
// q | k | v projections of the NEXT layer on the tile's fresh output, which never leaves the workgroup (one kernel boundary and
// one round trip of the activations through memory less per layer).  `tile` = index among the B * L16 / 16 tiles.
// `roles`: which of the next layer's projections (bit 0 q, 1 k, 2 v); q alone when the next layer attends to a cached memory.
// `AP`: the attention output arrives as key-split partials (then L <= 16).
__device__ __forceinline__ void out_ffn_qkv_body(const OutFfnArgs& A, const QkvArgs& Q, int L, int L16, int roles, const AttPartials& AP,
                                                 int tile, TileLds& S) {
  const int tpb = L16 / 16;
  const int b = tile / tpb, l0 = (tile % tpb) * 16;
  // rows of the tile beyond the batch element's L tokens are inert: the tile's token range ends at (b + 1) L
  const long long t0 = (long long)b * L + l0, tend = (long long)(b + 1) * L;
  const bool rot = Q.cs != nullptr, mod = Q.ss != nullptr && (roles & 1);
  if (rot) {
    if (mod)
      out_ffn_tile<true, true, true>(A, t0, tend, L, S, AP, Q, roles, b, l0, L16);
    else
      out_ffn_tile<true, true, false>(A, t0, tend, L, S, AP, Q, roles, b, l0, L16);
  } else {
    if (mod)
      out_ffn_tile<true, false, true>(A, t0, tend, L, S, AP, Q, roles, b, l0, L16);
    else
      out_ffn_tile<true, false, false>(A, t0, tend, L, S, AP, Q, roles, b, l0, L16);
  }
}

__global__ __launch_bounds__(256) void k_out_ffn_qkv(OutFfnArgs A, QkvArgs Q, int L, int L16, int roles, AttPartials AP) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  out_ffn_qkv_body(A, Q, L, L16, roles, AP, (int)blockIdx.x, S);
}

// two stacks: blocks [0, tiles) serve stack 0, [tiles, 2 tiles) stack 1
__global__ __launch_bounds__(256) void k_out_ffn_qkv2(OutFfnArgs A0, OutFfnArgs A1, QkvArgs Q0, QkvArgs Q1, int L, int L16, int tiles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  const AttPartials none{nullptr, 0, 0};
  if ((int)blockIdx.x >= tiles)
    out_ffn_qkv_body(A1, Q1, L, L16, 7, none, (int)blockIdx.x - tiles, S);
  else
    out_ffn_qkv_body(A0, Q0, L, L16, 7, none, (int)blockIdx.x, S);
}

MMF_DEFINE_WG_TRACE_SETTER(set_wg_trace_policy_layer)

// ---- launchers (weights: split fp16 matrices of launch_split_weight) ------------------------------------------------------------
static inline const _Float16* W16(const float* p) { return reinterpret_cast<const _Float16*>(p); }

static QkvArgs qkv_args(const float* const* q7, float* Qp, float* Kp, float* Vt) {
  return QkvArgs{q7[0], W16(q7[1]), q7[2], W16(q7[3]), q7[4], q7[5], q7[6], Qp, Kp, Vt};
}
static OutFfnArgs out_ffn_args(const float* const* a13, float eps1, float eps2, float* out) {
  return OutFfnArgs{a13[0], a13[1], W16(a13[2]), a13[3], a13[4], a13[5], a13[6], W16(a13[7]), a13[8], W16(a13[9]), a13[10], a13[11], a13[12],
                    eps1,   eps2,   out};
}

int launch_qkv_heads(const float* x, const float* ss, const float* Wq, const float* bq, const float* Wkv, const float* bkv, const float* cs,
                     const float* sn, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H, int roles, hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16;
  // roles: 7 = q | k | v (self-attention), 1 = q alone, 6 = k | v alone (a memory whose keys / values are cached)
  const int role0 = (roles & 1) ? 0 : 1, nroles = roles == 7 ? 3 : (roles == 1 ? 1 : 2);
  QkvArgs Q{ss, W16(Wq), bq, W16(Wkv), bkv, cs, sn, Qp, Kp, Vt};
  hipLaunchKernelGGL(k_qkv_heads, dim3(B * (L16 / 16), nroles), dim3(256), 0, s, x, Q, L, L16, role0);
  return 0;
}

// q14: {ss, Wq, bq, Wkv, bkv, cs, sn} of stack 0 then of stack 1; Qp / Kp / Vt: stack-major [2, B, H, ...] outputs
int launch_qkv_heads2(const float* x0, const float* x1, const float* const* q14, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H,
                      hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16;
  const size_t half = (size_t)B * kH * L16 * 16;
  const QkvArgs Q0 = qkv_args(q14, Qp, Kp, Vt), Q1 = qkv_args(q14 + 7, Qp + half, Kp + half, Vt + half);
  hipLaunchKernelGGL(k_qkv_heads2, dim3(B * (L16 / 16), 3, 2), dim3(256), 0, s, x0, x1, Q0, Q1, L, L16);
  return 0;
}

// a26: OutFfnArgs pointer fields (att .. be2) of stack 0 then of stack 1; out: stack-major [2, B, L, D]
int launch_out_ffn_mfma2(const float* const* a26, const float* eps4, float* out, int B, int L, int D, hipStream_t s) {
  if (D != kD) return 1;
  const long long tokens = (long long)B * L;
  const int tiles = (int)((tokens + 15) / 16);
  const OutFfnArgs A0 = out_ffn_args(a26, eps4[0], eps4[1], out), A1 = out_ffn_args(a26 + 13, eps4[2], eps4[3], out + tokens * kD);
  hipLaunchKernelGGL(k_out_ffn_mfma2, dim3(2 * tiles), dim3(256), 0, s, A0, A1, L, tokens, tiles);
  return 0;
}

int launch_out_ffn_mfma(const float* att, const float* res, const float* Wo, const float* bo, const float* g1, const float* be1, float eps1,
                        const float* ss, const float* W1, const float* b1, const float* W2, const float* b2, const float* g2,
                        const float* be2, float eps2, float* out, int B, int L, int D, hipStream_t s) {
  if (D != kD) return 1;
  const long long tokens = (long long)B * L;
  OutFfnArgs A{att, res, W16(Wo), bo, g1, be1, ss, W16(W1), b1, W16(W2), b2, g2, be2, eps1, eps2, out};
  const int tiles = (int)((tokens + 15) / 16);
  static const int warm = getenv("MMF_DEBUG_WARM") ? atoi(getenv("MMF_DEBUG_WARM")) : 0;
  hipLaunchKernelGGL(k_out_ffn_mfma, dim3((unsigned)(tiles + warm)), dim3(256), 0, s, A, L, tokens, AttPartials{nullptr, 0, 0}, tiles);
  return 0;
}

// the same with the attention output given as the key-split partials of launch_attention_heads_split (L = Lq <= 16)
int launch_out_ffn_mfma_partials(const float* partials, int n_split, const float* res, const float* Wo, const float* bo, const float* g1,
                                 const float* be1, float eps1, const float* ss, const float* W1, const float* b1, const float* W2,
                                 const float* b2, const float* g2, const float* be2, float eps2, float* out, int B, int L, int D,
                                 hipStream_t s) {
  if (D != kD || L > 16 || n_split < 1) return 1;
  const long long tokens = (long long)B * L;
  OutFfnArgs A{partials, res, W16(Wo), bo, g1, be1, ss, W16(W1), b1, W16(W2), b2, g2, be2, eps1, eps2, out};
  hipLaunchKernelGGL(k_out_ffn_mfma, dim3((unsigned)((tokens + 15) / 16)), dim3(256), 0, s, A, L, tokens, AttPartials{partials, n_split, L},
                     (int)((tokens + 15) / 16));
  return 0;
}

// args13: att, res, Wo, bo, g1, be1, ss, W1, b1, W2, b2, g2, be2 (OutFfnArgs order); next7: ss, Wq, bq, Wkv, bkv, cs, sn
int launch_out_ffn_qkv(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp, float* Kp,
                       float* Vt, int B, int L, int D, int H, int roles, const float* partials, int n_split, hipStream_t s) {
  if (D != kD || H != kH || (partials && L > 16)) return 1;
  const int L16 = (L + 15) / 16 * 16;
  const OutFfnArgs A = out_ffn_args(args13, eps1, eps2, out);
  const QkvArgs Q = qkv_args(next7, Qp, Kp, Vt);
  hipLaunchKernelGGL(k_out_ffn_qkv, dim3(B * (L16 / 16)), dim3(256), 0, s, A, Q, L, L16, roles, AttPartials{partials, n_split, L});
  return 0;
}

// the two-stack form: a26 / eps4 / out as launch_out_ffn_mfma2, q14 / Qp / Kp / Vt as launch_qkv_heads2 (roles 7)
int launch_out_ffn_qkv2(const float* const* a26, const float* eps4, float* out, const float* const* q14, float* Qp, float* Kp, float* Vt,
                        int B, int L, int D, int H, hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16;
  const int tiles = B * (L16 / 16);
  const size_t half = (size_t)B * kH * L16 * 16;
  const OutFfnArgs A0 = out_ffn_args(a26, eps4[0], eps4[1], out), A1 = out_ffn_args(a26 + 13, eps4[2], eps4[3], out + (size_t)B * L * kD);
  const QkvArgs Q0 = qkv_args(q14, Qp, Kp, Vt), Q1 = qkv_args(q14 + 7, Qp + half, Kp + half, Vt + half);
  hipLaunchKernelGGL(k_out_ffn_qkv2, dim3(2 * tiles), dim3(256), 0, s, A0, A1, Q0, Q1, L, L16, tiles);
  return 0;
}

}  // namespace mmf
