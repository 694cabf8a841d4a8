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
//   weights   pre-split once per model by k_split_weight (mmf_split_linear_weight) into the order in which the waves load them
//             (see load_bhalf), zero beyond D
//   A tiles   live in LDS as two fp16 planes [16 tokens][136]: written by the lanes that produce them (LayerNorm lanes own 8
//             adjacent channels of a token = one operand piece; GEMM epilogues write their D-layout values as halves), read as
//             16-byte pieces.  Reduction chunk c, lane (i = l & 15, s = l >> 4), element t  <->  channel 32 c + 8 s + t for A
//             and B alike
//   D tiles   row = 4 s + r, col = l & 15 (as every gfx950 MFMA)
//
//   k_qkv_heads      q = rotary(q_proj(modulated x)) | k = rotary(k_proj(x)) | v = v_proj(x), written HEAD-MAJOR, padded to 16
//                    channels and SPLIT (see role_store): Qp, Kp [B, H, L16, 16], Vt [B, H, 16, L16] -- what k_attention_heads
//                    loads as MFMA operands
//   k_out_ffn_mfma   x1 = LN(res + out_proj(att)); h = modulate(x1); out = LN(h + fc2(relu(fc1(h))))
//   k_out_ffn_qkv    the same, then the NEXT layer's q | k | v on the tile, which never leaves the workgroup
//   ...2             two independent stacks of identical shape (the rotation and the position stack) in one launch
//
// Loads are written REQUEST FIRST, USE LATER and pinned with scheduling barriers: the compiler keeps the program order of loads,
// puts a wait in front of the first use, and otherwise sinks every load to its use (one memory round trip per MFMA group).
#include "mmf_device.h"
#include "mmf_launch.h"
#include "mmf_trace_device.h"
#include "mmf_policy_attention.h"

namespace mmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));

constexpr int kD = 120, kH = 8, kDH = 15;  // the policy's embedding dim / heads / head dim (the kernels are built for these)
constexpr int kRS = 132;                    // LDS row stride of an f32 tile (floats)
constexpr int kPS = 136;                    // LDS row stride of an fp16 plane (halves): 272 B, rows 4 banks apart
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

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

// A workgroup is 16 waves: a wave pulls ~8 GB/s out of the memory pipeline however many requests it has in flight (4 waves:
// 31 GB/s per CU, 16 waves: 86 - 115 GB/s, measured on cold 64 - 192 KB) and the weights of a block are 61 KB per GEMM -- with 4
// waves their arrival, not the arithmetic, set the kernel time (2 - 3 us per matrix).  The price: four waves share a SIMD, so
// every vector instruction that all 16 waves execute costs 16 cycles of it -- 125 instructions per wave are a microsecond.  What
// every wave runs is kept to loads at immediate offsets from scalar bases, a dozen MFMAs and four LDS stores per GEMM; the
// element-wise work runs on the four waves of the piece lanes (one per SIMD).
// Wave w of a GEMM: column tile ct = w & 7 (16 output columns), reduction half kh = w >> 3 (chunks 2 kh, 2 kh + 1); the two
// halves leave their D tiles in two f32 LDS tiles, whose readers -- the piece lanes below -- add them and apply every epilogue
// (LDS float atomics into one tile cost 5 us per GEMM).
constexpr int kNT = 1024;
struct BHalf {  // B operand of one 16-column tile, one reduction half: 2 chunks x (hi, lo)
  h8 hi[2], lo[2];
};
// Split weights are stored IN FRAGMENT ORDER: per block of 128 output rows (120 real, 8 zero)
//   [column tile ct 8][reduction half kh 2][chunk c 2][plane hi | lo][lane (s 4, j 16)][8 halves]
// so that a wave's load of one operand register quad is 1 KB of contiguous memory (a lane reading its own 32 bytes of each of 16
// rows touched 16 cache lines per load instruction and halved the rate at which the weights stream in).  `woff`: the lane's
// byte offset into every matrix block, computed once -- a matrix is four loads at immediate offsets from a scalar base.
constexpr int kWBlock = 128 * 256;  // halves per 128-row block of a split weight matrix (64 KB)
__device__ __forceinline__ uint32_t bhalf_offset(int ct, int kh, int lane) { return (uint32_t)((ct * 2 + kh) * 4096 + lane * 16); }
__device__ __forceinline__ void load_bhalf(const _Float16* __restrict__ W, uint32_t woff, BHalf& B) {
  const char* p = reinterpret_cast<const char*>(W) + woff;
  B.hi[0] = *reinterpret_cast<const h8*>(p);
  B.lo[0] = *reinterpret_cast<const h8*>(p + 1024);
  B.hi[1] = *reinterpret_cast<const h8*>(p + 2048);
  B.lo[1] = *reinterpret_cast<const h8*>(p + 3072);
}

__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// the wave's half of one 16 x 16 output tile -> acc[kh][row][16 ct + col]; the readers add the two halves
__device__ __forceinline__ void gemm_half(const Planes& A, const BHalf& B, int ct, int kh, int j, int s, float (*acc)[16][kRS]) {
  f32x4 m = {0.f, 0.f, 0.f, 0.f}, x = m;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const h8 ah = *reinterpret_cast<const h8*>(&A.hi[j][64 * kh + 32 * c + 8 * s]);
    const h8 al = *reinterpret_cast<const h8*>(&A.lo[j][64 * kh + 32 * c + 8 * s]);
    m = mfma_h(ah, B.hi[c], m);
    x = mfma_h(ah, B.lo[c], x);
    x = mfma_h(al, B.hi[c], x);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[kh][4 * s + r][16 * ct + j] = m[r] + x[r] * kLoInv;
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
  v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
}

// weights [rows = blocks x D, in = D] f32 (torch.nn.Linear's own layout; blocks = 1, or 2 for the stacked key | value projection)
// -> the split layout, one 128-row block per D rows; one thread per 8-channel piece of a padded row
__global__ __launch_bounds__(256) void k_split_weight(const float* __restrict__ W, int blocks, _Float16* __restrict__ dst) {
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (e >= blocks * 128 * 16) return;
  const int prow = e >> 4, piece = e & 15;             // padded row, piece = 4 (2 kh + c) + s: channels 8 piece + t
  const int blk = prow >> 7, row = prow & 127;         // row of the block: column tile ct = row >> 4, j = row & 15
  const int kh = piece >> 3, c = (piece >> 2) & 1, s4 = piece & 3;
  h8 hi, lo;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int ch = 8 * piece + t;
    const float v = (row < kD && ch < kD) ? W[((size_t)blk * kD + row) * kD + ch] : 0.0f;
    // round to nearest for the constant operand (either rounding gives a 22-bit representation)
    hi[t] = (_Float16)v;
    lo[t] = (_Float16)((v - (float)hi[t]) * kLoScale);
  }
  _Float16* p = dst + (size_t)blk * kWBlock + (((row >> 4) * 2 + kh) * 4 + c * 2) * 512 + (s4 * 16 + (row & 15)) * 8;
  *reinterpret_cast<h8*>(p) = hi;
  *reinterpret_cast<h8*>(p + 512) = lo;
}

int launch_split_weight(const float* W, int rows, int cols, void* dst, hipStream_t s) {
  if (cols != kD || rows <= 0 || rows % kD != 0) return 1;
  const int blocks = rows / kD;
  hipLaunchKernelGGL(k_split_weight, dim3((unsigned)((blocks * 128 * 16 + 255) / 256)), dim3(256), 0, s, W, blocks, reinterpret_cast<_Float16*>(dst));
  return 0;
}

// Activations [rows, K] f32 -> [rows, 3 K + 64] fp16 = [hi | hi / 2048 | lo | 1, 1 / 2048, 0 ..] (lo = fp16((x - hi) * 2048) / 2048,
// i.e. the true low part, still a normal number for |x| > 0.125 and within 2^-24 absolutely below that) for LARGE GEMMs on a library
// kernel: with the weight matrix stored as [hi | lo * 2048 | hi | bias hi, bias lo * 2048, 0 ..] along the reduction axis, ONE plain
// fp16 GEMM with f32 accumulation returns x w^T + bias at f32 accuracy (x_hi w_hi + x_hi w_lo + x_lo w_hi: 22-bit mantissas) -- no
// scaling, no epilogue.  Needs |x| < 65 504.  One thread per 8 values (and the last thread of a row: the bias columns).
constexpr int kTail = 64;  // columns behind the three parts: 1, 1 / 2048 (the bias), zeros -- 64 keeps a row a multiple of 128 bytes
__global__ __launch_bounds__(256) void k_split_act3(const float* __restrict__ x, long long pieces, int K, _Float16* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= pieces) return;
  const int ppr = K / 8;  // pieces per row
  const long long r = e / ppr;
  const int p = (int)(e - r * ppr), c = p * 8;
  float v[8];
  load8(x + r * K + c, v);
  h8 hi, hs, lo;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    hi[t] = (_Float16)v[t];
    const _Float16 l = (_Float16)((v[t] - (float)hi[t]) * kLoScale);
    hs[t] = (_Float16)((float)hi[t] * kLoInv);  // exact power-of-two scaling (down to the subnormal grid)
    lo[t] = (_Float16)((float)l * kLoInv);
  }
  _Float16* o = out + r * (3 * K + kTail) + c;
  *reinterpret_cast<h8*>(o) = hi;
  *reinterpret_cast<h8*>(o + K) = hs;
  *reinterpret_cast<h8*>(o + 2 * K) = lo;
  if (p >= ppr - kTail / 8) {  // (a row has at least kTail / 8 pieces: K >= 64)
    const int tp = p - (ppr - kTail / 8);
    h8 one = {(_Float16)(tp == 0 ? 1.0f : 0.0f), (_Float16)(tp == 0 ? kLoInv : 0.0f), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
              (_Float16)0.f, (_Float16)0.f};
    *reinterpret_cast<h8*>(out + r * (3 * K + kTail) + 3 * K + 8 * tp) = one;
  }
}

int launch_split_act3(const float* x, long long rows, int K, void* out, hipStream_t s) {
  if (rows <= 0 || K < kTail || K % 8 != 0) return 1;
  const long long pieces = rows * (K / 8);
  hipLaunchKernelGGL(k_split_act3, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, x, pieces, K, reinterpret_cast<_Float16*>(out));
  return 0;
}

// ---- the frozen backbone's element-wise passes, fused with the split (training step: every one of them is an HBM round trip of a
// [32 768, 768 .. 3 072] float32 activation) ------------------------------------------------------------------
// One thread per 8 values like k_split_act3; SRC: 0 plain rows, 1 exact GELU of the input (0.5 x (1 + erf(x / sqrt 2)), torch's
// default) first, 2 the input is the attention output [B, H, L, d] (what SDPA returns) read as rows (b, l) of H d channels -- the
// transpose-and-reshape copy the composite ops make is skipped.
__device__ __forceinline__ void split3_store(const float (&v)[8], long long r, int c, int p, int K, _Float16* __restrict__ out) {
  h8 hi, hs, lo;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    hi[t] = (_Float16)v[t];
    const _Float16 l = (_Float16)((v[t] - (float)hi[t]) * kLoScale);
    hs[t] = (_Float16)((float)hi[t] * kLoInv);
    lo[t] = (_Float16)((float)l * kLoInv);
  }
  _Float16* o = out + r * (3 * K + kTail) + c;
  *reinterpret_cast<h8*>(o) = hi;
  *reinterpret_cast<h8*>(o + K) = hs;
  *reinterpret_cast<h8*>(o + 2 * K) = lo;
  const int ppr = K / 8;
  if (p >= ppr - kTail / 8) {
    const int tp = p - (ppr - kTail / 8);
    h8 one = {(_Float16)(tp == 0 ? 1.0f : 0.0f), (_Float16)(tp == 0 ? kLoInv : 0.0f), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
              (_Float16)0.f, (_Float16)0.f};
    *reinterpret_cast<h8*>(out + r * (3 * K + kTail) + 3 * K + 8 * tp) = one;
  }
}

template <int SRC>
__global__ __launch_bounds__(256) void k_split_act3_src(const float* __restrict__ x, long long pieces, int K, int heads, int L,
                                                      _Float16* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= pieces) return;
  const int ppr = K / 8;
  const long long r = e / ppr;
  const int p = (int)(e - r * ppr), c = p * 8;
  float v[8];
  if (SRC == 2) {
    const int d = K / heads, hh = c / d, j = c - hh * d;  // d is a multiple of 8: a piece lies inside one head
    const long long b = r / L, l = r - b * L;
    load8(x + ((b * heads + hh) * L + l) * d + j, v);
  } else {
    load8(x + r * K + c, v);
  }
  if (SRC == 1) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      v[t] = 0.5f * v[t] * (1.0f + erff(v[t] * 0.70710678118654752440f));
      // ONE value per element from here on: without the barrier the compiler re-evaluates the expression for the hi and for the
      // lo part (not bit-identically -- seen as hi and lo disagreeing by a whole fp16 ulp on exact ties)
      asm volatile("" : "+v"(v[t]));
    }
  }
  split3_store(v, r, c, p, K, out);
}

int launch_split_act3_src(int src, const float* x, long long rows, int K, int heads, int L, void* out, hipStream_t s) {
  if (rows <= 0 || K < kTail || K % 8 != 0) return 1;
  if (src == 2 && (heads <= 0 || L <= 0 || K % heads != 0 || (K / heads) % 8 != 0 || rows % L != 0)) return 1;
  const long long pieces = rows * (K / 8);
  const dim3 grid((unsigned)((pieces + 255) / 256));
  _Float16* o = reinterpret_cast<_Float16*>(out);
  if (src == 1)
    hipLaunchKernelGGL(k_split_act3_src<1>, grid, dim3(256), 0, s, x, pieces, K, heads, L, o);
  else if (src == 2)
    hipLaunchKernelGGL(k_split_act3_src<2>, grid, dim3(256), 0, s, x, pieces, K, heads, L, o);
  else
    return 1;
  return 0;
}

// LayerNorm (+ the residual add in front of it) + split: s = x (+ y);  [sum_out <- s];  out <- split3((s - mean) rstd gamma + beta).
// One wave per row, the row in registers (K = 256 NCH, NCH float4 per lane): two-pass mean / variance like torch's
// (sum of squared deviations from the mean: no cancellation), float32 throughout.
template <int NCH>
__global__ __launch_bounds__(256) void k_ln_split3(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float eps, long long rows, float* __restrict__ sum_out,
                                                  _Float16* __restrict__ out) {
  constexpr int K = 256 * NCH;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float4 v[NCH];
  float sum = 0.0f;
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int c = q * 256 + lane * 4;
    v[q] = *reinterpret_cast<const float4*>(x + r * K + c);
    if (y) {
      const float4 w = *reinterpret_cast<const float4*>(y + r * K + c);
      v[q].x += w.x, v[q].y += w.y, v[q].z += w.z, v[q].w += w.w;
      *reinterpret_cast<float4*>(sum_out + r * K + c) = v[q];
    }
    sum += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum * (1.0f / K);
  float sq = 0.0f;
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const float a = v[q].x - mean, b = v[q].y - mean, c = v[q].z - mean, d = v[q].w - mean;
    sq += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq * (1.0f / K) + eps);
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  _Float16* o = out + r * (3 * K + kTail);
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int c = q * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
    float n[4] = {(v[q].x - mean) * rstd * g.x + bt.x, (v[q].y - mean) * rstd * g.y + bt.y, (v[q].z - mean) * rstd * g.z + bt.z,
                  (v[q].w - mean) * rstd * g.w + bt.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(n[t]));  // (one value per element for the hi and the lo part, see the GELU kernel)
    h4 hi, hs, lo;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      hi[t] = (_Float16)n[t];
      const _Float16 l = (_Float16)((n[t] - (float)hi[t]) * kLoScale);
      hs[t] = (_Float16)((float)hi[t] * kLoInv);
      lo[t] = (_Float16)((float)l * kLoInv);
    }
    *reinterpret_cast<h4*>(o + c) = hi;
    *reinterpret_cast<h4*>(o + K + c) = hs;
    *reinterpret_cast<h4*>(o + 2 * K + c) = lo;
  }
  if (lane < kTail / 8) {
    h8 one = {(_Float16)(lane == 0 ? 1.0f : 0.0f), (_Float16)(lane == 0 ? kLoInv : 0.0f), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
              (_Float16)0.f, (_Float16)0.f};
    *reinterpret_cast<h8*>(o + 3 * K + 8 * lane) = one;
  }
}

int launch_ln_split3(const float* x, const float* y, const float* gamma, const float* beta, float eps, long long rows, int K, float* sum_out,
                     void* out, hipStream_t s) {
  if (rows <= 0 || K % 256 != 0 || K < 256 || K > 1024 || (y != nullptr) != (sum_out != nullptr)) return 1;
  const dim3 grid((unsigned)((rows + 3) / 4));
  _Float16* o = reinterpret_cast<_Float16*>(out);
  switch (K / 256) {
    case 1: hipLaunchKernelGGL(k_ln_split3<1>, grid, dim3(256), 0, s, x, y, gamma, beta, eps, rows, sum_out, o); break;
    case 2: hipLaunchKernelGGL(k_ln_split3<2>, grid, dim3(256), 0, s, x, y, gamma, beta, eps, rows, sum_out, o); break;
    case 3: hipLaunchKernelGGL(k_ln_split3<3>, grid, dim3(256), 0, s, x, y, gamma, beta, eps, rows, sum_out, o); break;
    default: hipLaunchKernelGGL(k_ln_split3<4>, grid, dim3(256), 0, s, x, y, gamma, beta, eps, rows, sum_out, o); break;
  }
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

// PIECE LANES: thread (tl = t >> 4, q = t & 15) of a 256-thread group owns channels 8 q .. 8 q + 7 of token tl -- one 16-byte
// operand piece of an A plane, two 16-byte pieces of every f32 row (lane q = 15: the zero padding).  They read a GEMM's result
// from the two half tiles and apply bias / LayerNorm / ReLU / rotary.
__device__ __forceinline__ void take_acc(const float (*acc)[16][kRS], int tl, int q, float (&v)[8]) {
  const float4* p0 = reinterpret_cast<const float4*>(&acc[0][tl][8 * q]);
  const float4* p1 = reinterpret_cast<const float4*>(&acc[1][tl][8 * q]);
  const float4 a = p0[0], b = p0[1], c = p1[0], d = p1[1];
  v[0] = a.x + c.x, v[1] = a.y + c.y, v[2] = a.z + c.z, v[3] = a.w + c.w;
  v[4] = b.x + d.x, v[5] = b.y + d.y, v[6] = b.z + d.z, v[7] = b.w + d.w;
}

// role (0 = q, 1 = k, 2 = v) epilogue, first half, by a piece lane: y = acc + bias [rotary] written back IN PLACE into half
// tile 0 of the role's accumulation pair (rows beyond L as zeros).  bias / cs / sn: the lane's pieces of the role's bias and
// rotary tables.
template <int ROLE, bool ROT>
__device__ __forceinline__ void role_finish(float (*acc)[16][kRS], const float (&bias)[8], const float (&cs)[8], const float (&sn)[8], int l0,
                                            int L, int tl, int q) {
  float y[8];
  take_acc(acc, tl, q, y);
  const bool live = l0 + tl < L && q < 15;
#pragma unroll
  for (int t = 0; t < 8; ++t) y[t] += bias[t];
  if (ROLE < 2 && ROT) {  // out[c] = y[c] cos[c] + (c odd ? y[c-1] : -y[c+1]) sin[c]: the pairs lie inside a piece
    float o[8];
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
      o[t] = y[t] * cs[t] + (-y[t + 1]) * sn[t];
      o[t + 1] = y[t + 1] * cs[t + 1] + y[t] * sn[t + 1];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = o[t];
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) y[t] = live ? y[t] : 0.0f;
  float4* p = reinterpret_cast<float4*>(&acc[0][tl][8 * q]);
  p[0] = make_float4(y[0], y[1], y[2], y[3]);
  p[1] = make_float4(y[4], y[5], y[6], y[7]);
}

// ... second half, by 512 GROUP lanes per role (after a barrier): the operand layouts of k_attention_heads -- HEAD-MAJOR, the
// head's 15 channels padded to 16, every aligned group of four values stored as {4 fp16 hi | 4 fp16 lo} in the 16 bytes four
// floats would take:
//   Qp, Kp [B, H, L16, 16]: group = 4 channels of a token;  Vt [B, H, 16, L16]: group = 4 tokens of a channel.
// Group lane u: q / k  (head h = u >> 6, token tl = (u >> 2) & 15, channel group g = u & 3): one head's 16 tokens are 1 KB of
// contiguous stores;  v  (head h = u >> 6, channel ch = (u >> 2) & 15, token group g = u & 3).
template <int ROLE>
__device__ __forceinline__ void role_store(const float (*tile)[kRS], const QkvArgs& Q, int b, int l0, int L16, int u) {
  const int h = u >> 6, mid = (u >> 2) & 15, g = u & 3;
  float v[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int tl = ROLE == 2 ? 4 * g + p : mid, ch = ROLE == 2 ? mid : 4 * g + p;
    v[p] = ch < kDH ? tile[tl][kDH * h + ch] : 0.0f;  // (channel 15: the padding)
  }
  uint4 o;
  split2(v[0], v[1], o.x, o.z);
  split2(v[2], v[3], o.y, o.w);
  float* dst = ROLE == 2 ? Q.Vt + (((size_t)b * kH + h) * 16 + mid) * L16 + l0 + 4 * g
                         : (ROLE == 0 ? Q.Qp : Q.Kp) + (((size_t)b * kH + h) * L16 + l0 + mid) * 16 + 4 * g;
  *reinterpret_cast<uint4*>(dst) = o;
}

__device__ __forceinline__ const _Float16* role_weights(const QkvArgs& Q, int role) {
  return role == 0 ? Q.Wq : Q.Wkv + (role == 2 ? kWBlock : 0);  // values: the second block of Wkv
}

struct QkvLds {
  float acc[2][16][kRS];
  Planes P;
};

// One (tile, role) of k_qkv_heads / k_qkv_heads2
template <int ROLE, bool ROT, bool MOD>
__device__ __forceinline__ void qkv_heads_tile(const float* __restrict__ x, const QkvArgs& Q, int L, int L16, QkvLds& S) {
  const int tpb = L16 / 16;
  const int b = (int)blockIdx.x / tpb, l0 = ((int)blockIdx.x % tpb) * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, s = lane >> 4;
  const int ct = w & 7, kh = w >> 3;
  const int tl = (tid >> 4) & 15, q = tid & 15, c0 = 8 * min(q, 14);
  const int tok = l0 + tl;
  const bool piece = tid < 256;
  MMF_PT(kPtQkv, 0, 0.0f);
  float v[8], g[8], h[8], rb[8], rc[8], rn[8];
  if (piece) {
    load8(x + ((size_t)b * L + min(tok, L - 1)) * kD + c0, v);
    if (MOD) {  // AdaLN modulation of the query input
      load8(Q.ss + (size_t)b * 2 * kD + c0, g);
      load8(Q.ss + (size_t)b * 2 * kD + kD + c0, h);
    }
  }
  BHalf B;
  load_bhalf(role_weights(Q, ROLE), bhalf_offset(ct, kh, lane), B);
  if (piece) {
    load8((ROLE == 0 ? Q.bq : Q.bkv + (ROLE == 2 ? kD : 0)) + c0, rb);
    if (ROLE < 2 && ROT) {
      const size_t e = ((size_t)b * L + min(tok, L - 1)) * kD + c0;
      load8(Q.cs + e, rc);
      load8(Q.sn + e, rn);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (piece) {
    const bool live = tok < L && q < 15;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (MOD) v[t] = v[t] * (1.0f + g[t]) + h[t];
      v[t] = live ? v[t] : 0.0f;
    }
    store_piece(S.P, tl, 8 * q, v);
  }
  __syncthreads();
  MMF_PT(kPtQkv, 1, 0.0f);
  gemm_half(S.P, B, ct, kh, j, s, S.acc);
  __syncthreads();
  MMF_PT(kPtQkv, 3, 0.0f);
  if (piece) role_finish<ROLE, ROT>(S.acc, rb, rc, rn, l0, L, tl, q);
  __syncthreads();
  if (tid < 512) role_store<ROLE>(S.acc[0], Q, b, l0, L16, tid);
  MMF_PT(kPtQkv, 4, 0.0f);
}

__device__ __forceinline__ void qkv_heads_body(const float* __restrict__ x, const QkvArgs& Q, int L, int L16, int role, QkvLds& S) {
  const bool rot = Q.cs != nullptr, mod = Q.ss != nullptr;
  if (role == 0) {
    if (rot) {
      if (mod)
        qkv_heads_tile<0, true, true>(x, Q, L, L16, S);
      else
        qkv_heads_tile<0, true, false>(x, Q, L, L16, S);
    } else {
      if (mod)
        qkv_heads_tile<0, false, true>(x, Q, L, L16, S);
      else
        qkv_heads_tile<0, false, false>(x, Q, L, L16, S);
    }
  } else if (role == 1) {
    if (rot)
      qkv_heads_tile<1, true, false>(x, Q, L, L16, S);
    else
      qkv_heads_tile<1, false, false>(x, Q, L, L16, S);
  } else {
    qkv_heads_tile<2, false, false>(x, Q, L, L16, S);
  }
}

// grid (B * L16 / 16, roles), 1024 threads
__global__ __launch_bounds__(kNT) void k_qkv_heads(const float* __restrict__ x, QkvArgs Q, int L, int L16, int role0) {
  __shared__ __attribute__((aligned(16))) QkvLds S;
  qkv_heads_body(x, Q, L, L16, role0 + (int)blockIdx.y, S);  // 0 = q, 1 = k, 2 = v
}

// (The stack's argument block is chosen by a BRANCH around two calls, not by `second ? Q1 : Q0`: a reference picked at run
// time makes the compiler copy the block to scratch memory and read every pointer back from there.)
__global__ __launch_bounds__(kNT) void k_qkv_heads2(const float* __restrict__ x0, const float* __restrict__ x1, QkvArgs Q0, QkvArgs Q1,
                                                   int L, int L16) {
  __shared__ __attribute__((aligned(16))) QkvLds S;
  const int role = (int)blockIdx.y;  // 0 = q, 1 = k, 2 = v
  if (blockIdx.z != 0)
    qkv_heads_body(x1, Q1, L, L16, role, S);
  else
    qkv_heads_body(x0, Q0, L, L16, role, S);
}

// ---- out_proj + LayerNorm + feed-forward block (+ the next layer's projections) -----------------------------------------------
constexpr int kPartRows = 18;  // rows of one key-split attention partial: 16 channels of O^T, then the maxima, then the sums
struct AttPartials {            // k_attention_heads<.., SPLIT > 1> output to merge instead of reading `att` (null: read att)
  const float* part;            // [B, H, n_split, 18, 16]
  int n_split, Lq;              // Lq <= 16 query rows per batch element
  const unsigned long long* tagged;  // the same as self-validating words {tag | f32 bits}, written by workgroups of THIS launch
  unsigned tag;                      // (k_cross_layer); `part` is then any non-null pointer
  int* fail;                         // set when the words do not arrive (a peer workgroup is not running)
};
constexpr unsigned kCrossSpinLimit = 1u << 22;

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

// Small operands are STAGED IN LDS by all 1024 threads, one or two 16-byte pieces each: biases and LayerNorm vectors, and per
// token the residual row, the AdaLN scale / shift and the next layer's rotary rows.  (Held in registers by the lanes that use
// them they are 48 registers beside 48 of weights: the kernel spilled.)
constexpr int kVecRows = 12;  // bo g1 be1 b1 b2 g2 be2 | next layer: bq bk bv, query scale, query shift
constexpr int kTokRows = 5;   // residual, scale, shift | next layer: cos, sin
enum : int { kVBo = 0, kVG1, kVBe1, kVB1, kVB2, kVG2, kVBe2, kVBq, kVBk, kVBv, kVQg, kVQh };
enum : int { kTRes = 0, kTSc, kTSh, kTCos, kTSin };
struct TileLds {
  float acc[2][16][kRS];   // the two reduction halves' D tiles of the current GEMM (the queries' in the projection stage)
  float accK[2][16][kRS];  // ... of the next layer's keys
  float accV[2][16][kRS];  // ... and values
  float sH[16][kRS];       // h = modulate(LN1(..)) in f32: the residual of fc2 (before that: scratch of the partial merge)
  Planes P0, P1;           // A tiles: att -> u = relu(fc1 h) -> modulated x2 | h -> x2
  float vec[kVecRows][128];
  float tok[kTokRows][16][128];
};

// sum over the 16 lanes of a token row (= one DPP row) in four cross-lane adds, every lane receiving the total: pairs, quads,
// the mirrored half (the other quad), the mirrored row (the other half).  (__shfl_xor goes through ds_bpermute: eight of them in
// a chain were half of a LayerNorm's time.)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_get<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_get<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_get<0x141>(v);  // row_half_mirror
  v += dpp_get<0x140>(v);  // row_mirror
  return v;
}

// LayerNorm of a piece lane's token by its 16 lanes: o = LN(v) [* (1 + sc) + sh]; lane 15 (no channels) contributes zeros
__device__ __forceinline__ void ln_piece(float (&v)[8], int q, const float (&g)[8], const float (&be)[8], float eps, const float (&sc)[8],
                                         const float (&sh)[8], float (&o)[8]) {
  const bool own = q < 15;
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = own ? v[i] : 0.0f;
    sum += v[i];
  }
  sum = row_sum16(sum);
  const float mean = sum / (float)kD;
  float var = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float d = own ? v[i] - mean : 0.0f;
    v[i] = d;
    var += d * d;
  }
  var = row_sum16(var);
  const float inv = rsqrtf(var / (float)kD + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float n = v[i] * inv * g[i] + be[i];
    o[i] = own ? n * (1.0f + sc[i]) + sh[i] : 0.0f;
  }
}

// x1 = LN1(res + out_proj(att)); h = modulate(x1); out = LN2(h + fc2(relu(fc1(h)))) for the 16 tokens t0 .. t0 + 15 of the
// flattened [B L] token axis (`tokens` = end of the tile's token range; rows beyond it are inert).  QKV: then the next layer's
// projections `roles` (7 = q | k | v, 1 = q alone) of the tile (batch element b, first token l0 = t0 - b L); QROT / QMOD: that
// layer has rotary tables / modulates its query input.
template <bool QKV, bool QROT, bool QMOD>
__device__ __forceinline__ void out_ffn_tile(const OutFfnArgs& A, long long t0, long long tokens, int L, TileLds& S, const AttPartials& AP,
                                             const QkvArgs& Q, int roles, int b, int l0, int L16) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, s = lane >> 4;
  const int ct = w & 7, kh = w >> 3;
  const int grp = tid >> 8;  // piece-lane group: 0 runs the block's epilogues (and the queries'), 1 the keys', 2 the values'
  const int tl = (tid >> 4) & 15, q = tid & 15, c0 = 8 * min(q, 14);
  const long long ltok = t0 + tl, ltokc = min(ltok, tokens - 1);
  const bool live = ltok < tokens && q < 15;
  MMF_PT(kPtOutFfn, 0, 0.0f);
  // Requests in the order of first use: the input piece, the small operands (on their way to LDS: every WAVE takes whole rows
  // of 128 floats, two floats per lane, so that the choice of the row's source is scalar work), the wave's share of the three
  // weight matrices (12 pieces per lane).
  float av[8];
  const bool att_tagged = AP.part == nullptr && AP.tagged != nullptr;  // the attention output arrives inside this launch (k_self_layer)
  if (grp == 0 && AP.part == nullptr && !att_tagged) load8(A.att + ltokc * kD + c0, av);
  constexpr int kNVec = QKV ? kVecRows : 7, kNTok = QKV ? kTokRows : 3, kRows = kNVec + 16 * kNTok, kPer = (kRows + 15) / 16;
  const int wv = __builtin_amdgcn_readfirstlane(w);
  float2 opv[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int r = wv + 16 * k;  // scalar
    const float* base = nullptr;
    if (r < kNVec) {
      base = r == kVBo ? A.bo : r == kVG1 ? A.g1 : r == kVBe1 ? A.be1 : r == kVB1 ? A.b1 : r == kVB2 ? A.b2 : r == kVG2 ? A.g2
           : r == kVBe2 ? A.be2 : r == kVBq ? Q.bq : r == kVBk ? Q.bkv : r == kVBv ? (Q.bkv ? Q.bkv + kD : nullptr)
           : r == kVQg ? (QMOD ? Q.ss + (size_t)b * 2 * kD : nullptr) : (QMOD ? Q.ss + (size_t)b * 2 * kD + kD : nullptr);
    } else if (r < kRows) {
      const int u = r - kNVec, op = u >> 4, utl = u & 15;
      const long long tok = t0 + utl;
      if (op == kTRes) {
        if (tok < tokens) base = A.res + tok * kD;
      } else if (op == kTSc || op == kTSh) {
        if (A.ss != nullptr) base = A.ss + (size_t)((int)min(tok, tokens - 1) / L) * 2 * kD + (op == kTSh ? kD : 0);
      } else if (QROT) {
        base = (op == kTCos ? Q.cs : Q.sn) + ((size_t)b * L + min(l0 + utl, L - 1)) * kD;
      }
    }
    opv[k] = make_float2(0.f, 0.f);  // absent operands, rows beyond the tile's tokens and the padding channels: zeros
    if (base != nullptr && lane < kD / 2) opv[k] = *reinterpret_cast<const float2*>(base + 2 * lane);
  }
  const uint32_t woff = bhalf_offset(ct, kh, lane);
  BHalf Bo, B1, B2;
  load_bhalf(A.Wo, woff, Bo);
  __builtin_amdgcn_sched_barrier(0);

  if (AP.part != nullptr) {
    const float* pbase = AP.part;
    if (AP.tagged != nullptr) {
      // the partials of this batch element come from workgroups of the same launch: poll their words (requested nine per thread
      // at a time, checked afterwards; stale ones again) into the accumulation tiles' LDS, which nobody uses before GEMM 1
      // (only the columns of the Lq live query rows are written and read)
      float* stage = &S.acc[0][0][0];
      const int words = kH * AP.n_split * kPartRows * 16, eb = (int)(t0 / AP.Lq), live = kH * AP.n_split * kPartRows * AP.Lq;
      const unsigned long long* src = AP.tagged + (size_t)eb * words;
      for (int e0 = tid; e0 < live; e0 += 9 * kNT) {
        unsigned long long got[9];
        int at[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const int e = min(e0 + i * kNT, live - 1);
          at[i] = (e / AP.Lq) * 16 + e % AP.Lq;
          got[i] = __hip_atomic_load(src + at[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          if (e0 + i * kNT < live) {
            unsigned spins = 0;
            while ((unsigned)(got[i] >> 32) != AP.tag) {
              if (++spins > kCrossSpinLimit) {
                *AP.fail = 1;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
              got[i] = __hip_atomic_load(src + at[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stage[at[i]] = __uint_as_float((unsigned)got[i]);
          }
        }
      }
      __syncthreads();
      pbase = stage;
    }
    const int eb_stage = AP.tagged != nullptr ? (int)(t0 / AP.Lq) : 0;  // (the staged partials are this batch element's alone)
    // the attention output of this tile, merged from the key splits: element (token, channel c = 15 h + ch) =
    // sum_sp e^(m_sp - M) O_sp[ch][row] / sum_sp e^(m_sp - M) l_sp, M = max_sp m_sp
    for (int e = tid; e < 16 * 128; e += kNT) {
      const int et = e >> 7, c = e & 127;
      const long long tok = t0 + et;
      float v = 0.0f;
      if (c < kD && tok < tokens) {
        const int eb = (int)(tok / AP.Lq), row = (int)(tok - (long long)eb * AP.Lq), h = c / kDH, ch = c - h * kDH;
        const float* Pp = pbase + ((size_t)(eb - eb_stage) * kH + h) * AP.n_split * kPartRows * 16;
        float M = -INFINITY;
        for (int sp = 0; sp < AP.n_split; ++sp) M = fmaxf(M, Pp[(sp * kPartRows + 16) * 16 + row]);
        float num = 0.0f, den = 0.0f;
        for (int sp = 0; sp < AP.n_split; ++sp) {
          const float m = Pp[(sp * kPartRows + 16) * 16 + row];
          const float f = (m == -INFINITY) ? 0.0f : __expf(m - M);
          num += f * Pp[(sp * kPartRows + ch) * 16 + row];
          den += f * Pp[(sp * kPartRows + 17) * 16 + row];
        }
        v = num / den;
      }
      S.sH[et][c] = v;
    }
    __syncthreads();
    if (grp == 0) load8(&S.sH[tl][c0], av);
  }
  if (grp == 0) {
    if (att_tagged && live) {  // the lane's 8 channels: 8 words from (up to 2) attention workgroups of this launch
      const unsigned long long* src = AP.tagged + ltok * kD + 8 * q;
      unsigned long long got[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) got[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned spins = 0;
        while ((unsigned)(got[i] >> 32) != AP.tag) {
          if (++spins > kCrossSpinLimit) {
            *AP.fail = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          got[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        av[i] = __uint_as_float((unsigned)got[i]);
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) av[t] = live ? av[t] : 0.0f;
    store_piece(S.P0, tl, 8 * q, av);
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int r = wv + 16 * k;
    if (r < kRows) {
      float* dst = r < kNVec ? &S.vec[r][0] : &S.tok[(r - kNVec) >> 4][(r - kNVec) & 15][0];
      *reinterpret_cast<float2*>(dst + 2 * lane) = opv[k];
    }
  }
  __syncthreads();
  MMF_PT(kPtOutFfn, 1, 0.0f);

  // ---- x1 = LN1(res + out_proj(att)), h = modulate(x1)
  load_bhalf(A.W1, woff, B1);
  __builtin_amdgcn_sched_barrier(0);
  gemm_half(S.P0, Bo, ct, kh, j, s, S.acc);
  // fc2's weights are requested now -- two matrices ahead: everything up front is 192 KB through the CU before the first MFMA --
  // and the next layer's key weights into the registers of Bo
  load_bhalf(A.W2, woff, B2);
  if (QKV && (roles & 2)) load_bhalf(role_weights(Q, 1), woff, Bo);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  if (grp == 0) {
    float v[8], o[8], x0[8], x1[8], x2[8], x3[8];
    take_acc(S.acc, tl, q, v);
    load8(&S.vec[kVBo][c0], x0);
    load8(&S.tok[kTRes][tl][c0], x1);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = (v[t] + x0[t]) + x1[t];
    load8(&S.vec[kVG1][c0], x0);
    load8(&S.vec[kVBe1][c0], x1);
    load8(&S.tok[kTSc][tl][c0], x2);
    load8(&S.tok[kTSh][tl][c0], x3);
    ln_piece(v, q, x0, x1, A.eps1, x2, x3, o);
    *reinterpret_cast<float4*>(&S.sH[tl][8 * q]) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(&S.sH[tl][8 * q + 4]) = make_float4(o[4], o[5], o[6], o[7]);
    store_piece(S.P1, tl, 8 * q, o);
  }
  __syncthreads();
  MMF_PT(kPtOutFfn, 2, 0.0f);

  // ---- u = relu(fc1(h))
  gemm_half(S.P1, B1, ct, kh, j, s, S.acc);
  if (QKV && (roles & 2)) {  // ... the value weights into the registers of B1
    load_bhalf(role_weights(Q, 2), woff, B1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (grp == 0) {
    float v[8], x0[8];
    take_acc(S.acc, tl, q, v);
    load8(&S.vec[kVB1][c0], x0);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = q < 15 ? fmaxf(v[t] + x0[t], 0.0f) : 0.0f;
    store_piece(S.P0, tl, 8 * q, v);
  }
  __syncthreads();
  MMF_PT(kPtOutFfn, 3, 0.0f);

  // ---- out = LN2(h + fc2(u))
  gemm_half(S.P0, B2, ct, kh, j, s, S.acc);
  if (QKV) {  // ... the query weights into the registers of B2
    load_bhalf(Q.Wq, woff, B2);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (grp == 0) {
    const float zero8[8] = {};
    float v[8], o[8], x0[8], x1[8];
    take_acc(S.acc, tl, q, v);
    load8(&S.sH[tl][c0], x0);
    load8(&S.vec[kVB2][c0], x1);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = x0[t] + (v[t] + x1[t]);
    load8(&S.vec[kVG2][c0], x0);
    load8(&S.vec[kVBe2][c0], x1);
    ln_piece(v, q, x0, x1, A.eps2, zero8, zero8, o);
    if (live) {
      *reinterpret_cast<float4*>(A.out + ltok * kD + 8 * q) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(A.out + ltok * kD + 8 * q + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (QKV) {  // the next layer's A tiles: the block's output (rows beyond the batch element's tokens zero), and its modulated form
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = live ? o[t] : 0.0f;
      store_piece(S.P1, tl, 8 * q, o);
      if (QMOD) {
        load8(&S.vec[kVQg][c0], x0);
        load8(&S.vec[kVQh][c0], x1);
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = live ? o[t] * (1.0f + x0[t]) + x1[t] : 0.0f;
        store_piece(S.P0, tl, 8 * q, o);
      }
    }
  }
  MMF_PT(kPtOutFfn, 4, 0.0f);
  if (!QKV) return;
  __syncthreads();
  // ---- the next layer's projections: three GEMMs into three pairs of half tiles, one barrier, then three groups of piece lanes
  if (roles & 2) {
    gemm_half(S.P1, Bo, ct, kh, j, s, S.accK);
    gemm_half(S.P1, B1, ct, kh, j, s, S.accV);
  }
  gemm_half(QMOD ? S.P0 : S.P1, B2, ct, kh, j, s, S.acc);
  __syncthreads();
  MMF_PT(kPtOutFfn, 5, 0.0f);
  if (grp < 3 && (grp == 0 || (roles & 2))) {
    float bias[8], cs[8], sn[8];
    load8(&S.vec[grp == 0 ? kVBq : (grp == 1 ? kVBk : kVBv)][c0], bias);
    load8(&S.tok[kTCos][tl][c0], cs);
    load8(&S.tok[kTSin][tl][c0], sn);
    if (grp == 0)
      role_finish<0, QROT>(S.acc, bias, cs, sn, l0, L, tl, q);
    else if (grp == 1)
      role_finish<1, QROT>(S.accK, bias, cs, sn, l0, L, tl, q);
    else
      role_finish<2, false>(S.accV, bias, cs, sn, l0, L, tl, q);
  }
  __syncthreads();
  if (tid < 512) {
    role_store<0>(S.acc[0], Q, b, l0, L16, tid);
    if (roles & 2) role_store<2>(S.accV[0], Q, b, l0, L16, tid);
  } else if (roles & 2) {
    role_store<1>(S.accK[0], Q, b, l0, L16, tid - 512);
  }
  MMF_PT(kPtOutFfn, 6, 0.0f);
}

// grid = 16-token tiles of the flattened [B L] token axis; 1024 threads
__global__ __launch_bounds__(kNT) void k_out_ffn_mfma(OutFfnArgs A, int L, long long tokens, AttPartials AP) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  out_ffn_tile<false, false, false>(A, (long long)blockIdx.x * 16, tokens, L, S, AP, QkvArgs{}, 0, 0, 0, 0);
}

__global__ __launch_bounds__(kNT) void k_out_ffn_mfma2(OutFfnArgs A0, OutFfnArgs A1, int L, long long tokens, int tiles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  const AttPartials none{nullptr, 0, 0, nullptr, 0u, nullptr};
  if ((int)blockIdx.x >= tiles)
    out_ffn_tile<false, false, false>(A1, (long long)((int)blockIdx.x - tiles) * 16, tokens, L, S, none, QkvArgs{}, 0, 0, 0, 0);
  else
    out_ffn_tile<false, false, false>(A0, (long long)blockIdx.x * 16, tokens, L, S, none, QkvArgs{}, 0, 0, 0, 0);
}

// The tail of layer i and the head of layer i + 1 in one launch: the block on a 16-token tile of ONE batch element, then the
// q | k | v projections of the NEXT layer on the tile's fresh output, which never leaves the workgroup (one kernel boundary and
// one round trip of the activations through memory less per layer).  `tile` = index among the B * L16 / 16 tiles.
// `roles`: 7 = q | k | v; 1 = q alone (the next layer attends to a cached memory).
// `AP`: the attention output arrives as key-split partials (then L <= 16).
__device__ __forceinline__ void out_ffn_qkv_body(const OutFfnArgs& A, const QkvArgs& Q, int L, int L16, int roles, const AttPartials& AP,
                                                 int tile, TileLds& S) {
  const int tpb = L16 / 16;
  const int b = tile / tpb, l0 = (tile % tpb) * 16;
  // rows of the tile beyond the batch element's L tokens are inert: the tile's token range ends at (b + 1) L
  const long long t0 = (long long)b * L + l0, tend = (long long)(b + 1) * L;
  const bool rot = Q.cs != nullptr, mod = Q.ss != nullptr;
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

__global__ __launch_bounds__(kNT) void k_out_ffn_qkv(OutFfnArgs A, QkvArgs Q, int L, int L16, int roles, AttPartials AP) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  out_ffn_qkv_body(A, Q, L, L16, roles, AP, (int)blockIdx.x, S);
}

// two stacks: blocks [0, tiles) serve stack 0, [tiles, 2 tiles) stack 1
__global__ __launch_bounds__(kNT) void k_out_ffn_qkv2(OutFfnArgs A0, OutFfnArgs A1, QkvArgs Q0, QkvArgs Q1, int L, int L16, int tiles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  const AttPartials none{nullptr, 0, 0, nullptr, 0u, nullptr};
  if ((int)blockIdx.x >= tiles)
    out_ffn_qkv_body(A1, Q1, L, L16, 7, none, (int)blockIdx.x - tiles, S);
  else
    out_ffn_qkv_body(A0, Q0, L, L16, 7, none, (int)blockIdx.x, S);
}

// A cross-attention layer of a handful of query rows (Lq <= 16 per batch element) over a long cached context in ONE launch:
//   blocks [0, B H kCrossSplit)   attention: k_attention_heads<16, 3, kCrossSplit> of (batch element, head, key split), the
//                                 partial written as self-validating words {tag | f32 bits}
//   blocks behind them, one per batch element: the block kernel (out_proj + LN + FFN + LN [+ the next layer's queries]), which
//                                 requests its weights and operands, THEN polls the partials of its batch element -- its 2 us of
//                                 start-up run beside the attention instead of behind a kernel boundary.
// The producers never wait and lead the grid (workgroups are dispatched in index order), so a waiting consumer cannot keep a
// producer off the chip; the wait is bounded and raises *fail.  `tag` must differ from every tag the buffer has seen since it was
// last zeroed (the caller counts launches).
constexpr int kCrossSplit = 4;
struct CrossAtt {
  const float *Qp, *Kp, *Vt;
  const uint8_t* pad;
  int Lq, Lk, Lk16;
  float scale;
};
__global__ __launch_bounds__(kNT) void k_cross_layer(OutFfnArgs A, QkvArgs Q, CrossAtt C, unsigned long long* tagged, unsigned tag, int* fail,
                                                    int B, int roles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  const int n_att = B * kH * kCrossSplit;
  if ((int)blockIdx.x < n_att) {
    const int split = (int)blockIdx.x % kCrossSplit, h = ((int)blockIdx.x / kCrossSplit) % kH, b = (int)blockIdx.x / (kCrossSplit * kH);
    att::attention_body<16, 3, kCrossSplit>(C.Qp, C.Kp, C.Vt, C.pad, nullptr, tagged, tag, C.Lq, 16, C.Lk, C.Lk16, C.scale, split, 0, h, b,
                                            *reinterpret_cast<att::AttLds*>(&S));
    return;
  }
  const AttPartials AP{reinterpret_cast<const float*>(tagged), kCrossSplit, C.Lq, tagged, tag, fail};
  const int tile = (int)blockIdx.x - n_att;  // = batch element: its Lq <= 16 rows are one tile
  if (roles == 0) {
    const long long t0 = (long long)tile * C.Lq;
    out_ffn_tile<false, false, false>(A, t0, t0 + C.Lq, C.Lq, S, AP, QkvArgs{}, 0, 0, 0, 0);
  } else {
    out_ffn_qkv_body(A, Q, C.Lq, 16, roles, AP, tile, S);
  }
}

// A SELF-attention layer in one launch, the same way: the attention workgroups lead the grid -- two (query tile, head) units of 8
// waves each per 16-wave workgroup -- and write their output rows as tagged words; the block workgroup of a tile sits behind
// them, requests its weights and operands, and then each of its piece lanes polls the 8 words it owns.
struct SelfAtt {
  const float *Qp, *Kp, *Vt;
  const uint8_t* pad;
  int L, L16;
  float scale;
};
__device__ __forceinline__ void self_layer_attention(const SelfAtt& C, unsigned long long* tagged, unsigned tag, int units, int blk, TileLds& S) {
  const int tiles = C.L16 / 16, half = (int)threadIdx.x >> 9;
  const int u = min(2 * blk + half, units - 1);  // (an odd count: the last unit is computed twice, with identical stores)
  const int tile = u % tiles, h = (u / tiles) % kH, b = u / (tiles * kH);
  att::AttLds* LD = reinterpret_cast<att::AttLds*>(&S) + half;
  att::attention_body<8, 5, 1>(C.Qp, C.Kp, C.Vt, C.pad, nullptr, tagged, tag, C.L, C.L16, C.L, C.L16, C.scale, 0, tile * 16, h, b, *LD, half * 8);
}
__global__ __launch_bounds__(kNT) void k_self_layer(OutFfnArgs A, QkvArgs Q, SelfAtt C, unsigned long long* tagged, unsigned tag, int* fail,
                                                   int B, int roles) {
  __shared__ __attribute__((aligned(16))) TileLds S;
  const int tiles = C.L16 / 16, units = tiles * kH * B, n_att = (units + 1) / 2;
  if ((int)blockIdx.x < n_att) {
    self_layer_attention(C, tagged, tag, units, (int)blockIdx.x, S);
    return;
  }
  const AttPartials AP{nullptr, 0, 0, tagged, tag, fail};
  const int tile = (int)blockIdx.x - n_att;
  if (roles == 0) {
    const int b = tile / tiles, l0 = (tile % tiles) * 16;
    out_ffn_tile<false, false, false>(A, (long long)b * C.L + l0, (long long)(b + 1) * C.L, C.L, S, AP, QkvArgs{}, 0, 0, 0, 0);
  } else {
    out_ffn_qkv_body(A, Q, C.L, C.L16, roles, AP, tile, S);
  }
}

// (The two-stack form of this was built and measured: 624 attention units = 312 workgroups of 1 024 threads + 78 block workgroups
// do not fit the chip at once, the block workgroups -- last in the grid -- start behind the first attention workgroups instead of
// beside them: 23.4 us against 22.4 / 15.9 us for the separate launches.  Removed.)

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
  hipLaunchKernelGGL(k_qkv_heads, dim3(B * (L16 / 16), nroles), dim3(kNT), 0, s, x, Q, L, L16, role0);
  return 0;
}

// q14: {ss, Wq, bq, Wkv, bkv, cs, sn} of stack 0 then of stack 1; Qp / Kp / Vt: stack-major [2, B, H, ...] outputs
int launch_qkv_heads2(const float* x0, const float* x1, const float* const* q14, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H,
                      hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16;
  const size_t half = (size_t)B * kH * L16 * 16;
  const QkvArgs Q0 = qkv_args(q14, Qp, Kp, Vt), Q1 = qkv_args(q14 + 7, Qp + half, Kp + half, Vt + half);
  hipLaunchKernelGGL(k_qkv_heads2, dim3(B * (L16 / 16), 3, 2), dim3(kNT), 0, s, x0, x1, Q0, Q1, L, L16);
  return 0;
}

// a26: OutFfnArgs pointer fields (att .. be2) of stack 0 then of stack 1; out: stack-major [2, B, L, D]
int launch_out_ffn_mfma2(const float* const* a26, const float* eps4, float* out, int B, int L, int D, hipStream_t s) {
  if (D != kD) return 1;
  const long long tokens = (long long)B * L;
  const int tiles = (int)((tokens + 15) / 16);
  const OutFfnArgs A0 = out_ffn_args(a26, eps4[0], eps4[1], out), A1 = out_ffn_args(a26 + 13, eps4[2], eps4[3], out + tokens * kD);
  hipLaunchKernelGGL(k_out_ffn_mfma2, dim3(2 * tiles), dim3(kNT), 0, s, A0, A1, L, tokens, tiles);
  return 0;
}

int launch_out_ffn_mfma(const float* att, const float* res, const float* Wo, const float* bo, const float* g1, const float* be1, float eps1,
                        const float* ss, const float* W1, const float* b1, const float* W2, const float* b2, const float* g2,
                        const float* be2, float eps2, float* out, int B, int L, int D, hipStream_t s) {
  if (D != kD) return 1;
  const long long tokens = (long long)B * L;
  OutFfnArgs A{att, res, W16(Wo), bo, g1, be1, ss, W16(W1), b1, W16(W2), b2, g2, be2, eps1, eps2, out};
  hipLaunchKernelGGL(k_out_ffn_mfma, dim3((unsigned)((tokens + 15) / 16)), dim3(kNT), 0, s, A, L, tokens, AttPartials{nullptr, 0, 0, nullptr, 0u, nullptr});
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
  hipLaunchKernelGGL(k_out_ffn_mfma, dim3((unsigned)((tokens + 15) / 16)), dim3(kNT), 0, s, A, L, tokens, AttPartials{partials, n_split, L, nullptr, 0u, nullptr});
  return 0;
}

// args13: att, res, Wo, bo, g1, be1, ss, W1, b1, W2, b2, g2, be2 (OutFfnArgs order); next7: ss, Wq, bq, Wkv, bkv, cs, sn
int launch_out_ffn_qkv(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp, float* Kp,
                       float* Vt, int B, int L, int D, int H, int roles, const float* partials, int n_split, hipStream_t s) {
  if (D != kD || H != kH || (partials && L > 16)) return 1;
  const int L16 = (L + 15) / 16 * 16;
  const OutFfnArgs A = out_ffn_args(args13, eps1, eps2, out);
  const QkvArgs Q = qkv_args(next7, Qp, Kp, Vt);
  hipLaunchKernelGGL(k_out_ffn_qkv, dim3(B * (L16 / 16)), dim3(kNT), 0, s, A, Q, L, L16, roles, AttPartials{partials, n_split, L, nullptr, 0u, nullptr});
  return 0;
}

// args13 / next7 as launch_out_ffn_qkv (args13[0] unused; next7 null: no projection, roles 0); qkv3: the layer's head-major q,
// the context's cached k, v (launch_qkv_heads roles 1 / 6); tagged: [B, H, 4, 18, 16] 64-bit words, zeroed once
int launch_cross_layer(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp_next,
                       const float* const* qkv3, const uint8_t* pad, unsigned long long* tagged, unsigned tag, int* fail, int B, int Lq,
                       int Lk, int D, int H, hipStream_t s) {
  if (D != kD || H != kH || Lq > 16 || Lq < 1) return 1;
  const OutFfnArgs A = out_ffn_args(args13, eps1, eps2, out);
  QkvArgs Q{};
  if (next7) Q = qkv_args(next7, Qp_next, nullptr, nullptr);
  const CrossAtt C{qkv3[0], qkv3[1], qkv3[2], pad, Lq, Lk, (Lk + 15) / 16 * 16, 1.0f / sqrtf((float)kDH)};
  hipLaunchKernelGGL(k_cross_layer, dim3(B * kH * kCrossSplit + B), dim3(kNT), 0, s, A, Q, C, tagged, tag, fail, B, next7 ? 1 : 0);
  return 0;
}

// A self-attention layer: args13 / next7 as launch_out_ffn_qkv (args13[0] unused; next7 null: no projections), qkv3: this
// layer's head-major q, k, v; Qp / Kp / Vt: the next layer's; tagged: [B, L, D] 64-bit words, zeroed once
int launch_self_layer(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp, float* Kp, float* Vt,
                      const float* const* qkv3, const uint8_t* pad, unsigned long long* tagged, unsigned tag, int* fail, int B, int L, int D,
                      int H, hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16, tiles = L16 / 16;
  const OutFfnArgs A = out_ffn_args(args13, eps1, eps2, out);
  QkvArgs Q{};
  if (next7) Q = qkv_args(next7, Qp, Kp, Vt);
  const SelfAtt C{qkv3[0], qkv3[1], qkv3[2], pad, L, L16, 1.0f / sqrtf((float)kDH)};
  hipLaunchKernelGGL(k_self_layer, dim3((tiles * kH * B + 1) / 2 + tiles * B), dim3(kNT), 0, s, A, Q, C, tagged, tag, fail, B, next7 ? 7 : 0);
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
  hipLaunchKernelGGL(k_out_ffn_qkv2, dim3(2 * tiles), dim3(kNT), 0, s, A0, A1, Q0, Q1, L, L16, tiles);
  return 0;
}

}  // namespace mmf
