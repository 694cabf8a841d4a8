// Matrix-core (MFMA) forms of the per-layer inference kernels of the diffusion head (mmf_kernels_policy.hip holds the
// thread-per-channel forms and everything that is not GEMM-shaped).
//
// A denoising step works on ~616 tokens x 120 channels, 8 heads of 15 channels.  All three kernels use the f32-input
// v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation: an fmaf chain) on tiles of 16 tokens:
//
//   operand maps (lane l, j = l & 15, s = l >> 4):   A[i = j][k = s]   B[k = s][col = j]   D[row = 4 s + r][col = j], r = 0..3
//
// The reduction index is free to be permuted as long as A and B agree, so every lane loads its share of an operand row
// as ONE 16-byte piece: step (m, kk) of a 120-channel reduction is channel 16 m + 4 s + kk (k_qkv_heads, k_out_ffn_mfma),
// and the 16-key (16-channel) reductions of the attention kernel use key (channel) 4 s + kk.
//
//   k_qkv_heads       q = rotary(q_proj(modulated x)) | k = rotary(k_proj(x)) | v = v_proj(x), written HEAD-MAJOR and padded
//                     to 16 channels: Qp, Kp [B, H, L16, 16], Vt [B, H, 16, L16] (values transposed) -- the layouts from which
//                     k_attention_heads loads every operand with one aligned 16-byte access per lane
//   k_attention_heads softmax(q k^T / sqrt(dh) + padding) v: scores are produced TRANSPOSED (S^T = K Q^T), which leaves each
//                     lane holding, for its query row, exactly the four probabilities the B operand of O^T = V^T P^T needs,
//                     and makes the softmax statistics of a query row lane-aligned with the columns of O^T: no transposes, no
//                     LDS, two shuffles per reduction.  The keys of a (query tile, head) are split over the 4 waves of the
//                     workgroup and merged once through LDS
//   k_out_ffn_mfma    x1 = LN(res + out_proj(att)); h = modulate(x1); out = LN(h + fc2(relu(fc1(h)))): three chained GEMMs on a
//                     16-token tile, the tile moves between them through LDS (D layout -> A layout)
#include "mmf_device.h"
#include "mmf_launch.h"
#include "mmf_trace_device.h"

namespace mmf {

// Phase marks of the instrumented build (`make WG_TRACE=1`, tools/policy_phase_trace.py): thread 0 of a workgroup stores the
// 100 MHz wall clock at up to 8 points, behind the 6 x 8192 frame records of the trace buffer, at record (8 u64) `base` +
// linear workgroup index.  `dep` pins the mark behind the value's producer.  Compiled out of the default build.
#ifdef MMF_WG_TRACE
constexpr int kPtQkv = 0, kPtAtt = 256, kPtOutFfn = 1024;
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int kD = 120, kH = 8, kDH = 15;  // the policy's embedding dim / heads / head dim (the kernels are built for these)
constexpr int kKSteps = 32;                 // 8 x 4 reduction steps cover 128 >= 120 channels

// All loads below are UNCONDITIONAL from clamped (always valid) addresses, with a select on the value where it matters: a
// guarded load compiles to a branch plus register copies, and these kernels run once through straight-line code fetched
// through a cold instruction cache -- their duration follows their code size.

// Loads are written REQUEST FIRST, USE LATER: the compiler keeps the program order of loads and puts a wait in front of the
// first use, and waits count in issue order -- a select on a loaded value between two groups of loads makes the second group a
// second round trip (1.1 us here: the first touch of a kernel goes to HBM), and a value requested late but used early drags
// every earlier request into its wait.  So: every kernel issues its loads in the order of their first use, all of them before
// the first use of any.

// Offset of the 16-byte piece of reduction steps 4 m .. 4 m + 3 in a 120-float row: 16 m + 4 s; the pieces of m = 7, s >= 2
// lie beyond D (their A values are zero) and are read from 112 + 4 (s & 1) instead.  (Written as arithmetic: from
// `min(c0, D - 4)` the compiler builds a two-way select of loaded vectors that it indexes through scratch memory.)
__device__ __forceinline__ int piece_offset(int m, int s) { return 16 * m + 4 * (m < 7 ? s : (s & 1)); }

// A-operand share of one token row: a[4 m + kk] = row[16 m + 4 s + kk] (0 beyond D, 0 if !ok; `row` must be readable)
struct RowRaw {
  float4 v[8];
};
__device__ __forceinline__ RowRaw load_row_raw(const float* __restrict__ row, int s) {
  RowRaw R;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    R.v[m] = *reinterpret_cast<const float4*>(row + piece_offset(m, s));
  }
  return R;
}
__device__ __forceinline__ void row_share(const RowRaw& R, int s, bool ok, float (&a)[kKSteps]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const bool keep = ok && (m < 7 || 16 * m + 4 * s < kD);
    a[4 * m] = keep ? R.v[m].x : 0.0f;
    a[4 * m + 1] = keep ? R.v[m].y : 0.0f;
    a[4 * m + 2] = keep ? R.v[m].z : 0.0f;
    a[4 * m + 3] = keep ? R.v[m].w : 0.0f;
  }
}

// B-operand share of one output column: w[4 m + kk] = W[col][16 m + 4 s + kk] -- W = the Linear's own [out, in] weight matrix,
// so a lane's share of a reduction step is ONE aligned 16-byte piece (the transposed [in, out] layout needs four 4-byte loads
// for it: 64 instead of 16 load instructions per wave and tile, and the load unit's issue rate, not the latency, set the
// time to the first MFMA).  `col` must be a valid row of W; pieces beyond D are read from the row's last 16 bytes: their A
// values are zero.
__device__ __forceinline__ void load_col_share(const float* __restrict__ W, int col, int s, float (&w)[kKSteps]) {
  const float* row = W + (size_t)col * kD;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float4 v = *reinterpret_cast<const float4*>(row + piece_offset(m, s));
    w[4 * m] = v.x, w[4 * m + 1] = v.y, w[4 * m + 2] = v.z, w[4 * m + 3] = v.w;
  }
}

__device__ __forceinline__ f32x4 tile_gemm(const float (&a)[kKSteps], const float (&w)[kKSteps]) {
  // two accumulators: the dependent-accumulator latency of the 16x16x4 form (40 cycles) exceeds its issue interval (32)
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < kKSteps; i += 2) {
    c0 = mfma4(a[i], w[i], c0);
    c1 = mfma4(a[i + 1], w[i + 1], c1);
  }
  return c0 + c1;
}

// ---- q | k | v projections, rotary, head-major outputs ---------------------------------------------------------------------
struct QkvArgs {
  const float *ss, *Wq, *bq, *Wkv, *bkv, *cs, *sn;  // Wq [D, D], Wkv [2 D, D]: the Linears' [out, in] weights; ss: AdaLN (scale | shift) [B, 2 D] of the query input or null; cs / sn [B, L, D] or null
  float *Qp, *Kp, *Vt;
};

// AdaLN modulation of an A-operand share: a <- a (1 + scale) + shift with the lane's pieces of (scale | shift) [2 D]
struct ModRaw {
  float4 g[8], h[8];
};
__device__ __forceinline__ ModRaw load_mod_raw(const float* __restrict__ sc, int s) {  // sc: 2 D readable floats
  ModRaw M;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int c0 = piece_offset(m, s);
    M.g[m] = *reinterpret_cast<const float4*>(sc + c0);
    M.h[m] = *reinterpret_cast<const float4*>(sc + kD + c0);
  }
  return M;
}
__device__ __forceinline__ void modulate_share(const ModRaw& M, int s, bool apply, float (&a)[kKSteps]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const bool ap = apply && (m < 7 || 16 * m + 4 * s < kD);  // pieces beyond D stay zero
    a[4 * m] = ap ? a[4 * m] * (1.0f + M.g[m].x) + M.h[m].x : a[4 * m];
    a[4 * m + 1] = ap ? a[4 * m + 1] * (1.0f + M.g[m].y) + M.h[m].y : a[4 * m + 1];
    a[4 * m + 2] = ap ? a[4 * m + 2] * (1.0f + M.g[m].z) + M.h[m].z : a[4 * m + 2];
    a[4 * m + 3] = ap ? a[4 * m + 3] * (1.0f + M.g[m].w) + M.h[m].w : a[4 * m + 3];
  }
}

// One role (0 = q, 1 = k, 2 = v) of a 16-token tile (batch element b, first token l0) for the wave's two heads 2 w, 2 w + 1
// (30 channels: rotary pairs never leave the wave); `a` = the wave's A-operand share of the (modulated / raw) input rows.
// Lanes of the padding column (j = 15) and of rows beyond L compute on clamped addresses; their results are zeroed at the store.
// ROLE / ROT (rotary tables present) / MOD (AdaLN on the query input) are TEMPLATE parameters and the kernels branch ONCE, at
// the top, into a straight-line body: a run-time condition around a group of loads makes the compiler split the request
// sequence at the branch, wait there, and start a second round trip behind it.
struct QkvOps {  // the role's operands: weights of the two heads' columns, bias, rotary cos / sin of the lane's outputs
  float wv[2][kKSteps];
  float bb[2], cv[2][4], sv[2][4];
};
template <int ROLE>
__device__ __forceinline__ const float* qkv_role_weights(const QkvArgs& Q) {
  return ROLE == 0 ? Q.Wq : Q.Wkv + (ROLE == 2 ? (size_t)kD * kD : 0);  // the value projection: rows D .. 2 D - 1 of Wkv
}
template <int ROLE, bool ROT>
__device__ __forceinline__ void qkv_role_loads(const QkvArgs& Q, int b, int l0, int L, int w, int j, int s, QkvOps& O) {
  const float* W = qkv_role_weights<ROLE>(Q);
  const float* bias = ROLE == 0 ? Q.bq : Q.bkv + (ROLE == 2 ? kD : 0);
  const int jc = min(j, kDH - 1);
#pragma unroll
  for (int n = 0; n < 2; ++n) load_col_share(W, kDH * (2 * w + n) + jc, s, O.wv[n]);
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
__device__ __forceinline__ void qkv_role_compute(const float (&a)[kKSteps], const QkvOps& O, const QkvArgs& Q, int b, int l0, int L,
                                                 int L16, int w, int j, int s) {
  MMF_PT(kPtQkv, 2, O.wv[0][31]);
  f32x4 y[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    y[n] = tile_gemm(a, O.wv[n]);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[n][r] += O.bb[n];  // (element by element: `vector += scalar` keeps the operand block in scratch)
  }
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

// a role on an A-operand share that is already in registers (the tile of k_out_ffn_qkv); MOD: modulate it first (in place)
template <int ROLE, bool ROT, bool MOD>
__device__ __forceinline__ void qkv_role_tile(float (&a)[kKSteps], const QkvArgs& Q, int b, int l0, int L, int L16, int w, int j, int s) {
  QkvOps O;
  qkv_role_loads<ROLE, ROT>(Q, b, l0, L, w, j, s, O);
  if (MOD) {
    const ModRaw M = load_mod_raw(Q.ss + (size_t)b * 2 * kD, s);
    __builtin_amdgcn_sched_barrier(0);
    modulate_share(M, s, l0 + j < L, a);
  } else {
    __builtin_amdgcn_sched_barrier(0);
  }
  qkv_role_compute<ROLE, ROT>(a, O, Q, b, l0, L, L16, w, j, s);
}

// One (tile, role) of k_qkv_heads / k_qkv_heads2: every load is requested before the first use (weights first: the largest)
template <int ROLE, bool ROT, bool MOD>
__device__ __forceinline__ void qkv_heads_tile(const float* __restrict__ x, const QkvArgs& Q, int L, int L16) {
  const int tpb = L16 / 16;
  const int b = (int)blockIdx.x / tpb, l0 = ((int)blockIdx.x % tpb) * 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, s = lane >> 4;
  const int tok = l0 + j;
  const bool ok = tok < L;
  MMF_PT(kPtQkv, 0, 0.0f);
  QkvOps O;
  qkv_role_loads<ROLE, ROT>(Q, b, l0, L, w, j, s, O);
  const RowRaw R = load_row_raw(x + ((size_t)b * L + min(tok, L - 1)) * kD, s);
  float a[kKSteps];
  if (MOD) {  // AdaLN modulation of the query input
    const ModRaw M = load_mod_raw(Q.ss + (size_t)b * 2 * kD, s);
    __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks each load to its use: one round trip per MFMA group
    row_share(R, s, ok, a);
    modulate_share(M, s, ok, a);
  } else {
    __builtin_amdgcn_sched_barrier(0);
    row_share(R, s, ok, a);
  }
  MMF_PT(kPtQkv, 1, a[31]);
  qkv_role_compute<ROLE, ROT>(a, O, Q, b, l0, L, L16, w, j, s);
  MMF_PT(kPtQkv, 4, 0.0f);
}

__device__ __forceinline__ void qkv_heads_body(const float* __restrict__ x, const QkvArgs& Q, int L, int L16, int role) {
  const bool rot = Q.cs != nullptr, mod = Q.ss != nullptr;
  if (role == 0) {
    if (rot) {
      if (mod)
        qkv_heads_tile<0, true, true>(x, Q, L, L16);
      else
        qkv_heads_tile<0, true, false>(x, Q, L, L16);
    } else {
      if (mod)
        qkv_heads_tile<0, false, true>(x, Q, L, L16);
      else
        qkv_heads_tile<0, false, false>(x, Q, L, L16);
    }
  } else if (role == 1) {
    if (rot)
      qkv_heads_tile<1, true, false>(x, Q, L, L16);
    else
      qkv_heads_tile<1, false, false>(x, Q, L, L16);
  } else {
    qkv_heads_tile<2, false, false>(x, Q, L, L16);
  }
}

// grid (B * L16 / 16, roles), 256 threads
__global__ __launch_bounds__(256) void k_qkv_heads(const float* __restrict__ x, QkvArgs Q, int L, int L16, int role0) {
  qkv_heads_body(x, Q, L, L16, role0 + (int)blockIdx.y);  // 0 = q, 1 = k, 2 = v
}

// ---- attention over head-major operands -------------------------------------------------------------------------------------
// CH = key tiles a wave scores before it runs the softmax update (4 CH score registers + 8 CH operand registers)

// NW waves per workgroup share the keys of one (query tile, head): 4 for self-attention over a few hundred keys, 16 when a
// handful of query rows attend to thousands of keys (the trajectory tokens over the full context).
// SPLIT > 1 (one query tile per batch element, i.e. Lq <= 16): the keys are additionally divided among SPLIT WORKGROUPS, each
// of which leaves an un-normalised partial result {O^T [16 channels][16 rows], running maximum [16], sum [16]} in `out`
// ([B, H, SPLIT, 18, 16] floats) -- k_out_ffn_mfma merges them while it loads its input tile.  A cross-workgroup merge inside
// this kernel would need a device-scope release per workgroup; the kernel boundary that follows anyway is free.
constexpr int kPartRows = 18;  // rows of one partial: 16 channels of O^T, then the maxima, then the sums

template <int NW, int CH, int SPLIT>
__global__ __launch_bounds__(64 * NW) void k_attention_heads(const float* __restrict__ Qp, const float* __restrict__ Kp,
                                                            const float* __restrict__ Vt, const uint8_t* __restrict__ pad,
                                                            float* __restrict__ out, int Lq, int Lq16, int Lk, int Lk16, float scale) {
  __shared__ float sM[NW][16], sL[NW][16];
  __shared__ float sO[NW][16][17];
  const int split = SPLIT > 1 ? (int)blockIdx.x : 0;
  const int q0 = SPLIT > 1 ? 0 : (int)blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, s = lane >> 4;
  const size_t bh = (size_t)b * kH + h;

  MMF_PT(kPtAtt, 0, 0.0f);
  float q[4];
  {
    const float4 v = *reinterpret_cast<const float4*>(Qp + (bh * Lq16 + q0 + j) * 16 + 4 * s);
    q[0] = v.x * scale, q[1] = v.y * scale, q[2] = v.z * scale, q[3] = v.w * scale;
  }
  const float* Kb = Kp + bh * Lk16 * 16;
  const float* Vb = Vt + (bh * 16 + j) * Lk16;
  // key padding: [B, Lk16] bytes (1 = ignore; keys >= Lk are marked too), one aligned 32-bit word per (tile, lane)
  const uint8_t* pb = pad ? pad + (size_t)b * Lk16 : nullptr;

  float m_run = -INFINITY, l_run = 0.0f;
  f32x4 O0 = {0.f, 0.f, 0.f, 0.f}, O1 = {0.f, 0.f, 0.f, 0.f};  // O^T: rows = channel 4 s + r, column = query row j
  const int all_tiles = Lk16 / 16, per_split = (all_tiles + SPLIT - 1) / SPLIT;
  const int t_begin = split * per_split, ntiles = min(t_begin + per_split, all_tiles);  // this workgroup's key tiles
  for (int tb = t_begin + w; tb < ntiles; tb += NW * CH) {  // this wave's tiles: tb, tb + NW, ...
    float4 kv[CH], vv[CH];
    uint32_t pw[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int t = tb + NW * i, tc = min(t, all_tiles - 1);  // tiles beyond the end: a valid tile's data, every key marked dead
      kv[i] = *reinterpret_cast<const float4*>(Kb + ((size_t)tc * 16 + j) * 16 + 4 * s);
      vv[i] = *reinterpret_cast<const float4*>(Vb + tc * 16 + 4 * s);
      uint32_t word = 0u;
      if (pb) word = *reinterpret_cast<const uint32_t*>(pb + tc * 16 + 4 * s);
      pw[i] = t < ntiles ? word : 0xffffffffu;
    }
    MMF_PT(kPtAtt, 1, kv[CH - 1].x + vv[CH - 1].x + q[0]);
    f32x4 S[CH];
    float cmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = mfma4(kv[i].x, q[0], c);
      c = mfma4(kv[i].y, q[1], c);
      c = mfma4(kv[i].z, q[2], c);
      c = mfma4(kv[i].w, q[3], c);
      const int t = tb + NW * i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
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
    O0 *= corr;
    O1 *= corr;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = (S[i][r] == -INFINITY) ? 0.0f : __expf(S[i][r] - m_new);
        l_run += p[r];
      }
      if (i & 1) {
        O1 = mfma4(vv[i].x, p[0], O1);
        O1 = mfma4(vv[i].y, p[1], O1);
        O1 = mfma4(vv[i].z, p[2], O1);
        O1 = mfma4(vv[i].w, p[3], O1);
      } else {
        O0 = mfma4(vv[i].x, p[0], O0);
        O0 = mfma4(vv[i].y, p[1], O0);
        O0 = mfma4(vv[i].z, p[2], O0);
        O0 = mfma4(vv[i].w, p[3], O0);
      }
    }
    m_run = m_new;
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
      if (SPLIT > 1)
        part[ch * 16 + j] = o;
      else if (ch < kDH && row < Lq)
        out[((size_t)b * Lq + row) * kD + h * kDH + ch] = o / l;
    }
    if (SPLIT > 1 && s == 0) {
      part[16 * 16 + j] = M;
      part[17 * 16 + j] = l;
    }
  }
  MMF_PT(kPtAtt, 5, 0.0f);
}

// ---- out_proj + LayerNorm + feed-forward block -----------------------------------------------------------------------------
constexpr int kRS = 132;  // LDS row stride of a 16-token tile (floats)

// LayerNorm of the 16 x 120 tile in `src` (wave w: tokens 4 w .. 4 w + 3, 16 lanes per token; lane q < 15 owns the 8 adjacent
// channels 8 q .. 8 q + 7, lane 15 idles): every operand of a lane -- gamma / beta / (scale, shift) / the residual row, fetched
// at kernel start -- and its LDS traffic are 16-byte pieces.  `add`: a second summand per channel (the residual input, or
// zeros).  The result (optionally AdaLN-modulated) goes to `dst` (LDS) and / or `gout` (global, row stride D)
struct LnShare {
  float g[8], b[8];
};
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
  v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
}
__device__ __forceinline__ LnShare load_ln_share(const float* __restrict__ gamma, const float* __restrict__ beta, int q) {
  LnShare P;
  const int c0 = 8 * min(q, 14);  // lane 15 reads lane 14's pieces: the values are not used
  load8(gamma + c0, P.g);
  load8(beta + c0, P.b);
  return P;
}

__device__ __forceinline__ void tile_layer_norm(const float (*src)[kRS], float (*dst)[kRS], float* __restrict__ gout, long long t0,
                                                long long tokens, const LnShare& P, float eps, bool modulate, const float (&sc)[8],
                                                const float (&sh)[8], const float (&add)[8], int lane, int w) {
  const int tl = 4 * w + (lane >> 4), q = lane & 15;
  const bool own = q < 15;
  const int c0 = 8 * min(q, 14);
  float v[8];
  {
    const float4 lo = *reinterpret_cast<const float4*>(&src[tl][c0]), hi = *reinterpret_cast<const float4*>(&src[tl][c0 + 4]);
    v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
  }
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
  const long long tok = t0 + tl;
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    o[i] = v[i] * inv * P.g[i] + P.b[i];
    if (modulate) o[i] = o[i] * (1.0f + sc[i]) + sh[i];
  }
  if (own) {
    if (dst) {
      *reinterpret_cast<float4*>(&dst[tl][c0]) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(&dst[tl][c0 + 4]) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (gout && tok < tokens) {
      *reinterpret_cast<float4*>(gout + tok * kD + c0) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(gout + tok * kD + c0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

__device__ __forceinline__ void lds_row_share(const float (*src)[kRS], int j, int s, float (&a)[kKSteps]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int c0 = 16 * m + 4 * s;
    const float4 v = *reinterpret_cast<const float4*>(&src[j][c0]);  // c0 <= 124: inside the 132-float row
    const bool keep = m < 7 || c0 < kD;
    a[4 * m] = keep ? v.x : 0.0f;
    a[4 * m + 1] = keep ? v.y : 0.0f;
    a[4 * m + 2] = keep ? v.z : 0.0f;
    a[4 * m + 3] = keep ? v.w : 0.0f;
  }
}

struct AttPartials {  // k_attention_heads<.., SPLIT > 1> output to merge instead of reading `att` (null: read att)
  const float* part;  // [B, H, n_split, 18, 16]
  int n_split, Lq;    // Lq <= 16 query rows per batch element
};

struct OutFfnArgs {
  const float *att, *res, *Wo, *bo, *g1, *be1, *ss, *W1, *b1, *W2, *b2, *g2, *be2;  // ss: AdaLN (scale | shift) of the FFN or null
  float eps1, eps2;
  float* out;
};

// x1 = LN1(res + out_proj(att)); h = modulate(x1); out = LN2(h + fc2(relu(fc1(h)))) for the 16 tokens t0 .. t0 + 15 of the
// flattened [B L] token axis (`tokens` = B L; rows beyond it are inert); wave w owns output columns [32 w, 32 w + 32).
// `keep`: where the result additionally stays in LDS (rows of invalid tokens zeroed) for a consumer in the same kernel.
__device__ __forceinline__ void out_ffn_tile(const OutFfnArgs& A, long long t0, long long tokens, int L, float (*sH)[kRS], float (*sU)[kRS],
                                             float (*sY)[kRS], float (*keep)[kRS], int lane, int w, int j, int s,
                                             const AttPartials& AP = AttPartials{nullptr, 0, 0}) {
  MMF_PT(kPtOutFfn, 0, 0.0f);
  // Requests in the order of first use (see the note at load_row_raw), all before the first use: out_proj weights and the input
  // rows | LayerNorm 1 operands | fc1 weights | fc2 weights | LayerNorm 2 operands.  The wave runs alone on its SIMD: 192
  // registers of weights are free, and nothing but LDS traffic and MFMAs sits between the barriers below.
  // (Output columns 120 .. 127 -- the last 8 lanes of wave 3's second tile -- compute on row 119's operands; their results land
  // in LDS columns that are never read.)
  const int colc[2] = {min(32 * w + j, kD - 1), min(32 * w + 16 + j, kD - 1)};
  float wo[2][kKSteps], w1[2][kKSteps], w2[2][kKSteps];
#pragma unroll
  for (int n = 0; n < 2; ++n) load_col_share(A.Wo, colc[n], s, wo[n]);
  const long long atok = t0 + j;
  RowRaw R;
  if (AP.part == nullptr) R = load_row_raw(A.att + min(atok, tokens - 1) * kD, s);
  float bbo[2], bb1[2], bb2[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) bbo[n] = A.bo[colc[n]];
  const int ln_q = lane & 15, ln_c0 = 8 * min(ln_q, 14);
  const LnShare P1 = load_ln_share(A.g1, A.be1, ln_q);
  float sc[8], sh[8], rs[8];
  const float zero8[8] = {};
  const long long ltok = t0 + 4 * w + (lane >> 4), ltokc = min(ltok, tokens - 1);
  load8(A.res + ltokc * kD + ln_c0, rs);  // the residual row of this lane's LayerNorm token
  {
    const float* sp = A.ss != nullptr ? A.ss + (size_t)((int)ltokc / L) * 2 * kD : A.g1;  // (no AdaLN: any readable floats, zeroed below)
    load8(sp + ln_c0, sc);
    load8(sp + (A.ss != nullptr ? kD : 0) + ln_c0, sh);
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) load_col_share(A.W1, colc[n], s, w1[n]);
#pragma unroll
  for (int n = 0; n < 2; ++n) bb1[n] = A.b1[colc[n]];
#pragma unroll
  for (int n = 0; n < 2; ++n) load_col_share(A.W2, colc[n], s, w2[n]);
#pragma unroll
  for (int n = 0; n < 2; ++n) bb2[n] = A.b2[colc[n]];
  const LnShare P2 = load_ln_share(A.g2, A.be2, ln_q);
  __builtin_amdgcn_sched_barrier(0);  // every request above is issued before anything below (the scheduler sinks loads to uses)

  float a[kKSteps];
  if (AP.part != nullptr) {
    // the attention output of this tile, merged from the key splits: element (token, channel c = 15 h + ch) =
    // sum_sp e^(m_sp - M) O_sp[ch][row] / sum_sp e^(m_sp - M) l_sp, M = max_sp m_sp
    for (int e = threadIdx.x; e < 16 * 128; e += 256) {
      const int tl = e >> 7, c = e & 127;
      const long long tok = t0 + tl;
      float v = 0.0f;
      if (c < kD && tok < tokens) {
        const int b = (int)(tok / AP.Lq), row = (int)(tok - (long long)b * AP.Lq), h = c / kDH, ch = c - h * kDH;
        const float* P = AP.part + ((size_t)b * kH + h) * AP.n_split * kPartRows * 16;
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
      sU[tl][c] = v;
    }
    __syncthreads();
    lds_row_share(sU, j, s, a);
    __syncthreads();  // sU is reused below
  } else {
    row_share(R, s, atok < tokens, a);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (ltok >= tokens) rs[i] = 0.0f;
    if (A.ss == nullptr) sc[i] = 0.0f, sh[i] = 0.0f;  // x (1 + 0) + 0 = x exactly
  }
  // ---- x1 = LN1(res + out_proj(att)), h = modulate(x1)
  MMF_PT(kPtOutFfn, 1, a[31] + wo[1][31]);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = 32 * w + 16 * n + j;
    const f32x4 y = tile_gemm(a, wo[n]);
#pragma unroll
    for (int r = 0; r < 4; ++r) sY[4 * s + r][col] = y[r] + bbo[n];
    if (n == 1) MMF_PT(kPtOutFfn, 2, y[0]);
  }
  __syncthreads();
  tile_layer_norm(sY, sH, nullptr, t0, tokens, P1, A.eps1, true, sc, sh, rs, lane, w);
  __syncthreads();
  // ---- u = relu(fc1(h))
  lds_row_share(sH, j, s, a);
  MMF_PT(kPtOutFfn, 3, a[31] + w1[1][31]);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = 32 * w + 16 * n + j;
    const f32x4 y = tile_gemm(a, w1[n]);
#pragma unroll
    for (int r = 0; r < 4; ++r) sU[4 * s + r][col] = fmaxf(y[r] + bb1[n], 0.0f);
    if (n == 1) MMF_PT(kPtOutFfn, 4, y[0]);
  }
  __syncthreads();
  // ---- out = LN2(h + fc2(u))
  lds_row_share(sU, j, s, a);
  MMF_PT(kPtOutFfn, 5, a[31] + w2[1][31]);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = 32 * w + 16 * n + j;
    const f32x4 y = tile_gemm(a, w2[n]);
#pragma unroll
    for (int r = 0; r < 4; ++r) sY[4 * s + r][col] = sH[4 * s + r][col] + (y[r] + bb2[n]);
    if (n == 1) MMF_PT(kPtOutFfn, 6, y[0]);
  }
  __syncthreads();  // also: every read of sH / sU above is done, `keep` may alias them
  tile_layer_norm(sY, keep, A.out, t0, tokens, P2, A.eps2, false, sc, sh, zero8, lane, w);
  MMF_PT(kPtOutFfn, 7, 0.0f);
}

// grid = 16-token tiles of the flattened [B L] token axis
__global__ __launch_bounds__(256) void k_out_ffn_mfma(OutFfnArgs A, int L, long long tokens, AttPartials AP) {
  __shared__ __attribute__((aligned(16))) float sH[16][kRS];  // h = modulate(LN1(..)): A operand of fc1 and residual of fc2
  __shared__ __attribute__((aligned(16))) float sU[16][kRS];  // relu(fc1(h)): A operand of fc2
  __shared__ __attribute__((aligned(16))) float sY[16][kRS];  // pre-LayerNorm sums
  const int lane = threadIdx.x & 63;
  out_ffn_tile(A, (long long)blockIdx.x * 16, tokens, L, sH, sU, sY, nullptr, lane, threadIdx.x >> 6, lane & 15, lane >> 4, AP);
}

// Two independent stacks of identical shape (the rotation and the position stack of the diffusion head) in ONE launch:
// blockIdx.z / the upper half of blockIdx.x selects the stack's operands.  Activations and head-major outputs are stack-major
// ([2, B, ...]), so the attention kernel sees the pair as a batch of 2 B.
// (The stack's argument block is chosen by a BRANCH around two calls, not by `second ? Q1 : Q0`: a reference picked at run
// time makes the compiler copy the 168-byte block to scratch memory and read every pointer back from there.)
__global__ __launch_bounds__(256) void k_qkv_heads2(const float* __restrict__ x0, const float* __restrict__ x1, QkvArgs Q0, QkvArgs Q1,
                                                   int L, int L16) {
  const int role = (int)blockIdx.y;  // 0 = q, 1 = k, 2 = v
  if (blockIdx.z != 0)
    qkv_heads_body(x1, Q1, L, L16, role);
  else
    qkv_heads_body(x0, Q0, L, L16, role);
}

__global__ __launch_bounds__(256) void k_out_ffn_mfma2(OutFfnArgs A0, OutFfnArgs A1, int L, long long tokens, int tiles) {
  __shared__ __attribute__((aligned(16))) float sH[16][kRS];
  __shared__ __attribute__((aligned(16))) float sU[16][kRS];
  __shared__ __attribute__((aligned(16))) float sY[16][kRS];
  const bool second = (int)blockIdx.x >= tiles;
  const OutFfnArgs& A = second ? A1 : A0;
  const int lane = threadIdx.x & 63;
  out_ffn_tile(A, (long long)((int)blockIdx.x - (second ? tiles : 0)) * 16, tokens, L, sH, sU, sY, nullptr, lane, threadIdx.x >> 6, lane & 15,
               lane >> 4);
}

// The tail of layer i and the head of layer i + 1 in one launch: k_out_ffn_mfma on a 16-token tile of ONE batch element, then
// the q | k | v projections of the NEXT layer on the tile's fresh output, which never leaves the workgroup (one kernel boundary
// and one round trip of the activations through memory less per layer).  grid (B * L16 / 16), 256 threads.
// `roles`: which of the next layer's projections (bit 0 q, 1 k, 2 v); q alone when the next layer attends to a cached memory.
// `AP`: the attention output arrives as key-split partials (then L <= 16).
__global__ __launch_bounds__(256) void k_out_ffn_qkv(OutFfnArgs A, QkvArgs Q, int L, int L16, int roles, AttPartials AP) {
  __shared__ __attribute__((aligned(16))) float sH[16][kRS];
  __shared__ __attribute__((aligned(16))) float sU[16][kRS];
  __shared__ __attribute__((aligned(16))) float sY[16][kRS];
  const int tpb = L16 / 16;
  const int b = (int)blockIdx.x / tpb, l0 = ((int)blockIdx.x % tpb) * 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, s = lane >> 4;
  // rows of the tile beyond the batch element's L tokens are inert: the tile's token range ends at (b + 1) L
  out_ffn_tile(A, (long long)b * L + l0, (long long)(b + 1) * L, L, sH, sU, sY, sU, lane, w, j, s, AP);
  __syncthreads();
  float a[kKSteps];
  lds_row_share(sU, j, s, a);
  if (l0 + j >= L) {
#pragma unroll
    for (int i = 0; i < kKSteps; ++i) a[i] = 0.0f;
  }
  const bool rot = Q.cs != nullptr, mod = Q.ss != nullptr;
  if (rot) {
    if (roles & 2) qkv_role_tile<1, true, false>(a, Q, b, l0, L, L16, w, j, s);
    if (roles & 4) qkv_role_tile<2, false, false>(a, Q, b, l0, L, L16, w, j, s);
    if (roles & 1) {
      if (mod)
        qkv_role_tile<0, true, true>(a, Q, b, l0, L, L16, w, j, s);
      else
        qkv_role_tile<0, true, false>(a, Q, b, l0, L, L16, w, j, s);
    }
  } else {
    if (roles & 2) qkv_role_tile<1, false, false>(a, Q, b, l0, L, L16, w, j, s);
    if (roles & 4) qkv_role_tile<2, false, false>(a, Q, b, l0, L, L16, w, j, s);
    if (roles & 1) {
      if (mod)
        qkv_role_tile<0, false, true>(a, Q, b, l0, L, L16, w, j, s);
      else
        qkv_role_tile<0, false, false>(a, Q, b, l0, L, L16, w, j, s);
    }
  }
}

MMF_DEFINE_WG_TRACE_SETTER(set_wg_trace_policy)

// ---- launchers ----------------------------------------------------------------------------------------------------------------
int launch_qkv_heads(const float* x, const float* ss, const float* Wq, const float* bq, const float* Wkv, const float* bkv,
                     const float* cs, const float* sn, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H, int roles,
                     hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16;
  // roles: 7 = q | k | v (self-attention), 1 = q alone, 6 = k | v alone (a memory whose keys / values are cached)
  const int role0 = (roles & 1) ? 0 : 1, nroles = roles == 7 ? 3 : (roles == 1 ? 1 : 2);
  QkvArgs Q{ss, Wq, bq, Wkv, bkv, cs, sn, Qp, Kp, Vt};
  hipLaunchKernelGGL(k_qkv_heads, dim3(B * (L16 / 16), nroles), dim3(256), 0, s, x, Q, L, L16, role0);
  return 0;
}

// q14: {ss, Wq, bq, Wkv, bkv, cs, sn} of stack 0 then of stack 1; Qp / Kp / Vt: stack-major [2, B, H, ...] outputs
int launch_qkv_heads2(const float* x0, const float* x1, const float* const* q14, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H,
                      hipStream_t s) {
  if (D != kD || H != kH) return 1;
  const int L16 = (L + 15) / 16 * 16;
  const size_t half = (size_t)B * kH * L16 * 16;
  QkvArgs Q0{q14[0], q14[1], q14[2], q14[3], q14[4], q14[5], q14[6], Qp, Kp, Vt};
  QkvArgs Q1{q14[7], q14[8], q14[9], q14[10], q14[11], q14[12], q14[13], Qp + half, Kp + half, Vt + half};
  hipLaunchKernelGGL(k_qkv_heads2, dim3(B * (L16 / 16), 3, 2), dim3(256), 0, s, x0, x1, Q0, Q1, L, L16);
  return 0;
}

// a26: OutFfnArgs pointer fields (att .. be2) of stack 0 then of stack 1; out: stack-major [2, B, L, D]
int launch_out_ffn_mfma2(const float* const* a26, const float* eps4, float* out, int B, int L, int D, hipStream_t s) {
  if (D != kD) return 1;
  const long long tokens = (long long)B * L;
  const int tiles = (int)((tokens + 15) / 16);
  OutFfnArgs A0{a26[0], a26[1], a26[2], a26[3], a26[4], a26[5], a26[6], a26[7], a26[8], a26[9], a26[10], a26[11], a26[12], eps4[0], eps4[1], out};
  OutFfnArgs A1{a26[13], a26[14], a26[15], a26[16], a26[17], a26[18], a26[19], a26[20], a26[21], a26[22], a26[23], a26[24], a26[25], eps4[2],
                eps4[3], out + tokens * kD};
  hipLaunchKernelGGL(k_out_ffn_mfma2, dim3(2 * tiles), dim3(256), 0, s, A0, A1, L, tokens, tiles);
  return 0;
}

int launch_attention_heads(const float* Qp, const float* Kp, const float* Vt, const uint8_t* pad, float* out, int B, int Lq, int Lk,
                           int H, int dh, hipStream_t s) {
  if (H != kH || dh != kDH) return 1;
  const int Lq16 = (Lq + 15) / 16 * 16, Lk16 = (Lk + 15) / 16 * 16;
  const float scale = 1.0f / sqrtf((float)dh);
  if (Lk16 / 16 > 80 && Lq16 / 16 * H * B < 128)  // long key axis, few workgroups: spread the keys over 16 waves
    hipLaunchKernelGGL((k_attention_heads<16, 6, 1>), dim3(Lq16 / 16, H, B), dim3(1024), 0, s, Qp, Kp, Vt, pad, out, Lq, Lq16, Lk, Lk16, scale);
  else
    hipLaunchKernelGGL((k_attention_heads<4, 10, 1>), dim3(Lq16 / 16, H, B), dim3(256), 0, s, Qp, Kp, Vt, pad, out, Lq, Lq16, Lk, Lk16, scale);
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

int launch_out_ffn_mfma(const float* att, const float* res, const float* Wo, const float* bo, const float* g1, const float* be1,
                        float eps1, const float* ss, const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* g2, const float* be2, float eps2, float* out, int B, int L, int D, hipStream_t s) {
  if (D != kD) return 1;
  const long long tokens = (long long)B * L;
  OutFfnArgs A{att, res, Wo, bo, g1, be1, ss, W1, b1, W2, b2, g2, be2, eps1, eps2, out};
  hipLaunchKernelGGL(k_out_ffn_mfma, dim3((unsigned)((tokens + 15) / 16)), dim3(256), 0, s, A, L, tokens, AttPartials{nullptr, 0, 0});
  return 0;
}

// the same with the attention output given as the key-split partials of launch_attention_heads_split (L = Lq <= 16)
int launch_out_ffn_mfma_partials(const float* partials, int n_split, const float* res, const float* Wo, const float* bo, const float* g1,
                                 const float* be1, float eps1, const float* ss, const float* W1, const float* b1, const float* W2,
                                 const float* b2, const float* g2, const float* be2, float eps2, float* out, int B, int L, int D,
                                 hipStream_t s) {
  if (D != kD || L > 16 || n_split < 1) return 1;
  const long long tokens = (long long)B * L;
  OutFfnArgs A{partials, res, Wo, bo, g1, be1, ss, W1, b1, W2, b2, g2, be2, eps1, eps2, out};
  hipLaunchKernelGGL(k_out_ffn_mfma, dim3((unsigned)((tokens + 15) / 16)), dim3(256), 0, s, A, L, tokens, AttPartials{partials, n_split, L});
  return 0;
}

// args13: att, res, Wo, bo, g1, be1, ss, W1, b1, W2, b2, g2, be2 (OutFfnArgs order); next7: ss, Wq, bq, Wkv, bkv, cs, sn
int launch_out_ffn_qkv(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp, float* Kp,
                       float* Vt, int B, int L, int D, int H, int roles, const float* partials, int n_split, hipStream_t s) {
  if (D != kD || H != kH || (partials && L > 16)) return 1;
  const int L16 = (L + 15) / 16 * 16;
  OutFfnArgs A{args13[0], args13[1], args13[2], args13[3], args13[4], args13[5], args13[6], args13[7], args13[8], args13[9], args13[10],
               args13[11], args13[12], eps1, eps2, out};
  QkvArgs Q{next7[0], next7[1], next7[2], next7[3], next7[4], next7[5], next7[6], Qp, Kp, Vt};
  hipLaunchKernelGGL(k_out_ffn_qkv, dim3(B * (L16 / 16)), dim3(256), 0, s, A, Q, L, L16, roles, AttPartials{partials, n_split, L});
  return 0;
}

}  // namespace mmf
