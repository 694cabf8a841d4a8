// mmf_kernels_fps.hip -- farthest-point sampling in feature space for the policy encoder.  gfx950 / wave64.
//
// Replaces dgl.geometry.farthest_point_sampler(x[B,N,C], npoints, start_idx=0) (a dgl CUDA op without a ROCm
// build) called at mindmap/diffuser_actor/encoder.py:366-370 with B = 32, N = 3072 tokens, C = 120 channels,
// npoints = N/5 = 614.
//
// Algorithm (classic FPS): dist[i] = +inf; pick start; repeat npoints-1 times: dist[i] = min(dist[i], |x_i - x_sel|^2)
// with the squared distance accumulated channel by channel in float32 (acc += d*d, no fma), select argmax_i dist[i]
// (first index on ties).  The picks are sequential, so the only parallelism is inside one pick.
//
// Resident-point kernel (k_fps_resident, the path for N <= 4096, C <= 128):
//   * one batch element is spread over W = ceil(N/64) single-wave workgroups; lane l of wave w owns point 64*w + l and
//     keeps its CP (C padded to 16/32/64/96/128) channels IN REGISTERS for the whole kernel: after the first touch no
//     point data is read from memory again (the first version streamed the 1.5 MB point matrix from L2 on every pick,
//     uncoalesced: 65 us per pick);
//   * the selected row is wave-uniform: 512 B of scalar loads from a zero-padded copy of x made by k_fps_pad;
//   * the argmax is a wave butterfly + an exchange through global memory: every wave publishes one 64-bit key
//     {f32 bits of dist | 12 bits ~index | 20-bit pick number} into slot[batch][pick & 1][wave] with a relaxed
//     agent-scope store and polls the W slots of its batch element until all carry the current pick number.  Two
//     slot rows suffice: a wave can publish pick p+2 only after it has read every wave's pick p+1 key, which that wave
//     stored after reading all keys of pick p.  No other data is ordered by the exchange, so no fences are needed;
//   * the waves of one batch element take block ids of the same residue mod 8 (same XCD), adjacent in launch order.
//     Launches are chunked so that the whole grid is co-resident; the spin is bounded (writes -1 and leaves).
// Fallback kernel (k_fps_stream) for larger N or C: one 1024-thread workgroup per batch element, points streamed.
#include <math.h>
#include <stdlib.h>

#include "mmf_launch.h"

namespace mmf {

constexpr int kFpsMaxResidentN = 4096;  // 12 index bits in the key, <= 64 waves polled by one wave
constexpr int kFpsMaxResidentC = 128;
constexpr unsigned kFpsSpinLimit = 1u << 22;

template <int CP>
__global__ __launch_bounds__(256) void k_fps_pad(const float* __restrict__ x, long long rows, int C, float* __restrict__ xp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * CP) return;
  const long long r = i / CP;
  const int c = (int)(i - r * CP);
  xp[i] = c < C ? x[r * C + c] : 0.0f;
}

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const u64 o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

template <int CP>
__global__ __launch_bounds__(64) void k_fps_resident(const float* __restrict__ xp, int b0, int bend, int N, int W, int npoints,
                                                    int start, u64* slots, long long* __restrict__ out_idx) {
  const int id = blockIdx.x, lane = threadIdx.x;
  const int r = id >> 3;
  const int g = b0 + (id & 7) + 8 * (r / W);  // batch element: its W waves share id % 8 (one XCD)
  const int w = r % W;
  if (g >= bend) return;  // rounded-up tail of this launch: belongs to the next chunk (or to nobody)
  const int p = w * 64 + lane;
  const bool valid = p < N;
  const float* xb = xp + (size_t)g * N * CP;
  float v[CP];
  {
    const float4* src = reinterpret_cast<const float4*>(xb + (size_t)(valid ? p : 0) * CP);
#pragma unroll
    for (int c = 0; c < CP / 4; ++c) {
      const float4 q = src[c];
      v[4 * c] = q.x, v[4 * c + 1] = q.y, v[4 * c + 2] = q.z, v[4 * c + 3] = q.w;
    }
  }
  u64* myslots = slots + (size_t)g * 2 * 64;
  long long* ob = out_idx + (size_t)g * npoints;
  float dist = INFINITY;
  int cur = start;
  if (w == 0 && lane == 0) ob[0] = cur;
  const u64 pidx = (u64)(0xFFFu - (unsigned)p) << 20;
  for (int it = 1; it < npoints; ++it) {
    const float* row = xb + (size_t)__builtin_amdgcn_readfirstlane(cur) * CP;  // wave-uniform: scalar loads
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float d = v[c] - row[c];
      acc += d * d;
    }
    dist = fminf(dist, acc);
    const u64 tag = (u64)(it & 0xFFFFF);
    u64 key = valid ? (((u64)__float_as_uint(dist) << 32) | pidx | tag) : tag;
    key = wave_max_u64(key);
    u64* row_slots = myslots + (it & 1) * 64;
    if (lane == 0) __hip_atomic_store(row_slots + w, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 got = tag;
    if (lane < W) {
      unsigned spins = 0;
      for (;;) {
        got = __hip_atomic_load(row_slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((got & 0xFFFFF) == tag) break;
        if (++spins > kFpsSpinLimit) {
          got = ~0ull;  // the exchange never completed (a peer wave is not running): give up, flagged below
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    got = wave_max_u64(got);
    if (got == ~0ull) {
      if (lane == 0) ob[it] = -1;
      return;
    }
    cur = 0xFFF - (int)((got >> 20) & 0xFFF);
    if (w == 0 && lane == 0) ob[it] = cur;
  }
}

// ---- grouped form: a batch element = G workgroups of NWV waves ----------------------------------------------------------------
// The single-wave form above pays two dependent memory round trips per pick: the key exchange, then the fetch of the winner's
// row.  Here the waves of a workgroup settle their best point through LDS, and each workgroup publishes its best point's KEY
// AND ROW (CP + 1 self-validating 64-bit words: {tag | payload}, relaxed agent-scope stores, no fences); every workgroup
// then polls the G x (CP + 1) words of its batch element -- one word per thread -- and finds the winner's row already in
// hand: ONE round trip per pick.  Publishing rows is affordable because G is 4, not 48.
constexpr int kFpsMaxGroups = 8;

template <int CP, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_fps_groups(const float* __restrict__ xp, int b0, int bend, int N, int G, int npoints, int start,
                                                        u64* pub, long long* __restrict__ out_idx) {
  constexpr int NT = 64 * NWV;
  __shared__ u64 s_wkey[NWV];
  __shared__ __attribute__((aligned(16))) float s_pub[CP];
  __shared__ __attribute__((aligned(16))) float s_cand[kFpsMaxGroups][CP];
  __shared__ u64 s_ckey[kFpsMaxGroups];
  __shared__ int s_fail;
  const int id = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = id >> 3;
  const int b = b0 + (id & 7) + 8 * (r / G);  // batch element: its G workgroups share id % 8 (one XCD)
  const int g = r % G;
  if (b >= bend) return;  // rounded-up tail of this launch
  const int p = (g * NWV + wave) * 64 + lane;
  const bool valid = p < N;
  const float* xb = xp + (size_t)b * N * CP;
  float v[CP];
  {
    const float4* src = reinterpret_cast<const float4*>(xb + (size_t)(valid ? p : 0) * CP);
#pragma unroll
    for (int c = 0; c < CP / 4; ++c) {
      const float4 q = src[c];
      v[4 * c] = q.x, v[4 * c + 1] = q.y, v[4 * c + 2] = q.z, v[4 * c + 3] = q.w;
    }
  }
  // the first selected row
  for (int c = tid; c < CP; c += NT) s_cand[0][c] = xb[(size_t)start * CP + c];
  if (tid == 0) s_fail = 0;
  __syncthreads();
  u64* mypub = pub + (size_t)b * 2 * G * (CP + 1);
  long long* ob = out_idx + (size_t)b * npoints;
  float dist = INFINITY;
  int win = 0;  // which candidate row holds the current selection
  if (g == 0 && tid == 0) ob[0] = start;
  const u64 pidx = (u64)(0xFFFu - (unsigned)p) << 20;
  for (int it = 1; it < npoints; ++it) {
    const float* row = s_cand[win];
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CP; c += 4) {
      const float4 q = *reinterpret_cast<const float4*>(row + c);
      float d = v[c] - q.x;
      acc += d * d;
      d = v[c + 1] - q.y;
      acc += d * d;
      d = v[c + 2] - q.z;
      acc += d * d;
      d = v[c + 3] - q.w;
      acc += d * d;
    }
    dist = fminf(dist, acc);
    const u64 tag = (u64)(it & 0xFFFFF);
    u64 key = valid ? (((u64)__float_as_uint(dist) << 32) | pidx | tag) : tag;
    key = wave_max_u64(key);
    if (lane == 0) s_wkey[wave] = key;
    __syncthreads();
    u64 best = s_wkey[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) best = s_wkey[w] > best ? s_wkey[w] : best;
    const int best_p = 0xFFF - (int)((best >> 20) & 0xFFF);
    if (valid && p == best_p) {  // the owner of the workgroup's best point hands its row over
#pragma unroll
      for (int c = 0; c < CP; c += 4) *reinterpret_cast<float4*>(s_pub + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    }
    __syncthreads();
    u64* slot_row = mypub + (size_t)(it & 1) * G * (CP + 1);
    if (tid <= CP) {
      const u64 word = tid < CP ? (((u64)(unsigned)it << 32) | (u64)__float_as_uint(s_pub[tid])) : best;
      __hip_atomic_store(slot_row + (size_t)g * (CP + 1) + tid, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int idx = tid; idx < G * (CP + 1); idx += NT) {
      const int gg = idx / (CP + 1), c = idx - gg * (CP + 1);
      u64 got = 0;
      unsigned spins = 0;
      bool ok = false;
      for (;;) {
        got = __hip_atomic_load(slot_row + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = c < CP ? (unsigned)(got >> 32) == (unsigned)it : (got & 0xFFFFF) == tag;
        if (ok || ++spins > kFpsSpinLimit) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) s_fail = 1;  // a peer workgroup is not running: give up (flagged below)
      if (c < CP)
        s_cand[gg][c] = __uint_as_float((unsigned)got);
      else
        s_ckey[gg] = got;
    }
    __syncthreads();
    if (s_fail) {
      if (g == 0 && tid == 0) ob[it] = -1;
      return;
    }
    u64 top = s_ckey[0];
    win = 0;
    for (int gg = 1; gg < G; ++gg)
      if (s_ckey[gg] > top) {
        top = s_ckey[gg];
        win = gg;
      }
    if (g == 0 && tid == 0) ob[it] = 0xFFF - (int)((top >> 20) & 0xFFF);
  }
}

// Scratch of a launcher: either carved out of the caller's workspace (mmf_farthest_point_sampling_ws: no runtime allocation, so the
// call can sit inside a captured HIP graph without memory nodes -- those make hipGraphLaunch execute the graph synchronously from the
// host) or stream-ordered allocations of its own.
struct FpsScratch {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  bool own = false;
  void* a = nullptr;
  void* b = nullptr;
  hipStream_t s = nullptr;
  bool get(size_t bytes_a, size_t bytes_b) {
    if (base) {
      const size_t oa = (used + 255) & ~(size_t)255, ob = (oa + bytes_a + 255) & ~(size_t)255;
      if (ob + bytes_b > cap) return false;
      a = base + oa;
      b = base + ob;
      return true;
    }
    own = true;
    if (hipMallocAsync(&a, bytes_a, s) != hipSuccess) return false;
    if (hipMallocAsync(&b, bytes_b, s) != hipSuccess) {
      (void)hipFreeAsync(a, s);
      a = nullptr;
      return false;
    }
    return true;
  }
  void release() {
    if (own) {
      (void)hipFreeAsync(b, s);
      (void)hipFreeAsync(a, s);
    }
  }
};

size_t fps_workspace_bytes(int B, int N, int C) {
  // padded rows (the resident / grouped kernels, C <= 128) + the largest publication area of any launcher + alignment slack
  return sizeof(float) * (size_t)B * N * 128 + sizeof(unsigned long long) * (size_t)B * 2 * 64 * 129 + 1024;
}

template <int CP, int NWV>
static int fps_groups(const float* x, int B, int N, int C, int npoints, int start, long long* out_idx, hipStream_t s, FpsScratch sc) {
  const int G = (N + 64 * NWV - 1) / (64 * NWV);
  if (G > kFpsMaxGroups || CP + 1 > 64 * NWV) return 1;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fps_groups<CP, NWV>, 64 * NWV, 0) != hipSuccess || per_cu <= 0)
    return 2;
  int chunk = (int)((long long)cus * per_cu / G) / 8 * 8;  // batch elements per launch: the whole grid must be co-resident
  if (chunk < 8) return 1;
  float* xp = nullptr;
  u64* pub = nullptr;
  const size_t xp_bytes = sizeof(float) * (size_t)B * N * CP, pub_bytes = sizeof(u64) * (size_t)B * 2 * G * (CP + 1);
  if (!sc.get(xp_bytes, pub_bytes)) return 2;
  xp = (float*)sc.a;
  pub = (u64*)sc.b;
  (void)hipMemsetAsync(pub, 0, pub_bytes, s);
  const long long rows = (long long)B * N;
  hipLaunchKernelGGL(k_fps_pad<CP>, dim3((unsigned)((rows * CP + 255) / 256)), dim3(256), 0, s, x, rows, C, xp);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (B - b0 < chunk ? B - b0 : chunk);
    hipLaunchKernelGGL((k_fps_groups<CP, NWV>), dim3(8 * G * ((nb + 7) / 8)), dim3(64 * NWV), 0, s, xp, b0, b0 + nb, N, G, npoints, start, pub,
                       out_idx);
  }
  sc.release();
  return 0;
}

// ---- several picks per exchange ------------------------------------------------------------------------------------------------
// A pick costs one cross-workgroup exchange (an L2 round trip of ~1 us plus the barriers around it) on top of the distance
// update, and with more workgroups per batch element the update shrinks while the exchange does not.  So an exchange carries
// more than one pick, EXACTLY:
//   * every workgroup publishes its K best points by key (key = distance bits | ~index: a total order, first index on ties), keys
//     and rows;
//   * T = the largest K-th key of any workgroup bounds the key of every point that was NOT published (a point's key only ever
//     decreases); the candidates with key >= T are fetched by everybody (rows: they are few);
//   * pick 1 is the largest key (it is among each workgroup's best).  Then every workgroup applies the pick to its own points and,
//     on a fourth wave, to the fetched candidates -- the same channel-by-channel sum their owners compute -- and the best
//     candidate is pick 2 if its NEW key is still >= T: no unpublished point can beat it.  And so on, until the best candidate
//     falls below T (or K picks are made): then the next exchange.
// On the policy's shape (3 072 x 120, 16 workgroups of 3 + 1 waves, K = 4) 3.8 - 4.0 picks ride on one exchange, 21 candidates
// are fetched on average; 614 picks: 2.5 ms (one pick per exchange, 4 workgroups of 12 waves) -> 2.0 ms.  An exchange still costs
// ~9 us beside its picks (publish -> keys -> rows are three dependent trips through the L2, at the clock a 16-CU job gets), so
// the shipped arrangement offers more picks per exchange: 8 workgroups of 6 + 1 waves, K = 8 -> 1.6 ms.
constexpr int kFpsMultiGroups = 16, kFpsElig = 64;  // (one candidate per lane of the auxiliary wave)

// max of a 64-bit key over the wave, the result in every lane (as a scalar): four cross-lane steps inside the 16-lane rows
// (pairs, quads, mirrored half, mirrored row), two row broadcasts, one read of lane 63 -- data-parallel-primitive moves instead
// of the six ds_bpermute round trips of the butterfly (0.6 us each time on one wave; an exchange takes ten such maxima).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_max_step(u64 v) {
  const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  const u64 o = ((u64)ohi << 32) | olo;
  return o > v ? o : v;
}
__device__ __forceinline__ u64 wave_max_u64_dpp(u64 v) {
  v = dpp_max_step<0xB1, 0xf>(v);   // quad_perm [1, 0, 3, 2]
  v = dpp_max_step<0x4E, 0xf>(v);   // quad_perm [2, 3, 0, 1]
  v = dpp_max_step<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_max_step<0x140, 0xf>(v);  // row_mirror: every lane holds its row's maximum
  v = dpp_max_step<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave's maximum
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((u64)hi << 32) | lo;
}

template <int CP, int NWV, int K>
__global__ __launch_bounds__(64 * (NWV + 1)) void k_fps_multi(const float* __restrict__ xp, int b0, int bend, int N, int G, int npoints,
                                                             int start, u64* pub, long long* __restrict__ out_idx) {
  constexpr int NT = 64 * (NWV + 1), RS = CP + 4;
  __shared__ u64 s_wtop[NWV][K];
  __shared__ __attribute__((aligned(16))) float s_pub[K][CP];
  __shared__ u64 s_pubkey[K];
  __shared__ u64 s_keys[64];
  __shared__ __attribute__((aligned(16))) float s_rows[kFpsElig][RS];
  __shared__ int s_elig[kFpsElig];
  __shared__ int s_nelig, s_fail;
  __shared__ int s_pick[2];
  const int id = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = id >> 3;
  const int b = b0 + (id & 7) + 8 * (r / G);  // batch element: its G workgroups share id % 8 (one XCD)
  const int g = r % G;
  if (b >= bend) return;  // rounded-up tail of this launch
  const bool main_wave = wave < NWV;
  const int p = (g * NWV + wave) * 64 + lane;  // main waves: the lane's point
  const bool valid = main_wave && p < N;
  const float* xb = xp + (size_t)b * N * CP;
  float v[CP];
  if (main_wave) {
    const float4* src = reinterpret_cast<const float4*>(xb + (size_t)(valid ? p : 0) * CP);
#pragma unroll
    for (int c = 0; c < CP / 4; ++c) {
      const float4 q = src[c];
      v[4 * c] = q.x, v[4 * c + 1] = q.y, v[4 * c + 2] = q.z, v[4 * c + 3] = q.w;
    }
  }
  for (int c = tid; c < CP; c += NT) s_rows[0][c] = xb[(size_t)start * CP + c];  // the first selected row
  if (tid == 0) s_fail = 0, s_pick[0] = 0, s_nelig = 1;
  __syncthreads();
  u64* mypub = pub + (size_t)b * 2 * G * K * (CP + 1);
  long long* ob = out_idx + (size_t)b * npoints;
  float dist = INFINITY;
  if (g == 0 && tid == 0) ob[0] = start;
  const u64 pidx = (u64)(0xFFFu - (unsigned)p) << 20;
  int it = 1;                // picks made so far
  u64 ck = 0, T_eff = 0;     // aux wave: the lane's candidate key (slot = lane), the bound of everything not fetched
  int n = 0, pick = 0;       // picks of the current exchange; slot of the pick being applied
  bool first = true;         // the start point is applied like a pick
  for (unsigned round = 1;; ++round) {
    // ---- apply picks: slot `pick` of s_rows, then ask the aux wave for the next one
    for (;;) {
      if (main_wave) {
        const float* row = s_rows[pick];
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < CP; c += 4) {
          const float4 q = *reinterpret_cast<const float4*>(row + c);
          float d = v[c] - q.x;
          acc += d * d;
          d = v[c + 1] - q.y;
          acc += d * d;
          d = v[c + 2] - q.z;
          acc += d * d;
          d = v[c + 3] - q.w;
          acc += d * d;
        }
        dist = fminf(dist, acc);
      } else {
        // the candidates (one per lane) against the pick: what their owners compute for them
        int nxt = -1;
        if (!first) {
          const int ne = s_nelig;
          const float* mine = s_rows[lane < ne ? lane : 0];
          const float* row = s_rows[pick];
          float acc = 0.0f;
#pragma unroll 4
          for (int c = 0; c < CP; c += 4) {
            const float4 a = *reinterpret_cast<const float4*>(mine + c), q = *reinterpret_cast<const float4*>(row + c);
            float d = a.x - q.x;
            acc += d * d;
            d = a.y - q.y;
            acc += d * d;
            d = a.z - q.z;
            acc += d * d;
            d = a.w - q.w;
            acc += d * d;
          }
          if (lane < ne) {
            const float dn = fminf(__uint_as_float((unsigned)(ck >> 32)), acc);
            ck = ((u64)__float_as_uint(dn) << 32) | (ck & 0xFFFFFFFFull);
          }
          // the next pick of this exchange: the best candidate, if nothing outside the fetched set can beat it
          const u64 best = wave_max_u64_dpp(lane < ne ? ck : 0ull);
          if (n < K && it + n < npoints && best >= T_eff) {
            const unsigned long long m = __ballot(lane < ne && ck == best);
            nxt = __ffsll((long long)m) - 1;
            if (g == 0 && lane == 0) ob[it + n] = 0xFFF - (int)((best >> 20) & 0xFFF);
          }
        }
        if (lane == 0) s_pick[(n + 1) & 1] = nxt;
      }
      __syncthreads();
      if (first) break;
      const int nxt = s_pick[(n + 1) & 1];
      if (nxt < 0) break;
      pick = nxt;
      ++n;
    }
    if (!first) it += n;
    first = false;
    if (it >= npoints) return;
    // ---- this workgroup's K best points: keys and rows.  Every main wave takes ITS K best (K wave maxima, no barrier in
    // between); a point is among the workgroup's K best iff fewer than K of the NWV K wave candidates exceed it -- its rank is
    // its slot.  (K rounds of {wave maximum, barrier, owner writes its row} were a fifth of an exchange.)
    const u64 tag = (u64)(round & 0xFFFFF);
    const u64 mykey = valid ? (((u64)__float_as_uint(dist) << 32) | pidx | tag) : tag;
    bool sel = false;
    if (main_wave) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const u64 m = wave_max_u64_dpp(sel ? tag : mykey);
        if (lane == 0) s_wtop[wave][k] = m;
        if (valid && !sel && mykey == m) sel = true;  // (keys of valid points are distinct)
      }
    }
    if (tid < K) s_pubkey[tid] = tag;  // (a slot without a point: the empty key)
    __syncthreads();
    if (sel) {
      int rank = 0;
#pragma unroll
      for (int e = 0; e < NWV * K; ++e) rank += (&s_wtop[0][0])[e] > mykey ? 1 : 0;
      if (rank < K) {  // the owner hands its row over
        s_pubkey[rank] = mykey;
#pragma unroll
        for (int c = 0; c < CP; c += 4) *reinterpret_cast<float4*>(&s_pub[rank][c]) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
      }
    }
    __syncthreads();
    u64* slot_row = mypub + (size_t)(round & 1) * G * K * (CP + 1);
    for (int e = tid; e < K * (CP + 1); e += NT) {
      const int k = e / (CP + 1), c = e - k * (CP + 1);
      const u64 word = c < CP ? (((u64)round << 32) | (u64)__float_as_uint(s_pub[k][c])) : s_pubkey[k];
      __hip_atomic_store(slot_row + (size_t)(g * K + k) * (CP + 1) + c, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- everybody's keys; the bound T; the candidates to fetch (aux wave)
    if (!main_wave) {
      u64 key = 0;
      if (lane < G * K) {
        unsigned spins = 0;
        for (;;) {
          key = __hip_atomic_load(slot_row + (size_t)lane * (CP + 1) + CP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((key & 0xFFFFF) == tag) break;
          if (++spins > kFpsSpinLimit) {
            s_fail = 1;  // a peer workgroup is not running: give up (flagged below)
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      const u64 T = wave_max_u64_dpp((lane < G * K && lane % K == K - 1) ? key : 0ull);
      bool elig = lane < G * K && key >= T;
      unsigned long long m = __ballot(elig);
      u64 bound = T;
      if (__popcll(m) > kFpsElig) {  // (not seen in practice) too many to fetch: one pick in this exchange
        const u64 top = wave_max_u64_dpp(lane < G * K ? key : 0ull);
        elig = lane < G * K && key == top;
        m = __ballot(elig);
        bound = ~0ull;
      }
      const int slot = __popcll(m & ((1ull << lane) - 1ull));
      if (elig) s_elig[slot] = lane;
      if (lane == 0) s_nelig = __popcll(m);
      s_keys[lane] = key;
      T_eff = bound;
    }
    __syncthreads();
    if (s_fail) {
      if (g == 0 && tid == 0) ob[it] = -1;
      return;
    }
    // ---- their rows
    // (requested sixteen words per thread at a time, checked afterwards: one dependent round trip per word made this the longest
    // part of an exchange; the rows were stored before the keys that have been seen, a stale word is the exception)
    const int ne = s_nelig;
    for (int e0 = tid; e0 < ne * CP; e0 += 16 * NT) {
      u64 got[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = min(e0 + i * NT, ne * CP - 1), sl = e / CP, c = e - sl * CP;
        got[i] = __hip_atomic_load(slot_row + (size_t)s_elig[sl] * (CP + 1) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = e0 + i * NT;
        if (e < ne * CP) {
          const int sl = e / CP, c = e - sl * CP;
          unsigned spins = 0;
          while ((unsigned)(got[i] >> 32) != round) {
            if (++spins > kFpsSpinLimit) {
              s_fail = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            got[i] = __hip_atomic_load(slot_row + (size_t)s_elig[sl] * (CP + 1) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          s_rows[sl][c] = __uint_as_float((unsigned)got[i]);
        }
      }
    }
    // ---- the first pick of the exchange: the largest key (aux wave)
    n = 0;
    if (!main_wave) {
      ck = lane < ne ? s_keys[s_elig[lane]] : 0ull;
      const u64 best = wave_max_u64_dpp(ck);
      const unsigned long long m = __ballot(lane < ne && ck == best);
      const int first_slot = __ffsll((long long)m) - 1;
      if (lane == 0) s_pick[1] = first_slot;  // (the apply loop's first iteration, n = 1, writes s_pick[0])
      if (g == 0 && lane == 0) ob[it] = 0xFFF - (int)((best >> 20) & 0xFFF);
    }
    __syncthreads();
    if (s_fail) {
      if (g == 0 && tid == 0) ob[it] = -1;
      return;
    }
    pick = s_pick[1];
    n = 1;
  }
}

template <int CP, int NWV, int K>
static int fps_multi(const float* x, int B, int N, int C, int npoints, int start, long long* out_idx, hipStream_t s, FpsScratch sc) {
  const int G = (N + 64 * NWV - 1) / (64 * NWV);
  if (G * K > 64 || G > kFpsMultiGroups || N > 4096 || N - (G - 1) * 64 * NWV < K) return 1;  // (every workgroup publishes K real points)
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fps_multi<CP, NWV, K>, 64 * (NWV + 1), 0) != hipSuccess || per_cu <= 0)
    return 2;
  int chunk = (int)((long long)cus * per_cu / G) / 8 * 8;  // batch elements per launch: the whole grid must be co-resident
  if (chunk < 8) return 1;
  float* xp = nullptr;
  u64* pub = nullptr;
  const size_t xp_bytes = sizeof(float) * (size_t)B * N * CP, pub_bytes = sizeof(u64) * (size_t)B * 2 * G * K * (CP + 1);
  if (!sc.get(xp_bytes, pub_bytes)) return 2;
  xp = (float*)sc.a;
  pub = (u64*)sc.b;
  (void)hipMemsetAsync(pub, 0, pub_bytes, s);
  const long long rows = (long long)B * N;
  hipLaunchKernelGGL(k_fps_pad<CP>, dim3((unsigned)((rows * CP + 255) / 256)), dim3(256), 0, s, x, rows, C, xp);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (B - b0 < chunk ? B - b0 : chunk);
    hipLaunchKernelGGL((k_fps_multi<CP, NWV, K>), dim3(8 * G * ((nb + 7) / 8)), dim3(64 * (NWV + 1)), 0, s, xp, b0, b0 + nb, N, G, npoints, start,
                       pub, out_idx);
  }
  sc.release();
  return 0;
}

constexpr int kFpsThreads = 1024;
constexpr int kFpsMaxPerThread = 8;  // N <= 8192
constexpr int kFpsMaxC = 1024;

__global__ __launch_bounds__(kFpsThreads) void k_fps_stream(const float* __restrict__ x, int N, int C, int npoints, int start,
                                                           long long* __restrict__ out_idx) {
  __shared__ float s_sel[kFpsMaxC];
  __shared__ float s_best_d[kFpsThreads / 64];
  __shared__ int s_best_i[kFpsThreads / 64];
  __shared__ int s_cur;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * N * C;
  long long* ob = out_idx + (size_t)b * npoints;
  float dist[kFpsMaxPerThread];
#pragma unroll
  for (int k = 0; k < kFpsMaxPerThread; ++k) dist[k] = INFINITY;
  int cur = start;
  if (tid == 0) ob[0] = cur;
  for (int it = 1; it < npoints; ++it) {
    for (int c = tid; c < C; c += kFpsThreads) s_sel[c] = xb[(size_t)cur * C + c];
    __syncthreads();
    float best_d = -1.0f;
    int best_i = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < kFpsMaxPerThread; ++k) {
      const int i = tid + k * kFpsThreads;
      if (i < N) {
        const float* xi = xb + (size_t)i * C;
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
          const float d = xi[c] - s_sel[c];
          acc += d * d;
        }
        const float dm = fminf(dist[k], acc);
        dist[k] = dm;
        if (dm > best_d) {  // ascending i within a thread: keeps the first index on ties
          best_d = dm;
          best_i = i;
        }
      }
    }
    // argmax over the workgroup, ties -> smallest index
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_down(best_d, off, 64);
      const int oi = __shfl_down(best_i, off, 64);
      if (od > best_d || (od == best_d && oi < best_i)) {
        best_d = od;
        best_i = oi;
      }
    }
    if (lane == 0) {
      s_best_d[wave] = best_d;
      s_best_i[wave] = best_i;
    }
    __syncthreads();
    if (tid == 0) {
      float bd = s_best_d[0];
      int bi = s_best_i[0];
      for (int w = 1; w < kFpsThreads / 64; ++w)
        if (s_best_d[w] > bd || (s_best_d[w] == bd && s_best_i[w] < bi)) {
          bd = s_best_d[w];
          bi = s_best_i[w];
        }
      s_cur = bi;
      ob[it] = bi;
    }
    __syncthreads();
    cur = s_cur;
  }
}

template <int CP>
static int fps_resident(const float* x, int B, int N, int C, int npoints, int start, long long* out_idx, hipStream_t s, FpsScratch sc) {
  const int W = (N + 63) / 64;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fps_resident<CP>, 64, 0) != hipSuccess || per_cu <= 0)
    return 2;
  // batch elements per launch: the whole grid must be co-resident (waves of one element wait for each other)
  int chunk = (int)((long long)cus * per_cu / W) / 8 * 8;
  if (chunk < 8) return 1;
  float* xp = nullptr;
  u64* slots = nullptr;
  const size_t xp_bytes = sizeof(float) * (size_t)B * N * CP, slot_bytes = sizeof(u64) * (size_t)B * 2 * 64;
  if (!sc.get(xp_bytes, slot_bytes)) return 2;
  xp = (float*)sc.a;
  slots = (u64*)sc.b;
  (void)hipMemsetAsync(slots, 0, slot_bytes, s);
  const long long rows = (long long)B * N;
  hipLaunchKernelGGL(k_fps_pad<CP>, dim3((unsigned)((rows * CP + 255) / 256)), dim3(256), 0, s, x, rows, C, xp);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (B - b0 < chunk ? B - b0 : chunk);
    hipLaunchKernelGGL(k_fps_resident<CP>, dim3(8 * W * ((nb + 7) / 8)), dim3(64), 0, s, xp, b0, b0 + nb, N, W, npoints, start, slots,
                       out_idx);
  }
  sc.release();
  return 0;
}

// 0 = launched, 1 = unsupported shape, 2 = HIP runtime error
int launch_fps(const float* x, int B, int N, int C, int npoints, int start, long long* out_idx, hipStream_t s, void* workspace,
               size_t workspace_bytes) {
  FpsScratch sc;
  sc.s = s;
  sc.base = (char*)workspace;
  sc.cap = workspace_bytes;
  if (N > 1024 && N <= kFpsMaxResidentN && C > 96 && C <= kFpsMaxResidentC) {  // the policy's shape (3072 x 120)
    static const int single = getenv("MMF_DEBUG_FPS_SINGLE") ? 1 : 0;  // (diagnostics: one pick per exchange)
    // 8 workgroups of 6 + 1 waves, 8 picks offered per exchange (round 4: 1.81 -> 1.61 ms at B = 1, 2.19 -> 1.72 ms at B = 32 against
    // 16 workgroups of 3 + 1 waves with 4 picks: an exchange costs the same, more picks ride on it); MMF_DEBUG_FPS_VARIANT=1 is the
    // former arrangement (diagnostics)
    static const int variant = getenv("MMF_DEBUG_FPS_VARIANT") ? atoi(getenv("MMF_DEBUG_FPS_VARIANT")) : 0;
    int rc = single         ? 1
             : variant == 1 ? fps_multi<128, 3, 4>(x, B, N, C, npoints, start, out_idx, s, sc)
                            : fps_multi<128, 6, 8>(x, B, N, C, npoints, start, out_idx, s, sc);
    if (rc == 1) rc = fps_groups<128, 12>(x, B, N, C, npoints, start, out_idx, s, sc);  // (8 groups of 6 waves: 7 % slower)
    if (rc != 1) return rc;
  }
  if (N <= kFpsMaxResidentN && C <= kFpsMaxResidentC) {
    const int rc = C <= 16   ? fps_resident<16>(x, B, N, C, npoints, start, out_idx, s, sc)
                   : C <= 32 ? fps_resident<32>(x, B, N, C, npoints, start, out_idx, s, sc)
                   : C <= 64 ? fps_resident<64>(x, B, N, C, npoints, start, out_idx, s, sc)
                   : C <= 96 ? fps_resident<96>(x, B, N, C, npoints, start, out_idx, s, sc)
                             : fps_resident<128>(x, B, N, C, npoints, start, out_idx, s, sc);
    if (rc != 1) return rc;  // 1: the waves of 8 batch elements cannot be co-resident on this device -> stream
  }
  if (N > kFpsThreads * kFpsMaxPerThread || C > kFpsMaxC) return 1;
  hipLaunchKernelGGL(k_fps_stream, dim3(B), dim3(kFpsThreads), 0, s, x, N, C, npoints, start, out_idx);
  return 0;
}

}  // namespace mmf
