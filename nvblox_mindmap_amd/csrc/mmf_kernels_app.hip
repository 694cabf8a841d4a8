// mmf_kernels_app.hip -- appearance (colour / N-channel feature) fusion: candidate selection,
// sphere-traced synthetic depth for the occlusion test, and the per-voxel weighted update.
// gfx950 / wave64.
//
// Replaces the CUDA behind nvblox_torch Mapper.add_color_frame / add_feature_frame, reached by the
// reference at mindmap/mapping/helpers/nvblox_mapping_helpers.py:212-218 and :255-261.
#include <hip/hip_ext.h>

#include "mmf_launch.h"
#include "mmf_trace_device.h"
#include "mmf_alloc_device.h"
#include "mmf_app_device.h"

namespace mmf {

// ------------------------------------------------------------------------------------------------
// Candidate blocks: live TSDF blocks with a voxel inside the truncation band (W > 0, |D| < trunc)
// whose centre projects into the appearance image.  One workgroup per live block, 2 voxels/thread.
// ------------------------------------------------------------------------------------------------
template <int FMA>
__global__ __launch_bounds__(256) void k_app_candidates(LayerDev T, MapConsts mc, Cam cam, Rigid T_C_L,
                                                       uint8_t* __restrict__ flags, u64* __restrict__ cell_key) {
  const int n = T.ctr[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = T.live[i];
    int bx, by, bz;
    const u64 key = T.slot_key[slot];
    unpack_key(key, bx, by, bz);
    const float4 a = reinterpret_cast<const float4*>(T.pool)[(size_t)slot * (kVPB / 2) + threadIdx.x];
    int hit = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float D = r ? a.z : a.x, W = r ? a.w : a.y;
      if (!(W > 0.0f) || !(fabsf(D) < mc.trunc)) continue;
      float c[3], p[3], u, v;
      voxel_centre<FMA>(mc, bx, by, bz, threadIdx.x * 2 + r, c);
      xform<FMA>(T_C_L, c, p);
      if (!project<FMA>(cam, p, u, v)) continue;
      if (mc.max_dist > 0.0f && p[2] > mc.max_dist) continue;
      hit = 1;
    }
    const int any = __syncthreads_or(hit);
    if (threadIdx.x == 0) {
      flags[i] = any ? 1 : 0;
      if (any) cell_key[i] = key;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Sphere tracing, cooperative form: 16 lanes per (subsampled) ray, workgroup = 4x4 ray patch.
// The march of the spec is a chain of dependent samples, most of them exactly `trunc` apart (unobserved space
// and observed free space whose distance is clamped to +trunc).  A thread-per-ray kernel has ~19k threads and
// one wave per SIMD: pure latency (47 us).  Here lane k of a ray group evaluates the sample the march WOULD
// take after k trunc-steps (t_k built by the same k sequential float adds), a ballot finds the first lane whose
// sample ends the run of trunc-steps (termination, or a step != trunc), and the group resumes from that lane's
// state.  Every sample that is consumed is the spec's sample at bitwise the same t, so the image is identical;
// the kernel now has ~300k threads and is throughput-bound.
// Exact shortcuts kept from the scalar form: no lookup outside the workspace bounds (no block can exist there),
// arithmetic-only fast-forward up to the entry into the (one-block padded) bounds, stop after leaving them,
// no voxel read inside all-free blocks (LayerDev::block_free).
// ------------------------------------------------------------------------------------------------
constexpr int kRayLanes = 16;

// One 4x4 ray patch = 256 threads (4 whole waves; no LDS, no barriers: the body may share a workgroup with other roles).
// LAZY: the map is lazily decayed (LayerDev::epoch): a sampled voxel's weight may be cur_epoch - epoch[slot] decays behind -- the
// missing multiplications are applied to the sampled value (block_free is kept current by the decay's compaction).
// BDIV: mmf_params.block_index_by_division (the stand-alone launch k_sphere_trace only): the sampled point's block / voxel by division.
template <bool LAZY = false, bool BDIV = false>
__device__ inline void sphere_patch(const LayerDev& T, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C,
                                    float* __restrict__ synth, int Ws, int Hs, int patches_x, int patch, int tid,
                                    int* trace_iters = nullptr) {
  const int k = tid & (kRayLanes - 1);          // sample lane within the ray group
  const int rl = tid >> 4;                      // ray within the 4x4 patch
  const int gshift = (tid & 63) & ~(kRayLanes - 1);  // first lane of this group inside its wave
  const int cs = (patch % patches_x) * 4 + (rl & 3), rs = (patch / patches_x) * 4 + (rl >> 2);
  if (cs >= Ws || rs >= Hs) return;  // whole groups leave together
  const int idx = rs * Ws + cs;
  const float sf = (float)mc.st_sf;
  const float u = ((float)cs + 0.5f) * sf, v = ((float)rs + 0.5f) * sf;
  const float x = (u - cam.cx) / cam.fx, y = (v - cam.cy) / cam.fy;
  const float nrm = sqrtf((x * x + y * y) + 1.0f);
  const float dC[3] = {x / nrm, y / nrm, 1.0f / nrm};
  float dL[3];
  rotate(T_L_C, dC, dL);
  const float o[3] = {T_L_C.t[0], T_L_C.t[1], T_L_C.t[2]};
  // Loop constants of the march live in VECTOR registers: the kernel has VGPRs to spare (55 of 72 at its occupancy) and is out of
  // scalar registers -- the compiler was parking loop-invariant scalars in VGPR lanes and fetching them back with
  // v_readlane + s_nop inside the march (18 + 10 of the ~640 instructions of an iteration; the tracer is issue-bound).
  float c_inv_bs = mc.inv_bs, c_bs = mc.bs, c_inv_v = mc.inv_v, c_trunc = mc.trunc, c_eps = mc.st_eps, c_max_len = mc.st_max_len,
        c_half_v = 0.5f * mc.v;
  asm volatile("" : "+v"(c_inv_bs), "+v"(c_bs), "+v"(c_inv_v), "+v"(c_trunc), "+v"(c_eps), "+v"(c_max_len), "+v"(c_half_v));
  int w_lo0 = mc.ws_lo[0], w_lo1 = mc.ws_lo[1], w_lo2 = mc.ws_lo[2], w_hi0 = mc.ws_hi[0], w_hi1 = mc.ws_hi[1], w_hi2 = mc.ws_hi[2];
  int d_lo0 = T.d_lo[0], d_lo1 = T.d_lo[1], d_lo2 = T.d_lo[2], d_ny = T.d_ny, d_nz = T.d_nz;
  asm volatile("" : "+v"(w_lo0), "+v"(w_lo1), "+v"(w_lo2), "+v"(w_hi0), "+v"(w_hi1), "+v"(w_hi2));
  asm volatile("" : "+v"(d_lo0), "+v"(d_lo1), "+v"(d_lo2), "+v"(d_ny), "+v"(d_nz));
  auto inws = [&](int x, int y, int z) {  // in_workspace(mc, ...) on the pinned bounds
    if (mc.ws_type == 0) return true;
    if (z < w_lo2 || z > w_hi2) return false;
    if (mc.ws_type == 1) return true;
    return !(x < w_lo0 || x > w_hi0 || y < w_lo1 || y > w_hi1);
  };
  auto cell_of = [&](int x, int y, int z) { return __mul24(__mul24(x - d_lo0, d_ny) + (y - d_lo1), d_nz) + (z - d_lo2); };  // dense_cell(T, ...)

  // [t_enter, t_exit]: ray parameters at which the ray can be inside the workspace bounds padded by one block
  float t_enter = -3.0e38f, t_exit = 3.0e38f;
  if (mc.ws_type != 0) {
    float tmin = -3.0e38f, tmax = 3.0e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (mc.ws_type == 1 && a < 2) continue;
      const float lo = (float)(mc.ws_lo[a] - 1) * c_bs, hi = (float)(mc.ws_hi[a] + 2) * c_bs;
      if (fabsf(dL[a]) > 1e-9f) {
        const float t1 = (lo - o[a]) / dL[a], t2 = (hi - o[a]) / dL[a];
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
      } else if (o[a] < lo || o[a] > hi) {
        tmax = -3.0e38f;
      }
    }
    if (tmax >= tmin) {
      t_enter = tmin;
      t_exit = tmax;
    } else {
      t_exit = -1.0f;  // never inside: the first sample already fails
    }
  }

  bool last_pos = false, ok = false, done = false;
  float t = 0.0f;
  int i = 0;
  // fast-forward: before the padded bounds every sample is unobserved and (nothing observed yet) the step is trunc
  if (t_exit >= 0.0f)
    while (i < mc.st_max_steps && t < c_max_len && t < t_enter) {
      t += c_trunc;
      ++i;
    }

  // Two modes per ray (uniform over its 16 lanes).
  // WIDE: lane k speculates the sample after k trunc-steps (free / unobserved space: 16 steps per memory round trip).
  // NARROW: inside the truncation band every step is a distance read from the map and the next position depends on it,
  // but the VOXELS the march can reach next are the ones the ray crosses just ahead: lane k fetches the voxel at t + k/2
  // voxel sizes (lane 0: at t, the march's own sample), then the group marches through the fetched voxels without further
  // loads -- each new position is looked up among the 16 probes by (block, voxel) identity; a position whose voxel was
  // not fetched ends the round and the next one probes from there.  A round trip costs ~2 us under load whatever it
  // carries, so the work per ray is its number of rounds: 3-4 wide + up to 14 one-step rounds became 3-4 + 1-2.
  // Entered when a step came from the map (a positive distance other than trunc), left on the first plain trunc-step.
  // Either mode consumes exactly the samples of the sequential march, at bitwise the same t.
  struct Sample {
    bool inws, exists, valid;
    float D;
    int b0, b1, b2, lin;
  };
  auto locate = [&](float tt, int& b0, int& b1, int& b2, int& lin) {  // block / voxel containing the ray point at tt
    const float p0 = o[0] + tt * dL[0], p1 = o[1] + tt * dL[1], p2 = o[2] + tt * dL[2];
    int q0, q1, q2;
    if constexpr (BDIV) {
      const float c_v = mc.v;
      b0 = ifloor(p0 / c_bs), b1 = ifloor(p1 / c_bs), b2 = ifloor(p2 / c_bs);
      q0 = ifloor((p0 - (float)b0 * c_bs) / c_v), q1 = ifloor((p1 - (float)b1 * c_bs) / c_v), q2 = ifloor((p2 - (float)b2 * c_bs) / c_v);
    } else {
      b0 = ifloor(p0 * c_inv_bs), b1 = ifloor(p1 * c_inv_bs), b2 = ifloor(p2 * c_inv_bs);
      q0 = ifloor((p0 - (float)b0 * c_bs) * c_inv_v), q1 = ifloor((p1 - (float)b1 * c_bs) * c_inv_v),
      q2 = ifloor((p2 - (float)b2 * c_bs) * c_inv_v);
    }
    q0 = q0 < 0 ? 0 : (q0 > 7 ? 7 : q0);
    q1 = q1 < 0 ? 0 : (q1 > 7 ? 7 : q1);
    q2 = q2 < 0 ? 0 : (q2 > 7 ? 7 : q2);
    lin = (q0 * 8 + q1) * 8 + q2;
    return inws(b0, b1, b2);  // blocks are only ever allocated inside the workspace bounds
  };
  auto sample = [&](float tt) {
    Sample S;
    S.exists = S.valid = false;
    S.D = 0.0f;
    S.inws = locate(tt, S.b0, S.b1, S.b2, S.lin);
    if (S.inws) {
      const int slot = T.dense ? (int)T.dense[cell_of(S.b0, S.b1, S.b2)] - 1 : hash_find(T, pack_key(S.b0, S.b1, S.b2));
      if (slot >= 0) {
        S.exists = true;
        if (T.block_free[slot]) {
          S.valid = true;
          S.D = c_trunc;
        } else {
          const float2 dw = reinterpret_cast<const float2*>(T.pool)[(size_t)slot * kVPB + S.lin];
          float w = dw.y;
          if (LAZY) {
            for (int lag = T.cur_epoch - T.epoch[slot]; lag > 0; --lag) w = w * T.lag_f;
          }
          if (w > 1e-4f) {
            S.valid = true;
            S.D = dw.x;
          }
        }
      }
    }
    return S;
  };
  bool narrow = false;
#ifdef MMF_WG_TRACE
  int n_wide = 0, n_narrow = 0;
#endif
  while (!done) {
    if (narrow) {
#ifdef MMF_WG_TRACE
      ++n_narrow;
#endif
      const Sample S = sample(k == 0 ? t : t + (float)k * c_half_v);
      int hold = 0;  // the probe that holds the march's current sample (lane 0's at first); -1: a point outside the workspace
      for (int step = 0; step < kRayLanes; ++step) {
        const int src = gshift + (hold < 0 ? 0 : hold);
        const bool v_valid = hold >= 0 && __shfl((int)S.valid, src, 64) != 0;
        const float v_D = __shfl(S.D, src, 64);
        if (!(i < mc.st_max_steps && t < c_max_len) || !v_valid) {  // (last_pos holds in this mode: unobserved = failure)
          done = true;
          break;
        }
        if (v_D < c_eps) {  // the surface, reached from a valid positive distance
          t = t + v_D;
          ok = true;
          done = true;
          break;
        }
        const float tn = t + v_D;
        const bool plain = tn == t + c_trunc;
        t = tn;
        ++i;
        if (plain) {  // out of the band: speculate again
          narrow = false;
          break;
        }
        int n0, n1, n2, nlin;
        if (!locate(t, n0, n1, n2, nlin)) {
          hold = -1;
          continue;
        }
        const bool mine = S.inws && S.b0 == n0 && S.b1 == n1 && S.b2 == n2 && (!S.exists || S.lin == nlin);
        const unsigned match = (unsigned)((__ballot(mine) >> gshift) & 0xffffu);
        if (match == 0u) break;  // not fetched: the next round probes from here
        hold = __ffs(match) - 1;
      }
      continue;
    }
#ifdef MMF_WG_TRACE
    ++n_wide;
#endif
    // this lane's sample: the march position after k further trunc-steps (k sequential float additions, as the march makes
    // them; straight-line add + select instead of a lane-divergent loop)
    float tk = t;
#pragma unroll
    for (int j = 0; j < kRayLanes - 1; ++j) {
      const float nx = tk + c_trunc;
      tk = j < k ? nx : tk;
    }
    const float tk_next_guess = tk + c_trunc;  // == lane k+1's tk
    const Sample S = sample(tk);
    const bool valid = S.valid;
    const float D = S.D;
    const bool pos = valid && !(D < c_eps);  // a sample that sets "previous sample was a valid positive distance"
    const unsigned mpos = (unsigned)((__ballot(pos) >> gshift) & 0xffffu);
    const bool last_pos_k = last_pos || (mpos & ((1u << k) - 1u)) != 0u;
    const bool bound = !((i + k) < mc.st_max_steps && tk < c_max_len);
    const bool fail_unobserved = !valid && (last_pos_k || tk > t_exit);
    const bool surface = valid && D < c_eps;
    const float tnext = tk + (valid ? D : c_trunc);
    const bool deviate = pos && !(tnext == tk_next_guess);
    const unsigned mev = (unsigned)((__ballot(bound || fail_unobserved || surface || deviate) >> gshift) & 0xffffu);
    if (mev == 0u) {  // 16 plain trunc-steps
      t = __shfl(tnext, gshift + kRayLanes - 1, 64);
      i += kRayLanes;
      last_pos = last_pos || mpos != 0u;
      continue;
    }
    const int e = __ffs(mev) - 1;  // first lane whose sample ends the run
    const int src = gshift + e;
    const bool e_bound = __shfl((int)bound, src, 64) != 0;
    const bool e_fail = __shfl((int)fail_unobserved, src, 64) != 0;
    const bool e_surface = __shfl((int)surface, src, 64) != 0;
    const bool e_last = __shfl((int)last_pos_k, src, 64) != 0;
    const float e_tk = __shfl(tk, src, 64), e_D = __shfl(D, src, 64), e_tnext = __shfl(tnext, src, 64);
    if (e_bound || e_fail) {
      done = true;
    } else if (e_surface) {
      if (e_last) {
        t = e_tk + e_D;
        ok = true;
      }
      done = true;
    } else {  // a positive distance other than trunc: resume the march from there, through the fetched voxels ahead
      t = e_tnext;
      i += e + 1;
      last_pos = true;
      narrow = true;
    }
  }
  if (k == 0) synth[idx] = ok ? t * dC[2] : -1.0f;
#ifdef MMF_WG_TRACE
  if (trace_iters) {
    atomicMax(&trace_iters[0], n_wide);
    atomicMax(&trace_iters[1], n_narrow);
  }
#endif
}

template <bool LAZY, bool BDIV = false>
__global__ __launch_bounds__(256) void k_sphere_trace(LayerDev T, MapConsts mc, Cam cam, Rigid T_L_C, float* __restrict__ synth,
                                                     int Ws, int Hs, int patches_x) {
  sphere_patch<LAZY, BDIV>(T, mc, cam, T_L_C, synth, Ws, Hs, patches_x, blockIdx.x, threadIdx.x);
}

// Large pools: the scalable colour + feature allocation (alloc_big_body: [job 0 chunks | job 1 chunks], look-back chains) and the
// sphere trace as roles of ONE launch -- the allocation reads the candidate flags of the TSDF pass and touches only the appearance
// layers' lists and index, the trace reads the TSDF layer: independent, as in the bounded workspace's k_sphere_alloc.  The chunks
// lead the grid (their look-back needs its predecessors resident first).  FLAT (1: from the feature image, 2: from a low-res map): in a
// pipelined stream the previous frame's row update follows as a fourth role (as in k_sphere_alloc_flat).
template <bool LAZY, int FLAT>
__global__ __launch_bounds__(256) void k_sphere_alloc_big(LayerDev T, MapConsts mc, Cam cam, Rigid T_L_C, float* __restrict__ synth, int Ws, int Hs,
                                                         int patches_x, int n_patches, AllocJob J0, AllocJob J1, int nwg0, int nwg1, int G0, int G1,
                                                         long long* stats, AppArgs F, int lpv, int n_flat) {
  __shared__ AllocBigLds S;
  __shared__ int s_prefix[FLAT ? kFlatSubLists + 1 : 1];
  int b = (int)blockIdx.x;
  if (b < nwg0) {
    if (J0.ks.mode == 0) alloc_big_body<0>(J0, stats, S, b, nwg0, G0);
    else alloc_big_body<1>(J0, stats, S, b, nwg0, G0);
    return;
  }
  b -= nwg0;
  if (b < nwg1) {
    if (J1.ks.mode == 0) alloc_big_body<0>(J1, stats, S, b, nwg1, G1);
    else alloc_big_body<1>(J1, stats, S, b, nwg1, G1);
    return;
  }
  b -= nwg1;
  if (!FLAT || b < n_patches) return sphere_patch<LAZY>(T, mc, cam, T_L_C, synth, Ws, Hs, patches_x, b, threadIdx.x);
  if constexpr (FLAT != 0) feature_flat_role<FLAT == 2>(F, mc, lpv, b - n_patches, n_flat, s_prefix);
}

// Horizontal fusion: the sphere trace (reads the TSDF layer) and the block allocation of the colour and the feature layer
// (touch only their own hash / lists; inputs = the candidate flags) are independent once the TSDF update and the candidate
// selection are done.  Workgroups [0, njobs) run one allocation job each, the others one 4x4 ray patch each.
// 256-thread workgroups at <= 102 registers: five fit a compute unit, so all ~1 200 patches of a 640x480 frame are resident
// at once.  (With four patches per 1 024-thread workgroup only 256 of the 300 workgroups fitted, and the 44 late ones -- each
// as long as any other -- doubled the duration of the launch.)  The allocation jobs run in the 4-wave form: three passes
// instead of one, still well inside the trace's shadow.
struct SphereLds {
  int lds[34];
  int carry[2];
  int ctx[4];
};

template <bool DENSE, int MODE>
__device__ inline void sphere_alloc_role(const SphereArgs& A, SphereLds& Q, int job) {
  const long long tr0 = wg_trace_begin();
  alloc_job_body<DENSE, MODE, 4>(job == 0 ? A.J0 : A.J1, A.stats, Q.lds, Q.carry, Q.ctx);
  wg_trace_end(tr0, kTrSphereAlloc);
}

__device__ inline void sphere_patch_role(const SphereArgs& A, SphereLds& Q, int patch) {
  const long long tr0 = wg_trace_begin();
  if (A.patch_flags && patch < A.n_patches) {  // (workgroup-uniform) skip patches no gate of this frame can read
    int mine = 0;
    if (threadIdx.x < 9) {
      const int patches_y = A.n_patches / A.patches_x;
      const int px = patch % A.patches_x + (int)(threadIdx.x % 3) - 1, py = patch / A.patches_x + (int)(threadIdx.x / 3) - 1;
      if (px >= 0 && py >= 0 && px < A.patches_x && py < patches_y) mine = A.patch_flags[py * A.patches_x + px] == (uint8_t)A.patch_tag;
    }
    if (!__syncthreads_or(mine)) {
      wg_trace_end(tr0, kTrSphereTrace);
      return;
    }
  }
  if (wg_trace_on()) {  // diagnostics only: longest wide / narrow iteration counts of the workgroup ride in the record id
    if (threadIdx.x < 2) Q.lds[threadIdx.x] = 0;
    __syncthreads();
    if (patch < A.n_patches) sphere_patch(A.T, A.mc, A.cam, A.T_L_C, A.synth, A.Ws, A.Hs, A.patches_x, patch, threadIdx.x, Q.lds);
    __syncthreads();
    wg_trace_end(tr0, kTrSphereTrace + (Q.lds[0] << 8) + (Q.lds[1] << 20));
    return;
  }
  if (patch < A.n_patches) sphere_patch(A.T, A.mc, A.cam, A.T_L_C, A.synth, A.Ws, A.Hs, A.patches_x, patch, threadIdx.x);
}

template <bool DENSE, int MODE>
__global__ __launch_bounds__(256, 5) void k_sphere_alloc(SphereArgs A) {
  __shared__ SphereLds Q;
  if ((int)blockIdx.x < A.njobs) return sphere_alloc_role<DENSE, MODE>(A, Q, (int)blockIdx.x);
  sphere_patch_role(A, Q, (int)blockIdx.x - A.njobs);
}

// two frames (mmf_integrate_frame_multi): [allocation jobs 0 | allocation jobs 1 | ray patches 0 | ray patches 1]
template <bool DENSE, int MODE>
__global__ __launch_bounds__(256, 5) void k_sphere_alloc2(SphereArgs A0, SphereArgs A1) {
  __shared__ SphereLds Q;
  int b = (int)blockIdx.x;
  if (b < A0.njobs) return sphere_alloc_role<DENSE, MODE>(A0, Q, b);
  b -= A0.njobs;
  if (b < A1.njobs) return sphere_alloc_role<DENSE, MODE>(A1, Q, b);
  b -= A1.njobs;
  if (b < A0.n_patches) return sphere_patch_role(A0, Q, b);
  sphere_patch_role(A1, Q, b - A0.n_patches);
}

// N frames (mmf_integrate_frame_batch): [allocation jobs 0 .. n-1 | ray patches 0 .. n-1]
template <bool DENSE, int MODE>
__global__ __launch_bounds__(256, 8) void k_sphere_alloc_batch(SphereBatch P) {
  __shared__ SphereLds Q;
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    if (b < P.a[q].njobs) return sphere_alloc_role<DENSE, MODE>(P.a[q], Q, b);
    b -= P.a[q].njobs;
  }
  for (int q = 0; q < P.n; ++q) {
    if (b < P.a[q].n_patches) return sphere_patch_role(P.a[q], Q, b);
    b -= P.a[q].n_patches;
  }
}


template <bool DIV, int FMA>
__global__ __launch_bounds__(256) void k_color_integrate(AppArgs A, MapConsts mc, const float* __restrict__ synth, int Ws, int Hs) {
  color_body<DIV, FMA>(A, mc, synth, Ws, Hs, blockIdx.x, gridDim.x);
}


template <bool LOW, int FMA = 0>
__global__ __launch_bounds__(256) void k_feature_flat(AppArgs A, MapConsts mc, int lpv) {
  __shared__ int s_prefix[kFlatSubLists + 1];
  feature_flat_role<LOW, FMA>(A, mc, lpv, (int)blockIdx.x, (int)gridDim.x, s_prefix);
}

// two frames' survivor lists in one launch: the first nb0 workgroups walk list 0, the rest list 1
template <bool LOW>
__global__ __launch_bounds__(256) void k_feature_flat2(AppArgs A0, MapConsts mc0, AppArgs A1, MapConsts mc1, int lpv, int nb0) {
  __shared__ int s_prefix[kFlatSubLists + 1];
  if ((int)blockIdx.x < nb0)
    feature_flat_role<LOW>(A0, mc0, lpv, (int)blockIdx.x, nb0, s_prefix);
  else
    feature_flat_role<LOW>(A1, mc1, lpv, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, s_prefix);
}

// The PREVIOUS frame's row update as a role of this frame's sphere-trace launch (mmf_set_deferred_feature_rows): the rows of
// frame N are independent of everything frame N + 1 does before its own row update (they are read by map consumers only, and
// those flush first), the tracer is a chain of dependent rounds at 53 registers on 1 202 of the chip's 2 048 workgroup slots
// and 0.03 of the HBM rate, the row update is a bandwidth stream of ~1 200 short workgroups at 60 registers: they fill the
// empty slots and end long before the slowest ray patch.  Grid: [allocation jobs | ray patches | row-update workgroups] --
// the patches first, each is as long as the launch.
// LOW: the row update samples a low-res feature map (101 registers: five workgroups per CU -- the ray patches still all start at
// once, the rows take the slots they leave; at 768 channels the rows are the longer role and the trace disappears behind THEM).
template <bool DENSE, int MODE, bool LOW>
__global__ __launch_bounds__(256, 5) void k_sphere_alloc_flat(SphereArgs A, AppArgs F, int lpv, int n_flat) {
  __shared__ SphereLds Q;
  __shared__ int s_prefix[kFlatSubLists + 1];
  int b = (int)blockIdx.x;
  if (b < A.njobs) return sphere_alloc_role<DENSE, MODE>(A, Q, b);
  b -= A.njobs;
  if (b < A.n_patches) return sphere_patch_role(A, Q, b);
  feature_flat_role<LOW>(F, A.mc, lpv, b - A.n_patches, n_flat, s_prefix);
}

// N frames: [allocation jobs 0 .. n-1 | ray patches 0 .. n-1 | pending row updates of the previous frames of the batch's mappers]
template <bool DENSE, int MODE, bool LOW>
__global__ __launch_bounds__(256, LOW ? 5 : 8) void k_sphere_alloc_batch_flat(SphereBatch P, FlatBatch F) {
  __shared__ SphereLds Q;
  __shared__ int s_prefix[kFlatSubLists + 1];
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    if (b < P.a[q].njobs) return sphere_alloc_role<DENSE, MODE>(P.a[q], Q, b);
    b -= P.a[q].njobs;
  }
  for (int q = 0; q < P.n; ++q) {
    if (b < P.a[q].n_patches) return sphere_patch_role(P.a[q], Q, b);
    b -= P.a[q].n_patches;
  }
  for (int q = 0; q < F.n; ++q) {
    if (b < F.nb[q]) return feature_flat_role<LOW>(F.a[q], F.mc[q], F.lpv[q], b, F.nb[q], s_prefix);
    b -= F.nb[q];
  }
}

// N frames' survivor lists in one launch: frame q's list is walked by its own nb[q] workgroups
template <bool LOW>
__global__ __launch_bounds__(256) void k_feature_flat_batch(FlatBatch P) {
  __shared__ int s_prefix[kFlatSubLists + 1];
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    if (b < P.nb[q]) return feature_flat_role<LOW>(P.a[q], P.mc[q], P.lpv[q], b, P.nb[q], s_prefix);
    b -= P.nb[q];
  }
}


template <bool LOW, bool DIV, int FMA>
__global__ __launch_bounds__(256) void k_feature_integrate(AppArgs A, MapConsts mc, const float* __restrict__ synth, int Ws,
                                                          int Hs) {
  __shared__ FeatLds S;
  feature_body<LOW, DIV, FMA>(A, mc, synth, Ws, Hs, blockIdx.x, gridDim.x, S);
}

// Horizontal fusion: colour and feature update of one frame in ONE launch (different layers, same TSDF / synthetic
// depth inputs): the first g_col workgroups walk the colour candidates, the rest the feature candidates.
template <bool LOW, int FMA>
__global__ __launch_bounds__(256) void k_app_integrate2(AppArgs Acol, AppArgs Afeat, MapConsts mc, const float* __restrict__ synth,
                                                       int Ws, int Hs, int g_col) {
  __shared__ FeatLds S;
  if ((int)blockIdx.x < g_col)
    color_body<false, FMA>(Acol, mc, synth, Ws, Hs, blockIdx.x, g_col);
  else
    feature_body<LOW, false, FMA>(Afeat, mc, synth, Ws, Hs, (int)blockIdx.x - g_col, (int)gridDim.x - g_col, S);
}


// PUB: the frame has a survivor list (every fused frame): publish-only gating, independent of LOW (launched as <false, true>)
template <bool LOW, bool PUB, int FMA = 0>
__global__ __launch_bounds__(256) void k_app_frame(AppArgs Acol, AppArgs Afeat, MapConsts mc, const float* __restrict__ synth, int Ws,
                                                  int Hs) {
  __shared__ FeatLds S;
  const long long tr0 = wg_trace_begin();
  app_frame_body<LOW, PUB, FMA>(Acol, Afeat, mc, synth, Ws, Hs, blockIdx.x, gridDim.x, S);
#ifdef MMF_WG_TRACE
  {  // diagnostics: survivors of the (last) block and whether it was new ride in the record id
    const int n = *Acol.sc.cand_count;
    const int i = xcd_candidate((int)blockIdx.x, (n + 7) >> 3);
    const int extra = (i < n) ? (S.n & 0x3ff) | ((Afeat.sc.cand_new[i] != 0) << 10) | ((Acol.sc.cand_new[i] != 0) << 11) : 0xfff;
    wg_trace_end(tr0, kTrAppFrame + (extra << 8));
  }
#else
  wg_trace_end(tr0, kTrAppFrame);
#endif
}

// two frames' candidate lists in one launch (mmf_integrate_frame_multi)
template <bool LOW, bool PUB>
__global__ __launch_bounds__(256) void k_app_frame2(AppFrameArgs F0, AppFrameArgs F1, int nb0) {
  __shared__ FeatLds S;
  const long long tr0 = wg_trace_begin();
  if ((int)blockIdx.x < nb0)
    app_frame_body<LOW, PUB>(F0.Ac, F0.Af, F0.mc, F0.synth, F0.Ws, F0.Hs, (int)blockIdx.x, nb0, S);
  else
    app_frame_body<LOW, PUB>(F1.Ac, F1.Af, F1.mc, F1.synth, F1.Ws, F1.Hs, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, S);
  wg_trace_end(tr0, kTrAppFrame);
}

// N frames' candidate lists in one launch
template <bool LOW, bool PUB>
__global__ __launch_bounds__(256) void k_app_frame_batch(AppFrameBatch P) {
  __shared__ FeatLds S;
  const long long tr0 = wg_trace_begin();
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    const AppFrameArgs& F = P.a[q];
    if (b < F.nb) {
      app_frame_body<LOW, PUB>(F.Ac, F.Af, F.mc, F.synth, F.Ws, F.Hs, b, F.nb, S);
      break;
    }
    b -= F.nb;
  }
  wg_trace_end(tr0, kTrAppFrame);
}

MMF_DEFINE_WG_TRACE_SETTER(set_wg_trace_app)

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static inline int grid8(int upper, int cap) {
  int g = upper < cap ? upper : cap;
  g = (g + 7) & ~7;
  return g < 8 ? 8 : g;
}

void launch_app_candidates(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, uint8_t* flags,
                           u64* cell_key, hipStream_t s) {
  if (mc.spec_flags & kSpecFma)
    hipLaunchKernelGGL(k_app_candidates<true>, dim3(grid8(hinted(tsdf.hint_live, tsdf.cap), 8192)), dim3(256), 0, s, tsdf, mc, cam, T_C_L,
                       flags, cell_key);
  else
    hipLaunchKernelGGL(k_app_candidates<false>, dim3(grid8(hinted(tsdf.hint_live, tsdf.cap), 8192)), dim3(256), 0, s, tsdf, mc, cam, T_C_L,
                       flags, cell_key);
}

void launch_sphere_trace(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, float* synth, int Ws,
                         int Hs, hipStream_t s) {
  const int patches_x = (Ws + 3) / 4, patches_y = (Hs + 3) / 4;
  const int n = patches_x * patches_y;
  if (n <= 0) return;
  const bool bdiv = (mc.spec_flags & kSpecBlockDiv) != 0;  // mmf_params.block_index_by_division
  if (tsdf.epoch) {  // a lazily decayed map: sampled weights are brought up to date on the fly
    if (bdiv)
      hipLaunchKernelGGL((k_sphere_trace<true, true>), dim3(n), dim3(256), 0, s, tsdf, mc, cam, T_L_C, synth, Ws, Hs, patches_x);
    else
      hipLaunchKernelGGL((k_sphere_trace<true, false>), dim3(n), dim3(256), 0, s, tsdf, mc, cam, T_L_C, synth, Ws, Hs, patches_x);
  } else if (bdiv)
    hipLaunchKernelGGL((k_sphere_trace<false, true>), dim3(n), dim3(256), 0, s, tsdf, mc, cam, T_L_C, synth, Ws, Hs, patches_x);
  else
    hipLaunchKernelGGL((k_sphere_trace<false, false>), dim3(n), dim3(256), 0, s, tsdf, mc, cam, T_L_C, synth, Ws, Hs, patches_x);
}

static int flat_lanes_per_voxel(const MapConsts& mc);
static int flat_grid(const FlatList& fl, int lpv);
void launch_sphere_alloc_big(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, float* synth, int Ws, int Hs,
                             const AllocJob* jobs, long long* stats, const AppArgs* rows, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int patches_x = (Ws + 3) / 4, patches_y = (Hs + 3) / 4;
  const int n = patches_x * patches_y > 0 ? patches_x * patches_y : 0;
  const int G0 = alloc_big_groups(jobs[0].ncells), G1 = alloc_big_groups(jobs[1].ncells);
  const int n0 = (alloc_big_wgs(jobs[0].ncells) + G0 - 1) / G0, n1 = (alloc_big_wgs(jobs[1].ncells) + G1 - 1) / G1;
  const bool lazy = tsdf.epoch != nullptr;  // a lazily decayed map: sampled weights are brought up to date on the fly
  const int lpv = flat_lanes_per_voxel(mc);
  const int n_flat = rows ? flat_grid(rows->flat, lpv) : 0;
  const int flat = !rows ? 0 : (rows->low.data ? 2 : 1);
  const AppArgs F = rows ? *rows : AppArgs{};
  const dim3 grid(n0 + n1 + n + n_flat);
#define MMF_SAB(LAZYV, FLATV)                                                                                                                  \
  hipExtLaunchKernelGGL((k_sphere_alloc_big<LAZYV, FLATV>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, tsdf, mc, cam, T_L_C, synth, Ws, Hs, patches_x, n, \
                        jobs[0], jobs[1], n0, n1, G0, G1, stats, F, lpv, n_flat)
  if (lazy) {
    if (flat == 0) MMF_SAB(true, 0);
    else if (flat == 1) MMF_SAB(true, 1);
    else MMF_SAB(true, 2);
  } else {
    if (flat == 0) MMF_SAB(false, 0);
    else if (flat == 1) MMF_SAB(false, 1);
    else MMF_SAB(false, 2);
  }
#undef MMF_SAB
}

SphereArgs make_sphere_args(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, float* synth, int Ws, int Hs,
                            const AllocJob* jobs, int njobs, long long* stats) {
  SphereArgs A;
  A.T = tsdf;
  A.mc = mc;
  A.cam = cam;
  A.T_L_C = T_L_C;
  A.synth = synth;
  A.Ws = Ws;
  A.Hs = Hs;
  A.patches_x = (Ws + 3) / 4;
  A.n_patches = A.patches_x * ((Hs + 3) / 4);
  A.J0 = jobs[0];
  A.J1 = jobs[njobs > 1 ? 1 : 0];
  A.njobs = njobs;
  A.stats = stats;
  return A;
}

static inline bool sphere_dense(const SphereArgs& A) {  // bounded workspace, list cells: no hash paths
  return A.J0.L.dense && A.J1.L.dense && A.J0.ks.mode == 1 && A.J1.ks.mode == 1;
}

void launch_sphere_alloc(const SphereArgs* A, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n == 1) {
    const dim3 grid(A[0].njobs + A[0].n_patches);
    if (sphere_dense(A[0]))
      hipExtLaunchKernelGGL((k_sphere_alloc<true, 1>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A[0]);
    else
      hipExtLaunchKernelGGL((k_sphere_alloc<false, -1>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A[0]);
  } else {
    const dim3 grid(A[0].njobs + A[1].njobs + A[0].n_patches + A[1].n_patches);
    if (sphere_dense(A[0]) && sphere_dense(A[1]))
      hipExtLaunchKernelGGL((k_sphere_alloc2<true, 1>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A[0], A[1]);
    else
      hipExtLaunchKernelGGL((k_sphere_alloc2<false, -1>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A[0], A[1]);
  }
}

void launch_sphere_alloc_batch(const SphereArgs* A, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  SphereBatch P;
  P.n = n;
  int total = 0;
  bool dense = true;
  for (int q = 0; q < n; ++q) {
    P.a[q] = A[q];
    total += A[q].njobs + A[q].n_patches;
    dense = dense && sphere_dense(A[q]);
  }
  if (dense)
    hipExtLaunchKernelGGL((k_sphere_alloc_batch<true, 1>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
  else
    hipExtLaunchKernelGGL((k_sphere_alloc_batch<false, -1>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
}

static int flat_lanes_per_voxel(const MapConsts& mc);
static int flat_grid(const FlatList& fl, int lpv);

// (rows: all from full-resolution feature images or all from low-res maps -- the two row updates are different instantiations)
void launch_sphere_alloc_batch_flat(const SphereArgs* A, int n, const AppArgs* rows, const MapConsts* mcs, int nr, hipStream_t s,
                                    hipEvent_t ev_start, hipEvent_t ev_stop) {
  SphereBatch P;
  FlatBatch F;
  P.n = n;
  F.n = nr;
  int total = 0;
  bool dense = true;
  for (int q = 0; q < n; ++q) {
    P.a[q] = A[q];
    total += A[q].njobs + A[q].n_patches;
    dense = dense && sphere_dense(A[q]);
  }
  for (int q = 0; q < nr; ++q) {
    F.a[q] = rows[q];
    F.mc[q] = mcs[q];
    F.lpv[q] = flat_lanes_per_voxel(mcs[q]);
    F.nb[q] = flat_grid(rows[q].flat, F.lpv[q]);
    total += F.nb[q];
  }
  const bool low = nr > 0 && rows[0].low.data != nullptr;
  if (dense && !low)
    hipExtLaunchKernelGGL((k_sphere_alloc_batch_flat<true, 1, false>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P, F);
  else if (dense)
    hipExtLaunchKernelGGL((k_sphere_alloc_batch_flat<true, 1, true>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P, F);
  else if (!low)
    hipExtLaunchKernelGGL((k_sphere_alloc_batch_flat<false, -1, false>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P, F);
  else
    hipExtLaunchKernelGGL((k_sphere_alloc_batch_flat<false, -1, true>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P, F);
}

static AppArgs make_app_args(const LayerDev& L, const Cam& cam, const Rigid& T_C_L, const void* image, const uint8_t* mask,
                             const Scratch& sc, long long* stats = nullptr, const LowRes* low = nullptr,
                             const FlatList* flat = nullptr) {
  AppArgs A;
  A.low = low ? *low : LowRes{};
  A.flat = flat ? *flat : FlatList{};
  A.stats = stats;
  A.L = L;
  A.cam = cam;
  A.T_C_L = T_C_L;
  A.image = image;
  A.mask = mask;
  A.sc = sc;
  return A;
}

void launch_color_integrate(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const uint8_t* rgb,
                            const uint8_t* mask, const float* synth, int Ws, int Hs, const Scratch& sc, int max_cand,
                            hipStream_t s) {
  const dim3 grid(grid8(hinted(sc.hint_cand, max_cand), 8192));
  const bool div = (mc.spec_flags & 2) != 0;  // mmf_params.appearance_blend_division
  const int ar = arith_mode(mc.spec_flags);   // bit 0 .fma_contraction, bit 1 .bilinear_four_weight_sum
  const AppArgs A = make_app_args(L, cam, T_C_L, rgb, mask, sc);
#define MMF_CI(DIVV, ARV) hipLaunchKernelGGL((k_color_integrate<DIVV, ARV>), grid, dim3(256), 0, s, A, mc, synth, Ws, Hs)
  if (div) {
    if (ar == 3) MMF_CI(true, 3);
    else if (ar == 2) MMF_CI(true, 2);
    else if (ar == 1) MMF_CI(true, 1);
    else MMF_CI(true, 0);
  } else {
    if (ar == 3) MMF_CI(false, 3);
    else if (ar == 2) MMF_CI(false, 2);
    else if (ar == 1) MMF_CI(false, 1);
    else MMF_CI(false, 0);
  }
#undef MMF_CI
}

static AppArgs make_app_args(const LayerDev& L, const Cam& cam, const Rigid& T_C_L, const void* image, const uint8_t* mask,
                             const Scratch& sc, long long* stats, const LowRes* low, const FlatList* flat);
static int flat_lanes_per_voxel(const MapConsts& mc);
static int flat_grid(const FlatList& fl, int lpv);

// balanced phase 2 over the frame's survivor list (enqueued right behind the gating launch)
static int flat_lanes_per_voxel(const MapConsts& mc) {
  const int nch = mc.C >> 3;
  return nch <= 8 ? 8 : nch <= 16 ? 16 : nch <= 32 ? 32 : (nch % 64 == 0 ? 64 : 32);
}
static int flat_grid(const FlatList& fl, int lpv) {
  const int vpw = 256 / lpv;
  const long long vox = hinted(fl.hint, fl.cap);
  long long wgs = (vox + vpw - 1) / vpw;
  wgs = wgs > 16384 ? 16384 : wgs;
  return grid8((int)wgs, 16384);
}

AppArgs make_flat_args(const LayerDev& L, const Cam& cam, const __half* feat, const LowRes* lowres, const FlatList& fl, long long* stats) {
  return make_app_args(L, cam, Rigid{}, feat, nullptr, Scratch{}, stats, lowres, &fl);
}

template <bool LOW>
static void launch_flat_kernel(dim3 grid, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, const AppArgs& Af, const MapConsts& mc, int lpv) {
  switch (arith_mode(mc.spec_flags)) {  // bit 0 mmf_params.fma_contraction, bit 1 .bilinear_four_weight_sum
    case 1: hipExtLaunchKernelGGL((k_feature_flat<LOW, 1>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, Af, mc, lpv); break;
    case 2: hipExtLaunchKernelGGL((k_feature_flat<LOW, 2>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, Af, mc, lpv); break;
    case 3: hipExtLaunchKernelGGL((k_feature_flat<LOW, 3>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, Af, mc, lpv); break;
    default: hipExtLaunchKernelGGL((k_feature_flat<LOW, 0>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, Af, mc, lpv); break;
  }
}

// the stand-alone row update from a saved argument block (the flush of a deferred one)
void launch_feature_flat_args(const AppArgs& Af, const MapConsts& mc, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (!Af.flat.rec) return;
  const int lpv = flat_lanes_per_voxel(mc);
  const dim3 grid((unsigned)flat_grid(Af.flat, lpv));
  if (Af.low.data)
    launch_flat_kernel<true>(grid, s, ev_start, ev_stop, Af, mc, lpv);
  else
    launch_flat_kernel<false>(grid, s, ev_start, ev_stop, Af, mc, lpv);
}

// sphere trace | colour allocation | feature allocation of this frame | row update of the previous one
void launch_sphere_alloc_flat(const SphereArgs& A, const AppArgs& F, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int lpv = flat_lanes_per_voxel(A.mc);
  const int n_flat = flat_grid(F.flat, lpv);
  const dim3 grid(A.njobs + A.n_patches + n_flat);
  const bool low = F.low.data != nullptr;
  if (sphere_dense(A) && !low)
    hipExtLaunchKernelGGL((k_sphere_alloc_flat<true, 1, false>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A, F, lpv, n_flat);
  else if (sphere_dense(A))
    hipExtLaunchKernelGGL((k_sphere_alloc_flat<true, 1, true>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A, F, lpv, n_flat);
  else if (!low)
    hipExtLaunchKernelGGL((k_sphere_alloc_flat<false, -1, false>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A, F, lpv, n_flat);
  else
    hipExtLaunchKernelGGL((k_sphere_alloc_flat<false, -1, true>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A, F, lpv, n_flat);
}

void launch_feature_flat(const LayerDev& L, const MapConsts& mc, const Cam& cam, const __half* feat, const LowRes* lowres,
                         const FlatList& fl, long long* stats, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (!fl.rec) return;
  const bool low = lowres != nullptr;
  const AppArgs Af = make_app_args(L, cam, Rigid{}, feat, nullptr, Scratch{}, stats, lowres, &fl);
  const int lpv = flat_lanes_per_voxel(mc);
  const dim3 grid((unsigned)flat_grid(fl, lpv));
  // with events: the extension launch stamps them with the dispatch's own begin / end times (what rocprofv3 reports as
  // the kernel duration), no marker packets around the kernel
  if (low)
    launch_flat_kernel<true>(grid, s, ev_start, ev_stop, Af, mc, lpv);
  else
    launch_flat_kernel<false>(grid, s, ev_start, ev_stop, Af, mc, lpv);
}

void launch_feature_integrate(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const __half* feat,
                              const uint8_t* mask, const float* synth, int Ws, int Hs, const Scratch& sc, int max_cand,
                              long long* stats, hipStream_t s, const LowRes* low, const FlatList* flat) {
  const dim3 grid(grid8(hinted(sc.hint_cand, max_cand), 8192));
  const bool div = (mc.spec_flags & 2) != 0;  // mmf_params.appearance_blend_division: rows are updated inside the gating workgroup
  const AppArgs A = make_app_args(L, cam, T_C_L, feat, mask, sc, stats, low, div ? nullptr : flat);
  const int ar = arith_mode(mc.spec_flags);  // bit 0 mmf_params.fma_contraction, bit 1 .bilinear_four_weight_sum
#define MMF_FI(LOWV, DIVV, ARV) hipLaunchKernelGGL((k_feature_integrate<LOWV, DIVV, ARV>), grid, dim3(256), 0, s, A, mc, synth, Ws, Hs)
#define MMF_FI4(ARV)                              \
  do {                                            \
    if (low && div) MMF_FI(true, true, ARV);      \
    else if (low) MMF_FI(true, false, ARV);       \
    else if (div) MMF_FI(false, true, ARV);       \
    else MMF_FI(false, false, ARV);               \
  } while (0)
  if (ar == 3) MMF_FI4(3);
  else if (ar == 2) MMF_FI4(2);
  else if (ar == 1) MMF_FI4(1);
  else MMF_FI4(0);
#undef MMF_FI4
#undef MMF_FI
}

AppTail make_app_tail(const LayerDev& Lc, const Cam& ccam, const uint8_t* rgb, const uint8_t* cmask, const Scratch& csc, const LayerDev& Lf,
                      const Cam& fcam, const __half* feat, const uint8_t* fmask, const Scratch& fsc, const MapConsts& mc, const Rigid& T_C_L,
                      const float* synth, int Ws, int Hs, int max_cand, long long* stats, const FlatList& flat) {
  AppTail T;
  T.Ac = make_app_args(Lc, ccam, T_C_L, rgb, cmask, csc, nullptr, nullptr, nullptr);
  T.Af = make_app_args(Lf, fcam, T_C_L, feat, fmask, fsc, stats, nullptr, &flat);
  T.mc = mc;
  T.synth = synth;
  T.Ws = Ws;
  T.Hs = Hs;
  T.max_cand = max_cand;
  return T;
}

AppFrameArgs app_frame_args_of(const AppTail& T) {
  AppFrameArgs F;
  F.Ac = T.Ac;
  F.Af = T.Af;
  F.mc = T.mc;
  F.synth = T.synth;
  F.Ws = T.Ws;
  F.Hs = T.Hs;
  F.nb = app_tail_grid(T);
  return F;
}

AppTail app_tail_of(const AppFrameArgs& F, int max_cand) {
  AppTail T;
  T.Ac = F.Ac;
  T.Af = F.Af;
  T.mc = F.mc;
  T.synth = F.synth;
  T.Ws = F.Ws;
  T.Hs = F.Hs;
  T.max_cand = max_cand;
  return T;
}

int app_tail_grid(const AppTail& T) { return grid8(hinted(T.Ac.sc.hint_cand, T.max_cand), 8192); }

void launch_app_tail(const AppTail& T, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
#define MMF_AT(ARV) hipExtLaunchKernelGGL((k_app_frame<false, true, ARV>), dim3(app_tail_grid(T)), dim3(256), 0, s, ev_start, ev_stop, 0, T.Ac, T.Af, T.mc, T.synth, T.Ws, T.Hs)
  switch (arith_mode(T.mc.spec_flags)) {  // bit 0 mmf_params.fma_contraction, bit 1 .bilinear_four_weight_sum
    case 1: MMF_AT(1); break;
    case 2: MMF_AT(2); break;
    case 3: MMF_AT(3); break;
    default: MMF_AT(0); break;
  }
#undef MMF_AT
}

// colour + feature update of one frame (gating launch, then the balanced feature pass when a survivor list is given)
void launch_app_integrate2(const LayerDev& Lc, const Cam& ccam, const uint8_t* rgb, const uint8_t* cmask, const Scratch& csc,
                           const LayerDev& Lf, const Cam& fcam, const __half* feat, const uint8_t* fmask, const Scratch& fsc,
                           const MapConsts& mc, const Rigid& T_C_L, const float* synth, int Ws, int Hs, int max_cand, long long* stats,
                           hipStream_t s, const LowRes* low, const FlatList* flat, bool same_candidates, hipEvent_t ev_start,
                           hipEvent_t ev_stop) {
  const AppArgs Ac = make_app_args(Lc, ccam, T_C_L, rgb, cmask, csc),
                Af = make_app_args(Lf, fcam, T_C_L, feat, fmask, fsc, stats, low, flat);
  // same_candidates: both allocation jobs compacted the same flag array, so candidate i is the same block in both lists
  const bool same_cam = same_candidates && ccam.W == fcam.W && ccam.H == fcam.H && ccam.fx == fcam.fx && ccam.fy == fcam.fy &&
                        ccam.cx == fcam.cx && ccam.cy == fcam.cy;
  const int ar = arith_mode(mc.spec_flags);  // bit 0 mmf_params.fma_contraction, bit 1 .bilinear_four_weight_sum
  if (same_cam) {  // one candidate list, one geometric gate per voxel
    const dim3 grid(grid8(hinted(csc.hint_cand, max_cand), 8192));
#define MMF_AF(LOWV, PUBV, ARV) hipExtLaunchKernelGGL((k_app_frame<LOWV, PUBV, ARV>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, Ac, Af, mc, synth, Ws, Hs)
#define MMF_AF3(ARV)                                      \
  do {                                                    \
    if (flat && flat->rec) MMF_AF(false, true, ARV);      \
    else if (low) MMF_AF(true, false, ARV);               \
    else MMF_AF(false, false, ARV);                       \
  } while (0)
    if (ar == 3) MMF_AF3(3);
    else if (ar == 2) MMF_AF3(2);
    else if (ar == 1) MMF_AF3(1);
    else MMF_AF3(0);
#undef MMF_AF3
#undef MMF_AF
  } else {
    const int gc = grid8(hinted(csc.hint_cand, max_cand), 4096), gf = grid8(hinted(fsc.hint_cand, max_cand), 4096);
#define MMF_AI(LOWV, ARV) hipExtLaunchKernelGGL((k_app_integrate2<LOWV, ARV>), dim3(gc + gf), dim3(256), 0, s, ev_start, ev_stop, 0, Ac, Af, mc, synth, Ws, Hs, gc)
#define MMF_AI2(ARV)               \
  do {                             \
    if (low) MMF_AI(true, ARV);    \
    else MMF_AI(false, ARV);       \
  } while (0)
    if (ar == 3) MMF_AI2(3);
    else if (ar == 2) MMF_AI2(2);
    else if (ar == 1) MMF_AI2(1);
    else MMF_AI2(0);
#undef MMF_AI2
#undef MMF_AI
  }
}

// ---- two frames in one launch (mmf_integrate_frame_multi): same camera for colour and features in each frame ------------
AppFrameArgs make_app_frame_args(const LayerDev& Lc, const Cam& cam, const uint8_t* rgb, const uint8_t* cmask, const Scratch& csc,
                                 const LayerDev& Lf, const __half* feat, const uint8_t* fmask, const Scratch& fsc, const MapConsts& mc,
                                 const Rigid& T_C_L, const float* synth, int Ws, int Hs, int max_cand, long long* stats, const LowRes* low,
                                 const FlatList* flat) {
  AppFrameArgs F;
  F.Ac = make_app_args(Lc, cam, T_C_L, rgb, cmask, csc);
  F.Af = make_app_args(Lf, cam, T_C_L, feat, fmask, fsc, stats, low, flat);
  F.mc = mc;
  F.synth = synth;
  F.Ws = Ws;
  F.Hs = Hs;
  F.nb = grid8(hinted(csc.hint_cand, max_cand), 8192);
  return F;
}

void launch_app_frame2(const AppFrameArgs& F0, const AppFrameArgs& F1, bool low, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const dim3 grid(F0.nb + F1.nb);
  if (F0.Af.flat.rec && F1.Af.flat.rec)
    hipExtLaunchKernelGGL((k_app_frame2<false, true>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, F0, F1, F0.nb);
  else if (low)
    hipExtLaunchKernelGGL((k_app_frame2<true, false>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, F0, F1, F0.nb);
  else
    hipExtLaunchKernelGGL((k_app_frame2<false, false>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, F0, F1, F0.nb);
}

void launch_app_frame_batch(const AppFrameArgs* F, int n, bool low, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  AppFrameBatch P;
  P.n = n;
  int total = 0;
  bool pub = true;
  for (int q = 0; q < n; ++q) {
    P.a[q] = F[q];
    total += F[q].nb;
    pub = pub && F[q].Af.flat.rec != nullptr;
  }
  if (pub)
    hipExtLaunchKernelGGL((k_app_frame_batch<false, true>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
  else if (low)
    hipExtLaunchKernelGGL((k_app_frame_batch<true, false>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
  else
    hipExtLaunchKernelGGL((k_app_frame_batch<false, false>), dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
}

static int flat_wgs(const MapConsts& mc, const FlatList& fl, int& lpv) {
  const int nch = mc.C >> 3;
  lpv = nch <= 8 ? 8 : nch <= 16 ? 16 : nch <= 32 ? 32 : (nch % 64 == 0 ? 64 : 32);  // lanes per voxel row
  const int vpw = 256 / lpv;
  const long long vox = hinted(fl.hint, fl.cap);
  long long wgs = (vox + vpw - 1) / vpw;
  wgs = wgs > 16384 ? 16384 : wgs;
  return grid8((int)wgs, 16384);
}

void launch_feature_flat2(const LayerDev& L0, const MapConsts& mc0, const FlatList& fl0, long long* stats0, const LayerDev& L1,
                          const MapConsts& mc1, const FlatList& fl1, long long* stats1, const Cam& cam, const __half* feat,
                          const LowRes* lowres, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const AppArgs A0 = make_app_args(L0, cam, Rigid{}, feat, nullptr, Scratch{}, stats0, lowres, &fl0);
  const AppArgs A1 = make_app_args(L1, cam, Rigid{}, feat, nullptr, Scratch{}, stats1, lowres, &fl1);
  int lpv0, lpv1;
  const int nb0 = flat_wgs(mc0, fl0, lpv0), nb1 = flat_wgs(mc1, fl1, lpv1);
  const dim3 grid(nb0 + nb1);  // (both mappers of one Mapper object hold the same channel count: lpv0 == lpv1)
  if (lowres)
    hipExtLaunchKernelGGL(k_feature_flat2<true>, grid, dim3(256), 0, s, ev_start, ev_stop, 0, A0, mc0, A1, mc1, lpv0, nb0);
  else
    hipExtLaunchKernelGGL(k_feature_flat2<false>, grid, dim3(256), 0, s, ev_start, ev_stop, 0, A0, mc0, A1, mc1, lpv0, nb0);
}

// (every frame of a batch has the same kind of feature source: all images or all low-res maps)
void launch_feature_flat_batch(const FlatFrame* F, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  FlatBatch P;
  P.n = n;
  int total = 0;
  for (int q = 0; q < n; ++q) {
    P.a[q] = make_app_args(F[q].L, F[q].cam, Rigid{}, F[q].feat, nullptr, Scratch{}, F[q].stats, F[q].low, &F[q].fl);
    P.mc[q] = F[q].mc;
    P.nb[q] = flat_wgs(F[q].mc, F[q].fl, P.lpv[q]);
    total += P.nb[q];
  }
  if (F[0].low)
    hipExtLaunchKernelGGL(k_feature_flat_batch<true>, dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
  else
    hipExtLaunchKernelGGL(k_feature_flat_batch<false>, dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
}

}  // namespace mmf
