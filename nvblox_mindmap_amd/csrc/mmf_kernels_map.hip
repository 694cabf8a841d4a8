// mmf_kernels_map.hip -- voxel-block allocation (raycast marking, flag compaction, hash insertion),
// projective TSDF update, decay, and the layer views.  gfx950 / wave64.
//
// Replaces (through the C ABI in include/mmfusion.h) the CUDA behind nvblox_torch
// Mapper.add_depth_frame / decay / clear / layer views, reached by the reference at
// mindmap/mapping/helpers/nvblox_mapping_helpers.py:207-209 and
// mindmap/mapping/isaaclab_nvblox_mapper.py:252-258.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "mmf_launch.h"
#include "mmf_trace_device.h"
#include "mmf_alloc_device.h"
#include "mmf_mask_device.h"
#include "mmf_app_device.h"

namespace mmf {

static inline int grid_for(int upper, int cap);

// ------------------------------------------------------------------------------------------------
// 1. Blocks in view: one thread per (subsampled) depth pixel walks the block grid from the camera
//    centre to (depth + trunc) along its ray and flags every traversed block of the view grid.
//    Bound: latency / L2 stores; the depth image is read once, coalesced (4 B per lane).
// ------------------------------------------------------------------------------------------------
struct Walk {
  int c[3], g[3], st[3], n;
  float tm[3], dt[3];
};

__device__ inline void walk_init(Walk& w, const float* s, const float* e) {
  w.n = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float fs = floorf(s[a]);
    w.c[a] = (int)fs;
    w.g[a] = ifloor(e[a]);
    int diff = w.g[a] - w.c[a];
    w.n += diff < 0 ? -diff : diff;
    float r = e[a] - s[a];
    w.st[a] = r > 0.0f ? 1 : (r < 0.0f ? -1 : 0);
    if (w.st[a] != 0) {
      float corr = w.st[a] > 0 ? 1.0f : 0.0f;
      float dist = corr - (s[a] - fs);
      w.tm[a] = dist / r;
      w.dt[a] = (float)w.st[a] / r;
    } else {
      w.tm[a] = 2.0f;
      w.dt[a] = 2.0f;
    }
  }
}

__device__ inline void walk_step(Walk& w) {
  // the axis with the smallest boundary parameter among those that have not reached their goal cell; ties and the choice
  // among finished axes resolve to the lowest axis, exactly like a first-unfinished / strictly-smaller scan.  Branch-free:
  // this is the hot loop of the most instruction-bound launch of a frame.
  const bool f0 = w.c[0] == w.g[0], f1 = w.c[1] == w.g[1], f2 = w.c[2] == w.g[2];
  const bool lt10 = w.tm[1] < w.tm[0];
  const bool p1 = !f1 & (f0 | lt10);  // (bitwise on purpose: no short-circuit branches in this loop)
  const float t01 = p1 ? w.tm[1] : w.tm[0];
  const bool lt2 = w.tm[2] < t01;
  const bool p2 = !f2 & ((f0 & f1) | lt2);
  const bool s0 = !p1 & !p2 & !f0, s1 = p1 & !p2;  // p2: axis 2
  w.c[0] += s0 ? w.st[0] : 0;
  w.tm[0] = s0 ? w.tm[0] + w.dt[0] : w.tm[0];
  w.c[1] += s1 ? w.st[1] : 0;
  w.tm[1] = s1 ? w.tm[1] + w.dt[1] : w.tm[1];
  w.c[2] += p2 ? w.st[2] : 0;
  w.tm[2] = p2 ? w.tm[2] + w.dt[2] : w.tm[2];
}

// LDSFLAGS: the view grid has at most kRaycastLdsCells cells, so a workgroup first collects the cells its 256
// rays traverse as byte flags in LDS (plain same-value stores, no atomics) and then writes one global flag
// byte per DISTINCT cell: the hot loop issues no global stores at all.
constexpr int kRaycastLdsCells = 32768;


// SPEC: the stand-alone launch of a mapper with mmf_params.block_index_by_division / .view_truncation_band_marking set (uniform
// branches on MapConsts::spec_flags); the fused kernels instantiate SPEC = false and contain neither.
template <bool LDSFLAGS, bool SPEC = false>
__device__ inline void raycast_body(const RaycastJob& R, int bid, unsigned* s_words) {
  const MapConsts& mc = R.mc;
  const Cam& cam = R.cam;
  const Rigid& T_L_C = R.T_L_C;
  const float* __restrict__ depth = R.depth;
  const uint8_t* __restrict__ mask = R.mask;
  const float min_d = R.min_d;
  const int sub = R.sub, Wsub = R.Wsub, Hsub = R.Hsub;
  const ViewGrid& vg = R.vg;
  uint8_t* __restrict__ flags = R.flags;
  uint8_t* s_flags = reinterpret_cast<uint8_t*>(s_words);
  const int ncells = vg.nx * vg.ny * vg.nz;
  const int nwords = (ncells + 3) >> 2;
  if (LDSFLAGS) {
    for (int w = threadIdx.x; w < nwords; w += 256) s_words[w] = 0u;
    __syncthreads();
  }
  // one wave = one 8x8 tile of (subsampled) pixels: neighbouring rays traverse the same blocks
  const int tiles_x = (Wsub + 7) >> 3;
  const int tile = bid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int cs = (tile % tiles_x) * 8 + (lane & 7), rs = (tile / tiles_x) * 8 + (lane >> 3);
  bool active = cs < Wsub && rs < Hsub;
  int r = rs * sub, c = cs * sub;
  float d = 0.0f;
  if (active) {
    const size_t pix = (size_t)r * cam.W + c;
    d = depth[pix];
    if (!(d > min_d)) active = false;  // min_d >= 0: "depth > 0" and the caller's "depth > min distance" mask in one test
    if (active && mask && ((mask[pix] == 0) != (R.mask_invert != 0))) active = false;
  }
  if (active) {
    if (mc.max_dist > 0.0f && d > mc.max_dist) d = mc.max_dist;
    if (!(d <= 3.4028235e38f)) active = false;  // +inf ("the ray hit nothing") without a maximum distance to clamp it to: no ray
  }
  if (active) {
    float s = d + mc.reach;
    float ray[3] = {((float)c + 0.5f - cam.cx) / cam.fx, ((float)r + 0.5f - cam.cy) / cam.fy, 1.0f};
    float pC[3] = {s * ray[0], s * ray[1], s * ray[2]};
    float pL[3];
    xform(T_L_C, pC, pL);
    float s0[3] = {T_L_C.t[0] * mc.inv_bs, T_L_C.t[1] * mc.inv_bs, T_L_C.t[2] * mc.inv_bs};
    float e[3] = {pL[0] * mc.inv_bs, pL[1] * mc.inv_bs, pL[2] * mc.inv_bs};
    if (SPEC && (mc.spec_flags & kSpecBlockDiv)) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        s0[a] = T_L_C.t[a] / mc.bs;
        e[a] = pL[a] / mc.bs;
      }
    }
    // Only blocks inside the workspace bounds can be in view: the walk starts where the ray enters them, two cells early
    // (oracle/mmf_oracle.c clip_walk_start: same operations in the same order), not at the camera -- a quarter of the steps of a
    // camera that orbits the task's box.
    if (mc.ws_type != 0 && !(mc.spec_flags & 1)) {
      float r[3], t0 = 0.0f, big = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        r[a] = e[a] - s0[a];
        const float ar = fabsf(r[a]);
        if (ar > big) big = ar;
        if (mc.ws_type == 1 && a < 2) continue;
        const float lo = (float)mc.ws_lo[a], hi = (float)(mc.ws_hi[a] + 1);
        float ta = 0.0f;
        if (r[a] > 0.0f) ta = (lo - s0[a]) / r[a];
        else if (r[a] < 0.0f) ta = (hi - s0[a]) / r[a];
        if (ta > t0) t0 = ta;
      }
      if (t0 > 0.0f && big > 0.0f) {
        t0 = t0 - 2.0f / big;
        if (t0 > 0.0f) {
          if (t0 > 1.0f) t0 = 1.0f;
#pragma unroll
          for (int a = 0; a < 3; ++a) s0[a] = s0[a] + t0 * r[a];
        }
      }
    }
    Walk w;
    walk_init(w, s0, e);
    // The walk is monotone along every axis: once it is past the view grid in its direction of travel (or off the grid
    // along an axis it does not move on) no later cell can be inside the grid -> stop.  This bounds the longest rays (far
    // depths clamped to the maximum distance walk ~100 cells, most of them outside).  A coordinate changes by one per
    // step, so "past the grid" is first reached as EQUALITY with the first cell beyond it: the test is made once in full
    // before the loop (a ray that starts beyond never enters) and as three compares per step inside it.
    // From here on the walk lives in grid-relative coordinates (cell (0,0,0) = the grid's origin block).
    const int n3[3] = {vg.nx, vg.ny, vg.nz}, o3[3] = {vg.ox, vg.oy, vg.oz};
    int beyond[3];
    bool never = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      w.c[a] -= o3[a];
      w.g[a] -= o3[a];
      never = never || (w.st[a] >= 0 && w.c[a] >= n3[a]) || (w.st[a] <= 0 && w.c[a] < 0);
      beyond[a] = w.st[a] > 0 ? n3[a] : (w.st[a] < 0 ? -1 : (int)0x80000000);
    }
    for (int i = 0; i <= w.n && !never; ++i) {
      if (w.c[0] == beyond[0] || w.c[1] == beyond[1] || w.c[2] == beyond[2]) break;
      const int gx = w.c[0], gy = w.c[1], gz = w.c[2];
      // the view grid already is the intersection with the workspace bounds
      if ((unsigned)gx < (unsigned)vg.nx && (unsigned)gy < (unsigned)vg.ny && (unsigned)gz < (unsigned)vg.nz) {
        // inside the grid the coordinates are small and non-negative: 24-bit multiplies (full rate; 32-bit ones are quarter rate)
        const int cell = (int)(__umul24(__umul24((unsigned)gx, (unsigned)vg.ny) + (unsigned)gy, (unsigned)vg.nz) + (unsigned)gz);
        if (LDSFLAGS)
          s_flags[cell] = 1;
        else
          flags[cell] = (uint8_t)R.flag_value;
      }
      walk_step(w);
    }
    if (SPEC && (mc.spec_flags & kSpecBandMark)) {
      // the blocks that intersect the cube [p - trunc, p + trunc]^3 around the pixel's surface point p = T_L_C (d * ray)
      // (oracle/mmf_oracle.c blocks_in_view, view_truncation_band_marking); the view grid is the workspace intersection, padded by a block
      const float qC[3] = {d * ray[0], d * ray[1], d * ray[2]};
      float qL[3];
      xform(T_L_C, qC, qL);
      int lo[3], hi[3];
      const bool bdiv = (mc.spec_flags & kSpecBlockDiv) != 0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float l = qL[a] - mc.trunc, h = qL[a] + mc.trunc;
        lo[a] = ifloor(bdiv ? l / mc.bs : l * mc.inv_bs) - o3[a];
        hi[a] = ifloor(bdiv ? h / mc.bs : h * mc.inv_bs) - o3[a];
      }
      for (int gx = lo[0]; gx <= hi[0]; ++gx)
        for (int gy = lo[1]; gy <= hi[1]; ++gy)
          for (int gz = lo[2]; gz <= hi[2]; ++gz)
            if ((unsigned)gx < (unsigned)vg.nx && (unsigned)gy < (unsigned)vg.ny && (unsigned)gz < (unsigned)vg.nz) {
              const int cell = (gx * vg.ny + gy) * vg.nz + gz;
              if (LDSFLAGS)
                s_flags[cell] = 1;
              else
                flags[cell] = (uint8_t)R.flag_value;
            }
    }
  }
  if (LDSFLAGS) {
    __syncthreads();
    for (int wd = threadIdx.x; wd < nwords; wd += 256) {
      const unsigned v = s_words[wd];
      if (!v) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((v >> (8 * k)) & 0xffu) flags[wd * 4 + k] = (uint8_t)R.flag_value;
    }
  }
}

template <bool LDSFLAGS, bool SPEC = false>
__global__ __launch_bounds__(256) void k_raycast_mark(RaycastJob R) {
  extern __shared__ unsigned s_words[];  // LDSFLAGS: ceil(ncells/4) words of 4 byte flags
  raycast_body<LDSFLAGS, SPEC>(R, blockIdx.x, s_words);
}

// Horizontal fusion: the raycast tiles and the row pass of the frame's mask job in ONE launch (independent work:
// both only read the depth image / input mask).
template <bool ERASE>
__device__ inline void decay_body(const LayerDev& L, const MapConsts& mc, uint8_t* __restrict__ kill, int* any_kill, int bid, int nb);




struct FrontLds {
  u64 s_in[kMaxMaskWords], s_d[kMaxMaskWords];
  int s_scan[10], s_carry[2];
};

// Workgroup `bid` of the frame's first launch.
template <bool LDSFLAGS>
__device__ inline void front_role(const FrontArgs& A, int bid, unsigned* s_words, FrontLds& S) {
  const RaycastJob& R = A.R;
  const MaskJob& M = A.M;
  const DecayJob& D = A.D;
  int* snap_ctr = A.snap_ctr;
  const long long tr0 = wg_trace_begin();
  // A pending Mapper.decay() touches only the TSDF layer and its lists, which neither other role reads.
  if (D.light && bid == 0) {
    // wmax is current: ONE workgroup -- the first of the grid, so that it is long done when the last ray is -- decides the
    // dead blocks from it and compacts the live list / pushes the freed slots right here, beside the raycast: the
    // allocation workgroup of the next launch starts on a clean list.
    // (8 entries per thread, not 16: the launch's register count is that of its largest role, and at 86 VGPRs the raycast
    // ran at 5 waves per SIMD instead of 7 -- the mask rows had to wait for a slot.)
    if (R.mc.dealloc_decayed)
      live_compact_body<4, 8, true, false, true>(D.L, nullptr, nullptr, S.s_scan, S.s_carry, R.mc.decay_factor, R.mc.decay_thr);
    // (thread 0 wrote ctr[0] itself.)  ctr[6] = the live count k_alloc_tsdf's pass over the EXISTING blocks runs to, while the
    // allocation workgroup of that launch appends to the list and moves ctr[0]
    if (snap_ctr && threadIdx.x == 0) snap_ctr[6] = snap_ctr[0];
    wg_trace_end(tr0, kTrFrontDecay);
    return;
  }
  const int b = bid - (D.light ? 1 : 0);
  if (snap_ctr && !D.light && b == 0 && threadIdx.x == 0) snap_ctr[6] = snap_ctr[0];
  if (b < A.n_ray_wgs) {
    raycast_body<LDSFLAGS>(R, b, s_words);
    wg_trace_end(tr0, kTrFrontRay);
  } else if (b < A.n_ray_wgs + M.H) {
    mask_rowbits_row(M, b - A.n_ray_wgs, S.s_in, S.s_d);
    wg_trace_end(tr0, kTrFrontMaskRows);
  } else {
    // wmax stale (a stand-alone call wrote TSDF weights): the full pass over the voxels.  Each workgroup also drops the
    // blocks it finds dead from the hash / dense table (in parallel, off the critical path); the order-preserving
    // compaction of the live list is the first thing the allocation workgroup of the next launch does.
    // (Tried: decay workgroups first + "last one compacts" inside this launch -- an agent-scope fence per workgroup is a
    // full L2 write-back (339 us), atomics on one arrival counter serialise (69 us), and even with a two-level counter
    // the launch grew by 9 us while the next one shrank by 2.)
    decay_body<true>(D.L, R.mc, D.kill, D.any_kill, b - A.n_ray_wgs - M.H, D.n_wgs);
    wg_trace_end(tr0, kTrFrontDecay);
  }
}

template <bool LDSFLAGS>
__global__ __launch_bounds__(256) void k_front(FrontArgs A) {
  extern __shared__ unsigned s_words[];
  __shared__ FrontLds S;
  front_role<LDSFLAGS>(A, (int)blockIdx.x, s_words, S);
}

// Large hash-indexed maps: the light decay's scalable list compaction (live_compact_big_body: decided from wmax, no voxel touched) as
// one more role of the frame's first launch -- [n_compact compaction chunks | raycast tiles | mask rows].  The compaction touches the
// TSDF layer's lists, index and summaries; the raycast and the mask rows the depth image and the view grid: nothing in common (the
// bounded workspace's single decay workgroup rides in k_front the same way).  The chunks lead the grid: their look-back chain is
// resident before the rays fill the chip, and long done when the last ray is.  APP: in a pipelined stream the previous frame's colour
// update + feature gating follows as a fourth role (as in k_front_app); n_compact == 0: a frame without a pending decay.
template <bool LDSFLAGS, bool APP>
__global__ __launch_bounds__(256) MMF_SGPR96 void k_front_compact_big(FrontArgs A, LayerDev L, u64* lb, unsigned tag, int* rebuild, float decay_f, float decay_thr,
                                                          int n_compact, AppArgs Acol, AppArgs Afeat, const float* __restrict__ synth, int Ws, int Hs,
                                                          int nb_gate) {
  extern __shared__ unsigned s_words[];
  __shared__ FrontLds S;
  __shared__ int sh[8];
  int b = (int)blockIdx.x;
  if (b < n_compact) return live_compact_big_body<true>(L, nullptr, lb, tag, rebuild, nullptr, decay_f, decay_thr, S.s_scan, sh, b, n_compact);
  b -= n_compact;
  if (!APP || b < A.n_wgs) return front_role<LDSFLAGS>(A, b, s_words, S);
  if constexpr (APP) {
    const long long tr0 = wg_trace_begin();
    app_frame_body<false, true>(Acol, Afeat, A.R.mc, synth, Ws, Hs, b - A.n_wgs, nb_gate, *reinterpret_cast<FeatLds*>(s_words));
    wg_trace_end(tr0, kTrAppFrame);
  }
}

// Two frames (two mappers fed by the same camera frame: mmf_integrate_frame_multi) in ONE launch: the workgroups of the second
// follow those of the first.  Same role code, same results; the launch is as long as its slower half instead of their sum.
template <bool LDSFLAGS>
__global__ __launch_bounds__(256) MMF_SGPR96 void k_front2(FrontArgs A0, FrontArgs A1) {
  extern __shared__ unsigned s_words[];
  __shared__ FrontLds S;
  const int b = (int)blockIdx.x;
  if (b < A0.n_wgs)
    front_role<LDSFLAGS>(A0, b, s_words, S);
  else
    front_role<LDSFLAGS>(A1, b - A0.n_wgs, s_words, S);
}

// Deferred mode: the colour update + feature gating of the PREVIOUS frame as one more role of this frame's first launch.  It
// reads the previous frame's candidate lists, synthetic depth, masks and images and writes the appearance layers; the raycast,
// the mask rows and the decay touch the depth image, the view grid and the TSDF layer's lists -- nothing in common.  The
// gating workgroups follow the frame's own (the longest rays decide when the launch ends; in front of the mask rows they cost
// 1.7 us more); their LDS is the launch's dynamic block.
template <bool LDSFLAGS>
__global__ __launch_bounds__(256) MMF_SGPR96 void k_front_app(FrontArgs A, AppArgs Acol, AppArgs Afeat, const float* __restrict__ synth, int Ws, int Hs,
                                                  int nb_gate) {
  extern __shared__ unsigned s_words[];
  __shared__ FrontLds S;
  if ((int)blockIdx.x < A.n_wgs) return front_role<LDSFLAGS>(A, (int)blockIdx.x, s_words, S);
  const long long tr0 = wg_trace_begin();
  app_frame_body<false, true>(Acol, Afeat, A.R.mc, synth, Ws, Hs, (int)blockIdx.x - A.n_wgs, nb_gate, *reinterpret_cast<FeatLds*>(s_words));
  wg_trace_end(tr0, kTrAppFrame);
}

// N frames (independent mappers: mmf_integrate_frame_batch): frame q's workgroups follow frame q-1's.
template <bool LDSFLAGS>
__global__ __launch_bounds__(256, 7) void k_front_batch(FrontBatch P) {  // 72 VGPRs (natural: 83 = 6 waves per SIMD; 8 waves spill too much): 86 -> 80 us at N = 8
  extern __shared__ unsigned s_words[];
  __shared__ FrontLds S;
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    const int nw = P.a[q].n_wgs;
    if (b < nw) return front_role<LDSFLAGS>(P.a[q], b, s_words, S);
    b -= nw;
  }
}

// ... | the pending colour update + feature gating of the PREVIOUS frame of every mapper of the batch that has one (deferred mode)
template <bool LDSFLAGS>
__global__ __launch_bounds__(256, 7) void k_front_batch_app(FrontBatch P, AppFrameBatch G) {
  extern __shared__ unsigned s_words[];
  __shared__ FrontLds S;
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    const int nw = P.a[q].n_wgs;
    if (b < nw) return front_role<LDSFLAGS>(P.a[q], b, s_words, S);
    b -= nw;
  }
  for (int q = 0; q < G.n; ++q) {
    const AppFrameArgs& F = G.a[q];
    if (b < F.nb) return app_frame_body<false, true>(F.Ac, F.Af, F.mc, F.synth, F.Ws, F.Hs, b, F.nb, *reinterpret_cast<FeatLds*>(s_words));
    b -= F.nb;
  }
}

// ------------------------------------------------------------------------------------------------
// 2. Flag compaction + hash lookup / insertion (shared by TSDF, colour and feature allocation).
//    count tiles -> scan tiles -> emit.  A tile is 1024 cells (256 threads x 4 flag bytes).
//    Candidate order = cell order (lexicographic block index for the view grid), so the order of
//    the live list and of every output derived from it is deterministic.  New blocks take their
//    pool slot from (rank among the new ones): wave ballot/prefix-sum, no per-block atomics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_tiles(LayerDev L, KeySrc ks, Scratch sc, int ncells) {
  __shared__ int lds[10];
  if (ks.mode == 1) {
    const int nl = *ks.n_live;
    ncells = ncells < nl ? ncells : nl;
  }
  const int cell0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  uint32_t f4 = 0;
  if (cell0 < ncells) f4 = *reinterpret_cast<const uint32_t*>(sc.flags + cell0);
  int nf = 0, nn = 0;
  if (f4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (((f4 >> (8 * k)) & 0xffu) && cell0 + k < ncells) {
        int cell = cell0 + k;
        int slot = layer_lookup(L, cell_key(ks, sc, cell));
        sc.cell_slot[cell] = slot;
        nf++;
        nn += slot < 0;
      }
    }
  }
  int ea, eb, ta, tb;
  block_excl_scan2<4>(nf, nn, lds, ea, eb, ta, tb);
  if (threadIdx.x == 0) sc.tile_counts[blockIdx.x] = make_int2(ta, tb);
}

// Single workgroup: exclusive scan of the tile counts, slot grant, counter update, statistics.
__global__ __launch_bounds__(256) void k_scan_tiles(LayerDev L, Scratch sc, int ntiles, long long* stats, int stat_upd,
                                                   int stat_new) {
  __shared__ int lds[10];
  __shared__ int carry[2];
  if (threadIdx.x == 0) {
    carry[0] = 0;
    carry[1] = 0;
  }
  __syncthreads();
  for (int base = 0; base < ntiles; base += 256) {
    int t = base + threadIdx.x;
    int2 c = t < ntiles ? sc.tile_counts[t] : make_int2(0, 0);
    int ea, eb, ta, tb;
    block_excl_scan2<4>(c.x, c.y, lds, ea, eb, ta, tb);
    if (t < ntiles) sc.tile_offs[t] = make_int2(carry[0] + ea, carry[1] + eb);
    __syncthreads();
    if (threadIdx.x == 0) {
      carry[0] += ta;
      carry[1] += tb;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int n_cand = carry[0], n_new = carry[1];
    int n_live = L.ctr[0], n_free = L.ctr[1], bump = L.ctr[2];
    int room = n_free + (L.cap - bump);
    int granted = n_new < room ? n_new : room;
    if (granted < n_new) atomicOr(&L.ctr[3], 1);
    sc.alloc_ctx[0] = n_live;
    sc.alloc_ctx[1] = n_free;
    sc.alloc_ctx[2] = bump;
    sc.alloc_ctx[3] = granted;
    int from_free = granted < n_free ? granted : n_free;
    L.ctr[0] = n_live + granted;
    L.ctr[1] = n_free - from_free;
    L.ctr[2] = bump + (granted - from_free);
    *sc.cand_count = n_cand;
    if (sc.hint_cand) *sc.hint_cand = n_cand;
    if (L.hint_live) *L.hint_live = L.ctr[0];
    if (stats) {
      if (stat_upd >= 0) stats[stat_upd] += n_cand;
      if (stat_new >= 0) stats[stat_new] += granted;
    }
  }
}

__global__ __launch_bounds__(256) void k_emit(LayerDev L, KeySrc ks, Scratch sc, int ncells) {
  __shared__ int lds[10];
  if (ks.mode == 1) {
    const int nl = *ks.n_live;
    ncells = ncells < nl ? ncells : nl;
  }
  const int cell0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  uint32_t f4 = 0;
  if (cell0 < ncells) f4 = *reinterpret_cast<const uint32_t*>(sc.flags + cell0);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (cell0 + k >= ncells) f4 &= ~(0xffu << (8 * k));  // stale flags beyond the live count
  int slot4[4];
  int nf = 0, nn = 0;
  if (f4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      slot4[k] = 0;
      if ((f4 >> (8 * k)) & 0xffu) {
        slot4[k] = sc.cell_slot[cell0 + k];
        nf++;
        nn += slot4[k] < 0;
      }
    }
  }
  int ea, eb, ta, tb;
  block_excl_scan2<4>(nf, nn, lds, ea, eb, ta, tb);
  if (!f4) return;
  const int2 off = sc.tile_offs[blockIdx.x];
  const int old_live = sc.alloc_ctx[0], old_free = sc.alloc_ctx[1], old_bump = sc.alloc_ctx[2], granted = sc.alloc_ctx[3];
  int pos = off.x + ea, rnk = off.y + eb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!((f4 >> (8 * k)) & 0xffu)) continue;
    const int cell = cell0 + k;
    const u64 key = cell_key(ks, sc, cell);
    int slot = slot4[k];
    const bool is_new = slot < 0;
    if (is_new) {
      if (rnk < granted) {
        slot = rnk < old_free ? L.free_stack[old_free - 1 - rnk] : old_bump + (rnk - old_free);
        if (hash_insert(L, key, slot)) atomicSub(&L.ctr[4], 1);  // a tombstone became a key again
        dense_set(L, key, slot + 1);
        L.slot_key[slot] = key;
        L.live[old_live + rnk] = slot;
      }
      rnk++;
    }
    sc.cand_slot[pos] = slot;
    sc.cand_key[pos] = key;
    sc.cand_new[pos] = is_new ? 1 : 0;
    pos++;
  }
  if (ks.mode == 0) *reinterpret_cast<uint32_t*>(sc.flags + cell0) = 0u;  // grid flags: all-zero for the next frame
}


// alloc_job_body (count + scan + emit of a small cell set in one 1024-thread workgroup): mmf_alloc_device.h
template <bool DENSE>
__global__ __launch_bounds__(1024) void k_alloc_fused(AllocJob J, long long* stats) {
  __shared__ int lds[34];
  __shared__ int carry[2];
  __shared__ int ctx[4];
  alloc_job_body<DENSE>(J, stats, lds, carry, ctx);
}

// Horizontal fusion: up to two allocation jobs (one workgroup each) and the column pass of the frame's mask job
// (one workgroup per output row) in ONE launch.  The roles are independent; this only removes launch boundaries.
// TWO = false: a single job (J1 is ignored: half the kernel arguments to keep in scalar registers).
template <bool DENSE, int MODE, bool TWO>
__global__ __launch_bounds__(1024) void k_alloc_jobs(AllocJob J0, AllocJob J1, int njobs, long long* stats, MaskJob M,
                                                    int mask_rows) {
  __shared__ int lds[34];
  __shared__ int carry[2];
  __shared__ int ctx[4];
  __shared__ u64 s_bad[kMaxMaskWords];
  const long long tr0 = wg_trace_begin();
  if ((int)blockIdx.x < njobs) {
    alloc_job_body<DENSE, MODE>((!TWO || blockIdx.x == 0) ? J0 : J1, stats, lds, carry, ctx);
    wg_trace_end(tr0, kTrAllocJob);
  } else if ((int)blockIdx.x - njobs < mask_rows) {
    const long long t0 = J0.timeline ? wall_clock64() : 0;
    mask_colemit_row(M, (int)blockIdx.x - njobs, s_bad);
    wg_trace_end(tr0, kTrAllocMaskCols);
    if (J0.timeline && threadIdx.x == 0) {  // diagnostics: earliest start / latest end over the mask workgroups
      const long long t1 = wall_clock64();
      atomicMin(reinterpret_cast<unsigned long long*>(J0.timeline + 7), (unsigned long long)t0);
      atomicMax(reinterpret_cast<unsigned long long*>(J0.timeline + 6), (unsigned long long)t1);
      atomicMax(reinterpret_cast<unsigned long long*>(J0.timeline + 8), (unsigned long long)(t1 - t0));   // longest one
      atomicMax(reinterpret_cast<unsigned long long*>(J0.timeline + 9), (unsigned long long)t0);          // latest start
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3. Projective TSDF update: one workgroup (512 threads = 8 waves) per 8x8x8 block, thread = voxel,
//    z fastest so a wave reads/writes 512 contiguous bytes of {distance, weight}.  HBM-bound
//    read-modify-write of 4 KB per block; the 4 depth taps come through L1/L2 (1.2 MB image).
// ------------------------------------------------------------------------------------------------
__device__ inline bool depth_ok(float d, float min_d) { return d > min_d; }

typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load with dword alignment

// Measured surface depth at image-plane point (u,v) (spec: oracle/mmf_oracle.c:sample_depth).
// `depth` is either the raw image with `mask` (valid iff depth > min_d and mask != 0) or, with mask == nullptr,
// an image whose invalid pixels are already <= min_d.  The 2x2 bilinear footprint is fetched as two 8-byte
// row-pair loads; the nearest tap (pixel floor(u), floor(v)) is one of the four whenever the footprint is inside
// the image (u - 0.5 is exact in float32, so floor(u) = x0 + (wx >= 0.5)), which turns 5 scattered depth reads
// (+5 mask reads) per voxel into 2.
template <int FMA = 0>
__device__ inline bool sample_depth(const MapConsts& mc, const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                                    float min_d, const Cam& cam, float u, float v, float& out) {
  int x0, y0;
  float wx, wy;
  if (bilin_setup(u, v, cam.W, cam.H, x0, y0, wx, wy)) {
    const size_t i0 = (size_t)y0 * cam.W + x0;
    const float2_u r0 = *reinterpret_cast<const float2_u*>(depth + i0);
    const float2_u r1 = *reinterpret_cast<const float2_u*>(depth + i0 + cam.W);
    bool v00 = depth_ok(r0.x, min_d), v10 = depth_ok(r0.y, min_d), v01 = depth_ok(r1.x, min_d), v11 = depth_ok(r1.y, min_d);
    if (mask) {
      v00 = v00 && mask[i0];
      v10 = v10 && mask[i0 + 1];
      v01 = v01 && mask[i0 + cam.W];
      v11 = v11 && mask[i0 + cam.W + 1];
    }
    const bool nx = wx >= 0.5f, ny = wy >= 0.5f;
    const float dn = ny ? (nx ? r1.y : r1.x) : (nx ? r0.y : r0.x);
    const bool vn = ny ? (nx ? v11 : v01) : (nx ? v10 : v00);
    if (!vn) return false;
    if (v00 && v10 && v01 && v11) {
      bool ok = true;
      if (mc.lin_md > 0.0f) {
        if (fabsf(r0.x - dn) > mc.lin_md || fabsf(r0.y - dn) > mc.lin_md || fabsf(r1.x - dn) > mc.lin_md ||
            fabsf(r1.y - dn) > mc.lin_md)
          ok = false;
      }
      if (ok) {
        out = bilin<FMA>(r0.x, r0.y, r1.x, r1.y, wx, wy);
        return true;
      }
    }
    out = dn;
    return true;
  }
  // footprint leaves the image: nearest tap only
  int xn = ifloor(u), yn = ifloor(v);
  if (xn > cam.W - 1) xn = cam.W - 1;
  if (yn > cam.H - 1) yn = cam.H - 1;
  const size_t i = (size_t)yn * cam.W + xn;
  const float dn = depth[i];
  if (!depth_ok(dn, min_d)) return false;
  if (mask && !mask[i]) return false;
  out = dn;
  return true;
}

// ---- THE projective TSDF update of one voxel (spec: oracle/mmf_oracle.c, "TSDF update"), in its branching form: k_tsdf_integrate and the
// MASKED passes (stand-alone add_depth_frame with a mask) come here; the unmasked passes of the fused frame and of the hash path run the
// same rules branch-free (tsdf_voxel_group_bf below -- the second and last copy, bit-identical to this one over every suite).
//   in_view <- the voxel centre projects into the image, not beyond the maximum integration distance (evaluated when `want_view`);
//   if `cand` (the block is integrated this frame) and in view and the depth sample is valid and sdf >= -trunc and w > 0:
//     D <- clamp((sdf w + D W) / (w + W), +-trunc),  W <- min(W + w, max_weight);  returns true iff D / W changed.
template <bool MASKED, int FMA = 0>
__device__ inline bool tsdf_voxel_update(const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* __restrict__ depth,
                                         const uint8_t* __restrict__ mask, float min_d, int bx, int by, int bz, int lin, bool cand,
                                         bool want_view, float& D, float& W, bool& in_view) {
  float c[3], p[3], u, v;
  in_view = false;
  if (!want_view) return false;
  voxel_centre<FMA>(mc, bx, by, bz, lin, c);
  xform<FMA>(T_C_L, c, p);
  in_view = project<FMA>(cam, p, u, v) && !(mc.max_dist > 0.0f && p[2] > mc.max_dist);
  if (!(cand && in_view)) return false;
  float d;
  if (!sample_depth<FMA>(mc, depth, MASKED ? mask : nullptr, min_d, cam, u, v, d)) return false;
  const float sdf = d - p[2];
  if (sdf < -mc.trunc) return false;
  const float wm = tsdf_measurement_weight(mc, d, sdf);
  if (!(wm > 0.0f)) return false;
  float Dn = madd2<FMA>(sdf, wm, D, W) / (wm + W);
  Dn = Dn > 0.0f ? fminf(mc.trunc, Dn) : fmaxf(-mc.trunc, Dn);
  D = Dn;
  W = fminf(W + wm, mc.max_weight);
  return true;
}

// What a pass learns about a block while it holds its voxels: an appearance-candidate voxel in view (hit), every voxel observed
// free space (freev: the sphere tracer's empty-space summary), the largest weight (wmx: the lazy decay's deallocation test).
struct TsdfBlockAcc {
  int hit = 0, freev = 1, band = 0;
  float wmx = 0.0f, wmn = 3.0e38f;
};

// VPT z-adjacent voxels of one thread (VPT / 2 float4 = {D, W, D, W}): load (or zeros for a new block), the pending decay's
// W *= f, the update above per voxel, write back when anything changed.  The only copy of the per-thread voxel loop.
// `lag` (block-uniform): how many decays the stored weights are behind -- applied one multiplication at a time, exactly as the
// eager decays would have been (1 for the pending decay of a bounded map, cur_epoch - epoch[slot] in lazy mode, 0: none).
// `may_write` false: a read-only visit (lazy mode, a block this frame does not integrate: only its appearance flag is wanted).
// LAGLOOP false: lag is 0 or 1 (the hot bounded kernels: one predicated multiplication, no loop).
template <int VPT, bool MASKED, bool LAGLOOP = false, int FMA = 0>
__device__ inline void tsdf_voxel_group(const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* __restrict__ depth,
                                        const uint8_t* __restrict__ mask, float min_d, float decay_f, int lag, bool may_write, int bx, int by,
                                        int bz, int lin0, bool cand, bool is_new, float4* __restrict__ vox, TsdfBlockAcc& acc,
                                        long long* ph = nullptr) {
  constexpr int NP = VPT / 2;
  float4 av[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) av[q] = is_new ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : vox[q];
#ifdef MMF_WG_TRACE
  if (ph) {  // phase marks of tools/wg_trace.py --phases (instrumented build only): the voxels have arrived
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ph[1] = (long long)wall_clock64();
  }
#endif
  const bool decayed = lag > 0;  // uniform
  if (LAGLOOP) {
    for (int l = 0; l < lag; ++l) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        av[q].y = av[q].y * decay_f;
        av[q].w = av[q].w * decay_f;
      }
    }
  } else if (decayed) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      av[q].y = av[q].y * decay_f;
      av[q].w = av[q].w * decay_f;
    }
  }
  bool upd = false;
#pragma unroll
  for (int r = 0; r < VPT; ++r) {
    float4& a = av[r >> 1];
    const bool hi = (r & 1) != 0;
    float D = hi ? a.z : a.x, W = hi ? a.w : a.y;
    // the projection is needed by the update (cand) and by the appearance flag of a near-surface voxel only
    bool in_view;
    upd |= tsdf_voxel_update<MASKED, FMA>(mc, cam, T_C_L, depth, mask, min_d, bx, by, bz, lin0 + r, cand, cand || (W > 0.0f && fabsf(D) < mc.trunc), D,
                                     W, in_view);
    if (hi) {
      a.z = D;
      a.w = W;
    } else {
      a.x = D;
      a.y = W;
    }
    const bool near = W > 0.0f && fabsf(D) < mc.trunc;
    acc.hit |= (near && in_view) ? 1 : 0;
    acc.band |= near ? 1 : 0;
    acc.freev &= (W > 1e-4f && D == mc.trunc) ? 1 : 0;
    acc.wmx = fmaxf(acc.wmx, W);
    acc.wmn = fminf(acc.wmn, W);
  }
#ifdef MMF_WG_TRACE
  if (ph) ph[2] = (long long)wall_clock64();  // the voxel loop is done
#endif
  if (may_write && (upd || is_new || decayed)) {
#pragma unroll
    for (int q = 0; q < NP; ++q) vox[q] = av[q];
  }
#ifdef MMF_WG_TRACE
  if (ph) {  // the stores have been acknowledged
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ph[3] = (long long)wall_clock64();
  }
#endif
}

// ---- the same voxel loop BRANCH-FREE (unmasked depth image only: the fused frame's / the hash path's) -----------------------------
// tsdf_voxel_update is a cascade of early-outs: per voxel ~60 scalar instructions of exec-mask management and ~15 branches around
// ~144 vector instructions that nearly every wave executes anyway (some lane needs them).  Here every voxel of a block that is
// integrated runs the whole update and the result is SELECTED: the 2 x 2 footprint is clamped into the image (it always contains the
// nearest tap and IS the bilinear footprint whenever that exists), voxels that are not updated compute on garbage that is discarded
// (no trap on this target).  The same float operations on every voxel that is updated: bit-identical to tsdf_voxel_group over the parity,
// fuzz, soak and hash suites.  Measured (round 5, profiles/r05p_branch_free.txt): k_alloc_tsdf 14.6 -> 14.3 us, the lazy pass of the hash
// path 116 -> 112 us.  tsdf_voxel_group stays for the masked stand-alone calls (its mask taps are worth skipping).
template <int VPT, bool LAGLOOP, int FMA>
__device__ inline void tsdf_voxel_group_bf(const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* __restrict__ depth, float min_d,
                                           float decay_f, int lag, bool may_write, int bx, int by, int bz, int lin0, bool cand, bool is_new,
                                           float4* __restrict__ vox, TsdfBlockAcc& acc, long long* ph = nullptr) {
  constexpr int NP = VPT / 2;
  float4 av[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) av[q] = is_new ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : vox[q];
#ifdef MMF_WG_TRACE
  if (ph) {  // phase marks of tools/wg_trace.py --phases (instrumented build only): the voxels have arrived
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ph[1] = (long long)wall_clock64();
  }
#endif
  const bool decayed = lag > 0;  // uniform
  if (LAGLOOP) {
    for (int l = 0; l < lag; ++l) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        av[q].y = av[q].y * decay_f;
        av[q].w = av[q].w * decay_f;
      }
    }
  } else if (decayed) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      av[q].y = av[q].y * decay_f;
      av[q].w = av[q].w * decay_f;
    }
  }
  bool upd_any = false;
#pragma unroll
  for (int r = 0; r < VPT; ++r) {
    float4& a = av[r >> 1];
    const bool hi = (r & 1) != 0;
    float D = hi ? a.z : a.x, W = hi ? a.w : a.y;
    const bool want = cand || (W > 0.0f && fabsf(D) < mc.trunc);
    float c[3], p[3];
    voxel_centre<FMA>(mc, bx, by, bz, lin0 + r, c);
    xform<FMA>(T_C_L, c, p);
    const float iz = 1.0f / p[2];
    const float uu = madd<FMA>(cam.fx, p[0] * iz, cam.cx);
    const float vv = madd<FMA>(cam.fy, p[1] * iz, cam.cy);
    const bool proj = !(p[2] <= 1e-6f) && !(uu < 0.0f || vv < 0.0f || uu > (float)cam.W || vv > (float)cam.H);
    const bool inview = want && proj && !(mc.max_dist > 0.0f && p[2] > mc.max_dist);
    if (cand) {  // uniform over the block
      // This form converts for EVERY voxel of a candidate block, also those that do not project (p[2] ~ 0: uu = inf / NaN / beyond
      // int's range, whose conversion is undefined): clamp in float first.  A projecting voxel has uu in [0, W], vv in [0, H] -- inside
      // the clamps, so its values are unchanged; the others get defined indices and are discarded by `inview` as before.
      const float uc = fminf(fmaxf(uu - 0.5f, -1.0f), (float)cam.W), vc = fminf(fmaxf(vv - 0.5f, -1.0f), (float)cam.H);
      const float fx0 = floorf(uc), fy0 = floorf(vc);
      const int ix = (int)fx0, iy = (int)fy0;
      const bool fp_ok = !(ix < 0 || iy < 0 || ix > cam.W - 2 || iy > cam.H - 2);
      const float wx = uc - fx0, wy = vc - fy0;
      const int x0c = ix < 0 ? 0 : (ix > cam.W - 2 ? cam.W - 2 : ix), y0c = iy < 0 ? 0 : (iy > cam.H - 2 ? cam.H - 2 : iy);
      int xn = ifloor(fminf(fmaxf(uu, 0.0f), (float)cam.W)), yn = ifloor(fminf(fmaxf(vv, 0.0f), (float)cam.H));
      xn = xn > cam.W - 1 ? cam.W - 1 : (xn < 0 ? 0 : xn);
      yn = yn > cam.H - 1 ? cam.H - 1 : (yn < 0 ? 0 : yn);
      const int sel = ((yn - y0c) << 1) | (xn - x0c);  // which of the four taps is the nearest one (always among them for a projecting voxel)
      const unsigned i0 = (unsigned)y0c * (unsigned)cam.W + (unsigned)x0c;
      const float2_u q0 = *reinterpret_cast<const float2_u*>(depth + i0);
      const float2_u q1 = *reinterpret_cast<const float2_u*>(depth + i0 + cam.W);
      const float a00 = q0.x, a10 = q0.y, a01 = q1.x, a11 = q1.y;
      const float dn = (sel & 2) ? ((sel & 1) ? a11 : a01) : ((sel & 1) ? a10 : a00);
      const bool vn = depth_ok(dn, min_d);
      bool lin = fp_ok && depth_ok(a00, min_d) && depth_ok(a10, min_d) && depth_ok(a01, min_d) && depth_ok(a11, min_d);
      if (mc.lin_md > 0.0f)  // uniform
        lin = lin && !(fabsf(a00 - dn) > mc.lin_md || fabsf(a10 - dn) > mc.lin_md || fabsf(a01 - dn) > mc.lin_md || fabsf(a11 - dn) > mc.lin_md);
      const float d = lin ? bilin<FMA>(a00, a10, a01, a11, wx, wy) : dn;
      const float sdf = d - p[2];
      const float wm = tsdf_measurement_weight(mc, d, sdf);
      float Dn = madd2<FMA>(sdf, wm, D, W) / (wm + W);
      Dn = Dn > 0.0f ? fminf(mc.trunc, Dn) : fmaxf(-mc.trunc, Dn);
      const bool upd = inview && vn && !(sdf < -mc.trunc) && (wm > 0.0f);
      D = upd ? Dn : D;
      W = upd ? fminf(W + wm, mc.max_weight) : W;
      upd_any |= upd;
    }
    if (hi) {
      a.z = D;
      a.w = W;
    } else {
      a.x = D;
      a.y = W;
    }
    const bool near = W > 0.0f && fabsf(D) < mc.trunc;
    acc.hit |= (near && inview) ? 1 : 0;
    acc.band |= near ? 1 : 0;
    acc.freev &= (W > 1e-4f && D == mc.trunc) ? 1 : 0;
    acc.wmx = fmaxf(acc.wmx, W);
    acc.wmn = fminf(acc.wmn, W);
  }
#ifdef MMF_WG_TRACE
  if (ph) ph[2] = (long long)wall_clock64();  // the voxel loop is done
#endif
  if (may_write && (upd_any || is_new || decayed)) {
#pragma unroll
    for (int q = 0; q < NP; ++q) vox[q] = av[q];
  }
#ifdef MMF_WG_TRACE
  if (ph) {  // the stores have been acknowledged
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ph[3] = (long long)wall_clock64();
  }
#endif
}

template <int FMA>
__global__ __launch_bounds__(512) void k_tsdf_integrate(LayerDev L, MapConsts mc, Cam cam, Rigid T_C_L,
                                                       const float* __restrict__ depth,
                                                       const uint8_t* __restrict__ mask, float min_d, Scratch sc) {
  const int n = *sc.cand_count;
  const int chunk = (n + 7) >> 3;
  const int lin = threadIdx.x;
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int i = xcd_candidate(j, chunk);
    if (i >= n) continue;
    const int slot = sc.cand_slot[i];
    if (slot < 0) continue;
    const bool is_new = sc.cand_new[i] != 0;
    int bx, by, bz;
    unpack_key(sc.cand_key[i], bx, by, bz);
    float2* vox = reinterpret_cast<float2*>(L.pool) + (size_t)slot * kVPB + lin;
    float2 dw = is_new ? make_float2(0.0f, 0.0f) : *vox;
    bool in_view;
    const bool upd = tsdf_voxel_update<true, FMA>(mc, cam, T_C_L, depth, mask, min_d, bx, by, bz, lin, true, true, dw.x, dw.y, in_view);
    if (upd || is_new) *vox = dw;
    // block summary for the sphere tracer's empty-space skipping
    const int all_free = __syncthreads_and((dw.y > 1e-4f && dw.x == mc.trunc) ? 1 : 0);
    if (lin == 0) L.block_free[slot] = all_free ? 1 : 0;
  }
}

// Fused-frame form of the TSDF update: ONE pass over the live list that (a) integrates the blocks stamped by this
// frame's allocation job exactly as k_tsdf_integrate does and (b) evaluates, on the voxels it already holds, the
// appearance-candidate test of k_app_candidates (mmf_kernels_app.hip) for every live block -- the colour / feature camera
// of a fused frame is the depth camera, so the projection is shared.  Saves a launch and a second read of the layer.
// VPT voxels per thread (z-adjacent, 16-byte accesses), 512 / VPT threads per block.  VPT = 4: two waves per block, so every
// live block of a bounded workspace is resident at once (at four waves per block 1 792 of ~2 300 fitted and the rest formed a
// second round).
// MASKED = false: the depth image is already masked (the fused frame's masked depth): the mask taps are compiled out.
// LAZY (large hash-indexed maps): the pass walks the WORK LIST k_tsdf_classify made of the live list --
// the blocks this frame integrates plus the near-surface blocks whose appearance flag needs their voxels -- instead of every live
// block; a block's missing decays (cur_epoch - epoch[slot]) are applied before it is integrated, a block that is only looked at
// is not written.  Not LAZY on a layer with lazy summaries (L.epoch != nullptr): the full pass that (re)establishes them.
template <int VPT, bool MASKED, bool LAZY, int FMA>
__global__ __launch_bounds__(512 / VPT) MMF_SGPR96 void k_tsdf_pass(LayerDev L, MapConsts mc, Cam cam, Rigid T_C_L,
                                                        const float* __restrict__ depth, const uint8_t* __restrict__ mask_arg,
                                                        float min_d, int stamp, uint8_t* __restrict__ flags,
                                                        u64* __restrict__ cell_key, float decay_f, const int* __restrict__ work_n,
                                                        const int* __restrict__ work) {
  // decay_f > 0: a Mapper.decay() is pending whose deallocations k_front already made from L.wmax -- its W *= f is applied
  // here, on the voxels this pass loads anyway (and every block is written back).  L.wmax is refreshed for every live block.
  static_assert(VPT == 2 || VPT == 4 || VPT == 8, "one, two or four 16-byte voxel pairs per thread");
  // (two sets, alternating: thread 0 reads an iteration's set after the barrier while the other waves may already fill the next one)
  __shared__ float s_wmax2[2][512 / VPT / 64], s_wmin2[2][512 / VPT / 64];
  __shared__ int s_band2[2][512 / VPT / 64];
  int par = 0;
  constexpr int NP = VPT / 2;  // float4 = two {distance, weight} voxels
  const uint8_t* __restrict__ mask = MASKED ? mask_arg : nullptr;
  const long long tr0 = wg_trace_begin();
  // (uniform constants of the voxel loop in vector registers: mmf_device.h vgpr())
  rigid_to_vgprs(T_C_L);
  cam_to_vgprs(cam);
  if constexpr (MASKED) update_consts_to_vgprs(mc);  // (the branch-free voxel group keeps them scalar: 68 VGPRs = seven waves per SIMD; 74 with them)
  const int n = LAZY ? *work_n : L.ctr[0];
  const int chunk = (n + 7) >> 3;
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int iw = xcd_candidate(j, chunk);
    if (iw >= n) continue;
    par ^= 1;
    float* s_wmax = s_wmax2[par];
    float* s_wmin = s_wmin2[par];
    int* s_band = s_band2[par];
    const int i = LAZY ? work[iw] : iw;
    const int slot = L.live[i];
    const u64 key = L.slot_key[slot];
    const int st = L.stamp[slot];
    const bool cand = (st >> 1) == stamp, is_new = cand && (st & 1);
    int bx, by, bz;
    unpack_key(key, bx, by, bz);
    float4* vox = reinterpret_cast<float4*>(L.pool) + (size_t)slot * (kVPB / 2) + threadIdx.x * NP;
    const int lag = LAZY ? (is_new ? 0 : L.cur_epoch - L.epoch[slot]) : (decay_f > 0.0f ? 1 : 0);  // uniform
    const float f = LAZY ? L.lag_f : decay_f;
    const bool writes = !LAZY || cand;
    TsdfBlockAcc acc;
    if constexpr (!MASKED)
      tsdf_voxel_group_bf<VPT, LAZY, FMA>(mc, cam, T_C_L, depth, min_d, f, lag, writes, bx, by, bz, threadIdx.x * VPT, cand, is_new, vox, acc);
    else
      tsdf_voxel_group<VPT, MASKED, LAZY, FMA>(mc, cam, T_C_L, depth, mask, min_d, f, lag, writes, bx, by, bz, threadIdx.x * VPT, cand, is_new, vox, acc);
    const int hit = acc.hit, freev = acc.freev;
    float wmx = acc.wmx, wmn = acc.wmn;
    if (writes && (cand || lag > 0)) {  // workgroup-uniform: the block's voxels changed -> its empty-space summary may have
      const int all_free = __syncthreads_and(freev);
      if (threadIdx.x == 0) L.block_free[slot] = all_free ? 1 : 0;
    }
    wmx = wave_max_f32(wmx);
    wmn = -wave_max_f32(-wmn);
    const int w_band = __any(acc.band) ? 1 : 0;
    if ((threadIdx.x & 63) == 0) {
      s_wmax[threadIdx.x >> 6] = wmx;
      s_wmin[threadIdx.x >> 6] = wmn;
      s_band[threadIdx.x >> 6] = w_band;
    }
    const int any = __syncthreads_or(hit);  // (also orders s_wmax / s_wmin / s_band)
    if (threadIdx.x == 0) {
      int any2 = 0;
#pragma unroll
      for (int q = 0; q < 512 / VPT / 64; ++q) any2 |= s_band[q];
      any2 <<= 1;
      flags[i] = any ? 1 : 0;
      if (any) cell_key[i] = key;
      if (writes) {
        float m = s_wmax[0], mn = s_wmin[0];
#pragma unroll
        for (int q = 1; q < 512 / VPT / 64; ++q) {
          m = fmaxf(m, s_wmax[q]);
          mn = fminf(mn, s_wmin[q]);
        }
        L.wmax[slot] = m;
        if (L.epoch) {  // the lazy summaries of a large map
          L.wmin[slot] = mn;
          L.band[slot] = (unsigned char)(any2 >> 1);
          L.epoch[slot] = L.cur_epoch;
        }
      }
    }
  }
  wg_trace_end(tr0, kTrTsdfPass);
}

// The work list of a lazy pass: one thread per live block.  A block the frame integrates (its allocation job stamped it), or one
// with near-surface voxels (band: its appearance flag depends on which of them are in view), goes on the list; any other block's
// flag is 0 and its voxels are not touched.  *count = the list's length, list[..] = live-list positions, in no particular order
// (blocks are independent).  Two counters alternate between frames: this frame's was zeroed by the previous frame's classify
// (*zero_next = 0 here, for the next one) -- no memset node per frame.  1 024 threads per workgroup: one same-address atomic per
// workgroup, and those atomics (serialised at the L2) are most of this kernel's time.
__global__ __launch_bounds__(1024) void k_tsdf_classify(LayerDev L, int stamp, uint8_t* __restrict__ flags, int* __restrict__ count,
                                                       int* __restrict__ zero_next, int* __restrict__ list) {
  __shared__ int s_cnt[16], s_base;
  const int n = L.ctr[0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) *zero_next = 0;
  for (int i0 = (int)blockIdx.x * 1024; i0 < n; i0 += (int)gridDim.x * 1024) {  // workgroup-uniform trip count
    const int i = i0 + (int)threadIdx.x;
    bool take = false;
    if (i < n) {
      const int slot = L.live[i];
      take = (L.stamp[slot] >> 1) == stamp || L.band[slot] != 0;
      if (!take) flags[i] = 0;
    }
    const unsigned long long b = __ballot(take);
    if (lane == 0) s_cnt[wv] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) tot += s_cnt[q];
      s_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    int base = s_base;
    for (int q = 0; q < wv; ++q) base += s_cnt[q];
    if (take) list[base + __popcll(b & ((1ull << lane) - 1ull))] = i;
    __syncthreads();  // (s_cnt / s_base are rewritten by the next round)
  }
}

// Every live block's voxels brought up to the current decay epoch (before anything reads voxel weights of a lazily decayed map,
// or writes them outside the lazy pass): the missing multiplications, one by one.
__global__ __launch_bounds__(128) void k_lazy_catchup(LayerDev L) {
  const int n = L.ctr[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = L.live[i];
    const int lag = L.cur_epoch - L.epoch[slot];  // uniform
    if (lag <= 0) continue;
    float4* vox = reinterpret_cast<float4*>(L.pool) + (size_t)slot * (kVPB / 2) + threadIdx.x * 2;
    float4 a = vox[0], b = vox[1];
    for (int l = 0; l < lag; ++l) {
      a.y = a.y * L.lag_f, a.w = a.w * L.lag_f;
      b.y = b.y * L.lag_f, b.w = b.w * L.lag_f;
    }
    vox[0] = a, vox[1] = b;
    __syncthreads();
    if (threadIdx.x == 0) L.epoch[slot] = L.cur_epoch;
  }
}

// ------------------------------------------------------------------------------------------------
// 3c. Allocation | mask columns | TSDF pass in ONE launch (bounded workspace, fused frame).
//     The TSDF pass does not need the allocation's result for the blocks that already exist: whether such a block is
//     integrated this frame is the raycast flag of its view-grid cell, and its position in the live list is final once
//     k_front has compacted the list.  So the existing blocks (all of them in steady state but a few dozen) are
//     processed BESIDE the allocation workgroup, whose ~8 us chain used to stand alone on the critical path with a launch
//     boundary on either side.  The new blocks of the frame are handled by a few workgroups at the end of the grid that
//     wait for the allocation workgroup of the same launch: it publishes {slot, key} of every new block and their number
//     as self-validating tagged words with relaxed agent-scope atomics (alloc_grid_multi_body; the scheme of the grouped
//     FPS kernel: no fences, bounded polling).  The allocation workgroup is block 0, so it is resident before any waiter.
//     256-thread workgroups; a TSDF workgroup holds TWO blocks (one per half, 4 voxels per thread as in k_tsdf_pass).
// ------------------------------------------------------------------------------------------------

struct TsdfPairLds {
  int free_[2][4], hit[2][4];
  float wmax[2][4];
  int slot[2], act[2];
  unsigned klo[2], khi[2];
};

// One block by one half (128 threads) of the workgroup; `par` alternates between consecutive calls of a workgroup so that
// the reduction scratch of one call is not overwritten before everybody has read it.  Contains ONE barrier: call uniformly.
__device__ inline void tsdf_frame_block(const LayerDev& L, const TsdfFrameArgs& P, TsdfPairLds& S, int par, bool act, int i,
                                        int slot, u64 key, bool cand, bool is_new, long long* ph = nullptr) {
  const int t = threadIdx.x & 127, wv = threadIdx.x >> 6;
  const MapConsts& mc = P.mc;
  int hit = 0, freev = 1;
  float wmx = 0.0f;
  const bool decayed = P.decay_f > 0.0f;  // uniform
  if (act) {
    int bx, by, bz;
    unpack_key(key, bx, by, bz);
    float4* vox = reinterpret_cast<float4*>(L.pool) + (size_t)slot * (kVPB / 2) + t * 2;
    TsdfBlockAcc acc;
    tsdf_voxel_group_bf<4, false, false>(mc, P.cam, P.T_C_L, P.depth, 0.0f, P.decay_f, decayed ? 1 : 0, true, bx, by, bz, t * 4, cand, is_new, vox, acc, ph);
    hit = acc.hit;
    freev = acc.freev;
    wmx = acc.wmx;
  }
  const int w_free = __all(freev), w_hit = __any(hit);
  wmx = wave_max_f32(wmx);
  if ((threadIdx.x & 63) == 0) {
    S.free_[par][wv] = w_free;
    S.hit[par][wv] = w_hit;
    S.wmax[par][wv] = wmx;
  }
  __syncthreads();
  if (act && t == 0) {
    const int h2 = (threadIdx.x >> 7) * 2;
    if (cand || decayed) L.block_free[slot] = (S.free_[par][h2] && S.free_[par][h2 + 1]) ? 1 : 0;
    const int any = S.hit[par][h2] | S.hit[par][h2 + 1];
    P.flags_out[i] = any ? 1 : 0;
    if (any) P.cell_key_out[i] = key;
    L.wmax[slot] = fmaxf(S.wmax[par][h2], S.wmax[par][h2 + 1]);
  }
}

// One round of the new-block hand-over: the two halves of the workgroup take ranks k0 and k0 + 1 as soon as the allocation
// workgroups of this launch have published them (they may still be scanning later cells) and integrate them from zeroed
// voxels.  Workgroup-uniform result: 1 = round done, 0 = the published total says there is no rank k0 (ranks are granted in
// order: nothing beyond a missing one), -1 = a half did not see its record within `bound` polls.
__device__ inline int new_block_round(const LayerDev& L, const TsdfFrameArgs& P, TsdfPairLds& S, int& par, int k0, int n_old, int bound,
                                      bool pretend_timeout) {
  const int half = threadIdx.x >> 7;
  const int k = k0 + half;
  if ((threadIdx.x & 127) == 0) {
    int got = 0;
    u64 w0 = 0;
    for (int spins = 0; spins < bound && !pretend_timeout; ++spins) {
      w0 = __hip_atomic_load(P.pub + kPubRec + 3 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(w0 >> 32) == P.tag) {
        got = 1;
        break;
      }
      const u64 tot = __hip_atomic_load(P.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(tot >> 32) == P.tag) {
        // (a rank below the total was published before the total: one more look at its word settles it)
        if (k >= (int)(unsigned)(tot & 0xffffffffull)) {
          got = -1;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(4);
    }
    if (got == 1) {
      u64 w1, w2;
      int spins = 0;
      do {
        w1 = __hip_atomic_load(P.pub + kPubRec + 1 + 3 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w2 = __hip_atomic_load(P.pub + kPubRec + 2 + 3 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while (((unsigned)(w1 >> 32) != P.tag || (unsigned)(w2 >> 32) != P.tag) && ++spins < bound);
      if ((unsigned)(w1 >> 32) != P.tag || (unsigned)(w2 >> 32) != P.tag) got = 0;
      S.slot[half] = (int)(unsigned)(w0 & 0xffffffffull);
      S.klo[half] = (unsigned)(w1 & 0xffffffffull);
      S.khi[half] = (unsigned)(w2 & 0xffffffffull);
    }
    S.act[half] = got == 1 ? 1 : (got == 0 ? -1 : 0);
  }
  __syncthreads();
  const int a0 = S.act[0], a1 = S.act[1];
  int status = 1;
  if (a0 < 0 || a1 < 0)
    status = -1;
  else if (!a0)
    status = 0;
  if (status == 1) {
    const bool act = S.act[half] != 0;
    const int slot = S.slot[half];
    const u64 key = ((u64)S.khi[half] << 32) | (u64)S.klo[half];
    tsdf_frame_block(L, P, S, par, act, n_old + k, slot, key, true, true);
    par ^= 1;
  }
  __syncthreads();  // S.slot / klo / khi / act are rewritten by the next round
  return status;
}

// The waiters of k_alloc_tsdf: workgroup d of n_new_wgs takes rounds d, d + n_new_wgs, ... (two ranks per round).
// The hand-over relies on the allocation workgroups of the SAME launch making progress while a waiter polls; workgroups are
// dispatched in index order and the producers lead the grid, so they are resident first -- but that is an observation about
// the hardware, not a guarantee of the programming model.  A waiter therefore polls with a bound and, instead of leaving
// allocated-but-never-initialised blocks behind when it expires, ABANDONS its remaining rounds: it publishes the first one as a
// tagged word and counts itself out.  The workgroup that terminates last sees in the same counter whether anybody abandoned
// and, if so, becomes the sweeper: it waits (much longer) for the published total and integrates every abandoned round itself.
// Common path cost: one atomic per waiter workgroup at exit.  Only if the sweeper cannot finish either is the map incomplete:
// it then raises error bit 1 (ctr[3]) and the pinned host flag, and the next API call on the mapper fails with
// MMF_ERR_BAD_STATE instead of integrating on top of a broken map.
__device__ inline void new_blocks_role(const LayerDev& L, const TsdfFrameArgs& P, TsdfPairLds& S, int d, int n_old) {
  const u64 tg = (u64)P.tag << 32;
  const int stride = P.n_new_wgs * 2;
  int par = 0, abandoned_at = -1;
  for (int k0 = d * 2;; k0 += stride) {
    const int st = new_block_round(L, P, S, par, k0, n_old, 1 << 22, P.debug_abandon != 0 && (d & 1) != 0);
    if (st < 0) abandoned_at = k0;
    if (st <= 0) break;
  }
  if (threadIdx.x == 0) {
    if (abandoned_at >= 0)
      __hip_atomic_store(P.ctl + 2 + d, tg | (u64)(unsigned)abandoned_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 prev = __hip_atomic_fetch_add(P.ctl, 1ull + (abandoned_at >= 0 ? (1ull << 32) : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int count = (int)(prev & 0xffffffffull) + 1, n_ab = (int)(prev >> 32) + (abandoned_at >= 0 ? 1 : 0);
    S.slot[0] = count == P.n_new_wgs ? n_ab : -1;
    if (count == P.n_new_wgs) __hip_atomic_store(P.ctl, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next frame counts from zero
  }
  __syncthreads();
  const int n_ab = S.slot[0];
  __syncthreads();
  if (n_ab <= 0) return;  // not the last workgroup, or nobody abandoned (always, so far)
  // ---- sweeper
  const int bound = P.debug_abandon == 2 ? 0 : (1 << 26);
  int total = -1;
  if (threadIdx.x == 0) {
    for (int spins = 0; spins < bound; ++spins) {
      const u64 tot = __hip_atomic_load(P.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(tot >> 32) == P.tag) {
        total = (int)(unsigned)(tot & 0xffffffffull);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    S.slot[1] = total;
  }
  __syncthreads();
  total = S.slot[1];
  __syncthreads();
  bool failed = total < 0;
  int found = 0;
  for (int w = 0; w < P.n_new_wgs && !failed; ++w) {
    // which workgroups abandoned is in their tagged words (written before they counted themselves out; polled until n_ab of
    // them have been seen -- a word of another frame never carries this frame's tag)
    if (threadIdx.x == 0) {
      u64 x = 0;
      for (int spins = 0; spins < (found + (P.n_new_wgs - w) > n_ab ? 1 : bound); ++spins) {
        x = __hip_atomic_load(P.ctl + 2 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == P.tag) break;
      }
      S.slot[1] = (unsigned)(x >> 32) == P.tag ? (int)(unsigned)(x & 0xffffffffull) : -1;
    }
    __syncthreads();
    const int from = S.slot[1];
    __syncthreads();
    if (from < 0) continue;
    found++;
    for (int k0 = from; k0 < total; k0 += stride) {
      const int st = new_block_round(L, P, S, par, k0, n_old, bound, false);
      if (st < 0) failed = true;
      if (st <= 0) break;
      if (threadIdx.x == 0) __hip_atomic_fetch_add(P.ctl + 1, (u64)((k0 + 1 < total) ? 2 : 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (found < n_ab) failed = true;
  if (failed && threadIdx.x == 0) {
    atomicOr(P.err, 2);
    if (P.host_err) *P.host_err = 1;
  }
}


struct AllocTsdfLds {
  int lds[34];
  int carry[4];  // [2]: the allocation workgroup's wait for its predecessors failed
  int ctx[4];
  u64 s_bad[kMaxMaskWords];
  TsdfPairLds S;
};

__device__ inline void alloc_role(const AllocTsdfArgs& A, AllocTsdfLds& Q, int w) {
  const long long tr0 = wg_trace_begin();
  alloc_grid_multi_body<4, 2>(A.J, A.stats, Q.lds, Q.carry, Q.ctx, w, A.alloc_wgs);
  wg_trace_end(tr0, kTrAllocJob);
}

__device__ inline void mask_cols_role(const AllocTsdfArgs& A, AllocTsdfLds& Q, int row) {
  const long long tr0 = wg_trace_begin();
  mask_colemit_row(A.M, row, Q.s_bad);
  wg_trace_end(tr0, kTrAllocMaskCols);
}

// existing blocks, two list-adjacent ones per workgroup, pairs in XCD-contiguous chunks (workgroup c of n_pair_wgs; the
// caller places the role at a multiple of 8 so that c mod 8 is the workgroup's XCD)
__device__ inline void tsdf_pairs_role(const AllocTsdfArgs& A, AllocTsdfLds& Q, int c) {
  const long long tr0 = wg_trace_begin();
  const LayerDev& L = A.J.L;
  const TsdfFrameArgs& P = A.P;  // (its constants in vector registers -- as k_tsdf_pass keeps them -- measured: 15.05 -> 14.87 us, not taken:
                                 // the batch form of this role is capped at 64 VGPRs)
  const int half = threadIdx.x >> 7;
  const int n_old = *P.n_old;
  const int npairs = (n_old + 1) >> 1;
  const int chunk = (npairs + 7) >> 3;
  int par = 0;
  for (int j = c; j < chunk * 8; j += P.n_pair_wgs, par ^= 1) {
    const int i = 2 * xcd_candidate(j, chunk) + half;
    const bool act = i < n_old;
    int slot = 0;
    u64 key = 0;
    bool cand = false;
    if (act) {
      slot = L.live[i];
      key = L.slot_key[slot];
      int bx, by, bz;
      unpack_key(key, bx, by, bz);
      const int gx = bx - P.ox, gy = by - P.oy, gz = bz - P.oz;
      if (gx >= 0 && gy >= 0 && gz >= 0 && gx < P.nx && gy < P.ny && gz < P.nz)
        cand = P.grid_flags[(gx * P.ny + gy) * P.nz + gz] == (uint8_t)P.grid_tag;
    }
#ifdef MMF_WG_TRACE
    // in-workgroup phases of thread 0's block (tools/wg_trace.py --phases): slots 6 .. 8 of a 9 x 8192-record trace buffer
    long long ph[5] = {0, 0, 0, 0, 0};
    if (wg_trace_on()) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      ph[0] = (long long)wall_clock64();  // slot, key and the raycast flag are known
    }
    tsdf_frame_block(L, P, Q.S, par, act, i, slot, key, cand, false, wg_trace_on() ? ph : nullptr);
    if (wg_trace_on() && threadIdx.x == 0 && blockIdx.x < 8192 && g_wg_trace_cap >= 9 * 8192) {
      ph[4] = (long long)wall_clock64();
      unsigned long long* r = g_wg_trace + 3 * (6 * 8192 + (int)blockIdx.x);
      r[0] = (unsigned long long)(cand ? 2 : 1), r[1] = (unsigned long long)tr0, r[2] = (unsigned long long)ph[0];
      r = g_wg_trace + 3 * (7 * 8192 + (int)blockIdx.x);
      r[0] = (unsigned long long)ph[1], r[1] = (unsigned long long)ph[2], r[2] = (unsigned long long)ph[3];
      r = g_wg_trace + 3 * (8 * 8192 + (int)blockIdx.x);
      r[0] = (unsigned long long)ph[4];
    }
#else
    tsdf_frame_block(L, P, Q.S, par, act, i, slot, key, cand, false);
#endif
  }
  wg_trace_end(tr0, kTrTsdfPass);
}

__device__ inline void tsdf_new_role(const AllocTsdfArgs& A, AllocTsdfLds& Q, int d) {
  const long long tr0 = wg_trace_begin();
  new_blocks_role(A.J.L, A.P, Q.S, d, *A.P.n_old);
  wg_trace_end(tr0, kTrTsdfNew);
}

// grid: [allocation | new-block waiters | padding to `lead` (a multiple of 8) | existing-block pairs | mask columns]
// The waiters poll for words of the allocation workgroups of their own launch: producers first, in dispatch order.  They
// follow them directly -- resident from the start, asleep between polls -- so that a new block is integrated as soon as it
// is published (at the end of the grid they only got a slot when earlier workgroups retired, and closed the launch ~2 us
// after the pass over the existing blocks had ended).  The mask columns, short and needed by nobody in this launch, come
// last: the TSDF pairs take every free slot at once and the columns fill in behind them.
__global__ __launch_bounds__(256) MMF_SGPR96 void k_alloc_tsdf(AllocTsdfArgs A, int lead) {
  __shared__ AllocTsdfLds Q;
  const int b = (int)blockIdx.x;
  if (b < A.alloc_wgs) return alloc_role(A, Q, b);
  if (b - A.alloc_wgs < A.P.n_new_wgs) return tsdf_new_role(A, Q, b - A.alloc_wgs);
  if (b < lead) return;  // padding: the TSDF pairs start at a multiple of 8 (workgroup -> XCD residue)
  const int c = b - lead;
  if (c < A.P.n_pair_wgs) return tsdf_pairs_role(A, Q, c);
  if (c - A.P.n_pair_wgs < A.mask_rows) mask_cols_role(A, Q, c - A.P.n_pair_wgs);
}

// Two frames in one launch (mmf_integrate_frame_multi): BOTH frames' producers lead the grid,
//   [alloc 0 | alloc 1 | new 0 | new 1 | padding | pairs 0 | pairs 1 | mask columns 0 | mask columns 1].
__global__ __launch_bounds__(256) MMF_SGPR96 void k_alloc_tsdf2(AllocTsdfArgs A0, AllocTsdfArgs A1, int lead) {
  __shared__ AllocTsdfLds Q;
  int b = (int)blockIdx.x;
  if (b < A0.alloc_wgs) return alloc_role(A0, Q, b);
  b -= A0.alloc_wgs;
  if (b < A1.alloc_wgs) return alloc_role(A1, Q, b);
  b -= A1.alloc_wgs;
  if (b < A0.P.n_new_wgs) return tsdf_new_role(A0, Q, b);
  b -= A0.P.n_new_wgs;
  if (b < A1.P.n_new_wgs) return tsdf_new_role(A1, Q, b);
  if ((int)blockIdx.x < lead) return;
  int c = (int)blockIdx.x - lead;
  if (c < A0.P.n_pair_wgs) return tsdf_pairs_role(A0, Q, c);
  c -= A0.P.n_pair_wgs;
  if (c < A1.P.n_pair_wgs) return tsdf_pairs_role(A1, Q, c);
  c -= A1.P.n_pair_wgs;
  if (c < A0.mask_rows) return mask_cols_role(A0, Q, c);
  c -= A0.mask_rows;
  if (c < A1.mask_rows) mask_cols_role(A1, Q, c);
}

// N frames: every frame's producers lead the grid,
//   [alloc 0 .. alloc n-1 | new 0 .. new n-1 | padding | pairs 0 .. pairs n-1 | mask columns 0 .. n-1].
__global__ __launch_bounds__(256, 8) void k_alloc_tsdf_batch(AllocTsdfBatch P) {
  __shared__ AllocTsdfLds Q;
  int b = (int)blockIdx.x;
  for (int q = 0; q < P.n; ++q) {
    if (b < P.a[q].alloc_wgs) return alloc_role(P.a[q], Q, b);
    b -= P.a[q].alloc_wgs;
  }
  for (int q = 0; q < P.n; ++q) {
    if (b < P.a[q].P.n_new_wgs) return tsdf_new_role(P.a[q], Q, b);
    b -= P.a[q].P.n_new_wgs;
  }
  if ((int)blockIdx.x < P.lead) return;
  int c = (int)blockIdx.x - P.lead;
  for (int q = 0; q < P.n; ++q) {
    if (c < P.a[q].P.n_pair_wgs) return tsdf_pairs_role(P.a[q], Q, c);
    c -= P.a[q].P.n_pair_wgs;
  }
  for (int q = 0; q < P.n; ++q) {
    if (c < P.a[q].mask_rows) return mask_cols_role(P.a[q], Q, c);
    c -= P.a[q].mask_rows;
  }
}

MMF_DEFINE_WG_TRACE_SETTER(set_wg_trace_map)

// ------------------------------------------------------------------------------------------------
// 4. Decay: W *= factor over every live TSDF block; blocks whose voxels all fell below the threshold
//    are flagged, then one workgroup compacts the live list in place (order preserving), pushes the
//    freed slots and the hash is rebuilt from the survivors.
// ------------------------------------------------------------------------------------------------
template <bool ERASE>
__device__ inline void decay_body(const LayerDev& L, const MapConsts& mc, uint8_t* __restrict__ kill, int* any_kill, int bid, int nb) {
  const int n = L.ctr[0];
  for (int i = bid; i < n; i += nb) {
    const int slot = L.live[i];
    float4* vox = reinterpret_cast<float4*>(L.pool) + (size_t)slot * (kVPB / 2) + threadIdx.x;  // 2 voxels
    float4 a = *vox;
    a.y = a.y * mc.decay_factor;
    a.w = a.w * mc.decay_factor;
    *vox = a;
    int alive = (!(a.y < mc.decay_thr)) || (!(a.w < mc.decay_thr));
    int any_alive = __syncthreads_or(alive);
    const int all_free = __syncthreads_and((a.y > 1e-4f && a.x == mc.trunc && a.w > 1e-4f && a.z == mc.trunc) ? 1 : 0);
    if (threadIdx.x == 0) L.block_free[slot] = all_free ? 1 : 0;
    if (threadIdx.x == 0 && !any_alive && mc.dealloc_decayed) {
      kill[i] = 1;
      *any_kill = 1;
      if (ERASE) {  // the compaction (live_compact_body<.., false>) then only moves list entries
        const u64 key = L.slot_key[slot];
        hash_erase(L, key);
        dense_set(L, key, 0);
        L.slot_key[slot] = kEmptyKey;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_decay(LayerDev L, MapConsts mc, uint8_t* __restrict__ kill, int* any_kill) {
  decay_body<false>(L, mc, kill, any_kill, blockIdx.x, gridDim.x);
}

// Scalable allocation (alloc_big_body): roles [job 0: nwg0 chunks | job 1: nwg1 chunks | mask column rows].  One launch for the
// TSDF allocation of a large view grid (+ the frame's mask columns) or for the colour + feature allocation of a large pool.
__global__ __launch_bounds__(256) void k_alloc_big(AllocJob J0, AllocJob J1, int nwg0, int nwg1, int G0, int G1, long long* stats, MaskJob M,
                                                  int mask_rows) {
  __shared__ AllocBigLds S;
  __shared__ u64 s_bad[kMaxMaskWords];
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
  if (b < mask_rows) mask_colemit_row(M, b, s_bad);
}

// Scalable deallocation (live_compact_big_body): the dead blocks of a decay leave the live list / the index in one launch of
// ceil(live / 1024) workgroups.  WMAX: decided from the blocks' largest weights (the light decay of a fused frame).
template <bool WMAX>
__global__ __launch_bounds__(256) void k_live_compact_big(LayerDev L, uint8_t* kill, u64* lb, unsigned tag, int* rebuild, int* snap6,
                                                         float decay_f, float decay_thr, int* any_kill) {
  __shared__ int lds[10];
  __shared__ int sh[8];
  live_compact_big_body<WMAX>(L, kill, lb, tag, rebuild, snap6, decay_f, decay_thr, lds, sh, (int)blockIdx.x, (int)gridDim.x);
  if (!WMAX && any_kill && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *any_kill = 0;
}

__global__ __launch_bounds__(1024) void k_live_compact(LayerDev L, uint8_t* __restrict__ kill, int* any_kill) {
  __shared__ int lds[34];
  __shared__ int carry[2];
  live_compact_body<16, 4, true>(L, kill, any_kill, lds, carry);
}

__global__ __launch_bounds__(256) void k_hash_clear_if(LayerDev L, const int* cond) {
  if (cond && !*cond) return;
  for (unsigned h = blockIdx.x * blockDim.x + threadIdx.x; h <= L.hmask; h += gridDim.x * blockDim.x) L.htab[h].key = kEmptyKey;
}

// diagnostics (mmf_debug_count_tombstones): the tombstones that are in the table, to hold against the layer's counter
__global__ __launch_bounds__(256) void k_count_tombstones(LayerDev L, unsigned long long* out) {
  unsigned n = 0;
  for (unsigned h = blockIdx.x * 256 + threadIdx.x; h <= L.hmask; h += gridDim.x * 256) n += L.htab[h].key == kTombKey ? 1u : 0u;
  if (n) atomicAdd(out, (unsigned long long)n);
}

__global__ __launch_bounds__(256) void k_hash_insert_live_if(LayerDev L, const int* cond) {
  if (cond && !*cond) return;
  if (cond && blockIdx.x == 0 && threadIdx.x == 0) {  // the requested rebuild is being served: no tombstone is left (k_hash_clear_if ran)
    L.ctr[4] = 0;
    L.ctr[5]++;  // diagnostics (mmf_debug_hash_state): table rebuilds since the layer was reset
  }
  const int n = L.ctr[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int slot = L.live[i];
    hash_insert(L, L.slot_key[slot], slot);
  }
}

__global__ void k_reset_layer(LayerDev L) {
  if (threadIdx.x < 8 && blockIdx.x == 0) L.ctr[threadIdx.x] = 0;
  if (L.dense)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < L.d_ncells; c += gridDim.x * blockDim.x) L.dense[c] = 0;
}

__global__ void k_set_int(int* p, int v) { *p = v; }
__global__ void k_clear_bits(int* p, int bits) { atomicAnd(p, ~bits); }
void launch_clear_bits(int* word, int bits, hipStream_t s) { hipLaunchKernelGGL(k_clear_bits, dim3(1), dim3(1), 0, s, word, bits); }

__global__ __launch_bounds__(256) void k_invert_mask(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] == 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// 5. Views (allocation order) and point queries.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_get_indices(LayerDev L, int32_t* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || i >= L.ctr[0]) return;
  int x, y, z;
  unpack_key(L.slot_key[L.live[i]], x, y, z);
  out[3 * i] = x;
  out[3 * i + 1] = y;
  out[3 * i + 2] = z;
}

// copy `bytes_per_block` (multiple of 16) of payload A of every live block, in live order
__global__ __launch_bounds__(256) void k_gather_pool(LayerDev L, size_t bytes_per_block, char* __restrict__ out, int n) {
  const int nl = L.ctr[0];
  const size_t n16 = bytes_per_block / 16;
  for (int i = blockIdx.x; i < n && i < nl; i += gridDim.x) {
    const uint4* src = reinterpret_cast<const uint4*>(L.pool + (size_t)L.live[i] * bytes_per_block);
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)i * bytes_per_block);
    for (size_t k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
  }
}

__global__ __launch_bounds__(256) void k_gather_poolw(LayerDev L, float* __restrict__ out, int n) {
  const int nl = L.ctr[0];
  for (int i = blockIdx.x; i < n && i < nl; i += gridDim.x) {
    const float* src = L.poolw + (size_t)L.live[i] * kVPB;
    float* dst = out + (size_t)i * kVPB;
    for (int k = threadIdx.x; k < kVPB; k += blockDim.x) dst[k] = src[k];
  }
}

// colour blocks are stored as {uchar4 rgba, float w}; unpack to rgb[512][3] + w[512]
__global__ __launch_bounds__(256) void k_gather_color(LayerDev L, uint8_t* __restrict__ rgb, float* __restrict__ w, int n) {
  const int nl = L.ctr[0];
  for (int i = blockIdx.x; i < n && i < nl; i += gridDim.x) {
    const uint2* src = reinterpret_cast<const uint2*>(L.pool) + (size_t)L.live[i] * kVPB;
    for (int k = threadIdx.x; k < kVPB; k += blockDim.x) {
      uint2 e = src[k];
      uint8_t* o = rgb + ((size_t)i * kVPB + k) * 3;
      o[0] = e.x & 0xff;
      o[1] = (e.x >> 8) & 0xff;
      o[2] = (e.x >> 16) & 0xff;
      w[(size_t)i * kVPB + k] = __uint_as_float(e.y);
    }
  }
}

__global__ __launch_bounds__(256) void k_query_tsdf(LayerDev L, MapConsts mc, const float* __restrict__ pts, int n,
                                                   float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  int lin;
  u64 key = voxel_at(mc, p, lin);
  int slot = layer_lookup(L, key);
  float2 r = make_float2(0.0f, 0.0f);
  if (slot >= 0) r = reinterpret_cast<const float2*>(L.pool)[(size_t)slot * kVPB + lin];
  out[2 * i] = r.x;
  out[2 * i + 1] = r.y;
}

// one wave per query point: out[i][0..C-1] = features (f32), out[i][C] = weight
__global__ __launch_bounds__(256) void k_query_feature(LayerDev L, MapConsts mc, const float* __restrict__ pts, int n,
                                                      float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  int lin;
  u64 key = voxel_at(mc, p, lin);
  int slot = layer_lookup(L, key);
  const int C = mc.C;
  float* o = out + (size_t)i * (C + 1);
  if (slot < 0) {
    for (int k = lane; k <= C; k += 64) o[k] = 0.0f;
    return;
  }
  const __half* f = reinterpret_cast<const __half*>(L.pool) + ((size_t)slot * kVPB + lin) * C;
  for (int k = lane; k < C; k += 64) o[k] = __half2float(f[k]);
  if (lane == 0) o[C] = L.poolw[(size_t)slot * kVPB + lin];
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static RaycastJob make_raycast_job(const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, const float* depth, const uint8_t* mask,
                                   float min_d, int sub, const ViewGrid& vg, uint8_t* flags, int& n_wgs) {
  RaycastJob R;
  R.mc = mc;
  R.cam = cam;
  R.T_L_C = T_L_C;
  R.depth = depth;
  R.mask = mask;
  R.mask_invert = 0;
  R.min_d = min_d;
  R.sub = sub;
  R.Wsub = (cam.W + sub - 1) / sub;
  R.Hsub = (cam.H + sub - 1) / sub;
  R.vg = vg;
  R.flags = flags;
  R.flag_value = 1;
  const int ntiles = ((R.Wsub + 7) / 8) * ((R.Hsub + 7) / 8);
  n_wgs = (ntiles + 3) / 4;
  return R;
}

void launch_raycast(const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, const float* depth, const uint8_t* mask, float min_d,
                    int sub, const ViewGrid& vg, uint8_t* flags, hipStream_t s) {
  int n_wgs;
  RaycastJob R = make_raycast_job(mc, cam, T_L_C, depth, mask, min_d, sub, vg, flags, n_wgs);
  if (n_wgs <= 0) return;
  const int ncells = vg.nx * vg.ny * vg.nz;
  const bool spec = (mc.spec_flags & (kSpecBlockDiv | kSpecBandMark)) != 0;  // (the stand-alone launch is the only one built with them)
  const size_t shm = (size_t)((ncells + 3) / 4) * 4;
  if (ncells <= kRaycastLdsCells) {
    if (spec)
      hipLaunchKernelGGL((k_raycast_mark<true, true>), dim3(n_wgs), dim3(256), shm, s, R);
    else
      hipLaunchKernelGGL((k_raycast_mark<true, false>), dim3(n_wgs), dim3(256), shm, s, R);
  } else if (spec)
    hipLaunchKernelGGL((k_raycast_mark<false, true>), dim3(n_wgs), dim3(256), 0, s, R);
  else
    hipLaunchKernelGGL((k_raycast_mark<false, false>), dim3(n_wgs), dim3(256), 0, s, R);
}

// raycast + mask row pass (+ pending decay) of a frame: arguments of its share of the first launch
FrontArgs make_front_args(const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, const float* depth, const uint8_t* mask, float min_d,
                          int sub, const ViewGrid& vg, uint8_t* flags, const MaskJob& M, const LayerDev* decay_layer, bool light_decay,
                          uint8_t* kill, int* any_kill, int* snap_ctr, int flag_value) {
  FrontArgs A;
  A.R = make_raycast_job(mc, cam, T_L_C, depth, mask, min_d, sub, vg, flags, A.n_ray_wgs);
  A.R.flag_value = flag_value;
  A.R.mask_invert = M.invert;
  A.M = M;
  A.D = DecayJob{};
  if (decay_layer) {
    A.D.L = *decay_layer;
    A.D.kill = kill;
    A.D.any_kill = any_kill;
    A.D.light = light_decay ? 1 : 0;
    A.D.n_wgs = light_decay ? 1 : grid_for(hinted(decay_layer->hint_live, decay_layer->cap), 4096);
  }
  A.snap_ctr = snap_ctr;
  A.n_wgs = A.n_ray_wgs + M.H + A.D.n_wgs;
  return A;
}

static inline int front_cells(const FrontArgs& A) { return A.R.vg.nx * A.R.vg.ny * A.R.vg.nz; }

void launch_front(const FrontArgs* A, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  int ncells = front_cells(A[0]);
  if (n > 1 && front_cells(A[1]) > ncells) ncells = front_cells(A[1]);
  const bool lds = ncells <= kRaycastLdsCells;
  const size_t shm = lds ? (size_t)((ncells + 3) / 4) * 4 : 0;
  if (n == 1) {
    const dim3 grid(A[0].n_wgs);
    if (lds)
      hipExtLaunchKernelGGL(k_front<true>, grid, dim3(256), shm, s, ev_start, ev_stop, 0, A[0]);
    else
      hipExtLaunchKernelGGL(k_front<false>, grid, dim3(256), 0, s, ev_start, ev_stop, 0, A[0]);
  } else {
    const dim3 grid(A[0].n_wgs + A[1].n_wgs);
    if (lds)
      hipExtLaunchKernelGGL(k_front2<true>, grid, dim3(256), shm, s, ev_start, ev_stop, 0, A[0], A[1]);
    else
      hipExtLaunchKernelGGL(k_front2<false>, grid, dim3(256), 0, s, ev_start, ev_stop, 0, A[0], A[1]);
  }
}

void launch_hash_rebuild_if(const LayerDev& L, const int* rebuild, hipStream_t s);

// the light decay's scalable compaction | this frame's raycast + mask rows, then (serve_rebuild) the conditional rebuild pair
void launch_front_compact_big(const FrontArgs& A, const LayerDev* compact, u64* lb, unsigned tag, int* rebuild, float decay_f, float decay_thr,
                              int live_upper, bool serve_rebuild, const AppTail* tail, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int ncells = front_cells(A);
  const bool lds = ncells <= kRaycastLdsCells;
  size_t shm = lds ? (size_t)((ncells + 3) / 4) * 4 : 0;
  const int nc = compact ? alloc_big_wgs(live_upper) : 0;
  const LayerDev L = compact ? *compact : LayerDev{};
  if (tail) {
    shm = shm < sizeof(FeatLds) ? sizeof(FeatLds) : shm;
    const int nb_gate = app_tail_grid(*tail);
    const dim3 grid(nc + A.n_wgs + nb_gate);
    if (lds)
      hipExtLaunchKernelGGL((k_front_compact_big<true, true>), grid, dim3(256), shm, s, ev_start, ev_stop, 0, A, L, lb, tag, rebuild, decay_f, decay_thr, nc,
                            tail->Ac, tail->Af, tail->synth, tail->Ws, tail->Hs, nb_gate);
    else
      hipExtLaunchKernelGGL((k_front_compact_big<false, true>), grid, dim3(256), shm, s, ev_start, ev_stop, 0, A, L, lb, tag, rebuild, decay_f, decay_thr, nc,
                            tail->Ac, tail->Af, tail->synth, tail->Ws, tail->Hs, nb_gate);
  } else {
    const dim3 grid(nc + A.n_wgs);
    const AppArgs none{};
    if (lds)
      hipExtLaunchKernelGGL((k_front_compact_big<true, false>), grid, dim3(256), shm, s, ev_start, ev_stop, 0, A, L, lb, tag, rebuild, decay_f, decay_thr, nc,
                            none, none, (const float*)nullptr, 0, 0, 0);
    else
      hipExtLaunchKernelGGL((k_front_compact_big<false, false>), grid, dim3(256), 0, s, ev_start, ev_stop, 0, A, L, lb, tag, rebuild, decay_f, decay_thr, nc,
                            none, none, (const float*)nullptr, 0, 0, 0);
  }
  if (compact && !L.dense && serve_rebuild) launch_hash_rebuild_if(L, rebuild, s);
}

void launch_front_app(const FrontArgs& A, const AppTail& T, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int ncells = front_cells(A);
  const bool lds = ncells <= kRaycastLdsCells;
  size_t shm = lds ? (size_t)((ncells + 3) / 4) * 4 : 0;
  shm = shm < sizeof(FeatLds) ? sizeof(FeatLds) : shm;
  const int nb_gate = app_tail_grid(T);
  const dim3 grid(A.n_wgs + nb_gate);
  if (lds)
    hipExtLaunchKernelGGL(k_front_app<true>, grid, dim3(256), shm, s, ev_start, ev_stop, 0, A, T.Ac, T.Af, T.synth, T.Ws, T.Hs, nb_gate);
  else
    hipExtLaunchKernelGGL(k_front_app<false>, grid, dim3(256), shm, s, ev_start, ev_stop, 0, A, T.Ac, T.Af, T.synth, T.Ws, T.Hs, nb_gate);
}

void launch_front_batch(const FrontArgs* A, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  FrontBatch P;
  P.n = n;
  int ncells = 0, total = 0;
  for (int q = 0; q < n; ++q) {
    P.a[q] = A[q];
    ncells = front_cells(A[q]) > ncells ? front_cells(A[q]) : ncells;
    total += A[q].n_wgs;
  }
  const bool lds = ncells <= kRaycastLdsCells;
  if (lds)
    hipExtLaunchKernelGGL(k_front_batch<true>, dim3(total), dim3(256), (size_t)((ncells + 3) / 4) * 4, s, ev_start, ev_stop, 0, P);
  else
    hipExtLaunchKernelGGL(k_front_batch<false>, dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, P);
}

void launch_front_batch_app(const FrontArgs* A, int n, const AppFrameArgs* G, int ng, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  FrontBatch P;
  AppFrameBatch Q;
  P.n = n;
  Q.n = ng;
  int ncells = 0, total = 0;
  for (int q = 0; q < n; ++q) {
    P.a[q] = A[q];
    ncells = front_cells(A[q]) > ncells ? front_cells(A[q]) : ncells;
    total += A[q].n_wgs;
  }
  for (int q = 0; q < ng; ++q) {
    Q.a[q] = G[q];
    total += G[q].nb;
  }
  const bool lds = ncells <= kRaycastLdsCells;
  size_t shm = lds ? (size_t)((ncells + 3) / 4) * 4 : 0;
  shm = shm < sizeof(FeatLds) ? sizeof(FeatLds) : shm;
  if (lds)
    hipExtLaunchKernelGGL(k_front_batch_app<true>, dim3(total), dim3(256), shm, s, ev_start, ev_stop, 0, P, Q);
  else
    hipExtLaunchKernelGGL(k_front_batch_app<false>, dim3(total), dim3(256), shm, s, ev_start, ev_stop, 0, P, Q);
}

constexpr int kFusedAllocMaxCells = 16384;

static AllocJob make_alloc_job(const LayerDev& L, const KeySrc& ks, const Scratch& sc, int ncells, int stat_upd, int stat_new) {
  AllocJob J;
  J.L = L;
  J.ks = ks;
  J.sc = sc;
  J.ncells = ncells;
  J.stat_upd = stat_upd;
  J.stat_new = stat_new;
  return J;
}

void launch_compact_alloc(const LayerDev& L, const KeySrc& ks, const Scratch& sc, int ncells, long long* stats, int stat_upd,
                          int stat_new, hipStream_t s) {
  if (ncells <= kFusedAllocMaxCells) {
    if (L.dense)
      hipLaunchKernelGGL(k_alloc_fused<true>, dim3(1), dim3(1024), 0, s, make_alloc_job(L, ks, sc, ncells, stat_upd, stat_new), stats);
    else
      hipLaunchKernelGGL(k_alloc_fused<false>, dim3(1), dim3(1024), 0, s, make_alloc_job(L, ks, sc, ncells, stat_upd, stat_new), stats);
    return;
  }
  int ntiles = (ncells + 1023) / 1024;
  if (ntiles <= 0) ntiles = 1;
  hipLaunchKernelGGL(k_count_tiles, dim3(ntiles), dim3(256), 0, s, L, ks, sc, ncells);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, s, L, sc, ntiles, stats, stat_upd, stat_new);
  hipLaunchKernelGGL(k_emit, dim3(ntiles), dim3(256), 0, s, L, ks, sc, ncells);
}

int alloc_big_wgs(int ncells) { return ncells <= 0 ? 1 : (ncells + kBigChunk - 1) / kBigChunk; }
// groups of 4 cells per thread so that a job is at most ~128 chunks where it can (look-back hops are memory round trips)
int alloc_big_groups(int ncells) {
  static const int forced = std::getenv("MMF_DEBUG_BIG_G") ? std::atoi(std::getenv("MMF_DEBUG_BIG_G")) : 0;
  if (forced > 0) return forced > kBigMaxG ? kBigMaxG : forced;
  int G = (alloc_big_wgs(ncells) + 127) / 128;
  return G < 1 ? 1 : (G > kBigMaxG ? kBigMaxG : G);
}
bool alloc_big_supported(const LayerDev& L) { return L.cap < (int)kLbCountMax; }

// jobs: 1 or 2 allocation jobs (AllocJob::sc.lb / lb_tag set by the caller); M: the frame's mask job whose column pass rides
// along (may be null)
void launch_alloc_big(const AllocJob* jobs, int njobs, long long* stats, const MaskJob* M, hipStream_t s) {
  MaskJob mj{};
  int rows = 0;
  if (M) {
    mj = *M;
    rows = M->Hf;
  }
  const int G0 = alloc_big_groups(jobs[0].ncells), G1 = njobs > 1 ? alloc_big_groups(jobs[1].ncells) : 1;
  const int n0 = (alloc_big_wgs(jobs[0].ncells) + G0 - 1) / G0, n1 = njobs > 1 ? (alloc_big_wgs(jobs[1].ncells) + G1 - 1) / G1 : 0;
  hipLaunchKernelGGL(k_alloc_big, dim3(n0 + n1 + rows), dim3(256), 0, s, jobs[0], jobs[njobs > 1 ? 1 : 0], n0, n1, G0, G1, stats, mj, rows);
}

// live_upper: an upper bound of the live count known to the host (the pool capacity if nothing better)
void launch_live_compact_big(const LayerDev& L, bool wmax, uint8_t* kill, int* any_kill, u64* lb, unsigned tag, int* rebuild, int* snap6,
                             float decay_f, float decay_thr, int live_upper, hipStream_t s, bool serve_rebuild) {
  const int nwg = alloc_big_wgs(live_upper);
  if (wmax)
    hipLaunchKernelGGL(k_live_compact_big<true>, dim3(nwg), dim3(256), 0, s, L, kill, lb, tag, rebuild, snap6, decay_f, decay_thr, any_kill);
  else
    hipLaunchKernelGGL(k_live_compact_big<false>, dim3(nwg), dim3(256), 0, s, L, kill, lb, tag, rebuild, snap6, decay_f, decay_thr, any_kill);
  if (!L.dense && serve_rebuild) {  // the amortised rebuild, on the flag the last chunk raised
    // (these two launches almost always find the flag down and leave: 4.5 us each, whatever their grid -- the cost is the launch and
    // the flag's load.  So the caller asks for them only every few compactions: the table's layout is not observable, a request
    // stays up until it is served, and between two chances the tombstones can grow by a few frames' deallocations at most)
    launch_hash_rebuild_if(L, rebuild, s);
  }
}

void launch_hash_rebuild_if(const LayerDev& L, const int* rebuild, hipStream_t s) {
  int hb = (int)((L.hmask + 1 + 255) / 256);
  if (hb > 1024) hb = 1024;
  hipLaunchKernelGGL(k_hash_clear_if, dim3(hb), dim3(256), 0, s, L, rebuild);
  hipLaunchKernelGGL(k_hash_insert_live_if, dim3(grid_for((L.cap + 255) / 256, 1024)), dim3(256), 0, s, L, rebuild);
}

bool alloc_jobs_fusable(int ncells0, int ncells1) { return ncells0 <= kFusedAllocMaxCells && ncells1 <= kFusedAllocMaxCells; }

// njobs (1 or 2) allocation jobs + (optionally) the mask column pass in one launch
void launch_alloc_jobs(const AllocJob* jobs, int njobs, long long* stats, const MaskJob* M, hipStream_t s) {
  MaskJob mj{};
  int rows = 0;
  if (M) {
    mj = *M;
    rows = M->Hf;
  }
  const AllocJob& j1 = jobs[njobs > 1 ? 1 : 0];
  const bool dense = jobs[0].L.dense && j1.L.dense;  // bounded workspace: the specialisations without the hash paths
  if (dense && njobs == 1 && jobs[0].ks.mode == 0)  // the TSDF allocation of the fused frame
    hipLaunchKernelGGL((k_alloc_jobs<true, 0, false>), dim3(njobs + rows), dim3(1024), 0, s, jobs[0], j1, njobs, stats, mj, rows);
  else if (dense)
    hipLaunchKernelGGL((k_alloc_jobs<true, -1, true>), dim3(njobs + rows), dim3(1024), 0, s, jobs[0], j1, njobs, stats, mj, rows);
  else
    hipLaunchKernelGGL((k_alloc_jobs<false, -1, true>), dim3(njobs + rows), dim3(1024), 0, s, jobs[0], j1, njobs, stats, mj, rows);
}

// Grid size from the last count the device published to pinned host memory (stale by a frame or two, which is
// fine: every block kernel strides over its work list, so any grid size is correct).  Empty workgroups are not
// free -- dispatching 3 744 of them costs ~6 us -- so grids follow the real count with 25 % headroom.
int hinted(const int* hint, int upper) {
  if (!hint) return upper;
  const int v = *reinterpret_cast<const volatile int*>(hint);
  if (v <= 0) return upper;
  long long g = (long long)v + v / 4 + 64;
  return g < upper ? (int)g : upper;
}

static inline int grid_for(int upper, int cap) {
  int g = upper < cap ? upper : cap;
  g = (g + 7) & ~7;  // multiple of 8: a workgroup keeps its XCD residue across the stride loop
  return g < 8 ? 8 : g;
}

void launch_tsdf_integrate(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth,
                           const uint8_t* mask, float min_d, const Scratch& sc, int max_cand, hipStream_t s) {
  const dim3 grid(grid_for(hinted(sc.hint_cand, max_cand), 8192));
  switch (arith_mode(mc.spec_flags)) {  // bit 0 fma_contraction, bit 1 bilinear_four_weight_sum
    case 1: hipLaunchKernelGGL(k_tsdf_integrate<1>, grid, dim3(512), 0, s, L, mc, cam, T_C_L, depth, mask, min_d, sc); break;
    case 2: hipLaunchKernelGGL(k_tsdf_integrate<2>, grid, dim3(512), 0, s, L, mc, cam, T_C_L, depth, mask, min_d, sc); break;
    case 3: hipLaunchKernelGGL(k_tsdf_integrate<3>, grid, dim3(512), 0, s, L, mc, cam, T_C_L, depth, mask, min_d, sc); break;
    default: hipLaunchKernelGGL(k_tsdf_integrate<0>, grid, dim3(512), 0, s, L, mc, cam, T_C_L, depth, mask, min_d, sc); break;
  }
}

// ---- import of a saved layer (Mapper.load_from_file): block i of the file takes pool slot i, live position i -----------
__global__ __launch_bounds__(256) void k_import_index(LayerDev L, const int32_t* __restrict__ idx, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    L.ctr[0] = n;
    L.ctr[1] = 0;
    L.ctr[2] = n;
    L.ctr[4] = 0;
    if (L.hint_live) *L.hint_live = n;
  }
  if (i >= n) return;
  const int x = idx[3 * i], y = idx[3 * i + 1], z = idx[3 * i + 2];
  if (x <= -kKeyOff || x >= kKeyOff || y <= -kKeyOff || y >= kKeyOff || z <= -kKeyOff || z >= kKeyOff) {
    atomicOr(&L.ctr[3], 2);
    return;
  }
  const u64 key = pack_key(x, y, z);
  if (L.dense) {  // the saved map must fit this mapper's workspace
    const int dx = x - L.d_lo[0], dy = y - L.d_lo[1], dz = z - L.d_lo[2];
    if (dx < 0 || (unsigned)dy >= (unsigned)L.d_ny || (unsigned)dz >= (unsigned)L.d_nz ||
        (dx * L.d_ny + dy) * L.d_nz + dz >= L.d_ncells) {
      atomicOr(&L.ctr[3], 2);
      return;
    }
  }
  L.slot_key[i] = key;
  L.live[i] = i;
  hash_insert(L, key, i);
  dense_set(L, key, i + 1);
  if (L.stamp) L.stamp[i] = 0;
}

// colour payload {uchar4 rgb_, f32 w}[512] from the exported planes rgb [n,512,3] u8 and w [n,512] f32
__global__ __launch_bounds__(256) void k_import_color(LayerDev L, const uint8_t* __restrict__ rgb, const float* __restrict__ w, int n) {
  for (int b = blockIdx.x; b < n; b += gridDim.x)
    for (int v = threadIdx.x; v < kVPB; v += 256) {
      const size_t q = (size_t)b * kVPB + v;
      uint2 e;
      e.x = (unsigned)rgb[3 * q] | ((unsigned)rgb[3 * q + 1] << 8) | ((unsigned)rgb[3 * q + 2] << 16);
      e.y = __float_as_uint(w[q]);
      reinterpret_cast<uint2*>(L.pool)[q] = e;
    }
}

// block_free summary of every live TSDF block (after an import)
__global__ __launch_bounds__(256) void k_block_free_all(LayerDev L, MapConsts mc) {
  const int n = L.ctr[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = L.live[i];
    const float4 a = reinterpret_cast<const float4*>(L.pool)[(size_t)slot * (kVPB / 2) + threadIdx.x];
    const int all_free = __syncthreads_and((a.y > 1e-4f && a.x == mc.trunc && a.w > 1e-4f && a.z == mc.trunc) ? 1 : 0);
    if (threadIdx.x == 0) L.block_free[slot] = all_free ? 1 : 0;
  }
}

void launch_import_index(const LayerDev& L, const int32_t* idx, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_import_index, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, s, L, idx, n);
}
void launch_import_color(const LayerDev& L, const uint8_t* rgb, const float* w, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_import_color, dim3(n < 4096 ? n : 4096), dim3(256), 0, s, L, rgb, w, n);
}
void launch_block_free_all(const LayerDev& L, const MapConsts& mc, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_block_free_all, dim3(n < 4096 ? n : 4096), dim3(256), 0, s, L, mc);
}

void launch_invert_mask(const uint8_t* in, uint8_t* out, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_invert_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

template <int FMA>
static void launch_tsdf_pass_t(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth,
                               const uint8_t* mask, float min_d, int stamp, uint8_t* flags, u64* cell_key, float decay_f, hipStream_t s) {
  const dim3 grid(grid_for(hinted(L.hint_live, L.cap), 8192));
  // (an image of one row or one column has no 2 x 2 footprint to clamp into: the branch-free voxel loop needs W, H >= 2 -- the branching
  // form, which accepts a null mask, takes such images)
  if (mask || cam.W < 2 || cam.H < 2)
    hipLaunchKernelGGL((k_tsdf_pass<4, true, false, FMA>), grid, dim3(128), 0, s, L, mc, cam, T_C_L, depth, mask, min_d, stamp, flags, cell_key, decay_f,
                       (const int*)nullptr, (const int*)nullptr);
  else
    hipLaunchKernelGGL((k_tsdf_pass<4, false, false, FMA>), grid, dim3(128), 0, s, L, mc, cam, T_C_L, depth, mask, min_d, stamp, flags, cell_key, decay_f,
                       (const int*)nullptr, (const int*)nullptr);
}
void launch_tsdf_pass(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth,
                      const uint8_t* mask, float min_d, int stamp, uint8_t* flags, u64* cell_key, float decay_f, hipStream_t s) {
  switch (arith_mode(mc.spec_flags)) {  // bit 0 mmf_params.fma_contraction, bit 1 .bilinear_four_weight_sum
    case 1: launch_tsdf_pass_t<1>(L, mc, cam, T_C_L, depth, mask, min_d, stamp, flags, cell_key, decay_f, s); break;
    case 2: launch_tsdf_pass_t<2>(L, mc, cam, T_C_L, depth, mask, min_d, stamp, flags, cell_key, decay_f, s); break;
    case 3: launch_tsdf_pass_t<3>(L, mc, cam, T_C_L, depth, mask, min_d, stamp, flags, cell_key, decay_f, s); break;
    default: launch_tsdf_pass_t<0>(L, mc, cam, T_C_L, depth, mask, min_d, stamp, flags, cell_key, decay_f, s); break;
  }
}

template <int AR>
static void launch_tsdf_pass_lazy_t(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth, int stamp,
                                    uint8_t* flags, u64* cell_key, const int* cnt, const int* list, int live, hipStream_t s) {
  if (cam.W < 2 || cam.H < 2)  // (degenerate image: the branching form, see launch_tsdf_pass_t)
    hipLaunchKernelGGL((k_tsdf_pass<4, true, true, AR>), dim3(grid_for(live, 8192)), dim3(128), 0, s, L, mc, cam, T_C_L, depth, (const uint8_t*)nullptr,
                       0.0f, stamp, flags, cell_key, 0.0f, cnt, list);
  else
    hipLaunchKernelGGL((k_tsdf_pass<4, false, true, AR>), dim3(grid_for(live, 8192)), dim3(128), 0, s, L, mc, cam, T_C_L, depth, (const uint8_t*)nullptr,
                       0.0f, stamp, flags, cell_key, 0.0f, cnt, list);
}

// the lazy form (L.epoch / L.cur_epoch / L.lag_f set): classify the live list, then pass over the work list only
void launch_tsdf_pass_lazy(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth, int stamp,
                           uint8_t* flags, u64* cell_key, int* work, int parity, hipStream_t s) {
  // work = {count of even frames, count of odd frames, list ...}: both counters zero when the buffer is made (ensure_lazy)
  const int live = hinted(L.hint_live, L.cap);
  int* cnt = work + (parity & 1);
  int* nxt = work + ((parity & 1) ^ 1);
  hipLaunchKernelGGL(k_tsdf_classify, dim3((unsigned)((live + 1023) / 1024)), dim3(1024), 0, s, L, stamp, flags, cnt, nxt, work + 2);
  switch (arith_mode(mc.spec_flags)) {
    case 1: launch_tsdf_pass_lazy_t<1>(L, mc, cam, T_C_L, depth, stamp, flags, cell_key, cnt, work + 2, live, s); break;
    case 2: launch_tsdf_pass_lazy_t<2>(L, mc, cam, T_C_L, depth, stamp, flags, cell_key, cnt, work + 2, live, s); break;
    case 3: launch_tsdf_pass_lazy_t<3>(L, mc, cam, T_C_L, depth, stamp, flags, cell_key, cnt, work + 2, live, s); break;
    default: launch_tsdf_pass_lazy_t<0>(L, mc, cam, T_C_L, depth, stamp, flags, cell_key, cnt, work + 2, live, s); break;
  }
}

void launch_lazy_catchup(const LayerDev& L, hipStream_t s) {
  hipLaunchKernelGGL(k_lazy_catchup, dim3(grid_for(hinted(L.hint_live, L.cap), 8192)), dim3(128), 0, s, L);
}

// allocation | mask columns | TSDF pass (existing blocks beside the allocation, new blocks behind it): a frame's share of k_alloc_tsdf
AllocTsdfArgs make_alloc_tsdf_args(const AllocJob& job, long long* stats, const MaskJob& M, const MapConsts& mc, const Cam& cam,
                                   const Rigid& T_C_L, const float* masked_depth, const ViewGrid& vg, uint8_t* flags_out, u64* cell_key_out,
                                   float decay_f) {
  AllocTsdfArgs A;
  A.J = job;
  A.stats = stats;
  A.M = M;
  TsdfFrameArgs& P = A.P;
  P.mc = mc;
  P.cam = cam;
  P.T_C_L = T_C_L;
  P.depth = masked_depth;
  P.grid_flags = job.sc.flags;
  P.grid_tag = job.flag_value;
  P.ox = vg.ox, P.oy = vg.oy, P.oz = vg.oz, P.nx = vg.nx, P.ny = vg.ny, P.nz = vg.nz;
  P.flags_out = flags_out;
  P.cell_key_out = cell_key_out;
  P.decay_f = decay_f;
  P.n_old = job.L.ctr + 6;
  P.pub = job.pub;
  P.tag = job.pub_tag;
  P.err = job.L.ctr + 3;
  {
    // the live count moves by a few dozen blocks per frame and the role strides over the list: 6 % headroom, not the 25 % of
    // the kernels whose work list can double from one frame to the next (idle workgroups still cost a dispatch slot each)
    const int v = job.L.hint_live ? *reinterpret_cast<const volatile int*>(job.L.hint_live) : 0;
    const int n = v > 0 ? (v + v / 16 + 16 < job.L.cap ? v + v / 16 + 16 : job.L.cap) : job.L.cap;
    P.n_pair_wgs = grid_for((n + 1) / 2, 8192);
  }
  P.n_new_wgs = kNewBlockWgs;
  P.ctl = job.pub + kPubRec + 3 * (size_t)job.L.cap;
  P.host_err = job.host_err;
  P.debug_abandon = job.debug_abandon;
  A.mask_rows = M.Hf;
  A.alloc_wgs = (job.ncells + 2047) / 2048;  // alloc_grid_multi_body<4, 2>: 2 048 cells per workgroup (<= kAllocMaxWgs)
  return A;
}

void launch_alloc_tsdf(const AllocTsdfArgs* A, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n == 1) {
    const int lead = (A[0].alloc_wgs + A[0].P.n_new_wgs + 7) & ~7;
    hipExtLaunchKernelGGL(k_alloc_tsdf, dim3(lead + A[0].P.n_pair_wgs + A[0].mask_rows), dim3(256), 0, s, ev_start, ev_stop, 0, A[0], lead);
  } else {
    const int lead = (A[0].alloc_wgs + A[1].alloc_wgs + A[0].P.n_new_wgs + A[1].P.n_new_wgs + 7) & ~7;
    const int total = lead + A[0].P.n_pair_wgs + A[1].P.n_pair_wgs + A[0].mask_rows + A[1].mask_rows;
    hipExtLaunchKernelGGL(k_alloc_tsdf2, dim3(total), dim3(256), 0, s, ev_start, ev_stop, 0, A[0], A[1], lead);
  }
}

void launch_alloc_tsdf_batch(const AllocTsdfArgs* A, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  AllocTsdfBatch P;
  P.n = n;
  int lead = 0, rest = 0;
  for (int q = 0; q < n; ++q) {
    P.a[q] = A[q];
    lead += A[q].alloc_wgs + A[q].P.n_new_wgs;
    rest += A[q].P.n_pair_wgs + A[q].mask_rows;
  }
  P.lead = (lead + 7) & ~7;
  hipExtLaunchKernelGGL(k_alloc_tsdf_batch, dim3(P.lead + rest), dim3(256), 0, s, ev_start, ev_stop, 0, P);
}

// the voxel pass of a decay alone (W *= f, dead blocks flagged in kill[]); the caller compacts
void launch_decay_mark(const LayerDev& L, const MapConsts& mc, uint8_t* kill, int* any_kill, hipStream_t s) {
  hipLaunchKernelGGL(k_decay, dim3(grid_for(hinted(L.hint_live, L.cap), 4096)), dim3(256), 0, s, L, mc, kill, any_kill);
}

void launch_decay(const LayerDev& L, const MapConsts& mc, uint8_t* kill, int* any_kill, hipStream_t s) {
  // any_kill is zero on entry (initialised at creation, reset by k_live_compact)
  hipLaunchKernelGGL(k_decay, dim3(grid_for(hinted(L.hint_live, L.cap), 4096)), dim3(256), 0, s, L, mc, kill, any_kill);
  if (mc.dealloc_decayed) hipLaunchKernelGGL(k_live_compact, dim3(1), dim3(1024), 0, s, L, kill, any_kill);
}

// Option decay_appearance_layers: W *= f on the colour / feature weights of every live block (no deallocation).
// has_w: feature layer (weights in poolw); else colour layer ({rgba, w} pairs in pool).
__global__ __launch_bounds__(256) void k_decay_app_weights(LayerDev L, float f, int has_w) {
  const int n = L.ctr[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = L.live[i];
    for (int lin = threadIdx.x; lin < kVPB; lin += 256) {
      if (has_w) {
        float* w = L.poolw + (size_t)slot * kVPB + lin;
        *w = *w * f;
      } else {
        uint2* e = reinterpret_cast<uint2*>(L.pool) + (size_t)slot * kVPB + lin;
        e->y = __float_as_uint(__uint_as_float(e->y) * f);
      }
    }
  }
}

void launch_decay_app_weights(const LayerDev& L, float f, bool has_w, hipStream_t s) {
  hipLaunchKernelGGL(k_decay_app_weights, dim3(grid_for(L.cap, 4096)), dim3(256), 0, s, L, f, has_w ? 1 : 0);
}

void launch_layer_reset(const LayerDev& L, hipStream_t s) {
  int hb = (int)((L.hmask + 1 + 255) / 256);
  if (hb > 1024) hb = 1024;
  hipLaunchKernelGGL(k_hash_clear_if, dim3(hb), dim3(256), 0, s, L, (const int*)nullptr);
  hipLaunchKernelGGL(k_reset_layer, dim3(L.dense ? 64 : 1), dim3(256), 0, s, L);
}

void launch_count_tombstones(const LayerDev& L, unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(k_count_tombstones, dim3(grid_for((int)((L.hmask + 256) / 256), 1024)), dim3(256), 0, s, L, out);
}

void launch_hash_rebuild(const LayerDev& L, hipStream_t s) {
  int hb = (int)((L.hmask + 1 + 255) / 256);
  if (hb > 1024) hb = 1024;
  hipLaunchKernelGGL(k_hash_clear_if, dim3(hb), dim3(256), 0, s, L, (const int*)nullptr);
  hipLaunchKernelGGL(k_hash_insert_live_if, dim3(grid_for((L.cap + 255) / 256, 1024)), dim3(256), 0, s, L,
                     (const int*)nullptr);
}

void launch_get_indices(const LayerDev& L, int32_t* out, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_get_indices, dim3((n + 255) / 256), dim3(256), 0, s, L, out, n);
}

void launch_gather_pool(const LayerDev& L, size_t bytes_per_block, void* out, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_pool, dim3(n < 4096 ? n : 4096), dim3(256), 0, s, L, bytes_per_block, (char*)out, n);
}

void launch_gather_poolw(const LayerDev& L, float* out, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_poolw, dim3(n < 4096 ? n : 4096), dim3(256), 0, s, L, out, n);
}

void launch_gather_color(const LayerDev& L, uint8_t* rgb, float* w, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_color, dim3(n < 4096 ? n : 4096), dim3(256), 0, s, L, rgb, w, n);
}

void launch_query_tsdf(const LayerDev& L, const MapConsts& mc, const float* pts, int n, float* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_query_tsdf, dim3((n + 255) / 256), dim3(256), 0, s, L, mc, pts, n, out);
}

void launch_query_feature(const LayerDev& L, const MapConsts& mc, const float* pts, int n, float* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_query_feature, dim3((n + 3) / 4), dim3(256), 0, s, L, mc, pts, n, out);
}

}  // namespace mmf
