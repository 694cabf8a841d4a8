// mmf_device.h -- device-side structures and arithmetic shared by all libmmfusion kernels.
//
// gfx950 (CDNA4) only: 64-wide wavefronts are assumed throughout (ballot masks are 64 bit).
// The float arithmetic below follows the operation order of the spec in DESIGN.md section 3
// exactly; the library is compiled with -ffp-contract=off so no multiply-add is fused and
// results can be compared bit-for-bit with the CPU oracle.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmf {

constexpr int kVPB = 512;  // voxels per 8x8x8 block
constexpr int kWave = 64;
typedef unsigned long long u64;
constexpr u64 kEmptyKey = ~0ull;
constexpr u64 kTombKey = ~0ull - 1ull;  // deleted entry: probes continue past it, the rebuild drops it

// One 16-byte hash entry so that a probe is a single dwordx4 load.
struct __attribute__((aligned(16))) HEntry {
  u64 key;
  int val;
  int pad;
};
constexpr int kKeyOff = 1 << 20;  // 21 bits per axis: block indices in [-2^20, 2^20)

struct Rigid {
  float R[9];
  float t[3];
};

struct Cam {
  float fx, fy, cx, cy;
  int W, H;
};

// Per-mapper constants derived from mmf_params on the host.
struct MapConsts {
  float v, bs, inv_bs, inv_v, trunc;
  float max_dist, max_weight, lin_md;
  int weighting_mode;
  float app_wm, app_max_w;
  int ws_type;
  int ws_lo[3], ws_hi[3];
  float decay_factor, decay_thr;
  int dealloc_decayed;
  float mesh_min_w;
  int st_sf, st_max_steps;
  float st_max_len, st_eps;
  int C;
  float reach;  // how far past the measured depth the raycast marks blocks: trunc (default) or 0
  int spec_flags;  // bit 0: raycast walks from the camera; bit 1: appearance blend divides per channel; bit 2: fma_contraction;
                   // bit 3: block_index_by_division; bit 4: view_truncation_band_marking; bit 5: bilinear_four_weight_sum
                   // (mmf_params spec switches)
};
constexpr int kSpecFma = 4;
constexpr int kSpecBlockDiv = 8;
constexpr int kSpecBandMark = 16;
constexpr int kSpecBilin4 = 32;
// switches only the stand-alone launches implement: frames of a mapper with any of them set are not fused, merged or pipelined
// (bilinear_four_weight_sum is NOT among them: like fma_contraction it is an arithmetic mode the un-merged fused launches are built for)
constexpr int kSpecStandalone = 2 | kSpecBlockDiv | kSpecBandMark;
constexpr int kSpecArith = kSpecFma | kSpecBilin4;  // non-default arithmetic: un-merged launches, not pipelined
// The arithmetic mode the per-voxel device functions are instantiated for (their template parameter `FMA`, an int):
//   bit 0: fma_contraction, bit 1: bilinear_four_weight_sum.  The fused kernels instantiate 0 (and 1: the un-merged FMA launches).
__host__ __device__ inline int arith_mode(int spec_flags) { return ((spec_flags & kSpecFma) ? 1 : 0) | ((spec_flags & kSpecBilin4) ? 2 : 0); }

// ---- spec switch mmf_params.fma_contraction (MapConsts::spec_flags & kSpecFma) ---------------------------------------------------
// The library is built with -ffp-contract=off: `a * b + c` is two rounded operations everywhere.  nvcc contracts by default
// (-fmad=true), so CUDA nvblox's results almost certainly contain fused multiply-adds; which ones is not knowable from here (the
// source is absent).  The switch applies the contraction a LLVM-family compiler makes to the spec's expressions as written:
//   a*b + c        -> fma(a, b, c)
//   a*x + b*y      -> fma(a, x, b*y)            (the left product is fused, the right one rounded)
//   (X + e*f) + t  -> fma(e, f, X) + t
// at the projection of a voxel centre (voxel_centre, xform, project), every bilinear sample (depth, synthetic depth, colour /
// feature taps, the low-res feature map incl. its source-index computation), the TSDF update's numerator and the appearance
// blend's numerator -- in the oracle (fmaf) and here (v_fma_f32 / v_pk_fma_f32), same tree.  FMA is a template parameter like DIV:
// the default kernels are not touched by it.
template <int FMA>
__device__ __forceinline__ float madd(float a, float b, float c) {
  if constexpr ((FMA & 1) != 0)
    return __builtin_fmaf(a, b, c);
  else
    return a * b + c;
}
// ---- scalar-register cap -------------------------------------------------------------------------------------------------------
// The CU admits floor(800 / (ceil(sgpr / 16) * 16 + 16)) waves per SIMD (MI355X_MICROARCH.md, "Residency"): the per-voxel kernels'
// natural 101-106 scalar registers mean SIX 256-thread workgroups per CU whatever the compiler's occupancy line says (7), and the
// grids of the fused frame are a little larger than 6 x 256: the surplus forms a second round (k_alloc_tsdf: 106 of its 1 608
// pair workgroups started 5-8 us late).  96 admits seven.  Applied where it measured faster (round 5, profiles/r05l_sgpr_cap.txt):
// k_front_app 18.3 -> 17.3 us, k_alloc_tsdf 15.0 -> 14.7, the hash path's k_tsdf_pass / k_front_compact_big (+2 % frames/s); NOT
// on k_front (its grid fits six per CU: 12.8 -> 12.9), the sphere tracers (+0.2 .. +0.5 us) or k_app_frame (neutral); 80 (eight per CU)
// was slower everywhere it was tried.
#define MMF_SGPR96 __attribute__((amdgpu_num_sgpr(96)))

// ---- uniform constants in VECTOR registers ---------------------------------------------------------------------------------------
// The per-voxel kernels keep ~100 launch constants live (transform, intrinsics, map constants, a dozen layer pointers) beside the
// exec masks of their nested early-outs: the scalar file (<= 102) overflows and the allocator spills scalars into lanes of a vector
// register -- v_writelane / v_readlane, VALU instructions on the busiest port (12 % of k_tsdf_pass's).  A constant moved to a vector
// register once costs one VGPR and no instruction per use; the float operations are the same (round 5: k_tsdf_pass 125 -> 120 us).
__device__ __forceinline__ float vgpr(float x) {
  float y;
  asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "s"(x));
  return y;
}
__device__ __forceinline__ void rigid_to_vgprs(Rigid& T) {
#pragma unroll
  for (int q = 0; q < 9; ++q) T.R[q] = vgpr(T.R[q]);
#pragma unroll
  for (int q = 0; q < 3; ++q) T.t[q] = vgpr(T.t[q]);
}
__device__ __forceinline__ void cam_to_vgprs(Cam& c) {
  c.fx = vgpr(c.fx);
  c.fy = vgpr(c.fy);
  c.cx = vgpr(c.cx);
  c.cy = vgpr(c.cy);
}
// the map constants of the projective update (voxel / block size, truncation, distance and weight limits, depth-interpolation limit)
__device__ __forceinline__ void update_consts_to_vgprs(MapConsts& mc) {
  mc.v = vgpr(mc.v);
  mc.bs = vgpr(mc.bs);
  mc.trunc = vgpr(mc.trunc);
  mc.max_dist = vgpr(mc.max_dist);
  mc.max_weight = vgpr(mc.max_weight);
  mc.lin_md = vgpr(mc.lin_md);
}

// a*x + b*y
template <int FMA>
__device__ __forceinline__ float madd2(float a, float x, float b, float y) {
  if constexpr ((FMA & 1) != 0)
    return __builtin_fmaf(a, x, b * y);
  else
    return a * x + b * y;
}

// Device view of one block layer: open-addressing hash (packed 64-bit keys -> pool slot),
// pool of 8x8x8 blocks, live list (allocation order) and free-slot stack.
struct LayerDev {
  HEntry* htab;    // [hmask+1] {key, pool slot}; key == kEmptyKey when free
  unsigned hmask;  // table size - 1 (power of two)
  u64* slot_key;   // [cap] key stored in each pool slot
  int* live;       // [cap] pool slots in allocation order
  int* free_stack; // [cap]
  int* ctr;        // [0] n_live  [1] n_free  [2] bump (first never-used slot)  [3] error flags  [4] tombstones  [5] hash rebuilds  [6] live count snapshot of k_front
  char* pool;      // payload A: cap * block_bytes
  float* poolw;    // payload B: cap * 512 floats (feature layer weights) or nullptr
  int cap;
  // Optional dense block table of a bounded workspace (TSDF layer only): dense[cell] = slot + 1, 0 = none.
  // It mirrors the hash (maintained by the same insert / erase sites) and exists so that latency-bound
  // kernels (sphere tracing) can stage the whole index in LDS.
  unsigned short* dense;
  // TSDF layer only: block_free[slot] = 1 iff every voxel of the block is observed free space (W > 1e-4 and
  // D == +trunc).  Recomputed by k_tsdf_integrate / k_decay for every block they touch; lets the sphere tracer
  // step through such blocks without reading voxels (the sample result is known: valid, distance = trunc).
  unsigned char* block_free;
  // TSDF layer only: wmax[slot] = largest voxel weight of the block, written by k_tsdf_pass for every live block.  A pending
  // Mapper.decay() then needs no voxel pass to find the blocks it deallocates (all W f < threshold <=> max(W) f < threshold,
  // exactly: multiplication by f > 0 is monotone) and the multiplication itself rides in the next k_tsdf_pass.
  float* wmax;
  // TSDF layer only: stamp[slot] = (frame stamp << 1) | is_new, written by the allocation job of a fused frame for every
  // candidate block, so that a pass over the LIVE list (k_tsdf_pass) knows which blocks this frame integrates.
  int* stamp;
  int* hint_live;  // pinned host int (may be null): last live-block count, read by the host to size later grids
  int d_lo[3];
  int d_ny, d_nz, d_ncells;
  // TSDF layer only: LAZY DECAY of large hash-indexed maps.  A decay multiplies every voxel weight by f; in a
  // map of 10^5 live blocks of which a frame touches a fifth, that is the frame's dominant traffic.  Instead a block's voxels
  // stay `cur_epoch - epoch[slot]` decays behind until the block is next integrated (the TSDF pass applies the missing
  // multiplications first, one by one: the same float operations as the eager decays) or read (k_lazy_catchup before any
  // consumer of voxel weights; the sphere tracer applies them to the weight it samples).  The per-block summaries are kept
  // CURRENT by the decay's list compaction, which touches one word per live block anyway: wmax (deallocation: all W f < thr
  // <=> max W f < thr), wmin and block_free (all-free needs min W > 1e-4); multiplication by f > 0 is monotone, so max / min
  // commute with it exactly.  epoch == nullptr: the layer is not in lazy mode (every kernel then behaves as before).
  int* epoch;            // [cap] the decay epoch the block's voxel weights are current to
  float* wmin;           // [cap] smallest voxel weight of the block, current
  unsigned char* band;   // [cap] 1 iff a voxel had W > 0 and |D| < trunc when the block was last written (appearance candidates)
  int cur_epoch;         // decays applied lazily so far
  float lag_f;           // their factor
};

__device__ inline int dense_cell(const LayerDev& L, int x, int y, int z) {
  // 24-bit multiplies: full rate (a 32-bit v_mul_lo is quarter rate); the offsets and extents of a dense table are far below 2^23
  return __mul24(__mul24(x - L.d_lo[0], L.d_ny) + (y - L.d_lo[1]), L.d_nz) + (z - L.d_lo[2]);
}
__device__ inline void dense_set(const LayerDev& L, unsigned long long key, int slot_plus_1);


// Compaction scratch shared by the three "flag -> ordered candidate list" passes.
struct Scratch {
  uint8_t* flags;    // [ncells] (multiple of 4) -- all zero between calls
  int* cell_slot;    // [ncells]
  u64* cell_key;     // [ncells] key of a flagged cell when the cells are list positions (KeySrc mode 1)
  int2* tile_counts; // [ntiles]
  int2* tile_offs;   // [ntiles]
  int* cand_slot;    // [ncells]
  u64* cand_key;     // [ncells]
  uint8_t* cand_new; // [ncells]
  int* cand_count;   // [1]
  int* alloc_ctx;    // [4] old n_live, old n_free, old bump, granted
  int* hint_cand;    // pinned host int (may be null): last candidate count, read by the host to size later grids
  u64* lb;           // [2 (ncells / 1024 + 2) + 8] counter / count / prefix words of the scalable allocation (alloc_big_body)
};

// Where the key of a compaction cell comes from.
struct KeySrc {
  int mode;  // 0: dense view grid (cell -> block index), 1: live list of another layer
  int ox, oy, oz, ny, nz;
  const u64* slot_key;
  const int* live;
  const int* n_live;
};

// One compaction + allocation job (flags -> ordered candidate list, new blocks inserted) for one layer.
constexpr int kNewBlockWgs = 32;  // waiter workgroups of k_alloc_tsdf (new blocks of the frame)
constexpr int kPubRec = 16;       // AllocJob::pub: [0] total, [1 .. 8] per-workgroup counts, [kPubRec ..] new-block records
constexpr int kAllocMaxWgs = 8;   // kFusedAllocMaxCells / 2 048

struct AllocJob {
  LayerDev L;
  KeySrc ks;
  Scratch sc;
  int ncells;
  int stat_upd, stat_new;  // indices into the mapper's statistics array (-1: none)
  int stamp = 0;           // != 0: mark every candidate slot in L.stamp (see LayerDev::stamp)
  uint8_t* kill = nullptr; // != null: first apply the kill flags of a decay pass (live_compact_body)
  int* any_kill = nullptr;
  int* zero_me = nullptr;  // != null: counters this job resets: zero_me[k * zero_stride], k < zero_n (the frame's survivor sub-lists)
  int zero_n = 0, zero_stride = 0;
  long long* timeline = nullptr;  // != null: thread 0 stores wall_clock64() (100 MHz) at 6 points of the job (diagnostics)
  u64* pub = nullptr;      // alloc_grid_multi_body: [16 + 3 * cap + 2 + kNewBlockWgs] published counts, new blocks (see there) and
                           // the control words of k_alloc_tsdf's hand-over (TsdfFrameArgs::ctl)
  unsigned pub_tag = 0;
  int* host_err = nullptr; // pinned host int: set when the in-launch hand-over failed for good
  int debug_abandon = 0;   // test hook, see TsdfFrameArgs
  int flag_value = 1;      // alloc_grid_multi_body: a grid cell is flagged iff its byte equals this (the frame's grid tag)
  unsigned lb_tag = 0;     // alloc_big_body: tag of this launch's look-back words (22 bits, never 0)
};

__host__ __device__ inline u64 pack_key(int x, int y, int z) {
  return ((u64)(unsigned)(x + kKeyOff) << 42) | ((u64)(unsigned)(y + kKeyOff) << 21) | (u64)(unsigned)(z + kKeyOff);
}
__host__ __device__ inline void unpack_key(u64 k, int& x, int& y, int& z) {
  x = (int)((k >> 42) & 0x1fffffu) - kKeyOff;
  y = (int)((k >> 21) & 0x1fffffu) - kKeyOff;
  z = (int)(k & 0x1fffffu) - kKeyOff;
}

__device__ inline void dense_set(const LayerDev& L, unsigned long long key, int slot_plus_1) {
  if (!L.dense) return;
  int x, y, z;
  unpack_key(key, x, y, z);
  L.dense[dense_cell(L, x, y, z)] = (unsigned short)slot_plus_1;
}

__device__ inline unsigned hash_key(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  return (unsigned)k;
}

__device__ inline uint4 hash_load(const LayerDev& L, unsigned h) {
  return *reinterpret_cast<const uint4*>(&L.htab[h]);
}
__device__ inline u64 entry_key(const uint4& e) { return ((u64)e.y << 32) | (u64)e.x; }

// Continue a probe sequence whose first entry `e` (at position h) has already been loaded.
__device__ inline int hash_resolve(const LayerDev& L, u64 key, unsigned h, uint4 e) {
  for (unsigned probe = 0; probe <= L.hmask; ++probe) {
    const u64 k = entry_key(e);
    if (k == key) return (int)e.z;
    if (k == kEmptyKey) return -1;
    h = (h + 1) & L.hmask;
    e = hash_load(L, h);
  }
  return -1;
}

__device__ inline int hash_find(const LayerDev& L, u64 key) {
  const unsigned h = hash_key(key) & L.hmask;
  return hash_resolve(L, key, h, hash_load(L, h));
}

// Insert a key known to be absent. Distinct keys may race for a cell: CAS on the 64-bit key.
// Layers with a dense block table (bounded workspace) do not keep the hash at all: the table is the index.
// The first EMPTY or TOMBSTONE entry of the probe sequence is claimed.  Reusing tombstones matters: a block that leaves and
// re-enters the view (the frustum's edge, every orbit) would otherwise leave one more tombstone IN ITS OWN CHAIN per cycle, and
// the chains of the few thousand keys that cycle grew to dozens of dependent probes between two rebuilds (the unbounded bench
// stream: allocation 31 -> 198 us, deallocation 23 -> 240 us as the tombstones went 0 -> 120 k).  Safe beside concurrent lookups
// of OTHER keys: an entry goes tombstone -> key, both of which a probe for another key walks past.
// Returns 1 when the claimed entry was a tombstone: the caller takes it off the layer's tombstone count (ctr[4]; one aggregated
// atomic per thread that reused any), so that the amortised rebuild is triggered by the tombstones that ARE in the table.
__device__ inline int hash_insert(const LayerDev& L, u64 key, int slot) {
  if (L.dense) return 0;
  unsigned h = hash_key(key) & L.hmask;
  for (unsigned probe = 0; probe <= L.hmask; ++probe) {
    u64 prev = atomicCAS(&L.htab[h].key, kEmptyKey, key);
    int reused = 0;
    if (prev == kTombKey) {
      reused = atomicCAS(&L.htab[h].key, kTombKey, key) == kTombKey ? 1 : 0;
      prev = reused ? kEmptyKey : 0ull;
    }
    if (prev == kEmptyKey) {
      L.htab[h].val = slot;
      return reused;
    }
    h = (h + 1) & L.hmask;
  }
  return 0;
}

// Mark the entry of `key` deleted (no-op if absent).
__device__ inline void hash_erase(const LayerDev& L, u64 key) {
  if (L.dense) return;
  unsigned h = hash_key(key) & L.hmask;
  for (unsigned probe = 0; probe <= L.hmask; ++probe) {
    const u64 k = L.htab[h].key;
    if (k == key) {
      L.htab[h].key = kTombKey;
      return;
    }
    if (k == kEmptyKey) return;
    h = (h + 1) & L.hmask;
  }
}

// Pool slot of a block key (-1 if absent): dense table when the layer has one, hash otherwise.
__device__ inline int layer_lookup(const LayerDev& L, u64 key) {
  if (L.dense) {  // blocks only ever exist inside the workspace bounds the table covers
    int x, y, z;
    unpack_key(key, x, y, z);
    const int dx = x - L.d_lo[0], dy = y - L.d_lo[1], dz = z - L.d_lo[2];
    if (dx < 0 || (unsigned)dy >= (unsigned)L.d_ny || (unsigned)dz >= (unsigned)L.d_nz) return -1;
    const int cell = (dx * L.d_ny + dy) * L.d_nz + dz;
    if (cell >= L.d_ncells) return -1;
    return (int)L.dense[cell] - 1;
  }
  return hash_find(L, key);
}

__device__ inline int ifloor(float x) { return (int)floorf(x); }

template <int FMA = 0>
__device__ inline void xform(const Rigid& T, const float* p, float* q) {
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = madd<FMA>(T.R[i * 3 + 2], p[2], madd2<FMA>(T.R[i * 3 + 0], p[0], T.R[i * 3 + 1], p[1])) + T.t[i];
}
__device__ inline void rotate(const Rigid& T, const float* p, float* q) {
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = (T.R[i * 3 + 0] * p[0] + T.R[i * 3 + 1] * p[1]) + T.R[i * 3 + 2] * p[2];
}

template <int FMA = 0>
__device__ inline void voxel_centre(const MapConsts& mc, int bx, int by, int bz, int lin, float* c) {
  int vx = lin >> 6, vy = (lin >> 3) & 7, vz = lin & 7;
  c[0] = madd2<FMA>((float)bx, mc.bs, (float)vx + 0.5f, mc.v);
  c[1] = madd2<FMA>((float)by, mc.bs, (float)vy + 0.5f, mc.v);
  c[2] = madd2<FMA>((float)bz, mc.bs, (float)vz + 0.5f, mc.v);
}

template <int FMA = 0>
__device__ inline bool project(const Cam& c, const float* p, float& u, float& v) {
  if (p[2] <= 1e-6f) return false;
  float iz = 1.0f / p[2];
  float uu = madd<FMA>(c.fx, p[0] * iz, c.cx);
  float vv = madd<FMA>(c.fy, p[1] * iz, c.cy);
  if (uu < 0.0f || vv < 0.0f || uu > (float)c.W || vv > (float)c.H) return false;
  u = uu;
  v = vv;
  return true;
}

__device__ inline bool bilin_setup(float u, float v, int W, int H, int& x0, int& y0, float& wx, float& wy) {
  float uc = u - 0.5f, vc = v - 0.5f;
  float fx0 = floorf(uc), fy0 = floorf(vc);
  int ix = (int)fx0, iy = (int)fy0;
  if (ix < 0 || iy < 0 || ix + 1 > W - 1 || iy + 1 > H - 1) return false;
  x0 = ix;
  y0 = iy;
  wx = uc - fx0;
  wy = vc - fy0;
  return true;
}

// (FMA & 2: mmf_params.bilinear_four_weight_sum -- four weighted taps in upstream's order of terms, oracle/mmf_oracle.c bilin_c)
template <int FMA = 0>
__device__ inline float bilin(float a00, float a10, float a01, float a11, float wx, float wy) {
  if constexpr ((FMA & 2) != 0) {
    const float w00 = (1.0f - wx) * (1.0f - wy), w01 = (1.0f - wx) * wy, w10 = wx * (1.0f - wy), w11 = wx * wy;
    float t = madd2<FMA>(w00, a00, w01, a01);
    t = madd<FMA>(w10, a10, t);
    return madd<FMA>(w11, a11, t);
  }
  float top = madd2<FMA>(1.0f - wx, a00, wx, a10);
  float bot = madd2<FMA>(1.0f - wx, a01, wx, a11);
  return madd2<FMA>(1.0f - wy, top, wy, bot);
}

// Measurement weight of a TSDF update: upstream nvblox's WeightingFunctionType family as restated in oracle/mmf_oracle.c
// (tsdf_weight: same operations in the same order).  Mode 1 (1/d^2) is this spec's default.
__device__ inline float tsdf_measurement_weight(const MapConsts& mc, float d, float sdf) {
  const int mode = mc.weighting_mode;
  if (mode == 1) return 1.0f / (d * d);
  if (mode == 0) return 1.0f;
  const float trunc = mc.trunc;
  if (mode == 2) return sdf >= 0.0f ? 1.0f : fmaxf((trunc + sdf) / trunc, 0.0f);
  if (mode == 3) return (1.0f / (d * d)) * (sdf >= 0.0f ? 1.0f : fmaxf((trunc + sdf) / trunc, 0.0f));
  if (mode == 4) return (1.0f / (d * d)) * (1.0f - 0.5f * (fminf(fabsf(sdf), trunc) / trunc));
  return fminf(1.0f / d, 1.0f);
}

__device__ inline bool in_workspace(const MapConsts& mc, int x, int y, int z) {
  if (mc.ws_type == 0) return true;
  if (z < mc.ws_lo[2] || z > mc.ws_hi[2]) return false;
  if (mc.ws_type == 1) return true;
  if (x < mc.ws_lo[0] || x > mc.ws_hi[0]) return false;
  if (y < mc.ws_lo[1] || y > mc.ws_hi[1]) return false;
  return true;
}

// Voxel containing p: block floor(p*inv_bs), voxel clamp(floor((p - b*bs)*inv_v), 0, 7)  (mmf_params.block_index_by_division: by
// division instead -- a uniform branch; the callers are the mesh's / the queries' look-ups, not the per-frame kernels).
__device__ inline u64 voxel_at(const MapConsts& mc, const float* p, int& lin) {
  int b[3], vi[3];
  const bool bdiv = (mc.spec_flags & kSpecBlockDiv) != 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    b[a] = ifloor(bdiv ? p[a] / mc.bs : p[a] * mc.inv_bs);
    const float r = p[a] - (float)b[a] * mc.bs;
    int q = ifloor(bdiv ? r / mc.v : r * mc.inv_v);
    vi[a] = q < 0 ? 0 : (q > 7 ? 7 : q);
  }
  lin = (vi[0] * 8 + vi[1]) * 8 + vi[2];
  return pack_key(b[0], b[1], b[2]);
}

// ---- wave / workgroup scans (wave64) ---------------------------------------------------------
// Six data-parallel-primitive adds: shifts by 1, 2, 4, 8 inside the 16-lane rows (lanes without a source add 0), then the
// total of row 0 / 2 into rows 1 / 3 and the total of rows 0 - 1 into rows 2 - 3.  (__shfl_up goes through ds_bpermute: six of
// them in a chain are ~0.6 us, and the allocation chain of a frame runs several scans.)
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_shifted(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
__device__ inline int wave_incl_scan(int v) {
  v += dpp_shifted<0x111, 0xf>(v);  // row_shr:1
  v += dpp_shifted<0x112, 0xf>(v);  // row_shr:2
  v += dpp_shifted<0x114, 0xf>(v);  // row_shr:4
  v += dpp_shifted<0x118, 0xf>(v);  // row_shr:8
  v += dpp_shifted<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_shifted<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
// maximum of a non-negative float over the wave, in every lane
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_max_f32_step(float v) {
  const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return fmaxf(v, __builtin_bit_cast(float, o));
}
__device__ inline float wave_max_f32(float v) {
  v = dpp_max_f32_step<0xB1, 0xf>(v);   // quad_perm [1, 0, 3, 2]
  v = dpp_max_f32_step<0x4E, 0xf>(v);   // quad_perm [2, 3, 0, 1]
  v = dpp_max_f32_step<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_max_f32_step<0x140, 0xf>(v);  // row_mirror
  v = dpp_max_f32_step<0x142, 0xa>(v);  // row_bcast:15
  v = dpp_max_f32_step<0x143, 0xc>(v);  // row_bcast:31: lane 63 holds the maximum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Exclusive scan of two ints over a workgroup of NW waves; `lds` holds 2*NW+2 ints.
template <int NW>
__device__ inline void block_excl_scan2(int a, int b, int* lds, int& ex_a, int& ex_b, int& tot_a, int& tot_b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ia = wave_incl_scan(a), ib = wave_incl_scan(b);
  if (lane == 63) {
    lds[wave] = ia;
    lds[NW + wave] = ib;
  }
  __syncthreads();
  int ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    int xa = lds[w], xb = lds[NW + w];
    if (w < wave) {
      ba += xa;
      bb += xb;
    }
    ta += xa;
    tb += xb;
  }
  ex_a = ba + ia - a;
  ex_b = bb + ib - b;
  tot_a = ta;
  tot_b = tb;
  __syncthreads();
}

// XCD-aware candidate mapping: workgroup j handles candidate (j%8)*chunk + j/8 so that each of the
// 8 XCDs (block j runs on XCD j%8) walks one contiguous, spatially coherent eighth of the sorted
// candidate list and keeps its image footprint in its own L2.
__device__ inline int xcd_candidate(int j, int chunk) { return (j & 7) * chunk + (j >> 3); }

}  // namespace mmf
