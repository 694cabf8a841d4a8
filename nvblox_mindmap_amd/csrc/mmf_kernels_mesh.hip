// mmf_kernels_mesh.hip -- surface-vertex extraction (the welded marching-cubes vertex set) and the
// per-vertex feature gather that feeds diffuser_actor.  gfx950 / wave64.
//
// Replaces the CUDA behind nvblox_torch Mapper.update_feature_mesh / get_feature_mesh().vertices() /
// .vertex_features(), reached by the reference at mindmap/mapping/helpers/nvblox_output_helpers.py:49-52.
//
// One workgroup (256 threads) per live TSDF block.  The 9x9x9 lattice of (distance, valid) of the block
// and its seven +x/+y/+z neighbours is staged in LDS (each lattice value is used by up to 6 edges and
// 8 cubes).  The 3*729 lattice edges are visited in canonical order (lattice point lexicographic, axis
// 0,1,2), 9 consecutive edges per thread, so a workgroup prefix sum gives every vertex its final,
// deterministic output position: count pass -> scan over blocks -> emit pass.
#include "mmf_launch.h"

#define MMF_MC_QUAL static __device__ const
#include "../../include/mmf_mc_table.h"  // generated marching-cubes table (tools/gen_mc_table.py), shared with the oracle

namespace mmf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kLat = 729;
constexpr int kEdges = 3 * kLat;   // 2187
constexpr int kEdgesPerThread = 9; // 243 active threads

struct MeshLds {
  float D[kLat];
  uint8_t V[kLat];
  uint8_t CV[kVPB];
  int nslot[8];
  int scan[10];
};

__device__ inline void load_lattice(const LayerDev& T, const MapConsts& mc, int slot, int bx, int by, int bz, MeshLds& m) {
  const int tid = threadIdx.x;
  if (tid < 8) {
    const int dx = tid >> 2, dy = (tid >> 1) & 1, dz = tid & 1;
    m.nslot[tid] = tid == 0 ? slot : layer_lookup(T, pack_key(bx + dx, by + dy, bz + dz));
  }
  __syncthreads();
  for (int q = tid; q < kLat; q += 256) {
    const int x = q / 81, y = (q / 9) % 9, z = q % 9;
    const int s = m.nslot[((x >> 3) << 2) | ((y >> 3) << 1) | (z >> 3)];
    const int lin = ((x & 7) * 8 + (y & 7)) * 8 + (z & 7);
    float2 dw = make_float2(0.0f, 0.0f);
    if (s >= 0) dw = reinterpret_cast<const float2*>(T.pool)[(size_t)s * kVPB + lin];
    const bool valid = s >= 0 && dw.y >= mc.mesh_min_w;
    m.D[q] = valid ? dw.x : 0.0f;
    m.V[q] = valid ? 1 : 0;
  }
  __syncthreads();
  for (int c = tid; c < kVPB; c += 256) {
    const int x = c >> 6, y = (c >> 3) & 7, z = c & 7;
    const int q = x * 81 + y * 9 + z;
    m.CV[c] = m.V[q] & m.V[q + 81] & m.V[q + 9] & m.V[q + 90] & m.V[q + 1] & m.V[q + 82] & m.V[q + 10] & m.V[q + 91];
  }
  __syncthreads();
}

// does lattice edge e (canonical id) carry a vertex?  Returns the interpolation parameter in t.
__device__ inline bool edge_vertex(const MeshLds& m, int e, int& qx, int& qy, int& qz, int& a, float& t) {
  const int q = e / 3;
  a = e - 3 * q;
  qx = q / 81;
  qy = (q / 9) % 9;
  qz = q % 9;
  const int qa = a == 0 ? qx : (a == 1 ? qy : qz);
  if (qa >= 8) return false;
  const int r = q + (a == 0 ? 81 : (a == 1 ? 9 : 1));
  const float Da = m.D[q], Db = m.D[r];
  if ((Da < 0.0f) == (Db < 0.0f)) return false;
  // cubes containing the edge: origin o with o_a = q_a and o_b in {q_b - 1, q_b} on the two other axes
  // (ca, sa): coordinate / cube-array stride along a; (cb, sb), (cc, sc): the axes (a+1)%3, (a+2)%3
  int ca, cb, cc, sa, sb, sc;
  if (a == 0) {
    ca = qx; cb = qy; cc = qz; sa = 64; sb = 8; sc = 1;
  } else if (a == 1) {
    ca = qy; cb = qz; cc = qx; sa = 8; sb = 1; sc = 64;
  } else {
    ca = qz; cb = qx; cc = qy; sa = 1; sb = 64; sc = 8;
  }
  bool found = false;
#pragma unroll
  for (int s1 = -1; s1 <= 0; ++s1)
#pragma unroll
    for (int s2 = -1; s2 <= 0; ++s2) {
      const int ob = cb + s1, oc = cc + s2;
      if (ob < 0 || oc < 0 || ob > 7 || oc > 7) continue;  // ca <= 7 already
      if (m.CV[ca * sa + ob * sb + oc * sc]) found = true;
    }
  if (!found) return false;
  t = Da / (Da - Db);
  return true;
}

__global__ __launch_bounds__(256) void k_mesh_count(LayerDev T, MapConsts mc, int* __restrict__ counts) {
  __shared__ MeshLds m;
  const int n = T.ctr[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = T.live[i];
    int bx, by, bz;
    unpack_key(T.slot_key[slot], bx, by, bz);
    load_lattice(T, mc, slot, bx, by, bz, m);
    int cnt = 0;
    const int e0 = threadIdx.x * kEdgesPerThread;
    for (int k = 0; k < kEdgesPerThread; ++k) {
      const int e = e0 + k;
      if (e >= kEdges) break;
      int qx, qy, qz, a;
      float t;
      cnt += edge_vertex(m, e, qx, qy, qz, a, t) ? 1 : 0;
    }
    int ea, eb, ta, tb;
    block_excl_scan2<4>(cnt, 0, m.scan, ea, eb, ta, tb);
    if (threadIdx.x == 0) counts[i] = ta;
    __syncthreads();
  }
}

// single workgroup: offsets[i] = exclusive prefix of counts; out2[0] = total vertices, out2[1] = n_live
__global__ __launch_bounds__(256) void k_mesh_scan(LayerDev T, const int* __restrict__ counts, int* __restrict__ offsets,
                                                  int* __restrict__ out2) {
  __shared__ int lds[10];
  __shared__ int carry;
  const int n = T.ctr[0];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    const int c = i < n ? counts[i] : 0;
    int ea, eb, ta, tb;
    block_excl_scan2<4>(c, 0, lds, ea, eb, ta, tb);
    if (i < n) offsets[i] = carry + ea;
    __syncthreads();
    if (threadIdx.x == 0) carry += ta;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = carry;
    out2[1] = n;
  }
}

__global__ __launch_bounds__(256) void k_mesh_emit(LayerDev T, LayerDev F, MapConsts mc, const int* __restrict__ offsets,
                                                  int n_blocks, float* __restrict__ verts, __half* __restrict__ vfeat, int V) {
  __shared__ MeshLds m;
  __shared__ int s_vox[kEdges];  // feature voxel (slot*512 + lin) of every vertex of the block, -1 = none
  const int C = mc.C, nch = C >> 3;
  const int group = threadIdx.x >> 3, gl = threadIdx.x & 7;
  for (int i = blockIdx.x; i < n_blocks; i += gridDim.x) {
    const int slot = T.live[i];
    int bx, by, bz;
    unpack_key(T.slot_key[slot], bx, by, bz);
    load_lattice(T, mc, slot, bx, by, bz, m);
    // pass 1: count (same walk as k_mesh_count) to get this thread's first vertex id
    int cnt = 0;
    const int e0 = threadIdx.x * kEdgesPerThread;
    for (int k = 0; k < kEdgesPerThread; ++k) {
      const int e = e0 + k;
      if (e >= kEdges) break;
      int qx, qy, qz, a;
      float t;
      cnt += edge_vertex(m, e, qx, qy, qz, a, t) ? 1 : 0;
    }
    int ea, eb, nv, tb;
    block_excl_scan2<4>(cnt, 0, m.scan, ea, eb, nv, tb);
    const int off = offsets[i];
    // pass 2: emit positions, look up the feature voxel of every vertex
    int local = ea;
    if (cnt > 0) {
      for (int k = 0; k < kEdgesPerThread; ++k) {
        const int e = e0 + k;
        if (e >= kEdges) break;
        int qx, qy, qz, a;
        float t;
        if (!edge_vertex(m, e, qx, qy, qz, a, t)) continue;
        float pos[3] = {(float)bx * mc.bs + ((float)qx + 0.5f) * mc.v, (float)by * mc.bs + ((float)qy + 0.5f) * mc.v,
                        (float)bz * mc.bs + ((float)qz + 0.5f) * mc.v};
        const int qa = a == 0 ? qx : (a == 1 ? qy : qz);
        const int ba = a == 0 ? bx : (a == 1 ? by : bz);
        const float pa = a == 0 ? pos[0] : (a == 1 ? pos[1] : pos[2]);
        const float pb = (float)ba * mc.bs + ((float)(qa + 1) + 0.5f) * mc.v;
        const float pn = pa + t * (pb - pa);
        if (a == 0) pos[0] = pn;
        else if (a == 1) pos[1] = pn;
        else pos[2] = pn;
        const int vid = off + local;
        if (vid < V) {
          verts[3 * (size_t)vid] = pos[0];
          verts[3 * (size_t)vid + 1] = pos[1];
          verts[3 * (size_t)vid + 2] = pos[2];
        }
        int lin;
        const u64 key = voxel_at(mc, pos, lin);
        int fv = -1;
        if (F.pool) {
          const int fs = layer_lookup(F, key);
          if (fs >= 0 && F.poolw[(size_t)fs * kVPB + lin] > 0.0f) fv = fs * kVPB + lin;
        }
        s_vox[local] = fv;
        local++;
      }
    }
    __syncthreads();
    // pass 3: copy feature rows, 8 lanes x 16 B per 128-byte piece
    half8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
    for (int vi = group; vi < nv; vi += 32) {
      const int vid = off + vi;
      if (vid >= V) continue;
      const int fv = s_vox[vi];
      __half* dst = vfeat + (size_t)vid * C;
      if (fv >= 0) {
        const __half* src = reinterpret_cast<const __half*>(F.pool) + (size_t)fv * C;
        for (int ch = gl; ch < nch; ch += 8) *reinterpret_cast<half8*>(dst + ch * 8) = *reinterpret_cast<const half8*>(src + ch * 8);
      } else {
        for (int ch = gl; ch < nch; ch += 8) *reinterpret_cast<half8*>(dst + ch * 8) = z;
      }
    }
    __syncthreads();
  }
}

void launch_mesh_count(const LayerDev& tsdf, const MapConsts& mc, int* counts, int* offsets, int* out2, hipStream_t s) {
  int g = hinted(tsdf.hint_live, tsdf.cap);
  g = g < 8192 ? g : 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_mesh_count, dim3(g), dim3(256), 0, s, tsdf, mc, counts);
  hipLaunchKernelGGL(k_mesh_scan, dim3(1), dim3(256), 0, s, tsdf, (const int*)counts, offsets, out2);
}

void launch_mesh_emit(const LayerDev& tsdf, const LayerDev& feat, const MapConsts& mc, const int* offsets, int n_blocks,
                      float* verts, __half* vfeat, int V, hipStream_t s) {
  if (n_blocks <= 0 || V <= 0) return;
  int g = n_blocks < 8192 ? n_blocks : 8192;
  hipLaunchKernelGGL(k_mesh_emit, dim3(g), dim3(256), 0, s, tsdf, feat, mc, offsets, n_blocks, verts, vfeat, V);
}

// ------------------------------------------------------------------------------------------------
// Map -> model-input tensors in two launches (mmf_model_inputs_prepare / _gather): what the reference does as
// update_feature_mesh + get_feature_mesh + three boolean-mask copies of the [V, C_pad] feature matrix + the final row
// selection (mindmap/mapping/helpers/nvblox_output_helpers.py:49-91) -- without materialising a feature row that is not kept.
//
// k_mesh_keep (workgroup per live TSDF block, lattice staged once): the block's vertices in canonical order -> strict AABB
// test (:57-60) -> feature voxel of each vertex -> "row has a non-zero used channel" test (:68-74; 8 lanes per row, 16 B
// pieces, stops at the first non-zero piece: one 128 B line per vertex in practice) -> the kept vertices {x, y, z, feature
// voxel} are appended, in order, to a chunk of the list reserved with ONE atomic per block; counts[i] / chunk[i] record it.
// The atomic counter's final value is the number of kept rows -- the only number the host needs (for its RNG draw).
// k_model_inputs_gather: rank r of the kept rows (ranks follow block order, then canonical vertex order: the order of the
// reference's filtered mesh) is found by a search in the prefix sums of counts (LDS) and its row written as f32 / f16.
// ------------------------------------------------------------------------------------------------
struct KeepArgs {
  float lo[3], hi[3];
  int used;         // leading feature channels the model uses (C - num_excess_features)
  int remove_zero;  // drop vertices whose used channels are all zero (incl. vertices without a feature voxel)
};

__global__ __launch_bounds__(256) void k_mesh_keep(LayerDev T, LayerDev F, MapConsts mc, KeepArgs ka, int* __restrict__ counts,
                                                  int* __restrict__ chunk, int* __restrict__ total, uint4* __restrict__ list,
                                                  int list_cap) {
  __shared__ MeshLds m;
  __shared__ int s_vox[kEdges];        // feature voxel of vertex k of the block; -1 none, -2 vertex dropped
  __shared__ float s_pos[3 * kEdges];
  __shared__ int s_base;
  const int n = T.ctr[0];
  const int C = mc.C;
  const int group = threadIdx.x >> 3, gl = threadIdx.x & 7, lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) total[1] = n;  // the host reads {kept rows, live blocks} in one copy
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = T.live[i];
    int bx, by, bz;
    unpack_key(T.slot_key[slot], bx, by, bz);
    load_lattice(T, mc, slot, bx, by, bz, m);
    int cnt = 0;
    const int e0 = threadIdx.x * kEdgesPerThread;
    for (int k = 0; k < kEdgesPerThread; ++k) {
      const int e = e0 + k;
      if (e >= kEdges) break;
      int qx, qy, qz, a;
      float t;
      cnt += edge_vertex(m, e, qx, qy, qz, a, t) ? 1 : 0;
    }
    int ea, eb, nv, tb;
    block_excl_scan2<4>(cnt, 0, m.scan, ea, eb, nv, tb);
    if (nv == 0) {  // (uniform) most live blocks are free space
      if (threadIdx.x == 0) {
        counts[i] = 0;
        chunk[i] = 0;
      }
      continue;
    }
    if (cnt > 0) {
      int local = ea;
      for (int k = 0; k < kEdgesPerThread; ++k) {
        const int e = e0 + k;
        if (e >= kEdges) break;
        int qx, qy, qz, a;
        float t;
        if (!edge_vertex(m, e, qx, qy, qz, a, t)) continue;
        // same position arithmetic as k_mesh_emit (the vertices of get_feature_mesh, bit for bit)
        float pos[3] = {(float)bx * mc.bs + ((float)qx + 0.5f) * mc.v, (float)by * mc.bs + ((float)qy + 0.5f) * mc.v,
                        (float)bz * mc.bs + ((float)qz + 0.5f) * mc.v};
        const int qa = a == 0 ? qx : (a == 1 ? qy : qz);
        const int ba = a == 0 ? bx : (a == 1 ? by : bz);
        const float pa = a == 0 ? pos[0] : (a == 1 ? pos[1] : pos[2]);
        const float pb = (float)ba * mc.bs + ((float)(qa + 1) + 0.5f) * mc.v;
        const float pn = pa + t * (pb - pa);
        if (a == 0) pos[0] = pn;
        else if (a == 1) pos[1] = pn;
        else pos[2] = pn;
        const bool inside = pos[0] > ka.lo[0] && pos[0] < ka.hi[0] && pos[1] > ka.lo[1] && pos[1] < ka.hi[1] && pos[2] > ka.lo[2] &&
                            pos[2] < ka.hi[2];  // strict on both sides, like the reference
        int fv = -2;
        if (inside) {
          fv = -1;
          if (F.pool) {
            int lin;
            const u64 key = voxel_at(mc, pos, lin);
            const int fs = layer_lookup(F, key);
            if (fs >= 0 && F.poolw[(size_t)fs * kVPB + lin] > 0.0f) fv = fs * kVPB + lin;
          }
          if (fv < 0 && ka.remove_zero) fv = -2;  // no feature voxel: an all-zero row
        }
        s_vox[local] = fv;
        s_pos[3 * local] = pos[0];
        s_pos[3 * local + 1] = pos[1];
        s_pos[3 * local + 2] = pos[2];
        local++;
      }
    }
    __syncthreads();
    if (ka.remove_zero) {
      for (int v0 = 0; v0 < nv; v0 += 32) {
        const int vi = v0 + group;
        const int fv = vi < nv ? s_vox[vi] : -2;
        if (fv >= 0) {  // (uniform over the 8 lanes of a group)
          const unsigned short* row = reinterpret_cast<const unsigned short*>(F.pool) + (size_t)fv * C;
          bool nzrow = false;
          for (int p0 = 0; 8 * p0 < ka.used && !nzrow; p0 += 8) {
            const int p = p0 + gl;
            bool nz = false;
            if (8 * p < ka.used) {
              const uint4 q = *reinterpret_cast<const uint4*>(row + 8 * p);
              const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {  // -0.0 == 0, like `!= 0` on the half values
                if ((w[j] & 0x7fffu) != 0u && 8 * p + 2 * j < ka.used) nz = true;
                if ((w[j] & 0x7fff0000u) != 0u && 8 * p + 2 * j + 1 < ka.used) nz = true;
              }
            }
            const unsigned long long b = __ballot(nz);
            nzrow = ((b >> (lane & ~7)) & 0xffull) != 0ull;
          }
          if (gl == 0 && !nzrow) s_vox[vi] = -2;
        }
      }
      __syncthreads();
    }
    int kc = 0;
    for (int k = 0; k < cnt; ++k) kc += s_vox[ea + k] != -2 ? 1 : 0;
    int ka_off, kb_off, ktot, ktb;
    block_excl_scan2<4>(kc, 0, m.scan, ka_off, kb_off, ktot, ktb);
    if (threadIdx.x == 0) {
      const int base = ktot > 0 ? atomicAdd(total, ktot) : 0;
      s_base = base;
      counts[i] = ktot;
      chunk[i] = base;
    }
    __syncthreads();
    int o = s_base + ka_off;
    for (int k = 0; k < cnt; ++k) {
      const int fv = s_vox[ea + k];
      if (fv == -2) continue;
      if (o < list_cap)
        list[o] = make_uint4(__float_as_uint(s_pos[3 * (ea + k)]), __float_as_uint(s_pos[3 * (ea + k) + 1]),
                             __float_as_uint(s_pos[3 * (ea + k) + 2]), (unsigned)fv);
      o++;
    }
    __syncthreads();
  }
}

constexpr int kGatherLdsBlocks = 8192;  // prefix sums of up to this many live blocks are rebuilt in LDS by every workgroup

// rows: ranks (int64) among the kept vertices, or null = identity.  Output row j < n_take = kept vertex rows[j];
// rows n_take <= j < n_out are zero padding (valid 0).  32 lanes per row, 8 channels per lane and step.
template <bool F32>
__global__ __launch_bounds__(256) void k_model_inputs_gather(const int* __restrict__ counts, const int* __restrict__ chunk,
                                                            const int* __restrict__ goffsets, int n_blocks,
                                                            const uint4* __restrict__ list, LayerDev F, int C, int U,
                                                            const long long* __restrict__ rows, int n_take, int n_out,
                                                            float* __restrict__ verts, void* __restrict__ feats,
                                                            uint8_t* __restrict__ valid) {
  __shared__ int s_off[kGatherLdsBlocks + 1];
  __shared__ int s_scan[10];
  const bool in_lds = goffsets == nullptr;
  if (in_lds) {
    // exclusive prefix sums of counts: a thread owns a run of consecutive blocks
    const int per = (n_blocks + 255) / 256;
    const int b0 = threadIdx.x * per;
    int sum = 0;
    for (int k = 0; k < per; ++k) sum += (b0 + k < n_blocks) ? counts[b0 + k] : 0;
    int ea, eb, ta, tb;
    block_excl_scan2<4>(sum, 0, s_scan, ea, eb, ta, tb);
    int run = ea;
    for (int k = 0; k < per; ++k) {
      if (b0 + k < n_blocks) {
        s_off[b0 + k] = run;
        run += counts[b0 + k];
      }
    }
    if (threadIdx.x == 0) s_off[n_blocks] = ta;
    __syncthreads();
  }
  const int* off = in_lds ? s_off : goffsets;
  const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
  for (int j = blockIdx.x * 8 + g; j < n_out; j += gridDim.x * 8) {
    uint4 e = make_uint4(0u, 0u, 0u, 0xffffffffu);
    const bool take = j < n_take;
    if (take) {
      const int r = rows ? (int)rows[j] : j;
      int lo = 0, hi = n_blocks;  // largest b with off[b] <= r (blocks without kept vertices share their successor's offset)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= r) lo = mid;
        else hi = mid;
      }
      e = list[chunk[lo] + (r - off[lo])];
    }
    if (l < 3) verts[3 * (size_t)j + l] = __uint_as_float(l == 0 ? e.x : (l == 1 ? e.y : e.z));
    if (l == 3 && valid) valid[j] = take ? 1 : 0;
    if (!feats) continue;
    const int fv = (int)e.w;
    const __half* src = fv >= 0 ? reinterpret_cast<const __half*>(F.pool) + (size_t)fv * C : nullptr;
    for (int c0 = 8 * l; c0 < U; c0 += 256) {
      half8 h;
#pragma unroll
      for (int k = 0; k < 8; ++k) h[k] = (_Float16)0.0f;
      if (src) h = *reinterpret_cast<const half8*>(src + c0);  // C is a multiple of 8 and c0 < U <= C: in range
      if (F32) {
        float* dst = reinterpret_cast<float*>(feats) + (size_t)j * U + c0;
        if (c0 + 8 <= U && (U & 3) == 0) {
          *reinterpret_cast<float4*>(dst) = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
        } else {
          for (int k = 0; k < 8 && c0 + k < U; ++k) dst[k] = (float)h[k];
        }
      } else {
        __half* dst = reinterpret_cast<__half*>(feats) + (size_t)j * U + c0;
        if (c0 + 8 <= U && (U & 7) == 0) {
          *reinterpret_cast<half8*>(dst) = h;
        } else {
          for (int k = 0; k < 8 && c0 + k < U; ++k) reinterpret_cast<_Float16*>(dst)[k] = h[k];
        }
      }
    }
  }
}

void launch_mesh_keep(const LayerDev& tsdf, const LayerDev& feat, const MapConsts& mc, const float* lo, const float* hi, int used,
                      int remove_zero, int* counts, int* chunk, int* total, uint4* list, int list_cap, hipStream_t s) {
  KeepArgs ka;
  for (int a = 0; a < 3; ++a) {
    ka.lo[a] = lo[a];
    ka.hi[a] = hi[a];
  }
  ka.used = used;
  ka.remove_zero = remove_zero;
  int g = hinted(tsdf.hint_live, tsdf.cap);
  g = g < 8192 ? g : 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_mesh_keep, dim3(g), dim3(256), 0, s, tsdf, feat, mc, ka, counts, chunk, total, list, list_cap);
}

// offsets: device prefix sums of counts (k_mesh_scan) -- required when n_blocks > kGatherLdsBlocks, else pass null
void launch_model_inputs_gather(const int* counts, const int* chunk, const int* offsets, int n_blocks, const uint4* list,
                                const LayerDev& feat, int C, int used, const long long* rows, int n_take, int n_out, float* verts,
                                void* feats, bool f32, uint8_t* valid, hipStream_t s) {
  if (n_out <= 0) return;
  int g = (n_out + 7) / 8;
  g = g < 2048 ? g : 2048;
  const int* goff = n_blocks > kGatherLdsBlocks ? offsets : nullptr;
  if (f32)
    hipLaunchKernelGGL(k_model_inputs_gather<true>, dim3(g), dim3(256), 0, s, counts, chunk, goff, n_blocks, list, feat, C, used, rows,
                       n_take, n_out, verts, feats, valid);
  else
    hipLaunchKernelGGL(k_model_inputs_gather<false>, dim3(g), dim3(256), 0, s, counts, chunk, goff, n_blocks, list, feat, C, used, rows,
                       n_take, n_out, verts, feats, valid);
}

int model_inputs_lds_blocks() { return kGatherLdsBlocks; }
void launch_mesh_scan_counts(const LayerDev& tsdf, const int* counts, int* offsets, int* out2, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_scan, dim3(1), dim3(256), 0, s, tsdf, counts, offsets, out2);
}

// ------------------------------------------------------------------------------------------------
// Triangle connectivity + per-vertex colour (Mapper.get_color_mesh / FeatureMesh.triangles(), consumed by the
// reference's visualiser: visualization/visualizer.py:656-672, paper/utils/utils.py:84-92).  Not on the per-frame path.
// Per block: the vertex walk of k_mesh_emit gives every lattice edge its vertex index (LDS, u16); the 512 cubes, two per
// thread in lexicographic order, emit the triangles of their corner pattern (mmf_mc_table.h).  Same order as the oracle.
// ------------------------------------------------------------------------------------------------
__device__ inline int cube_pattern(const MeshLds& m, int c) {
  if (!m.CV[c]) return 0;
  const int x = c >> 6, y = (c >> 3) & 7, z = c & 7;
  const int q = x * 81 + y * 9 + z;
  int pat = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (m.D[q + (k >> 2) * 81 + ((k >> 1) & 1) * 9 + (k & 1)] < 0.0f) pat |= 1 << k;
  return pat;
}

__global__ __launch_bounds__(256) void k_mesh_tri_count(LayerDev T, MapConsts mc, int* __restrict__ tcounts) {
  __shared__ MeshLds m;
  const int n = T.ctr[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = T.live[i];
    int bx, by, bz;
    unpack_key(T.slot_key[slot], bx, by, bz);
    load_lattice(T, mc, slot, bx, by, bz, m);
    const int cnt = mmf_mc_num_tris[cube_pattern(m, 2 * threadIdx.x)] + mmf_mc_num_tris[cube_pattern(m, 2 * threadIdx.x + 1)];
    int ea, eb, ta, tb;
    block_excl_scan2<4>(cnt, 0, m.scan, ea, eb, ta, tb);
    if (threadIdx.x == 0) tcounts[i] = ta;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_mesh_tri_emit(LayerDev T, LayerDev Cl, MapConsts mc, const int* __restrict__ voffsets,
                                                      const int* __restrict__ toffsets, int n_blocks, int32_t* __restrict__ tris,
                                                      uint8_t* __restrict__ vcolors, int V, int Tn) {
  __shared__ MeshLds m;
  __shared__ uint16_t s_vid[kEdges];  // vertex index (within the block) of every lattice edge, 0xffff = none
  for (int i = blockIdx.x; i < n_blocks; i += gridDim.x) {
    const int slot = T.live[i];
    int bx, by, bz;
    unpack_key(T.slot_key[slot], bx, by, bz);
    load_lattice(T, mc, slot, bx, by, bz, m);
    int cnt = 0;
    const int e0 = threadIdx.x * kEdgesPerThread;
    for (int k = 0; k < kEdgesPerThread; ++k) {
      const int e = e0 + k;
      if (e >= kEdges) break;
      int qx, qy, qz, a;
      float t;
      cnt += edge_vertex(m, e, qx, qy, qz, a, t) ? 1 : 0;
    }
    int ea, eb, nv, tb;
    block_excl_scan2<4>(cnt, 0, m.scan, ea, eb, nv, tb);
    const int voff = voffsets[i];
    int local = ea;
    for (int k = 0; k < kEdgesPerThread; ++k) {
      const int e = e0 + k;
      if (e >= kEdges) break;
      int qx, qy, qz, a;
      float t;
      if (!edge_vertex(m, e, qx, qy, qz, a, t)) {
        s_vid[e] = 0xffffu;
        continue;
      }
      s_vid[e] = (uint16_t)local;
      if (vcolors && voff + local < V) {  // colour voxel containing the vertex (same position arithmetic as k_mesh_emit)
        float pos[3] = {(float)bx * mc.bs + ((float)qx + 0.5f) * mc.v, (float)by * mc.bs + ((float)qy + 0.5f) * mc.v,
                        (float)bz * mc.bs + ((float)qz + 0.5f) * mc.v};
        const int qa = a == 0 ? qx : (a == 1 ? qy : qz);
        const int ba = a == 0 ? bx : (a == 1 ? by : bz);
        const float pa = a == 0 ? pos[0] : (a == 1 ? pos[1] : pos[2]);
        const float pb = (float)ba * mc.bs + ((float)(qa + 1) + 0.5f) * mc.v;
        const float pn = pa + t * (pb - pa);
        if (a == 0) pos[0] = pn;
        else if (a == 1) pos[1] = pn;
        else pos[2] = pn;
        int lin;
        const u64 key = voxel_at(mc, pos, lin);
        unsigned rgb = 0;
        if (Cl.pool) {
          const int cs = layer_lookup(Cl, key);
          if (cs >= 0) {
            const uint2 e2 = reinterpret_cast<const uint2*>(Cl.pool)[(size_t)cs * kVPB + lin];
            if (__uint_as_float(e2.y) > 0.0f) rgb = e2.x;
          }
        }
        uint8_t* o = vcolors + 3 * (size_t)(voff + local);
        o[0] = (uint8_t)(rgb & 0xffu);
        o[1] = (uint8_t)((rgb >> 8) & 0xffu);
        o[2] = (uint8_t)((rgb >> 16) & 0xffu);
      }
      local++;
    }
    __syncthreads();
    const int p0 = cube_pattern(m, 2 * threadIdx.x), p1 = cube_pattern(m, 2 * threadIdx.x + 1);
    int tea, teb, nt, ttb;
    block_excl_scan2<4>((int)mmf_mc_num_tris[p0] + (int)mmf_mc_num_tris[p1], 0, m.scan, tea, teb, nt, ttb);
    int tpos = toffsets[i] + tea;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c = 2 * threadIdx.x + r, pat = r ? p1 : p0;
      const int x = c >> 6, y = (c >> 3) & 7, z = c & 7;
      for (int k = 0; k < (int)mmf_mc_num_tris[pat]; ++k) {
        int v3[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int e = mmf_mc_tris[pat][3 * k + j];
          const int a = e >> 2, s1 = (e >> 1) & 1, s2 = e & 1;
          const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
          int qx = x, qy = y, qz = z;  // lattice point the cube edge starts at (no dynamic register indexing)
          if (a1 == 0) qx += s1; else if (a1 == 1) qy += s1; else qz += s1;
          if (a2 == 0) qx += s2; else if (a2 == 1) qy += s2; else qz += s2;
          v3[j] = voff + (int)s_vid[3 * (qx * 81 + qy * 9 + qz) + a];
        }
        if (tpos < Tn) {
          tris[3 * (size_t)tpos] = v3[0];
          tris[3 * (size_t)tpos + 1] = v3[1];
          tris[3 * (size_t)tpos + 2] = v3[2];
        }
        tpos++;
      }
    }
    __syncthreads();
  }
}

void launch_mesh_tri_count(const LayerDev& tsdf, const MapConsts& mc, int* tcounts, int* toffsets, int* out2, hipStream_t s) {
  int g = hinted(tsdf.hint_live, tsdf.cap);
  g = g < 8192 ? g : 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_mesh_tri_count, dim3(g), dim3(256), 0, s, tsdf, mc, tcounts);
  hipLaunchKernelGGL(k_mesh_scan, dim3(1), dim3(256), 0, s, tsdf, (const int*)tcounts, toffsets, out2);
}

void launch_mesh_tri_emit(const LayerDev& tsdf, const LayerDev& color, const MapConsts& mc, const int* voffsets, const int* toffsets,
                          int n_blocks, int32_t* tris, uint8_t* vcolors, int V, int Tn, hipStream_t s) {
  if (n_blocks <= 0 || V <= 0) return;
  int g = n_blocks < 8192 ? n_blocks : 8192;
  hipLaunchKernelGGL(k_mesh_tri_emit, dim3(g), dim3(256), 0, s, tsdf, color, mc, voffsets, toffsets, n_blocks, tris, vcolors, V, Tn);
}

}  // namespace mmf
