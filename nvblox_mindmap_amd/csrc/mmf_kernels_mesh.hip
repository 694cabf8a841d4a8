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
