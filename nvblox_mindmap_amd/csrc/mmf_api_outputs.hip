// mmf_api_outputs.hip -- the reading half of the C ABI (include/mmfusion.h): feature / colour mesh, map -> model inputs, block
// export / import, point queries, diagnostics and the per-launch profiling accessors.
#include "mmf_api_internal.h"

using namespace mmf;
using namespace mmf_host;

namespace {

__global__ void k_unpack_keys(const u64* keys, int n, int32_t* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x, y, z;
  unpack_key(keys[i], x, y, z);
  out[3 * i] = x;
  out[3 * i + 1] = y;
  out[3 * i + 2] = z;
}

}  // namespace

extern "C" {

int mmf_update_feature_mesh(mmf_handle h, int mapper_id, void* stream, int* num_vertices) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (!num_vertices) return fail(MMF_ERR_INVALID_ARG, "null num_vertices");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(h, MMF_K_MESH, s);
    launch_mesh_count(m->tsdf.d, m->mc, m->mesh_counts, m->mesh_offsets, m->mesh_out2, s);
  }
  HIP_TRY(hipMemcpyAsync(h->pinned, m->mesh_out2, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h->pinned + 11, m->tsdf.d.ctr + 3, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  m->mesh_V = h->pinned[0];
  m->mesh_nblocks = h->pinned[1];
  m->mesh_epoch = m->tsdf_epoch;
  *num_vertices = m->mesh_V;
  {  // a natural synchronisation point: asynchronous device errors (hand-over failure, pool exhaustion) surface here, once
    const int bits = h->pinned[11] & 2;  // (exhaustion keeps its own reporting point: mmf_num_allocated_blocks)
    MMF_TRY(report_device_errors(h, *m, &m->tsdf, &bits, s));
  }
  return check_launch();
}

int mmf_get_feature_mesh(mmf_handle h, int mapper_id, float* verts, void* vfeat, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (m->mesh_epoch != m->tsdf_epoch)
    return fail(MMF_ERR_BAD_STATE, "the map changed since mmf_update_feature_mesh; call it again");
  if (m->mesh_V == 0) return MMF_OK;
  if (!verts || !vfeat) return fail(MMF_ERR_INVALID_ARG, "null output buffer");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  LayerDev F = m->feat.allocated ? m->feat.d : LayerDev{};
  {
    ProfScope ps(h, MMF_K_MESH, s);
    launch_mesh_emit(m->tsdf.d, F, m->mc, m->mesh_offsets, m->mesh_nblocks, verts, (__half*)vfeat, m->mesh_V, s);
  }
  return check_launch();
}

int mmf_model_inputs_prepare(mmf_handle h, int mapper_id, const float* lo, const float* hi, int used, int remove_zero, void* stream,
                             int* num_kept) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (!lo || !hi || !num_kept) return fail(MMF_ERR_INVALID_ARG, "null argument");
  if (used < 1 || used > m->mc.C) return fail(MMF_ERR_INVALID_ARG, "used_channels must be in 1 .. feature_channels");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  if (!m->mi_counts) {
    HIP_TRY(hipMalloc(&m->mi_counts, sizeof(int) * (size_t)m->mesh_cap));
    HIP_TRY(hipMalloc(&m->mi_chunk, sizeof(int) * (size_t)m->mesh_cap));
    HIP_TRY(hipMalloc(&m->mi_offsets, sizeof(int) * ((size_t)m->mesh_cap + 2)));
    HIP_TRY(hipMalloc(&m->mi_total, sizeof(int) * 2));
    m->mi_list_cap = 1 << 16;
    HIP_TRY(hipMalloc(&m->mi_list, sizeof(uint4) * (size_t)m->mi_list_cap));
  }
  LayerDev F = m->feat.allocated ? m->feat.d : LayerDev{};
  for (int attempt = 0;; ++attempt) {
    HIP_TRY(hipMemsetAsync(m->mi_total, 0, sizeof(int), s));
    {
      ProfScope ps(h, MMF_K_MESH, s);
      launch_mesh_keep(m->tsdf.d, F, m->mc, lo, hi, used, remove_zero ? 1 : 0, m->mi_counts, m->mi_chunk, m->mi_total, m->mi_list,
                       m->mi_list_cap, s);
    }
    HIP_TRY(hipMemcpyAsync(h->pinned, m->mi_total, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h->pinned + 11, m->tsdf.d.ctr + 3, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    m->mi_n = h->pinned[0];
    m->mi_nblocks = h->pinned[1];
    if (m->mi_n <= m->mi_list_cap) break;
    if (attempt) return fail(MMF_ERR_BAD_STATE, "model inputs: vertex list overflow after growing it");
    // more kept vertices than the list holds (they were counted, not stored): grow and run the pass again
    (void)hipFree(m->mi_list);
    m->mi_list = nullptr;
    size_t cap = (size_t)m->mi_list_cap;
    while (cap < (size_t)m->mi_n + (size_t)m->mi_n / 2) cap *= 2;
    if (cap > ((size_t)1 << 30)) return fail(MMF_ERR_INVALID_ARG, "model inputs: too many vertices");
    HIP_TRY(hipMalloc(&m->mi_list, sizeof(uint4) * cap));
    m->mi_list_cap = (int)cap;
  }
  if (m->mi_nblocks > model_inputs_lds_blocks())  // maps with more live blocks than the gather kernel scans in LDS
    launch_mesh_scan_counts(m->tsdf.d, m->mi_counts, m->mi_offsets, m->mi_offsets + m->mesh_cap, s);
  m->mi_epoch = m->tsdf_epoch;
  m->mi_feat_frames = m->frames[2];
  m->mi_used = used;
  *num_kept = m->mi_n;
  {
    const int bits = h->pinned[11] & 2;
    MMF_TRY(report_device_errors(h, *m, &m->tsdf, &bits, s));
  }
  return check_launch();
}

int mmf_model_inputs_gather(mmf_handle h, int mapper_id, const int64_t* rows, int n_take, int n_out, float* verts, void* feats,
                            int features_f32, uint8_t* valid, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  if (m->mi_epoch != m->tsdf_epoch || m->mi_feat_frames != m->frames[2] || m->pending_decay)
    return fail(MMF_ERR_BAD_STATE, "the map changed since mmf_model_inputs_prepare; call it again");
  if (n_take < 0 || n_out < n_take) return fail(MMF_ERR_INVALID_ARG, "need 0 <= n_take <= n_out");
  if (!rows && n_take > m->mi_n) return fail(MMF_ERR_INVALID_ARG, "n_take exceeds the kept rows");
  if (n_take > 0 && m->mi_n == 0) return fail(MMF_ERR_INVALID_ARG, "no kept rows to take from");
  if (n_out == 0) return MMF_OK;
  if (!verts) return fail(MMF_ERR_INVALID_ARG, "null vertex buffer");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  LayerDev F = m->feat.allocated ? m->feat.d : LayerDev{};
  {
    ProfScope ps(h, MMF_K_MESH, s);
    launch_model_inputs_gather(m->mi_counts, m->mi_chunk, m->mi_offsets, m->mi_nblocks, m->mi_list, F, m->mc.C, m->mi_used,
                               (const long long*)rows, n_take, n_out, verts, feats, features_f32 != 0, valid, s);
  }
  return check_launch();
}

int mmf_update_mesh_topology(mmf_handle h, int mapper_id, void* stream, int* num_vertices, int* num_triangles) {
  if (!num_vertices || !num_triangles) return fail(MMF_ERR_INVALID_ARG, "null output");
  MMF_TRY(mmf_update_feature_mesh(h, mapper_id, stream, num_vertices));  // vertex counts / offsets (flushes a pending decay)
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(h, MMF_K_MESH, s);
    launch_mesh_tri_count(m->tsdf.d, m->mc, m->mesh_tcounts, m->mesh_toffsets, m->mesh_tout2, s);
  }
  HIP_TRY(hipMemcpyAsync(h->pinned, m->mesh_tout2, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  m->mesh_T = h->pinned[0];
  m->mesh_tepoch = m->tsdf_epoch;
  *num_triangles = m->mesh_T;
  return check_launch();
}

int mmf_get_mesh_topology(mmf_handle h, int mapper_id, int32_t* triangles, uint8_t* vertex_colors, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (m->mesh_tepoch != m->tsdf_epoch || m->mesh_epoch != m->tsdf_epoch)
    return fail(MMF_ERR_BAD_STATE, "the map changed since mmf_update_mesh_topology; call it again");
  if (m->mesh_V == 0) return MMF_OK;
  if (!triangles && m->mesh_T > 0) return fail(MMF_ERR_INVALID_ARG, "null triangle buffer");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  LayerDev Cl = m->color.allocated ? m->color.d : LayerDev{};
  {
    ProfScope ps(h, MMF_K_MESH, s);
    launch_mesh_tri_emit(m->tsdf.d, Cl, m->mc, m->mesh_offsets, m->mesh_toffsets, m->mesh_nblocks, triangles, vertex_colors,
                         m->mesh_V, m->mesh_T, s);
  }
  return check_launch();
}

static Layer* pick_layer(Mapper* m, int layer) {
  return layer == MMF_LAYER_TSDF ? &m->tsdf : (layer == MMF_LAYER_COLOR ? &m->color : (layer == MMF_LAYER_FEATURE ? &m->feat : nullptr));
}

int mmf_num_allocated_blocks(mmf_handle h, int mapper_id, int layer, void* stream, int* out) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  Layer* L = pick_layer(m, layer);
  if (!L || !out) return fail(MMF_ERR_INVALID_ARG, "bad layer / null out");
  if (!L->allocated) {
    *out = 0;
    return MMF_OK;
  }
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(h->pinned + 8, L->d.ctr, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *out = h->pinned[8];
  return report_device_errors(h, *m, L, &h->pinned[11], s);
}

int mmf_get_block_indices(mmf_handle h, int mapper_id, int layer, int32_t* out, int n, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  Layer* L = pick_layer(m, layer);
  if (!L) return fail(MMF_ERR_INVALID_ARG, "bad layer");
  if (n <= 0 || !L->allocated) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  launch_get_indices(L->d, out, n, (hipStream_t)stream);
  return check_launch();
}

int mmf_get_tsdf_blocks(mmf_handle h, int mapper_id, float* out, int n, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (n <= 0) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  launch_gather_pool(m->tsdf.d, m->tsdf.block_bytes, out, n, (hipStream_t)stream);
  return check_launch();
}

int mmf_get_feature_blocks(mmf_handle h, int mapper_id, void* feats, float* weights, int n, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (n <= 0 || !m->feat.allocated) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  launch_gather_pool(m->feat.d, m->feat.block_bytes, feats, n, (hipStream_t)stream);
  launch_gather_poolw(m->feat.d, weights, n, (hipStream_t)stream);
  return check_launch();
}

int mmf_get_color_blocks(mmf_handle h, int mapper_id, uint8_t* rgb, float* weights, int n, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (n <= 0 || !m->color.allocated) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  launch_gather_color(m->color.d, rgb, weights, n, (hipStream_t)stream);
  return check_launch();
}

int mmf_import_blocks(mmf_handle h, int mapper_id, int layer, const int32_t* idx, const void* payload, const float* weights, int n,
                      void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  if (n < 0 || (n > 0 && (!idx || !payload))) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_import_blocks");
  if (layer < MMF_LAYER_TSDF || layer > MMF_LAYER_FEATURE) return fail(MMF_ERR_INVALID_ARG, "bad layer id");
  if (layer != MMF_LAYER_TSDF && n > 0 && !weights) return fail(MMF_ERR_INVALID_ARG, "appearance layers need the weight plane");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  if (layer == MMF_LAYER_COLOR) MMF_TRY(ensure_app_layer(*m, m->color, sizeof(uint2) * kVPB, false));
  if (layer == MMF_LAYER_FEATURE)
    MMF_TRY(ensure_app_layer(*m, m->feat, sizeof(__half) * kVPB * (size_t)m->P.feature_channels, true));
  Layer& L = layer == MMF_LAYER_TSDF ? m->tsdf : layer == MMF_LAYER_COLOR ? m->color : m->feat;
  if (n > L.d.cap)
    return fail(MMF_ERR_POOL_EXHAUSTED, "saved layer has " + std::to_string(n) + " blocks, the pool holds " + std::to_string(L.d.cap));
  if (layer == MMF_LAYER_TSDF) m->pending_decay = false;  // the content it would have decayed is replaced
  if (layer == MMF_LAYER_TSDF) {
    m->wmax_valid = false;  // imported weights: wmax is rebuilt by the next fused frame
    m->lazy_lag = false;    // (the voxels that were behind are replaced)
    drop_lazy(*m);
  }
  launch_layer_reset(L.d, s);
  launch_import_index(L.d, idx, n, s);
  if (n > 0) {
    if (layer == MMF_LAYER_COLOR) {
      launch_import_color(L.d, (const uint8_t*)payload, weights, n, s);
    } else {
      HIP_TRY(hipMemcpyAsync(L.d.pool, payload, L.block_bytes * (size_t)n, hipMemcpyDeviceToDevice, s));
      if (layer == MMF_LAYER_FEATURE)
        HIP_TRY(hipMemcpyAsync(L.d.poolw, weights, sizeof(float) * kVPB * (size_t)n, hipMemcpyDeviceToDevice, s));
    }
    if (layer == MMF_LAYER_TSDF) launch_block_free_all(L.d, m->mc, n, s);
  }
  if (layer == MMF_LAYER_TSDF) {
    m->touched = m->touched || n > 0;
    m->tsdf_epoch++;
    m->mesh_epoch = -1;
    m->mi_epoch = -1;
  }
  // out-of-range indices are flagged on the device; report them now (loading is not a hot path)
  HIP_TRY(hipStreamSynchronize(s));
  int err = 0;
  HIP_TRY(hipMemcpy(&err, L.d.ctr + 3, sizeof(int), hipMemcpyDeviceToHost));
  if (err & 2) return fail(MMF_ERR_INVALID_ARG, "saved block indices lie outside this mapper's workspace bounds / key range");
  return check_launch();
}

int mmf_query_layer(mmf_handle h, int mapper_id, int layer, const float* pts, int n, float* out, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (n <= 0) return MMF_OK;
  if (!pts || !out) return fail(MMF_ERR_INVALID_ARG, "null buffer");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  if (layer == MMF_LAYER_TSDF) {
    launch_query_tsdf(m->tsdf.d, m->mc, pts, n, out, s);
  } else if (layer == MMF_LAYER_FEATURE) {
    if (!m->feat.allocated) {
      HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)n * (m->mc.C + 1), s));
      return MMF_OK;
    }
    launch_query_feature(m->feat.d, m->mc, pts, n, out, s);
  } else {
    return fail(MMF_ERR_INVALID_ARG, "query_layer supports the TSDF and feature layers");
  }
  return check_launch();
}

// ---- image-side ops -----------------------------------------------------------------------------
// ---- diagnostics ----------------------------------------------------------------------------------
int mmf_get_synthetic_depth_dims(mmf_handle h, int mapper_id, int* Hs, int* Ws) {
  Mapper* m;
  MMF_TRY(get_mapper(h, mapper_id, &m));
  *Hs = m->synth_H;
  *Ws = m->synth_W;
  return MMF_OK;
}

int mmf_get_synthetic_depth(mmf_handle h, int mapper_id, float* out, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  if (!m->synth || m->synth_W * m->synth_H == 0) return fail(MMF_ERR_BAD_STATE, "no synthetic depth rendered yet");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(out, m->synth, sizeof(float) * (size_t)m->synth_W * m->synth_H, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMF_OK;
}

int mmf_render_synthetic_depth(mmf_handle h, int mapper_id, int H, int W, const float* T16, const float* K9, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  HIP_TRY(hipSetDevice(h->device));
  Cam cam = cam_from_K(K9, W, H);
  Rigid T_L_C;
  rigid_from_T(T16, T_L_C);
  m->synth_epoch = -1;  // force
  MMF_TRY(ensure_synth(h, *m, cam, T_L_C, T16, K9, (hipStream_t)stream));
  return check_launch();
}

int mmf_last_view_block_count(mmf_handle h, int mapper_id, void* stream, int* out) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(h->pinned + 16, m->sc[0].cand_count, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *out = h->pinned[16];
  return MMF_OK;
}

int mmf_get_last_view_blocks(mmf_handle h, int mapper_id, int32_t* out, int n, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  if (n <= 0) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_unpack_keys, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const mmf::u64*)m->sc[0].cand_key, n, out);
  return check_launch();
}

int mmf_get_stats(mmf_handle h, int mapper_id, void* stream, int64_t* out8) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  long long* host = reinterpret_cast<long long*>(h->pinned + 32);
  HIP_TRY(hipMemcpyAsync(host, m->stats, sizeof(long long) * MMF_NUM_STATS, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < MMF_NUM_STATS; ++i) out8[i] = host[i];
  out8[0] = m->frames[0];
  out8[3] = m->frames[1];
  out8[5] = m->frames[2];
  return MMF_OK;
}

int mmf_debug_alloc_recoveries(mmf_handle h, int mapper_id, void* stream, int64_t* out) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  if (!out) return fail(MMF_ERR_INVALID_ARG, "null out");
  HIP_TRY(hipSetDevice(h->device));
  unsigned long long v = 0;
  HIP_TRY(hipMemcpyAsync(&v, m->pub + kPubRec + 3 * (size_t)m->tsdf.d.cap + 1, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  *out = (int64_t)v;
  return MMF_OK;
}

int mmf_debug_hash_state(mmf_handle h, int mapper_id, int layer, void* stream, int64_t* out8) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  Layer* L = pick_layer(m, layer);
  if (!L || !out8) return fail(MMF_ERR_INVALID_ARG, "bad layer / null out");
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  out8[4] = m->last_vg.nx;
  out8[5] = m->last_vg.ny;
  out8[6] = m->last_vg.nz;
  out8[7] = m->lazy_epoch;  // decays applied lazily so far (large maps: one multiplication per live block, voxels catch up later)
  if (!L->allocated) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  int c[8];
  HIP_TRY(hipMemcpyAsync(c, L->d.ctr, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  out8[0] = L->d.dense ? 0 : (int64_t)L->d.hmask + 1;  // 0: the layer is indexed by its dense table, the hash is not maintained
  out8[1] = c[4];
  out8[2] = c[5];
  out8[3] = c[0];
  return MMF_OK;
}

int mmf_debug_count_tombstones(mmf_handle h, int mapper_id, int layer, void* stream, int64_t* out) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  Layer* L = pick_layer(m, layer);
  if (!L || !out) return fail(MMF_ERR_INVALID_ARG, "bad layer / null out");
  *out = 0;
  if (!L->allocated || L->d.dense) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  unsigned long long* n_dev = nullptr;
  HIP_TRY(hipMalloc(&n_dev, sizeof(*n_dev)));
  unsigned long long n = 0;
  hipError_t e = hipMemsetAsync(n_dev, 0, sizeof(*n_dev), (hipStream_t)stream);
  if (e == hipSuccess) {
    mmf::launch_count_tombstones(L->d, n_dev, (hipStream_t)stream);
    e = hipMemcpyAsync(&n, n_dev, sizeof(n), hipMemcpyDeviceToHost, (hipStream_t)stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(n_dev);
  HIP_TRY(e);
  *out = (int64_t)n;
  return MMF_OK;
}

int mmf_get_alloc_timeline(mmf_handle h, int mapper_id, int enable, int64_t* out6) {
  // out6 is really out8: [6] latest end / [7] earliest start of the mask column workgroups sharing the launch
  Mapper* m;
  MMF_TRY(get_mapper(h, mapper_id, &m));
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  if (out6) {
    for (int i = 0; i < 10; ++i) out6[i] = 0;
    if (m->timeline) {
      HIP_TRY(hipMemcpy(out6, m->timeline, sizeof(long long) * 10, hipMemcpyDeviceToHost));
      const long long reset[4] = {0, 0x7fffffffffffffffll, 0, 0};
      HIP_TRY(hipMemcpy(m->timeline + 6, reset, sizeof(reset), hipMemcpyHostToDevice));
    }
  }
  if (enable && !m->timeline) {
    HIP_TRY(hipMalloc(&m->timeline, sizeof(long long) * 16));
    HIP_TRY(hipMemset(m->timeline, 0, sizeof(long long) * 16));
    const long long big = 0x7fffffffffffffffll;
    HIP_TRY(hipMemcpy(m->timeline + 7, &big, sizeof(big), hipMemcpyHostToDevice));
  } else if (!enable && m->timeline) {
    (void)hipFree(m->timeline);
    m->timeline = nullptr;
  }
  return MMF_OK;
}

int mmf_reset_stats(mmf_handle h, int mapper_id, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemsetAsync(m->stats, 0, sizeof(long long) * MMF_NUM_STATS, (hipStream_t)stream));
  m->frames[0] = m->frames[1] = m->frames[2] = 0;
  return MMF_OK;
}

int mmf_profile_enable(mmf_handle h, int enable) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  h->prof = (unsigned)enable;
  return MMF_OK;
}

int mmf_profile_set_stride(mmf_handle h, int stride) {
  if (!h || stride < 1) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_profile_set_stride");
  h->prof_stride = (unsigned)stride;
  for (int i = 0; i < MMF_NUM_KERNEL_IDS; ++i) h->prof_seen[i] = 0;
  return MMF_OK;
}

int mmf_profile_get(mmf_handle h, int kernel_id, double* total_ms, int64_t* launches) {
  if (!h || kernel_id < 0 || kernel_id >= MMF_NUM_KERNEL_IDS) return fail(MMF_ERR_INVALID_ARG, "bad kernel id");
  HIP_TRY(hipSetDevice(h->device));
  MMF_TRY(prof_collect(h));
  if (total_ms) *total_ms = h->prof_ms[kernel_id];
  if (launches) *launches = h->prof_n[kernel_id];
  return MMF_OK;
}

int mmf_profile_reset(mmf_handle h) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  MMF_TRY(prof_collect(h));
  for (int i = 0; i < MMF_NUM_KERNEL_IDS; ++i) {
    h->prof_ms[i] = 0;
    h->prof_n[i] = 0;
  }
  return MMF_OK;
}

const char* mmf_kernel_name(int id) {
  static const char* names[MMF_NUM_KERNEL_IDS] = {
      "k_raycast_mark / k_front",  "k_alloc_jobs / k_count_tiles+k_scan_tiles+k_emit", "k_tsdf_integrate / k_tsdf_pass",
      "k_app_candidates",          "k_sphere_trace / k_sphere_alloc",                  "k_color_integrate",
      "k_feature_integrate / k_app_frame (gating)", "k_decay(+compact)",              "k_mesh_count/emit",
      "k_feature_flat"};
  return (id >= 0 && id < MMF_NUM_KERNEL_IDS) ? names[id] : "?";
}

}  // extern "C"
