// mmf_api.hip -- host side of libmmfusion.so: the C ABI declared in include/mmfusion.h (this file: handles, parameters, the
// frame entry points, decay / flush / clear; see mmf_api_internal.h for the other translation units).
// Owns the block pools / hash tables of every mapper and sequences the kernels of one frame on the
// caller's HIP stream.  No device->host synchronisation on the per-frame path.
#include "mmf_api_internal.h"

using namespace mmf;
using namespace mmf_host;

namespace mmf_host {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int prof_collect(mmf_mapper_s* h) {
  for (auto& r : h->prof_recs) {
    HIP_TRY(hipEventSynchronize(r.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    h->prof_ms[r.id] += ms;
    h->prof_n[r.id] += 1;
    h->ev_pool.push_back(r.a);
    h->ev_pool.push_back(r.b);
  }
  h->prof_recs.clear();
  return MMF_OK;
}

unsigned next_pow2(unsigned v) {
  unsigned p = 1;
  while (p < v) p <<= 1;
  return p;
}

int ifloor_h(float x) { return (int)floorf(x); }

void rigid_from_T(const float* T, Rigid& o) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = T[i * 4 + j];
    o.t[i] = T[i * 4 + 3];
  }
}

// Rinv = R^T, tinv_i = -((Rinv_i0*t0 + Rinv_i1*t1) + Rinv_i2*t2)   
void rigid_inverse(const Rigid& a, Rigid& o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a.R[j * 3 + i];
  for (int i = 0; i < 3; ++i) o.t[i] = -((o.R[i * 3 + 0] * a.t[0] + o.R[i * 3 + 1] * a.t[1]) + o.R[i * 3 + 2] * a.t[2]);
}

void xform_h(const Rigid& T, const float* p, float* q) {
  for (int i = 0; i < 3; ++i) q[i] = ((T.R[i * 3 + 0] * p[0] + T.R[i * 3 + 1] * p[1]) + T.R[i * 3 + 2] * p[2]) + T.t[i];
}

Cam cam_from_K(const float* K, int W, int H) {
  Cam c;
  c.fx = K[0];
  c.fy = K[4];
  c.cx = K[2];
  c.cy = K[5];
  c.W = W;
  c.H = H;
  return c;
}

void derive_consts(const mmf_params& P, MapConsts& mc) {
  mc.v = P.voxel_size_m;
  mc.bs = 8.0f * mc.v;
  mc.inv_bs = 1.0f / mc.bs;
  mc.inv_v = 1.0f / mc.v;
  mc.trunc = P.truncation_distance_vox * mc.v;
  mc.max_dist = P.max_integration_distance_m;
  mc.max_weight = P.max_weight;
  mc.lin_md = P.lin_interp_max_diff_vox * mc.v;
  mc.weighting_mode = P.weighting_mode;
  mc.app_wm = P.appearance_measurement_weight;
  mc.app_max_w = P.appearance_max_weight;
  mc.ws_type = P.workspace_bounds_type;
  for (int a = 0; a < 3; ++a) {  // (mmf_params.block_index_by_division: the bounds' own blocks by the same rule as any point's)
    mc.ws_lo[a] = ifloor_h(P.block_index_by_division ? P.ws_min[a] / mc.bs : P.ws_min[a] * mc.inv_bs);
    mc.ws_hi[a] = ifloor_h(P.block_index_by_division ? P.ws_max[a] / mc.bs : P.ws_max[a] * mc.inv_bs);
  }
  mc.decay_factor = P.tsdf_decay_factor;
  mc.decay_thr = P.decayed_weight_threshold;
  mc.dealloc_decayed = P.deallocate_decayed_blocks;
  mc.mesh_min_w = P.mesh_min_weight;
  mc.st_sf = P.st_subsampling < 1 ? 1 : P.st_subsampling;
  mc.st_max_steps = P.st_max_steps;
  mc.st_max_len = P.st_max_ray_length_m;
  mc.st_eps = P.st_surface_eps_vox * mc.v;
  mc.C = P.feature_channels;
  mc.reach = P.raycast_to_truncation ? mc.trunc : 0.0f;
  mc.spec_flags = (P.raycast_walk_from_camera ? 1 : 0) | (P.appearance_blend_division ? 2 : 0) | (P.fma_contraction ? kSpecFma : 0) |
                  (P.block_index_by_division ? kSpecBlockDiv : 0) | (P.view_truncation_band_marking ? kSpecBandMark : 0) |
                  (P.bilinear_four_weight_sum ? kSpecBilin4 : 0);
}

int alloc_layer(Layer& L, int cap, size_t block_bytes, bool has_w) {
  L.allocated = true;  // from here on free_layer releases whatever exists (a failure below must not leak the earlier buffers)
  L.block_bytes = block_bytes;
  L.has_w = has_w;
  L.d.cap = cap;
  unsigned tcap = next_pow2((unsigned)(2 * cap < 1024 ? 1024 : 2 * cap));
  L.d.hmask = tcap - 1;
  HIP_TRY(hipMalloc(&L.d.htab, sizeof(HEntry) * tcap));
  HIP_TRY(hipMalloc(&L.d.slot_key, sizeof(u64) * cap));
  HIP_TRY(hipMalloc(&L.d.live, sizeof(int) * cap));
  HIP_TRY(hipMalloc(&L.d.free_stack, sizeof(int) * cap));
  HIP_TRY(hipMalloc(&L.d.ctr, sizeof(int) * 8));
  HIP_TRY(hipMalloc(&L.d.pool, block_bytes * (size_t)cap));
  if (has_w) HIP_TRY(hipMalloc(&L.d.poolw, sizeof(float) * kVPB * (size_t)cap));
  HIP_TRY(hipMemset(L.d.htab, 0xff, sizeof(HEntry) * tcap));
  HIP_TRY(hipMemset(L.d.ctr, 0, sizeof(int) * 8));
  HIP_TRY(hipMemset(L.d.slot_key, 0xff, sizeof(u64) * cap));
  L.allocated = true;
  return MMF_OK;
}

void free_layer(Layer& L) {
  if (!L.allocated) return;
  (void)hipFree(L.d.htab);
  (void)hipFree(L.d.slot_key);
  (void)hipFree(L.d.live);
  (void)hipFree(L.d.free_stack);
  (void)hipFree(L.d.ctr);
  (void)hipFree(L.d.pool);
  if (L.d.poolw) (void)hipFree(L.d.poolw);
  if (L.d.dense) (void)hipFree(L.d.dense);
  if (L.d.block_free) (void)hipFree(L.d.block_free);
  if (L.d.wmax) (void)hipFree(L.d.wmax);
  if (L.d.stamp) (void)hipFree(L.d.stamp);
  L = Layer{};  // (a mapper's lazy-decay arrays are its own: destroy_mapper)
}

void free_scratch(Scratch& sc) {
  (void)hipFree(sc.flags);
  (void)hipFree(sc.cell_slot);
  (void)hipFree(sc.cell_key);
  (void)hipFree(sc.tile_counts);
  (void)hipFree(sc.tile_offs);
  (void)hipFree(sc.cand_slot);
  (void)hipFree(sc.cand_key);
  (void)hipFree(sc.cand_new);
  (void)hipFree(sc.lb);
  sc.lb = nullptr;
  sc.flags = nullptr;
}

// (Re)allocate compaction scratch `which` for `ncells` cells.  Growing synchronises the device (rare:
// only when an unbounded map sees a larger view grid than ever before).
int ensure_scratch(Mapper& m, int which, int ncells) {
  if (ncells <= m.sc_cap[which]) return MMF_OK;
  HIP_TRY(hipDeviceSynchronize());
  Scratch& sc = m.sc[which];
  int* cand_count = sc.cand_count;
  int* alloc_ctx = sc.alloc_ctx;
  if (sc.flags) free_scratch(sc);
  int n = (ncells + 1023) & ~1023;
  int ntiles = n / 1024;
  HIP_TRY(hipMalloc(&sc.flags, (size_t)n));
  HIP_TRY(hipMemset(sc.flags, 0, (size_t)n));
  HIP_TRY(hipMalloc(&sc.cell_slot, sizeof(int) * (size_t)n));
  HIP_TRY(hipMalloc(&sc.cell_key, sizeof(u64) * (size_t)n));
  HIP_TRY(hipMalloc(&sc.tile_counts, sizeof(int2) * (size_t)ntiles));
  HIP_TRY(hipMalloc(&sc.tile_offs, sizeof(int2) * (size_t)ntiles));
  HIP_TRY(hipMalloc(&sc.cand_slot, sizeof(int) * (size_t)n));
  HIP_TRY(hipMalloc(&sc.cand_key, sizeof(u64) * (size_t)n));
  HIP_TRY(hipMalloc(&sc.cand_new, (size_t)n));
  HIP_TRY(hipMalloc(&sc.lb, sizeof(u64) * (2 * ((size_t)n / 1024 + 2) + 8)));
  HIP_TRY(hipMemset(sc.lb, 0, sizeof(u64) * (2 * ((size_t)n / 1024 + 2) + 8)));
  if (!cand_count) {
    HIP_TRY(hipMalloc(&cand_count, sizeof(int)));
    HIP_TRY(hipMemset(cand_count, 0, sizeof(int)));
    HIP_TRY(hipMalloc(&alloc_ctx, sizeof(int) * 4));
    HIP_TRY(hipMemset(alloc_ctx, 0, sizeof(int) * 4));
  }
  sc.cand_count = cand_count;
  sc.alloc_ctx = alloc_ctx;
  sc.hint_cand = m.hints ? m.hints + which : nullptr;
  m.sc_cap[which] = n;
  return MMF_OK;
}

// Dense block table of a bounding-box workspace (mirrors the hash; see LayerDev::dense).
int attach_dense_table(const Mapper& m, Layer& L) {
  if (m.P.workspace_bounds_type != 2 || L.d.cap >= 65535) return MMF_OK;
  LayerDev& d = L.d;
  long long nc = 1;
  for (int a = 0; a < 3; ++a) {
    d.d_lo[a] = m.mc.ws_lo[a];
    nc *= (long long)(m.mc.ws_hi[a] - m.mc.ws_lo[a] + 1);
  }
  d.d_ny = m.mc.ws_hi[1] - m.mc.ws_lo[1] + 1;
  d.d_nz = m.mc.ws_hi[2] - m.mc.ws_lo[2] + 1;
  if (nc > 32768) return MMF_OK;
  d.d_ncells = (int)nc;
  size_t bytes = ((size_t)nc * 2 + 15) / 16 * 16;
  HIP_TRY(hipMalloc(&d.dense, bytes));
  HIP_TRY(hipMemset(d.dense, 0, bytes));
  return MMF_OK;
}

size_t pub_words(const Mapper& m) { return kPubRec + 3 * (size_t)m.tsdf.d.cap + 2 + kNewBlockWgs; }

void destroy_mapper(Mapper* m);

// Fills a freshly constructed Mapper; on failure the caller (create_mapper) releases whatever was allocated so far.
int create_mapper_impl(const mmf_params& P, Mapper* m) {
  m->P = P;
  derive_consts(P, m->mc);
  int cap, app_cap;
  if (P.num_preallocated_blocks > 0) {
    cap = app_cap = P.num_preallocated_blocks;
  } else if (P.workspace_bounds_type == 2) {
    long long n = 1;
    for (int a = 0; a < 3; ++a) {
      long long d = (long long)m->mc.ws_hi[a] - m->mc.ws_lo[a] + 1;
      if (d <= 0) return fail(MMF_ERR_INVALID_ARG, "empty workspace bounds");
      n *= d;
    }
    if (n > (1ll << 24)) return fail(MMF_ERR_INVALID_ARG, "workspace bounding box too large; set num_preallocated_blocks");
    cap = app_cap = (int)n;
  } else {
    cap = 65536;
    app_cap = 16384;
  }
  m->app_cap = app_cap;
  HIP_TRY(hipHostMalloc(&m->hints, sizeof(int) * 8));
  for (int i = 0; i < 8; ++i) m->hints[i] = 0;
  MMF_TRY(alloc_layer(m->tsdf, cap, sizeof(float2) * kVPB, false));
  MMF_TRY(attach_dense_table(*m, m->tsdf));
  m->tsdf.d.hint_live = m->hints + 3;
  HIP_TRY(hipMalloc(&m->tsdf.d.block_free, (size_t)cap));
  HIP_TRY(hipMemset(m->tsdf.d.block_free, 0, (size_t)cap));
  HIP_TRY(hipMalloc(&m->tsdf.d.wmax, sizeof(float) * (size_t)cap));
  HIP_TRY(hipMemset(m->tsdf.d.wmax, 0, sizeof(float) * (size_t)cap));
  HIP_TRY(hipMalloc(&m->tsdf.d.stamp, sizeof(int) * (size_t)cap));
  HIP_TRY(hipMemset(m->tsdf.d.stamp, 0, sizeof(int) * (size_t)cap));
  HIP_TRY(hipMalloc(&m->kill, (size_t)cap));
  HIP_TRY(hipMemset(m->kill, 0, (size_t)cap));
  {
    const char* e = std::getenv("MMF_NO_ALLOC_TSDF");
    {
      const char* e2 = std::getenv("MMF_NO_BIG_MERGE");
      m->allow_big_merge = !(e2 && e2[0] == '1');
    }
    // (spec switches fma_contraction, block_index_by_division, view_truncation_band_marking, bilinear_four_weight_sum: the merged launch
    // k_alloc_tsdf and the front launch are built with the default forms only -- also for the stand-alone mmf_add_depth_frame)
    m->allow_merged = !(e && e[0] == '1') && !P.fma_contraction && !P.block_index_by_division && !P.view_truncation_band_marking &&
                      !P.bilinear_four_weight_sum;
  }
  HIP_TRY(hipMalloc(&m->pub, sizeof(u64) * pub_words(*m)));
  HIP_TRY(hipMemset(m->pub, 0, sizeof(u64) * pub_words(*m)));
  {
    const char* e = std::getenv("MMF_DEBUG_FORCE_ALLOC_TIMEOUT");
    m->debug_abandon = e ? std::atoi(e) : 0;
  }
  HIP_TRY(hipMalloc(&m->any_kill, sizeof(int)));
  HIP_TRY(hipMemset(m->any_kill, 0, sizeof(int)));
  m->lb_compact_words = 2 * ((size_t)cap / 1024 + 2) + 8;
  HIP_TRY(hipMalloc(&m->lb_compact, sizeof(u64) * m->lb_compact_words));
  HIP_TRY(hipMemset(m->lb_compact, 0, sizeof(u64) * m->lb_compact_words));
  HIP_TRY(hipMalloc(&m->rebuild_flag, sizeof(int)));
  HIP_TRY(hipMemset(m->rebuild_flag, 0, sizeof(int)));
  HIP_TRY(hipMalloc(&m->stats, sizeof(long long) * MMF_NUM_STATS));
  HIP_TRY(hipMemset(m->stats, 0, sizeof(long long) * MMF_NUM_STATS));
  HIP_TRY(hipMalloc(&m->mesh_counts, sizeof(int) * (size_t)cap));
  HIP_TRY(hipMalloc(&m->mesh_offsets, sizeof(int) * (size_t)cap));
  HIP_TRY(hipMalloc(&m->mesh_out2, sizeof(int) * 2));
  HIP_TRY(hipMalloc(&m->mesh_tcounts, sizeof(int) * (size_t)cap));
  HIP_TRY(hipMalloc(&m->mesh_toffsets, sizeof(int) * (size_t)cap));
  HIP_TRY(hipMalloc(&m->mesh_tout2, sizeof(int) * 2));
  m->mesh_cap = cap;
  return ensure_scratch(*m, 0, cap);
}

int create_mapper(const mmf_params& P, Mapper** out) {
  if (!(P.voxel_size_m > 0.f)) return fail(MMF_ERR_INVALID_ARG, "voxel_size_m must be > 0");
  if (P.feature_channels <= 0 || P.feature_channels % 8 != 0)
    return fail(MMF_ERR_INVALID_ARG, "feature_channels must be a positive multiple of 8");
  if (P.workspace_bounds_type < 0 || P.workspace_bounds_type > 2)
    return fail(MMF_ERR_INVALID_ARG, "workspace_bounds_type must be 0, 1 or 2");
  if (P.weighting_mode < 0 || P.weighting_mode > 5) return fail(MMF_ERR_INVALID_ARG, "weighting_mode must be 0 .. 5");
  if (P.workspace_bounds_type != 2 && !(P.max_integration_distance_m > 0.f))
    return fail(MMF_ERR_INVALID_ARG, "max_integration_distance_m must be > 0 unless the workspace is a bounding box");
  Mapper* m = new Mapper();
  const int rc = create_mapper_impl(P, m);
  if (rc != MMF_OK) {
    const std::string msg = g_err;  // the releases below must not clobber the message of the failure
    destroy_mapper(m);              // every earlier allocation (hipFree(nullptr) is a no-op)
    g_err = msg;
    return rc;
  }
  *out = m;
  return MMF_OK;
}

void destroy_mapper(Mapper* m) {
  free_layer(m->tsdf);
  free_layer(m->color);
  free_layer(m->feat);
  for (int w = 0; w < 3; ++w) {
    if (m->sc[w].flags) free_scratch(m->sc[w]);
    (void)hipFree(m->sc[w].cand_count);
    (void)hipFree(m->sc[w].alloc_ctx);
  }
  (void)hipFree(m->lazy_epoch_of);
  (void)hipFree(m->lazy_wmin);
  (void)hipFree(m->lazy_band);
  (void)hipFree(m->lazy_work);
  (void)hipFree(m->mask_tmp);
  (void)hipFree(m->masked_depth);
  (void)hipFree(m->patch_flags);
  (void)hipFree(m->inv_mask);
  (void)hipFree(m->timeline);
  (void)hipFree(m->flat.rec);
  (void)hipFree(m->flat.w);
  (void)hipFree(m->flat.count);
  (void)hipFree(m->flat_other.rec);
  (void)hipFree(m->flat_other.w);
  (void)hipFree(m->flat_other.count);
  if (m->hints) (void)hipHostFree(m->hints);
  (void)hipFree(m->kill);
  (void)hipFree(m->pub);
  (void)hipFree(m->any_kill);
  (void)hipFree(m->lb_compact);
  (void)hipFree(m->rebuild_flag);
  (void)hipFree(m->stats);
  (void)hipFree(m->synth);
  (void)hipFree(m->mesh_counts);
  (void)hipFree(m->mesh_offsets);
  (void)hipFree(m->mesh_out2);
  (void)hipFree(m->mesh_tcounts);
  (void)hipFree(m->mesh_toffsets);
  (void)hipFree(m->mesh_tout2);
  (void)hipFree(m->mi_counts);
  (void)hipFree(m->mi_chunk);
  (void)hipFree(m->mi_total);
  (void)hipFree(m->mi_offsets);
  (void)hipFree(m->mi_list);
  delete m;
}

// View grid: bounding box (in block indices, padded by one block) of the camera centre and the far
// corners of the frustum at (max distance + truncation), intersected with the workspace bounds.
int compute_view_grid(const Mapper& m, const Cam& cam, const Rigid& T_L_C, ViewGrid& vg) {
  const MapConsts& mc = m.mc;
  int lo[3], hi[3];
  if (mc.max_dist > 0.f) {
    const float s = mc.max_dist + mc.trunc;
    const bool bdiv = (mc.spec_flags & kSpecBlockDiv) != 0;
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = ifloor_h(bdiv ? T_L_C.t[a] / mc.bs : T_L_C.t[a] * mc.inv_bs);
    const float us[2] = {0.f, (float)cam.W}, vs[2] = {0.f, (float)cam.H};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) {
        float pC[3] = {s * ((us[i] - cam.cx) / cam.fx), s * ((vs[j] - cam.cy) / cam.fy), s};
        float pL[3];
        xform_h(T_L_C, pC, pL);
        for (int a = 0; a < 3; ++a) {
          int b = ifloor_h(bdiv ? pL[a] / mc.bs : pL[a] * mc.inv_bs);
          if (b < lo[a]) lo[a] = b;
          if (b > hi[a]) hi[a] = b;
        }
      }
    for (int a = 0; a < 3; ++a) {
      lo[a] -= 1;
      hi[a] += 1;
    }
  } else {
    for (int a = 0; a < 3; ++a) {
      lo[a] = mc.ws_lo[a];
      hi[a] = mc.ws_hi[a];
    }
  }
  if (mc.ws_type >= 1) {
    if (lo[2] < mc.ws_lo[2]) lo[2] = mc.ws_lo[2];
    if (hi[2] > mc.ws_hi[2]) hi[2] = mc.ws_hi[2];
  }
  if (mc.ws_type == 2) {
    for (int a = 0; a < 2; ++a) {
      if (lo[a] < mc.ws_lo[a]) lo[a] = mc.ws_lo[a];
      if (hi[a] > mc.ws_hi[a]) hi[a] = mc.ws_hi[a];
    }
  }
  vg.ox = lo[0];
  vg.oy = lo[1];
  vg.oz = lo[2];
  vg.nx = hi[0] - lo[0] + 1;
  vg.ny = hi[1] - lo[1] + 1;
  vg.nz = hi[2] - lo[2] + 1;
  if (vg.nx <= 0 || vg.ny <= 0 || vg.nz <= 0) vg.nx = vg.ny = vg.nz = 0;
  long long n = (long long)vg.nx * vg.ny * vg.nz;
  if (n > (1ll << 27)) return fail(MMF_ERR_INVALID_ARG, "view grid too large (check pose / max integration distance)");
  for (int a = 0; a < 3; ++a)
    if (lo[a] <= -kKeyOff || hi[a] >= kKeyOff) return fail(MMF_ERR_INVALID_ARG, "block index out of the 21-bit key range");
  return MMF_OK;
}

// A deferred row update runs now, as the launch it would have been, on the stream of the frame it belongs to (where the
// undeferred launch would have been enqueued: the ordering against later work is what it would have been).
int flush_rows(mmf_handle h, Mapper& m) {
  if (!m.rows_pending && !m.tail_pending) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  if (m.tail_pending) {
    m.tail_pending = false;
    m.rows_pending = true;
    ProfExt pe(h, MMF_K_FEATURE);
    launch_app_tail(m.tail, m.rows_stream, pe.a(), pe.b());
  }
  m.rows_pending = false;
  ProfExt pe(h, MMF_K_FEATURE_FLAT);
  launch_feature_flat_args(m.rows_args, m.mc, m.rows_stream, pe.a(), pe.b());
  return MMF_OK;
}

// A frame on stream s is about to host what is pending: fine on the stream it was deferred on.  On another stream the tail goes
// where it would have been enqueued, and -- it is enqueued later than it would have been, after whatever synchronisation the
// caller placed between the two streams -- stream s waits for it.
int flush_rows_for(mmf_handle h, Mapper& m, hipStream_t s);
int adopt_pending(mmf_handle h, Mapper& m, hipStream_t s) {
  if (!(m.rows_pending || m.tail_pending) || m.rows_stream == s) return MMF_OK;
  return flush_rows_for(h, m, s);
}

// the fused frame's accessor: a pending row update stays pending (the frame hosts it)
int get_mapper_keep_rows(mmf_handle h, int id, Mapper** out) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  if (id < 0 || id >= (int)h->mappers.size()) return fail(MMF_ERR_INVALID_ARG, "mapper_id out of range");
  *out = h->mappers[id];
  return MMF_OK;
}

// every other entry point: whatever it reads or writes, the map is what the calls so far made it
int get_mapper(mmf_handle h, int id, Mapper** out) {
  MMF_TRY(get_mapper_keep_rows(h, id, out));
  return flush_rows(h, **out);
}

// ... and an entry point that knows the stream it works on: the tail goes onto the stream it was deferred on (where the
// undeferred launches would be), and when that is another stream, the caller's stream waits for it -- a caller that ordered
// its stream after the frame's (event / wait) sees the finished frame, exactly as without deferral.
int flush_rows_for(mmf_handle h, Mapper& m, hipStream_t s) {
  if (!m.rows_pending && !m.tail_pending) return MMF_OK;
  hipStream_t old_stream = m.rows_stream;
  MMF_TRY(flush_rows(h, m));
  if (old_stream == s) return MMF_OK;
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ev, old_stream));
  HIP_TRY(hipStreamWaitEvent(s, ev, 0));
  HIP_TRY(hipEventDestroy(ev));
  return MMF_OK;
}

int get_mapper_on(mmf_handle h, int id, Mapper** out, void* stream) {
  MMF_TRY(get_mapper_keep_rows(h, id, out));
  return flush_rows_for(h, **out, (hipStream_t)stream);
}

// The two conditional hash-rebuild launches behind a list compaction find their request flag down on all but one frame in hundreds
// and cost 4.5 us each: they are enqueued behind every kRebuildEvery-th compaction of a mapper only.  A request stays up until it
// is served; in between the tombstones grow by what kRebuildEvery - 1 compactions deallocate (a few percent of the table).
// Only for tables large enough that 15 frames' deallocations are a small part of them (>= 2^18 entries: a churn of thousands of
// blocks per frame adds a few percent); a small table under heavy churn is checked behind every compaction, as before.
constexpr int kRebuildEvery = 16;
bool serve_rebuild_now(Mapper& m) {
  const unsigned every = (m.tsdf.d.hmask + 1u >= (1u << 18)) ? (unsigned)kRebuildEvery : 1u;
  return (m.compactions++ % every) == 0;
}

// A pending decay is applied now, as its own launches (every consumer of the map except the fused frame path).
int next_lb_tag(Mapper& m, hipStream_t s, unsigned* tag);
bool big_mode(const Mapper& m, int ncells);
int alloc_big_one(Mapper& m, const LayerDev& L, const KeySrc& ks, const Scratch& sc, int ncells, int stat_upd, int stat_new, hipStream_t s);

// ---- lazy decay of large maps (Mapper::lazy_*) ------------------------------------------------------------------------------------
int ensure_lazy(Mapper& m) {
  if (m.lazy_epoch_of) return MMF_OK;
  const size_t cap = (size_t)m.tsdf.d.cap;
  HIP_TRY(hipMalloc(&m.lazy_epoch_of, sizeof(int) * cap));
  HIP_TRY(hipMalloc(&m.lazy_wmin, sizeof(float) * cap));
  HIP_TRY(hipMalloc(&m.lazy_band, cap));
  HIP_TRY(hipMalloc(&m.lazy_work, sizeof(int) * (cap + 2)));  // {count (even frames), count (odd frames), list[cap]}
  HIP_TRY(hipMemset(m.lazy_work, 0, sizeof(int) * 2));
  m.lazy_parity = 0;
  HIP_TRY(hipMemset(m.lazy_epoch_of, 0, sizeof(int) * cap));
  HIP_TRY(hipMemset(m.lazy_wmin, 0, sizeof(float) * cap));
  HIP_TRY(hipMemset(m.lazy_band, 0, cap));
  return MMF_OK;
}

// the TSDF layer as the lazy-aware kernels see it
LayerDev lazy_view(const Mapper& m) {
  LayerDev L = m.tsdf.d;
  L.epoch = m.lazy_epoch_of;
  L.wmin = m.lazy_wmin;
  L.band = m.lazy_band;
  L.cur_epoch = m.lazy_epoch;
  L.lag_f = m.mc.decay_factor;
  return L;
}

int flush_lazy(mmf_handle h, Mapper& m, hipStream_t s) {
  if (!m.lazy_lag) return MMF_OK;
  HIP_TRY(hipSetDevice(h->device));
  ProfScope ps(h, MMF_K_DECAY, s);
  launch_lazy_catchup(lazy_view(m), s);
  m.lazy_lag = false;
  return MMF_OK;
}

void drop_lazy(Mapper& m) { m.lazy_valid = false; }

void flush_decay(mmf_handle h, Mapper& m, hipStream_t s) {
  if (!m.pending_decay) return;
  (void)flush_lazy(h, m, s);  // (the eager decay multiplies the voxels as they are)
  m.pending_decay = false;
  m.wmax_valid = false;  // the stand-alone decay does not maintain wmax
  m.lazy_valid = false;
  ProfScope ps(h, MMF_K_DECAY, s);
  if (m.tsdf.d.cap > 16384 && alloc_big_supported(m.tsdf.d) && m.mc.dealloc_decayed) {
    // large pools: the voxel pass marks the dead blocks, the scalable compaction drops them (one launch instead of a single
    // workgroup's dozens of passes over the live list)
    unsigned tag = 1;
    (void)next_lb_tag(m, s, &tag);
    launch_decay_mark(m.tsdf.d, m.mc, m.kill, m.any_kill, s);
    launch_live_compact_big(m.tsdf.d, false, m.kill, m.any_kill, m.lb_compact, tag, m.rebuild_flag, nullptr, 0.0f, 0.0f, m.tsdf.d.cap, s,
                            serve_rebuild_now(m));
    return;
  }
  launch_decay(m.tsdf.d, m.mc, m.kill, m.any_kill, s);
}

int get_mapper_ready(mmf_handle h, int id, Mapper** out, void* stream) {
  MMF_TRY(get_mapper_on(h, id, out, stream));
  MMF_TRY(flush_lazy(h, **out, (hipStream_t)stream));
  if ((*out)->pending_decay) {
    HIP_TRY(hipSetDevice(h->device));
    flush_decay(h, **out, (hipStream_t)stream);
  }
  return MMF_OK;
}

int ensure_app_layer(Mapper& m, Layer& L, size_t block_bytes, bool has_w) {
  if (L.allocated) return MMF_OK;
  MMF_TRY(alloc_layer(L, m.app_cap, block_bytes, has_w));
  if (has_w && !m.flat.rec) {
    // survivor list of a feature frame: every voxel of every block may survive.  Bounded workspaces (what the
    // reference configures) need a few tens of MB; very large pools keep phase 2 inside the gating workgroups.
    // (sub-list k takes the blocks in pool slots k, k + 64, ...: at most ceil(app_cap / 64) blocks of 512 voxels each)
    const size_t recs = (size_t)((m.app_cap + kFlatSubLists - 1) / kFlatSubLists) * kVPB * kFlatSubLists;
    // (20 B per record: 2.7 GB for a 262 144-block pool -- 1 % of the part's 288 GB)
    if (m.app_cap < (1 << 22) && recs * 20 <= ((size_t)8 << 30)) {
      HIP_TRY(hipMalloc(&m.flat.rec, sizeof(uint4) * recs));
      HIP_TRY(hipMalloc(&m.flat.w, sizeof(float) * recs));
      HIP_TRY(hipMalloc(&m.flat.count, sizeof(int) * kFlatSubLists * kFlatCountStride));
      HIP_TRY(hipMemset(m.flat.count, 0, sizeof(int) * kFlatSubLists * kFlatCountStride));
      m.flat.cap = (int)recs;
      m.flat.seg_cap = (int)(recs / kFlatSubLists);
      m.flat.hint = m.hints ? m.hints + 6 : nullptr;
    }
  }
  return attach_dense_table(m, L);
}

// second survivor list (first deferred frame): same size, same hint
int ensure_flat_other(Mapper& m) {
  if (m.flat_other.rec || !m.flat.rec) return MMF_OK;
  const size_t recs = (size_t)m.flat.cap;
  FlatList f = m.flat;
  f.rec = nullptr;
  f.w = nullptr;
  f.count = nullptr;
  if (hipMalloc(&f.rec, sizeof(uint4) * recs) != hipSuccess || hipMalloc(&f.w, sizeof(float) * recs) != hipSuccess ||
      hipMalloc(&f.count, sizeof(int) * kFlatSubLists * kFlatCountStride) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(f.rec);
    (void)hipFree(f.w);
    (void)hipFree(f.count);
    return MMF_OK;  // no room: frames keep their own row update
  }
  HIP_TRY(hipMemset(f.count, 0, sizeof(int) * kFlatSubLists * kFlatCountStride));
  m.flat_other = f;
  return MMF_OK;
}


// Is the cached synthetic depth image valid for this camera and TSDF state?
bool synth_cached(const Mapper& m, const Cam& cam, const float* T16, const float* K9) {
  return m.synth && m.synth_epoch == m.tsdf_epoch && m.synth_iw == cam.W && m.synth_ih == cam.H &&
         std::memcmp(m.synth_T, T16, sizeof(float) * 16) == 0 && std::memcmp(m.synth_K, K9, sizeof(float) * 9) == 0;
}

// Makes room for the synthetic depth image of `cam`; *need = the image must be rendered (not cached).
int synth_prepare(Mapper& m, const Cam& cam, const float* T16, const float* K9, int* Ws_out, int* Hs_out, bool* need) {
  const int sf = m.mc.st_sf;
  const int Ws = cam.W / sf, Hs = cam.H / sf;
  if (Ws <= 0 || Hs <= 0) return fail(MMF_ERR_INVALID_ARG, "image smaller than the sphere-tracing subsampling factor");
  *Ws_out = Ws;
  *Hs_out = Hs;
  *need = !synth_cached(m, cam, T16, K9);
  if (*need && Ws * Hs > m.synth_cap) {
    HIP_TRY(hipDeviceSynchronize());
    if (m.synth) HIP_TRY(hipFree(m.synth));
    m.synth = nullptr;
    HIP_TRY(hipMalloc(&m.synth, sizeof(float) * (size_t)Ws * Hs));
    m.synth_cap = Ws * Hs;
  }
  return MMF_OK;
}

// Records that m.synth now holds the image of `cam` for the current TSDF state.
void synth_commit(Mapper& m, const Cam& cam, const float* T16, const float* K9, int Ws, int Hs) {
  m.synth_W = Ws;
  m.synth_H = Hs;
  m.synth_epoch = m.tsdf_epoch;
  m.synth_iw = cam.W;
  m.synth_ih = cam.H;
  std::memcpy(m.synth_T, T16, sizeof(float) * 16);
  std::memcpy(m.synth_K, K9, sizeof(float) * 9);
}

// Sphere-trace the synthetic depth image of `cam` on stream `s` (no-op when cached).
int ensure_synth(mmf_handle h, Mapper& m, const Cam& cam, const Rigid& T_L_C, const float* T16, const float* K9, hipStream_t s) {
  int Ws, Hs;
  bool need;
  MMF_TRY(synth_prepare(m, cam, T16, K9, &Ws, &Hs, &need));
  if (!need) return MMF_OK;
  {
    ProfScope ps(h, MMF_K_SPHERE, s);
    launch_sphere_trace(m.tsdf.d, m.mc, cam, T_L_C, m.synth, Ws, Hs, s);
  }
  synth_commit(m, cam, T16, K9, Ws, Hs);
  return MMF_OK;
}

// candidate selection + allocation in an appearance layer on stream s, using scratch `which` (1 colour, 2 feature)
int app_alloc(mmf_handle h, Mapper& m, int which, Layer& L, const Cam& cam, const Rigid& T_C_L, int stat_upd, int stat_new,
              hipStream_t s) {
  MMF_TRY(ensure_scratch(m, which, m.tsdf.d.cap));
  {
    ProfScope ps(h, MMF_K_CANDIDATES, s);
    launch_app_candidates(m.tsdf.d, m.mc, cam, T_C_L, m.sc[which].flags, m.sc[which].cell_key, s);
  }
  {
    ProfScope ps(h, MMF_K_ALLOC, s);
    KeySrc ks{};
    ks.mode = 1;
    ks.n_live = m.tsdf.d.ctr;
    if (big_mode(m, m.tsdf.d.cap) && alloc_big_supported(L.d))
      MMF_TRY(alloc_big_one(m, L.d, ks, m.sc[which], m.tsdf.d.cap, stat_upd, stat_new, s));
    else
      launch_compact_alloc(L.d, ks, m.sc[which], m.tsdf.d.cap, m.stats, stat_upd, stat_new, s);
  }
  return MMF_OK;
}

// Stand-alone appearance call: synthetic depth (if not cached), then candidate selection + allocation.
int app_prepare(mmf_handle h, Mapper& m, int which, Layer& L, const Cam& cam, const Rigid& T_L_C, const Rigid& T_C_L,
                const float* T16, const float* K9, int stat_upd, int stat_new, hipStream_t s) {
  MMF_TRY(ensure_synth(h, m, cam, T_L_C, T16, K9, s));
  return app_alloc(h, m, which, L, cam, T_C_L, stat_upd, stat_new, s);
}

// bit-row scratch of the mask kernels + the masked depth image of a frame
int ensure_mask_scratch(Mapper& m, int H, int W) {
  // (the 16x16-pixel patch flags are sized by the image's ASPECT, not its area: an equal-area image of another shape needs more)
  const size_t patches = (size_t)(H / 16 + 1) * (W / 16 + 1);
  if ((size_t)H * W + 8 <= m.mask_tmp_cap && patches <= m.patch_cap) return MMF_OK;
  HIP_TRY(hipDeviceSynchronize());
  (void)hipFree(m.mask_tmp);
  (void)hipFree(m.masked_depth);
  (void)hipFree(m.patch_flags);
  m.mask_tmp = nullptr;
  m.masked_depth = nullptr;
  m.patch_flags = nullptr;
  const size_t area = (size_t)H * W + 8 > m.mask_tmp_cap ? (size_t)H * W + 8 : m.mask_tmp_cap;
  HIP_TRY(hipMalloc(&m.mask_tmp, area));
  HIP_TRY(hipMalloc(&m.masked_depth, sizeof(float) * area));
  m.patch_cap = patches > m.patch_cap ? patches : m.patch_cap;
  HIP_TRY(hipMalloc(&m.patch_flags, m.patch_cap));
  HIP_TRY(hipMemset(m.patch_flags, 0, m.patch_cap));
  m.mask_tmp_cap = area;
  return MMF_OK;
}

// Tag of the next k_alloc_tsdf launch.  Words of earlier launches stay in the buffer and are told apart by their tag, so the
// buffer is zeroed when the 30-bit counter wraps (no stale word may validate against a reused tag).
int next_pub_tag(Mapper& m, hipStream_t s, unsigned* tag) {
  m.pub_tag = (m.pub_tag + 1) & 0x3fffffffu;
  if (m.pub_tag == 0) {
    HIP_TRY(hipMemsetAsync(m.pub, 0, sizeof(u64) * pub_words(m), s));
    m.pub_tag = 1;
  }
  *tag = m.pub_tag;
  return MMF_OK;
}

// Tag of the next scalable (look-back) launch of this mapper: 22 bits, 0 never used; the words of earlier launches stay in the
// buffers and are told apart by their tag, so every look-back buffer is zeroed when the counter wraps.
int next_lb_tag(Mapper& m, hipStream_t s, unsigned* tag) {
  m.lb_tag = (m.lb_tag + 1) & 0x3fffffu;
  if (m.lb_tag == 0) {
    HIP_TRY(hipMemsetAsync(m.lb_compact, 0, sizeof(u64) * m.lb_compact_words, s));
    for (int w = 0; w < 3; ++w)
      if (m.sc[w].lb) HIP_TRY(hipMemsetAsync(m.sc[w].lb, 0, sizeof(u64) * (2 * ((size_t)m.sc_cap[w] / 1024 + 2) + 8), s));
    m.lb_tag = 1;
  }
  *tag = m.lb_tag;
  return MMF_OK;
}

// Large view grids / pools (an unbounded workspace, a bounding box of more than 16 384 blocks): the scalable single-launch
// allocation and list compaction instead of the single-workgroup roles.
bool big_mode(const Mapper& m, int ncells) { return !alloc_jobs_fusable(ncells, m.tsdf.d.cap) && alloc_big_supported(m.tsdf.d); }

// one allocation job through the scalable kernel (the replacement of the three-kernel count / scan / emit path)
int alloc_big_one(Mapper& m, const LayerDev& L, const KeySrc& ks, const Scratch& sc, int ncells, int stat_upd, int stat_new, hipStream_t s) {
  AllocJob J;
  J.L = L;
  J.ks = ks;
  J.sc = sc;
  J.ncells = ncells;
  J.stat_upd = stat_upd;
  J.stat_new = stat_new;
  MMF_TRY(next_lb_tag(m, s, &J.lb_tag));
  launch_alloc_big(&J, 1, m.stats, nullptr, s);
  return MMF_OK;
}

// Errors the device raised asynchronously (LayerDev::ctr[3] of the TSDF layer; the hard one also sets the pinned flag
// hints[7]).  Called where the API synchronises anyway (`synced`: ctr[3] has just been copied to *err_bits) and, for the pinned
// flag alone, at the start of every frame.  An error is reported ONCE and cleared: the caller decides how to go on (clear()).
int report_device_errors(mmf_handle h, Mapper& m, Layer* layer, const int* err_bits, hipStream_t s) {
  int bits = err_bits ? *err_bits : 0;
  if (m.hints && m.hints[7]) bits |= 2;
  if (!(bits & 3)) return MMF_OK;
  // only the bits that are being reported are cleared (an unreported pool exhaustion stays for mmf_num_allocated_blocks)
  if (m.hints[7]) {
    m.hints[7] = 0;
    launch_clear_bits(m.tsdf.d.ctr + 3, 2, s);
  }
  if (layer && layer->allocated) launch_clear_bits(layer->d.ctr + 3, bits & 3, s);
  if (bits & 2)
    return fail(MMF_ERR_BAD_STATE, "k_alloc_tsdf: the in-launch hand-over of new blocks failed (waiters and sweeper timed out waiting for the "
                                   "allocation workgroups of their launch); the TSDF map is incomplete -- clear() it, and set "
                                   "MMF_NO_ALLOC_TSDF=1 to keep allocation and TSDF pass as separate launches");
  return fail(MMF_ERR_POOL_EXHAUSTED, "voxel-block pool exhausted: raise BlockMemoryPoolParams.num_preallocated_blocks");
}

int untag_grid(Mapper& m, hipStream_t s) {
  if (m.grid_tagged && m.sc[0].flags) HIP_TRY(hipMemsetAsync(m.sc[0].flags, 0, (size_t)m.sc_cap[0], s));
  m.grid_tagged = false;
  m.grid_tag = 0;
  return MMF_OK;
}

int next_grid_tag(Mapper& m, hipStream_t s, int* tag) {
  m.grid_tagged = true;
  if (++m.grid_tag > 255) {
    HIP_TRY(hipMemsetAsync(m.sc[0].flags, 0, (size_t)m.sc_cap[0], s));
    if (m.patch_flags) HIP_TRY(hipMemsetAsync(m.patch_flags, 0, m.patch_cap, s));  // tagged with the same counter
    m.grid_tag = 1;
  }
  *tag = m.grid_tag;
  return MMF_OK;
}

// TSDF chain of one depth frame on stream s: raycast marking -> compaction/allocation -> TSDF update.
// A pixel is valid iff depth > min_d (min_d >= 0) and mask != 0.
int depth_chain(mmf_handle h, Mapper& m, const float* depth, const uint8_t* mask, float min_d, const Cam& cam,
                const Rigid& T_L_C, const Rigid& T_C_L, hipStream_t s) {
  ViewGrid vg;
  MMF_TRY(compute_view_grid(m, cam, T_L_C, vg));
  m.last_vg = vg;
  const int ncells = vg.nx * vg.ny * vg.nz;
  m.frames[0]++;
  m.tsdf_epoch++;
  m.touched = true;
  if (ncells == 0) {
    HIP_TRY(hipMemsetAsync(m.sc[0].cand_count, 0, sizeof(int), s));
    return MMF_OK;
  }
  MMF_TRY(ensure_scratch(m, 0, ncells));
  MMF_TRY(untag_grid(m, s));
  const int sub = m.P.raycast_subsampling < 1 ? 1 : m.P.raycast_subsampling;
  {
    ProfScope ps(h, MMF_K_RAYCAST, s);
    launch_raycast(m.mc, cam, T_L_C, depth, mask, min_d, sub, vg, m.sc[0].flags, s);
  }
  {
    ProfScope ps(h, MMF_K_ALLOC, s);
    KeySrc ks{};
    ks.mode = 0;
    ks.ox = vg.ox;
    ks.oy = vg.oy;
    ks.oz = vg.oz;
    ks.ny = vg.ny;
    ks.nz = vg.nz;
    if (big_mode(m, ncells))
      MMF_TRY(alloc_big_one(m, m.tsdf.d, ks, m.sc[0], ncells, 1, 2, s));
    else
      launch_compact_alloc(m.tsdf.d, ks, m.sc[0], ncells, m.stats, 1, 2, s);
  }
  {
    ProfScope ps(h, MMF_K_TSDF, s);
    m.wmax_valid = false;  // the stand-alone integrator does not maintain wmax
    m.lazy_valid = false;
    launch_tsdf_integrate(m.tsdf.d, m.mc, cam, T_C_L, depth, mask, min_d, m.sc[0], ncells < m.tsdf.d.cap ? ncells : m.tsdf.d.cap, s);
  }
  return MMF_OK;
}

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MMF_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return MMF_OK;
}

}  // namespace mmf_host


// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int mmf_params_size(void) { return (int)sizeof(mmf_params); }
int mmf_abi_version(void) { return MMF_ABI_VERSION; }
const char* mmf_last_error(void) { return g_err.c_str(); }

int mmf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int mmf_default_params(mmf_params* p) {
  if (!p) return fail(MMF_ERR_INVALID_ARG, "null params");
  std::memset(p, 0, sizeof(*p));
  p->voxel_size_m = 0.05f;
  p->max_integration_distance_m = 7.0f;
  p->truncation_distance_vox = 4.0f;
  p->max_weight = 5.0f;
  p->weighting_mode = 1;
  p->lin_interp_max_diff_vox = 2.0f;
  p->appearance_measurement_weight = 1.0f;
  p->appearance_max_weight = 5.0f;
  p->raycast_subsampling = 4;
  p->workspace_bounds_type = 0;
  p->tsdf_decay_factor = 0.95f;
  p->decayed_weight_threshold = 1e-3f;
  p->deallocate_decayed_blocks = 1;
  p->mesh_min_weight = 1e-4f;
  p->st_subsampling = 4;
  p->st_max_steps = 100;
  p->st_max_ray_length_m = 15.0f;
  p->st_surface_eps_vox = 0.1f;
  p->feature_channels = 768;
  p->num_preallocated_blocks = 0;
  p->expansion_factor = 1.5f;
  p->raycast_to_truncation = 1;
  p->decay_appearance_layers = 0;
  p->raycast_walk_from_camera = 0;
  p->appearance_blend_division = 0;
  p->fma_contraction = 0;
  p->block_index_by_division = 0;
  p->view_truncation_band_marking = 0;
  p->bilinear_four_weight_sum = 0;
  return MMF_OK;
}

int mmf_mapper_create(int n_mappers, const mmf_params* params, int device, mmf_handle* out) {
  if (n_mappers <= 0 || !params || !out) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_mapper_create");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(MMF_ERR_HIP, "no HIP device visible: libmmfusion has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(MMF_ERR_INVALID_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  mmf_mapper_s* h = new mmf_mapper_s();
  h->device = device;
  for (int i = 0; i < n_mappers; ++i) {
    Mapper* m = nullptr;
    int rc = create_mapper(params[i], &m);
    if (rc != MMF_OK) {
      for (Mapper* q : h->mappers) destroy_mapper(q);
      delete h;
      return rc;
    }
    h->mappers.push_back(m);
  }
  HIP_TRY(hipHostMalloc(&h->pinned, sizeof(int) * 64));
  HIP_TRY(hipDeviceSynchronize());
  *out = h;
  return MMF_OK;
}

int mmf_mapper_destroy(mmf_handle h) {
  if (!h) return MMF_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (Mapper* m : h->mappers) destroy_mapper(m);
  for (auto& r : h->prof_recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->pinned) (void)hipHostFree(h->pinned);
  delete h;
  return MMF_OK;
}

int mmf_num_mappers(mmf_handle h) { return h ? (int)h->mappers.size() : 0; }

int mmf_add_depth_frame(mmf_handle h, int mapper_id, const float* depth, const uint8_t* mask, int H, int W, const float* T16,
                        const float* K9, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  if (!depth || !T16 || !K9 || H <= 0 || W <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_add_depth_frame");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  MMF_TRY(report_device_errors(h, *m, nullptr, nullptr, s));
  Cam cam = cam_from_K(K9, W, H);
  Rigid T_L_C, T_C_L;
  rigid_from_T(T16, T_L_C);
  rigid_inverse(T_L_C, T_C_L);
  // Bounded workspace: the two-launch form of the fused frame's TSDF half -- k_front (raycast | masked depth | a pending decay's
  // deallocations) + k_alloc_tsdf (allocation | TSDF pass) -- instead of raycast, allocation, update and an eager decay pass.
  ViewGrid vg;
  MMF_TRY(compute_view_grid(*m, cam, T_L_C, vg));
  const int ncells = vg.nx * vg.ny * vg.nz;
  // (H, W >= 2: the fused launch's branch-free voxel loop clamps a 2 x 2 footprint into the image; one-row / one-column images take the chain below)
  bool fast = m->allow_merged && m->tsdf.d.dense != nullptr && ncells > 0 && alloc_jobs_fusable(ncells, m->tsdf.d.cap) && H >= 2 && W >= 2;
  MaskJob M;
  if (fast) {
    MMF_TRY(ensure_mask_scratch(*m, H, W));
    fast = make_mask_job(mask, depth, H, W, 0.0f, 0, 0, 0, H, W, nullptr, nullptr, m->mask_tmp, M);
  }
  if (!fast) {
    flush_decay(h, *m, s);
    MMF_TRY(depth_chain(h, *m, depth, mask, 0.0f, cam, T_L_C, T_C_L, s));
    return check_launch();
  }
  MMF_TRY(flush_lazy(h, *m, s));
  m->lazy_valid = false;  // (this launch does not maintain the lazy summaries)
  if (m->pending_decay && !m->wmax_valid) flush_decay(h, *m, s);  // wmax stale: the decay needs its pass over the voxels
  const bool do_decay = m->pending_decay;                          // (else: light -- decided from wmax, W *= f in the TSDF pass)
  m->pending_decay = false;
  m->last_vg = vg;
  m->frames[0]++;
  m->tsdf_epoch++;
  m->touched = true;
  const int stamp = (int)(m->tsdf_epoch & 0x3fffffff) ? (int)(m->tsdf_epoch & 0x3fffffff) : 1;
  MMF_TRY(ensure_scratch(*m, 0, ncells));
  MMF_TRY(ensure_scratch(*m, 1, m->tsdf.d.cap));
  M.masked_depth_out = m->masked_depth;
  int grid_tag = 1;
  MMF_TRY(next_grid_tag(*m, s, &grid_tag));
  const int sub = m->P.raycast_subsampling < 1 ? 1 : m->P.raycast_subsampling;
  {
    ProfScope ps(h, MMF_K_RAYCAST, s);
    const FrontArgs FA = make_front_args(m->mc, cam, T_L_C, depth, mask, 0.0f, sub, vg, m->sc[0].flags, M, do_decay ? &m->tsdf.d : nullptr,
                                         do_decay, m->kill, m->any_kill, m->tsdf.d.ctr, grid_tag);
    launch_front(&FA, 1, s);
  }
  {
    ProfScope ps(h, MMF_K_TSDF, s);
    KeySrc ks{};
    ks.mode = 0;
    ks.ox = vg.ox;
    ks.oy = vg.oy;
    ks.oz = vg.oz;
    ks.ny = vg.ny;
    ks.nz = vg.nz;
    AllocJob job;
    job.L = m->tsdf.d;
    job.ks = ks;
    job.sc = m->sc[0];
    job.ncells = ncells;
    job.stat_upd = 1;
    job.stat_new = 2;
    job.stamp = stamp;
    job.pub = m->pub;
    MMF_TRY(next_pub_tag(*m, s, &job.pub_tag));
    job.flag_value = grid_tag;
    job.host_err = m->hints + 7;
    job.debug_abandon = m->debug_abandon;
    MaskJob Mc = M;
    Mc.Hf = 0;  // no column pass: there is no feature mask to emit
    const AllocTsdfArgs TA = make_alloc_tsdf_args(job, m->stats, Mc, m->mc, cam, T_C_L, m->masked_depth, vg, m->sc[1].flags,
                                                  m->sc[1].cell_key, do_decay ? m->mc.decay_factor : 0.0f);
    launch_alloc_tsdf(&TA, 1, s);
    m->wmax_valid = true;
  }
  return check_launch();
}

int mmf_add_color_frame(mmf_handle h, int mapper_id, const uint8_t* rgb, const uint8_t* mask, int H, int W, const float* T16,
                        const float* K9, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if (!rgb || !T16 || !K9 || H <= 0 || W <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_add_color_frame");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  MMF_TRY(ensure_app_layer(*m, m->color, sizeof(uint2) * kVPB, false));
  Cam cam = cam_from_K(K9, W, H);
  Rigid T_L_C, T_C_L;
  rigid_from_T(T16, T_L_C);
  rigid_inverse(T_L_C, T_C_L);
  m->frames[1]++;
  MMF_TRY(app_prepare(h, *m, 1, m->color, cam, T_L_C, T_C_L, T16, K9, 4, -1, s));
  {
    ProfScope ps(h, MMF_K_COLOR, s);
    launch_color_integrate(m->color.d, m->mc, cam, T_C_L, rgb, mask, m->synth, m->synth_W, m->synth_H, m->sc[1], m->color.d.cap, s);
  }
  return check_launch();
}

// Validates a low-res feature source and fills the kernel-side descriptor (scales as in launch_upsample_features).
static int make_lowres(const Mapper& m, const float* lowres, int lh, int lw, int Cin, int Hf, int Wf, LowRes& lr) {
  if (!lowres || lh <= 0 || lw <= 0 || Cin <= 0 || Hf <= 1 || Wf <= 1) return fail(MMF_ERR_INVALID_ARG, "bad low-res feature map");
  if (Cin > m.P.feature_channels)
    return fail(MMF_ERR_INVALID_ARG, "low-res feature map has " + std::to_string(Cin) + " channels, the mapper holds " +
                                         std::to_string(m.P.feature_channels));
  if (Cin % 8 != 0 || ((uintptr_t)lowres & 15) != 0)
    return fail(MMF_ERR_INVALID_ARG, "the fused low-res path needs Cin % 8 == 0 and a 16-byte aligned map; "
                                     "use mmf_upsample_features + mmf_add_feature_frame otherwise");
  lr.data = lowres;
  lr.h = lh;
  lr.w = lw;
  lr.cin = Cin;
  lr.sh = (float)lh / (float)Hf;
  lr.sw = (float)lw / (float)Wf;
  return MMF_OK;
}

static int add_feature_frame_impl(mmf_handle h, int mapper_id, const void* feat, const LowRes* low, const uint8_t* mask, int Hf,
                                  int Wf, int C, const float* T16, const float* K9, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_ready(h, mapper_id, &m, stream));
  if ((!feat && !low) || !T16 || !K9 || Hf <= 1 || Wf <= 1) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_add_feature_frame");
  if (C != m->P.feature_channels)
    return fail(MMF_ERR_INVALID_ARG, "feature frame has " + std::to_string(C) + " channels, the mapper was created with " +
                                         std::to_string(m->P.feature_channels));
  if (((uintptr_t)feat & 15) != 0) return fail(MMF_ERR_INVALID_ARG, "feature frame must be 16-byte aligned");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  MMF_TRY(ensure_app_layer(*m, m->feat, sizeof(__half) * kVPB * (size_t)C, true));
  Cam cam = cam_from_K(K9, Wf, Hf);
  Rigid T_L_C, T_C_L;
  rigid_from_T(T16, T_L_C);
  rigid_inverse(T_L_C, T_C_L);
  m->frames[2]++;
  MMF_TRY(app_prepare(h, *m, 2, m->feat, cam, T_L_C, T_C_L, T16, K9, 6, 7, s));
  if (m->flat.rec) HIP_TRY(hipMemsetAsync(m->flat.count, 0, sizeof(int) * kFlatSubLists * kFlatCountStride, s));
  {
    ProfScope ps(h, MMF_K_FEATURE, s);
    launch_feature_integrate(m->feat.d, m->mc, cam, T_C_L, (const __half*)feat, mask, m->synth, m->synth_W, m->synth_H, m->sc[2],
                             m->feat.d.cap, m->stats, s, low, &m->flat);
  }
  if (!(m->mc.spec_flags & 2)) {  // (appearance_blend_division: the gating workgroups updated the rows themselves)
    ProfExt pe(h, MMF_K_FEATURE_FLAT);
    launch_feature_flat(m->feat.d, m->mc, cam, (const __half*)feat, low, m->flat, m->stats, s, pe.a(), pe.b());
  }
  return check_launch();
}

int mmf_add_feature_frame(mmf_handle h, int mapper_id, const void* feat, const uint8_t* mask, int Hf, int Wf, int C,
                          const float* T16, const float* K9, void* stream) {
  if (!feat) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_add_feature_frame");
  return add_feature_frame_impl(h, mapper_id, feat, nullptr, mask, Hf, Wf, C, T16, K9, stream);
}

int mmf_add_feature_frame_lowres(mmf_handle h, int mapper_id, const float* lowres, int lh, int lw, int Cin, const uint8_t* mask,
                                 int Hf, int Wf, const float* T16, const float* K9, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_on(h, mapper_id, &m, stream));
  LowRes lr;
  MMF_TRY(make_lowres(*m, lowres, lh, lw, Cin, Hf, Wf, lr));
  return add_feature_frame_impl(h, mapper_id, nullptr, &lr, mask, Hf, Wf, m->P.feature_channels, T16, K9, stream);
}

static int integrate_frame_impl(mmf_handle h, int mapper_id, const float* depth, const uint8_t* rgb, const void* feat,
                                const LowRes* low, const uint8_t* input_mask, int H, int W, int Hf, int Wf, int C, const float* T16,
                                const float* K9, float min_depth_m, int k_in, int k_depth, int border_percent,
                                uint8_t* depth_mask_out, uint8_t* feature_mask_out, void* stream, bool invert_mask = false,
                                bool may_defer = true) {
  // may_defer: false for the frames mmf_integrate_frame_multi hands on one by one (that call completes what is pending and does
  // not defer)
  Mapper* m;
  MMF_TRY(get_mapper_keep_rows(h, mapper_id, &m));  // (a deferred row update of the previous frame rides in this one's launch 3)
  if (!depth || !rgb || (!feat && !low) || !input_mask || !T16 || !K9 || !depth_mask_out || !feature_mask_out || H <= 1 || W <= 1 ||
      Hf <= 1 || Wf <= 1 || k_in < 0 || k_depth < 0 || !(min_depth_m >= 0.0f))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_integrate_frame");
  if (C != m->P.feature_channels)
    return fail(MMF_ERR_INVALID_ARG, "feature frame has " + std::to_string(C) + " channels, the mapper was created with " +
                                         std::to_string(m->P.feature_channels));
  if (((uintptr_t)feat & 15) != 0) return fail(MMF_ERR_INVALID_ARG, "feature frame must be 16-byte aligned");
  if (Hf != H || Wf != W)  // one synthetic depth image then serves colour and features
    return fail(MMF_ERR_INVALID_ARG, "mmf_integrate_frame needs the feature image at the depth resolution; "
                                     "use the separate add_*_frame calls otherwise");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  MMF_TRY(adopt_pending(h, *m, s));
  MMF_TRY(report_device_errors(h, *m, nullptr, nullptr, s));  // an earlier frame's hand-over failed for good: do not integrate on top
  MMF_TRY(ensure_app_layer(*m, m->color, sizeof(uint2) * kVPB, false));
  MMF_TRY(ensure_app_layer(*m, m->feat, sizeof(__half) * kVPB * (size_t)C, true));
  MMF_TRY(ensure_scratch(*m, 1, m->tsdf.d.cap));
  MMF_TRY(ensure_scratch(*m, 2, m->tsdf.d.cap));
  MMF_TRY(ensure_mask_scratch(*m, H, W));
  const Cam cam = cam_from_K(K9, W, H);
  // intrinsics of the feature image: first two rows scaled per axis (nvblox_mapping_helpers.py:229-234)
  float Kf[9];
  std::memcpy(Kf, K9, sizeof(Kf));
  const float sx = (float)Wf / (float)W, sy = (float)Hf / (float)H;
  for (int j = 0; j < 3; ++j) {
    Kf[j] *= sx;
    Kf[3 + j] *= sy;
  }
  const Cam fcam = cam_from_K(Kf, Wf, Hf);
  Rigid T_L_C, T_C_L;
  rigid_from_T(T16, T_L_C);
  rigid_inverse(T_L_C, T_C_L);

  // ---- in-order path with horizontally fused launches (default) --------------------------------------------
  //   1  raycast tiles            | mask row pass
  //   2  TSDF allocation (1 WG)   | mask column pass          (k_alloc_jobs)
  //   3  TSDF update
  //   4  candidate selection (once: colour and feature camera coincide)
  //   5  sphere trace
  //   6  colour allocation | feature allocation              (k_alloc_jobs, 2 WGs)
  //   7  colour update     | feature update                  (k_app_integrate2)
  MaskJob M;
  const bool packed = make_mask_job(input_mask, depth, H, W, min_depth_m, k_in, k_depth, border_percent, Hf, Wf, depth_mask_out,
                                    feature_mask_out, m->mask_tmp, M);
  ViewGrid vg;
  MMF_TRY(compute_view_grid(*m, cam, T_L_C, vg));
  const int ncells = vg.nx * vg.ny * vg.nz;
  // big: a view grid / pool beyond the single-workgroup allocation roles (an unbounded workspace, a very large box): the same
  // fused frame with the scalable allocation and list compaction as launches of their own (mmf_alloc_device.h)
  const bool big = packed && ncells > 0 && big_mode(*m, ncells) && alloc_big_supported(m->color.d) && alloc_big_supported(m->feat.d);
  // (spec switch appearance_blend_division: only the stand-alone appearance kernels are built with the per-channel division)
  // (and the spec switches of kSpecStandalone: block_index_by_division, view_truncation_band_marking, bilinear_four_weight_sum)
  const bool fusable = packed && ncells > 0 && (alloc_jobs_fusable(ncells, m->tsdf.d.cap) || big) && !(m->mc.spec_flags & kSpecStandalone);
  if (fusable) M.masked_depth_out = m->masked_depth;  // consumed by the TSDF update (no mask gathers there)
  if (fusable) M.invert = invert_mask ? 1 : 0;
  if (!fusable && invert_mask) {  // stand-alone kernels take the mask as it is: invert it once into scratch
    if (!m->inv_mask || m->inv_mask_cap < (size_t)H * W) {
      HIP_TRY(hipDeviceSynchronize());
      (void)hipFree(m->inv_mask);
      m->inv_mask = nullptr;
      HIP_TRY(hipMalloc(&m->inv_mask, (size_t)H * W));
      m->inv_mask_cap = (size_t)H * W;
    }
    launch_invert_mask(input_mask, m->inv_mask, (size_t)H * W, s);
    input_mask = m->inv_mask;
  }
  const bool can_host = fusable && (!big || m->allow_big_merge);  // (large maps: the merged launches of round 5 host a pending tail too)
  if (!can_host) MMF_TRY(flush_rows(h, *m));  // no launch of this frame can host it
  if (m->defer_rows && can_host && !(m->mc.spec_flags & kSpecArith)) MMF_TRY(ensure_flat_other(*m));
  if (!fusable) {
    // odd shapes / very large grids: the plain sequence of stand-alone launches
    flush_decay(h, *m, s);
    launch_frame_masks(input_mask, depth, H, W, min_depth_m, k_in, k_depth, border_percent, Hf, Wf, depth_mask_out,
                       feature_mask_out, m->mask_tmp, s);
    MMF_TRY(depth_chain(h, *m, depth, input_mask, min_depth_m, cam, T_L_C, T_C_L, s));
    m->frames[1]++;
    MMF_TRY(app_alloc(h, *m, 1, m->color, cam, T_C_L, 4, -1, s));
    MMF_TRY(ensure_synth(h, *m, cam, T_L_C, T16, K9, s));
    {
      ProfScope ps(h, MMF_K_COLOR, s);
      launch_color_integrate(m->color.d, m->mc, cam, T_C_L, rgb, depth_mask_out, m->synth, m->synth_W, m->synth_H, m->sc[1],
                             m->color.d.cap, s);
    }
    m->frames[2]++;
    MMF_TRY(app_alloc(h, *m, 2, m->feat, fcam, T_C_L, 6, 7, s));
    if (m->flat.rec) HIP_TRY(hipMemsetAsync(m->flat.count, 0, sizeof(int) * kFlatSubLists * kFlatCountStride, s));
    {
      ProfScope ps(h, MMF_K_FEATURE, s);
      launch_feature_integrate(m->feat.d, m->mc, fcam, T_C_L, (const __half*)feat, feature_mask_out, m->synth, m->synth_W,
                               m->synth_H, m->sc[2], m->feat.d.cap, m->stats, s, low, &m->flat);
    }
    if (!(m->mc.spec_flags & 2)) {
      ProfScope ps(h, MMF_K_FEATURE_FLAT, s);
      launch_feature_flat(m->feat.d, m->mc, fcam, (const __half*)feat, low, m->flat, m->stats, s);
    }
    return check_launch();
  }
  m->last_vg = vg;
  m->frames[0]++;
  m->frames[1]++;
  m->frames[2]++;
  m->tsdf_epoch++;
  m->touched = true;
  const int stamp = (int)(m->tsdf_epoch & 0x3fffffff) ? (int)(m->tsdf_epoch & 0x3fffffff) : 1;
  MMF_TRY(ensure_scratch(*m, 0, ncells));
  const int sub = m->P.raycast_subsampling < 1 ? 1 : m->P.raycast_subsampling;
  if (big && m->pending_decay && !m->wmax_valid) flush_decay(h, *m, s);  // that decay needs its voxel pass: eager, scalable compaction
  // Lazy decay (large maps): with the per-block summaries in place (a previous fused frame of this kind
  // established them) a decay costs one multiplication per live block, the TSDF pass visits only the blocks the frame integrates
  // and the near-surface blocks whose appearance flag needs their voxels, and a block's voxels catch up when it is next visited.
  // Needs the decay's deallocation rule (the compaction launch is what brings the summaries forward).
  if (big) MMF_TRY(ensure_lazy(*m));
  const bool lazy = big && m->lazy_valid && m->wmax_valid && m->mc.dealloc_decayed && m->mc.decay_thr > 0.0f && m->mc.decay_factor > 0.0f;
  if (!lazy) MMF_TRY(flush_lazy(h, *m, s));
  if (!big) m->lazy_valid = false;  // (the bounded launches do not maintain the lazy summaries)
  const bool do_decay = m->pending_decay;
  m->pending_decay = false;
  // wmax current (the previous writer of the TSDF weights was a fused frame): the decay's deallocations are decided from it by
  // a dozen workgroups and its W *= f rides in this frame's k_tsdf_pass -- no extra pass over the layer
  const bool light_decay = do_decay && m->wmax_valid;
  // Merged launch 2 (k_alloc_tsdf): bounded workspace and no voxel-pass decay in this frame -- the live list is final when
  // k_front ends, so the TSDF pass of the existing blocks runs beside the allocation workgroup instead of after it.
  // (spec switch fma_contraction: built into the un-merged launches only -- allocation, then k_tsdf_pass; the gating launch and the
  // row update of this frame, not deferred)
  const bool fma = (m->mc.spec_flags & kSpecArith) != 0;  // (fma_contraction and / or bilinear_four_weight_sum: a non-default arithmetic mode)
  const bool merged = !big && !fma && m->allow_merged && m->tsdf.d.dense != nullptr && (!do_decay || light_decay);
  // the light decay's deallocations for a large pool: the scalable compaction (decided from wmax, no voxel touched; k_front's single
  // decay workgroup would need dozens of serial passes over 10^5 list entries) -- as roles of this frame's FIRST launch, beside the
  // raycast and the mask rows (round 5; MMF_NO_BIG_MERGE=1: a launch of its own in front of it)
  const bool compact_big = big && light_decay && m->mc.dealloc_decayed;
  const bool compact_in_front = compact_big && m->allow_big_merge;
  unsigned compact_tag = 1;
  if (compact_big) {
    MMF_TRY(next_lb_tag(*m, s, &compact_tag));
    if (!compact_in_front) {
      ProfScope ps(h, MMF_K_DECAY, s);
      // (lazy: the same launch multiplies the survivors' wmax / wmin and keeps block_free current -- their voxels stay behind)
      launch_live_compact_big(lazy ? lazy_view(*m) : m->tsdf.d, true, nullptr, nullptr, m->lb_compact, compact_tag, m->rebuild_flag, nullptr,
                              m->mc.decay_factor, m->mc.decay_thr, m->tsdf.d.cap, s, serve_rebuild_now(*m));
    }
    if (lazy) {
      m->lazy_epoch++;
      m->lazy_lag = true;
    }
  }
  int grid_tag = 1;
  if (merged)
    MMF_TRY(next_grid_tag(*m, s, &grid_tag));
  else
    MMF_TRY(untag_grid(*m, s));
  {
    // raycast tiles | mask row pass | pending decay of the TSDF layer
    ProfExt pe(h, MMF_K_RAYCAST);
    const FrontArgs FA = make_front_args(m->mc, cam, T_L_C, depth, input_mask, min_depth_m, sub, vg, m->sc[0].flags, M,
                                         (do_decay && !big) ? &m->tsdf.d : nullptr, light_decay, m->kill, m->any_kill,
                                         merged ? m->tsdf.d.ctr : nullptr, grid_tag);
    if (big && m->allow_big_merge && (compact_in_front || m->tail_pending)) {
      // large map: the decay's list compaction | ... | the previous frame's colour update + feature gating (pipelined stream)
      const LayerDev Lc = lazy ? lazy_view(*m) : m->tsdf.d;
      const bool tail = m->tail_pending;
      if (tail) {
        m->tail_pending = false;
        m->rows_pending = true;
      }
      launch_front_compact_big(FA, compact_in_front ? &Lc : nullptr, m->lb_compact, compact_tag, m->rebuild_flag, m->mc.decay_factor,
                               m->mc.decay_thr, m->tsdf.d.cap, compact_in_front && serve_rebuild_now(*m), tail ? &m->tail : nullptr, s, pe.a(),
                               pe.b());
    } else if (m->tail_pending) {  // ... | the previous frame's colour update + feature gating
      m->tail_pending = false;
      m->rows_pending = true;
      launch_front_app(FA, m->tail, s, pe.a(), pe.b());
    } else {
      launch_front(&FA, 1, s, pe.a(), pe.b());
    }
  }
  KeySrc ks0{};
  ks0.mode = 0;
  ks0.ox = vg.ox;
  ks0.oy = vg.oy;
  ks0.oz = vg.oz;
  ks0.ny = vg.ny;
  ks0.nz = vg.nz;
  AllocJob job0;
  job0.L = m->tsdf.d;
  job0.ks = ks0;
  job0.sc = m->sc[0];
  job0.ncells = ncells;
  job0.stat_upd = 1;
  job0.stat_new = 2;
  job0.stamp = stamp;
  job0.timeline = m->timeline;
  if (merged) {
    // TSDF allocation | mask column pass | TSDF update + appearance-candidate flags of every live block
    ProfExt pe(h, MMF_K_TSDF);
    job0.pub = m->pub;
    MMF_TRY(next_pub_tag(*m, s, &job0.pub_tag));
    job0.flag_value = grid_tag;
    job0.host_err = m->hints + 7;
    job0.debug_abandon = m->debug_abandon;
    const AllocTsdfArgs TA = make_alloc_tsdf_args(job0, m->stats, M, m->mc, cam, T_C_L, m->masked_depth, vg, m->sc[1].flags,
                                                  m->sc[1].cell_key, light_decay ? m->mc.decay_factor : 0.0f);
    launch_alloc_tsdf(&TA, 1, s, pe.a(), pe.b());
    m->wmax_valid = true;  // refreshed for every live block
  } else {
    {
      ProfScope ps(h, MMF_K_ALLOC, s);
      if (do_decay && !light_decay && m->mc.dealloc_decayed) {  // dead blocks leave the live list before the allocation hands
                                                                 // out slots (the light decay compacted in k_front already)
        job0.kill = m->kill;
        job0.any_kill = m->any_kill;
      }
      if (big) {
        MMF_TRY(next_lb_tag(*m, s, &job0.lb_tag));
        launch_alloc_big(&job0, 1, m->stats, &M, s);  // TSDF allocation (hash lookups + CAS insertion) | mask columns
      } else {
        launch_alloc_jobs(&job0, 1, m->stats, &M, s);
      }
    }
    {
      // TSDF update of the stamped blocks + appearance-candidate flags of every live block: one pass
      ProfScope ps(h, MMF_K_TSDF, s);
      if (lazy) {
        launch_tsdf_pass_lazy(lazy_view(*m), m->mc, cam, T_C_L, m->masked_depth, stamp, m->sc[1].flags, m->sc[1].cell_key, m->lazy_work,
                              m->lazy_parity++, s);
      } else {
        // (a large map's full pass also establishes the lazy summaries: the next fused frame can decay lazily)
        launch_tsdf_pass(big ? lazy_view(*m) : m->tsdf.d, m->mc, cam, T_C_L, m->masked_depth, nullptr, 0.0f, stamp, m->sc[1].flags,
                         m->sc[1].cell_key, light_decay ? m->mc.decay_factor : 0.0f, s);
        m->lazy_valid = big;
      }
      m->wmax_valid = true;  // k_tsdf_pass refreshed it for every live block (lazy: the compaction keeps it current)
    }
  }
  {
    // sphere trace | colour allocation | feature allocation: one launch
    KeySrc ks{};
    ks.mode = 1;
    ks.n_live = m->tsdf.d.ctr;
    AllocJob jobs[2];
    jobs[0].L = m->color.d;
    jobs[0].ks = ks;
    jobs[0].sc = m->sc[1];
    jobs[0].ncells = m->tsdf.d.cap;
    jobs[0].stat_upd = 4;
    jobs[0].stat_new = -1;
    jobs[1].L = m->feat.d;
    jobs[1].ks = ks;
    jobs[1].sc = m->sc[2];
    jobs[1].sc.flags = m->sc[1].flags;  // same camera: one candidate selection serves both layers
    jobs[1].sc.cell_key = m->sc[1].cell_key;
    jobs[1].ncells = m->tsdf.d.cap;
    jobs[1].stat_upd = 6;
    jobs[1].stat_new = 7;
    jobs[1].zero_me = m->flat.count;  // survivor counters of this frame's feature update
    jobs[1].zero_n = kFlatSubLists;
    jobs[1].zero_stride = kFlatCountStride;
    int Ws, Hs;
    bool need;
    MMF_TRY(synth_prepare(*m, cam, T16, K9, &Ws, &Hs, &need));
    if (big && need && m->allow_big_merge) {
      // colour allocation | feature allocation | sphere trace: one launch (round 5)
      ProfExt pe(h, MMF_K_SPHERE);
      MMF_TRY(next_lb_tag(*m, s, &jobs[0].lb_tag));
      jobs[1].lb_tag = jobs[0].lb_tag;
      const bool rows = m->rows_pending;  // ... | the previous frame's row update (its list is the other one of the pair)
      m->rows_pending = false;
      launch_sphere_alloc_big(m->lazy_lag ? lazy_view(*m) : m->tsdf.d, m->mc, cam, T_L_C, m->synth, Ws, Hs, jobs, m->stats,
                              rows ? &m->rows_args : nullptr, s, pe.a(), pe.b());
      synth_commit(*m, cam, T16, K9, Ws, Hs);
    } else if (big) {
      MMF_TRY(flush_rows(h, *m));
      if (need) {
        ProfScope ps(h, MMF_K_SPHERE, s);
        launch_sphere_trace(m->lazy_lag ? lazy_view(*m) : m->tsdf.d, m->mc, cam, T_L_C, m->synth, Ws, Hs, s);
        synth_commit(*m, cam, T16, K9, Ws, Hs);
      }
      ProfScope ps(h, MMF_K_ALLOC, s);
      MMF_TRY(next_lb_tag(*m, s, &jobs[0].lb_tag));
      jobs[1].lb_tag = jobs[0].lb_tag;
      launch_alloc_big(jobs, 2, m->stats, nullptr, s);  // colour allocation | feature allocation
    } else if (need) {
      ProfExt pe(h, MMF_K_SPHERE);
      const SphereArgs SA = make_sphere_args(m->tsdf.d, m->mc, cam, T_L_C, m->synth, Ws, Hs, jobs, 2, m->stats);
      if (m->rows_pending) {  // ... | the previous frame's row update (its list is the other one of the pair: nobody zeroes it here)
        m->rows_pending = false;
        launch_sphere_alloc_flat(SA, m->rows_args, s, pe.a(), pe.b());
      } else {
        launch_sphere_alloc(&SA, 1, s, pe.a(), pe.b());
      }
      synth_commit(*m, cam, T16, K9, Ws, Hs);
    } else {
      MMF_TRY(flush_rows(h, *m));
      ProfScope ps(h, MMF_K_SPHERE, s);
      launch_alloc_jobs(jobs, 2, m->stats, nullptr, s);
    }
  }
  if (may_defer && m->defer_rows && (!big || m->allow_big_merge) && !fma && m->flat.rec && m->flat_other.rec) {
    // launches 4 and 5 are left to the next fused frame (roles of its launches 1 and 3) or to whatever takes the mapper first
    MMF_TRY(flush_rows(h, *m));  // (nothing hosted the previous frame's: before this frame's own)
    m->tail = make_app_tail(m->color.d, cam, rgb, depth_mask_out, m->sc[1], m->feat.d, fcam, (const __half*)feat, feature_mask_out,
                            m->sc[2], m->mc, T_C_L, m->synth, m->synth_W, m->synth_H, m->feat.d.cap, m->stats, m->flat);
    m->rows_args = make_flat_args(m->feat.d, fcam, (const __half*)feat, low, m->flat, m->stats);
    m->rows_stream = s;
    m->tail_pending = true;
    std::swap(m->flat, m->flat_other);  // the next frame fills (and its launch 3 zeroes) the other list
    return check_launch();
  }
  MMF_TRY(flush_rows(h, *m));
  {
    ProfExt pe(h, MMF_K_FEATURE);
    launch_app_integrate2(m->color.d, cam, rgb, depth_mask_out, m->sc[1], m->feat.d, fcam, (const __half*)feat, feature_mask_out,
                          m->sc[2], m->mc, T_C_L, m->synth, m->synth_W, m->synth_H, m->feat.d.cap, m->stats, s, low, &m->flat, true,
                          pe.a(), pe.b());
  }
  {
    ProfExt pe(h, MMF_K_FEATURE_FLAT);
    launch_feature_flat(m->feat.d, m->mc, fcam, (const __half*)feat, low, m->flat, m->stats, s, pe.a(), pe.b());
  }
  return check_launch();
}

// ---- two mappers, one camera frame, ONE set of launches ---------------------------------------------------------------
// The reference's control step feeds every camera frame to the static AND the dynamic mapper (nvblox_integrate with
// include_dynamic, nvblox_mapping_helpers.py:128-156): two independent maps, the same images, two different masks.  A fused
// frame is five dependent launches that each fill the chip for a few microseconds, so two frames cost twice the latency
// chain when issued one after the other.  Here the two frames are ROLES OF THE SAME FIVE LAUNCHES (k_front2, k_alloc_tsdf2,
// k_sphere_alloc2, k_app_frame2, k_feature_flat2: the second frame's workgroups follow the first's in every grid): one
// latency chain for both, one host enqueue.  Results are bit-identical to the two calls in sequence (same role code).
struct FrameIn {
  const float* depth;
  const uint8_t* rgb;
  const void* feat;
  LowRes low;
  bool has_low;
  const uint8_t* input_mask;
  int H, W, Hf, Wf, C;
  const float* T16;
  const float* K9;
  float min_depth_m;
  int k_in, k_depth, border_percent;
  uint8_t* depth_mask_out;
  uint8_t* feature_mask_out;
  bool invert_mask;
};

static int frame_in_from_desc(const Mapper& m, const mmf_frame* f, FrameIn& in) {
  if (!f || f->struct_size != (int)sizeof(mmf_frame)) return fail(MMF_ERR_INVALID_ARG, "mmf_frame: null or struct_size mismatch");
  if ((f->features_f16 != nullptr) == (f->lowres_features != nullptr))
    return fail(MMF_ERR_INVALID_ARG, "mmf_frame: give exactly one of features_f16 / lowres_features");
  in.has_low = f->lowres_features != nullptr;
  if (in.has_low) MMF_TRY(make_lowres(m, f->lowres_features, f->lowres_h, f->lowres_w, f->lowres_channels, f->Hf, f->Wf, in.low));
  in.depth = f->depth, in.rgb = f->rgb, in.feat = f->features_f16, in.input_mask = f->input_mask;
  in.H = f->H, in.W = f->W, in.Hf = f->Hf, in.Wf = f->Wf;
  in.C = in.has_low ? m.P.feature_channels : f->feature_channels;
  in.T16 = f->T_W_C, in.K9 = f->K;
  in.min_depth_m = f->min_depth_m;
  in.k_in = f->input_mask_erosion_iterations, in.k_depth = f->valid_depth_mask_erosion_iterations, in.border_percent = f->border_percent;
  in.depth_mask_out = f->depth_mask_out, in.feature_mask_out = f->feature_mask_out;
  in.invert_mask = f->invert_input_mask != 0;
  return MMF_OK;
}

static int integrate_frame_in(mmf_handle h, int mapper_id, const FrameIn& in, void* stream, bool may_defer = false) {
  return integrate_frame_impl(h, mapper_id, in.depth, in.rgb, in.feat, in.has_low ? &in.low : nullptr, in.input_mask, in.H, in.W, in.Hf,
                              in.Wf, in.C, in.T16, in.K9, in.min_depth_m, in.k_in, in.k_depth, in.border_percent, in.depth_mask_out,
                              in.feature_mask_out, stream, in.invert_mask, may_defer);
}

// Can this frame take the five-launch merged path (the only one the pair kernels implement)?  No side effects.
static bool pair_eligible(const Mapper& m, const FrameIn& in, MaskJob& M, ViewGrid& vg, Cam& cam, Rigid& T_L_C, Rigid& T_C_L) {
  if (!in.depth || !in.rgb || (!in.feat && !in.has_low) || !in.input_mask || !in.T16 || !in.K9 || !in.depth_mask_out ||
      !in.feature_mask_out || in.H <= 1 || in.W <= 1 || in.Hf != in.H || in.Wf != in.W || in.k_in < 0 || in.k_depth < 0 ||
      !(in.min_depth_m >= 0.0f) || in.C != m.P.feature_channels || ((uintptr_t)in.feat & 15) != 0)
    return false;
  if (!m.mask_tmp || (size_t)in.H * in.W + 8 > m.mask_tmp_cap || (size_t)(in.H / 16 + 1) * (in.W / 16 + 1) > m.patch_cap)
    return false;  // (first frame of a mapper / a new image shape: the single path sizes the scratch)
  if (!m.color.allocated || !m.feat.allocated || !m.flat.rec) return false;
  cam = cam_from_K(in.K9, in.W, in.H);
  rigid_from_T(in.T16, T_L_C);
  rigid_inverse(T_L_C, T_C_L);
  if (!make_mask_job(in.input_mask, in.depth, in.H, in.W, in.min_depth_m, in.k_in, in.k_depth, in.border_percent, in.Hf, in.Wf,
                     in.depth_mask_out, in.feature_mask_out, m.mask_tmp, M))
    return false;
  if (compute_view_grid(m, cam, T_L_C, vg) != MMF_OK) return false;
  const int ncells = vg.nx * vg.ny * vg.nz;
  if (ncells <= 0 || !alloc_jobs_fusable(ncells, m.tsdf.d.cap) || ncells > m.sc_cap[0] || m.tsdf.d.cap > m.sc_cap[1] ||
      m.tsdf.d.cap > m.sc_cap[2])
    return false;
  if (!m.allow_merged || !m.tsdf.d.dense || (m.mc.spec_flags & (kSpecStandalone | kSpecArith)) || m.lazy_lag) return false;
  if (m.pending_decay && !m.wmax_valid) return false;  // that decay needs its voxel pass: separate launches
  const int sf = m.mc.st_sf;
  if (in.W / sf <= 0 || in.H / sf <= 0 || (in.W / sf) * (in.H / sf) > m.synth_cap) return false;
  return true;
}

// One frame of a pair: the bookkeeping of integrate_frame_impl's merged path + the argument blocks of its five launches.
struct PairFrame {
  FrontArgs front;
  AllocTsdfArgs at;
  SphereArgs sphere;
  AppFrameArgs app;
};

static int pair_prepare(mmf_handle h, Mapper& m, const FrameIn& in, MaskJob M, const ViewGrid& vg, const Cam& cam, const Rigid& T_L_C,
                        const Rigid& T_C_L, hipStream_t s, PairFrame& F) {
  const int ncells = vg.nx * vg.ny * vg.nz;
  M.masked_depth_out = m.masked_depth;
  M.invert = in.invert_mask ? 1 : 0;
  m.last_vg = vg;
  m.frames[0]++;
  m.frames[1]++;
  m.frames[2]++;
  m.tsdf_epoch++;
  m.touched = true;
  const int stamp = (int)(m.tsdf_epoch & 0x3fffffff) ? (int)(m.tsdf_epoch & 0x3fffffff) : 1;
  const int sub = m.P.raycast_subsampling < 1 ? 1 : m.P.raycast_subsampling;
  const bool do_decay = m.pending_decay;  // (eligible: wmax is current, so it is the light form)
  m.pending_decay = false;
  int grid_tag = 1;
  MMF_TRY(next_grid_tag(m, s, &grid_tag));
  // Two frames share every launch, so the chip is oversubscribed where one frame alone fills it (2 x 1 200 ray patches): the
  // sphere tracer of a paired frame skips the patches no gate can read (for the dynamic mapper most of the image).
  const bool skip_patches = m.mc.st_sf == 4 && m.patch_flags != nullptr &&
                            (size_t)((in.W / 4 + 3) / 4) * ((in.H / 4 + 3) / 4) <= m.patch_cap;
  if (skip_patches) {
    M.patch_flags = m.patch_flags;
    M.patches_x = (in.W / 4 + 3) / 4;
    M.patches_y = (in.H / 4 + 3) / 4;
    M.patch_tag = grid_tag;
  }
  F.front = make_front_args(m.mc, cam, T_L_C, in.depth, in.input_mask, in.min_depth_m, sub, vg, m.sc[0].flags, M,
                            do_decay ? &m.tsdf.d : nullptr, do_decay, m.kill, m.any_kill, m.tsdf.d.ctr, grid_tag);
  KeySrc ks0{};
  ks0.mode = 0;
  ks0.ox = vg.ox, ks0.oy = vg.oy, ks0.oz = vg.oz, ks0.ny = vg.ny, ks0.nz = vg.nz;
  AllocJob job0;
  job0.L = m.tsdf.d;
  job0.ks = ks0;
  job0.sc = m.sc[0];
  job0.ncells = ncells;
  job0.stat_upd = 1;
  job0.stat_new = 2;
  job0.stamp = stamp;
  job0.timeline = m.timeline;
  job0.pub = m.pub;
  MMF_TRY(next_pub_tag(m, s, &job0.pub_tag));
  job0.flag_value = grid_tag;
  job0.host_err = m.hints + 7;
  job0.debug_abandon = m.debug_abandon;
  F.at = make_alloc_tsdf_args(job0, m.stats, M, m.mc, cam, T_C_L, m.masked_depth, vg, m.sc[1].flags, m.sc[1].cell_key,
                              do_decay ? m.mc.decay_factor : 0.0f);
  m.wmax_valid = true;
  m.lazy_valid = false;
  KeySrc ks{};
  ks.mode = 1;
  ks.n_live = m.tsdf.d.ctr;
  AllocJob jobs[2];
  jobs[0].L = m.color.d;
  jobs[0].ks = ks;
  jobs[0].sc = m.sc[1];
  jobs[0].ncells = m.tsdf.d.cap;
  jobs[0].stat_upd = 4;
  jobs[0].stat_new = -1;
  jobs[1].L = m.feat.d;
  jobs[1].ks = ks;
  jobs[1].sc = m.sc[2];
  jobs[1].sc.flags = m.sc[1].flags;  // same camera: one candidate selection serves both layers
  jobs[1].sc.cell_key = m.sc[1].cell_key;
  jobs[1].ncells = m.tsdf.d.cap;
  jobs[1].stat_upd = 6;
  jobs[1].stat_new = 7;
  jobs[1].zero_me = m.flat.count;
  jobs[1].zero_n = kFlatSubLists;
  jobs[1].zero_stride = kFlatCountStride;
  int Ws, Hs;
  bool need;
  MMF_TRY(synth_prepare(m, cam, in.T16, in.K9, &Ws, &Hs, &need));  // (need: the TSDF epoch moved on in this call)
  F.sphere = make_sphere_args(m.tsdf.d, m.mc, cam, T_L_C, m.synth, Ws, Hs, jobs, 2, m.stats);
  synth_commit(m, cam, in.T16, in.K9, Ws, Hs);
  if (skip_patches) {
    F.sphere.patch_flags = m.patch_flags;
    F.sphere.patch_tag = grid_tag;
    m.synth_epoch = -1;  // the image is partial: never reused by a later stand-alone appearance call
  }
  F.app = make_app_frame_args(m.color.d, cam, in.rgb, in.depth_mask_out, m.sc[1], m.feat.d, (const __half*)in.feat, in.feature_mask_out,
                              m.sc[2], m.mc, T_C_L, m.synth, m.synth_W, m.synth_H, m.feat.d.cap, m.stats,
                              in.has_low ? &in.low : nullptr, &m.flat);
  return MMF_OK;
}

int mmf_integrate_frame_multi(mmf_handle h, int n_frames, const int* mapper_ids, const mmf_frame* frames, void* stream) {
  if (!h || n_frames <= 0 || !mapper_ids || !frames) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_integrate_frame_multi");
  // deferred mode on one of the mappers: the batch entry point hosts and defers per mapper (same role code, same results)
  for (int i = 0; i < n_frames; ++i)
    if (mapper_ids[i] >= 0 && mapper_ids[i] < (int)h->mappers.size() && h->mappers[mapper_ids[i]]->defer_rows) {
      std::vector<mmf_handle> hs(n_frames, h);
      return mmf_integrate_frame_batch(n_frames, hs.data(), mapper_ids, frames, stream);
    }
  std::vector<Mapper*> ms(n_frames);
  std::vector<FrameIn> ins(n_frames);
  for (int i = 0; i < n_frames; ++i) {
    MMF_TRY(get_mapper_on(h, mapper_ids[i], &ms[i], stream));
    MMF_TRY(frame_in_from_desc(*ms[i], &frames[i], ins[i]));
    for (int j = 0; j < i; ++j)
      if (ms[j] == ms[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_integrate_frame_multi: every frame must go to a different mapper");
  }
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < n_frames) {
    bool pair = false;
    MaskJob M[2];
    ViewGrid vg[2];
    Cam cam[2];
    Rigid T_L_C[2], T_C_L[2];
    if (i + 1 < n_frames) {
      const FrameIn &a = ins[i], &b = ins[i + 1];
      // one feature source and one image size for both (k_feature_flat2 reads one image): what nvblox_integrate passes
      pair = a.feat == b.feat && a.has_low == b.has_low && (!a.has_low || (a.low.data == b.low.data && a.low.h == b.low.h && a.low.w == b.low.w)) &&
             a.H == b.H && a.W == b.W && ms[i]->mc.C == ms[i + 1]->mc.C &&
             pair_eligible(*ms[i], a, M[0], vg[0], cam[0], T_L_C[0], T_C_L[0]) &&
             pair_eligible(*ms[i + 1], b, M[1], vg[1], cam[1], T_L_C[1], T_C_L[1]);
    }
    if (!pair) {
      MMF_TRY(integrate_frame_in(h, mapper_ids[i], ins[i], stream));
      ++i;
      continue;
    }
    MMF_TRY(report_device_errors(h, *ms[i], nullptr, nullptr, s));
    MMF_TRY(report_device_errors(h, *ms[i + 1], nullptr, nullptr, s));
    PairFrame F[2];
    for (int q = 0; q < 2; ++q) MMF_TRY(pair_prepare(h, *ms[i + q], ins[i + q], M[q], vg[q], cam[q], T_L_C[q], T_C_L[q], s, F[q]));
    {
      ProfExt pe(h, MMF_K_RAYCAST);
      const FrontArgs A[2] = {F[0].front, F[1].front};
      launch_front(A, 2, s, pe.a(), pe.b());
    }
    {
      ProfExt pe(h, MMF_K_TSDF);
      const AllocTsdfArgs A[2] = {F[0].at, F[1].at};
      launch_alloc_tsdf(A, 2, s, pe.a(), pe.b());
    }
    {
      ProfExt pe(h, MMF_K_SPHERE);
      const SphereArgs A[2] = {F[0].sphere, F[1].sphere};
      launch_sphere_alloc(A, 2, s, pe.a(), pe.b());
    }
    {
      ProfExt pe(h, MMF_K_FEATURE);
      launch_app_frame2(F[0].app, F[1].app, ins[i].has_low, s, pe.a(), pe.b());
    }
    {
      ProfExt pe(h, MMF_K_FEATURE_FLAT);
      launch_feature_flat2(ms[i]->feat.d, ms[i]->mc, ms[i]->flat, ms[i]->stats, ms[i + 1]->feat.d, ms[i + 1]->mc, ms[i + 1]->flat,
                           ms[i + 1]->stats, cam[0],
                           (const __half*)ins[i].feat, ins[i].has_low ? &ins[i].low : nullptr, s, pe.a(), pe.b());
    }
    MMF_TRY(check_launch());
    i += 2;
  }
  return MMF_OK;
}

// ---- N independent frames, ONE set of launches (mmf_integrate_frame_batch) --------------------------------------------
// Generalises the pair above from {static, dynamic} mapper of one Mapper object to up to kMaxBatch frames of ANY mappers
// (different handles, different cameras, different images): replicas of the fusion path on one GPU -- data generation over
// several demos (run_isaaclab_datagen.py:213-216), several environments per GPU -- fill the chip that a single
// latency-bound chain leaves half idle.  Same role code as the single frame, so every map is bit-identical to the one the
// calls in sequence build.  Frames that cannot take the merged five-launch path are integrated on their own.
int mmf_integrate_frame_batch(int n_frames, const mmf_handle* handles, const int* mapper_ids, const mmf_frame* frames, void* stream) {
  if (n_frames <= 0 || !handles || !mapper_ids || !frames) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_integrate_frame_batch");
  std::vector<Mapper*> ms(n_frames);
  std::vector<FrameIn> ins(n_frames);
  for (int i = 0; i < n_frames; ++i) {
    if (!handles[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_integrate_frame_batch: null handle");
    if (handles[i]->device != handles[0]->device) return fail(MMF_ERR_INVALID_ARG, "mmf_integrate_frame_batch: all mappers must live on one device");
    MMF_TRY(get_mapper_keep_rows(handles[i], mapper_ids[i], &ms[i]));  // (a deferred tail rides in this call's launches)
    MMF_TRY(frame_in_from_desc(*ms[i], &frames[i], ins[i]));
    for (int j = 0; j < i; ++j)
      if (ms[j] == ms[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_integrate_frame_batch: every frame must go to a different mapper");
  }
  HIP_TRY(hipSetDevice(handles[0]->device));
  hipStream_t s = (hipStream_t)stream;
  struct Prep {
    MaskJob M;
    ViewGrid vg;
    Cam cam;
    Rigid T_L_C, T_C_L;
  };
  std::vector<Prep> pp(n_frames);
  std::vector<int> batch;  // indices of the frames that share launches
  auto flush = [&]() -> int {
    if (batch.empty()) return MMF_OK;
    if (batch.size() == 1) {
      const int i = batch[0];
      batch.clear();
      return integrate_frame_in(handles[i], mapper_ids[i], ins[i], stream, true);
    }
    const int nb = (int)batch.size();
    mmf_handle h0 = handles[batch[0]];  // (the profile of a batch is booked on the first frame's handle)
    PairFrame F[kMaxBatch];
    for (int q = 0; q < nb; ++q) {
      const int i = batch[q];
      Mapper& m = *ms[i];
      MMF_TRY(adopt_pending(handles[i], m, s));
      // (a larger image re-allocates the synthetic depth image while the frames are prepared, before launch 1 is enqueued: a pending
      // gating that reads the old one runs first.  The single-frame path prepares it after its launch 1.)
      if ((m.tail_pending || m.rows_pending) && (ins[i].W / m.mc.st_sf) * (ins[i].H / m.mc.st_sf) > m.synth_cap)
        MMF_TRY(flush_rows(handles[i], m));
      if (m.defer_rows) MMF_TRY(ensure_flat_other(m));
      MMF_TRY(report_device_errors(handles[i], m, nullptr, nullptr, s));
      MMF_TRY(pair_prepare(handles[i], m, ins[i], pp[i].M, pp[i].vg, pp[i].cam, pp[i].T_L_C, pp[i].T_C_L, s, F[q]));
    }
    {
      // ... | the pending colour update + feature gating of the previous frame of every mapper that deferred it (section 4.11)
      ProfExt pe(h0, MMF_K_RAYCAST);
      FrontArgs A[kMaxBatch];
      AppFrameArgs G[kMaxBatch];
      int ng = 0;
      for (int q = 0; q < nb; ++q) {
        A[q] = F[q].front;
        Mapper& m = *ms[batch[q]];
        if (m.tail_pending) {
          G[ng++] = app_frame_args_of(m.tail);
          m.tail_pending = false;
          m.rows_pending = true;
        }
      }
      if (ng)
        launch_front_batch_app(A, nb, G, ng, s, pe.a(), pe.b());
      else
        launch_front_batch(A, nb, s, pe.a(), pe.b());
    }
    {
      ProfExt pe(h0, MMF_K_TSDF);
      AllocTsdfArgs A[kMaxBatch];
      for (int q = 0; q < nb; ++q) A[q] = F[q].at;
      launch_alloc_tsdf_batch(A, nb, s, pe.a(), pe.b());
    }
    {
      // ... | the pending row updates (of one kind -- from feature images or from low-res maps: two instantiations --, the first
      // mapper's; the others' run on their own in front)
      SphereArgs A[kMaxBatch];
      AppArgs R[kMaxBatch];
      MapConsts RM[kMaxBatch];
      int nr = 0;
      for (int q = 0; q < nb; ++q) {
        A[q] = F[q].sphere;
        Mapper& m = *ms[batch[q]];
        if (!m.rows_pending) continue;
        if (nr && (m.rows_args.low.data != nullptr) != (R[0].low.data != nullptr)) {
          MMF_TRY(flush_rows(handles[batch[q]], m));
          continue;
        }
        R[nr] = m.rows_args;
        RM[nr++] = m.mc;
        m.rows_pending = false;
      }
      ProfExt pe(h0, MMF_K_SPHERE);
      if (nr)
        launch_sphere_alloc_batch_flat(A, nb, R, RM, nr, s, pe.a(), pe.b());
      else
        launch_sphere_alloc_batch(A, nb, s, pe.a(), pe.b());
    }
    // launches 4 and 5: of the frames whose mapper does not defer them
    AppFrameArgs A4[kMaxBatch];
    FlatFrame A5[kMaxBatch];
    int n45 = 0;
    for (int q = 0; q < nb; ++q) {
      const int i = batch[q];
      Mapper& m = *ms[i];
      if (m.defer_rows && m.flat.rec && m.flat_other.rec) {
        m.tail = app_tail_of(F[q].app, m.feat.d.cap);
        m.rows_args = make_flat_args(m.feat.d, pp[i].cam, (const __half*)ins[i].feat, ins[i].has_low ? &ins[i].low : nullptr, m.flat, m.stats);
        m.rows_stream = s;
        m.tail_pending = true;
        std::swap(m.flat, m.flat_other);
        continue;
      }
      A4[n45] = F[q].app;
      A5[n45++] = FlatFrame{m.feat.d, m.mc, m.flat, m.stats, pp[i].cam, (const __half*)ins[i].feat, ins[i].has_low ? &ins[i].low : nullptr};
    }
    if (n45) {
      {
        ProfExt pe(h0, MMF_K_FEATURE);
        launch_app_frame_batch(A4, n45, ins[batch[0]].has_low, s, pe.a(), pe.b());
      }
      ProfExt pe(h0, MMF_K_FEATURE_FLAT);
      launch_feature_flat_batch(A5, n45, s, pe.a(), pe.b());
    }
    batch.clear();
    return check_launch();
  };
  for (int i = 0; i < n_frames; ++i) {
    const bool ok = pair_eligible(*ms[i], ins[i], pp[i].M, pp[i].vg, pp[i].cam, pp[i].T_L_C, pp[i].T_C_L);
    if (!ok) {  // (its first frame, an unbounded workspace, a decay that needs its voxel pass, ...): on its own, in order
      MMF_TRY(flush());
      MMF_TRY(integrate_frame_in(handles[i], mapper_ids[i], ins[i], stream, true));
      continue;
    }
    if (!batch.empty() && (ins[batch[0]].has_low != ins[i].has_low || (int)batch.size() == kMaxBatch)) MMF_TRY(flush());
    batch.push_back(i);
  }
  return flush();
}

int mmf_integrate_frame(mmf_handle h, int mapper_id, const float* depth, const uint8_t* rgb, const void* feat,
                        const uint8_t* input_mask, int H, int W, int Hf, int Wf, int C, const float* T16, const float* K9,
                        float min_depth_m, int k_in, int k_depth, int border_percent, uint8_t* depth_mask_out,
                        uint8_t* feature_mask_out, void* stream) {
  if (!feat) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_integrate_frame");
  return integrate_frame_impl(h, mapper_id, depth, rgb, feat, nullptr, input_mask, H, W, Hf, Wf, C, T16, K9, min_depth_m, k_in,
                              k_depth, border_percent, depth_mask_out, feature_mask_out, stream);
}

int mmf_integrate_frame_lowres(mmf_handle h, int mapper_id, const float* depth, const uint8_t* rgb, const float* lowres, int lh,
                               int lw, int Cin, const uint8_t* input_mask, int H, int W, int Hf, int Wf, const float* T16,
                               const float* K9, float min_depth_m, int k_in, int k_depth, int border_percent,
                               uint8_t* depth_mask_out, uint8_t* feature_mask_out, void* stream) {
  Mapper* m;
  MMF_TRY(get_mapper_keep_rows(h, mapper_id, &m));
  LowRes lr;
  MMF_TRY(make_lowres(*m, lowres, lh, lw, Cin, Hf, Wf, lr));
  return integrate_frame_impl(h, mapper_id, depth, rgb, nullptr, &lr, input_mask, H, W, Hf, Wf, m->P.feature_channels, T16, K9,
                              min_depth_m, k_in, k_depth, border_percent, depth_mask_out, feature_mask_out, stream);
}

int mmf_integrate_frame_desc(mmf_handle h, int mapper_id, const mmf_frame* f, void* stream) {
  if (!f || f->struct_size != (int)sizeof(mmf_frame)) return fail(MMF_ERR_INVALID_ARG, "mmf_frame: null or struct_size mismatch");
  if ((f->features_f16 != nullptr) == (f->lowres_features != nullptr))
    return fail(MMF_ERR_INVALID_ARG, "mmf_frame: give exactly one of features_f16 / lowres_features");
  Mapper* m;
  MMF_TRY(get_mapper_keep_rows(h, mapper_id, &m));
  LowRes lr;
  if (f->lowres_features) MMF_TRY(make_lowres(*m, f->lowres_features, f->lowres_h, f->lowres_w, f->lowres_channels, f->Hf, f->Wf, lr));
  return integrate_frame_impl(h, mapper_id, f->depth, f->rgb, f->features_f16, f->lowres_features ? &lr : nullptr, f->input_mask, f->H,
                              f->W, f->Hf, f->Wf, f->lowres_features ? m->P.feature_channels : f->feature_channels, f->T_W_C, f->K,
                              f->min_depth_m, f->input_mask_erosion_iterations, f->valid_depth_mask_erosion_iterations,
                              f->border_percent, f->depth_mask_out, f->feature_mask_out, stream, f->invert_input_mask != 0);
}

int mmf_decay(mmf_handle h, int mapper_id, void* stream) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < (int)h->mappers.size(); ++i) {
    if (mapper_id >= 0 && i != mapper_id) continue;
    Mapper* m = h->mappers[i];
    if (!m->touched) continue;  // nothing was ever allocated in this mapper
    // Lazy: the fused frame path (mmf_integrate_frame) applies it inside its first two launches; anything else that
    // touches the map first flushes it as stand-alone launches.  Same arithmetic, same order of effects.
    flush_decay(h, *m, s);  // an earlier decay that nothing consumed
    m->pending_decay = true;
    m->tsdf_epoch++;
    if (m->P.decay_appearance_layers) {  // spec switch: the appearance weights fade too (eager; independent of the TSDF layer)
      MMF_TRY(flush_rows(h, *m));        // (after the weights of a deferred frame)
      if (m->color.allocated) launch_decay_app_weights(m->color.d, m->mc.decay_factor, false, s);
      if (m->feat.allocated) launch_decay_app_weights(m->feat.d, m->mc.decay_factor, true, s);
      m->frames[2]++;  // (cached model-input rows read feature weights)
    }
  }
  if (mapper_id >= (int)h->mappers.size()) return fail(MMF_ERR_INVALID_ARG, "mapper_id out of range");
  return check_launch();
}

int mmf_set_deferred_feature_rows(mmf_handle h, int mapper_id, int on) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  if (mapper_id >= (int)h->mappers.size()) return fail(MMF_ERR_INVALID_ARG, "mapper_id out of range");
  for (int i = 0; i < (int)h->mappers.size(); ++i) {
    if (mapper_id >= 0 && i != mapper_id) continue;
    Mapper* m = h->mappers[i];
    if (!on) MMF_TRY(flush_rows(h, *m));
    m->defer_rows = on != 0;
  }
  return check_launch();
}

int mmf_deferred_feature_rows_pending(mmf_handle h, int mapper_id) {
  Mapper* m;
  if (get_mapper_keep_rows(h, mapper_id, &m) != MMF_OK) return MMF_ERR_INVALID_ARG;
  return (m->rows_pending || m->tail_pending) ? 1 : 0;
}

int mmf_flush(mmf_handle h, int mapper_id, void* stream) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  if (mapper_id >= (int)h->mappers.size()) return fail(MMF_ERR_INVALID_ARG, "mapper_id out of range");
  HIP_TRY(hipSetDevice(h->device));
  for (int i = 0; i < (int)h->mappers.size(); ++i) {
    if (mapper_id >= 0 && i != mapper_id) continue;
    Mapper* m = h->mappers[i];
    MMF_TRY(flush_rows_for(h, *m, (hipStream_t)stream));
    flush_decay(h, *m, (hipStream_t)stream);
  }
  return check_launch();
}

int mmf_clear(mmf_handle h, int mapper_id, void* stream) {
  if (!h) return fail(MMF_ERR_INVALID_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  if (mapper_id >= (int)h->mappers.size()) return fail(MMF_ERR_INVALID_ARG, "mapper_id out of range");
  for (int i = 0; i < (int)h->mappers.size(); ++i) {
    if (mapper_id >= 0 && i != mapper_id) continue;
    Mapper* m = h->mappers[i];
    m->pending_decay = false;  // decaying blocks that are about to be dropped is a no-op
    m->rows_pending = false;   // and so is updating their appearance
    m->tail_pending = false;
    m->hints[7] = 0;           // (k_reset_layer zeroes the layer's error bits)
    launch_layer_reset(m->tsdf.d, s);
    m->wmax_valid = true;  // no live block
    m->lazy_valid = m->lazy_epoch_of != nullptr;  // (ditto: the summaries hold trivially)
    m->lazy_lag = false;
    if (m->color.allocated) launch_layer_reset(m->color.d, s);
    if (m->feat.allocated) launch_layer_reset(m->feat.d, s);
    m->tsdf_epoch++;
    m->mesh_epoch = -1;
    m->mi_epoch = -1;
    m->touched = false;
    for (int q = 0; q < 8; ++q) m->hints[q] = 0;
  }
  return check_launch();
}

}  // extern "C"
