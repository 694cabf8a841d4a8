// mmf_api_internal.h -- what the translation units of the C ABI share: the host-side state of a mapper, the handle, the error /
// profiling helpers and the accessors every entry point starts with.  Not installed; include/mmfusion.h is the public surface.
//   mmf_api.hip          handles, parameters, the frame entry points (add_* / integrate_frame*), decay / flush / clear
//   mmf_api_outputs.hip  mesh, map -> model inputs, block export / import, queries, diagnostics, profiling
//   mmf_api_ops.hip      stateless image ops (back-projection, masks, feature resize) and the policy-side ops (FPS, fused layers)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mmfusion.h"
#include "mmf_launch.h"

namespace mmf_host {
using namespace mmf;

int fail(int code, const std::string& msg);  // sets the thread's last-error string (mmf_last_error), returns code

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess)                                                                             \
      return fail(MMF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                   \
  } while (0)

#define MMF_TRY(expr)          \
  do {                         \
    int rc__ = (expr);         \
    if (rc__ != MMF_OK) return rc__; \
  } while (0)

struct Layer {
  LayerDev d{};
  size_t block_bytes = 0;
  bool has_w = false;
  bool allocated = false;
};

struct Mapper {
  mmf_params P{};
  MapConsts mc{};
  Layer tsdf, color, feat;
  Scratch sc[3]{};  // compaction scratch of the three chains: 0 TSDF, 1 colour, 2 feature
  int sc_cap[3] = {0, 0, 0};
  uint8_t* mask_tmp = nullptr;  // bit-row scratch of the mask kernels
  size_t mask_tmp_cap = 0;
  size_t patch_cap = 0;
  float* masked_depth = nullptr;
  uint8_t* patch_flags = nullptr;  // [(H/16+1) * (W/16+1)] tagged like the grid flags: 16x16-pixel patches that hold a depth-mask pixel
  int* hints = nullptr;  // pinned host ints the device publishes counts to: [0..2] candidates of sc[0..2], [3..5] live blocks  // depth with invalid / masked pixels zeroed (written by the mask row pass)
  uint8_t* kill = nullptr;
  int* any_kill = nullptr;
  // Raycast flags of the view grid (sc[0].flags).  Stand-alone chains set a touched cell to 1 and their allocation clears the
  // grid again (all-zero between calls).  k_alloc_tsdf frames have two readers of the flags in one launch, so nobody clears:
  // the raycast writes the frame's tag (1 .. 255) and the readers compare for equality; the buffer is zeroed when the tag
  // wraps and when a stand-alone chain follows (untag_grid).
  bool grid_tagged = false;
  int grid_tag = 0;
  bool allow_merged = true;    // false: environment MMF_NO_ALLOC_TSDF=1 at creation -- keep allocation and TSDF pass as separate launches
                               // (the reference point of the parity tests of k_alloc_tsdf)
  bool allow_big_merge = true; // false: MMF_NO_BIG_MERGE=1 at creation -- a large map's list compaction / appearance allocation stay launches of their own
  unsigned compactions = 0;    // scalable list compactions so far (serve_rebuild_now: the conditional rebuild rides on every 16th)
  u64* pub = nullptr;          // [16 + 3 * cap + 2 + kNewBlockWgs] new blocks published by the allocation workgroups of k_alloc_tsdf to
                               // their own launch + the control words of the hand-over (AllocJob::pub)
  unsigned pub_tag = 0;        // tag of the last k_alloc_tsdf launch (30 bits, incremented by those launches only; 0 is never used)
  // scalable allocation / deallocation of large grids and pools (mmf_alloc_device.h: alloc_big_body, live_compact_big_body)
  u64* lb_compact = nullptr;   // [cap / 1024 + 4] look-back words of the list compaction
  size_t lb_compact_words = 0;
  int* rebuild_flag = nullptr; // device int: the compaction's last chunk asks for a hash rebuild
  unsigned lb_tag = 0;         // tag of the last scalable launch (22 bits; 0 is never used; every look-back buffer is zeroed on wrap)
  int debug_abandon = 0;       // environment MMF_DEBUG_FORCE_ALLOC_TIMEOUT at creation (test hook of the hand-over's recovery)
  long long* stats = nullptr;  // device [MMF_NUM_STATS]
  long long frames[3] = {0, 0, 0};
  // synthetic depth + cache key
  float* synth = nullptr;
  int synth_cap = 0, synth_W = 0, synth_H = 0;
  long long synth_epoch = -1;
  float synth_T[16]{}, synth_K[9]{};
  int synth_iw = 0, synth_ih = 0;
  long long tsdf_epoch = 0;
  uint8_t* inv_mask = nullptr;  // scratch for an inverted input mask (stand-alone kernel path only)
  size_t inv_mask_cap = 0;
  long long* timeline = nullptr;  // 8 device int64: timestamps of the last TSDF allocation job (mmf_get_alloc_timeline)
  FlatList flat;               // survivor list of a feature frame (balanced phase 2); rec == null: not in use
  // Deferred row update (mmf_set_deferred_feature_rows): a fused frame leaves its last launch -- the rows of its survivor list --
  // to the NEXT fused frame, which runs it as a role of its sphere-trace launch; whatever else touches the mapper first runs it
  // as the stand-alone launch it would have been (get_mapper flushes).  Two lists: frame N's is read while frame N + 1's fills.
  FlatList flat_other;
  bool defer_rows = false;     // the caller keeps a frame's feature image valid and unchanged until the next call on this mapper
  bool rows_pending = false;
  AppArgs rows_args{};         // argument block of the pending row update (list, image, pool)
  hipStream_t rows_stream = nullptr;
  bool tail_pending = false;   // the frame's launch 4 (colour update + feature gating) is pending too: it fills rows_args' list
  AppTail tail{};
  bool pending_decay = false;  // Mapper.decay() not applied yet: consumed by the next fused frame or flushed eagerly
  bool wmax_valid = true;      // (an empty map trivially) tsdf.d.wmax holds every live block's largest weight (set by a fused frame, cleared by whatever
                               // else writes TSDF weights): a pending decay can then take the light path
  // Lazy decay of a large (hash-indexed) map, fused frames only (LayerDev::epoch): a decay is one multiplication
  // per live BLOCK (its summaries, in the list compaction) instead of one per voxel; a block's voxels catch up when the block is next
  // integrated, or all at once (flush_lazy) before anything else reads or writes voxel weights.
  int* lazy_epoch_of = nullptr;   // [cap] device: LayerDev::epoch
  float* lazy_wmin = nullptr;     // [cap] device: LayerDev::wmin
  uint8_t* lazy_band = nullptr;   // [cap] device: LayerDev::band
  int* lazy_work = nullptr;       // [cap + 2] device: work list of the lazy pass behind its two alternating counters
  int lazy_parity = 0;            // which counter the next lazy frame uses (its classify kernel zeroes the other one)
  int lazy_epoch = 0;             // decays applied lazily so far
  bool lazy_valid = false;        // the three summaries hold for every live block (established by a full pass, kept by lazy passes)
  bool lazy_lag = false;          // some block's voxels may be behind lazy_epoch
  // mesh
  int* mesh_counts = nullptr;
  int* mesh_offsets = nullptr;
  int* mesh_out2 = nullptr;
  int mesh_cap = 0, mesh_V = 0, mesh_nblocks = 0;
  long long mesh_epoch = -1;
  int* mesh_tcounts = nullptr;   // triangle counts / offsets per live block (mmf_update_mesh_topology)
  int* mesh_toffsets = nullptr;
  int* mesh_tout2 = nullptr;
  int mesh_T = 0;
  long long mesh_tepoch = -1;
  // map -> model inputs (mmf_model_inputs_prepare / _gather)
  int* mi_counts = nullptr;   // [cap] kept vertices per live block
  int* mi_chunk = nullptr;    // [cap] start of the block's chunk in mi_list
  int* mi_offsets = nullptr;  // [cap] prefix sums of mi_counts (only maps with more live blocks than the gather kernel scans in LDS)
  int* mi_total = nullptr;    // device int[2]: kept rows, live blocks
  uint4* mi_list = nullptr;   // {x, y, z, feature voxel} of every kept vertex
  int mi_list_cap = 0, mi_n = 0, mi_nblocks = 0, mi_used = 0;
  long long mi_epoch = -1, mi_feat_frames = -1;
  // last view grid (diagnostics)
  ViewGrid last_vg{};
  int app_cap = 0;
  bool touched = false;  // a depth frame was integrated since creation / clear
};

struct ProfRec {
  hipEvent_t a, b;
  int id;
};

}  // namespace mmf_host

struct mmf_mapper_s {
  int device = 0;
  std::vector<mmf_host::Mapper*> mappers;
  int* pinned = nullptr;  // host pinned scratch (16 ints)
  unsigned prof = 0;  // bitmask of kernel ids to time
  std::vector<mmf_host::ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
  unsigned prof_stride = 1;   // time every prof_stride-th eligible launch of a kernel class (mmf_profile_set_stride)
  unsigned prof_seen[MMF_NUM_KERNEL_IDS] = {0};
  double prof_ms[MMF_NUM_KERNEL_IDS] = {0};
  long long prof_n[MMF_NUM_KERNEL_IDS] = {0};
};

namespace mmf_host {

struct ProfScope {
  mmf_mapper_s* h;
  hipStream_t s;
  ProfRec r{};
  bool on;
  ProfScope(mmf_mapper_s* h_, int id, hipStream_t s_) : h(h_), s(s_), on(id >= 0 && ((h_->prof >> id) & 1u) != 0) {
    if (on) on = (h->prof_seen[id]++ % h->prof_stride) == 0;
    if (!on) return;
    r.id = id;
    r.a = take();
    r.b = take();
    (void)hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    h->prof_recs.push_back(r);
  }
  hipEvent_t take() {
    if (!h->ev_pool.empty()) {
      hipEvent_t e = h->ev_pool.back();
      h->ev_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
};

// Timing of ONE kernel with the events of an extension launch (hipExtLaunchKernelGGL): the launcher receives a() / b().
struct ProfExt {
  mmf_mapper_s* h;
  ProfRec r{};
  bool on;
  ProfExt(mmf_mapper_s* h_, int id) : h(h_), on(((h_->prof >> id) & 1u) != 0) {
    if (on) on = (h->prof_seen[id]++ % h->prof_stride) == 0;
    if (!on) return;
    ProfScope tmp(h_, -1, nullptr);  // only for its event pool accessor
    r.id = id;
    r.a = tmp.take();
    r.b = tmp.take();
  }
  ~ProfExt() {
    if (on) h->prof_recs.push_back(r);
  }
  hipEvent_t a() const { return on ? r.a : nullptr; }
  hipEvent_t b() const { return on ? r.b : nullptr; }
};

// ---- defined in mmf_api.hip ----------------------------------------------------------------------------------------------------
int check_launch();
int prof_collect(mmf_mapper_s* h);
void rigid_from_T(const float* T, Rigid& o);
void rigid_inverse(const Rigid& a, Rigid& o);
Cam cam_from_K(const float* K, int W, int H);
int get_mapper_keep_rows(mmf_handle h, int id, Mapper** out);
int get_mapper(mmf_handle h, int id, Mapper** out);                       // flushes a deferred row update first
int get_mapper_on(mmf_handle h, int id, Mapper** out, void* stream);      // ... and orders `stream` after it
int get_mapper_ready(mmf_handle h, int id, Mapper** out, void* stream);   // ... and applies a pending lazy decay
int ensure_synth(mmf_handle h, Mapper& m, const Cam& cam, const Rigid& T_L_C, const float* T16, const float* K9, hipStream_t s);
int report_device_errors(mmf_handle h, Mapper& m, Layer* layer, const int* err_bits, hipStream_t s);
int ensure_app_layer(Mapper& m, Layer& L, size_t block_bytes, bool has_w);
int attach_dense_table(const Mapper& m, Layer& L);
int flush_lazy(mmf_handle h, Mapper& m, hipStream_t s);  // voxels of a lazily decayed map brought up to date (no-op otherwise)
void drop_lazy(Mapper& m);                               // something else wrote TSDF weights: the lazy summaries no longer hold

}  // namespace mmf_host
