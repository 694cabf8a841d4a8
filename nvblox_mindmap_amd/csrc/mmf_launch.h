// mmf_launch.h -- host-side launch interface between the C ABI (mmf_api.hip) and the kernel files.
#pragma once
#include "mmf_device.h"
#include "mmf_mask_device.h"

namespace mmf {

// low-res feature source of the fused up-sample + integrate path: [h,w,cin] f32 channels-last, cin % 8 == 0,
// sh = h / Hf, sw = w / Wf (float division, as k_upsample_features)
struct LowRes {
  const float* data = nullptr;
  int h = 0, w = 0, cin = 0;
  float sh = 0.0f, sw = 0.0f;
};

// Survivor list of one feature frame: voxels that passed the gate, appended by the gating workgroups and consumed by
// k_feature_flat.  rec = {is_new << 31 | slot << 9 | voxel, top-left tap pixel, wx, wy}, w = weight before the update.
// The list is kFlatSubLists sub-lists with a counter each (a block appends to sub-list (pool slot) % kFlatSubLists): one
// counter for ~800 gating workgroups serialised their appends at the L2 (~20 ns per same-address atomic: the tail of
// k_app_frame was twice its mean workgroup).  Sub-list k owns records [k * seg_cap, (k + 1) * seg_cap) and counter
// count[k * kFlatCountStride] (a cache line apart).  k_feature_flat walks the concatenation (prefix sums of the counters).
constexpr int kFlatSubLists = 64;
constexpr int kFlatCountStride = 16;
struct FlatList {
  uint4* rec = nullptr;
  float* w = nullptr;
  int* count = nullptr;  // [kFlatSubLists * kFlatCountStride] device counters, zero before the gating launch
  int* hint = nullptr;   // pinned host int: last total (sizes the next grid)
  int cap = 0;           // kFlatSubLists * seg_cap records
  int seg_cap = 0;
};

struct ViewGrid {
  int ox, oy, oz;  // block index of cell (0,0,0)
  int nx, ny, nz;
};

// ---- argument blocks of the fused-frame launches (one per frame; the *2 kernels take two) -----------------------------
struct RaycastJob {
  MapConsts mc;
  Cam cam;
  Rigid T_L_C;
  const float* depth;
  const uint8_t* mask;
  int mask_invert;  // != 0: a pixel is valid where the mask byte is 0
  float min_d;
  int sub, Wsub, Hsub;
  ViewGrid vg;
  uint8_t* flags;
  int flag_value;  // the byte a touched cell is set to: 1 (consumers clear the grid after use) or the frame's tag of a k_alloc_tsdf
                   // frame, whose consumers compare for equality and never clear (Mapper::grid_tagged)
};

struct DecayJob {
  LayerDev L;
  uint8_t* kill;
  int* any_kill;
  int n_wgs;  // 0: no decay pending
  int light;  // != 0: L.wmax is current -- ONE workgroup decides the deallocations from it and compacts the lists (no voxel
              // access); the weights themselves are multiplied by the next k_tsdf_pass
};

// Everything one frame's first launch needs; `n_wgs` = its workgroups (light decay 0/1 + rays + mask rows + decay pass).
struct FrontArgs {
  RaycastJob R;
  int n_ray_wgs;
  MaskJob M;
  DecayJob D;
  int* snap_ctr;
  int n_wgs;
};

struct TsdfFrameArgs {
  MapConsts mc;
  Cam cam;
  Rigid T_C_L;
  const float* depth;         // the frame's masked depth image
  const uint8_t* grid_flags;  // raycast flags of the view grid: a cell was touched this frame iff its byte == grid_tag
  int grid_tag;
  int ox, oy, oz, nx, ny, nz; // view grid
  uint8_t* flags_out;         // [live position] appearance-candidate flag
  u64* cell_key_out;
  float decay_f;              // > 0: pending decay's W *= f (its deallocations were made by k_front)
  const int* n_old;           // live blocks before this frame's allocation (ctr[6], published by k_front)
  const u64* pub;             // the allocation job's published new blocks
  unsigned tag;
  int* err;                   // layer error flags (ctr[3]): bit 1 = the hand-over failed for good (see new_blocks_role)
  int n_pair_wgs, n_new_wgs;
  u64* ctl;                   // control words of the hand-over, behind the records: [0] terminated / abandoned workgroup counter,
                              // [1] ranks integrated by the sweeper (diagnostics), [2 + d] {tag | first abandoned round} of waiter d
  int* host_err;              // pinned host int (may be null): set when the hand-over failed for good
  int debug_abandon;          // test hook (MMF_DEBUG_FORCE_ALLOC_TIMEOUT): 1 = odd waiters "time out" at once, 2 = so does the sweeper
};

// One frame's share of the launch: its allocation job, mask job and TSDF-pass arguments + role sizes.
struct AllocTsdfArgs {
  AllocJob J;
  long long* stats;
  MaskJob M;
  TsdfFrameArgs P;
  int alloc_wgs, mask_rows;
};

struct AppArgs {
  LayerDev L;
  Cam cam;
  Rigid T_C_L;
  const void* image;     // rgb u8 [H,W,3]  or  features f16 [Hf,Wf,C]  (null when `low` is the feature source)
  FlatList flat;         // features only: survivor list of the frame (rec == null: phase 2 stays in the gating workgroup)
  LowRes low;            // features only: low-res backbone map sampled in the kernel instead of a materialised image
  const uint8_t* mask;
  Scratch sc;
  long long* stats;  // mapper statistics (may be null)
};

// a frame's share of the sphere-trace | appearance-allocation launch
struct SphereArgs {
  LayerDev T;
  MapConsts mc;
  Cam cam;
  Rigid T_L_C;
  float* synth;
  int Ws, Hs, patches_x, n_patches;
  AllocJob J0, J1;
  int njobs;
  long long* stats;
  // optional (MaskJob::patch_flags of the same frame, 16x16-pixel patches = 4x4 ray patches at sphere-tracing subsampling 4):
  // a ray patch is traced only if it or one of its 8 neighbours holds a pixel of the depth mask -- the synthetic depth of the
  // others is never read by a gate that can pass (a voxel's synthetic-depth taps lie within 2 pixels of its mask footprint,
  // and the gate needs the whole footprint inside the mask).  The image is then PARTIAL: the caller invalidates its cache.
  const uint8_t* patch_flags = nullptr;
  int patch_tag = 0;
};

// a frame's share of the colour-update + feature-gating launch (one camera for both layers)
struct AppFrameArgs {
  AppArgs Ac, Af;
  MapConsts mc;
  const float* synth;
  int Ws, Hs;
  int nb;  // workgroups
};

// ---- N independent frames (different mappers, possibly of different Mapper objects) as roles of the SAME five launches ----
// (mmf_integrate_frame_batch).  The per-frame argument blocks travel by value, as arrays: a workgroup finds its frame by
// walking the n role sizes (scalar loads from the kernel-argument segment); the role code is the single frame's.
constexpr int kMaxBatch = 8;
struct FrontBatch {
  FrontArgs a[kMaxBatch];
  int n;
};
struct AllocTsdfBatch {
  AllocTsdfArgs a[kMaxBatch];
  int n, lead;  // lead: first workgroup of the existing-block pairs (a multiple of 8)
};
struct SphereBatch {
  SphereArgs a[kMaxBatch];
  int n;
};
struct AppFrameBatch {
  AppFrameArgs a[kMaxBatch];
  int n;
};
struct FlatBatch {
  AppArgs a[kMaxBatch];
  MapConsts mc[kMaxBatch];
  int nb[kMaxBatch];  // workgroups of each frame's list
  int lpv[kMaxBatch];
  int n;
};

// mmf_kernels_map.hip
int hinted(const int* hint, int upper);
void launch_front_batch(const FrontArgs* A, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_alloc_tsdf_batch(const AllocTsdfArgs* A, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_raycast(const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, const float* depth, const uint8_t* mask, float min_d,
                    int sub, const ViewGrid& vg, uint8_t* flags, hipStream_t s);
void launch_compact_alloc(const LayerDev& L, const KeySrc& ks, const Scratch& sc, int ncells, long long* stats, int stat_upd,
                          int stat_new, hipStream_t s);
FrontArgs make_front_args(const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, const float* depth, const uint8_t* mask, float min_d,
                          int sub, const ViewGrid& vg, uint8_t* flags, const MaskJob& M, const LayerDev* decay_layer, bool light_decay,
                          uint8_t* kill, int* any_kill, int* snap_ctr, int flag_value);
// n = 1 or 2 frames in the launch; events: stamped with the dispatch's own begin / end (extension launch)
void launch_front(const FrontArgs* A, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
AllocTsdfArgs make_alloc_tsdf_args(const AllocJob& job, long long* stats, const MaskJob& M, const MapConsts& mc, const Cam& cam,
                                   const Rigid& T_C_L, const float* masked_depth, const ViewGrid& vg, uint8_t* flags_out, u64* cell_key_out,
                                   float decay_f);
void launch_alloc_tsdf(const AllocTsdfArgs* A, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
bool alloc_jobs_fusable(int ncells0, int ncells1);
// scalable single-launch forms for large view grids / pools (mmf_alloc_device.h: alloc_big_body, live_compact_big_body)
int alloc_big_wgs(int ncells);
bool alloc_big_supported(const LayerDev& L);
void launch_alloc_big(const AllocJob* jobs, int njobs, long long* stats, const MaskJob* M, hipStream_t s);
// serve_rebuild: also enqueue the two conditional hash-rebuild launches behind it (the request flag stays up until they have run)
void launch_live_compact_big(const LayerDev& L, bool wmax, uint8_t* kill, int* any_kill, u64* lb, unsigned tag, int* rebuild, int* snap6,
                             float decay_f, float decay_thr, int live_upper, hipStream_t s, bool serve_rebuild = true);
int alloc_big_wgs(int ncells);
int alloc_big_groups(int ncells);
struct AppTail;
// large maps: [light decay's list compaction (compact != null) | raycast | mask rows | the previous frame's colour update + feature
// gating (tail != null)] as ONE launch (+ the conditional hash-rebuild pair when serve_rebuild)
void launch_front_compact_big(const FrontArgs& A, const LayerDev* compact, u64* lb, unsigned tag, int* rebuild, float decay_f, float decay_thr,
                              int live_upper, bool serve_rebuild, const AppTail* tail, hipStream_t s, hipEvent_t ev_start = nullptr,
                              hipEvent_t ev_stop = nullptr);
// large maps: [colour allocation | feature allocation | sphere trace | the previous frame's feature rows (rows != null)] as ONE launch
void launch_sphere_alloc_big(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, float* synth, int Ws, int Hs,
                             const AllocJob* jobs, long long* stats, const AppArgs* rows, hipStream_t s, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr);
void launch_alloc_jobs(const AllocJob* jobs, int njobs, long long* stats, const MaskJob* M, hipStream_t s);
void launch_tsdf_integrate(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth,
                           const uint8_t* mask, float min_d, const Scratch& sc, int max_cand, hipStream_t s);
void launch_import_index(const LayerDev& L, const int32_t* idx, int n, hipStream_t s);
void launch_import_color(const LayerDev& L, const uint8_t* rgb, const float* w, int n, hipStream_t s);
void launch_block_free_all(const LayerDev& L, const MapConsts& mc, int n, hipStream_t s);
void launch_invert_mask(const uint8_t* in, uint8_t* out, size_t n, hipStream_t s);
void launch_tsdf_pass(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth,
                      const uint8_t* mask, float min_d, int stamp, uint8_t* flags, u64* cell_key, float decay_f, hipStream_t s);
void launch_tsdf_pass_lazy(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const float* depth, int stamp,
                           uint8_t* flags, u64* cell_key, int* work, int parity, hipStream_t s);
void launch_lazy_catchup(const LayerDev& L, hipStream_t s);
void launch_decay(const LayerDev& L, const MapConsts& mc, uint8_t* kill, int* any_kill, hipStream_t s);
void launch_decay_mark(const LayerDev& L, const MapConsts& mc, uint8_t* kill, int* any_kill, hipStream_t s);
void launch_decay_app_weights(const LayerDev& L, float f, bool has_w, hipStream_t s);
void launch_layer_reset(const LayerDev& L, hipStream_t s);
void launch_clear_bits(int* word, int bits, hipStream_t s);
void launch_hash_rebuild(const LayerDev& L, hipStream_t s);
void launch_count_tombstones(const LayerDev& L, unsigned long long* out, hipStream_t s);
void launch_get_indices(const LayerDev& L, int32_t* out, int n, hipStream_t s);
void launch_gather_pool(const LayerDev& L, size_t bytes_per_block, void* out, int n, hipStream_t s);
void launch_gather_poolw(const LayerDev& L, float* out, int n, hipStream_t s);
void launch_gather_color(const LayerDev& L, uint8_t* rgb, float* w, int n, hipStream_t s);
void launch_query_tsdf(const LayerDev& L, const MapConsts& mc, const float* pts, int n, float* out, hipStream_t s);
void launch_query_feature(const LayerDev& L, const MapConsts& mc, const float* pts, int n, float* out, hipStream_t s);

// mmf_kernels_app.hip
void launch_app_candidates(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, uint8_t* flags,
                           u64* cell_key, hipStream_t s);
void launch_sphere_trace(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, float* synth, int Ws,
                         int Hs, hipStream_t s);
SphereArgs make_sphere_args(const LayerDev& tsdf, const MapConsts& mc, const Cam& cam, const Rigid& T_L_C, float* synth, int Ws, int Hs,
                            const AllocJob* jobs, int njobs, long long* stats);
void launch_sphere_alloc(const SphereArgs* A, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
AppFrameArgs make_app_frame_args(const LayerDev& Lc, const Cam& cam, const uint8_t* rgb, const uint8_t* cmask, const Scratch& csc,
                                 const LayerDev& Lf, const __half* feat, const uint8_t* fmask, const Scratch& fsc, const MapConsts& mc,
                                 const Rigid& T_C_L, const float* synth, int Ws, int Hs, int max_cand, long long* stats, const LowRes* low,
                                 const FlatList* flat);
void launch_sphere_alloc_batch(const SphereArgs* A, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_app_frame_batch(const AppFrameArgs* F, int n, bool low, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// per frame: its feature layer, constants, survivor list, statistics, feature camera and feature source (image or low-res map)
struct FlatFrame {
  LayerDev L;
  MapConsts mc;
  FlatList fl;
  long long* stats;
  Cam cam;
  const __half* feat;
  const LowRes* low;
};
void launch_feature_flat_batch(const FlatFrame* F, int n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_app_frame2(const AppFrameArgs& F0, const AppFrameArgs& F1, bool low, hipStream_t s, hipEvent_t ev_start = nullptr,
                       hipEvent_t ev_stop = nullptr);
void launch_feature_flat2(const LayerDev& L0, const MapConsts& mc0, const FlatList& fl0, long long* stats0, const LayerDev& L1,
                          const MapConsts& mc1, const FlatList& fl1, long long* stats1, const Cam& cam, const __half* feat,
                          const LowRes* lowres, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_color_integrate(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const uint8_t* rgb,
                            const uint8_t* mask, const float* synth, int Ws, int Hs, const Scratch& sc, int max_cand,
                            hipStream_t s);
void launch_feature_integrate(const LayerDev& L, const MapConsts& mc, const Cam& cam, const Rigid& T_C_L, const __half* feat,
                              const uint8_t* mask, const float* synth, int Ws, int Hs, const Scratch& sc, int max_cand,
                              long long* stats, hipStream_t s, const LowRes* low = nullptr, const FlatList* flat = nullptr);

void launch_app_integrate2(const LayerDev& Lc, const Cam& ccam, const uint8_t* rgb, const uint8_t* cmask, const Scratch& csc,
                           const LayerDev& Lf, const Cam& fcam, const __half* feat, const uint8_t* fmask, const Scratch& fsc,
                           const MapConsts& mc, const Rigid& T_C_L, const float* synth, int Ws, int Hs, int max_cand, long long* stats,
                           hipStream_t s, const LowRes* low = nullptr, const FlatList* flat = nullptr, bool same_candidates = false,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// balanced phase 2 of a feature update whose gating launch was given the survivor list `fl` (no-op without a list)
void launch_feature_flat(const LayerDev& L, const MapConsts& mc, const Cam& cam, const __half* feat, const LowRes* low,
                         const FlatList& fl, long long* stats, hipStream_t s, hipEvent_t ev_start = nullptr,
                         hipEvent_t ev_stop = nullptr);  // stats: the frame's survivor total is added to stats[8] here
// The appearance tail of a fused frame -- launch 4 (colour update + feature gating) and launch 5 (rows of the survivors) -- as
// saved argument blocks: in deferred mode (mmf_set_deferred_feature_rows) the NEXT fused frame runs them as roles of its
// launches 1 and 3, anything else that takes the mapper runs them stand-alone first.
struct AppTail {
  AppArgs Ac, Af;
  MapConsts mc;
  const float* synth;
  int Ws, Hs, max_cand;
};
AppTail make_app_tail(const LayerDev& Lc, const Cam& ccam, const uint8_t* rgb, const uint8_t* cmask, const Scratch& csc, const LayerDev& Lf,
                      const Cam& fcam, const __half* feat, const uint8_t* fmask, const Scratch& fsc, const MapConsts& mc, const Rigid& T_C_L,
                      const float* synth, int Ws, int Hs, int max_cand, long long* stats, const FlatList& flat);
int app_tail_grid(const AppTail& T);
AppFrameArgs app_frame_args_of(const AppTail& T);
AppTail app_tail_of(const AppFrameArgs& F, int max_cand);
// the batch forms of the two hosting launches: n frames + the pending tails / row updates of (some of) their mappers
void launch_front_batch_app(const FrontArgs* A, int n, const AppFrameArgs* G, int ng, hipStream_t s, hipEvent_t ev_start = nullptr,
                            hipEvent_t ev_stop = nullptr);
void launch_sphere_alloc_batch_flat(const SphereArgs* A, int n, const AppArgs* rows, const MapConsts* mcs, int nr, hipStream_t s,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_app_tail(const AppTail& T, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// raycast | mask rows | pending decay of THIS frame | colour update + feature gating of the PREVIOUS one
void launch_front_app(const FrontArgs& A, const AppTail& T, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// the row update from a saved argument block, stand-alone or as a role of the NEXT frame's sphere-trace launch
AppArgs make_flat_args(const LayerDev& L, const Cam& cam, const __half* feat, const LowRes* low, const FlatList& fl, long long* stats);
void launch_feature_flat_args(const AppArgs& Af, const MapConsts& mc, hipStream_t s, hipEvent_t ev_start = nullptr,
                              hipEvent_t ev_stop = nullptr);
void launch_sphere_alloc_flat(const SphereArgs& A, const AppArgs& F, hipStream_t s, hipEvent_t ev_start = nullptr,
                              hipEvent_t ev_stop = nullptr);

// mmf_kernels_mesh.hip
void launch_mesh_count(const LayerDev& tsdf, const MapConsts& mc, int* counts, int* offsets, int* total_host_mapped,
                       hipStream_t s);
void launch_mesh_emit(const LayerDev& tsdf, const LayerDev& feat, const MapConsts& mc, const int* offsets, int n_blocks,
                      float* verts, __half* vfeat, int V, hipStream_t s);

// map -> model inputs (mmf_model_inputs_prepare / _gather): total = device int[2] {kept rows (zero before the launch), live blocks}
void launch_mesh_keep(const LayerDev& tsdf, const LayerDev& feat, const MapConsts& mc, const float* lo, const float* hi, int used,
                      int remove_zero, int* counts, int* chunk, int* total, uint4* list, int list_cap, hipStream_t s);
void launch_model_inputs_gather(const int* counts, const int* chunk, const int* offsets, int n_blocks, const uint4* list,
                                const LayerDev& feat, int C, int used, const long long* rows, int n_take, int n_out, float* verts,
                                void* feats, bool f32, uint8_t* valid, hipStream_t s);
int model_inputs_lds_blocks();
void launch_mesh_scan_counts(const LayerDev& tsdf, const int* counts, int* offsets, int* out2, hipStream_t s);

void launch_mesh_tri_count(const LayerDev& tsdf, const MapConsts& mc, int* tcounts, int* toffsets, int* out2, hipStream_t s);
void launch_mesh_tri_emit(const LayerDev& tsdf, const LayerDev& color, const MapConsts& mc, const int* voffsets, const int* toffsets,
                          int n_blocks, int32_t* tris, uint8_t* vcolors, int V, int Tn, hipStream_t s);

// mmf_kernels_image.hip
void launch_backproject(const float* depth, const float* K, const float* T, int B, int H, int W, float* out, hipStream_t s);
int sample_inputs_scratch_floats();
void launch_sample_inputs(const float* rgb_chw, int H, int W, const float* pose7, const float* K9, uint8_t* rgb_out, float* small,
                          float* scratch, hipStream_t s, unsigned* host_flag = nullptr, unsigned seq = 0);
void launch_erode(const uint8_t* mask, uint8_t* out, uint8_t* tmp, int H, int W, int k, hipStream_t s);
void launch_feature_mask(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, int k_in, int k_depth,
                         int border_percent, int Hf, int Wf, uint8_t* out, uint8_t* tmp, hipStream_t s);
void launch_frame_masks(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, int k_in, int k_depth,
                        int border_percent, int Hf, int Wf, uint8_t* depth_mask_out, uint8_t* feature_mask_out, uint8_t* tmp,
                        hipStream_t s);
bool make_mask_job(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, int k_in, int k_depth,
                   int border_percent, int Hf, int Wf, uint8_t* depth_mask_out, uint8_t* feature_mask_out, uint8_t* tmp,
                   MaskJob& J);
void launch_depth_mask(const uint8_t* input_mask, const float* depth, int H, int W, float min_d, uint8_t* out, hipStream_t s);
void launch_upsample_features(const float* lowres, int h, int w, int Cin, __half* out, int Hf, int Wf, int Cpad, hipStream_t s,
                              bool fma = false);

}  // namespace mmf

// the policy side (FPS, the diffusion head's inference kernels, the frozen backbone's and the training step's operators): its own
// header, so that the sources the FUSION kernels are built from -- what _lib.source_hash() stamps the counter summaries with --
// do not change when a policy-side launcher is added
#include "mmf_launch_policy.h"

namespace mmf {

// diagnostics: per-workgroup timeline buffer of the fused frame kernels (mmf_trace_device.h); one setter per translation unit
int set_wg_trace_map(unsigned long long* buf, int cap);
int set_wg_trace_app(unsigned long long* buf, int cap);
int set_wg_trace_policy(unsigned long long* buf, int cap);
int set_wg_trace_policy_layer(unsigned long long* buf, int cap);

}  // namespace mmf
