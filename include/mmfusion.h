/*
 * mmfusion.h -- C ABI of libmmfusion.so, the MI355X (gfx950) voxel-fusion library.
 *
 * This is the drop-in boundary for the spatial-memory hot path of NVlabs/nvblox_mindmap:
 * everything the reference reaches through the `nvblox_torch` Python module (a pybind
 * wrapper around CUDA nvblox, absent from the reference tree) plus the image-side tensor
 * ops of mindmap/image_processing that feed it.  Plain C: opaque handle, raw device
 * pointers + sizes, int return codes (0 = ok), no torch types, no exceptions.
 *
 * Conventions
 *   - every `*_dev` pointer is HIP device memory owned by the caller (a torch tensor's
 *     data_ptr()); it must stay alive until the work enqueued on `stream` completed.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     All work is enqueued asynchronously; only functions documented "synchronises" block.
 *   - poses are row-major 4x4 float32 camera->world (T_W_C) in HOST memory, intrinsics are
 *     row-major 3x3 float32 in HOST memory: exactly what the reference passes
 *     (`camera_pose.cpu(), intrinsics.cpu()`, nvblox_mapping_helpers.py:207-218,255-261).
 *   - a handle is not thread-safe (the reference drives one Mapper from one Python thread).
 *   - layer ids: 0 = TSDF, 1 = colour, 2 = feature.
 *
 * Reference interface replaced by each entry point is cited as file:line under
 * /root/reference/mindmap.
 */
#ifndef MMFUSION_H_
#define MMFUSION_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_ABI_VERSION 1

#define MMF_LAYER_TSDF 0
#define MMF_LAYER_COLOR 1
#define MMF_LAYER_FEATURE 2

/* error codes */
#define MMF_OK 0
#define MMF_ERR_INVALID_ARG 1
#define MMF_ERR_HIP 2
#define MMF_ERR_POOL_EXHAUSTED 3
#define MMF_ERR_BAD_STATE 4

typedef struct mmf_mapper_s* mmf_handle;

/*
 * Parameters of one mapper.  Mirrors the nvblox_torch parameter bags the reference fills in
 * get_nvblox_mapper (mapping/helpers/nvblox_mapping_helpers.py:40-70):
 * ProjectiveIntegratorParams, TsdfDecayIntegratorParams, ViewCalculatorParams,
 * BlockMemoryPoolParams.  Fields the reference leaves at nvblox defaults are exposed too.
 */
typedef struct {
  float voxel_size_m;
  /* ProjectiveIntegratorParams */
  float max_integration_distance_m;    /* projective_integrator_max_integration_distance_m   */
  float truncation_distance_vox;       /* projective_integrator_truncation_distance_vox      */
  float max_weight;                    /* projective_integrator_max_weight                   */
  int32_t weighting_mode;              /* upstream's WeightingFunctionType family (recalled; oracle/mmf_oracle.c tsdf_weight):
                                          0 constant, 1 inverse-square (default), 2 constant-dropoff, 3 inverse-square-dropoff,
                                          4 inverse-square-TSDF-distance-penalty, 5 linear-with-max                          */
  float lin_interp_max_diff_vox;       /* bilinear depth only where the 4 taps agree         */
  float appearance_measurement_weight; /* projective_appearance_integrator_measurement_weight */
  float appearance_max_weight;
  /* ViewCalculatorParams */
  int32_t raycast_subsampling;   /* raycast_subsampling_factor                               */
  int32_t workspace_bounds_type; /* 0 kUnbounded, 1 kHeightBounds, 2 kBoundingBox            */
  float ws_min[3];               /* workspace_bounds_min_corner_{x,y}_m, _min_height_m       */
  float ws_max[3];               /* workspace_bounds_max_corner_{x,y}_m, _max_height_m       */
  /* TsdfDecayIntegratorParams */
  float tsdf_decay_factor;
  float decayed_weight_threshold;
  int32_t deallocate_decayed_blocks;
  /* mesh */
  float mesh_min_weight;
  /* sphere tracing used by the appearance integrators' occlusion test */
  int32_t st_subsampling;
  int32_t st_max_steps;
  float st_max_ray_length_m;
  float st_surface_eps_vox;
  /* feature layer */
  int32_t feature_channels; /* nvblox_torch.constants.feature_array_num_elements()           */
  /* BlockMemoryPoolParams */
  int32_t num_preallocated_blocks; /* 0: size from the workspace bounds / built-in default   */
  float expansion_factor;          /* kept for API parity; pools grow by doubling            */
  /* spec switches (items of the integrator that are code-level choices upstream; defaults = this spec, tests/pin_report.py flips them) */
  int32_t raycast_to_truncation;   /* 1 (default): blocks in view are marked up to depth + truncation; 0: up to the depth     */
  int32_t decay_appearance_layers; /* 0 (default): decay() leaves colour / feature weights alone; 1: multiplies them as well   */
  /* the two places where the spec was arranged for the GPU, switchable so that a pin against CUDA nvblox
   * can tell which form upstream has; defaults 0 = the arranged form */
  int32_t raycast_walk_from_camera;  /* 1: a ray's block walk starts at the camera instead of where it enters the workspace bounds
                                        (the same block sets by construction) */
  int32_t appearance_blend_division; /* 1: A' = (A W + a w) / (W + w) per channel instead of one reciprocal per voxel; frames then take
                                        the stand-alone appearance kernels (the fused / pipelined launches implement the default only) */
  /* the library computes every a*b + c as two rounded operations (-ffp-contract=off); nvcc contracts by default, so CUDA nvblox's
   * voxels almost certainly hold fused multiply-adds.  1: the contraction a LLVM-family compiler makes of the spec's expressions
   * (a*b + c -> fma(a, b, c); a*x + b*y -> fma(a, x, b*y)) at the voxel projection, every bilinear sample (depth, synthetic depth,
   * colour / feature taps, the low-res feature map), the TSDF update's and the appearance blend's numerator -- same tree in the
   * oracle (fmaf).  Frames then take the un-merged launches (allocation | TSDF pass | sphere trace | gating | rows), not pipelined. */
  int32_t fma_contraction;
  /* Three more switchable recollection risks (oracle/mmf_oracle.c orc_params documents the exact arithmetic; defaults 0).  Frames
   * of a mapper with one of the first two set take the stand-alone launches (raycast | allocation | TSDF | sphere trace | colour | gating |
   * rows), not merged, not pipelined: the fused kernels implement the default forms only.  bilinear_four_weight_sum is an ARITHMETIC mode
   * like fma_contraction: its frames take the un-merged fused launches (allocation | TSDF pass | sphere trace | gating | rows), not pipelined. */
  int32_t block_index_by_division;      /* 1: block / voxel of a point by floor(p / size) instead of floor(p * (1 / size)) */
  int32_t view_truncation_band_marking; /* 1: a pixel additionally marks every block intersecting the cube of half-width `truncation`
                                           around its surface point (SURVEY.md App. A.2's second kernel) */
  int32_t bilinear_four_weight_sum;     /* 1: bilinear samples as four weighted taps (upstream's order of terms) instead of nested lerps */
} mmf_params;

/* sizeof(mmf_params) as compiled into the library (binding self-check). */
int mmf_params_size(void);
int mmf_abi_version(void);
/* nvblox defaults (the values a default-constructed nvblox_torch MapperParams carries). */
int mmf_default_params(mmf_params* out);
/* Message of the last failing call on this thread ("" if none). */
const char* mmf_last_error(void);
/* Number of visible HIP devices (0 if none); does not create a context. */
int mmf_device_count(void);

/* ---- lifetime: nvblox_torch.mapper.Mapper(voxel_sizes_m, integrator_types, mapper_parameters)
 *      nvblox_mapping_helpers.py:72-76.  `params` has `n_mappers` entries (one per mapper_id). */
int mmf_mapper_create(int n_mappers, const mmf_params* params, int device, mmf_handle* out);
int mmf_mapper_destroy(mmf_handle h);
int mmf_num_mappers(mmf_handle h); /* Mapper.num_mappers(), visualization/visualizer.py:165 */

/* ---- integration -------------------------------------------------------------------------- */
/* Mapper.add_depth_frame(depth, T_W_C, K, mask, mapper_id)   nvblox_mapping_helpers.py:207-209
 * depth: [H,W] f32 metres, <=0 invalid.  mask: [H,W] u8 (1 = integrate) or NULL.
 * Bounded workspaces: two launches (raycast | masked depth | a pending mmf_decay, then allocation | TSDF update); environment
 * MMF_NO_ALLOC_TSDF=1 at mapper creation keeps the separate launches everywhere (same results; the tests' reference point). */
int mmf_add_depth_frame(mmf_handle h, int mapper_id, const float* depth_dev, const uint8_t* mask_dev, int H, int W,
                        const float* T_W_C_host, const float* K_host, void* stream);
/* Mapper.add_color_frame(rgb, T_W_C, K, mask_frame=, mapper_id=)  nvblox_mapping_helpers.py:212-218
 * rgb: [H,W,3] u8 contiguous. */
int mmf_add_color_frame(mmf_handle h, int mapper_id, const uint8_t* rgb_dev, const uint8_t* mask_dev, int H, int W,
                        const float* T_W_C_host, const float* K_host, void* stream);
/* Mapper.add_feature_frame(feat, T_W_C, K_feat, mask, mapper_id)  nvblox_mapping_helpers.py:255-261
 * feat: [Hf,Wf,C] f16 contiguous, C == feature_channels (multiple of 8). */
int mmf_add_feature_frame(mmf_handle h, int mapper_id, const void* feat_f16_dev, const uint8_t* mask_dev, int Hf, int Wf,
                          int C, const float* T_W_C_host, const float* K_host, void* stream);
/* integrate_frame(...) of the reference as ONE call (mapping/helpers/nvblox_mapping_helpers.py:162-273):
 *   depth_mask   = input_mask & (depth > min_depth_m)                                   -> depth_mask_out [H,W] u8
 *   feature_mask = border & nearest_upsample(erode(input_mask,k_in) & erode(depth>min_depth_m,k_depth)) -> feature_mask_out [Hf,Wf] u8
 *   add_depth_frame(depth, mask=depth_mask); add_color_frame(rgb, mask=depth_mask);
 *   add_feature_frame(feat, K scaled by (Wf/W, Hf/H), mask=feature_mask)
 * Results are identical to the four separate calls.  Everything is enqueued on `stream`, in order, as five launches
 * whose independent roles are fused horizontally (raycast | mask rows | a pending mmf_decay, TSDF allocation | mask
 * columns | TSDF update + candidate flags, sphere trace | colour / feature allocation, colour update + feature gating,
 * balanced feature-row update).  Requires (Hf,Wf) == (H,W).  input_mask: [H,W] u8 (or torch bool bytes). */
int mmf_integrate_frame(mmf_handle h, int mapper_id, const float* depth_dev, const uint8_t* rgb_dev, const void* feat_f16_dev,
                        const uint8_t* input_mask_dev, int H, int W, int Hf, int Wf, int C, const float* T_W_C_host,
                        const float* K_host, float min_depth_m, int k_in, int k_depth, int border_percent,
                        uint8_t* depth_mask_out_dev, uint8_t* feature_mask_out_dev, void* stream);
/* Mapper.decay() / Mapper.clear()   isaaclab_nvblox_mapper.py:252-258.  mapper_id < 0: all mappers.
 * mmf_decay is lazy: it is applied inside the next mmf_integrate_frame* on the same mapper, or as stand-alone launches
 * by the first other call that reads or writes the map -- observable results are those of an immediate decay. */
int mmf_decay(mmf_handle h, int mapper_id, void* stream);
/* Software pipelining of consecutive fused frames (no counterpart in the reference: nvblox consumes a frame before the call
 * returns the stream to the caller; this is the same trick as the lazy mmf_decay, on the other end of the frame).  With
 * on != 0, mmf_integrate_frame / _lowres / _desc on a bounded workspace leave their last TWO launches -- the colour update +
 * feature gating and the feature-row update of the voxels that passed the gate -- pending; the next such call on the same mapper
 * and stream runs them as roles of its first (raycast) and third (sphere trace) launch: three launches per frame instead of five,
 * the appearance half of frame N beside the geometry half of frame N + 1.  ANY other entry point that takes this mapper first runs
 * them as the stand-alone launches they would have been, on the stream of the frame they belong to.  Every observable result is
 * bit-identical to the undeferred sequence.
 * CONTRACT while on: the feature image (or low-res feature map), the colour image and the two mask outputs of a frame must stay
 * allocated and UNCHANGED until the next call on this mapper (or mmf_flush) has been enqueued.  mmf_integrate_frame_batch defers
 * and hosts per mapper in the same way (full-resolution feature images), and so does mmf_integrate_frame_multi when the mode is on
 * for one of its mappers (it then takes the batch entry point's launches).  mapper_id < 0: all mappers.  Turning it off flushes.  Default: off. */
int mmf_set_deferred_feature_rows(mmf_handle h, int mapper_id, int on);
/* 1: a frame's row update is pending on the mapper (its feature image is still in use), 0: not, < 0: error code. */
int mmf_deferred_feature_rows_pending(mmf_handle h, int mapper_id);
/* Enqueues whatever is pending on the mapper (a deferred row update on its frame's stream, a lazy decay on `stream`). */
int mmf_flush(mmf_handle h, int mapper_id, void* stream);
int mmf_clear(mmf_handle h, int mapper_id, void* stream);

/* ---- map -> model input --------------------------------------------------------------------- */
/* Mapper.update_feature_mesh(mapper_id)   nvblox_output_helpers.py:49.
 * Counts the surface vertices; synchronises `stream`; *num_vertices receives V. */
int mmf_update_feature_mesh(mmf_handle h, int mapper_id, void* stream, int* num_vertices);
/* Mapper.get_feature_mesh(mapper_id).vertices() / .vertex_features()  nvblox_output_helpers.py:50-52.
 * Writes V x 3 f32 and V x C f16 into caller-allocated buffers (V from the last update). */
int mmf_get_feature_mesh(mmf_handle h, int mapper_id, float* vertices_dev, void* vertex_features_f16_dev, void* stream);
/* get_vertices_and_features(mapper, mapper_id, cfg, remove_zero_features, num_excess_features, ...)
 * nvblox_output_helpers.py:22-91, fused: update_feature_mesh + get_feature_mesh (:49-52) + strict AABB filter (:57-60) + strip
 * of the pad channels (:63-66) + all-zero-row filter (:68-74) in ONE launch that never materialises a [V, C] matrix, then the
 * rows the caller selects (sample_to_n_vertices, data_loading/vertex_sampling.py:29-82 -- the RNG draw stays with the caller,
 * on the host, like in the reference) gathered by a second launch.
 * prepare: synchronises `stream`; *num_kept = rows that pass the filters, in the order of the reference's filtered mesh
 *   (= the order of mmf_get_feature_mesh).  used_channels = C - num_excess_features (1 .. C).
 * gather: output row j < n_take is kept row rows_dev[j] (int64 ranks in [0, num_kept); null = j itself); rows
 *   n_take <= j < n_out are zero padding.  vertices_out [n_out,3] f32; features_out [n_out, used_channels] f32
 *   (features_f32 != 0) or f16, may be null (vertices only); valid_out [n_out] u8 (1 = a real row), may be null.
 *   Fails with MMF_ERR_BAD_STATE if the map changed since prepare. */
int mmf_model_inputs_prepare(mmf_handle h, int mapper_id, const float* aabb_min_host, const float* aabb_max_host, int used_channels,
                             int remove_zero_features, void* stream, int* num_kept);
int mmf_model_inputs_gather(mmf_handle h, int mapper_id, const int64_t* rows_dev, int n_take, int n_out, float* vertices_out_dev,
                            void* features_out_dev, int features_f32, uint8_t* valid_out_dev, void* stream);
/* Host only (no device work): out[0..k) = torch.randperm(n)[:k] as drawn from torch's CPU default generator
 * (sample_to_n_vertices, data_loading/vertex_sampling.py:143-145), given that generator's serialised state
 * (torch.get_rng_state(): 5056 bytes), which is advanced in place by the n - 1 draws torch.randperm(n) makes -- in O(k) swaps
 * plus the engine advance.  Returns MMF_ERR_INVALID_ARG for a state blob of another size / layout (caller then uses torch). */
int mmf_host_randperm_prefix(uint8_t* torch_cpu_rng_state, int64_t state_bytes, int64_t n, int64_t k, int64_t* out_host);
/* Host only (no device work): the loader's two reads of a training sample, each ONE copy from the page cache into the caller's
 * buffer -- a row of a pinned batch buffer (replaces, for a dataset with raw copies beside its files, the PNG inflate and the
 * decompress-everything of NvbloxMindmapDataset.__getitem__, mindmap/data_loading/dataset.py:410-415,457-468, and the
 * per-sample tensors of its collation).  `nbytes` at `offset` of `path` -> dst_host. */
int mmf_host_read_file_at(const char* path, int64_t offset, void* dst_host, int64_t nbytes);
/* Rows `rows_host[0..n_rows)` of a raw vertex-feature file (io/vertex_cache.py layout: vertices f16 [V,3] at off_vertices,
 * features f16 [V,C] at off_features): vertices_f16_out_host [n_rows,3], features_f16_out_host [n_rows,C], in the order of
 * rows_host (sample_to_n_vertices' selection, data_loading/vertex_sampling.py:143-152). */
int mmf_host_sample_vertex_file(const char* path, int64_t off_vertices, int64_t off_features, int64_t num_vertices, int64_t channels,
                                const int64_t* rows_host, int64_t n_rows, void* vertices_f16_out_host, void* features_f16_out_host);

/* ---- layer views (nvblox_torch tsdf_layer_view / feature_layer_view; paper/utils/utils.py:101-121) */
/* Number of allocated blocks; synchronises `stream`. */
int mmf_num_allocated_blocks(mmf_handle h, int mapper_id, int layer, void* stream, int* out);
/* Block indices [n,3] i32 in allocation order (n from mmf_num_allocated_blocks). */
int mmf_get_block_indices(mmf_handle h, int mapper_id, int layer, int32_t* indices_dev, int n, void* stream);
/* All TSDF blocks: out [n,8,8,8,2] f32, [...,0] = distance, [...,1] = weight. */
int mmf_get_tsdf_blocks(mmf_handle h, int mapper_id, float* out_dev, int n, void* stream);
/* All feature blocks: feats [n,8,8,8,C] f16, weights [n,8,8,8] f32. */
int mmf_get_feature_blocks(mmf_handle h, int mapper_id, void* feats_f16_dev, float* weights_dev, int n, void* stream);
/* All colour blocks: rgb [n,8,8,8,3] u8, weights [n,8,8,8] f32. */
int mmf_get_color_blocks(mmf_handle h, int mapper_id, uint8_t* rgb_dev, float* weights_dev, int n, void* stream);
/* Mapper.query_layer(QueryType, points, mapper_id)  visualization/visualizer.py:686-691.
 * layer TSDF: out [n,2] (distance, weight); layer FEATURE: out [n,C+1] f32 (features, weight). */
int mmf_query_layer(mmf_handle h, int mapper_id, int layer, const float* points_dev, int n, float* out_dev, void* stream);

/* ---- image-side ops of the path (mindmap/image_processing) ---------------------------------- */
/* backproject_depth_to_pointcloud + nan_to_num + [B,3,H,W] layout
 * (image_processing/backprojection.py:51-146).  depth [B,H,W] f32, K [B,3,3] f32 device,
 * T [B,4,4] f32 device -> out [B,3,H,W] f32.  Integer pixel coordinates (u = col, v = row). */
int mmf_backproject_depth(const float* depth_dev, const float* K_dev, const float* T_dev, int B, int H, int W,
                          float* out_dev, void* stream);
/* get_nvblox_inputs_from_sample, the per-frame image work   mapping/helpers/nvblox_input_helpers.py:57-69.
 * rgb_chw_dev (3,H,W) f32 in [0,1] -> rgb_hwc_out_dev (H,W,3) u8 = (rgb * 255) truncated; and into small_out_dev[20] f32:
 * {min(rgb), max(rgb), 1 if a NaN was seen, 0, camera pose [x,y,z,qw,qx,qy,qz], intrinsics row-major 3x3} -- what the helper
 * needs on the host (its two range assertions, quat2mat, the K the mapper takes on the host), for ONE device->host copy.
 * scratch_dev: mmf_sample_inputs_scratch_floats() floats. */
int mmf_sample_inputs_scratch_floats(void);
int mmf_sample_frame_inputs(const float* rgb_chw_dev, int H, int W, const float* pose7_dev, const float* K_dev, uint8_t* rgb_hwc_out_dev,
                            float* small_out_dev, float* scratch_dev, void* stream);
/* The same work with the 20-float record delivered ON THE HOST when the call returns (host_out20: plain host memory): the kernel
 * stores the record into coherent pinned memory the library owns and the call polls its sequence number -- no copy engine and no
 * stream synchronisation on the common path (an idle stream: ~10 us; past 300 us of polling it waits on the stream).  Replaces the
 * reference's three synchronising reads of mapping/helpers/nvblox_input_helpers.py:50-69 (pose .cpu(), rgb.min() / rgb.max()) in the
 * closed loop's per-step path (mindmap/mapping/isaaclab_nvblox_mapper.py:96-122).  Calls are serialised per process. */
int mmf_sample_frame_inputs_host(const float* rgb_chw_dev, int H, int W, const float* pose7_dev, const float* K_dev, uint8_t* rgb_hwc_out_dev,
                                 float* scratch_dev, float* host_out20, void* stream);
/* erode_mask(mask, kernel_size=3, iterations=k) (image_processing/image_mask_operations.py:16-41):
 * out = NOT dilate_{(2k+1)x(2k+1)}(NOT mask).  mask/out [H,W] u8 (0/1), tmp [H,W] u8 scratch. */
int mmf_erode_mask(const uint8_t* mask_dev, uint8_t* out_dev, uint8_t* tmp_dev, int H, int W, int iterations, void* stream);
/* Fused feature-mask algebra of integrate_frame (nvblox_mapping_helpers.py:201-253):
 * out[Hf,Wf] = border(border_percent) & nearest_upsample( erode(input_mask,k_in) & erode(depth>min_d,k_depth) ). */
int mmf_feature_mask(const uint8_t* input_mask_dev, const float* depth_dev, int H, int W, float min_depth_m, int k_in,
                     int k_depth, int border_percent, int Hf, int Wf, uint8_t* out_dev, uint8_t* tmp_dev, void* stream);
/* Both masks of integrate_frame in one call (two kernels): depth_mask_out [H,W] (may be NULL) = input_mask & (depth > min_d),
 * feature_mask_out [Hf,Wf] as mmf_feature_mask.  tmp: H*W bytes, 8-byte aligned. */
int mmf_frame_masks(const uint8_t* input_mask_dev, const float* depth_dev, int H, int W, float min_depth_m, int k_in,
                    int k_depth, int border_percent, int Hf, int Wf, uint8_t* depth_mask_out_dev, uint8_t* feature_mask_out_dev,
                    uint8_t* tmp_dev, void* stream);
/* depth_mask = input_mask & (depth > min_d)  (nvblox_mapping_helpers.py:201-204) -> u8 [H,W]. */
int mmf_depth_mask(const uint8_t* input_mask_dev, const float* depth_dev, int H, int W, float min_depth_m,
                   uint8_t* out_dev, void* stream);
/* Bilinear (align_corners=False) upsample of a low-res feature map [h,w,Cin] f32 (channels last) to
 * [Hf,Wf,Cpad] f16 with zero padding of channels Cin..Cpad-1: the resize + rearrange + pad + cast chain of
 * image_processing/feature_extraction.py:188-191,198-210 and nvblox_mapping_helpers.py:256. */
int mmf_upsample_features(const float* lowres_dev, int h, int w, int Cin, void* out_f16_dev, int Hf, int Wf, int Cpad,
                          void* stream);
/* The same with the spec switch mmf_params.fma_contraction applied to its arithmetic (source index and the three blends): what a
 * mapper created with fma_contraction = 1 computes at the taps of mmf_add_feature_frame_lowres / mmf_integrate_frame_lowres. */
int mmf_upsample_features_spec(const float* lowres_dev, int h, int w, int Cin, void* out_f16_dev, int Hf, int Wf, int Cpad,
                               int fma_contraction, void* stream);

/* ---- fused up-sample + feature integration ("K11 inside the appearance kernel") ---------- */
/* mmf_add_feature_frame without the materialised [Hf,Wf,C_pad] f16 image: the kernel samples the low-res backbone map
 * lowres_dev [lh,lw,Cin] f32 (channels last) at each tap with the arithmetic of mmf_upsample_features
 * (feature_extraction.py:188-191,198-210: bilinear align_corners=False -> zero pad to the mapper's channel count -> f16)
 * and blends as mmf_add_feature_frame does.  Results are bit-identical to mmf_upsample_features followed by
 * mmf_add_feature_frame (nvblox_mapping_helpers.py:255-261); at the reference shape this removes a 403 MB image per
 * camera frame.  Cin % 8 == 0, Cin <= feature_channels, map 16-byte aligned.  mask: u8 [Hf,Wf] or NULL. */
int mmf_add_feature_frame_lowres(mmf_handle h, int mapper_id, const float* lowres_dev, int lh, int lw, int Cin,
                                 const uint8_t* mask_dev, int Hf, int Wf, const float* T_W_C_host16, const float* K_feat_host9,
                                 void* stream);
/* mmf_integrate_frame with the same low-res feature source (Hf == H, Wf == W as there). */
int mmf_integrate_frame_lowres(mmf_handle h, int mapper_id, const float* depth_dev, const uint8_t* rgb_dev,
                               const float* lowres_dev, int lh, int lw, int Cin, const uint8_t* input_mask_dev, int H, int W, int Hf,
                               int Wf, const float* T_W_C_host16, const float* K_host9, float min_depth_m, int k_in, int k_depth,
                               int border_percent, uint8_t* depth_mask_out_dev, uint8_t* feature_mask_out_dev, void* stream);

/* Triangle connectivity and per-vertex colour of the surface mesh: Mapper.update_color_mesh / get_color_mesh
 * (visualization/visualizer.py:656-672) and FeatureMesh.triangles() (paper/utils/utils.py:84-92).  Vertices are those of
 * mmf_get_feature_mesh (same order); triangles [T,3] i32 index them, wound so that normals point into free space;
 * vertex_colors [V,3] u8 = colour voxel containing the vertex (black if unobserved; may be NULL).  The update call
 * synchronises and returns V and T; it includes mmf_update_feature_mesh. */
int mmf_update_mesh_topology(mmf_handle h, int mapper_id, void* stream, int* num_vertices, int* num_triangles);
int mmf_get_mesh_topology(mmf_handle h, int mapper_id, int32_t* triangles_dev, uint8_t* vertex_colors_dev, void* stream);

/* Mapper.load_from_file (nvblox_to_disk_helpers.py:88-93 saves with save_map; paper/teaser/convert_maps_usd.py loads):
 * replace the content of one layer by n blocks in the order given (= allocation order of the saved map, what
 * mmf_get_block_indices / mmf_get_*_blocks export): block i takes live position i.  payload / weights as exported:
 * TSDF f32 [n,512,2] (weights NULL); COLOR u8 rgb [n,512,3] + f32 w [n,512]; FEATURE f16 [n,512,C] + f32 w [n,512].
 * Indices outside the mapper's workspace bounds are an error.  Synchronises. */
int mmf_import_blocks(mmf_handle h, int mapper_id, int layer, const int32_t* idx_dev, const void* payload_dev,
                      const float* weights_dev, int n, void* stream);

/* ---- frame descriptor form of mmf_integrate_frame / mmf_integrate_frame_lowres ---------------------------------- */
/* One frame of nvblox_integrate (nvblox_mapping_helpers.py:79-159) for one mapper.  Same work and results as
 * mmf_integrate_frame; in addition `invert_input_mask` makes the call use the input mask inverted (valid where the byte
 * is 0), which is the reference's `static_mask = ~dynamic_mask` (nvblox_mapping_helpers.py:116-117) without a separate
 * pass over the mask.  Pointers are device pointers except T_W_C / K (host).  Exactly one of features_f16 [Hf,Wf,C] and
 * lowres_features [lowres_h, lowres_w, lowres_channels] f32 is non-NULL. */
typedef struct mmf_frame {
  int struct_size; /* sizeof(mmf_frame) */
  const float* depth;
  const uint8_t* rgb;
  const void* features_f16;
  const float* lowres_features;
  int lowres_h, lowres_w, lowres_channels;
  const uint8_t* input_mask;
  int invert_input_mask;
  int H, W, Hf, Wf, feature_channels;
  const float* T_W_C; /* host, 16 floats row-major */
  const float* K;     /* host, 9 floats row-major */
  float min_depth_m;
  int input_mask_erosion_iterations, valid_depth_mask_erosion_iterations, border_percent;
  uint8_t* depth_mask_out;   /* [H,W] */
  uint8_t* feature_mask_out; /* [Hf,Wf] */
} mmf_frame;
int mmf_integrate_frame_desc(mmf_handle h, int mapper_id, const mmf_frame* frame, void* stream);
/* nvblox_integrate(..., include_dynamic=True) (nvblox_mapping_helpers.py:128-156) as ONE call: frame i goes to mapper
 * mapper_ids[i] (all different -- in the reference the STATIC mapper with mask = ~dynamic_mask and the DYNAMIC mapper with
 * mask = dynamic_mask, different erosion radii, same images).  Consecutive frames that share the feature source and image
 * size are integrated as roles of the SAME five launches (the second frame's workgroups follow the first's in every grid):
 * one latency chain and one enqueue for both maps.  Results are identical to mmf_integrate_frame_desc called per frame in
 * the order given, which is also what the call does for frames that cannot be paired (odd count, a mapper whose scratch is
 * not sized yet, unbounded workspace, MMF_NO_ALLOC_TSDF=1, a decay that needs its voxel pass). */
int mmf_integrate_frame_multi(mmf_handle h, int n_frames, const int* mapper_ids, const mmf_frame* frames, void* stream);
/* N independent frames -- any mappers, of any Mapper objects on one device, each with its own camera, images and masks --
 * as roles of ONE set of five launches (up to 8 frames per set; more are issued as further sets, in order).  What several
 * replicas of the fusion path on one GPU call instead of mmf_integrate_frame_desc per replica (data generation over several
 * demos, run_isaaclab_datagen.py:213-216; several environments per GPU): a single frame's five dependent launches leave
 * half the chip idle, N frames fill it.  Every map is bit-identical to the one the calls in sequence build; a pending
 * mmf_decay of a mapper is folded in as in the single call.  handles[i] / mapper_ids[i] / frames[i] describe frame i; every
 * frame must address a different mapper. */
int mmf_integrate_frame_batch(int n_frames, const mmf_handle* handles, const int* mapper_ids, const mmf_frame* frames, void* stream);

/* ---- policy-side op ---------------------------------------------------- */
/* dgl.geometry.farthest_point_sampler(x, npoints, start_idx) (diffuser_actor/encoder.py:366-370): farthest-point
 * sampling of x [B,N,C] f32 in C-dimensional feature space, squared L2, first index on ties.
 * out_idx [B,npoints] i64.  N <= 8192, C <= 1024. */
int mmf_farthest_point_sampling(const float* x_dev, int B, int N, int C, int npoints, int start_idx, int64_t* out_idx_dev,
                                void* stream);
/* The same sampling with its scratch in a caller-owned device buffer of >= mmf_fps_workspace_bytes(B, N, C) bytes: no runtime
 * allocation inside the call, so it can be captured in a HIP graph without memory nodes (the training step's graph,
 * nvblox_mindmap_amd/training/graphed.py: a graph with hipMallocAsync nodes is executed synchronously from the host). */
int64_t mmf_fps_workspace_bytes(int B, int N, int C);
int mmf_farthest_point_sampling_ws(const float* x_dev, int B, int N, int C, int npoints, int start_idx, int64_t* out_idx_dev,
                                   void* workspace_dev, int64_t workspace_bytes, void* stream);

/* Inference-side fused ops of the diffusion head (no autograd; the training path keeps the composite torch ops).
 * mmf_rotary_apply: apply_rotary of diffuser_actor/position_encodings.py (x*cos + rotate_pairs(x)*sin), x [rows,D] with a
 *   row stride (a column slice of a wider projection is fine), cos/sin/out [rows,D] contiguous -- same float operations.
 * mmf_adaln_modulate: x*(1+scale)+shift of the AdaLN blocks (diffuser_actor/layers.py), x/out [B,L,D], scale_shift [B,2D].
 * mmf_attention_small: softmax(q k^T/sqrt(d) + key padding) v per head for small problems (head_dim in {8,15,16,20,24,32}), fp32;
 *   q/out [B,Lq,heads*d], k/v [B,Lk,heads*d] with row strides, key_padding [B,Lk] bytes (1 = ignore) or NULL.  Agrees with
 *   torch's SDPA math path to float rounding (not bit for bit: different summation order).
 * mmf_ddpm_step: DDPMScheduler.step of the position (channels [0,split)) and rotation ([split,C)) schedulers in one launch:
 *   x0 = (x - s1*eps)*inv_s2 [clamp +-clip if clip > 0]; prev = c0*x0 + c1*x [+ sigma*noise if sigma > 0]; x/noise/out [rows,C],
 *   eps [rows, >=C] with a row stride; coef = {s1, inv_s2, c0, c1, sigma, clip} per scheduler (host). */
/* Whole-block kernels of the head's AttentionBlock / FeedForwardBlock at inference (embedding dim D = 120), weights as
 * the TRANSPOSE of what torch.nn.Linear stores, i.e. [in,out] row-major (coalesced weight fetch), biases [out],
 * scale_shift [B,2D] or NULL, cos/sin [B*L,D] or both NULL:
 *   mmf_ffn_block      h = x*(1+scale)+shift; out = LayerNorm(h + fc2(relu(fc1(h))))
 *   mmf_q_block        out = rotary(q_proj(x*(1+scale)+shift))
 *   mmf_kv_block       k_out = rotary(kv_proj(memory)[:, :D]), v_out = kv_proj(memory)[:, D:]
 *   mmf_attn_out_block out = LayerNorm(residual + out_proj(att))
 * One launch each instead of 3-7; results agree with the composite torch ops to float rounding.
 *   mmf_qkv_block      self-attention: mmf_q_block and mmf_kv_block of the same tokens in one launch
 *   mmf_out_ffn_block  mmf_attn_out_block followed by mmf_ffn_block (scale_shift is the FFN's) in one launch */
int mmf_qkv_block(const float* x_dev, const float* scale_shift_dev, const float* Wq_dev, const float* bq_dev, const float* Wkv_dev,
                  const float* bkv_dev, const float* cos_dev, const float* sin_dev, float* q_out_dev, float* k_out_dev, float* v_out_dev, int B,
                  int L, int D, void* stream);
int mmf_out_ffn_block(const float* att_dev, const float* residual_dev, const float* Wo_dev, const float* bo_dev, const float* ln1_weight_dev,
                      const float* ln1_bias_dev, float ln1_eps, const float* scale_shift_dev, const float* W1_dev, const float* b1_dev,
                      const float* W2_dev, const float* b2_dev, const float* ln2_weight_dev, const float* ln2_bias_dev, float ln2_eps,
                      float* out_dev, int B, int L, int D, void* stream);
/* Matrix-core forms of the per-layer kernels, built for D = 120, H = 8 (head_dim 15).  The attention kernel multiplies in
 * exact f32 (v_mfma_f32_16x16x4_f32); the projections and the out_proj + LayerNorm + feed-forward block compute every f32 GEMM
 * as three fp16 matrix-core products of operands split x = hi + lo / 2048 (22-bit mantissas, f32 accumulation: deviation from
 * the reference's outputs as small as the f32 form's, tests/test_gpu_policy_golden.py).  UNLIKE the block kernels above they
 * take every weight matrix (Wq, Wo, W1, W2 [D, D]; Wkv [2 D, D] as torch.nn.Linear stores them, [out, in]) PRE-SPLIT by
 *   mmf_split_linear_weight  weight [out_features = 120 n, in_features = 120] f32 -> split: one block of 65 536 bytes per 120
 *                        output rows (n = 2 for the stacked key | value projection), in the order in which the kernels' waves
 *                        load their operands; call once per weight
 * and passed through the `const float*` weight parameters below as opaque device pointers:
 *   mmf_qkv_heads        the projections of mmf_qkv_block written head-major and padded to 16 channels:
 *                        q_heads, k_heads [B, H, L16, 16], v_heads_t [B, H, 16, L16] (L16 = L rounded up to 16; padding = 0).
 *                        roles: 7 = q | k | v, 1 = q alone (Wkv / k / v may be null), 6 = k | v alone (Wq / q may be null)
 *   mmf_attention_heads  softmax(q k^T / sqrt(head_dim) + key padding) v over those layouts -> out [B, Lq, D]; key_padding: [B, Lk16]
 *                        bytes, 1 = ignore, the keys beyond Lk marked too (or null)
 *   mmf_out_ffn_mfma     same contract as mmf_out_ffn_block
 *   mmf_attention_heads_split + mmf_out_ffn_mfma_partials   Lq <= 16 query rows over a long key axis (the trajectory tokens over
 *                        the whole context): the keys are divided among n_split workgroups per (batch element, head), each leaves
 *                        an un-normalised partial {O [16 channels][16 rows], maxima [16], sums [16]} in partials
 *                        [B, H, n_split, 18, 16], and the following out-projection kernel merges them while it loads its input
 *                        (no cross-workgroup merge, i.e. no device-scope fence, inside the attention kernel).  partials_dev null:
 *                        only *n_split_out is set (size query)
 *   mmf_qkv_heads2, mmf_out_ffn_mfma2   the same kernels for TWO independent stacks of identical shape in one launch (the
 *                        rotation and the position stack of the diffusion head).  next14 / layer26: the next7 / layer13 arrays of
 *                        mmf_out_ffn_qkv for stack 0 then stack 1; eps4 = {ln1, ln2} of stack 0 then stack 1 (HOST array);
 *                        activations and outputs are stack-major ([2, B, ...]: mmf_attention_heads runs the pair as batch 2 B)
 *   mmf_out_ffn_qkv2     mmf_out_ffn_qkv (roles 7, no partials) for the two stacks in one launch: layer26 / eps4 / out as
 *                        mmf_out_ffn_mfma2, next14 / q / k / v as mmf_qkv_heads2
 *   mmf_out_ffn_qkv      mmf_out_ffn_mfma of layer i followed, in the same launch, by mmf_qkv_heads (roles 7) of layer i + 1 on
 *                        its output.  layer13 (HOST array of device pointers): att, residual, Wo, bo, ln1_weight, ln1_bias,
 *                        scale_shift (or null), W1, b1, W2, b2, ln2_weight, ln2_bias; next7: scale_shift of the next layer's
 *                        query input (or null), Wq, bq, Wkv, bkv, cos, sin (both null: no rotary).  roles: 7 = q | k | v,
 *                        1 = q alone (the next layer attends to a cached memory; Wkv / bkv / k / v may be null).
 *                        att_partials_dev non-null: layer13[0] is ignored and the attention output is merged from the key-split
 *                        partials of mmf_attention_heads_split (L <= 16) */
int mmf_qkv_heads(const float* x_dev, const float* scale_shift_dev, const float* Wq_dev, const float* bq_dev, const float* Wkv_dev,
                  const float* bkv_dev, const float* cos_dev, const float* sin_dev, float* q_heads_dev, float* k_heads_dev,
                  float* v_heads_t_dev, int B, int L, int D, int H, int roles, void* stream);
int mmf_attention_heads(const float* q_heads_dev, const float* k_heads_dev, const float* v_heads_t_dev, const uint8_t* key_padding_dev,
                        float* out_dev, int B, int Lq, int Lk, int H, int head_dim, void* stream);
int mmf_out_ffn_mfma(const float* att_dev, const float* residual_dev, const float* Wo_dev, const float* bo_dev, const float* ln1_weight_dev,
                     const float* ln1_bias_dev, float ln1_eps, const float* scale_shift_dev, const float* W1_dev, const float* b1_dev,
                     const float* W2_dev, const float* b2_dev, const float* ln2_weight_dev, const float* ln2_bias_dev, float ln2_eps,
                     float* out_dev, int B, int L, int D, void* stream);
/* Head and tail of a denoising step (D = 120), one launch each instead of ~15:
 *   mmf_step_prologue  tokens_out [B, num_tokens, D] = trajectory [B, num_tokens, 9] x traj_encoder + position_table [num_tokens, D];
 *                      adaln_out [B, adaln_width] = silu(time_embedding [D] + history [B, D]) x adaln_wt [D, adaln_width] + adaln_bias
 *                      (the stacked scale/shift projections of every AdaLN block); cos_out / sin_out: 3-D rotary tables of the
 *                      trajectory positions (channel thirds x, y, z; pair k of a third uses rotary_freq[k]), row i of batch b at
 *                      b * rotary_batch_stride + i * D (so they can be the first rows of sequence-wide tables)
 *   mmf_head_outputs   pred_out [B, L, G, 10] = (position 3 | rotation 6 | openness 1) and head_yaw_out [B, L] from rows
 *                      l * G + g of the two output stacks (batch stride seq_batch_stride floats).  weights20 (HOST array of 20
 *                      device pointers, transposed [in, out] weights then bias): rotation_proj, position_proj, rotation_out.0,
 *                      rotation_out.2, position_out.0, position_out.2, openness_out.0, openness_out.2, head_yaw_out.0,
 *                      head_yaw_out.2 (the last four entries null: no head yaw)
 *   mmf_step_tail      mmf_head_outputs, then the reverse-diffusion update of the trajectory (the arithmetic of mmf_ddpm_step on
 *                      channels [0,3) with coef_pos6 and [3,9) with coef_rot6 = {s1, inv_s2, c0, c1, sigma, clip}, HOST arrays),
 *                      then -- unless tokens_out is null -- the token / rotary part of mmf_step_prologue for the NEW trajectory:
 *                      a denoising step ends, and the next one begins, in this one launch.
 *                      mmf_step_prologue with adaln_width = 0 computes tokens and rotary codes only. */
int mmf_step_prologue(const float* trajectory_dev, int B, int num_tokens, const float* traj_encoder_wt_dev, const float* traj_encoder_bias_dev,
                      const float* position_table_dev, const float* time_embedding_dev, const float* history_dev,
                      const float* rotary_freq_dev, const float* adaln_wt_dev, const float* adaln_bias_dev, int adaln_width,
                      float* tokens_out_dev, float* adaln_out_dev, float* cos_out_dev, float* sin_out_dev, long long rotary_batch_stride,
                      int D, void* stream);
int mmf_head_outputs(const float* rotation_seq_dev, const float* position_seq_dev, long long seq_batch_stride, int B, int L, int G,
                     const float* const* weights20, float* pred_out_dev, float* head_yaw_out_dev, int D, void* stream);
int mmf_step_tail(const float* rotation_seq_dev, const float* position_seq_dev, long long seq_batch_stride, int B, int L, int G,
                  const float* const* weights20, float* pred_out_dev, float* head_yaw_out_dev, const float* trajectory_dev,
                  const float* noise_dev, const float* coef_pos6, const float* coef_rot6, float* trajectory_out_dev,
                  const float* traj_encoder_wt_dev, const float* traj_encoder_bias_dev, const float* position_table_dev,
                  const float* rotary_freq_dev, float* tokens_out_dev, float* cos_out_dev, float* sin_out_dev, long long rotary_batch_stride,
                  int D, void* stream);
int mmf_out_ffn_qkv(const float* const* layer13, float ln1_eps, float ln2_eps, float* out_dev, const float* const* next7,
                    float* q_heads_dev, float* k_heads_dev, float* v_heads_t_dev, int B, int L, int D, int H, int roles,
                    const float* att_partials_dev, int n_split, void* stream);
int mmf_qkv_heads2(const float* x0_dev, const float* x1_dev, const float* const* next14, float* q_heads_dev, float* k_heads_dev,
                   float* v_heads_t_dev, int B, int L, int D, int H, void* stream);
int mmf_out_ffn_mfma2(const float* const* layer26, const float* eps4, float* out_dev, int B, int L, int D, void* stream);
int mmf_out_ffn_qkv2(const float* const* layer26, const float* eps4, float* out_dev, const float* const* next14, float* q_heads_dev,
                     float* k_heads_dev, float* v_heads_t_dev, int B, int L, int D, int H, void* stream);
int mmf_split_linear_weight(const float* weight_dev, int out_features, int in_features, void* split_dev, void* stream);
/* The same split for LARGE GEMMs on a library kernel (the frozen image backbone; mindmap/image_processing/feature_extraction.py:322
 * runs its matmuls under TF32, this keeps 22 bits): x [rows, K] f32 -> out [rows, 3 K + 64] fp16 = [hi | hi / 2048 | lo | 1, 1 / 2048,
 * 0 x 62].  With the weight stored as [hi | lo * 2048 | hi | bias hi, bias lo * 2048, 0 x 62] along the reduction axis
 * (diffuser_actor/split_linear.py), ONE plain fp16 GEMM with f32 accumulation returns x w^T + bias at f32 accuracy.  K >= 64, a multiple
 * of 8; |x| < 65 504. */
int mmf_split_activations3(const float* x_dev, int64_t rows, int K, void* out_dev, void* stream);
/* The element-wise passes between the frozen backbone's GEMMs, fused with that split (one HBM round trip each instead of two or three):
 *   mmf_gelu_split_activations3:      out = split3(gelu(x)), exact (erf) GELU;
 *   mmf_split_attention_heads3:       out = split3 of the attention output att [B, heads, L, head_dim] read as rows (b, l) of
 *                                     heads * head_dim channels (the layout the next Linear wants, without the transpose copy);
 *   mmf_layernorm_split_activations3: s = x (+ residual, then sum_out <- s);  out = split3(LayerNorm(s; gamma, beta, eps)), two-pass
 *                                     float32 statistics; K in {256, 512, 768, 1024}. */
/* Self-attention of the frozen backbone at float32 accuracy on the fp16 matrix cores (csrc/mmf_kernels_backbone.hip): softmax(q k^T
 * scale) v per (batch, head), every product as hi hi + (hi lo + lo hi) / 2048 of operands split into two fp16 values (22-bit
 * mantissas, f32 accumulation), statistics and exponentials in f32.  q / k / v: rows of `head_dim` floats of head h at
 * base + b batch_stride + l row_stride + h head_dim (e.g. the three slices of a [B, L, 3, H, 64] projection); out [B, L, H head_dim].
 * head_dim = 64, L a multiple of 128.  split_out != 0: out is written as mmf_split_activations3 would split that result
 * ([B L, 3 H head_dim + 64] fp16: the operand of the next Linear's GEMM).  Replaces F.scaled_dot_product_attention inside the
 * backbone the reference runs under TF32 (image_processing/feature_extraction.py:318-323). */
int mmf_attention_split(const float* q_dev, const float* k_dev, const float* v_dev, int64_t row_stride, int64_t batch_stride, int B, int H,
                        int L, int head_dim, float scale, void* out_dev, int split_out, void* stream);
/* The gradient of mmf_adaln_modulate (y = x (1 + scale_b) + shift_b, (scale | shift) = scale_shift [B, 2 D]) for the training step:
 * grad_x [B, L, D] = grad_out (1 + scale), grad_scale_shift [B, 2 D] = (sum_L grad_out x | sum_L grad_out); D a multiple of 4 up to 128;
 * scratch: mmf_adaln_modulate_grad_scratch_bytes(B) bytes (column partials, added in a fixed order). */
int64_t mmf_adaln_modulate_grad_scratch_bytes(int B);
int mmf_adaln_modulate_grad(const float* grad_out_dev, const float* x_dev, const float* scale_shift_dev, int B, int L, int D, float* grad_x_dev,
                            float* grad_scale_shift_dev, float* scratch_dev, void* stream);
/* The parameter gradients of a Linear layer y = x W^T + b over many rows (the trainable stacks' 120 -> 120 / 240 projections over
 * 19 712 or 98 304 token rows): grad_weight [out, in] = grad_out^T x, grad_bias [out] = column sums of grad_out (null: not wanted);
 * grad_out [rows, out], x [rows, in] contiguous; out_features <= 256, in_features <= 128.  Rows split over the chip, f32 matrix
 * cores, splits added in a fixed order (deterministic).  scratch: mmf_linear_weight_grad_scratch_bytes(rows, out, in) bytes. */
int64_t mmf_linear_weight_grad_scratch_bytes(int64_t rows, int out_features, int in_features);
int mmf_linear_weight_grad(const float* grad_out_dev, const float* x_dev, int64_t rows, int out_features, int in_features, float* grad_weight_dev,
                           float* grad_bias_dev, float* scratch_dev, void* stream);
/* LayerNorm(a + b) of the trainable post-norm blocks (mindmap/diffuser_actor/layers.py: D = 120), forward and backward, rows of
 * D <= 128 channels (D a multiple of 4), float32.  forward: b may be null (plain LayerNorm(a)); with b, sum_out receives a + b (the
 * backward pass wants the normalised input); y, mean [rows], rstd [rows].  backward: grad_x (= the gradient of a AND of b), grad_gamma,
 * grad_beta [D] from grad_y, x (= a, or sum_out), gamma, mean, rstd; scratch: mmf_layernorm_train_scratch_bytes() bytes (per-workgroup
 * column partials, added in a fixed order: deterministic). */
int64_t mmf_layernorm_train_scratch_bytes(void);
int mmf_layernorm_train_forward(const float* a_dev, const float* b_dev, const float* gamma_dev, const float* beta_dev, float eps, int64_t rows, int D,
                                float* sum_out_dev, float* y_dev, float* mean_dev, float* rstd_dev, void* stream);
int mmf_layernorm_train_backward(const float* grad_y_dev, const float* x_dev, const float* gamma_dev, const float* mean_dev, const float* rstd_dev,
                                 int64_t rows, int D, float* grad_x_dev, float* grad_gamma_dev, float* grad_beta_dev, float* scratch_dev, void* stream);
/* Attention of the TRAINABLE transformer stacks, forward and backward, float32 on the f32 matrix cores: heads of up to 16 channels
 * (the policy's: 8 heads x 15), any Lq / Lk, optional key-padding mask ([B, Lk] bytes, != 0: ignore the key).  q / k / v: the rows of
 * head h at base + b * batch_stride + l * row_stride + h * head_dim (element strides in strides6 = {q_row, q_batch, k_row, k_batch,
 * v_row, v_batch}: the projections' own [B, L, heads * head_dim] layout or a chunk view of a wider projection -- no head transpose,
 * no channel padding).  forward: out [B, Lq, H * head_dim], lse [B, H, Lq] (base-2 log-sum-exp of the scaled logits, kept for the
 * backward).  backward: dq [B, Lq, H hd], dk / dv [B, Lk, H hd] from out, dout (both contiguous) and lse; P is recomputed.
 * Replaces F.scaled_dot_product_attention + its autograd in the reference's attention layers when they train
 * (mindmap/diffuser_actor/layers.py / multihead_custom_attention.py; torch has no kernel shaped for a 15-channel head). */
int mmf_train_attention_forward(const float* q_dev, const float* k_dev, const float* v_dev, const int64_t* strides6, const uint8_t* key_padding_dev,
                                int B, int H, int Lq, int Lk, int head_dim, float scale, float* out_dev, float* lse_dev, void* stream);
int mmf_train_attention_backward(const float* q_dev, const float* k_dev, const float* v_dev, const int64_t* strides6, const uint8_t* key_padding_dev,
                                 int B, int H, int Lq, int Lk, int head_dim, float scale, const float* out_dev, const float* dout_dev,
                                 const float* lse_dev, float* dsum_scratch_dev, float* dq_dev, float* dk_dev, float* dv_dev, void* stream);
int mmf_gelu_split_activations3(const float* x_dev, int64_t rows, int K, void* out_dev, void* stream);
int mmf_split_attention_heads3(const float* att_dev, int64_t B, int heads, int L, int head_dim, void* out_dev, void* stream);
int mmf_layernorm_split_activations3(const float* x_dev, const float* residual_dev, const float* gamma_dev, const float* beta_dev, float eps,
                                     int64_t rows, int K, float* sum_out_dev, void* out_dev, void* stream);
/* mmf_cross_layer: mmf_attention_heads_split and the block kernel that consumes its partials (mmf_out_ffn_mfma_partials, or with
 * next7 != NULL mmf_out_ffn_qkv with roles 1) in ONE launch: the attention workgroups lead the grid and hand their partials to
 * the block workgroup of their batch element as self-validating 64-bit words, which that workgroup polls after it has requested
 * its weights (its start-up runs beside the attention).  layer13 / next7: as mmf_out_ffn_qkv (layer13[0] unused); qkv3 (HOST
 * array): q_heads [B, H, 16, 16] of this layer, the context's k_heads [B, H, Lk16, 16] and v_heads_t [B, H, 16, Lk16];
 * handover: B * H * 4 * 18 * 16 + 1 uint64 words on the device, zeroed by the caller once -- the last word becomes non-zero if
 * a wait expired (a peer workgroup was not running: results undefined); tag: non-zero and different from every tag used on this
 * buffer since it was zeroed (count the launches).  Lq <= 16. */
/* mmf_self_layer: a SELF-attention layer in one launch, the same way -- mmf_attention_heads of this layer's q / k / v (qkv3), its
 * output rows handed to the block workgroups (mmf_out_ffn_mfma, or with next7 != NULL mmf_out_ffn_qkv with roles 7 producing the
 * NEXT layer's q / k / v) as self-validating words.  handover: B * L * D + 1 uint64 words, zeroed once; tag as below. */
int mmf_self_layer(const float* const* layer13, float ln1_eps, float ln2_eps, float* out_dev, const float* const* next7, float* q_heads_next_dev,
                   float* k_heads_next_dev, float* v_heads_t_next_dev, const float* const* qkv3, const uint8_t* key_padding16_dev,
                   uint64_t* handover_dev, uint32_t tag, int B, int L, int D, int H, void* stream);
int mmf_cross_layer(const float* const* layer13, float ln1_eps, float ln2_eps, float* out_dev, const float* const* next7,
                    float* q_heads_next_dev, const float* const* qkv3, const uint8_t* key_padding16_dev, uint64_t* handover_dev, uint32_t tag,
                    int B, int Lq, int Lk, int D, int H, void* stream);
int mmf_attention_heads_split(const float* q_heads_dev, const float* k_heads_dev, const float* v_heads_t_dev, const uint8_t* key_padding_dev,
                              float* partials_dev, int B, int Lq, int Lk, int H, int head_dim, int* n_split_out, void* stream);
int mmf_out_ffn_mfma_partials(const float* partials_dev, int n_split, const float* residual_dev, const float* Wo_dev, const float* bo_dev,
                              const float* ln1_weight_dev, const float* ln1_bias_dev, float ln1_eps, const float* scale_shift_dev,
                              const float* W1_dev, const float* b1_dev, const float* W2_dev, const float* b2_dev, const float* ln2_weight_dev,
                              const float* ln2_bias_dev, float ln2_eps, float* out_dev, int B, int L, int D, void* stream);
int mmf_ffn_block(const float* x_dev, const float* scale_shift_dev, const float* W1_dev, const float* b1_dev, const float* W2_dev,
                  const float* b2_dev, const float* ln_weight_dev, const float* ln_bias_dev, float ln_eps, float* out_dev, int B, int L,
                  int D, void* stream);
int mmf_q_block(const float* x_dev, const float* scale_shift_dev, const float* Wq_dev, const float* bq_dev, const float* cos_dev,
                const float* sin_dev, float* out_dev, int B, int L, int D, void* stream);
int mmf_kv_block(const float* memory_dev, const float* Wkv_dev, const float* bkv_dev, const float* cos_dev, const float* sin_dev,
                 float* k_out_dev, float* v_out_dev, long long tokens, int D, void* stream);
int mmf_attn_out_block(const float* att_dev, const float* residual_dev, const float* Wo_dev, const float* bo_dev,
                       const float* ln_weight_dev, const float* ln_bias_dev, float ln_eps, float* out_dev, long long tokens, int D,
                       void* stream);
int mmf_ddpm_step(const float* x_dev, const float* eps_dev, long long eps_row_stride, const float* noise_dev, float* out_dev,
                  long long rows, int C, int split, const float* coef_a_host6, const float* coef_b_host6, void* stream);
int mmf_rotary_apply(const float* x_dev, long long x_row_stride, const float* cos_dev, const float* sin_dev, float* out_dev,
                     long long rows, int D, void* stream);
/* The gradient of mmf_rotary_apply with respect to x (the training step): grad_out, cos, sin, grad_x all [rows, D] contiguous; the
 * same floats autograd computes for x * cos + rotate_pairs(x) * sin. */
int mmf_rotary_apply_grad(const float* grad_out_dev, const float* cos_dev, const float* sin_dev, float* grad_x_dev, long long rows, int D,
                          void* stream);
int mmf_adaln_modulate(const float* x_dev, const float* scale_shift_dev, float* out_dev, int B, int L, int D, void* stream);
int mmf_attention_small(const float* q_dev, const float* k_dev, long long k_row_stride, const float* v_dev, long long v_row_stride,
                        const uint8_t* key_padding_dev, float* out_dev, int B, int Lq, int Lk, int heads, int head_dim, void* stream);

/* ---- diagnostics / measurement --------------------------------------------------------------- */
/* Last sphere-traced synthetic depth image of the mapper: dims, then copy to out [Hs,Ws] f32. */
int mmf_get_synthetic_depth_dims(mmf_handle h, int mapper_id, int* Hs, int* Ws);
int mmf_get_synthetic_depth(mmf_handle h, int mapper_id, float* out_dev, void* stream);
/* Render only (no integration). */
int mmf_render_synthetic_depth(mmf_handle h, int mapper_id, int H, int W, const float* T_W_C_host, const float* K_host,
                               void* stream);
/* Blocks in view of the last add_depth_frame, sorted (x,y,z): count (synchronises), then indices [n,3]. */
int mmf_last_view_block_count(mmf_handle h, int mapper_id, void* stream, int* out);
int mmf_get_last_view_blocks(mmf_handle h, int mapper_id, int32_t* indices_dev, int n, void* stream);

/* Cumulative counters since creation / last reset (synchronises `stream`):
 *  [0] depth frames  [1] TSDF blocks updated  [2] TSDF blocks allocated
 *  [3] colour frames [4] colour blocks updated [5] feature frames [6] feature blocks updated
 *  [7] feature blocks allocated  [8] feature voxels updated (voxels that passed the occlusion / mask gate) */
#define MMF_NUM_STATS 9
int mmf_get_stats(mmf_handle h, int mapper_id, void* stream, int64_t* out /* [MMF_NUM_STATS] */);
int mmf_reset_stats(mmf_handle h, int mapper_id, void* stream);
/* Diagnostics: 100 MHz device timestamps taken by the TSDF allocation workgroup of the last fused frame (start, after the
 * decay compaction, table loads consumed, scan done, inserts done, counters published).  enable != 0 switches the
 * recording on for later frames (off otherwise); out10 receives the last recording (zeros if none): the six stamps, then
 * over the mask column workgroups that share the launch: latest end [6], earliest start [7], longest duration [8],
 * latest start [9].  Synchronises. */
int mmf_get_alloc_timeline(mmf_handle h, int mapper_id, int enable, int64_t* out10);
/* Diagnostics: number of new TSDF blocks that were integrated by the sweeper of k_alloc_tsdf since the mapper was created,
 * i.e. blocks whose waiter workgroup abandoned its wait for the allocation workgroups of the same launch (0 in normal
 * operation; the environment variable MMF_DEBUG_FORCE_ALLOC_TIMEOUT=1 at mapper creation makes half of the waiters abandon
 * at once, =2 makes the sweeper fail too: the next call on the mapper then returns MMF_ERR_BAD_STATE once).  Synchronises. */
int mmf_debug_alloc_recoveries(mmf_handle h, int mapper_id, void* stream, int64_t* out);
/* Diagnostics of a layer's block index: out8 = {hash table entries (0: bounded workspace, the dense block table is the index and
 * no hash is kept), tombstones in the table, table rebuilds since the layer was created / cleared, live blocks, and the last depth
 * frame's view grid nx, ny, nz (cells = blocks), 0}.  Synchronises. */
int mmf_debug_hash_state(mmf_handle h, int mapper_id, int layer, void* stream, int64_t* out8);
/* Diagnostics: the number of tombstone entries that ARE in a layer's hash table (a scan of the table), to hold against the
 * counter mmf_debug_hash_state reports (out8[1]) -- the two are equal: insertion takes a reused tombstone off the count.  0 for a
 * layer indexed by its dense table.  Synchronises. */
int mmf_debug_count_tombstones(mmf_handle h, int mapper_id, int layer, void* stream, int64_t* out);
/* Diagnostics: per-workgroup timeline of the fused frame kernels.  buffer_dev: uint64 [3 * capacity_records] on the device
 * (capacity_records >= 6 * 8192; the caller zeroes it), records {role id, start, end} in 100 MHz ticks at slot
 * (role id / 10 - 1) * 8192 + workgroup index; null = off (the default).
 * Role ids: 10 raycast, 11 mask rows, 12 decay (k_front); 20 allocation, 21 mask columns (k_alloc_jobs); 30 k_tsdf_pass;
 * 40 allocation, 41 sphere trace (k_sphere_alloc); 50 k_app_frame; 60 k_feature_flat.  tools/wg_trace.py prints it. */
int mmf_debug_wg_trace(uint64_t* buffer_dev, int capacity_records);

/* Kernel timing with HIP events on the launch stream.  kernel ids: */
#define MMF_K_RAYCAST 0
#define MMF_K_ALLOC 1
#define MMF_K_TSDF 2
#define MMF_K_CANDIDATES 3
#define MMF_K_SPHERE 4
#define MMF_K_COLOR 5
#define MMF_K_FEATURE 6
#define MMF_K_DECAY 7
#define MMF_K_MESH 8
#define MMF_K_FEATURE_FLAT 9 /* balanced phase 2 of the feature update (k_feature_flat) */
#define MMF_NUM_KERNEL_IDS 10
/* kernel_mask: bit k set = time kernel class k (0 = off, (1<<MMF_NUM_KERNEL_IDS)-1 = all). */
int mmf_profile_enable(mmf_handle h, int kernel_mask);
/* Time only every `stride`-th launch of each enabled kernel class (default 1 = every launch): keeps the event overhead
 * out of a throughput measurement while the timing is still taken live inside it. */
int mmf_profile_set_stride(mmf_handle h, int stride);
/* Sum of elapsed ms and number of timed launches of kernel class `kernel_id` (synchronises). */
int mmf_profile_get(mmf_handle h, int kernel_id, double* total_ms, int64_t* launches);
int mmf_profile_reset(mmf_handle h);
const char* mmf_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* MMFUSION_H_ */
