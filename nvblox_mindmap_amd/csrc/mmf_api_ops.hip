// mmf_api_ops.hip -- the stateless entry points of the C ABI (include/mmfusion.h): depth back-projection, mask algebra, feature
// resize and the policy-side ops (farthest-point sampling, the fused inference kernels).
#include <chrono>
#include <mutex>

#include "mmf_api_internal.h"

using namespace mmf;
using namespace mmf_host;

extern "C" {

int mmf_backproject_depth(const float* depth, const float* K, const float* T, int B, int H, int W, float* out, void* stream) {
  if (!depth || !K || !T || !out || B < 0 || H <= 0 || W <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_backproject_depth");
  launch_backproject(depth, K, T, B, H, W, out, (hipStream_t)stream);
  return check_launch();
}

int mmf_sample_inputs_scratch_floats(void) { return sample_inputs_scratch_floats(); }

int mmf_sample_frame_inputs(const float* rgb_chw, int H, int W, const float* pose7, const float* K9, uint8_t* rgb_hwc_out, float* small_out,
                            float* scratch, void* stream) {
  if (!rgb_chw || !pose7 || !K9 || !rgb_hwc_out || !small_out || !scratch || H <= 0 || W <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_sample_frame_inputs");
  launch_sample_inputs(rgb_chw, H, W, pose7, K9, rgb_hwc_out, small_out, scratch, (hipStream_t)stream);
  return check_launch();
}

namespace {
// The host-visible record of mmf_sample_frame_inputs_host: 20 floats + the sequence number of the call that wrote them, in coherent
// (fine-grained) pinned host memory the kernel stores to directly; one per device, calls serialised by the mutex.
struct HostRecord {
  float v[20];
  unsigned seq;
};
struct HostRecordSlot {
  HostRecord* host = nullptr;
  HostRecord* dev = nullptr;
  unsigned next = 1;
};
std::mutex g_rec_mutex;
HostRecordSlot g_rec[64];
}  // namespace

int mmf_sample_frame_inputs_host(const float* rgb_chw, int H, int W, const float* pose7, const float* K9, uint8_t* rgb_hwc_out, float* scratch,
                                 float* host_out20, void* stream) {
  if (!rgb_chw || !pose7 || !K9 || !rgb_hwc_out || !scratch || !host_out20 || H <= 0 || W <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_sample_frame_inputs_host");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(MMF_ERR_HIP, "mmf_sample_frame_inputs_host: no current device");
  std::lock_guard<std::mutex> lock(g_rec_mutex);
  HostRecordSlot& r = g_rec[dev];
  if (!r.host) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, sizeof(HostRecord), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      return fail(MMF_ERR_HIP, "mmf_sample_frame_inputs_host: cannot allocate the host-visible record");
    }
    std::memset(h, 0, sizeof(HostRecord));
    r.host = (HostRecord*)h;
    r.dev = (HostRecord*)d;
  }
  const unsigned seq = r.next++;
  if (r.next == 0) r.next = 1;
  launch_sample_inputs(rgb_chw, H, W, pose7, K9, rgb_hwc_out, r.dev->v, scratch, (hipStream_t)stream, &r.dev->seq, seq);
  const int rc = check_launch();
  if (rc != MMF_OK) return rc;
  // Poll the record: the kernels are microseconds of work behind whatever the stream still holds.  A short spin covers the common case
  // (an idle stream: ~10 us); past it the thread waits on the stream like any synchronising read would.
  volatile unsigned* flag = &r.host->seq;
  const auto t0 = std::chrono::steady_clock::now();
  bool seen = false;
  for (long it = 0;; ++it) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) {
      seen = true;
      break;
    }
    if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
  }
  if (!seen) {
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
      (void)hipGetLastError();
      return fail(MMF_ERR_HIP, "mmf_sample_frame_inputs_host: stream synchronisation failed");
    }
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return fail(MMF_ERR_HIP, "mmf_sample_frame_inputs_host: the record was not written");
  }
  std::memcpy(host_out20, (const void*)r.host->v, 20 * sizeof(float));
  return MMF_OK;
}

int mmf_erode_mask(const uint8_t* mask, uint8_t* out, uint8_t* tmp, int H, int W, int iterations, void* stream) {
  if (!mask || !out || !tmp || H <= 0 || W <= 0 || iterations < 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_erode_mask");
  launch_erode(mask, out, tmp, H, W, iterations, (hipStream_t)stream);
  return check_launch();
}

int mmf_feature_mask(const uint8_t* input_mask, const float* depth, int H, int W, float min_depth_m, int k_in, int k_depth,
                     int border_percent, int Hf, int Wf, uint8_t* out, uint8_t* tmp, void* stream) {
  if (!input_mask || !depth || !out || !tmp || H <= 0 || W <= 0 || Hf <= 0 || Wf <= 0 || k_in < 0 || k_depth < 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_feature_mask");
  launch_feature_mask(input_mask, depth, H, W, min_depth_m, k_in, k_depth, border_percent, Hf, Wf, out, tmp, (hipStream_t)stream);
  return check_launch();
}

int mmf_frame_masks(const uint8_t* input_mask, const float* depth, int H, int W, float min_depth_m, int k_in, int k_depth,
                    int border_percent, int Hf, int Wf, uint8_t* depth_mask_out, uint8_t* feature_mask_out, uint8_t* tmp,
                    void* stream) {
  if (!input_mask || !depth || !feature_mask_out || !tmp || H <= 0 || W <= 0 || Hf <= 0 || Wf <= 0 || k_in < 0 || k_depth < 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_frame_masks");
  launch_frame_masks(input_mask, depth, H, W, min_depth_m, k_in, k_depth, border_percent, Hf, Wf, depth_mask_out, feature_mask_out,
                     tmp, (hipStream_t)stream);
  return check_launch();
}

int mmf_depth_mask(const uint8_t* input_mask, const float* depth, int H, int W, float min_depth_m, uint8_t* out, void* stream) {
  if (!depth || !out || H <= 0 || W <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_depth_mask");
  launch_depth_mask(input_mask, depth, H, W, min_depth_m, out, (hipStream_t)stream);
  return check_launch();
}

int mmf_upsample_features_spec(const float* lowres, int hh, int ww, int Cin, void* out, int Hf, int Wf, int Cpad, int fma_contraction,
                               void* stream) {
  if (!lowres || !out || hh <= 0 || ww <= 0 || Cin <= 0 || Hf <= 0 || Wf <= 0 || Cpad < Cin || Cpad % 8 != 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_upsample_features (Cpad must be a multiple of 8 and >= Cin)");
  launch_upsample_features(lowres, hh, ww, Cin, (__half*)out, Hf, Wf, Cpad, (hipStream_t)stream, fma_contraction != 0);
  return check_launch();
}

int mmf_upsample_features(const float* lowres, int hh, int ww, int Cin, void* out, int Hf, int Wf, int Cpad, void* stream) {
  return mmf_upsample_features_spec(lowres, hh, ww, Cin, out, Hf, Wf, Cpad, 0, stream);
}

static int fps_entry(const float* x, int B, int N, int C, int npoints, int start_idx, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (!x || !out_idx || B <= 0 || N <= 0 || C <= 0 || npoints <= 0 || npoints > N || start_idx < 0 || start_idx >= N)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_farthest_point_sampling");
  const int rc = launch_fps(x, B, N, C, npoints, start_idx, reinterpret_cast<long long*>(out_idx), (hipStream_t)stream, workspace, workspace_bytes);
  if (rc == 1) return fail(MMF_ERR_INVALID_ARG, "mmf_farthest_point_sampling supports N <= 8192 and C <= 1024");
  if (rc != 0) return fail(MMF_ERR_HIP, "mmf_farthest_point_sampling: HIP runtime call failed (or the workspace is too small)");
  return check_launch();
}

int mmf_farthest_point_sampling(const float* x, int B, int N, int C, int npoints, int start_idx, int64_t* out_idx, void* stream) {
  return fps_entry(x, B, N, C, npoints, start_idx, out_idx, nullptr, 0, stream);
}

int64_t mmf_fps_workspace_bytes(int B, int N, int C) { return (B <= 0 || N <= 0 || C <= 0) ? 0 : (int64_t)fps_workspace_bytes(B, N, C); }

int mmf_farthest_point_sampling_ws(const float* x, int B, int N, int C, int npoints, int start_idx, int64_t* out_idx, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < mmf_fps_workspace_bytes(B, N, C))
    return fail(MMF_ERR_INVALID_ARG, "mmf_farthest_point_sampling_ws: workspace smaller than mmf_fps_workspace_bytes(B, N, C)");
  return fps_entry(x, B, N, C, npoints, start_idx, out_idx, workspace, (size_t)workspace_bytes, stream);
}

// ---- inference-side fused ops of the diffusion head ------------------------------------------------------------------
int mmf_rotary_apply(const float* x, long long x_row_stride, const float* cos_, const float* sin_, float* out, long long rows, int D,
                     void* stream) {
  if (!x || !cos_ || !sin_ || !out || rows < 0 || D <= 0 || (D & 1) || x_row_stride < D)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_rotary_apply");
  launch_rotary_apply(x, x_row_stride, cos_, sin_, out, rows, D, (hipStream_t)stream);
  return check_launch();
}

int mmf_rotary_apply_grad(const float* grad_out, const float* cos_, const float* sin_, float* grad_x, long long rows, int D, void* stream) {
  if (!grad_out || !cos_ || !sin_ || !grad_x || rows < 0 || D <= 0 || (D & 1))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_rotary_apply_grad");
  launch_rotary_apply_grad(grad_out, cos_, sin_, grad_x, rows, D, (hipStream_t)stream);
  return check_launch();
}

int mmf_adaln_modulate(const float* x, const float* scale_shift, float* out, int B, int L, int D, void* stream) {
  if (!x || !scale_shift || !out || B <= 0 || L <= 0 || D <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_adaln_modulate");
  launch_adaln_modulate(x, scale_shift, out, B, L, D, (hipStream_t)stream);
  return check_launch();
}

int mmf_qkv_block(const float* x, const float* scale_shift, const float* Wq, const float* bq, const float* Wkv, const float* bkv,
                  const float* cos_, const float* sin_, float* q_out, float* k_out, float* v_out, int B, int L, int D, void* stream) {
  if (!x || !Wq || !bq || !Wkv || !bkv || !q_out || !k_out || !v_out || B <= 0 || L <= 0 || ((cos_ == nullptr) != (sin_ == nullptr)))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_qkv_block");
  if (launch_qkv_block(x, scale_shift, Wq, bq, Wkv, bkv, cos_, sin_, q_out, k_out, v_out, B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_qkv_block is built for D = 120");
  return check_launch();
}

int mmf_out_ffn_block(const float* att, const float* residual, const float* Wo, const float* bo, const float* ln1_weight,
                      const float* ln1_bias, float ln1_eps, const float* scale_shift, const float* W1, const float* b1, const float* W2,
                      const float* b2, const float* ln2_weight, const float* ln2_bias, float ln2_eps, float* out, int B, int L, int D,
                      void* stream) {
  if (!att || !residual || !Wo || !bo || !ln1_weight || !ln1_bias || !W1 || !b1 || !W2 || !b2 || !ln2_weight || !ln2_bias || !out || B <= 0 ||
      L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_out_ffn_block");
  if (launch_out_ffn_block(att, residual, Wo, bo, ln1_weight, ln1_bias, ln1_eps, scale_shift, W1, b1, W2, b2, ln2_weight, ln2_bias, ln2_eps, out,
                           B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_block is built for D = 120");
  return check_launch();
}

int mmf_qkv_heads(const float* x, const float* scale_shift, const float* Wq, const float* bq, const float* Wkv, const float* bkv,
                  const float* cos_, const float* sin_, float* q_heads, float* k_heads, float* v_heads_t, int B, int L, int D, int H, int roles,
                  void* stream) {
  const bool need_q = (roles & 1) != 0, need_kv = (roles & 6) != 0;
  if (!x || B <= 0 || L <= 0 || (roles != 7 && roles != 1 && roles != 6) || (need_q && (!Wq || !bq || !q_heads)) ||
      (need_kv && (!Wkv || !bkv || !k_heads || !v_heads_t)) || ((cos_ == nullptr) != (sin_ == nullptr)))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_qkv_heads");
  if (launch_qkv_heads(x, scale_shift, Wq, bq, Wkv, bkv, cos_, sin_, q_heads, k_heads, v_heads_t, B, L, D, H, roles, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_qkv_heads is built for D = 120, H = 8");
  return check_launch();
}

int mmf_attention_heads(const float* q_heads, const float* k_heads, const float* v_heads_t, const uint8_t* key_padding, float* out, int B,
                        int Lq, int Lk, int H, int head_dim, void* stream) {
  if (!q_heads || !k_heads || !v_heads_t || !out || B <= 0 || Lq <= 0 || Lk <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_attention_heads");
  if (launch_attention_heads(q_heads, k_heads, v_heads_t, key_padding, out, B, Lq, Lk, H, head_dim, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_attention_heads is built for H = 8, head_dim = 15");
  return check_launch();
}

int mmf_out_ffn_mfma(const float* att, const float* residual, const float* Wo, const float* bo, const float* ln1_weight,
                     const float* ln1_bias, float ln1_eps, const float* scale_shift, const float* W1, const float* b1, const float* W2,
                     const float* b2, const float* ln2_weight, const float* ln2_bias, float ln2_eps, float* out, int B, int L, int D,
                     void* stream) {
  if (!att || !residual || !Wo || !bo || !ln1_weight || !ln1_bias || !W1 || !b1 || !W2 || !b2 || !ln2_weight || !ln2_bias || !out || B <= 0 ||
      L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_out_ffn_mfma");
  if (launch_out_ffn_mfma(att, residual, Wo, bo, ln1_weight, ln1_bias, ln1_eps, scale_shift, W1, b1, W2, b2, ln2_weight, ln2_bias, ln2_eps, out,
                          B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_mfma is built for D = 120");
  return check_launch();
}

int mmf_step_prologue(const float* trajectory, int B, int num_tokens, const float* traj_encoder_wt, const float* traj_encoder_bias,
                      const float* position_table, const float* time_embedding, const float* history, const float* rotary_freq,
                      const float* adaln_wt, const float* adaln_bias, int adaln_width, float* tokens_out, float* adaln_out, float* cos_out,
                      float* sin_out, long long rotary_batch_stride, int D, void* stream) {
  const bool ada = adaln_width > 0;  // adaln_width 0: tokens and rotary codes only
  if (!trajectory || !traj_encoder_wt || !traj_encoder_bias || !position_table || !rotary_freq || !tokens_out || !cos_out || !sin_out ||
      B <= 0 || num_tokens <= 0 || adaln_width < 0 || (ada && (!time_embedding || !history || !adaln_wt || !adaln_bias || !adaln_out)))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_step_prologue");
  if (D != 120) return fail(MMF_ERR_INVALID_ARG, "mmf_step_prologue is built for D = 120");
  launch_step_prologue(trajectory, B, num_tokens, traj_encoder_wt, traj_encoder_bias, position_table, time_embedding, history, rotary_freq,
                       adaln_wt, adaln_bias, adaln_width, tokens_out, adaln_out, cos_out, sin_out, rotary_batch_stride, (hipStream_t)stream);
  return check_launch();
}

int mmf_head_outputs(const float* rotation_seq, const float* position_seq, long long seq_batch_stride, int B, int L, int G,
                     const float* const* weights20, float* pred_out, float* head_yaw_out, int D, void* stream) {
  if (!rotation_seq || !position_seq || !weights20 || !pred_out || B <= 0 || L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_head_outputs");
  for (int i = 0; i < 16; ++i)
    if (!weights20[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_head_outputs: missing weight");
  if (weights20[16] && (!weights20[17] || !weights20[18] || !weights20[19] || !head_yaw_out))
    return fail(MMF_ERR_INVALID_ARG, "mmf_head_outputs: incomplete head-yaw arguments");
  if (D != 120 || launch_head_outputs(rotation_seq, position_seq, seq_batch_stride, B, L, G, weights20, pred_out, head_yaw_out,
                                      (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_head_outputs is built for D = 120 and at most 4 grippers");
  return check_launch();
}

int mmf_step_tail(const float* rotation_seq, const float* position_seq, long long seq_batch_stride, int B, int L, int G,
                  const float* const* weights20, float* pred_out, float* head_yaw_out, const float* trajectory, const float* noise,
                  const float* coef_pos6, const float* coef_rot6, float* trajectory_out, const float* traj_encoder_wt,
                  const float* traj_encoder_bias, const float* position_table, const float* rotary_freq, float* tokens_out, float* cos_out,
                  float* sin_out, long long rotary_batch_stride, int D, void* stream) {
  if (!rotation_seq || !position_seq || !weights20 || !pred_out || !trajectory || !noise || !coef_pos6 || !coef_rot6 || !trajectory_out ||
      B <= 0 || L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_step_tail");
  if (tokens_out && (!traj_encoder_wt || !traj_encoder_bias || !position_table || !rotary_freq || !cos_out || !sin_out))
    return fail(MMF_ERR_INVALID_ARG, "mmf_step_tail: incomplete next-step arguments");
  for (int i = 0; i < 16; ++i)
    if (!weights20[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_step_tail: missing weight");
  if (weights20[16] && (!weights20[17] || !weights20[18] || !weights20[19] || !head_yaw_out))
    return fail(MMF_ERR_INVALID_ARG, "mmf_step_tail: incomplete head-yaw arguments");
  if (D != 120 || launch_step_tail(rotation_seq, position_seq, seq_batch_stride, B, L, G, weights20, pred_out, head_yaw_out, trajectory, noise,
                                   coef_pos6, coef_rot6, trajectory_out, traj_encoder_wt, traj_encoder_bias, position_table, rotary_freq,
                                   tokens_out, cos_out, sin_out, rotary_batch_stride, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_step_tail is built for D = 120 and at most 4 grippers");
  return check_launch();
}

int mmf_out_ffn_qkv(const float* const* layer13, float ln1_eps, float ln2_eps, float* out, const float* const* next7, float* q_heads,
                    float* k_heads, float* v_heads_t, int B, int L, int D, int H, int roles, const float* att_partials, int n_split,
                    void* stream) {
  const bool need_q = (roles & 1) != 0, need_kv = (roles & 6) != 0;
  if (!layer13 || !next7 || !out || B <= 0 || L <= 0 || (roles != 7 && roles != 1) || (need_q && !q_heads) ||
      (need_kv && (!k_heads || !v_heads_t)) || (att_partials && n_split < 1))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_out_ffn_qkv");
  for (int i = att_partials ? 1 : 0; i < 13; ++i)
    if (i != 6 && !layer13[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_qkv: missing layer operand");  // [6] = scale_shift, optional
  if (!next7[1] || !next7[2] || (need_kv && (!next7[3] || !next7[4])) || ((next7[5] == nullptr) != (next7[6] == nullptr)))
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_qkv: missing next-layer operand");
  if (launch_out_ffn_qkv(layer13, ln1_eps, ln2_eps, out, next7, q_heads, k_heads, v_heads_t, B, L, D, H, roles, att_partials, n_split,
                         (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_qkv is built for D = 120, H = 8 (and L <= 16 with partials)");
  return check_launch();
}

int mmf_debug_wg_trace(uint64_t* buffer_dev, int capacity_records) {
  // buffer: {id, start, end} triples in 100 MHz ticks at slot (id / 10 - 1) * 8192 + workgroup index (zero it before the frame of
  // interest); null switches the trace off.  Synchronises the device (symbol copies).
  if (buffer_dev && capacity_records <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_debug_wg_trace");
  const int r0 = set_wg_trace_map(reinterpret_cast<unsigned long long*>(buffer_dev), capacity_records);
  const int r1 = set_wg_trace_app(reinterpret_cast<unsigned long long*>(buffer_dev), capacity_records);
  int r2 = set_wg_trace_policy(reinterpret_cast<unsigned long long*>(buffer_dev), capacity_records);
  if (r2 == 0) r2 = set_wg_trace_policy_layer(reinterpret_cast<unsigned long long*>(buffer_dev), capacity_records);
  if (r0 == 2 || r1 == 2 || r2 == 2) return fail(MMF_ERR_INVALID_ARG, "mmf_debug_wg_trace: this library was built without the hooks (make WG_TRACE=1)");
  if (r0 != 0 || r1 != 0 || r2 != 0) return fail(MMF_ERR_HIP, "mmf_debug_wg_trace: hipMemcpyToSymbol failed");
  return MMF_OK;
}

int mmf_qkv_heads2(const float* x0, const float* x1, const float* const* next14, float* q_heads, float* k_heads, float* v_heads_t, int B,
                   int L, int D, int H, void* stream) {
  if (!x0 || !x1 || !next14 || !q_heads || !k_heads || !v_heads_t || B <= 0 || L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_qkv_heads2");
  for (int st = 0; st < 2; ++st) {
    const float* const* q = next14 + 7 * st;
    if (!q[1] || !q[2] || !q[3] || !q[4] || ((q[5] == nullptr) != (q[6] == nullptr)))
      return fail(MMF_ERR_INVALID_ARG, "mmf_qkv_heads2: missing operand");
  }
  if (launch_qkv_heads2(x0, x1, next14, q_heads, k_heads, v_heads_t, B, L, D, H, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_qkv_heads2 is built for D = 120, H = 8");
  return check_launch();
}

int mmf_out_ffn_mfma2(const float* const* layer26, const float* eps4, float* out, int B, int L, int D, void* stream) {
  if (!layer26 || !eps4 || !out || B <= 0 || L <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_out_ffn_mfma2");
  for (int i = 0; i < 26; ++i)
    if (i % 13 != 6 && !layer26[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_mfma2: missing operand");  // [6]: scale_shift, optional
  if (launch_out_ffn_mfma2(layer26, eps4, out, B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_mfma2 is built for D = 120");
  return check_launch();
}

int mmf_out_ffn_qkv2(const float* const* layer26, const float* eps4, float* out, const float* const* next14, float* q_heads, float* k_heads,
                     float* v_heads_t, int B, int L, int D, int H, void* stream) {
  if (!layer26 || !eps4 || !out || !next14 || !q_heads || !k_heads || !v_heads_t || B <= 0 || L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_out_ffn_qkv2");
  for (int i = 0; i < 26; ++i)
    if (i % 13 != 6 && !layer26[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_qkv2: missing layer operand");  // [6]: scale_shift, optional
  for (int st = 0; st < 2; ++st) {
    const float* const* q = next14 + 7 * st;
    if (!q[1] || !q[2] || !q[3] || !q[4] || ((q[5] == nullptr) != (q[6] == nullptr)))
      return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_qkv2: missing next-layer operand");
  }
  if (launch_out_ffn_qkv2(layer26, eps4, out, next14, q_heads, k_heads, v_heads_t, B, L, D, H, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_qkv2 is built for D = 120, H = 8");
  return check_launch();
}

int mmf_cross_layer(const float* const* layer13, float ln1_eps, float ln2_eps, float* out, const float* const* next7, float* q_heads_next,
                    const float* const* qkv3, const uint8_t* key_padding16, uint64_t* handover, uint32_t tag, int B, int Lq, int Lk, int D,
                    int H, void* stream) {
  if (!layer13 || !out || !qkv3 || !qkv3[0] || !qkv3[1] || !qkv3[2] || !handover || tag == 0 || B <= 0 || Lq <= 0 || Lk <= 0 ||
      (next7 && !q_heads_next))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_cross_layer");
  for (int i = 1; i < 13; ++i)
    if (i != 6 && !layer13[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_cross_layer: missing layer operand");  // [0] unused, [6] = scale_shift, optional
  if (next7 && (!next7[1] || !next7[2] || ((next7[5] == nullptr) != (next7[6] == nullptr))))
    return fail(MMF_ERR_INVALID_ARG, "mmf_cross_layer: missing next-layer operand");
  const size_t words = (size_t)B * H * 4 * 18 * 16;
  if (launch_cross_layer(layer13, ln1_eps, ln2_eps, out, next7, q_heads_next, qkv3, key_padding16, reinterpret_cast<unsigned long long*>(handover), tag,
                         reinterpret_cast<int*>(handover + words), B, Lq, Lk, D, H, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_cross_layer is built for D = 120, H = 8, Lq <= 16");
  return check_launch();
}

int mmf_self_layer(const float* const* layer13, float ln1_eps, float ln2_eps, float* out, const float* const* next7, float* q_heads_next,
                   float* k_heads_next, float* v_heads_t_next, const float* const* qkv3, const uint8_t* key_padding16, uint64_t* handover,
                   uint32_t tag, int B, int L, int D, int H, void* stream) {
  if (!layer13 || !out || !qkv3 || !qkv3[0] || !qkv3[1] || !qkv3[2] || !handover || tag == 0 || B <= 0 || L <= 0 ||
      (next7 && (!q_heads_next || !k_heads_next || !v_heads_t_next)))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_self_layer");
  for (int i = 1; i < 13; ++i)
    if (i != 6 && !layer13[i]) return fail(MMF_ERR_INVALID_ARG, "mmf_self_layer: missing layer operand");  // [0] unused, [6] = scale_shift, optional
  if (next7 && (!next7[1] || !next7[2] || !next7[3] || !next7[4] || ((next7[5] == nullptr) != (next7[6] == nullptr))))
    return fail(MMF_ERR_INVALID_ARG, "mmf_self_layer: missing next-layer operand");
  const size_t words = (size_t)B * L * D;
  if (launch_self_layer(layer13, ln1_eps, ln2_eps, out, next7, q_heads_next, k_heads_next, v_heads_t_next, qkv3, key_padding16,
                        reinterpret_cast<unsigned long long*>(handover), tag, reinterpret_cast<int*>(handover + words), B, L, D, H,
                        (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_self_layer is built for D = 120, H = 8");
  return check_launch();
}

int mmf_split_activations3(const float* x, int64_t rows, int K, void* out, void* stream) {
  if (!x || !out) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_split_activations3");
  if (launch_split_act3(x, rows, K, out, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_split_activations3: rows > 0 and K a positive multiple of 8");
  return check_launch();
}

int mmf_gelu_split_activations3(const float* x, int64_t rows, int K, void* out, void* stream) {
  if (!x || !out) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_gelu_split_activations3");
  if (launch_split_act3_src(1, x, rows, K, 0, 0, out, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_gelu_split_activations3: rows > 0 and K >= 64 a multiple of 8");
  return check_launch();
}

int mmf_split_attention_heads3(const float* att, int64_t B, int heads, int L, int head_dim, void* out, void* stream) {
  if (!att || !out || B <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_split_attention_heads3");
  if (launch_split_act3_src(2, att, B * (int64_t)L, heads * head_dim, heads, L, out, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_split_attention_heads3: head_dim a multiple of 8, heads * head_dim >= 64");
  return check_launch();
}

int mmf_layernorm_split_activations3(const float* x, const float* residual, const float* gamma, const float* beta, float eps, int64_t rows,
                                     int K, float* sum_out, void* out, void* stream) {
  if (!x || !gamma || !beta || !out) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_layernorm_split_activations3");
  if (launch_ln_split3(x, residual, gamma, beta, eps, rows, K, sum_out, out, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_layernorm_split_activations3: K in {256, 512, 768, 1024}; residual and sum_out come together");
  return check_launch();
}

int mmf_attention_split(const float* q, const float* k, const float* v, int64_t row_stride, int64_t batch_stride, int B, int H, int L,
                        int head_dim, float scale, void* out, int split_out, void* stream) {
  if (!q || !k || !v || !out) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_attention_split");
  if (launch_attention_split(q, k, v, row_stride, batch_stride, B, H, L, head_dim, scale, out, split_out, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_attention_split: head_dim 64, L a multiple of 128, strides multiples of 4 floats");
  return check_launch();
}

int64_t mmf_layernorm_train_scratch_bytes(void) { return (int64_t)ln_train_partials_bytes(); }

int mmf_layernorm_train_forward(const float* a, const float* b, const float* gamma, const float* beta, float eps, int64_t rows, int D,
                                float* sum_out, float* y, float* mean, float* rstd, void* stream) {
  if (!a || !gamma || !beta || !y || !mean || !rstd) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_layernorm_train_forward");
  if (launch_ln_train_fwd(a, b, gamma, beta, eps, rows, D, sum_out, y, mean, rstd, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_layernorm_train_forward: D a multiple of 4 up to 128; sum_out comes with b");
  return check_launch();
}

int mmf_layernorm_train_backward(const float* grad_y, const float* x, const float* gamma, const float* mean, const float* rstd, int64_t rows, int D,
                                 float* grad_x, float* grad_gamma, float* grad_beta, float* scratch, void* stream) {
  if (!grad_y || !x || !gamma || !mean || !rstd || !grad_x || !grad_gamma || !grad_beta || !scratch)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_layernorm_train_backward");
  if (launch_ln_train_bwd(grad_y, x, gamma, mean, rstd, rows, D, grad_x, grad_gamma, grad_beta, scratch, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_layernorm_train_backward: D a multiple of 4 up to 128");
  return check_launch();
}

int64_t mmf_adaln_modulate_grad_scratch_bytes(int B) { return B > 0 ? (int64_t)adaln_train_scratch_bytes(B) : 0; }

int mmf_adaln_modulate_grad(const float* grad_out, const float* x, const float* scale_shift, int B, int L, int D, float* grad_x,
                            float* grad_scale_shift, float* scratch, void* stream) {
  if (!grad_out || !x || !scale_shift || !grad_x || !grad_scale_shift || !scratch)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_adaln_modulate_grad");
  if (launch_adaln_train_bwd(grad_out, x, scale_shift, B, L, D, grad_x, grad_scale_shift, scratch, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_adaln_modulate_grad: D a multiple of 4 up to 128");
  return check_launch();
}

int64_t mmf_linear_weight_grad_scratch_bytes(int64_t rows, int out_features, int in_features) {
  return (rows > 0 && out_features > 0 && in_features > 0) ? (int64_t)linear_wgrad_scratch_bytes(rows, out_features, in_features) : 0;
}

int mmf_linear_weight_grad(const float* grad_out, const float* x, int64_t rows, int out_features, int in_features, float* grad_weight,
                           float* grad_bias, float* scratch, void* stream) {
  if (!grad_out || !x || !grad_weight || !scratch) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_linear_weight_grad");
  if (launch_linear_wgrad(grad_out, x, rows, out_features, in_features, grad_weight, grad_bias, scratch, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_linear_weight_grad: out_features <= 256, in_features <= 128");
  return check_launch();
}

int mmf_train_attention_forward(const float* q, const float* k, const float* v, const int64_t* strides6, const uint8_t* key_padding, int B, int H,
                                int Lq, int Lk, int head_dim, float scale, float* out, float* lse, void* stream) {
  if (!q || !k || !v || !strides6 || !out || !lse) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_train_attention_forward");
  const long long st[6] = {strides6[0], strides6[1], strides6[2], strides6[3], strides6[4], strides6[5]};
  if (launch_train_attention_fwd(q, k, v, st, key_padding, B, H, Lq, Lk, head_dim, scale, out, lse, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_train_attention_forward: 1 <= head_dim <= 16, positive sizes");
  return check_launch();
}

int mmf_train_attention_backward(const float* q, const float* k, const float* v, const int64_t* strides6, const uint8_t* key_padding, int B, int H,
                                 int Lq, int Lk, int head_dim, float scale, const float* out, const float* dout, const float* lse, float* dsum,
                                 float* dq, float* dk, float* dv, void* stream) {
  if (!q || !k || !v || !strides6 || !out || !dout || !lse || !dsum || !dq || !dk || !dv)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_train_attention_backward");
  const long long st[6] = {strides6[0], strides6[1], strides6[2], strides6[3], strides6[4], strides6[5]};
  if (launch_train_attention_bwd(q, k, v, st, key_padding, B, H, Lq, Lk, head_dim, scale, out, dout, lse, dsum, dq, dk, dv, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_train_attention_backward: 1 <= head_dim <= 16, positive sizes");
  return check_launch();
}

int mmf_split_linear_weight(const float* weight, int out_features, int in_features, void* split, void* stream) {
  if (!weight || !split || out_features <= 0) return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_split_linear_weight");
  if (launch_split_weight(weight, out_features, in_features, split, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_split_linear_weight is built for in_features = 120");
  return check_launch();
}

int mmf_attention_heads_split(const float* q_heads, const float* k_heads, const float* v_heads_t, const uint8_t* key_padding, float* partials,
                              int B, int Lq, int Lk, int H, int head_dim, int* n_split_out, void* stream) {
  if (!q_heads || !k_heads || !v_heads_t || !n_split_out || B <= 0 || Lq <= 0 || Lk <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_attention_heads_split");
  if (!partials) {  // size query: partials = float [B, H, n_split, 18, 16]
    *n_split_out = 4;
    return MMF_OK;
  }
  const int rc = launch_attention_heads_split(q_heads, k_heads, v_heads_t, key_padding, partials, B, Lq, Lk, H, head_dim, (hipStream_t)stream);
  if (rc == 1) return fail(MMF_ERR_INVALID_ARG, "mmf_attention_heads_split is built for H = 8, head_dim = 15, Lq <= 16");
  *n_split_out = rc >> 8;
  return check_launch();
}

int mmf_out_ffn_mfma_partials(const float* partials, int n_split, const float* residual, const float* Wo, const float* bo,
                              const float* ln1_weight, const float* ln1_bias, float ln1_eps, const float* scale_shift, const float* W1,
                              const float* b1, const float* W2, const float* b2, const float* ln2_weight, const float* ln2_bias, float ln2_eps,
                              float* out, int B, int L, int D, void* stream) {
  if (!partials || !residual || !Wo || !bo || !ln1_weight || !ln1_bias || !W1 || !b1 || !W2 || !b2 || !ln2_weight || !ln2_bias || !out ||
      B <= 0 || L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_out_ffn_mfma_partials");
  if (launch_out_ffn_mfma_partials(partials, n_split, residual, Wo, bo, ln1_weight, ln1_bias, ln1_eps, scale_shift, W1, b1, W2, b2, ln2_weight,
                                   ln2_bias, ln2_eps, out, B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_out_ffn_mfma_partials is built for D = 120, L <= 16");
  return check_launch();
}

int mmf_ffn_block(const float* x, const float* scale_shift, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* ln_weight, const float* ln_bias, float ln_eps, float* out, int B, int L, int D, void* stream) {
  if (!x || !W1 || !b1 || !W2 || !b2 || !ln_weight || !ln_bias || !out || B <= 0 || L <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_ffn_block");
  if (launch_ffn_block(x, scale_shift, W1, b1, W2, b2, ln_weight, ln_bias, ln_eps, out, B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_ffn_block is built for D = 120");
  return check_launch();
}

int mmf_q_block(const float* x, const float* scale_shift, const float* Wq, const float* bq, const float* cos_, const float* sin_,
                float* out, int B, int L, int D, void* stream) {
  if (!x || !Wq || !bq || !out || B <= 0 || L <= 0 || ((cos_ == nullptr) != (sin_ == nullptr)))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_q_block");
  if (launch_q_block(x, scale_shift, Wq, bq, cos_, sin_, out, B, L, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_q_block is built for D = 120");
  return check_launch();
}

int mmf_kv_block(const float* memory, const float* Wkv, const float* bkv, const float* cos_, const float* sin_, float* k_out,
                 float* v_out, long long tokens, int D, void* stream) {
  if (!memory || !Wkv || !bkv || !k_out || !v_out || tokens <= 0 || ((cos_ == nullptr) != (sin_ == nullptr)))
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_kv_block");
  if (launch_kv_block(memory, Wkv, bkv, cos_, sin_, k_out, v_out, tokens, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_kv_block is built for D = 120");
  return check_launch();
}

int mmf_attn_out_block(const float* att, const float* residual, const float* Wo, const float* bo, const float* ln_weight,
                       const float* ln_bias, float ln_eps, float* out, long long tokens, int D, void* stream) {
  if (!att || !residual || !Wo || !bo || !ln_weight || !ln_bias || !out || tokens <= 0)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_attn_out_block");
  if (launch_attn_out_block(att, residual, Wo, bo, ln_weight, ln_bias, ln_eps, out, tokens, D, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_attn_out_block is built for D = 120");
  return check_launch();
}

int mmf_ddpm_step(const float* x, const float* eps, long long eps_row_stride, const float* noise, float* out, long long rows, int C,
                  int split, const float* coef_a_host6, const float* coef_b_host6, void* stream) {
  if (!x || !eps || !noise || !out || !coef_a_host6 || !coef_b_host6 || rows < 0 || C <= 0 || split < 0 || split > C || eps_row_stride < C)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_ddpm_step");
  launch_ddpm_step(x, eps, eps_row_stride, noise, out, rows, C, split, coef_a_host6, coef_b_host6, (hipStream_t)stream);
  return check_launch();
}

int mmf_attention_small(const float* q, const float* k, long long k_row_stride, const float* v, long long v_row_stride,
                        const uint8_t* key_padding, float* out, int B, int Lq, int Lk, int heads, int head_dim, void* stream) {
  if (!q || !k || !v || !out || k_row_stride < (long long)heads * head_dim || v_row_stride < (long long)heads * head_dim)
    return fail(MMF_ERR_INVALID_ARG, "bad arguments to mmf_attention_small");
  if (launch_attention_small(q, k, k_row_stride, v, v_row_stride, key_padding, out, B, Lq, Lk, heads, head_dim, (hipStream_t)stream) != 0)
    return fail(MMF_ERR_INVALID_ARG, "mmf_attention_small supports head_dim 8, 15, 16, 20, 24, 32");
  return check_launch();
}

}  // extern "C"
