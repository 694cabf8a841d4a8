// mmf_launch_policy.h -- host-side launch interface of the policy-side kernel files (mmf_kernels_{fps,policy*,backbone,train_*}.hip).
// Included by mmf_launch.h; kept apart from it because mmf_launch.h also defines the argument blocks of the fusion kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mmf {

// mmf_kernels_fps.hip
int launch_fps(const float* x, int B, int N, int C, int npoints, int start, long long* out_idx, hipStream_t s, void* workspace = nullptr,
               size_t workspace_bytes = 0);
size_t fps_workspace_bytes(int B, int N, int C);

// mmf_kernels_policy.hip (inference-side fused ops of the diffusion head)
void launch_rotary_apply(const float* x, long long x_stride, const float* cs, const float* sn, float* out, long long rows, int D,
                         hipStream_t s);
void launch_rotary_apply_grad(const float* g, const float* cs, const float* sn, float* dx, long long rows, int D, hipStream_t s);
void launch_adaln_modulate(const float* x, const float* ss, float* out, int B, int L, int D, hipStream_t s);
int launch_qkv_block(const float* x, const float* ss, const float* Wq, const float* bq, const float* Wkv, const float* bkv, const float* cs,
                     const float* sn, float* qout, float* kout, float* vout, int B, int L, int D, hipStream_t s);
int launch_out_ffn_block(const float* att, const float* res, const float* Wo, const float* bo, const float* g1, const float* be1, float eps1,
                         const float* ss, const float* W1, const float* b1, const float* W2, const float* b2, const float* g2,
                         const float* be2, float eps2, float* out, int B, int L, int D, hipStream_t s);
int launch_ffn_block(const float* x, const float* ss, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* gamma, const float* beta, float eps, float* out, int B, int L, int D, hipStream_t s);
int launch_q_block(const float* x, const float* ss, const float* Wq, const float* bq, const float* cs, const float* sn, float* out,
                   int B, int L, int D, hipStream_t s);
int launch_kv_block(const float* m, const float* Wkv, const float* bkv, const float* cs, const float* sn, float* kout, float* vout,
                    long long tokens, int D, hipStream_t s);
int launch_attn_out_block(const float* att, const float* res, const float* Wo, const float* bo, const float* gamma, const float* beta,
                          float eps, float* out, long long tokens, int D, hipStream_t s);
void launch_ddpm_step(const float* x, const float* eps, long long eps_stride, const float* noise, float* out, long long rows, int C,
                      int split, const float* coefA, const float* coefB, hipStream_t s);
int launch_attention_small(const float* q, const float* k, long long k_stride, const float* v, long long v_stride, const uint8_t* pad,
                           float* out, int B, int Lq, int Lk, int H, int d, hipStream_t s);

// mmf_kernels_policy_mfma.hip (matrix-core forms: head-major q/k/v, attention over them, out_proj + LN + FFN)
int launch_qkv_heads(const float* x, const float* ss, const float* WqT, const float* bq, const float* WkvT, const float* bkv,
                     const float* cs, const float* sn, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H, int roles,
                     hipStream_t s);
int launch_out_ffn_qkv(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp, float* Kp,
                       float* Vt, int B, int L, int D, int H, int roles, const float* partials, int n_split, hipStream_t s);
int launch_attention_heads_split(const float* Qp, const float* Kp, const float* Vt, const uint8_t* pad, float* partials, int B, int Lq, int Lk,
                                 int H, int dh, hipStream_t s);
int launch_out_ffn_mfma_partials(const float* partials, int n_split, const float* res, const float* WoT, const float* bo, const float* g1,
                                 const float* be1, float eps1, const float* ss, const float* W1T, const float* b1, const float* W2T,
                                 const float* b2, const float* g2, const float* be2, float eps2, float* out, int B, int L, int D,
                                 hipStream_t s);
int launch_qkv_heads2(const float* x0, const float* x1, const float* const* q14, float* Qp, float* Kp, float* Vt, int B, int L, int D, int H,
                      hipStream_t s);
int launch_out_ffn_mfma2(const float* const* a26, const float* eps4, float* out, int B, int L, int D, hipStream_t s);
int launch_out_ffn_qkv2(const float* const* a26, const float* eps4, float* out, const float* const* q14, float* Qp, float* Kp, float* Vt,
                        int B, int L, int D, int H, hipStream_t s);
int launch_split_weight(const float* W, int rows, int cols, void* dst, hipStream_t s);
int launch_split_act3(const float* x, long long rows, int K, void* out, hipStream_t s);
int launch_attention_split(const float* q, const float* k, const float* v, long long row_stride, long long batch_stride, int B, int H, int L,
                           int head_dim, float scale, void* out, int split_out, hipStream_t s);
int launch_split_act3_src(int src, const float* x, long long rows, int K, int heads, int L, void* out, hipStream_t s);
size_t ln_train_partials_bytes();
size_t linear_wgrad_scratch_bytes(long long R, int N, int K);
int launch_linear_wgrad(const float* g, const float* x, long long R, int N, int K, float* dW, float* db, float* partials, hipStream_t s);
size_t adaln_train_scratch_bytes(int B);
int launch_adaln_train_bwd(const float* g, const float* x, const float* ss, int B, int L, int D, float* dx, float* dss, float* partials,
                           hipStream_t s);
int launch_ln_train_fwd(const float* a, const float* b, const float* gamma, const float* beta, float eps, long long rows, int D, float* s_out,
                        float* y, float* mean, float* rstd, hipStream_t s);
int launch_ln_train_bwd(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, long long rows, int D, float* dx,
                        float* dgamma, float* dbeta, float* partials, hipStream_t s);
int launch_train_attention_fwd(const float* q, const float* k, const float* v, const long long* strides6, const uint8_t* pad, int B, int H,
                               int Lq, int Lk, int hd, float scale, float* out, float* lse, hipStream_t s);
int launch_train_attention_bwd(const float* q, const float* k, const float* v, const long long* strides6, const uint8_t* pad, int B, int H,
                               int Lq, int Lk, int hd, float scale, const float* out, const float* dout, const float* lse, float* dsum,
                               float* dq, float* dk, float* dv, hipStream_t s);
int launch_ln_split3(const float* x, const float* y, const float* gamma, const float* beta, float eps, long long rows, int K, float* sum_out,
                     void* out, hipStream_t s);
int launch_self_layer(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp, float* Kp, float* Vt,
                      const float* const* qkv3, const uint8_t* pad, unsigned long long* tagged, unsigned tag, int* fail, int B, int L, int D,
                      int H, hipStream_t s);
int launch_cross_layer(const float* const* args13, float eps1, float eps2, float* out, const float* const* next7, float* Qp_next,
                       const float* const* qkv3, const uint8_t* pad, unsigned long long* tagged, unsigned tag, int* fail, int B, int Lq,
                       int Lk, int D, int H, hipStream_t s);
int launch_attention_heads(const float* Qp, const float* Kp, const float* Vt, const uint8_t* pad, float* out, int B, int Lq, int Lk,
                           int H, int dh, hipStream_t s);
int launch_out_ffn_mfma(const float* att, const float* res, const float* WoT, const float* bo, const float* g1, const float* be1,
                        float eps1, const float* ss, const float* W1T, const float* b1, const float* W2T, const float* b2,
                        const float* g2, const float* be2, float eps2, float* out, int B, int L, int D, hipStream_t s);

// mmf_kernels_policy_head.hip (what runs before the first and after the last attention layer of a denoising step)
void launch_step_prologue(const float* traj, int B, int nt, const float* WeT, const float* be, const float* pos_table, const float* time_row,
                          const float* history, const float* freq, const float* AwT, const float* Ab, int NA, float* tokens, float* adaln,
                          float* cos_out, float* sin_out, long long rot_batch_stride, hipStream_t s);
int launch_head_outputs(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G, const float* const* w,
                        float* pred, float* head_yaw, hipStream_t s);
int launch_step_tail(const float* rot_seq, const float* pos_seq, long long seq_batch_stride, int B, int L, int G, const float* const* w,
                     float* pred, float* head_yaw, const float* traj, const float* noise, const float* coef_pos, const float* coef_rot,
                     float* traj_out, const float* WeT, const float* be, const float* pos_table, const float* freq, float* tokens_out,
                     float* cos_out, float* sin_out, long long rot_batch_stride, hipStream_t s);

}  // namespace mmf
