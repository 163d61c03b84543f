/* libtcx_hip.so — C ABI of the MI355X (gfx950) kernels behind the TrajectoryCrafter denoising path.
 *
 * The reference (alekseizhuravlev/TrajectoryCrafter) has no native code and no FFI: every kernel
 * below replaces an *implicit* torch/cuDNN/flash-SDPA dispatch inside the reference's Python hot
 * path.  Each entry point cites the reference statement(s) it replaces (file:line relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (face B2)
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller
 *   - bf16 tensors are raw uint16 storage; "row" tensors are row-major with the stated strides
 *     (strides are in ELEMENTS)
 *   - asynchronous: kernels are enqueued on `stream` (a hipStream_t passed as void*); the library
 *     never allocates, frees or synchronises
 *   - return value: 0 on success, a negative TCX_E_* code on argument errors, or a positive
 *     hipError_t from the launch; tcx_last_error_string() gives a thread-local message
 *   - stateless and re-entrant; safe from several host threads on different streams / devices
 */
#ifndef TCX_HIP_H
#define TCX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCX_OK 0
#define TCX_E_SHAPE (-1)      /* unsupported / inconsistent shape            */
#define TCX_E_DTYPE (-2)      /* unsupported dtype enum                      */
#define TCX_E_ALIGN (-3)      /* pointer or stride not aligned as required   */
#define TCX_E_NULL (-4)       /* required pointer is null                    */

#define TCX_BF16 0
#define TCX_F32 1

/* ---- library info ---------------------------------------------------------------------- */
int tcx_version(void);                       /* ABI version, currently 1 */
const char* tcx_last_error_string(void);     /* thread-local description of the last failure */
/* Fills {CU count, LDS bytes per CU, wavefront size, gcnArch major*100+minor*10+step}. */
int tcx_device_info(int device, int32_t out[4]);

/* ---- K1 / K3: fused attention forward -------------------------------------------------------
 * O = softmax(scale * Q K^T) V, no mask, online softmax in fp32, P rounded to bf16 before PV.
 * Replaces: F.scaled_dot_product_attention inside diffusers CogVideoXAttnProcessor2_0, called
 *   at models/crosstransformer3d.py:239-243 (joint text+video self-attention, D=64, scale 1/8)
 *   and the (q*s)@(k*s)^T -> softmax -> @v chain of PerceiverCrossAttention.forward,
 *   models/crosstransformer3d.py:391-395 (D=128; q,k pre-scaled -> scale = 1).
 * q/k/v/o: bf16 [B, S, H, D] views: element (b,s,h,d) at b*stride_b + s*stride_s + h*stride_h + d.
 * D in {64, 128}; all strides multiples of 8 elements and pointers 16-byte aligned.
 * flags: TCX_ATTN_LOG2_SCORES — Q K^T is already the base-2 exponent (q was stored pre-multiplied by
 *   scale*log2(e) by tcx_qk_layernorm_rope): P = exp2(Q K^T - max), `scale` must be 1.  The self-attention
 *   product path uses it (it removes one FMA per score from a VALU-bound loop); the row sum is then taken
 *   over the bf16-rounded probabilities (on the matrix pipe).
 * k_sqmax (optional, fp32 [B*H], only read with TCX_ATTN_LOG2_SCORES and D = 64): max over the keys of |k|^2 per
 *   (batch, head), as written by tcx_qk_layernorm_rope.  It lets the kernel centre the softmax of a query row on
 *   the Cauchy-Schwarz bound |q| * max|k| instead of tracking a running max (no max / rescale work in the loop);
 *   rows whose bound is too large for that to be provably underflow-free use the exact tracking loop.
 * out_dtype: TCX_BF16 (product path) or TCX_F32 (test-only higher precision output). */
#define TCX_ATTN_LOG2_SCORES 1
/* TCX_ATTN_BOUND_PROVEN (with TCX_ATTN_LOG2_SCORES and k_sqmax): the caller GUARANTEES that 1.002 * |q_row| * sqrt(k_sqmax[b,h]) + 1e-3
 * < 60 for every query row, e.g. analytically from the q/k LayerNorm parameters: |LN(x)|_2 <= sqrt(D), so |q| <= q_scale * (sqrt(D) *
 * max|gamma_q| + |beta_q|_2), RoPE being a rotation (what CrossTransformer3D's Attention does, once per weight version).  The
 * kernel then skips the per-workgroup test and the launch of the exact kernel on the (empty) complement: -63 us per call at the
 * product shape.  A false guarantee cannot corrupt memory, but rows whose scores sit more than ~120 below the bound underflow. */
#define TCX_ATTN_BOUND_PROVEN 2
int tcx_attn_fwd(const void* q, const void* k, const void* v, void* o,
                 int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D,
                 int64_t q_stride_b, int64_t q_stride_s, int64_t q_stride_h,
                 int64_t k_stride_b, int64_t k_stride_s, int64_t k_stride_h,
                 int64_t v_stride_b, int64_t v_stride_s, int64_t v_stride_h,
                 int64_t o_stride_b, int64_t o_stride_s, int64_t o_stride_h,
                 float scale, int32_t flags, const float* k_sqmax, int32_t out_dtype, void* stream);
/* Same call with a caller-owned scratch buffer (the product path).  With `workspace` (16-byte aligned, at least
 * tcx_attn_fwd_workspace_bytes(...) bytes, contents irrelevant before and after) the bound-centred D = 64 launch balances its
 * last, partly filled round of workgroups: the B*H*ceil(Sq/256) workgroups run in ceil(./#CU) rounds and the remainder (64 of
 * 6720 on 256 CUs at 49f 480x720 = a quarter-filled 27th round) is cut into #CU/remainder parts along the keys; a part writes its
 * un-normalised fp32 O and row sum to the workspace and a small second kernel adds the parts (every part exponentiates against
 * the same origin |q| max|k|, so the combination is a plain sum).  Null / too small workspace: exactly tcx_attn_fwd.
 * tcx_attn_fwd_workspace_bytes returns 0 when the call would not split (then no workspace is needed). */
int64_t tcx_attn_fwd_workspace_bytes(int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D, int32_t flags,
                                     int32_t has_k_sqmax, int32_t out_dtype);
int tcx_attn_fwd_ws(const void* q, const void* k, const void* v, void* o,
                    int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D,
                    int64_t q_stride_b, int64_t q_stride_s, int64_t q_stride_h,
                    int64_t k_stride_b, int64_t k_stride_s, int64_t k_stride_h,
                    int64_t v_stride_b, int64_t v_stride_s, int64_t v_stride_h,
                    int64_t o_stride_b, int64_t o_stride_s, int64_t o_stride_h,
                    float scale, int32_t flags, const float* k_sqmax, int32_t out_dtype,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- K2: per-head LayerNorm(D=64) on q and k + 3-D RoPE on the video tokens, in place ---------
 * Replaces: attn.norm_q / attn.norm_k (LayerNorm(64, eps 1e-6, affine)) and apply_rotary_emb on
 *   query[:, :, text_len:] / key[:, :, text_len:] in diffusers CogVideoXAttnProcessor2_0 (module
 *   built at models/crosstransformer3d.py:199-208; tables from
 *   models/pipeline_trajectorycrafter.py:616-649).
 * q, k: bf16 [B, S, H, 64] views (same stride convention as tcx_attn_fwd).  gamma/beta: bf16 [64].
 * cos/sin: fp32 [S - text_len, 64] (may be null -> no rotation).  fp32 math, one rounding.
 * q_scale multiplies q (not k) in fp32 before that rounding: the self-attention path passes
 * head_dim^-1/2 * log2(e) so that the attention kernel consumes base-2 scores (1.0 = plain).
 * k_sqmax (optional, fp32 [B*H]): receives max over the tokens of |k|^2 of the ROUNDED keys per (batch, head)
 * (zeroed by this call on the stream, then float atomic max); input of tcx_attn_fwd's bound-centred loop. */
int tcx_qk_layernorm_rope(void* q, void* k,
                          int32_t B, int32_t S, int32_t H, int32_t D,
                          int64_t stride_b, int64_t stride_s, int64_t stride_h,
                          const void* gamma_q, const void* beta_q, const void* gamma_k, const void* beta_k,
                          const float* cos, const float* sin, int32_t text_len, float eps, float q_scale,
                          float* k_sqmax, void* stream);

/* ---- K4: LayerNorm (+ optional AdaLN modulate) over rows of C channels ------------------------
 * y = LN(x) * gamma + beta, then optionally y = y * (1 + scale[b]) + shift[b], with separate
 * (shift, scale) for the first `text_len` rows of each batch item (text) and the rest (video).
 * Replaces: diffusers CogVideoXLayerNormZero.forward (norm1/norm2, models/crosstransformer3d.py:
 *   195-197,211-213,234-236,251-253), nn.LayerNorm norm_final (:553,849), diffusers AdaLayerNorm
 *   norm_out (:556-562,856) and PerceiverCrossAttention.norm1/norm2 (:314-315,379-380).
 * x, y: bf16 [B, rows_per_batch, C] with batch strides (elements); rows contiguous (stride C).
 * gamma/beta: bf16 [C] or null.  shift_v/scale_v/shift_t/scale_t: bf16 [B, C] with stride
 * mod_stride_b between batch items, or null (no modulation).  C % 8 == 0, C <= 8192. */
int tcx_layernorm_modulate(const void* x, void* y, int32_t B, int32_t rows_per_batch, int32_t C,
                           int64_t x_stride_b, int64_t y_stride_b,
                           const void* gamma, const void* beta,
                           const void* shift_v, const void* scale_v,
                           const void* shift_t, const void* scale_t, int64_t mod_stride_b,
                           int32_t text_len, float eps, void* stream);

/* ---- K5: gated residual, in place ---------------------------------------------------------------
 * x[b, r, :] += gate[b, :] * y[b, r, :]   (gate_t for rows < text_len, gate_v otherwise;
 * null gates mean gate = 1: the plain cross-attention residual).
 * Replaces: models/crosstransformer3d.py:245-248, 261-264 (gated residuals) and :833-837. */
int tcx_gated_residual(void* x, const void* y, int32_t B, int32_t rows_per_batch, int32_t C,
                       int64_t x_stride_b, int64_t y_stride_b,
                       const void* gate_v, const void* gate_t, int64_t gate_stride_b,
                       int32_t text_len, void* stream);

/* ---- K6 epilogue: y = gelu_tanh(x + bias) over [rows, C] bf16 (bias bf16 [C] or null) ---------
 * Replaces: the GELU(approximate="tanh") of diffusers FeedForward net.0 (module built at
 *   models/crosstransformer3d.py:215-222, run :259). */
int tcx_bias_gelu_tanh(const void* x, const void* bias, void* y, int64_t rows, int32_t C, void* stream);

/* ---- elementwise helpers -------------------------------------------------------------------- */
/* y = bf16(x * s): the q*scale / k*scale of PerceiverCrossAttention (crosstransformer3d.py:391-392). */
int tcx_scale_bf16(const void* x, void* y, int64_t n, float s, void* stream);
/* y[b,s,h,:] = bf16(x[b,s,h,:] * scale) (y contiguous [B,S,H,D]; x a [B,S,H*D] view with free batch / row strides, e.g. the
 * k half of the to_kv output) and sqmax[b,h] = max_s |y[b,s,h,:]|^2: the k*scale of PerceiverCrossAttention (:392) fused with the
 * bound input (k_sqmax) of tcx_attn_fwd's bound-centred loop.  D in {64, 128}.  sqmax: fp32 [B,H], zeroed by the call. */
int tcx_scale_sqmax_bf16(const void* x, void* y, int32_t B, int32_t S, int32_t H, int32_t D, int64_t x_stride_b,
                         int64_t x_stride_s, float scale, float* sqmax, void* stream);
/* y = silu(x), bf16 -> bf16: the SiLU in front of every AdaLN linear (diffusers LayerNormZero/AdaLayerNorm). */
int tcx_silu_bf16(const void* x, void* y, int64_t n, void* stream);

/* ---- K8: patchify / unpatchify -------------------------------------------------------------------
 * patchify: gathers [B,F,C,H,W] (optionally the channel-concat of two tensors a:[.,Ca,.,.] and
 * b:[.,Cb,.,.]) into the im2col matrix [B*F*(H/p)*(W/p), (Ca+Cb)*p*p] (column order c,py,px =
 * Conv2d weight.flatten(1)) so the patch embedding is one GEMM.  k_stride (0 = dense) >= (Ca+Cb)*p*p is the row
 * length of `out`, zero-filled past the data: tcx_gemm_bf16 wants K % 128 == 0 (132 -> 256 for the 5B model).
 * Replaces: torch.concat + nn.Conv2d(k=p, stride=p) + flatten/transpose in
 *   CogVideoXPatchEmbed.forward / RefPatchEmbed.forward (models/crosstransformer3d.py:736, 78-87,
 *   120-135).
 * unpatchify: [B, F*(H/p)*(W/p), Cout*p*p] -> [B,F,Cout,H,W] (models/crosstransformer3d.py:863-867). */
int tcx_patchify(const void* a, const void* b, void* out, int32_t B, int32_t F, int32_t Ca, int32_t Cb,
                 int32_t H, int32_t W, int32_t p, int32_t k_stride, void* stream);
int tcx_unpatchify(const void* x, void* out, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W,
                   int32_t p, int32_t out_dtype, void* stream);

/* ---- K10: classifier-free guidance + DDIM step (v-prediction, eta = 0), fused -----------------
 * noise = u + g*(c - u) in fp32; x0 = bf16r(sa*x) - sb*noise; eps = sa*noise + bf16r(sb*x);
 * x_prev = bf16r(sqrt(a_prev)*x0 + sqrt(1-a_prev)*eps).  u/c: the two halves of the transformer
 * output (dtype pred_dtype, TCX_BF16 or TCX_F32), x: bf16 latents, out: bf16 (may alias x).
 * If `c` is null no guidance is applied (noise = u).
 * Replaces: models/pipeline_trajectorycrafter.py:1117,1157-1167,1178 with diffusers
 *   DDIMScheduler.step. */
int tcx_cfg_ddim_step(const void* u, const void* c, const void* x, void* out, int64_t n,
                      float guidance, float alpha_t, float alpha_prev, int32_t pred_dtype, void* stream);

/* K10 with eta > 0 (stochastic DDIM, `DDIMScheduler.step(eta=...)`: the pipeline's `eta` argument, :1073,1166): as tcx_cfg_ddim_step with
 * the direction coefficient dir_coef = sqrt(1 - a_prev - std^2) in place of sqrt(1 - a_prev) and
 *   x_prev = bf16r( sqrt_alpha_prev x0 + dir_coef eps + std_dev variance_noise ),  std_dev = eta sqrt((1 - a_prev)/(1 - a_t) (1 - a_t/a_prev)),
 * variance_noise fp32 [n] = the library's randn_tensor draw.  All five coefficients are the scheduler's fp32 scalars. */
int tcx_cfg_ddim_eta_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance,
                          float sqrt_alpha_t, float sqrt_beta_t, float sqrt_alpha_prev, float dir_coef, float std_dev,
                          const float* variance_noise, int32_t pred_dtype, void* stream);

/* ---- K10b: the same fusion for the reference's "DDIM_Cog" sampler (diffusers CogVideoXDDIMScheduler.step, eta = 0,
 * v-prediction):  noise as above;  x0 = bf16r(sqrt_alpha_t * x) - sqrt_beta_t * noise;
 *   x_prev = bf16r( bf16r(coef_sample * x) + coef_x0 * x0 ),   coef_sample = sqrt((1 - a_prev) / (1 - a_t)),
 *   coef_x0 = sqrt(a_prev) - sqrt(a_t) * coef_sample — the four coefficients come from the scheduler's float64 tables (host).
 * Replaces: models/pipeline_trajectorycrafter.py:1117,1157-1167,1178 with the scheduler demo.py:652 selects. */
int tcx_cfg_ddim_cog_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance,
                          float sqrt_alpha_t, float sqrt_beta_t, float coef_sample, float coef_x0, int32_t pred_dtype,
                          void* stream);

/* ---- K10c: the same fusion for the sigma-parametrised samplers of the reference's table (demo.py:647-654: "Euler",
 * "Euler A", "DPM++" = diffusers EulerDiscreteScheduler / EulerAncestralDiscreteScheduler / DPMSolverMultistepScheduler,
 * v-prediction).  `coef` is a HOST array of 5 floats: the scheduler's own 0-dim fp32 scalars (scheduler.py computes them with the
 * library's operation sequence); the kernel applies the library's per-element fp32 operations in their order.
 *   noise-free guidance v = u + g (c - u) as above;  x = bf16 latents (upcast);  out = bf16r(prev), may alias x.
 *   TCX_STEP_EULER     coef = {a, sigma^2 + 1, sigma, dt, sigma_up}:  x0 = v a + x / coef[1];  d = (x - x0) / sigma;
 *                      prev = x + d dt;  with `noise` (fp32, n elements; Euler A): prev += noise sigma_up.  No history.
 *   TCX_STEP_DPMPP_2M  coef = {alpha_i, sig_i, A, B, 1/r0}:  x0 = bf16r(alpha_i x) - sig_i v, written to `hist_out` (fp32, n);
 *                      prev = A x - B x0, and with `hist_in` (the previous step's hist_out): prev -= (B/2) ((1/r0)(x0 - hist_in)).
 * Replaces: models/pipeline_trajectorycrafter.py:1117,1157-1167,1178 with the `step` of the scheduler demo.py:647-657 selects. */
#define TCX_STEP_EULER 0
#define TCX_STEP_DPMPP_2M 1
int tcx_cfg_sigma_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance, int32_t kind,
                       const float* coef, const float* hist_in, float* hist_out, const float* noise, int32_t pred_dtype, void* stream);

/* ---- K10d: the same fusion for "PNDM" (diffusers PNDMScheduler, v-prediction; 12 Runge-Kutta evaluations, then 4th-order linear
 * multistep).  mo = u + g (c - u); `coef` is a HOST array of 6 floats {w, sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev / a_t), a_prev - a_t,
 * a_t sqrt(1 - a_prev) + sqrt(a_t (1 - a_t) a_prev)} (the scheduler's 0-dim fp32 scalars); all history tensors are fp32, n elements.
 *   TCX_PNDM_PRK_FIRST  cur_out = (cur_in ? cur_in : 0) + w mo;  mo_out = mo;  eff = mo            (w = 1/6)
 *   TCX_PNDM_PRK_MID    cur_out = cur_in + w mo;                             eff = mo            (w = 1/3)
 *   TCX_PNDM_PRK_LAST   eff = cur_in + w mo                                                      (w = 1/6)
 *   TCX_PNDM_PLMS4      mo_out = mo;  eff = (1/24)(((55 mo - 59 e1) + 37 e2) - 9 e3)              (e1 newest)
 *   then eps = coef[1] eff + bf16r(coef[2] x);  out = bf16r( bf16r(coef[3] x) - (coef[4] eps) / coef[5] ),  x = the bf16 sample the
 *   library hands `_get_prev_sample` (the group's first sample during the Runge-Kutta evaluations).
 * Replaces: models/pipeline_trajectorycrafter.py:1117,1157-1167,1178 with PNDMScheduler.step (demo.py:651). */
#define TCX_PNDM_PRK_FIRST 0
#define TCX_PNDM_PRK_MID 1
#define TCX_PNDM_PRK_LAST 2
#define TCX_PNDM_PLMS4 3
int tcx_cfg_pndm_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance, int32_t mode,
                      const float* coef, const float* e1, const float* e2, const float* e3, const float* cur_in, float* cur_out,
                      float* mo_out, int32_t pred_dtype, void* stream);

/* y = bf16r(x / d), n bf16 elements: `scheduler.scale_model_input` of the Euler samplers (x / sqrt(sigma^2 + 1); a bf16 tensor
 * divided by a 0-dim fp32 tensor stays bf16).  Replaces: models/pipeline_trajectorycrafter.py:1099-1101 for those schedulers. */
int tcx_div_bf16(const void* x, void* y, int64_t n, float d, void* stream);

/* ---- K11 / K12 / 1x1x1 convs: bf16 implicit-GEMM convolution, channels-last -------------------
 * y[n, t, oy, ox, co] = bias[co] + sum_{dt,dy,dx,ci} X(t + dt, oy*stride + dy - pad_h, ox*stride + dx - pad_w, ci)
 *                                                    * w[co, dt, dy, dx, ci]   (+ res[n,t,oy,ox,co])
 * where X is the logical input: rows t < kT-1 come from `cache` (the previous chunk's last kT-1
 * input frames, or the first frame replicated when cache is null), the rest from `x`; spatial
 * zero padding; when `ups` = 1 the logical input is the nearest-neighbour x2 spatial upsample of
 * `x` (never materialised); `t_map` (int32 [T_out], device, or null = identity) maps an output
 * frame to the source frame of `x` (the temporal part of CogVideoXUpsample3D).
 * Layouts: x [N, T_in, H_in, W_in, Cin] bf16; w [Cout, kT, kH, kW, Cin] bf16 (pre-permuted by the
 * host from the reference's [Cout, Cin, kT, kH, kW]); y [N, T_out, H_out, W_out, Cout] bf16.
 * stride in {1, 2} (spatial; 2 = the encoder's CogVideoXDownsample3D conv with its (0,1,0,1) padding:
 * pad_h = pad_w = 0 and the extra bottom / right zero row comes from the range check).
 * Replaces: CogVideoXCausalConv3d.forward + CogVideoXSafeConv3d (models/autoencoder_magvit.py:
 *   41-73,136-163), the interpolate + Conv2d of diffusers CogVideoXUpsample3D (built :620-630),
 *   the 1x1x1 conv_shortcut (:312-318,351-352) and the residual add (:354).
 * The cache for the next chunk (last kT-1 logical input frames, :157) is a plain slice of
 * concat(cache, x) that the host keeps; `res` (or null) is added in fp32 before the rounding. */
int tcx_conv3d_cl(const void* x, const void* cache, const void* w, const void* bias, const void* res,
                  void* y,
                  int32_t N, int32_t T_in, int32_t H_in, int32_t W_in, int32_t Cin, int32_t Cout,
                  int32_t kT, int32_t kH, int32_t kW, int32_t T_out, int32_t ups, int32_t stride,
                  int32_t pad_h, int32_t pad_w, int32_t H_out, int32_t W_out, const int32_t* t_map,
                  void* stream);

/* Which kernel tcx_conv3d_cl launches for a shape (host-only query, no GPU touched; the same decision function the launch
 * uses).  Lets a caller / test assert that a model configuration reaches the kernels it was tuned for.
 *   TCX_CONV_ROUTE_MFMA_WIDE  conv_mfma_kernel<2,4>: 256 x 256 tile, LDS-DMA gather (Cin % 64 == 0, Cout >= 256)
 *   TCX_CONV_ROUTE_MFMA_TALL  conv_mfma_kernel<4,2>: 512 x 128 tile              (Cin % 64 == 0, 128 <= Cout < 256)
 *   TCX_CONV_ROUTE_NARROW     conv_narrow_kernel: Cout <= 4 dot-product kernel   (the decoder's conv_out)
 *   TCX_CONV_ROUTE_IGEMM      conv_igemm_kernel: register-staged 128 x 128 tile  (everything else)
 * Replaces: nothing in the reference (cuDNN picks its algorithm internally behind nn.Conv3d, models/autoencoder_magvit.py:41-73). */
#define TCX_CONV_ROUTE_MFMA_WIDE 1
#define TCX_CONV_ROUTE_MFMA_TALL 2
#define TCX_CONV_ROUTE_NARROW 3
#define TCX_CONV_ROUTE_IGEMM 4
int tcx_conv3d_route(int32_t T_in, int32_t H_in, int32_t W_in, int32_t Cin, int32_t Cout, int32_t kT, int32_t kH, int32_t kW,
                     int32_t ups, int32_t stride, int32_t H_out, int32_t W_out, int32_t has_t_map, int32_t has_res);

/* ---- temporal average pool of diffusers CogVideoXDownsample3D(compress_time) -------------------
 * x [N, T, S, C] channels-last bf16 (S = H*W) -> y [N, T', S, C]: T even: pairs averaged (T' = T/2);
 * T odd: frame 0 kept, frames 1.. averaged pairwise (T' = 1 + (T-1)/2).  fp32 mean, one rounding.
 * Replaces: the avg_pool1d branch the encoder's down blocks run (built models/autoencoder_magvit.py:423-435). */
int tcx_avgpool_t(const void* x, void* y, int32_t N, int32_t T, int64_t S, int32_t C, void* stream);

/* ---- seam blend of the tiled VAE decode ---------------------------------------------------------
 * b[o, y, i] = a[o, y, i] * (1 - y/ext) + b[o, y, i] * (y/ext) for o < outer, y < ext, i < inner, IN PLACE on b (bf16); `a` points at
 * the first of the LAST `ext` rows (blend_v) or columns (blend_h) of the neighbouring tile.  Element strides; rounding as the
 * reference's eager bf16 arithmetic: each product to bf16, then the sum (weights are python floats, i.e. fp32 scalars in the op).
 * Channels-last tiles [N, T, H, W, C]: blend_v = (outer N*T, ext along H, inner W*C), blend_h = (outer N*T*H, ext along W, inner C).
 * Replaces: AutoencoderKLCogVideoX.blend_v / blend_h (models/autoencoder_magvit.py:1282-1301) as tiled_decode calls them (:1376-1379). */
int tcx_blend_ramp_bf16(const void* a, void* b, int64_t outer, int32_t ext, int64_t inner, int64_t a_outer_stride, int64_t a_ext_stride,
                        int64_t b_outer_stride, int64_t b_ext_stride, void* stream);

/* ---- K13: GroupNorm statistics + fused GroupNorm * SpatialNorm modulate + SiLU ----------------
 * stats: per (n, group) mean / rstd over (T, H, W, C/G) of channels-last x [N, spatial, C] bf16.
 *   Two launches: per-channel shifted partial sums of `nsplit` row slabs into
 *   partial (fp32 [N, nsplit, 2, C]), then an fp64 combine -> stats fp32 [N, G, 2] = (mean, rstd).
 * apply: y = silu( GN(x) * Y[src] + Bt[src] ) when ytab != null, else y = silu(GN(x)) (silu optional).
 *   Y = conv_y(zq), Bt = conv_b(zq) are the SpatialNorm 1x1x1 convs evaluated at zq's own LOW
 *   resolution (channels-last [N, Tz, Hz, Wz, C] bf16, produced with tcx_conv3d_cl): a nearest resize
 *   commutes with a pointwise conv, so the resized zq is never materialised.  z_t_map int32 [T]
 *   (device) gives the zq frame of every frame of x (the first-frame/rest split of
 *   autoencoder_magvit.py:200-208); the spatial nearest index is floor(i * Hz / H).
 * Replaces: CogVideoXSpatialNorm3D.forward (models/autoencoder_magvit.py:199-212) + the SiLU at
 *   :333,347,951 and nn.GroupNorm + SiLU on the encoder side (:265-270,331-333,345-347,795-796). */
int tcx_groupnorm_stats(const void* x, float* stats, float* partial, int32_t N, int64_t spatial, int32_t C,
                        int32_t G, float eps, int32_t nsplit, void* stream);
int tcx_groupnorm_spatialnorm_silu(const void* x, void* y, const float* stats,
                                   const void* gn_w, const void* gn_b, const void* ytab, const void* btab,
                                   int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t G,
                                   int32_t Tz, int32_t Hz, int32_t Wz, const int32_t* z_t_map,
                                   int32_t apply_silu, void* stream);

/* ---- K14: layout + final clamp ---------------------------------------------------------------------
 * ncthw <-> channels-last transposes for the VAE boundary and frames = clamp(x/2 + .5, 0, 1) as fp32
 * [N, C, T, H, W] (models/pipeline_trajectorycrafter.py:515-517). */
int tcx_ncthw_to_cl(const void* x, void* y, int32_t N, int32_t C, int64_t spatial, float mul, void* stream);
int tcx_cl_to_ncthw_frames(const void* x, float* y, int32_t N, int32_t C, int64_t spatial, int64_t out_spatial_stride,
                           int64_t out_offset, void* stream);

/* ---- f0: bf16 GEMM with fused epilogues — the Linear layers of the transformer -----------------
 * y[M,N] = epilogue(x[M,K] . w[N,K]^T + bias[N]); bf16 operands, fp32 accumulation, bf16 result; w is a torch
 * nn.Linear weight as stored ([out, in], K contiguous); bias bf16 [N] or null.
 *   TCX_GEMM_BIAS           y = acc + bias                                  (to_q/k/v, to_out, proj, ff.net.2 ...)
 *   TCX_GEMM_BIAS_GELU      y = gelu_tanh(acc + bias)                       (FeedForward net.0, crosstransformer3d.py:215-222)
 *   TCX_GEMM_GATED_RESIDUAL y = res + gate[b(row)] * (acc + bias)           (:245-248, 261-264; res may alias y)
 *       gate_t for rows with (row % rows_per_batch) < text_len, gate_v otherwise, batch b = row / rows_per_batch,
 *       gate[b] = gate_x + b * gate_stride_b; both gates null -> gate = 1 (the plain residual of :833-837).
 * x rows are uniformly strided (ldx).  With rows_per_batch > 0, y and res are [B, rows_per_batch, N] views: row m =
 * (b, r) lives at y + b * y_stride_b + r * ldy (a row range of the joint text+video buffer); with 0 they are flat.
 * Needs N % 8 == 0, K % 8 == 0, ldx % 8 == 0, ldy % 8 == 0 (16-byte row-wise stores); any M < 2^31.  K % 128 == 0 (every Linear of
 * the 5B model) runs the plain loop, any other K the zero-filled K-tail instantiation.  256 x 256 output tiles (ragged edges are
 * masked), 128 KiB of LDS per workgroup.  M <= 8 with the bias epilogue (the AdaLN / time-embedding Linears, M = batch) takes a
 * weight-streaming dot-product kernel instead of an MFMA tile. */
#define TCX_GEMM_BIAS 0
#define TCX_GEMM_BIAS_GELU 1
#define TCX_GEMM_GATED_RESIDUAL 2
int tcx_gemm_bf16(const void* x, const void* w, const void* bias, void* y, int64_t M, int32_t N, int32_t K,
                  int64_t ldx, int64_t ldy, int64_t y_stride_b, int32_t epilogue, const void* res, int64_t ldres,
                  int64_t res_stride_b, const void* gate_v, const void* gate_t, int64_t gate_stride_b,
                  int32_t rows_per_batch, int32_t text_len, void* stream);

/* ---- f3: point-cloud render = forward warp by bilinear splatting, fp32 ---------------------------
 * One call = Warper.forward_warp(frame1, mask1, depth1, T1, T2, K1, K2, mask=False, twice=False) of the reference
 * (models/utils.py:220-293): compute_transformed_points (:350-421) then bilinear_splatting of the frame and of the
 * transformed depth (:422-583), fused: project -> float-atomic splat of (r,g,b,depth,weight) -> resolve.
 * frame [b,3,h,w], depth [b,1,h,w], mask1 [b,1,h,w] or null, all fp32 NCHW like the reference.
 * mats: fp32 [b,30] = K1^-1 (9) | rows of [R|t] of T2 T1^-1 (12) | K2 (9), prepared by the host (tiny inverses).
 * Outputs: flow [b,2,h,w], warped [b,3,h,w] in [-1,1] (-1 where empty), mask2 [b,1,h,w], wdepth [b,1,h,w].
 * Workspace: tdepth [b,h,w] and acc fp32 [b*(h+2)*(w+2)*5 + b] (zeroed by the call on the stream).
 * flags: TCX_WARP_PER_ITEM_MAX normalises the depth weight by each item's own max(log(1+depth)) instead of the max
 * over the batch (:478-479) — one call then equals b reference calls with batch 1, which is how the reference renders
 * a clip (demo.py:100-116 loops over frames).
 * Float atomics: matches the reference / oracle to ~1e-5, not bitwise. */
#define TCX_WARP_PER_ITEM_MAX 1
/* TCX_WARP_CLEAN_POINTS = forward_warp(mask=True): clean_points (:585-626) — the hole mask dilated by a 5x5 box
 * (cv2.dilate, borders ignored); the frame is blanked to -1 there, mask2 = 1 - dilated holes; depth is not cleaned. */
#define TCX_WARP_CLEAN_POINTS 2
int tcx_warp_forward(const float* frame, const float* mask1, const float* depth, const float* mats,
                     float* flow, float* tdepth, float* acc, float* warped, float* mask2, float* wdepth,
                     int32_t b, int32_t h, int32_t w, int32_t flags, void* stream);

/* One `Warper.bilinear_splatting(frame1, mask1, depth1, flow12, None, is_image)` of the reference (models/utils.py:422-583) on its
 * own: src [b,c,h,w] fp32 with 1 <= c <= 4, mask1 [b,1,h,w] or null, depth [b,h,w] (the weights exp(50 log(1+d) / max log(1+d)), max
 * over the whole batch like the reference), flow [b,2,h,w] multiplied by `flow_scale` (-1 for the reverse splat) -> out [b,c,h,w]
 * (is_image: holes = -1 and the result clamped to [-1, 1]; else holes = 0) and mask2 [b,1,h,w].  acc: scratch of
 * b (h+2)(w+2) 5 + 1 floats.  The building block of forward_warp(twice=True) (:294-347), whose first stage is tcx_warp_forward. */
int tcx_bilinear_splat(const float* src, const float* mask1, const float* depth, const float* flow, float* acc, float* out,
                       float* mask2, int32_t b, int32_t c, int32_t h, int32_t w, int32_t is_image, float flow_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TCX_HIP_H */
