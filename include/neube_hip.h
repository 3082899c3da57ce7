/*
 * neube_hip.h -- C ABI of the MI355X (gfx950) NeuBE generator kernels.
 *
 * This is the drop-in boundary for the reference's native plugin layer.  In the reference the
 * generator's arithmetic bottoms out in two pybind11/CUDA plugins plus cuDNN:
 *
 *   bias_act_plugin.bias_act(x,b,xref,yref,dy,grad,dim,act,alpha,gain,clamp)
 *       thirdparty/stylegan2_ada_pytorch/torch_utils/ops/bias_act.cpp:32-91  (kernel bias_act.cu:23-147)
 *   upfirdn2d_plugin.upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,pady0,pady1,flip,gain)
 *       thirdparty/stylegan2_ada_pytorch/torch_utils/ops/upfirdn2d.cpp:16-94 (kernels upfirdn2d.cu:29-200)
 *   torch.nn.functional.conv2d / conv_transpose2d (cuDNN) via
 *       thirdparty/stylegan2_ada_pytorch/torch_utils/ops/conv2d_gradfix.py:35-43
 *   loaded lazily by torch_utils/custom_ops.py:46-124 (JIT nvcc build).
 *
 * Here they are replaced by one ahead-of-time hipcc-built shared library with plain-C entry
 * points: raw device pointers, integer sizes, scalar parameters and the HIP stream to launch on.
 * No torch types cross this boundary.  All tensors are dense, contiguous, float32, NCHW.
 * Outputs are caller-allocated.  Every function returns 0 on success and a negative NB_E* code on
 * failure; nb_last_error() then returns a thread-local message.  Functions only enqueue work on
 * `stream` (no allocation, no synchronisation) and are therefore capturable into a hipGraph.
 *
 * `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 */
#ifndef NEUBE_HIP_H
#define NEUBE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_OK            0
#define NB_EINVAL       -1   /* bad argument (shape, null pointer, unsupported mode) */
#define NB_ELAUNCH      -2   /* the HIP runtime rejected a launch */
#define NB_EUNSUPPORTED -3

/* activation codes = the reference's cuda_idx (bias_act.py:22-32) */
#define NB_ACT_LINEAR  1
#define NB_ACT_RELU    2
#define NB_ACT_LRELU   3
#define NB_ACT_TANH    4
#define NB_ACT_SIGMOID 5
#define NB_ACT_ELU      6
#define NB_ACT_SELU     7
#define NB_ACT_SOFTPLUS 8
#define NB_ACT_SWISH    9

const char* nb_last_error(void);
int nb_abi_version(void);

/* ---- standalone operator parity with the reference plugins ------------------------------- */

/* y = clamp(act(x + b[(i / step_b) % size_b]) * gain, +-clamp); forward (grad=0) of
 * bias_act.cpp:32-91.  size_b = 0 -> no bias.  clamp < 0 -> no clamping.  x and y may alias. */
int nb_bias_act_f32(const float* x, const float* b, float* y, int64_t size_x, int size_b, int step_b,
                    int act, float alpha, float gain, float clamp, void* stream);

/* The plugin's full entry (bias_act.cpp:32 `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`,
 * kernel bias_act.cu:23-147) with the gradient modes the reference's autograd functions call
 * (bias_act.py:155-204):
 *   grad 0: forward as above (xref, yref, dy ignored).
 *   grad 1: x = incoming gradient, returns x * act'(.) * gain, zero where the forward output was clamped.
 *   grad 2: x = gradient w.r.t. the grad-1 result, dy = the original incoming gradient; returns
 *           x * dy * act''(.) * gain (identically zero for linear / relu / lrelu).
 * yref = the forward output (needed by every activation but linear-without-clamp and swish),
 * xref = the forward input, bias not yet added (needed by swish only); unused ones may be NULL. */
int nb_bias_act_grad_f32(const float* x, const float* b, const float* xref, const float* yref, const float* dy,
                         float* y, int64_t size_x, int size_b, int step_b, int grad, int act, float alpha,
                         float gain, float clamp, void* stream);

/* upfirdn2d.cpp:16-94 forward on a contiguous [major, in_h, in_w] stack of planes (major = N*C):
 * zero-insert by (upx,upy), pad/crop, correlate with f[f_h,f_w] (flipped unless `flip`), scale by
 * gain, keep every (downx,downy)-th sample.  y is [major, out_h, out_w] with
 * out_w = (in_w*upx + padx0 + padx1 - f_w + downx) / downx (same for h). */
int nb_upfirdn2d_f32(const float* x, const float* f, float* y, int major, int in_h, int in_w,
                     int f_h, int f_w, int upx, int upy, int downx, int downy,
                     int padx0, int padx1, int pady0, int pady1, int flip, float gain, void* stream);

/* ---- gradient building blocks (row f4; csrc/nb_grad.hip) ------------------------------------------ */

/* Generic fp32 convolution (cross-correlation, zero padding), what conv2d_gradfix.py:107-168 takes from cuDNN for the
 * gradient w.r.t. the input of a strided conv:
 *   y[n,co,oy,ox] = out_scale[n,co] * sum_{ci,a,b} (x[n,ci,oy*stride+a-pad,ox*stride+b-pad] * in_scale[n,ci]) * w[co,ci,a,b]
 * x [n,c_in,h,wd], w [c_out,c_in,kh,kw] (kh, kw <= 7), y [n,c_out,ho,wo], ho = (h + 2 pad - kh)/stride + 1; the per-sample
 * scales may be NULL. */
int nb_conv2d_f32(const float* x, const float* w, const float* in_scale, const float* out_scale, float* y, int n,
                  int c_in, int h, int wd, int c_out, int kh, int kw, int stride, int pad, void* stream);

/* Weight-gradient correlation of a 3x3 conv: a[n,cu,cv,ka,kb] = sum_{i,j} u[n,cu,i*stride+ka-pad,j*stride+kb-pad] * v[n,cv,i,j]
 * (zero outside u); a [n,cu,cv,3,3] is overwritten (partial sums are added atomically: the last bits depend on the
 * order). */
int nb_conv2d_wgrad_f32(const float* u, const float* v, float* a, int n, int cu, int hu, int wu, int cv, int hv, int wv,
                        int stride, int pad, void* stream);

/* The same on the f16 matrix cores with split (hi/lo) operands: scales = 2 floats in device memory, powers of two that bring
 * max|u| and max|v| near 2^10 (gradients may sit far below the f16 range); the result is divided by their product. */
int nb_conv2d_wgrad_h3(const float* u, const float* v, const float* scales, float* a, int n, int cu, int hu, int wu, int cv,
                       int hv, int wv, int stride, int pad, void* stream);

/* The same without atomics (the form the training path uses): every workgroup stores its partial block, one more launch
 * adds the blocks in a fixed order -- the result is reproducible and a is not zero-filled first.  sum_n != 0 also sums over
 * the samples: a is [cu,cv,3,3] (what a plain convolution's weight gradient needs; conv2d_gradfix.py:129-137).
 * ws: device scratch of at least nb_conv2d_wgrad_h3_ws_bytes(...) bytes, 16-byte aligned (may be NULL when that is 0). */
long long nb_conv2d_wgrad_h3_ws_bytes(int n, int cu, int cv, int hv, int sum_n);
int nb_conv2d_wgrad_h3_ws(const float* u, const float* v, const float* scales, int scales_are_absmax, float* a, float* ws,
                          long long ws_bytes, int sum_n, int n, int cu, int hu, int wu, int cv, int hv, int wv, int stride, int pad,
                          void* stream);
/* scales_are_absmax != 0: `scales` holds max|u|, max|v| (the two slots nb_absmax_f32 fills) and the kernel derives the powers
 * of two itself. */

/* The tail of modulated_conv2d's backward pass (what autograd spells out as einsums and element-wise passes in the reference,
 * training/networks.py:30-88 differentiated): dd[n,o] = sum_pix dy (y - noise) (noise: NULL, one shared plane with stride 0, or one
 * plane per sample); and, with A[n,o,c,3,3] the per-sample weight-gradient correlation (element strides given: the wgrad kernels
 * leave it as [n][c][o][9] for up = 1 and [n][o][c][9] for up = 2), s the styles, dq = d(loss)/d(sum under the demodulation rsqrt)
 * or NULL:  dW[o,c,t] = sum_n s A + 2 W sum_n dq s^2,   ds[n,c] = sum_{o,t} W A + 2 s sum_o dq sum_t W^2.  dW or ds may be NULL. */
int nb_modconv_bwd_dot_f32(const float* dy, const float* y, const float* noise, long long noise_stride_n, float* out, int n, int o,
                           int hw, void* stream);
int nb_modconv_bwd_finish_f32(const float* A, long long a_stride_n, long long a_stride_o, long long a_stride_c, const float* s,
                              const float* W, const float* dq, float* dW, float* ds, int n, int o, int c, void* stream);

/* Range scaling of the split-f16 training operators (no counterpart in the reference, whose cuDNN path computes in fp32 /
 * fp16 directly): slots = two 4-byte device words, zero-filled by the caller; afterwards slots[0] = max(|a|, |b|) and
 * slots[1] = max|c| as float bit patterns (a, b, c may be NULL with a zero count). */
int nb_absmax_f32(const float* a, long long na, const float* b, long long nb, const float* c, long long nc, void* slots, void* stream);
/* nb_pack_h2_f32 with the scale derived on the device: out = H2((x1 ++ x2) * scale[n,c] * k), k = the power of two that
 * brings slots[0] * slots[1] near `target`; also dco_out[i] = dco_in[i] / k for i < dco_count (may be 0). */
int nb_pack_h2_ranged_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out_h2, int n, int hw,
                          const void* slots, float target, const float* dco_in, float* dco_out, int dco_count, void* stream);

/* ---- generator path ---------------------------------------------------------------------- */

/* MappingNetwork.forward (training/networks.py:255-290) for c_dim = 0, without the broadcast:
 * w[n,:] = FC_{L-1}(...FC_0(z * rsqrt(mean(z^2)+1e-8))), each FC = lrelu(x W^T * lr_mul/sqrt(in) + b*lr_mul)*sqrt2.
 * fc_w: layer 0 [w_dim, z_dim] followed by layers 1.. [w_dim, w_dim]; fc_b: [L, w_dim].
 * z_dim, w_dim <= 512. */
int nb_mapping_f32(const float* z, const float* fc_w, const float* fc_b, float* w_out, int n, int z_dim,
                   int w_dim, int num_layers, float lr_mul, void* stream);

/* The same with the broadcast of networks.py:278-280 written by the kernel: ws_out [n, num_ws, w_dim]. */
int nb_mapping_ws_f32(const float* z, const float* fc_w, const float* fc_b, float* ws_out, int n, int z_dim,
                      int w_dim, int num_layers, float lr_mul, int num_ws, void* stream);

/* One entry of the per-layer table consumed by nb_styles_f32 / nb_noise_f32 (device memory, built
 * once when weights are packed).  Pointers are device addresses. */
typedef struct NbLayerDesc {
    const float* affine_w;   /* [c_aff, w_dim]  FullyConnectedLayer.weight of the layer's affine */
    const float* affine_b;   /* [c_aff] */
    const float* wsq;        /* [c_in, c_out]   sum_k W[o,i,k]^2 (transposed), NULL -> no demodulation */
    float*       styles;     /* out [n_max, c_aff]  affine(w) * style_scale (entries < n_plain are NOT scaled) */
    float*       dcoefs;     /* out [n_max, c_out]  rsqrt(sum_i (styles^2 * wsq) + 1e-8), unused when wsq NULL */
    const float* noise_const;/* [res, res] or NULL */
    const float* noise_lin;  /* [res] = noise_grid[0, :, 0, 0] (torch.linspace(0,1,res)) */
    float*       noise_out;  /* out [n_max or 1, res, res] = (shifted) noise_const * noise_strength */
    const float* noise_strength; /* [1] */
    int32_t c_aff;           /* affine outputs (= c_in, or c_in + 9 for the triad ToRGB) */
    int32_t n_plain;         /* leading affine outputs that are not styles (9 color scalars for ToRGB, else 0) */
    int32_t c_out;
    int32_t w_index;         /* which ws[:, w_index, :] feeds this layer */
    int32_t res;             /* output resolution of the layer (noise size) */
    float   style_scale;     /* 1 for conv layers, 1/sqrt(c_in) for ToRGB (networks.py:460) */
    int32_t pad_[2];
} NbLayerDesc;

/* For every layer l < n_layers and sample n: styles = affine_l(ws[n, w_index_l]) (weight_gain
 * 1/sqrt(w_dim), bias_gain 1: networks.py:106-118) and the demodulation coefficients
 * (networks.py:59-62 in the algebraically equal form d = rsqrt(sum_i s_i^2 * sum_k W_oik^2 + 1e-8)). */
int nb_styles_f32(const NbLayerDesc* layers_dev, int n_layers, const float* ws, int num_ws, int w_dim, int n,
                  void* stream);

/* Same results up to fp32 summation order, latency-oriented launch shape (the batch-1 step starts with it): needs
 * w_dim % 16 == 0, every layer's c_out % 4 == 0 and 16-byte aligned affine_w / wsq. */
int nb_styles_fast_f32(const NbLayerDesc* layers_dev, int n_layers, const float* ws, int num_ws, int w_dim, int n,
                       void* stream);

/* nb_styles_fast_f32 and nb_noise_f32 (per-sample shifted noise: exactly one of norm_pos / positions) in ONE launch -
 * the two are independent, and at batch 1 a launch costs more than either computes. */
int nb_styles_noise_f32(const NbLayerDesc* layers_dev, int n_layers, const float* ws, int num_ws, int w_dim,
                        const float* norm_pos, const int64_t* positions, int img_resolution, int n, void* stream);

/* Stand-alone demodulation coefficients for one layer (networks.py:59-62 in the form above):
 * dcoefs[n,o] = rsqrt(sum_i styles[n,i]^2 * wsq[i,o] + 1e-8); styles [n,c_in], wsq [c_in,c_out]. */
int nb_demod_coefs_f32(const float* styles, const float* wsq, float* dcoefs, int n, int c_in, int c_out,
                       void* stream);

/* Constant-noise inputs of all layers in one launch (networks.py:371-382).
 *   norm_pos == NULL && positions == NULL: noise_out[0] = noise_const * strength (shared by the batch).
 *   positions != NULL: [n,2] int64 (y,x) patch positions; the kernel forms the reference's
 *       (positions % img_resolution) / (img_resolution - 1) (networks_modified.py:351-353) itself, with
 *       python-style modulo and correctly rounded float32 division, so that it is bit-identical to the
 *       reference's CPU arithmetic (the wrapped coordinate is discontinuous: 1 ulp moves a noise row).
 *   norm_pos != NULL: [n,2] float32 already-normalised positions (the `norm_noise_positions` kwarg).
 * noise_out[n] is the bilinear, wrapped resampling of SURVEY note C.  Layers whose noise_const is
 * NULL are skipped.  max_res = largest `res` in the table.  The reference's noise_buffers override
 * (networks_modified.py:163-165) is expressed by passing a table with replaced noise_const pointers. */
int nb_noise_f32(const NbLayerDesc* layers_dev, int n_layers, int max_res, const float* norm_pos,
                 const int64_t* positions, int img_resolution, int n, void* stream);

/* The first step of that arithmetic on its own: positions [n,2] int64 -> norm_pos_out [n,2] float32 = ((p mod R) / (R - 1)), the values
 * nb_noise_f32 derives internally (networks.py:371-374; same function, same bits).  For callers that hand NbNoiseSrc to the split-f16
 * convolutions: those evaluate the normalisation at the top of every tile, and from `positions` that is four 64-bit modulo operations
 * per lane and tile -- convert once per batch and pass `norm_pos`. */
int nb_norm_positions_f32(const int64_t* positions, int img_resolution, float* norm_pos_out, int n, void* stream);

/* Modulated 3x3 convolution + fused epilogue = SynthesisLayer.forward (networks.py:362-391) after
 * the affine:  y = clamp(lrelu(conv(x * s) * d + noise + bias, alpha) * gain, +-clamp)
 *   up = 1: cross-correlation, zero padding 1 (conv2d_resample.py:145-147)
 *   up = 2: stride-2 transposed convolution (true convolution, 2H+1) followed by the 4x4
 *           [1,3,3,1]x[1,3,3,1]/64 FIR with padding 1 and gain 4 (conv2d_resample.py:124-142)
 * The input is the channel-concatenation of x1 [n,c1,h,w] and (optionally) x2 [n,c2,h,w] -- the
 * geometry feature -- so no torch.cat copy is needed (networks_modified.py:218-219).
 * wpk is the packed, zero-padded weight [ceil8(c1+c2)][9][ceil32(c_out)] (tap = ky*3+kx) produced by
 * nb_pack_conv_weight.  x1, x2, wpk and y must be 16-byte aligned.
 * noise is [n or 1, h*up, w*up] with sample stride noise_stride_n (0 = shared), or NULL.
 * y is [n, c_out, h*up, w*up].  h and w must be powers of two >= 4. */
int nb_modconv3x3_f32(const float* x1, int c1, const float* x2, int c2, const float* wpk, const float* styles,
                      const float* dcoefs, const float* noise, int64_t noise_stride_n, const float* bias,
                      float* y, int n, int h, int w, int c_out, int up, float alpha, float gain, float clamp,
                      void* stream);

/* Name of the kernel template instance nb_modconv3x3_f32 launches for this problem shape (the name
 * rocprofv3 reports), so that a benchmark can attribute its per-launch timings to kernels. */
int nb_modconv3x3_variant(int n, int h, int w, int c_out, int up, char* buf, int buflen);

/* ---- split-f16 ("h3") fast path for the large conv1 layers (csrc/nb_modconv_h3.hip) ----------------
 * Every fp32 operand is carried as hi + lo f16 halves and a product is three f16 MFMAs (xh*wh + xl*wh +
 * xh*wl, fp32 accumulate): fp32-grade results (5e-6 on pixels end to end) at ~5x the fp32-MFMA rate.
 * Activation format H2: _Float16 [n][ceil(c/8)][2 (hi,lo)][h][w][8], ALREADY multiplied by the consuming
 * layer's styles.  Weight format: produced by nb_pack_conv_weight_h3 (static, not modulated). */

/* fp32 NCHW (x1 [n,c1,hw] ++ x2 [n,c2,hw]) * scale[n, c1+c2] (or NULL) -> H2 */
int nb_pack_h2_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out_h2, int n,
                   int hw, void* stream);

/* Host helper: W[c_out,c_in,3,3] fp32 -> hi/lo f16 [ceil(c_in/16)][3][3][2][2][ceil64(c_out)][8]
 * (bytes: ceil(c_in/16)*9*4*ceil64(c_out)*16). */
int nb_pack_conv_weight_h3(const float* w, int c_out, int c_in, void* out);
/* The same packing on the DEVICE (w and out are device pointers): one launch instead of a host loop -- for weights that
 * change every step (the training path evaluates its 3x3 convolutions on the split-f16 kernels, ops.TRAIN_SPLIT_F16).
 * co_align = 64 (modconv kernels) or 128 (encoder-type kernels); transpose_flip != 0: w is [c_in][c_out][3][3] and the packed
 * weight is its transpose with reversed taps (the kernel of the input-gradient convolution). */
int nb_pack_conv_weight_h3_dev(const float* w, int c_out, int c_in, int co_align, int transpose_flip, void* out, void* stream);

/* SynthesisLayer.forward with up = 1 (networks.py:362-391) on an H2 input:
 * y[n,c_out,h,w] (fp32 NCHW) = clamp(lrelu(conv3x3(x_h2, W) * dcoefs + noise + bias, alpha) * gain).
 * Needs w % 32 == 0 and h % 16 == 0 (the large layers); smaller layers use nb_modconv3x3_f32. */
int nb_modconv3x3_up1_h3(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                         int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                         float alpha, float gain, float clamp, void* stream);

/* SynthesisLayer.forward with up = 2 (stride-2 transposed conv as 4 phases + fused 4x4 FIR, as in
 * nb_modconv3x3_f32) with split-f16 products: x_h2 is the H2 input [n, c_in, h, w] (already multiplied by this
 * layer's styles), y the fp32 NCHW output [n, c_out, 2h, 2w].  Needs w % 32 == 0. */
int nb_modconv3x3_up2_h3(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                         int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                         float alpha, float gain, float clamp, void* stream);

/* nb_modconv3x3_f32 with up = 2 whose fused epilogue writes the result in H2 format, multiplied by
 * out_scale[n, c_out] (the styles of the conv1 layer that consumes it; NULL = 1), instead of fp32 NCHW. */
int nb_modconv3x3_up2_f32_h2(const float* x1, int c1, const float* x2, int c2, const float* wpk,
                             const float* styles, const float* dcoefs, const float* noise,
                             int64_t noise_stride_n, const float* bias, const float* out_scale, void* y_h2,
                             int n, int h, int w, int c_out, float alpha, float gain, float clamp, void* stream);

/* ToRGBColorTriadLayer.forward (networks.py:451-485) after its affine, on x [n,c,hw]:
 *   logits_o = clamp(sum_c x_c * styles[n,c] * w[o,c] + bias[o]);  uvs = softmax_o(logits)
 *   img[n,ch] = sum_k uvs_k * colors[n,ch,k]            (colors = tanh(affine[:, :9] + color_bias))
 * plus, optionally, the paint engine's compositing (forger/ui/brush.py:763-792):
 *   rgba[:3] = sum_k uvs_k * col01[n,:,k], rgba[3] = u+v ('clear', render_mode 0) or 1 ('full', 1)
 *   with col01 = user_colors (where not NaN) else (colors+1)/2; if sfactor [n] is given, (u,v,s) are first remapped
 *   per StyleUVSMapper._map_style_s (forger/ui/mapper.py:52-72; brush.py:773-774): s' = min(sfactor*s, 1),
 *   (u',v') = (u,v) * (1-s')/(u+v) (0 where 1-s' <= 1e-6) -- only in the RGBA outputs; rgba_f32 is [n,4,hw];
 *   rgba_u8 = trunc(clip(rgba*255, 0, 255)) is [n,hw,4] bytes (HWC, as brush.py:377 hands it out).
 * colors_raw is the [n, c_aff] affine output whose first 9 entries are the un-biased color scalars.
 * Any of logits / uvs / img / colors_out / rgba_f32 / rgba_u8 may be NULL. */
int nb_torgb_triad_f32(const float* x, const float* styles, int styles_stride_n, const float* w, const float* bias,
                       const float* color_bias, float clamp, float* logits, float* uvs, float* img,
                       float* colors_out, const float* user_colors, const float* sfactor, int render_mode,
                       float* rgba_f32, uint8_t* rgba_u8, int n, int c, int hw, void* stream);

/* BlendedFeatures.blend (forger/train/stitching.py:24-25): y = alpha*F + (1-alpha)*x over
 * x [n,c,hw]; F is [nf,c,hw] and alpha [na,1,hw] with nf, na in {1, n} (broadcast). */
int nb_blend_f32(const float* features, int nf, const float* alpha, int na, const float* x, float* y, int n, int c,
                 int hw, void* stream);

/* ---- "f8" operand format of the split-f16 convolutions: the two correction products on one block-scaled fp8 MFMA per
 * tap pair (csrc/nb_modconv_h3.hip).  Same containers as H2 / nb_pack_conv_weight_h3, but the (cg, lo) slots of a
 * 16-channel chunk hold fp8 e4m3 values: activations (cg 2k, lo) = fp8(xl*2^9), (cg 2k+1, lo) = fp8(x/4); weights
 * (cg 0, lo) = fp8(w), (cg 1, lo) = fp8((w - f16(w))*2^11).  c_in % 16 == 0.  End-to-end pixel error ~1e-4 (H2: 5e-6). */
int nb_pack_h2f8_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out, int n, int hw,
                     void* stream);
/* c % 16 == 0 channels into the 16-channel chunks starting at channel group cg0 (even) of an f8-format tensor */
int nb_pack_h2f8_part_f32(const float* x, int c, const float* scale, int scale_stride, void* out, int c8_total, int cg0,
                          int n, int hw, void* stream);

/* ---- "f6" operand format (round 5): the two correction products on one block-scaled fp6 (e2m3) MFMA per tap pair, at the f16
 * instruction's cycles.  Same containers; the two lo slots of a 16-channel chunk (32 bytes per pixel / per (c_out, tap)) hold 32
 * six-bit fields (24 bytes), the chunk's E8M0 scale byte and zeros: activations field 2i = e2m3(xl[ch(i)]*2^11/S), field 2i+1 =
 * e2m3(x[ch(i)]/S), S = 2^(exponent of the chunk's largest |x| - 2), byte = S; weights field 2i = e2m3(w[ch(i)]/Sw), field 2i+1 =
 * e2m3((w - f16(w))[ch(i)]*2^11/Sw), byte = Sw*2^-11; ch(i) = channels 0-3, 8-11, 4-7, 12-15 of the chunk.  c_in % 16 == 0.
 * Operand format number 2 of nb_modconv3x3_up1_h3_ex / nb_modconv3x3_up2_h3_ex (0 = H2, 1 = f8).  Replaces nothing in the reference
 * (its convolutions are cuDNN calls, torch_utils/ops/conv2d_gradfix.py:35-43); the math it serves is training/networks.py:30-88. */
int nb_pack_h2f6_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out, int n, int hw,
                     void* stream);
/* c % 16 == 0 channels into the 16-channel chunks starting at channel group cg0 (even) of an f6-format tensor */
int nb_pack_h2f6_part_f32(const float* x, int c, const float* scale, int scale_stride, void* out, int c8_total, int cg0,
                          int n, int hw, void* stream);

/* Arguments of the triad ToRGB epilogue when it is fused into the last conv (same meaning as the parameters of
 * nb_torgb_triad_f32; any output pointer may be NULL). */
struct NbTorgbArgs {
    const float* styles;      /* [n, styles_stride_n]: 9 color scalars then c styles (affine output) */
    const float* w;           /* [3, c] */
    const float* bias;        /* [3] */
    const float* color_bias;  /* [9] */
    float* logits; float* uvs; float* img; float* colors_out;
    const float* user_colors; /* [n,3,3] or NULL */
    const float* sfactor;     /* [n] or NULL */
    float* rgba_f32; uint8_t* rgba_u8;
    int styles_stride_n, render_mode;
    float clamp;
};

/* Last SynthesisLayer (up = 1) + ToRGBColorTriadLayer + compositing in one launch: nb_modconv3x3_up1_h3 followed by
 * nb_torgb_triad_f32 on the tile while it is still in LDS (c_out <= 128).  y (the fp32 activations) may be NULL:
 * nothing but ToRGB reads them unless a caller taps the features. */
int nb_modconv3x3_up1_h3_torgb(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                               int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                               float alpha, float gain, float clamp, const struct NbTorgbArgs* t, void* stream);

/* Constant noise computed INSIDE the split-f16 convolutions instead of being read from a [n, H, W] tensor that
 * nb_noise_f32 wrote (networks.py:371-382; the position-shifted bilinear sample of a <= 256 x 256 constant is four
 * L2-resident loads and a dozen FMAs per output pixel -- cheaper than writing and re-reading 22 MB per batch of 32).
 * Pass noise_stride_n = NB_NOISE_IN_KERNEL and, as `noise`, a HOST pointer to this struct (copied at launch) to
 * nb_modconv3x3_up1_h3_ex / nb_modconv3x3_up2_h3_ex.  Same expressions, same order as nb_noise_f32: bit-identical. */
#define NB_NOISE_IN_KERNEL (-1)
struct NbNoiseSrc {
    const float* noise_const_t;   /* [res, res]: the layer's noise_const TRANSPOSED (the sampling grid transposes: SURVEY note C) */
    const float* noise_lin;       /* [res]: noise_grid[0, :, 0, 0] */
    const float* noise_strength;  /* [1] */
    const float* norm_pos;        /* [n, 2] normalised positions, or NULL */
    const int64_t* positions;     /* [n, 2] integer (y, x) positions (normalised in-kernel like nb_noise_f32), or NULL; exactly one */
    int res;                      /* resolution of the layer's OUTPUT */
    int img_resolution;           /* R of (positions % R) / (R - 1) */
};

/* General forms of the two split-f16 convolutions.  in_fmt / out_fmt: 0 = H2 (hi/lo f16), 1 = "f8" (hi f16 + fp8
 * correction operands, above); the weights must be packed for in_fmt (nb_pack_conv_weight_h3 or the f8 layout).
 * in_fmt 2 = "f6" (round-5 experiment: fp6 corrections, own weight layout).  in_fmt 3 (round 6) = f8 operands and weights with the
 * correction products SKIPPED where the launch runs on the large throughput kernels (ping-pong up=1 loop, 12-row up=2 kernel;
 * elsewhere it is in_fmt 1): a plain single-f16 evaluation, 1.5e-3 ... 3e-3 from fp32 end to end -- the reference's own shipped arithmetic for
 * blocks >= 32^2 (training/networks.py:634-638), outside this build's 1e-3 parity budget; Generator(conv_mode="f16"), a timing
 * data point, not a parity mode.
 * Exactly one destination: y_f32 (fp32 NCHW; out_fmt ignored), y_h2 (the consumer's input tensor [n, c_next, ...] in
 * out_fmt, multiplied by next_styles), or -- up1 only, t != NULL -- the fused ToRGB outputs (y_f32 optional). */
int nb_modconv3x3_up1_h3_ex(const void* x, int c_in, const void* wts, const float* dcoefs, const float* noise,
                            int64_t noise_stride_n, const float* bias, float* y_f32, void* y_h2, const float* next_styles,
                            int next_stride, int c_next, const struct NbTorgbArgs* t, int in_fmt, int out_fmt, int n, int h,
                            int w, int c_out, float alpha, float gain, float clamp, void* stream);
int nb_modconv3x3_up2_h3_ex(const void* x, int c_in, const void* wts, const float* dcoefs, const float* noise,
                            int64_t noise_stride_n, const float* bias, float* y_f32, void* y_h2, const float* next_styles,
                            int next_stride, int c_next, int in_fmt, int out_fmt, int n, int h, int w, int c_out,
                            float alpha, float gain, float clamp, void* stream);

/* Name of the kernel nb_modconv3x3_up2_h3 / _ex launches for this problem shape (the name rocprofv3 reports: the round-3 8-wave
 * kernel in one of its tile forms, or the software-pipelined 8-wave kernel of csrc/nb_modconv_up2v.hip), so that a benchmark can
 * attribute its per-launch timings to kernels.  in_fmt as in _ex. */
int nb_modconv3x3_up2_h3_variant(int in_fmt, int c_in, int c_out, int n, int h, int w, char* buf, int buflen);

/* The same layer for SMALL images (csrc/nb_modconv_small.hip; the <= 64x64 conv1 layers): split-f16 products on
 * 32 c_out x 32 position tiles with K split over the 4 waves of a workgroup, fp32 NCHW in and out, the per-sample styles
 * applied to the activations while they are split into hi/lo f16 on their way into LDS.  Replaces nb_modconv3x3_f32
 * (up = 1) where the fp32 matrix rate or the launch latency of few, long workgroups dominates: h, w powers of two >= 4
 * (4x4 images are processed two samples per tile), c_in <= 512.  w_h3 = the nb_pack_conv_weight_h3 format. */
int nb_modconv3x3_up1_small_h3(const float* x, int c_in, const void* w_h3, const float* styles, const float* dcoefs,
                               const float* noise, int64_t noise_stride_n, const float* bias, float* y, int n, int h,
                               int w, int c_out, float alpha, float gain, float clamp, void* stream);

/* up = 2 (conv2d_resample.py:124-142: stride-2 transposed conv + 4x4 FIR, pad 1, gain 4) for small images through the same
 * kernel: per output phase (py, px) the composite is a 3x3 correlation of the input grid with an effective kernel
 * Keff[py,px] = W folded with the FIR (static, no styles), so w_h3_phases = the four nb_pack_conv_weight_h3(Keff[2 py + px])
 * images back to back and the kernel runs the phases as grid.z.  (x1 ++ x2) = concatenated input, c1, c2 % 16 == 0;
 * h, w = INPUT size; y [n, c_out, 2h, 2w]; noise [.., 2h, 2w]. */
int nb_modconv3x3_up2_small_h3(const float* x1, int c1, const float* x2, int c2, const void* w_h3_phases, const float* styles,
                               const float* dcoefs, const float* noise, int64_t noise_stride_n, const float* bias, float* y,
                               int n, int h, int w, int c_out, float alpha, float gain, float clamp, void* stream);

/* The two split-f16 convolutions with the output written straight into the CONSUMER's H2 input tensor
 * y_h2 = H2 [n, c_next, h_out, w_out] (channel groups 0 .. c_out/8-1; c_out % 8 == 0), already multiplied by the
 * consumer's styles next_styles[n*next_stride + c] -- the fused form of SynthesisLayer.forward followed by the next
 * layer's `x * styles` (networks.py:67, 362-391).  Removes the separate nb_pack_h2_f32 pass between two such layers. */
int nb_modconv3x3_up1_h3_h2(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                            int64_t noise_stride_n, const float* bias, const float* next_styles, int next_stride,
                            void* y_h2, int c_next, int n, int h, int w, int c_out, float alpha, float gain, float clamp,
                            void* stream);
int nb_modconv3x3_up2_h3_h2(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                            int64_t noise_stride_n, const float* bias, const float* next_styles, int next_stride,
                            void* y_h2, int c_next, int n, int h, int w, int c_out, float alpha, float gain, float clamp,
                            void* stream);

/* fp32 NCHW x [n,c,hw] * scale[n*scale_stride + ch] (or NULL) -> channel groups cg0.. of an H2 tensor with c8_total
 * groups: the geometry features concatenated behind a producer that wrote its groups itself (NM:218-219). */
int nb_pack_h2_part_f32(const float* x, int c, const float* scale, int scale_stride, void* out_h2, int c8_total, int cg0,
                        int n, int hw, void* stream);

/* ---- canvas side of the painting engine (SURVEY 8 rows e / f2) ---------------------------------------------
 * Cells: the canvas is cut into NB_CELL_H x NB_CELL_W pixel cells (row-major, ceil(w/NB_CELL_W) per row);
 * cell_off [ncells+1] / cell_tiles is a CSR list of the tiles whose rectangle touches each cell, in ascending
 * tile order (= the reference's sequential paint order).  All pointers are device pointers. */
#define NB_CELL_H 4
#define NB_CELL_W 64

/* Geometry tiles for the encoder: out[t,0,y,x] = 1 - (255 - geom[ty+y, tx+x]) / 255  (outside the image: 1), i.e.
 * forger/viz/paint_image_main.py:162 (`255 - geom[y:y+P, x:x+P]`) followed by GanPaintEngine.prepare_geom_input
 * (forger/ui/brush.py:672-681).  geom is [gh,gw] uint8 (255 = background), tile_yx [t,2] int32. */
int nb_geom_tiles_f32(const uint8_t* geom, int gh, int gw, const int32_t* tile_yx, int t, int r, float* out,
                      void* stream);

/* Feature-canvas blending for a SEQUENCE of full tiles in one launch.  Replaces, for tiles 0..t-1 in order,
 * PaintingHelper._get_blended_features (forger/ui/brush.py:190-227) + BlendedFeatures.blend (forger/train/
 * stitching.py:24-25, applied in networks_modified.py:176-181) + FeatureCanvas.set_features (brush.py:82-92):
 *   upd = alpha0 > 0.99 | (mask & alpha0 > 0), cleared inside the `crop` border;  a = 1 - (mask ? alpha0 : 1);
 *   tiles[t] = a * canvas + (1 - a) * tiles[t];  canvas[upd] = tiles[t][upd];  mask |= upd.
 * tiles [t,c,hw,hw] holds the block output before blending on entry and the blended features on return;
 * tile_yx [t,2] are tile origins on the feature canvas; alpha0 [hw,hw] is generate_dirty_area_alpha
 * (brush.py:159-187) for a full tile; canvas [c,hc,wc] / mask_in / mask_out [hc,wc] are the FeatureCanvas state
 * (mask_out must not alias mask_in).  Cells are taken over the feature canvas. */
int nb_canvas_replay_f32(float* tiles, int t, int c, int hw, const int32_t* tile_yx, const float* alpha0, int crop,
                         float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc,
                         const int32_t* cell_off, const int32_t* cell_tiles, void* stream);

/* The same restricted to the cells [cell_y0, cell_y0 + cells_y) x [cell_x0, cell_x0 + cells_x) (cell = NB_CELL_H x NB_CELL_W
 * canvas pixels, the granularity of cell_off): for the interactive stroke, which replays ONE tile on a large canvas.
 * mask_out is only written inside the box - the caller starts it as a copy of mask_in. */
int nb_canvas_replay_box_f32(float* tiles, int t, int c, int hw, const int32_t* tile_yx, const float* alpha0,
                             int crop, float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc,
                             const int32_t* cell_off, const int32_t* cell_tiles, int cell_x0, int cell_y0,
                             int cells_x, int cells_y, void* stream);

/* The replay over tile PIECES, for the multi-GPU halo exchange ("P2P halo send/recv of only the overlapped
 * strips"): a rank replays its own tiles together with the strips of earlier foreign tiles that overlap them.  A piece is
 * a rectangle [ly0, ly0+h) x [lx0, lx0+w) of one tile, sitting at (cy, cx) on the feature canvas, with its values at
 * data[ch * cstride + row * rstride + col] (a full own tile: h = w = hw, rstride = hw, cstride = hw*hw; a received strip:
 * a compact buffer).  Pieces of one tile must be disjoint; ascending piece index = paint order (cell_pieces lists piece
 * indices).  Same per-pixel arithmetic as nb_canvas_replay_f32; only the cells of the given box run and mask_out is only
 * written there (start it as a copy of mask_in). */
typedef struct NbTilePiece {
    uint64_t data;               /* device pointer (float*) */
    int32_t cstride, rstride;    /* in floats */
    int32_t cy, cx, h, w;        /* piece rectangle on the feature canvas */
    int32_t ly0, lx0;            /* its origin inside the tile (alpha template / crop border lookup) */
} NbTilePiece;
int nb_canvas_replay_pieces_f32(const NbTilePiece* pieces, int n, int c, int hw, const float* alpha0, int crop,
                                float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc,
                                const int32_t* cell_off, const int32_t* cell_pieces, int cell_x0, int cell_y0,
                                int cells_x, int cells_y, void* stream);

/* Paste RGBA8 tiles [t,r,r,4] into canvas [h,w,4]: the interior [crop, r-crop)^2 of tile i lands at
 * dst_yx[i] + crop (brush.py:369-374 crop + out_meta, paint_image_main.py:173-177 paste); later tiles
 * overwrite earlier ones.  Cells are taken over the RGBA canvas from the interior rectangles. */
int nb_paste_tiles_u8(const uint8_t* tiles, int t, int r, const int32_t* dst_yx, int crop, uint8_t* canvas, int h, int w,
                      const int32_t* cell_off, const int32_t* cell_tiles, void* stream);

/* ---- geometry encoder (forger/experimental/autoenc/simple_autoencoder.py:88-121, 155-199,
 * 251-261).  Every layer is conv(reflect padding) + bias + LeakyReLU(slope) with eval-mode BatchNorm folded into
 * the weights and bias by the caller.  Activations between layers travel in the H2 format of the split-f16 convs. */

/* Stem: 1 -> 64 channels, 7x7, reflect padding 3.  x fp32 [n,1,h,w] (h % 16 == 0, w % 32 == 0), w50 [64][50] =
 * folded weights [c_out][ky*7+kx] padded with one zero, y_h2 = H2 [n,8,2,h,w,8].  preproc (autoenc/base.py:30-52):
 * 0 none, 1 '-11inverse' (1-x)*2-1, 2 'inverse' 1-x.  Exact fp32 products (v_mfma_f32_32x32x2_f32). */
int nb_enc_stem7x7_f32_h2(const float* x, const float* w50, const float* bias, void* y_h2, int n, int h, int w,
                          int preproc, float slope, void* stream);

/* 3x3 conv, stride 1 or 2, reflect padding 1, split-f16 products.  x_h2: H2 [n, c_in, h_in, w_in]; w_h3: hi/lo f16
 * [ceil(c_in/16)][3][3][2][2][ceil128(c_out)][8]; exactly one of y_f32 (fp32 NCHW [n,c_out,h_in/stride,w_in/stride])
 * and y_h2 (H2, c_out % 8 == 0) is non-NULL.  Output must be a multiple of 16 wide (rows % 16 == 0) or a multiple of 32 wide
 * (rows % 8 == 0). */
int nb_enc_conv3x3_h3(const void* x_h2, int c_in, const void* w_h3, const float* bias, float* y_f32, void* y_h2, int n,
                      int h_in, int w_in, int c_out, int stride, float slope, void* stream);

/* The same convolution writing straight into a CONSUMER's split-f16 operand tensor (the generator layer that takes these
 * geometry features as extra input channels, networks_modified.py:218-219 `torch.cat`): the result times
 * oscale[n * oscale_stride + co] (the consumer's styles of those channels; NULL = 1) goes into channel groups
 * cg0 .. cg0 + c_out/8 - 1 of y_h2 [n][c8_total][2][h][w][8]; out_fmt 0 = H2 (hi/lo f16), 1 = "f8" (hi f16 + fp8
 * correction operands: c_out % 16 == 0, cg0 even).  Removes the fp32 round trip + nb_pack_h2*_part_f32 pass. */
int nb_enc_conv3x3_h3_handoff(const void* x_h2, int c_in, const void* w_h3, const float* bias, void* y_h2,
                              const float* oscale, int oscale_stride, int c8_total, int cg0, int out_fmt,
                              int n, int h_in, int w_in, int c_out, int stride, float slope, void* stream);

/* Bilinear x2, align_corners=True (nn.Upsample in ScaleUp, simple_autoencoder.py:106-121): fp32 NCHW [n,c,h,w]
 * (c % 8 == 0) -> H2 [n, c, 2h, 2w]. */
int nb_enc_upsample2x_h2(const float* x, void* y_h2, int n, int c, int h, int w, void* stream);

/* The encoder layers with the "f8" operand format BETWEEN them (the format of the generator's f8 arithmetic mode: hi f16
 * + fp8 correction operands per 16-channel chunk; one f16 + half an fp8 MFMA per tap instead of three f16 MFMAs).
 * in_fmt / out_fmt: 0 = H2, 1 = f8 (c % 16 == 0).  For in_fmt 1 the weights are packed like the generator's f8 weights:
 * lo slot of chunk group 0 = fp8(w), of group 1 = fp8((w - f16(w)) 2^11), over the 16 channels of the chunk
 * ([ceil(c_in/16)][3][3][2][2][ceil128(c_out)][8] containers as for H2).  nb_enc_conv3x3_ex is the general form of the
 * two entry points above (oscale / c8_total / cg0 as in the hand-off; c8_total 0 = a plain tensor of c_out channels). */
int nb_enc_stem7x7_f32_h2_ex(const float* x, const float* w50, const float* bias, void* y_h2, int out_fmt, int n, int h, int w,
                             int preproc, float slope, void* stream);
int nb_enc_conv3x3_ex(const void* x, int c_in, const void* wts, const float* bias, float* y_f32, void* y_h2,
                      const float* oscale, int oscale_stride, int c8_total, int cg0, int in_fmt, int out_fmt,
                      int n, int h_in, int w_in, int c_out, int stride, float slope, void* stream);

/* Stem + first stride-2 stage in ONE launch (simple_autoencoder.py:155-176: _SingleConvolution 1 -> 64, 7 x 7, followed by the first
 * _Downsample stage 64 -> c_out, stride 2): the result of nb_enc_stem7x7_f32_h2_ex(out_fmt 1) followed by nb_enc_conv3x3_ex(in_fmt 1,
 * stride 2) without the 64-channel full-resolution tensor between them -- each workgroup computes the stem outputs its tile needs on the
 * matrix pipe, chunk by chunk, straight into LDS in operand format.  x fp32 [n,1,h,w] (h % 16 == 0, w % 64 == 0); w50 / bias0 / preproc
 * as nb_enc_stem7x7_f32_h2; wts1: the stage's weights in "f8" format (c_in = 64), bias1 [c_out], c_out % 16 == 0; y_h2
 * [n, c_out, h/2, w/2] in format out_fmt (0 = H2, 1 = f8).  Same split-f16 products; per stem output the summation order differs
 * from the stem kernel's (results agree to fp32 rounding, not bit for bit). */
int nb_enc_stem_conv3x3_f8(const float* x, const float* w50, const float* bias0, int preproc, const void* wts1, const float* bias1,
                           void* y_h2, int out_fmt, int n, int h, int w, int c_out, float slope, void* stream);

/* Stride-2 3x3 correlation WITHOUT padding on the same kernel -- the strided half of conv2d_resample's down-sampling branch
 * (conv2d_resample.py:96-113) and the input gradient of its up-sampling branch (:124-147), which cuDNN runs for the reference:
 * x H2 [n][c8][2][2ho+1][2wo+1][8], weights as above, y fp32 [n][c_out][ho][wo] = oscale[n][co] * (bias[co] + sum_{ci,a,b}
 * x[n,ci,2i+a,2j+b] w[co,ci,a,b]); oscale may be NULL.  Output sizes: wo % 32 == 0 and ho % 8 == 0, or wo == 16 and ho % 16 == 0. */
int nb_conv3x3_s2_valid_h3(const void* x_h2, int c_in, const void* w_h3, const float* bias, const float* oscale, int oscale_stride,
                           float* y_f32, int n, int h_in, int w_in, int c_out, void* stream);
int nb_enc_upsample2x_h2_ex(const float* x, void* y_h2, int out_fmt, int n, int c, int h, int w, void* stream);

/* Host-side helper (no GPU): repack W[c_out,c_in,3,3] into the zero-padded
 * wpk[ceil8(c_in)][9][ceil32(c_out)] and wsq[c_in][c_out] = sum_k W^2.  Either output may be NULL. */
int nb_pack_conv_weight(const float* w, int c_out, int c_in, float* wpk, float* wsq);

/* ---- box calibration (csrc/nb_calib.hip; measurement infrastructure, not on the generator's path) -----------------
 * A registers-only loop of back-to-back v_mfma_f32_32x32x16_f16 on random operands, one wave per SIMD on every CU of the current
 * device, for about target_ms (blocking).  *tflops = the dense f16 matrix rate this device sustains, *ms = duration of the measured
 * launch, *clock_mhz = median in-kernel shader clock during it (may be NULL).  MI355X boards differ by several percent in the
 * clock they hold under matrix load; bench.py reports the figure beside its own (box_calibration, roofline.frac_of_sustained).
 * The reference has no counterpart (forger/util/timer.py:11-32 is its only timer). */
int nb_calibrate_mfma_f16(double target_ms, double* tflops, double* ms, double* clock_mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUBE_HIP_H */
