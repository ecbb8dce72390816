/*
 * pcgan_hip.h — C ABI of libpcgan_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * Generator/Discriminator training step of flash4242/Promptable-Counterfactual-GAN.
 *
 * The reference has no FFI of its own: every FLOP of its hot path is a torch.nn / torch.optim call
 * (SURVEY.md §8b).  Each entry point below therefore cites the reference call site whose ATen
 * operator it replaces; the Python shim (promptable-counterfactual-gan_amd/) binds them with ctypes
 * and hands over `tensor.data_ptr()` of PyTorch-ROCm tensors.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types; no exceptions.  Every function returns PCG_OK (0)
 *    or a negative pcg_status; pcg_last_error() gives the text for the calling thread.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises, nothing
 *    allocates: every buffer, including workspaces, is owned by the caller and only borrowed until
 *    the enqueued work has run.  All launches are hipGraph-capturable.
 *  - Activations are fp32 NHWC  [B][H][W][C]  (PyTorch: torch.channels_last tensors of shape [B,C,H,W]).
 *    Conv2d weights are fp32 OHWI [Cout][KH][KW][Cin] (PyTorch: channels_last tensor [Cout,Cin,KH,KW]);
 *    ConvTranspose2d weights [Cin][KH][KW][Cout] (channels_last tensor [Cin,Cout,KH,KW]) — which is
 *    the OHWI weight of the adjoint convolution, so one geometry struct serves both layer kinds:
 *        Conv2d           forward = pcg_conv2d_fwd,   grad-input = pcg_conv2d_dgrad, grad-weight = pcg_conv2d_wgrad
 *        ConvTranspose2d  forward = pcg_conv2d_dgrad, grad-input = pcg_conv2d_fwd,   grad-weight = pcg_conv2d_wgrad
 *    (for the transposed layer, `x` of the geometry is the layer's OUTPUT side and `y` its INPUT side).
 *  - Labels / class indices are int64 (torch.long), as in the reference.
 */
#ifndef PCGAN_HIP_H
#define PCGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pcg_stream_t; /* hipStream_t */

typedef enum pcg_status {
  PCG_OK = 0,
  PCG_ERR_INVALID = -1,     /* bad argument / unsupported shape */
  PCG_ERR_WORKSPACE = -2,   /* workspace too small */
  PCG_ERR_LAUNCH = -3,      /* HIP launch error */
  PCG_ERR_UNSUPPORTED = -4
} pcg_status;

typedef enum pcg_act {
  PCG_ACT_NONE = 0,
  PCG_ACT_RELU = 1,
  PCG_ACT_LRELU = 2,   /* slope given separately */
  PCG_ACT_TANH = 3,
  PCG_ACT_SIGMOID = 4
} pcg_act;

/* y[b,oh,ow,co] = bias[co] + sum_{kh,kw,ci} x[b, oh*stride-pad+kh, ow*stride-pad+kw, ci] * w[co,kh,kw,ci] */
typedef struct pcg_conv_geom {
  int32_t B;
  int32_t IH, IW, Cin;   /* x: [B][IH][IW][Cin]  */
  int32_t OH, OW, Cout;  /* y: [B][OH][OW][Cout] */
  int32_t KH, KW, stride, pad;
} pcg_conv_geom;

/* ---- library ------------------------------------------------------------------------------- */
int pcg_abi_version(void);
const char* pcg_last_error(void);
/* "gfx950" — the only code object in the library */
const char* pcg_target_arch(void);

/* ---- convolution family (MFMA f32 implicit GEMM; thin Cin==1 / Cout==1 layers on the vector ALU) -
 * replaces nn.Conv2d / nn.ConvTranspose2d forward + autograd:
 *   dconv_gan/mnist/mnist_dcgan.py:76-88 (G ConvT stack), :100-111 (D Conv stack);
 *   conditional_counteRGAN/mnist/models/generator.py:11-14,39,49-50; models/discriminator.py:14-24;
 *   models/classifier.py:7-12.                                                                    */
/* Optional scratch for fwd / dgrad: 0 for MFMA layers; for a layer whose reduced side has ONE channel (Cout==1 fwd,
 * Cin==1 dgrad) it buys the two-stage "tap dot + col2im" path that reads the wide tensor exactly once.  Passing
 * workspace == NULL is always valid (a slower single-kernel path is used).                                        */
size_t pcg_conv2d_fwd_workspace_bytes(const pcg_conv_geom* g);
size_t pcg_conv2d_dgrad_workspace_bytes(const pcg_conv_geom* g);
int pcg_conv2d_fwd(const pcg_conv_geom* g, const float* x, const float* w, const float* bias /*nullable*/,
                   float* y, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_dgrad(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x /*nullable: added per Cin channel (ConvTranspose2d bias)*/,
                     float* dx, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
/* The same with the activation that follows the layer fused into the output write — y = act(conv(x) + bias):
 * Conv2d + LeakyReLU (mnist_dcgan.py:100-101, counteRGAN models/discriminator.py:17-24, generator.py:36-37,50),
 * Conv2d + ReLU (models/classifier.py:7-12), Conv2d + Sigmoid (mnist_dcgan.py:110-111), ConvTranspose2d + Tanh
 * (mnist_dcgan.py:88-89, mnist_wgan_conditional.py:70-71).  The backward needs only y (pcg_act_bwd). */
int pcg_conv2d_fwd_act(const pcg_conv_geom* g, const float* x, const float* w, const float* bias /*nullable*/, int act, float slope,
                       float* y, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_dgrad_act(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x /*nullable*/, int act, float slope,
                         float* dx, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
size_t pcg_conv2d_wgrad_workspace_bytes(const pcg_conv_geom* g);
/* Deferred slab reductions (r04).  A split-K weight gradient ends with a small launch that sums its K-slice slabs into dw; in a backward
 * sweep nothing reads dw before the sweep ends (optimizer.step(), mnist_dcgan.py:164,176; the gradient exchange).  Between
 * pcg_slab_defer_begin(stream) and pcg_slab_defer_flush(stream) the pcg_conv2d_wgrad* calls of THIS thread on `stream` leave their slabs
 * unreduced and the flush sums all of them in ONE launch — bit-identical to the per-call reductions (same order per element).  The
 * caller must give every deferred call its own workspace and keep it until the flush; two sums into the same dw stay in call order.
 * No nesting.  pcg_slab_defer_pending(): recorded entries, -1 when not deferring.                                                  */
int pcg_slab_defer_begin(pcg_stream_t stream);
int pcg_slab_defer_flush(pcg_stream_t stream);
int32_t pcg_slab_defer_pending(void);
/* dw[co,kh,kw,ci] (+)= sum_{b,oh,ow} dy[b,oh,ow,co] * x[b,oh*s-p+kh,ow*s-p+kw,ci];  accumulate!=0 adds into dw
 * (the reference accumulates .grad over two backward() calls: mnist_dcgan.py:153,161).             */
int pcg_conv2d_wgrad(const pcg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate,
                     void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* Convolution + training-mode BatchNorm statistics of its output in one call (MFMA layers only): the per-channel sum /
 * sum of squares are taken from the accumulator tile in the conv epilogue (no extra pass over y), partial rows are
 * finalised in a fixed order.  Same outputs as pcg_conv2d_fwd (resp. _dgrad) followed by pcg_bn_train_stats.
 * workspace: pcg_conv2d_{fwd,dgrad}_bn_workspace_bytes (0 = layer not eligible: use the two separate calls).          */
size_t pcg_conv2d_fwd_bn_workspace_bytes(const pcg_conv_geom* g);
size_t pcg_conv2d_dgrad_bn_workspace_bytes(const pcg_conv_geom* g);
int pcg_conv2d_fwd_bn(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                      float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_dgrad_bn(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, float* dx,
                        float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* ---- input transforms: BatchNorm + ReLU / LeakyReLU of the PRODUCING layer applied inside the consumer's gather ------------------
 * In `Conv -> BatchNorm(train) -> LeakyReLU -> Conv` (mnist_dcgan.py:102-110, G: :76-87) the activated tensor a = act(bn(z)) is
 * only ever read by the next convolution (forward, and again by that layer's weight gradient).  With a transform the consumer
 * reads the pre-BatchNorm tensor z itself and evaluates act(z*scale[c] + shift[c]) between the gather and the LDS write, so the
 * separate BatchNorm-apply pass and the activated copy of every such activation disappear.  scale / shift are the folded
 * BatchNorm (scale = gamma*invstd, shift = beta - mean*scale — exactly the expression pcg_bn_apply_act evaluates, so results are
 * bit-identical to the unfused sequence); zero padding stays zero.  MFMA layers only (Cin > 3, Cout > 3, channel counts % 4 == 0).
 *   _fwd_bn_xf / _dgrad_bn_xf : as pcg_conv2d_{fwd,dgrad}_bn with the transform `xf` (nullable) on the input operand (x resp. dy
 *                               — the forward input of a ConvTranspose2d layer) and, if coef_out != NULL, the folded scale / shift
 *                               [2][C] of THIS layer's BatchNorm (gamma, beta required) written by the statistics finalize —
 *                               ready to be the next consumer's transform;
 *   _fwd_xf / _dgrad_xf       : as pcg_conv2d_{fwd,dgrad}_act with a transform on the input operand;
 *   _wgrad_xf                 : weight gradient with the transform on x (Conv2d) or on dy (ConvTranspose2d, whose forward input
 *                               sits on the dy side of the adjoint geometry) — at most one of the two. */
typedef struct pcg_in_xform {
  const float* scale;   /* [C], 16-byte aligned; NULL: no transform */
  const float* shift;   /* [C] */
  int32_t act;          /* PCG_ACT_NONE / PCG_ACT_RELU / PCG_ACT_LRELU */
  float slope;
} pcg_in_xform;
int pcg_conv2d_fwd_bn_xf(const pcg_conv_geom* g, const float* x, const pcg_in_xform* xf, const float* w, const float* bias, float* y,
                         float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, const float* gamma, const float* beta, float* coef_out /*nullable [2][Cout]*/,
                         void* workspace, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_dgrad_bn_xf(const pcg_conv_geom* g, const float* dy, const pcg_in_xform* xf, const float* w, const float* bias_x, float* dx,
                           float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, const float* gamma, const float* beta, float* coef_out /*nullable [2][Cin]*/,
                           void* workspace, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_fwd_xf(const pcg_conv_geom* g, const float* x, const pcg_in_xform* xf, const float* w, const float* bias, int act,
                      float slope, float* y, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_dgrad_xf(const pcg_conv_geom* g, const float* dy, const pcg_in_xform* xf, const float* w, const float* bias_x, int act,
                        float slope, float* dx, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_wgrad_xf(const pcg_conv_geom* g, const float* x, const pcg_in_xform* xf_x, const float* dy, const pcg_in_xform* xf_dy,
                        float* dw, int accumulate, void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* Backward-pass epilogues (MFMA layers).  In autograd's sweep through `Conv -> [BatchNorm] -> ReLU/LeakyReLU -> Conv` (D:
 * mnist_dcgan.py:100-110, G: :76-87) the gradient w.r.t. a layer's OUTPUT a = act(..) is produced by the grad-input kernel of the
 * layer above; these entry points apply the lower layer's activation derivative while that tile is still in registers:
 *   _mask  : dx = conv_dgrad(dy, w) * act'(a_below)            (a_below = the post-activation output, same shape as dx) —
 *            replaces the pcg_act_bwd pass of a Conv + LeakyReLU layer without BatchNorm (mnist_dcgan.py:100-101);
 *   _bnbwd : dx = conv_dgrad(dy, w) * act'(bn(z_below)) with the mask recomputed from the pre-BatchNorm output z_below as
 *            fma(z, gamma*invstd, beta - mean*gamma*invstd) > 0, and the per-channel sums of dx and dx*xhat written as partial
 *            rows ([pcg_conv2d_*_bn_partial_rows][2][C], buffer of pcg_conv2d_*_bn_workspace_bytes) — the reduction pass of
 *            BatchNorm's backward disappears; finish with pcg_bn_bwd_partial.
 * The `fwd` forms are the same for ConvTranspose2d layers (their grad-input is a forward convolution).  act: none / ReLU / LeakyReLU. */
int pcg_conv2d_dgrad_mask(const pcg_conv_geom* g, const float* dy, const float* w, const float* a_below, int act, float slope,
                          float* dx, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
/* r04: pcg_conv2d_dgrad_mask also serves one-channel layers whose grad-input takes the row-block form (..._thin_ok: k3 / k4, Cout
 * <= 3, Cin a power-of-two multiple of 4): the mask is applied in the thin expand kernel (CounteRGAN conv_out, models/generator.py:50). */
int32_t pcg_conv2d_dgrad_mask_thin_ok(const pcg_conv_geom* g);
/* Thin layers that take an input transform (r04): a Cin = 1 geometry with Cout = 64 and <= 16 taps — DCGAN's last ConvTranspose2d(64, 1, 4, 2, 1)
 * (mnist_dcgan.py:88), whose input is BatchNorm2d + ReLU of the layer before (:86-87).  pcg_conv2d_dgrad_xf (its forward) and
 * pcg_conv2d_wgrad_xf with xf_dy (its weight gradient) then read the PRE-BatchNorm tensor: the BatchNorm-apply pass and the activated
 * copy of that activation disappear (both kernels are HBM-bound: the transform is free).  Needs the workspace of
 * pcg_conv2d_dgrad_workspace_bytes / pcg_conv2d_wgrad_workspace_bytes.                                                              */
int32_t pcg_conv2d_xf_thin_ok(const pcg_conv_geom* g);
/* y = (conv(x, w) + addend) * act'(a_below) — pcg_conv2d_*_add followed by pcg_act_bwd in ONE epilogue: the last skip-add of a residual
 * chain arriving at the entry convolution's LeakyReLU (models/generator.py:76 backward).  a_below: the activated output, output's shape. */
int pcg_conv2d_fwd_add_mask(const pcg_conv_geom* g, const float* x, const float* w, const float* addend, const float* a_below, int act,
                            float slope, float* y, pcg_stream_t stream);
int pcg_conv2d_dgrad_add_mask(const pcg_conv_geom* g, const float* dy, const float* w, const float* addend, const float* a_below, int act,
                              float slope, float* dx, pcg_stream_t stream);
int pcg_conv2d_fwd_mask(const pcg_conv_geom* g, const float* x, const float* w, const float* a_below, int act, float slope,
                        float* y, void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_dgrad_bnbwd(const pcg_conv_geom* g, const float* dy, const float* w, const float* z_below, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dx,
                           void* partial, size_t partial_bytes, pcg_stream_t stream);
int pcg_conv2d_fwd_bnbwd(const pcg_conv_geom* g, const float* x, const float* w, const float* z_below, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, int act, float slope, float* y,
                         void* partial, size_t partial_bytes, pcg_stream_t stream);
/* dx = conv_dgrad(dy, w) + addend  (addend may be dx itself: in-place accumulation) — the skip connection of a residual block
 * in the backward sweep (`x + 0.1*out`, conditional_counteRGAN/mnist/models/generator.py:20): no separate add pass.  MFMA layers. */
int pcg_conv2d_dgrad_add(const pcg_conv_geom* g, const float* dy, const float* w, const float* addend, float* dx,
                         void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
/* pcg_conv2d_dgrad_add + the BatchNorm-backward column sums of the SUM for the next BatchNorm down the skip chain
 * (`x + 0.1*bn2(conv2(...))`, models/generator.py:18-20: the gradient arriving at block i-1's bn2 is the sum block i's backward just
 * formed): partial rows get sum(sum_scale*dx) and sum(sum_scale*dx*xhat) with xhat from z_next / mean / invstd of that BatchNorm;
 * dx itself is stored unscaled.  Finish with pcg_bn_bwd_partial_db(dm = dx, dm_scale = sum_scale).  partial: the buffer of
 * pcg_conv2d_dgrad_bn_workspace_bytes, pcg_conv2d_dgrad_bn_partial_rows rows.  addend == NULL (r04, both forms): the plain grad-input
 * with the sums — the head of the chain, conv_mid's grad-input above the last block's bn2 (generator.py:49,78). */
int pcg_conv2d_dgrad_add_bnsum(const pcg_conv_geom* g, const float* dy, const float* w, const float* addend /*nullable*/, const float* z_next,
                               const float* mean, const float* invstd, float sum_scale, float* dx, void* partial, size_t partial_bytes,
                               pcg_stream_t stream);
/* The `fwd` forms of the skip-add epilogues (a ConvTranspose2d's grad-input; or a stride-1 Conv2d's grad-input run as a forward
 * convolution of dy with the adjoint weight — pcg_conv_weight_adjoint: w[co][kh][kw][ci] -> w_adj[ci][KH-1-kh][KW-1-kw][co], geometry
 * {B, OH, OW, Cout -> IH, IW, Cin, pad' = K-1-pad} — which makes both GEMM operands K-major). */
int pcg_conv2d_fwd_add(const pcg_conv_geom* g, const float* x, const float* w, const float* addend, float* y,
                       void* workspace /*nullable*/, size_t workspace_bytes, pcg_stream_t stream);
int pcg_conv2d_fwd_add_bnsum(const pcg_conv_geom* g, const float* x, const float* w, const float* addend, const float* z_next,
                             const float* mean, const float* invstd, float sum_scale, float* y, void* partial, size_t partial_bytes,
                             pcg_stream_t stream);
int pcg_conv_weight_adjoint(const float* w, float* w_adj, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, pcg_stream_t stream);
/* The same for n <= 16 layers of one shape in ONE launch (host arrays of device pointers). */
int pcg_conv_weight_adjoint_many(const float* const* w, float* const* w_adj, int32_t n, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin,
                                 pcg_stream_t stream);
int32_t pcg_conv2d_fwd_bn_partial_rows(const pcg_conv_geom* g);
int32_t pcg_conv2d_dgrad_bn_partial_rows(const pcg_conv_geom* g);
/* db[c] (+)= sum_rows dy[row][c]   (bias gradient of Conv2d / Linear; rows = B*OH*OW)             */
size_t pcg_colsum_workspace_bytes(int64_t rows, int32_t C);
int pcg_colsum(const float* dy, int64_t rows, int32_t C, float* db, int accumulate,
               void* workspace, size_t workspace_bytes, pcg_stream_t stream);

/* ---- BatchNorm (training mode) + activation --------------------------------------------------
 * replaces nn.BatchNorm2d(train) + nn.ReLU / nn.LeakyReLU(0.2): mnist_dcgan.py:77-87,103-110;
 * models/generator.py:12-15,18-20 (counteRGAN _ResBlock).  [torch] semantics: normalise with the
 * biased batch variance, eps inside the sqrt, running_var updated with the unbiased variance,
 * running = (1-momentum)*running + momentum*batch.                                                 */
size_t pcg_bn_workspace_bytes(int64_t rows, int32_t C);
/* stats over rows = B*H*W of x[rows][C]: writes save_mean[C], save_invstd[C]; updates running_* if non-null */
int pcg_bn_train_stats(const float* x, int64_t rows, int32_t C, float eps, float momentum,
                       float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked /*nullable: += 1*/,
                       void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* the same + the folded scale / shift [2][C] of y = x*scale + shift (gamma, beta required when coef_out != NULL) */
int pcg_bn_train_stats_coef(const float* x, int64_t rows, int32_t C, float eps, float momentum,
                            float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, const float* gamma, const float* beta, float* coef_out,
                            void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* y = act( (x-mean)*invstd*gamma + beta ).  var_eps < 0: `invstd_or_var` holds invstd (training: save_invstd);
 * var_eps >= 0: it holds a variance and invstd = rsqrt(var + var_eps) (eval mode: running_var, eps).   */
int pcg_bn_apply_act(const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd_or_var,
                     float var_eps, const float* gamma, const float* beta, int act, float slope,
                     const float* residual /*nullable*/, float alpha /* y = residual + alpha*act(bn(x)) :
                     counteRGAN _ResBlock `x + 0.1*out`, models/generator.py:20; plain BN: NULL, 1 */,
                     float* y, pcg_stream_t stream);
/* backward of y = act(bn(x)):  given dy (grad wrt y), x (pre-BN), y (post-activation: the sign mask)
 *   dgamma (+)= sum dz*xhat ; dbeta (+)= sum dz ; dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)),
 *   dz = dy*act'(.)     ([torch] LeakyReLU/ReLU sub-gradient at 0 is the negative-side one: uses y>0) */
int pcg_bn_act_bwd(const float* dy, const float* x, const float* y /*nullable when act==NONE*/, int64_t rows, int32_t C,
                   const float* mean, const float* invstd, const float* gamma,
                   int act, float slope, float dy_scale /* dz = dy_scale*dy*act'(.) : the alpha of the forward */,
                   float* dx, float* dgamma, float* dbeta, int accumulate,
                   void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* The same without reading y: for ReLU / LeakyReLU the mask act'(y) is the sign of the BatchNorm output, recomputed from x with the
 * forward's own expression fma(x, gamma*invstd, beta - mean*gamma*invstd) — two HBM reads of the activation less per layer. */
int pcg_bn_act_bwd_premask(const float* dy, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int act, float slope, float dy_scale, float* dx, float* dgamma,
                           float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, pcg_stream_t stream);

/* BatchNorm backward from the partial sums a pcg_conv2d_*_bnbwd call left behind: dm = dy*act'(.) (already masked), partial =
 * fp64 [nparts][2][C] (sum dm, sum dm*xhat: every per-channel sum of the BatchNorm family is accumulated in double, as the
 * reference's CPU path does — [torch] at::acc_type<float, false>) in the opaque, 8-byte-aligned buffer of
 * pcg_conv2d_*_bn_workspace_bytes (its tail is scratch for the two-level
 * finalize used when there are >= 4096 partial rows).  Fixed-order fp64 finalize (dgamma, dbeta as in pcg_bn_act_bwd) + the elementwise pass
 * dx = gamma*invstd*(dm - mean(dm) - xhat*mean(dm*xhat)).  workspace: pcg_bn_bwd_partial_workspace_bytes(C). */
size_t pcg_bn_bwd_partial_workspace_bytes(int32_t C);
int pcg_bn_bwd_partial(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                       const float* gamma, const void* partial, int32_t nparts, float* dx, float* dgamma, float* dbeta,
                       int accumulate, void* workspace, size_t workspace_bytes, pcg_stream_t stream);

/* The same two backward calls, additionally returning the column sums of the dx they write: dcol[c] (+)= sum_rows dx[row][c] — the
 * bias gradient of the convolution in front of this BatchNorm (models/generator.py:11-14: Conv2d(bias=True) -> BatchNorm2d; the sum is
 * analytically zero, the reference computes its rounding residue, so it is computed) taken while dx streams out of the apply pass
 * instead of by a separate pcg_colsum pass over it (fp64 per thread, fixed-order block sums, fixed-order finalize).
 * y == NULL with ReLU / LeakyReLU and beta given: the mask is recomputed (as pcg_bn_act_bwd_premask). */
size_t pcg_bn_db_workspace_bytes(int64_t rows, int32_t C);
int pcg_bn_act_bwd_db(const float* dy, const float* x, const float* y /*nullable*/, int64_t rows, int32_t C, const float* mean,
                      const float* invstd, const float* gamma, const float* beta /*nullable*/, int act, float slope, float dy_scale,
                      float* dx, float* dgamma, float* dbeta, int accumulate, float* dcol, int accumulate_col, void* workspace,
                      size_t workspace_bytes, pcg_stream_t stream);
size_t pcg_bn_bwd_partial_db_workspace_bytes(int32_t C);
/* dm_scale: the partial sums already include this factor, the apply pass multiplies dm by it (1: plain) — the 0.1 of a residual
 * branch whose incoming gradient is the unscaled skip-path sum (pcg_conv2d_dgrad_add_bnsum).  dcol may be NULL. */
int pcg_bn_bwd_partial_db(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                          const float* gamma, const void* partial, int32_t nparts, float dm_scale, float* dx, float* dgamma,
                          float* dbeta, int accumulate, float* dcol /*nullable*/, int accumulate_col, void* workspace,
                          size_t workspace_bytes, pcg_stream_t stream);

/* ---- grouped batches (r04): the discriminator's real and fake passes as ONE pass --------------------------------------------
 * mnist_dcgan.py:151-161 runs netD twice per D step — on the real batch and on fake.detach() — and adds the two backward passes
 * into .grad before optimizerD.step() (:164).  Neither pass depends on the other, so both run as one tensor of `groups` (= 2)
 * batches side by side along the batch axis: one convolution launch per layer and direction over 2B images, one weight-gradient
 * launch whose sum over pixels covers both batches (the .grad accumulation of :153,161).  BatchNorm keeps the reference's
 * semantics: every group has its OWN batch statistics (save_mean / save_invstd / the backward's two means are [groups][C], taken
 * from the group's own 64-row partial sums in the fixed order of the ungrouped finalize), the running statistics are updated
 * once per group in group order, num_batches_tracked advances by `groups`; dgamma / dbeta are the groups' sums added in group
 * order.  Against two separate passes the results differ only where a sum's ORDER differs: the weight gradients (one K-loop over
 * 2B images instead of two added results) — a few ulp, stated in tests/test_hip_groups.py.  Constraints: C / 4 a power of two
 * <= 256, (B / groups) * OH * OW a multiple of 128, equal sub-pixel phases; not available in the exact-BatchNorm data-parallel
 * mode.  Callers fall back to separate passes otherwise (pcgan_amd.nn.SequentialConvNet.supports_groups).                      */
int pcg_conv2d_fwd_bn_g(const pcg_conv_geom* g, const float* x, const float* w, const float* bias /*nullable*/, float* y, float eps,
                        float momentum, float* save_mean /*[groups][Cout]*/, float* save_invstd /*[groups][Cout]*/,
                        float* running_mean /*nullable*/, float* running_var /*nullable*/, int64_t* num_batches_tracked /*nullable*/,
                        int32_t groups, void* workspace /*pcg_conv2d_fwd_bn_workspace_bytes*/, size_t workspace_bytes, pcg_stream_t stream);
/* y = act(bn_g(x)) with group g's statistics on group g's rows (training mode only: mean / invstd are save_mean / save_invstd) */
int pcg_bn_apply_act_g(const float* x, int64_t rows, int32_t C, const float* mean /*[groups][C]*/, const float* invstd /*[groups][C]*/,
                       const float* gamma, const float* beta, int act, float slope, float* y, int32_t groups, pcg_stream_t stream);
/* pcg_conv2d_dgrad_bnbwd on `groups` side-by-side batches: mean / invstd of the layer below are [groups][Cin]; the partial rows keep
 * the ungrouped layout ([phase][tile row], pcg_conv2d_dgrad_bn_partial_rows of them in pcg_conv2d_dgrad_bn_phases phases).       */
int32_t pcg_conv2d_dgrad_bn_phases(const pcg_conv_geom* g);
int pcg_conv2d_dgrad_bnbwd_g(const pcg_conv_geom* g, const float* dy, const float* w, const float* z_below, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dx,
                             void* partial, size_t partial_bytes, int32_t groups, pcg_stream_t stream);
/* pcg_bn_bwd_partial on `groups` side-by-side batches (partial rows from pcg_conv2d_dgrad_bnbwd_g; nphases = the phases of that launch) */
size_t pcg_bn_bwd_partial_g_workspace_bytes(int32_t C, int32_t groups);
int pcg_bn_bwd_partial_g(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                         const float* gamma, const void* partial, int32_t nparts, int32_t nphases, float* dx, float* dgamma,
                         float* dbeta, int accumulate, int32_t groups, void* workspace, size_t workspace_bytes, pcg_stream_t stream);
/* pcg_bn_act_bwd_premask on `groups` side-by-side batches (BatchNorm + ReLU / LeakyReLU backward behind a thin layer: one column
 * reduction per group, one grouped finalize, one grouped apply)                                                               */
size_t pcg_bn_act_bwd_g_workspace_bytes(int64_t rows, int32_t C, int32_t groups);
int pcg_bn_act_bwd_premask_g(const float* dy, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int act, float slope, float* dx, float* dgamma, float* dbeta,
                             int accumulate, int32_t groups, void* workspace, size_t workspace_bytes, pcg_stream_t stream);

/* Thin forward fused with the BatchNorm backward of the layer below (r04).  A ConvTranspose2d with ONE output channel (DCGAN's G5,
 * mnist_dcgan.py:88) hands its grad-input — a forward convolution of the image gradient with Cin = 1 — to BatchNorm2d + ReLU
 * (:86-87).  The chain wrote that gradient (134 MB at batch 512), reduced it against z, and read both again; it costs 16 FMAs per
 * element to recompute, so here it never reaches memory: pass 1 leaves the two column sums, the usual finalize follows, pass 2
 * writes dz.  dgamma / dbeta as in pcg_bn_act_bwd.  Only k4, Cin = 1 geometries the row-block form covers (..._ok).               */
int32_t pcg_conv2d_fwd_bnbwd_thin_ok(const pcg_conv_geom* g);
size_t pcg_conv2d_fwd_bnbwd_thin_workspace_bytes(const pcg_conv_geom* g);
int pcg_conv2d_fwd_bnbwd_thin(const pcg_conv_geom* g, const float* x, const float* w, const float* z_below, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dz,
                              float* dgamma /*nullable*/, float* dbeta /*nullable*/, int accumulate, void* workspace,
                              size_t workspace_bytes, pcg_stream_t stream);
/* The same idea for a FULL-WINDOW Cout = 1 convolution above a BatchNorm layer (r04): DCGAN D's last conv, Conv2d(512, 1, 4, 1, 0) on
 * a 4x4 map (mnist_dcgan.py:110), whose grad-input dy[b] * w[hw][c] is ONE multiply per element; the layer below is Conv -> BatchNorm2d
 * -> LeakyReLU (:107-109).  Instead of  outer product (write d) -> column sums (read d, z) -> apply (read d, z; write dz)  the sums
 * pass reads z only and the apply pass reads z and writes dz.  `groups` side-by-side batches of B / groups samples with their own
 * statistics (mean / invstd [G][C]; the paired D step, section "grouped batches"); dgamma / dbeta: the groups' sums in group order.
 * Eligibility (..._ok): Cout = 1, window = input map, KH*KW a divisor of 256, (Cin/4) a multiple of 256/(KH*KW), B / groups a
 * multiple of 16; not in the exact-BatchNorm data-parallel mode.                                                                */
int32_t pcg_conv2d_dgrad_bnbwd_full_ok(const pcg_conv_geom* g, int32_t groups);
size_t pcg_conv2d_dgrad_bnbwd_full_workspace_bytes(const pcg_conv_geom* g, int32_t groups);
int pcg_conv2d_dgrad_bnbwd_full(const pcg_conv_geom* g, const float* dy, const float* w, const float* z_below, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dz,
                                float* dgamma /*nullable*/, float* dbeta /*nullable*/, int accumulate, int32_t groups, void* workspace,
                                size_t workspace_bytes, pcg_stream_t stream);

/* ... and its INPUT never needs the BatchNorm-apply pass (r04): the same full-window layer's forward and weight gradient read the
 * pre-BatchNorm output z of the layer below and evaluate act(bn(z)) in their loads, with that layer's batch statistics (mean / invstd
 * [groups][Cin], group = sample / (B / groups)) and the one bn_fold expression every BatchNorm pass uses — the values
 * pcg_bn_apply_act would have written.  Workspace of the weight gradient: pcg_conv2d_wgrad_workspace_bytes.
 * Eligibility: Cout = 1, window = input map, 256 % (Cin / 4) == 0, B / groups a multiple of 16.                                     */
int32_t pcg_conv2d_bnin_full_ok(const pcg_conv_geom* g, int32_t groups);
int pcg_conv2d_fwd_bnin_full(const pcg_conv_geom* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, int in_act, float in_slope, int32_t groups, const float* w, const float* bias /*nullable*/,
                             int act, float slope, float* y, pcg_stream_t stream);
int pcg_conv2d_wgrad_bnin_full(const pcg_conv_geom* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, int in_act, float in_slope, int32_t groups, const float* dy, float* dw, int accumulate,
                               void* workspace, size_t workspace_bytes, pcg_stream_t stream);

/* ---- pointwise activations (layers without BatchNorm) ----------------------------------------
 * nn.LeakyReLU after D's first conv (mnist_dcgan.py:101), nn.Tanh (:89), nn.Sigmoid (:112).        */
int pcg_act_fwd(const float* x, int64_t n, int act, float slope, float* y, pcg_stream_t stream);
/* dx = dy * act'(.) expressed through the OUTPUT y (relu/lrelu: y>0; tanh: 1-y^2; sigmoid: y(1-y)) */
int pcg_act_bwd(const float* dy, const float* y, int64_t n, int act, float slope, float* dx, pcg_stream_t stream);

/* ---- losses ----------------------------------------------------------------------------------
 * nn.BCELoss (mean): mnist_dcgan.py:125,152,160,172  ([torch]: log clamped at -100; backward
 * (p-t)/max(p(1-p),1e-12)/n).  loss is one float on the device; dp may be null (forward only).     */
int pcg_bce_fwd_bwd(const float* p, const float* target /*nullable → target_const*/, float target_const,
                    int64_t n, float grad_scale, const float* grad_out_dev /*nullable: one float, multiplies dp*/,
                    float* loss /*nullable*/, float* dp /*nullable*/, pcg_stream_t stream);
/* Two nn.BCELoss (mean) over the two halves of p in ONE launch: the discriminator's real and fake outputs side by side
 * (mnist_dcgan.py:152,160) and errD = errD_real + errD_fake (:163).  loss3 = {BCE(p[:n], target0), BCE(p[n:], target1), their fp32
 * sum}; the first two carry the bits pcg_bce_fwd_bwd gives on each half.  Backward: g0 / g1 / g2 = one-element device cotangents of
 * the three outputs (nullable: 0; all null: 1); dp[h*n + i] = (g_h + g2) / n * dBCE/dp.                                          */
int pcg_bce_pair(const float* p, int64_t n_half, float target0, float target1, const float* g0_dev /*nullable*/,
                 const float* g1_dev /*nullable*/, const float* g2_dev /*nullable*/, float* loss3 /*nullable*/, float* dp /*nullable*/,
                 pcg_stream_t stream);
/* nn.BCEWithLogitsLoss (mean): conditional_counteRGAN/mnist/trainer.py:79,106-107,117              */
int pcg_bce_logits_fwd_bwd(const float* z, float target_const, int64_t n, float grad_scale,
                           const float* grad_out_dev /*nullable*/, float* loss /*nullable*/, float* dz /*nullable*/,
                           pcg_stream_t stream);

/* ---- optimizer -------------------------------------------------------------------------------
 * torch.optim.Adam over one flat fp32 parameter buffer: mnist_dcgan.py:126-127,164,175;
 * mnist/trainer.py:77-78,112,123.  [torch] non-amsgrad, eps outside the sqrt of the bias-corrected
 * second moment: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  `step` is the 1-based step count.   */
/* Hyper-parameters are double, as in Python: 1-beta1, 1-beta2 and the bias corrections are formed in double and
 * rounded to fp32 once (that is what torch does). */
int pcg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled_wd,
                  int64_t step, pcg_stream_t stream);

/* hipGraph-capturable form: the step count lives on the device (*step_counter_dev is incremented by the
 * call) and the bias corrections are computed there in fp64, inside the update kernel itself: its last block to
 * finish stores the new count and the corrections of the next step.  hyper_scratch2_dev is 48 bytes (8-byte aligned)
 * the caller zero-initialises ONCE and then leaves alone. */
int pcg_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled_wd,
                             int64_t* step_counter_dev, float* hyper_scratch2_dev, pcg_stream_t stream);

/* ---- CounteRGAN step: non-convolution pieces (conditional_counteRGAN/mnist) -------------------------
 * nn.Embedding(num_classes, H*W) lookup + torch.cat along channels (models/generator.py:73-74,
 * models/discriminator.py:35-36): out[b][p] = (x[b][p], table[idx[b]][p] [, mask[b][p]]) — exact copies.
 * bwd: dtable[k][p] (+)= sum_{b: idx[b]==k, ascending b} dinp[b][p][1];  dx[b][p] = dinp[b][p][0] (nullable).   */
int pcg_embed_concat_fwd(const float* x, const int64_t* idx, const float* table, const float* mask /*C==3*/,
                         float* out, int32_t B, int32_t HW, int32_t C, int32_t K, pcg_stream_t stream);
int pcg_embed_concat_bwd(const float* dinp, const int64_t* idx, float* dtable /*nullable*/, float* dx /*nullable*/,
                         int32_t B, int32_t HW, int32_t C, int32_t K, int accumulate, pcg_stream_t stream);
/* r04: the table gradient alone from channel `ch` of a [B][HW][C] gradient, and the [rows] column `ch` of a [rows][C] tensor — together
 * they let conv_in's grad-input be computed for its label-map channel only (the image and mask channels' gradients are read by nobody). */
int pcg_embed_table_grad(const float* dinp, const int64_t* idx, float* dtable, int32_t B, int32_t HW, int32_t C, int32_t ch, int32_t K,
                         int accumulate, pcg_stream_t stream);
int pcg_gather_channel(const float* src, float* dst, int32_t rows, int32_t C, int32_t ch, pcg_stream_t stream);
/* out = a*x + b*y (y nullable): residual-path gradient sums */
int pcg_axpby(float* out, float a, const float* x, float b, const float* y, int64_t n, pcg_stream_t stream);
/* raw = scale*c ; masked = raw*mask (generator.py:80-82);  bwd: dc = scale*(d_raw + d_masked*mask) */
int pcg_scale_mask_fwd(const float* c, const float* mask, float scale, float* raw, float* masked, int64_t n, pcg_stream_t stream);
int pcg_scale_mask_bwd(const float* d_raw /*nullable*/, const float* d_masked /*nullable*/, const float* mask, float scale,
                       float* dc, int64_t n, pcg_stream_t stream);
/* x_cf = clamp(x + r, lo, hi) (trainer.py:97); bwd: dr = dy on lo <= x+r <= hi, else 0 ([torch] clamp) */
int pcg_clamp_add_fwd(const float* x, const float* r, float lo, float hi, float* y, int64_t n, pcg_stream_t stream);
int pcg_clamp_add_bwd(const float* dy, const float* x, const float* r, float lo, float hi, float* dr, int64_t n, pcg_stream_t stream);
/* out[0] = mean |a*w|, w = m, 1-m (one_minus_m) or 1 (m NULL): trainer.py:99,119; bwd da (+)= g*sign(a*w)*w/n */
size_t pcg_abs_mean_workspace_bytes(void);
int pcg_abs_mean_fwd(const float* a, const float* m, int one_minus_m, int64_t n, float* out, void* workspace,
                     size_t workspace_bytes, pcg_stream_t stream);
int pcg_abs_mean_bwd(const float* a, const float* m, int one_minus_m, int64_t n, const float* grad_out_dev /*nullable*/,
                     float grad_scale, float* da, int accumulate, pcg_stream_t stream);
/* nn.AdaptiveAvgPool2d(1) (discriminator.py:26): y[b][c] = mean over the HW pixels of x[b][p][c] */
int pcg_avgpool_fwd(const float* x, float* y, int32_t B, int32_t HW, int32_t C, pcg_stream_t stream);
int pcg_avgpool_bwd(const float* dy, float* dx, int32_t B, int32_t HW, int32_t C, pcg_stream_t stream);
/* nn.CrossEntropyLoss (mean) on logits [B][K] with int64 targets: trainer.py:80,118 */
int pcg_cross_entropy_fwd_bwd(const float* logits, const int64_t* target, int32_t B, int32_t K, float grad_scale,
                              const float* grad_out_dev /*nullable*/, float* loss /*nullable*/, float* dlogits /*nullable*/,
                              pcg_stream_t stream);

/* ---- device-side batch synthesis (counter-based Philox-4x32-10; deterministic in (seed, offset)) ---------------
 * Replaces the per-iteration host draws of the training loops: build_mask (trainer.py:45-72: per sample choose
 * `num_selected` of the (H/patch)x(W/patch) patches, nearest-upsample to HxW), torch.randint targets (trainer.py:94;
 * `exclude` != NULL applies house_sales trainer.py:248-249's rule: draw k uniformly over [low, high) and map a collision
 * k == exclude[i] to low + (k - low + 1) % (high - low), i.e. the next class gets probability 2/K, the others 1/K) and
 * torch.randn latent noise (mnist_dcgan.py:156).  Streams differ from torch's generators by design. */
int pcg_patch_mask(float* out /*[B][H][W]*/, int32_t B, int32_t H, int32_t W, int32_t patch_size, int32_t num_selected,
                   uint64_t seed, uint64_t offset, pcg_stream_t stream);
int pcg_randint(int64_t* out, int64_t n, int32_t low, int32_t high /*exclusive*/, const int64_t* exclude /*nullable*/,
                uint64_t seed, uint64_t offset, pcg_stream_t stream);
int pcg_randn(float* out, int64_t n, float mean, float std, uint64_t seed, uint64_t offset, pcg_stream_t stream);
/* Gumbel(0,1) noise  -log(-log(u))  — the draw inside F.gumbel_softmax (models/generator.py:90 of house_sales_kc_usa) */
int pcg_rand_gumbel(float* out, int64_t n, uint64_t seed, uint64_t offset, pcg_stream_t stream);
/* Bernoulli(1/2) feature mask [B][D] with the listed columns forced to 0 — house_sales_kc_usa/trainer.py:253-255 */
int pcg_feature_mask(float* out, int32_t B, int32_t D, const int32_t* zero_cols /*device, nullable*/, int32_t n_zero_cols,
                     uint64_t seed, uint64_t offset, pcg_stream_t stream);

/* ---- helpers ---------------------------------------------------------------------------------- */
int pcg_fill(float* p, int64_t n, float value, pcg_stream_t stream);
/* x[r][c] += bias[c] in place — the bias of a ConvTranspose2d that is run as one plain GEMM (a 1x1 input: all KH*KW output
 * positions of a channel share the bias; mnist_dcgan.py:76, mnist_wgan_conditional.py:61) */
int pcg_add_bias_rows(float* x, int64_t rows, int32_t C, const float* bias, pcg_stream_t stream);
/* out[0] (+)= sum p[i]^2   (grad_norm diagnostic: mnist/trainer.py:41-42) */
int pcg_sumsq(const float* p, int64_t n, float* out, int accumulate, pcg_stream_t stream);
/* out[0] = sum over segments s of || flat[seg[2s] .. seg[2s] + seg[2s+1]) ||_2 — the grad_norm diagnostic of
 * house_sales_kc_usa/trainer.py:182-183 (a SUM of per-parameter norms) in one launch over the net's flat gradient buffer;
 * seg_dev: int64 [nseg][2] = (offset, numel) of every parameter, on the device. */
int pcg_norm_sum(const float* flat, const int64_t* seg_dev, int32_t nseg, float* out, pcg_stream_t stream);

/* ---- tabular CounteRGAN (conditional_counteRGAN/house_sales_kc_usa), SURVEY.md section 8a row a15 -------------------
 * All operands are dense row-major fp32 [rows][features]; the layer widths (38, 32, 21, 17, 10, 9, 30, 6, 2, 5, 13, 1)
 * are too small and too ragged for the MFMA tiles, so these are bounds-checked VALU kernels.
 *
 * pcg_gemm: C[M][N] (+)= opA[M][K] . opB[K][N] (+ bias[N]); opA = A or A^T, opB = B or B^T (ld* = row stride of the
 * stored matrix).  nn.Linear forward  y = x W^T + b  -> (0,1, B,out,in, x, W, y, bias)
 *                  input gradient     dx = dy W       -> (0,0, B,in,out, dy, W, dx)
 *                  weight gradient    dW += dy^T x    -> (1,0, out,in,B, dy, x, dW, accumulate=1)
 * replaces the nn.Linear calls of models/generator.py:11-12,22-25,47,54-60, models/discriminator.py:9-15 and
 * models/nn_classifier.py:8-27. */
int pcg_gemm(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
             float* C, int32_t ldc, const float* bias, int accumulate, pcg_stream_t stream);
/* pcg_gemm with the layer's ReLU / LeakyReLU fused into the output write (Linear + LeakyReLU: discriminator.py:9-14, nn_classifier.py:8-25) */
int pcg_gemm_act(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                 float* C, int32_t ldc, const float* bias, int accumulate, int act, float slope, pcg_stream_t stream);
/* nn.Linear weight + bias gradient in one launch, reducing over the batch with a deterministic split over row slabs:
 * dW[O][I] (+)= dy^T x, db[O] (+)= column sums of dy (db nullable).  dy / x may be column slices (ld = row stride).
 * tickets: caller-owned int32[pcg_linear_wgrad_ticket_count()], zero before first use; the kernel leaves it zero. */
size_t pcg_linear_wgrad_workspace_bytes(int32_t B, int32_t O, int32_t I);
int32_t pcg_linear_wgrad_ticket_count(void);
int pcg_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int32_t B, int32_t O, int32_t I, float* dW, float* db,
                     int accumulate_w, int accumulate_b, void* workspace, size_t workspace_bytes, int32_t* tickets, pcg_stream_t stream);
/* The same for up to 40 LAYERS that reduce over the SAME B rows, in one launch on the matrix cores — the 29 Linear layers of the
 * tabular generator's backward.  The library cuts every layer into 32x32 output tiles (at most 64 per call) and the rows into
 * slabs; tile_x / tile_y are reserved (0).  tickets: int32[>= 64], zero before first use, left zero.
 * pcg_linear_wgrad_grouped_slabs: partial results per layer the workspace holds. */
typedef struct pcg_wgrad_item {
  const float* dy; const float* x; float* dW; float* db /*nullable*/;
  int32_t ldy, ldx, O, I, accumulate_w, accumulate_b, tile_x, tile_y;
} pcg_wgrad_item;
int32_t pcg_linear_wgrad_grouped_slabs(int32_t B);
size_t pcg_linear_wgrad_grouped_workspace_bytes(int32_t B, const pcg_wgrad_item* items, int32_t n_items);
int pcg_linear_wgrad_grouped(const pcg_wgrad_item* items, int32_t n_items, int32_t B, void* workspace, size_t workspace_bytes,
                             int32_t* tickets, pcg_stream_t stream);
/* F.one_hot(idx, K).float() — trainer.py:250,290 */
int pcg_onehot(const int64_t* idx, int32_t B, int32_t K, float* out, pcg_stream_t stream);
/* torch.cat([a, b], dim=1) and its backward — generator.py:73-74, discriminator.py:19 */
int pcg_concat_cols(const float* a, int32_t ca, const float* b, int32_t cb, int32_t rows, float* out, pcg_stream_t stream);
int pcg_split_cols(const float* d, int32_t ca, int32_t cb, int32_t rows, float* da /*nullable*/, float* db /*nullable*/,
                   pcg_stream_t stream);
/* FiLM modulation y = gamma*h + beta — generator.py:13-16; backward gives dgamma = dy*h, dh = dy*gamma (dbeta = dy) */
int pcg_film_fwd(const float* gamma, const float* h, const float* beta, float* y, int64_t n, pcg_stream_t stream);
int pcg_film_bwd(const float* dy, const float* gamma, const float* h, float* dgamma, float* dh, int64_t n, pcg_stream_t stream);
/* F.gumbel_softmax(logits, tau, hard=False) over S heads packed side by side in T columns (seg_offsets[S+1], device
 * int32), with the Gumbel(0,1) noise supplied — generator.py:86-90.  y = softmax((logits + noise)/tau) per head;
 * y_hard (nullable) = one-hot of the per-head argmax of y: the forward value of hard=True (eval_utils.py:77), whose
 * straight-through backward is the soft one below. */
int pcg_gumbel_softmax_fwd(const float* logits, const float* noise, const int32_t* seg_offsets, int32_t S, int32_t T, int32_t B,
                           float tau, float* y, float* y_hard /*nullable*/, pcg_stream_t stream);
int pcg_gumbel_softmax_bwd(const float* dy, const float* y, const int32_t* seg_offsets, int32_t S, int32_t T, int32_t B, float tau,
                           float* dlogits, pcg_stream_t stream);
/* residual_full — trainer.py:266-279: column cont_idx[i] = cont[:, i]; categorical column cat_idx[s] =
 * samples[:, head s] . norm_vals[head s] - x[:, cat_idx[s]].  ncont + S must equal D. */
int pcg_assemble_residual_fwd(const float* cont, int32_t ncont, const int32_t* cont_idx, const float* samples,
                              const int32_t* seg_offsets, int32_t S, int32_t T, const int32_t* cat_idx, const float* norm_vals,
                              const float* x, int32_t D, int32_t B, float* residual, pcg_stream_t stream);
int pcg_assemble_residual_bwd(const float* dres, int32_t ncont, const int32_t* cont_idx, const int32_t* seg_offsets, int32_t S,
                              int32_t T, const int32_t* cat_idx, const float* norm_vals, int32_t D, int32_t B, float* dcont,
                              float* dsamples, pcg_stream_t stream);
/* tensor.mean() (Wasserstein critic losses, trainer.py:292,299) and its backward dx = grad_out * grad_scale / n */
size_t pcg_mean_workspace_bytes(void);
int pcg_mean_fwd(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, pcg_stream_t stream);
int pcg_mean_bwd(const float* grad_out_dev /*nullable = 1*/, float grad_scale, int64_t n, float* dx, pcg_stream_t stream);
/* torch.nn.utils.spectral_norm (n_power_iterations = 1) — discriminator.py:9-15.  Forward: when power_iteration != 0
 * (training mode) v <- normalize(W^T u), u <- normalize(W v) in place, then sigma = u.(W v), w_bar = W / sigma.
 * u_used / v_used (nullable): copies of the vectors this forward used — what torch's `u.clone()` keeps for backward,
 * since a later forward overwrites u and v.  Backward (u, v constants): dW (+)= (dw_bar - <dw_bar, w_bar> u v^T) / sigma.
 * Dimensions up to 256. */
int pcg_spectral_norm_fwd(const float* w_orig, int32_t out_features, int32_t in_features, float* u, float* v, float eps,
                          int power_iteration, float* w_bar, float* sigma, float* u_used, float* v_used, pcg_stream_t stream);
/* all (up to 8) spectral-norm layers of a net in one launch — arrays of per-layer arguments, host-side arrays of device pointers */
int pcg_spectral_norm_fwd_batched(int32_t n, const float* const* w_orig, const int32_t* out_features, const int32_t* in_features,
                                  float* const* u, float* const* v, float eps, int power_iteration, float* const* w_bar,
                                  float* const* sigma, float* const* u_used, float* const* v_used, pcg_stream_t stream);
int pcg_spectral_norm_bwd_batched(int32_t n, const float* const* dw_bar, const float* const* w_bar, const int32_t* out_features,
                                  const int32_t* in_features, const float* const* u, const float* const* v, const float* const* sigma,
                                  float* const* dw_orig, const int32_t* accumulate, pcg_stream_t stream);
/* `reps` successive training-mode calls of every layer in one launch (each one power iteration from the previous call's u, v, as the
 * module's forward does: D(real) then D(fake)); the output arrays hold reps * n entries, call-major (call r of layer l: r * n + l).
 * At most 8 layers x calls. */
int pcg_spectral_norm_fwd_batched_reps(int32_t n, int32_t reps, const float* const* w_orig, const int32_t* out_features,
                                       const int32_t* in_features, float* const* u, float* const* v, float eps, int power_iteration,
                                       float* const* w_bar, float* const* sigma, float* const* u_used, float* const* v_used,
                                       pcg_stream_t stream);
/* The backward of `passes` calls of the same n layers, applied one after the other into the same dw_orig[l] (the first writes or
 * accumulates as accumulate[l] says, the others add — what chained launches compute); per-call arrays hold passes * n entries,
 * pass-major.  Then, where db_dst[l] is given, db_dst[l] += db_src[l] over out_features[l] values (the bias gradient of a later pass,
 * reduced into its own buffer because one grouped weight-gradient launch cannot order two writers of the same vector). */
int pcg_spectral_norm_bwd_batched_seq(int32_t n, int32_t passes, const float* const* dw_bar, const float* const* w_bar,
                                      const int32_t* out_features, const int32_t* in_features, const float* const* u, const float* const* v,
                                      const float* const* sigma, float* const* dw_orig, const int32_t* accumulate, float* const* db_dst /*nullable*/,
                                      const float* const* db_src /*nullable*/, pcg_stream_t stream);
int pcg_spectral_norm_bwd(const float* dw_bar, const float* w_bar, int32_t out_features, int32_t in_features, const float* u,
                          const float* v, const float* sigma, float* dw_orig, int accumulate, pcg_stream_t stream);

/* ---- conditional WGAN-GP (conditional_gan/mnist/mnist_wgan_conditional.py), SURVEY.md section 8a row a14 -----------------
 * nn.InstanceNorm2d(C, affine=True) (:88,91,94) on an NHWC activation [B][HW][C]: statistics per (sample, channel) over HW,
 * biased variance, eps inside the square root; `act` fuses the LeakyReLU(0.2) that follows (:89,92,95). */
int pcg_instnorm_fwd(const float* x, int32_t B, int32_t HW, int32_t C, const float* gamma, const float* beta, float eps, int act,
                     float slope, float* y, float* mean /*[B*C]*/, float* invstd /*[B*C]*/, pcg_stream_t stream);
/* backward: dx (nullable); per-sample partial sums dgamma_partial / dbeta_partial [B][C] (nullable; reduce over B with pcg_colsum) */
int pcg_instnorm_bwd(const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean, const float* invstd,
                     const float* gamma, float* dx, float* dgamma_partial, float* dbeta_partial, pcg_stream_t stream);
/* the same with the stage's neighbours folded in (one launch for what autograd runs as LeakyReLU backward, InstanceNorm backward, an
 * add and the conv bias gradient's reduction, :88-95): act_y (nullable) = the LeakyReLU OUTPUT, dy is multiplied by lrelu'(.) first
 * (dn_out, nullable, keeps that masked gradient for the double backward); addend (nullable) is added to the stored dx;
 * dxsum_partial (nullable) [B][C] = sum over HW of the stored dx (the conv bias gradient's per-sample partial).                  */
int pcg_instnorm_bwd_fused(const float* dy, const float* act_y, float neg_slope, const float* x, int32_t B, int32_t HW, int32_t C,
                           const float* mean, const float* invstd, const float* gamma, float* dn_out, const float* addend, float* dx,
                           float* dgamma_partial, float* dbeta_partial, float* dxsum_partial, pcg_stream_t stream);
/* sums over the rows of up to three [rows][C] partial arrays (the three above) in one launch; acc_k: dst_k += instead of =        */
int pcg_rowsum3(int32_t n, const float* src0, float* dst0, int acc0, const float* src1, float* dst1, int acc1, const float* src2,
                float* dst2, int acc2, int32_t rows, int32_t C, pcg_stream_t stream);
/* backward of pcg_instnorm_bwd — what autograd.grad(..., create_graph=True) + critic_loss.backward() (:149,154) need: with r
 * the cotangent on dx, returns the cotangents reaching dy (ddy), x (ez) and gamma (per-sample partials); any may be NULL. */
int pcg_instnorm_bwd_bwd(const float* r, const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean,
                         const float* invstd, const float* gamma, float* ddy, float* ez, float* dgamma_partial, pcg_stream_t stream);
/* ... with the cotangent ddy continuing through the stage's LeakyReLU (act_y = its output; nullable) in the same launch          */
int pcg_instnorm_bwd_bwd_act(const float* r, const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean,
                             const float* invstd, const float* gamma, const float* act_y, float neg_slope, float* ddy, float* ez,
                             float* dgamma_partial, pcg_stream_t stream);
/* nn.Flatten of an NCHW tensor (:96) from the NHWC activation, flat[b][c*HW + p] = act[b][p][c]; inverse != 0: the other way */
int pcg_nhwc_to_nchw_flat(const float* src, float* dst, int32_t B, int32_t HW, int32_t C, int inverse, pcg_stream_t stream);
/* x3 = [real | fake | alpha*real + (1-alpha)*fake] (3B samples): the input of the batched critic pass in one launch (r04) */
int pcg_interpolate_stack(const float* alpha, const float* real, const float* fake, float* x3, int32_t B, int32_t per_sample,
                          pcg_stream_t stream);
/* interpolates = alpha*real + (1-alpha)*fake, alpha per sample (:147) */
int pcg_interpolate(const float* alpha, const float* real, const float* fake, float* out, int32_t B, int32_t per_sample,
                    pcg_stream_t stream);
/* gradient penalty lambda * mean_b (||g_b||_2 - 1)^2 (:150): forward keeps the per-sample norms for the backward,
 * dg = grad_out * 2*lambda/B * (||g_b|| - 1) * g_b / ||g_b|| */
int pcg_gradient_penalty_fwd(const float* grads, int32_t B, int32_t per_sample, float lambda, float* norms /*[B]*/, float* penalty /*[1]*/,
                             pcg_stream_t stream);
int pcg_gradient_penalty_bwd(const float* grads, const float* norms, const float* grad_out_dev /*nullable = 1*/, int32_t B,
                             int32_t per_sample, float lambda, float* dgrads, pcg_stream_t stream);
/* torch.rand: uniform [0, 1) — the interpolation coefficients alpha (:146) */
int pcg_rand_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, pcg_stream_t stream);

/* ---- counterfactual evaluation -----------------------------------------------------------------
 * out[0] = class-flip rate = mean_b [argmax logits_cf[b] == target[b]];
 * out[1] = prediction gain = mean_b (softmax(logits_cf[b])[target[b]] - q_b) with q_b = softmax(logits_ref[b])[target[b]]
 * (house_sales_kc_usa/eval_utils.py:246-258) or, when logits_ref is NULL, softmax(logits_cf[b])[other[b]]
 * (mnist/eval_utils.py:61-64).  Actionability (mean |residual|) is pcg_abs_mean_fwd. */
int pcg_cf_metrics(const float* logits_cf, const float* logits_ref /*nullable*/, const int64_t* target, const int64_t* other /*nullable*/,
                   int32_t B, int32_t K, float* out /*[2]*/, pcg_stream_t stream);

/* ---- classifier pre-training -----------------------------------------------------------------
 * nn.CrossEntropyLoss(weight=class_weight), mean reduction = weighted sum / sum of the weights of the targets
 * (house_sales_kc_usa/trainer.py:55-57) */
int pcg_cross_entropy_weighted_fwd_bwd(const float* logits, const int64_t* target, const float* class_weight, int32_t B, int32_t K,
                                       float grad_scale, const float* grad_out_dev /*nullable*/, float* loss /*nullable*/,
                                       float* dlogits /*nullable*/, pcg_stream_t stream);
/* nn.Dropout / nn.Dropout2d in training mode, forward and backward (same map): y = x * mask * scale, scale = 1/(1-p).
 * Dropout: inner = 1, mask has n entries; Dropout2d on an NHWC activation [B][HW][C]: inner = HW, mask is [B][C]
 * (mnist/models/classifier.py:13,19; house_sales_kc_usa/models/nn_classifier.py:11,16,22).  The 0/1 mask is an input:
 * pcg_rand_bernoulli draws it on the device (keep probability 1-p); parity runs pass the reference's draws. */
int pcg_dropout_apply(const float* x, const float* mask, int64_t n, int32_t inner, int32_t C, float scale, float* y, pcg_stream_t stream);
int pcg_rand_bernoulli(float* out, int64_t n, float keep_prob, uint64_t seed, uint64_t offset, pcg_stream_t stream);

/* ---- input pipeline -------------------------------------------------------------------------
 * transforms.Resize -> ToTensor -> Normalize (dconv_gan/mnist/mnist_dcgan.py:42-46) for a batch of 8-bit one-channel images:
 * Pillow's separable fixed-point bilinear resize (bounds[2*i] = first source index, bounds[2*i+1] = taps, coefficients with 22
 * fractional bits, ksize per output coordinate — built by the host as Pillow's precompute_coeffs does), rounding to uint8
 * after each pass, then float32 v/255 and (t - mean)/std.  dst: [N][OH][OW] float32 (= [N,1,OH,OW]). */
int pcg_resize8_normalize(const uint8_t* src, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, const int32_t* x_bounds,
                          const int32_t* x_coeffs, int32_t x_ksize, const int32_t* y_bounds, const int32_t* y_coeffs, int32_t y_ksize,
                          float mean, float stdv, float* dst, pcg_stream_t stream);

/* ---- fused tabular generator (house_sales_kc_usa/models/generator.py:38-92) -------------------------------------------------
 * The whole ResidualGenerator forward (fc_in, 5 x [fc1, BatchNorm1d, FiLM, ReLU, fc2, BatchNorm1d, FiLM, residual add], the
 * continuous head and the packed Gumbel-softmax heads) as 11 launches and its backward (down to the pre-activation gradients
 * every Linear's weight gradient needs) as 11 launches: one thread per batch row, kernels cut at the BatchNorm statistics.
 * Built for hidden width 32 and 5 blocks (the reference's configuration, config.py:15-17); all buffers are the caller's.
 * Offsets are element offsets into ONE flat fp32 parameter buffer (and the gradient buffer of the same layout). */
#define PCG_HOUSE_ROWS_PER_BLOCK 64
typedef struct pcg_house_g_desc {
  int32_t fc_in_w, fc_in_b;
  int32_t fc1_w[5], fc1_b[5], bn1_g[5], bn1_b[5], fc2_w[5], fc2_b[5], bn2_g[5], bn2_b[5];
  int32_t film_gamma_w[5], film_gamma_b[5], film_beta_w[5], film_beta_b[5];
  int32_t cont_w, cont_b;
  int32_t head_w[8], head_b[8], seg[9];      /* categorical heads and their packed column offsets (seg[nheads] = T) */
  int32_t nheads, ncont, D, NC;              /* D = input_dim, NC = num_classes */
  int32_t hidden, nblocks;                   /* must be 32 and 5 */
} pcg_house_g_desc;

typedef struct pcg_house_g_fwd_args {
  const float* params;
  const float *x, *onehot, *mask, *noise;    /* [B][D], [B][NC], [B][D], Gumbel noise [B][T] */
  float* inp;                                /* out [B][D+NC+D]: (x, onehot, mask) — the cond operand of later weight gradients */
  float *H, *Z1, *Z2;                        /* out [6][B][32], [5][B][32], [5][B][32]: block inputs and pre-BatchNorm activations */
  float* P;                                  /* scratch [10][ceil(B/64)][2][32] partial statistics */
  float* SM;                                 /* out [10][2][32] saved mean / invstd (layer order bn1_0, bn2_0, bn1_1, ...) */
  float* running_mean[10]; float* running_var[10]; int64_t* num_batches_tracked[10];   /* nullable: BatchNorm buffers to update */
  float *cont, *logits, *soft, *hard;        /* out [B][ncont], [B][T], [B][T], nullable [B][T] one-hot of the soft arg-max */
  int32_t B;
  float eps, momentum, tau, res_scale;
} pcg_house_g_fwd_args;

typedef struct pcg_house_g_bwd_args {
  const float* params; float* grads;         /* BatchNorm gamma / beta gradients are written into `grads` (accumulate flag) */
  const float *onehot, *mask;
  const float *H, *Z1, *Z2, *SM, *soft;      /* from the forward */
  const float *d_cont, *d_logits, *d_samples;/* nullable cotangents of the three outputs */
  float *DH, *DZ1, *DZ2, *A1;                /* out [5][B][32] each: block-top gradients, pre-BN gradients (weight-gradient operands), ReLU outputs */
  float* DN1;                                /* scratch [B][32] */
  float *DG, *DB;                            /* out [5][B][32]: gradients at the FiLM gamma / beta Linear outputs */
  float* DZIN;                               /* out [B][32]: gradient at fc_in's output */
  float *DL, *DC;                            /* out [B][T], [B][ncont]: gradients at the head outputs */
  float* Q;                                  /* scratch [9 * ceil(B/64) + ceil(B/16)][2][32]: per-block partial sums of the ten BatchNorm backwards
                                                (64-row blocks; the last bn2's come from the 16-row head kernel) */
  int32_t B, accumulate;
  float tau, res_scale;
} pcg_house_g_bwd_args;

int pcg_house_g_fwd(const pcg_house_g_desc* desc, const pcg_house_g_fwd_args* args, pcg_stream_t stream);
int pcg_house_g_bwd(const pcg_house_g_desc* desc, const pcg_house_g_bwd_args* args, pcg_stream_t stream);

/* Loss composition on the device: total = sum_i w_i * term_i for up to 8 one-element loss tensors, e.g.
 * G_loss = G_adv + lambda_cls*G_cls + lambda_reg*G_reg + lambda_mask*mask_pen (house_sales_kc_usa/trainer.py:307-312); the backward
 * hands every term its w_i * grad_out.  One launch each instead of a dozen scalar tensor-op kernels. */
int pcg_weighted_sum_fwd(int32_t n, const float* const* terms, const float* weights, float* out, pcg_stream_t stream);
int pcg_weighted_sum_bwd(int32_t n, const float* weights, const float* grad_out_dev /*nullable = 1*/, float* const* grads /*entries nullable*/,
                         pcg_stream_t stream);

/* The tabular spectral-norm critic (house_sales_kc_usa/models/discriminator.py:5-20) as one forward and one backward launch, one
 * thread per row; w_bar[l] / bias[l]: the four layers' normalised weights (from pcg_spectral_norm_fwd_batched) and biases.
 * Forward writes the concatenated input a0 [B][21] and the post-LeakyReLU activations a1 [B][32], a2 [B][64], a3 [B][128];
 * backward writes the pre-activation gradients d3, d2, d1 (the dy operands of the weight gradients; layer 4's is dout itself)
 * and, if dx != NULL, the gradient of the first D input columns.  Built for input_dim + num_classes = 21, hidden width 32. */
int pcg_house_critic_fwd(const float* x, const float* onehot, int32_t B, int32_t D, int32_t NC, const float* const* w_bar,
                         const float* const* bias, float slope, float* a0, float* a1, float* a2, float* a3, float* out, pcg_stream_t stream);
int pcg_house_critic_bwd(const float* dout, int32_t B, int32_t D, const float* const* w_bar, float slope, const float* a1, const float* a2,
                         const float* a3, float* d3, float* d2, float* d1, float* dx /*nullable*/, pcg_stream_t stream);
/* The same for n_pass (1 or 2) independent passes in ONE launch each way — D(real) and D(fake) of the critic step (trainer.py:290-291),
 * which share the module but not the spectral-norm weights (two successive power iterations): every per-pass pointer becomes an
 * array of n_pass, w_bar has n_pass * 4 entries (pass-major).  Per pass the arithmetic is that of the one-pass calls. */
int pcg_house_critic_fwd_n(int32_t n_pass, const float* const* x, const float* const* onehot, int32_t B, int32_t D, int32_t NC,
                           const float* const* w_bar, const float* const* bias, float slope, float* const* a0, float* const* a1,
                           float* const* a2, float* const* a3, float* const* out, pcg_stream_t stream);
int pcg_house_critic_bwd_n(int32_t n_pass, const float* const* dout, int32_t B, int32_t D, const float* const* w_bar, float slope,
                           const float* const* a1, const float* const* a2, const float* const* a3, float* const* d3, float* const* d2,
                           float* const* d1, float* const* dx /*entries nullable*/, pcg_stream_t stream);

/* The frozen tabular classifier of the counterfactual loss (house_sales_kc_usa/models/nn_classifier.py:4-32 in eval mode, each
 * BatchNorm1d folded into the following Linear by the caller: Linear 17->256, 256->256, 256->128, 128->64 + LeakyReLU(0.1), Linear
 * 64->4) as one launch each way on the matrix cores, 16 rows per block, activations in LDS between layers.
 *   forward:  w_kmajor[l], l = 0..3: the folded weight TRANSPOSED, [K_l][N_l] row-major, layer 0 zero-padded to K = 20;
 *             w_kmajor[4]: the last layer as stored [4][64]; bias[l]; writes the post-activation outputs a1 [B][256], a2 [B][256],
 *             a3 [B][128], a4 [B][64] (the backward's masks) and logits [B][4].
 *   backward: w_stored[l]: the folded weights as stored [N_l][K_l]; dx [B][17] = d(logits . dlogits)/dx.  (trainer.py:301-302;
 *             the parameters are frozen, main.py:27-30: no weight gradients.) */
int pcg_house_classifier_fwd(const float* x, int32_t B, const float* const* w_kmajor, const float* const* bias, float* a1, float* a2,
                             float* a3, float* a4, float* logits, pcg_stream_t stream);
int pcg_house_classifier_bwd(const float* dlogits, int32_t B, const float* const* w_stored, const float* a1, const float* a2, const float* a3,
                             const float* a4, float* dx, pcg_stream_t stream);

/* The scalars the tabular trainer logs per step (house_sales_kc_usa/trainer.py:292, :299, :307-312, :318-330) in one launch:
 * out5 = { D_loss = mean(d_fake) - mean(d_real), G_loss = -mean(d_fake_g) + lambda_cls*g_cls + w_reg*am + lambda_mask*pen,
 *          g_adv = -mean(d_fake_g), g_reg = w_reg_log*am, mean(d_fake_g) } — the same reduction trees and fma chains as
 * pcg_mean_fwd + pcg_weighted_sum_fwd (bit-identical values), n <= 16384 critic outputs. */
int pcg_house_losses(const float* d_real, const float* d_fake, const float* d_fake_g, int64_t n, const float* g_cls, const float* am,
                     const float* pen, float lambda_cls, float w_reg, float lambda_mask, float w_reg_log, float* out5, pcg_stream_t stream);

/* The residual block of the tabular step (house_sales_kc_usa/trainer.py:266-287, :305) in one launch each way.
 * Forward: residual_full (as pcg_assemble_residual_fwd), masked = residual_full * mask, x_cf = x + masked, and the penalties
 * pen = mean|residual_full * (1 - mask)|, am = mean|masked| — bit-identical to pcg_assemble_residual_fwd + pcg_scale_mask_fwd +
 * pcg_axpby + 2 x pcg_abs_mean_fwd (same element arithmetic, same partition and trees of the sums).  col_src (HOST int32[D]):
 * >= 0 the index of a continuous column in cont, -(s + 1) categorical head s.  partial512: 512 floats of scratch; ticket: one
 * int32, zero before first use, left zero.
 * Backward: the gradient of  lambda_mask * pen + w_reg * am + <gx_a + gx_b, x_cf>  with respect to (cont, samples) — w_pen =
 * lambda_mask, w_am = lambda_reg * D as the trainer weighs them; gx_a, gx_b: the two addends of dLoss/dx_cf (critic, classifier).
 * Bit-identical to the chain pcg_axpby, pcg_weighted_sum_bwd, 2 x pcg_abs_mean_bwd, pcg_axpby, pcg_scale_mask_bwd, add,
 * pcg_assemble_residual_bwd. */
int pcg_house_residual_fwd(const float* cont, int32_t ncont, const float* samples, const int32_t* seg_dev, int32_t T, const float* norm,
                           const float* x, const float* mask, const int32_t* col_src, int32_t D, int32_t B, float* res, float* masked,
                           float* x_cf, float* partial512, int32_t* ticket, float* pen_out, float* am_out, pcg_stream_t stream);
int pcg_house_residual_bwd(const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b, float w_pen,
                           float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev, int32_t S, int32_t T,
                           const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B, float* dcont, float* dsamples,
                           pcg_stream_t stream);
/* "Riders": two launches of the tabular step that do not depend on each other issued as ONE launch whose blocks split between the
 * two kernel bodies (a HIP graph with parallel branches is launched node by node by the host; one launch is not).  Same bodies, same
 * bits as the separate calls.
 *   pcg_house_residual_fwd_sn      = pcg_house_residual_fwd + pcg_spectral_norm_fwd_batched_reps (training mode: the critic step's
 *                                    power iterations only need the critic's weights; trainer.py:266-287 beside models/discriminator.py:9-16)
 *   pcg_house_residual_bwd_losses  = pcg_house_residual_bwd + pcg_house_losses (the logged scalars do not feed the backward;
 *                                    trainer.py:314 beside :292, :307-312)
 *   pcg_house_classifier_fwd_snbwd = pcg_house_classifier_fwd + pcg_spectral_norm_bwd_batched_seq, and
 *   pcg_house_classifier_bwd_snfwd = pcg_house_classifier_bwd + pcg_spectral_norm_fwd_batched_reps (training mode): the frozen
 *                                    classifier's term (trainer.py:301-302) does not depend on the critic update (:290-295), so its
 *                                    two launches carry the critic's spectral-norm work that is on the chain at the same time */
int pcg_house_residual_fwd_sn(const float* cont, int32_t ncont, const float* samples, const int32_t* seg_dev, int32_t T, const float* norm,
                              const float* x, const float* mask, const int32_t* col_src, int32_t D, int32_t B, float* res, float* masked,
                              float* x_cf, float* partial512, int32_t* ticket, float* pen_out, float* am_out,
                              int32_t n_layers, int32_t reps, const float* const* w_orig, const int32_t* out_features,
                              const int32_t* in_features, float* const* u, float* const* v, float eps, float* const* w_bar,
                              float* const* sigma, float* const* u_used, float* const* v_used, pcg_stream_t stream);
int pcg_house_residual_bwd_losses(const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b,
                                  float w_pen, float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev, int32_t S,
                                  int32_t T, const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B, float* dcont,
                                  float* dsamples, const float* d_real, const float* d_fake, const float* d_fake_g, int32_t n,
                                  const float* g_cls, const float* am, const float* pen, float lambda_cls, float w_reg, float lambda_mask,
                                  float w_reg_log, const float* ce_row_loss, int32_t n_ce, float* out6, pcg_stream_t stream);
/* ce_target != NULL: the forward also evaluates the cross-entropy of its logits against ce_target (trainer.py:302) — per row the
 * term lse - z[target] into ce_row_loss[B] and ce_grad_scale / B * (softmax - onehot) into ce_dlogits[B][4], the expressions of
 * pcg_cross_entropy_fwd_bwd; pcg_house_residual_bwd_losses(ce_row_loss, n_ce = B) then forms the mean in that kernel's summation
 * order (out6[5], and uses it for G_loss instead of *g_cls): the same bits as the separate cross-entropy launch. */
int pcg_house_classifier_fwd_snbwd(const float* x, int32_t B, const float* const* w_kmajor, const float* const* bias, float* a1, float* a2,
                                   float* a3, float* a4, float* logits, int32_t n, int32_t passes, const float* const* dw_bar,
                                   const float* const* w_bar, const int32_t* out_features, const int32_t* in_features,
                                   const float* const* u, const float* const* v, const float* const* sigma, float* const* dw_orig,
                                   const int32_t* accumulate, float* const* db_dst, const float* const* db_src,
                                   const int64_t* ce_target, float ce_grad_scale, float* ce_dlogits, float* ce_row_loss, pcg_stream_t stream);
int pcg_house_classifier_bwd_snfwd(const float* dlogits, int32_t B, const float* const* w_stored, const float* a1, const float* a2,
                                   const float* a3, const float* a4, float* dx, int32_t n, int32_t reps, const float* const* w_orig,
                                   const int32_t* out_features, const int32_t* in_features, float* const* u, float* const* v, float eps,
                                   float* const* w_bar, float* const* sigma, float* const* u_used, float* const* v_used,
                                   pcg_stream_t stream);
/* The three per-iteration draws of the tabular trainer in one launch — target class != y (trainer.py:248-249, as pcg_randint with
 * exclude), feature mask (:253-255, as pcg_feature_mask), Gumbel noise [B][T] (generator.py:90, as pcg_rand_gumbel) — each from its
 * own counter offset: the values the three separate calls produce.  onehot_target / onehot_y (nullable, [B][num_classes]): the float
 * one-hot rows of the drawn targets and of y (trainer.py:250, :290), written by the same launch. */
int pcg_house_draws(int64_t* target_y, int32_t B, int32_t num_classes, const int64_t* y, uint64_t offset_target, float* mask, int32_t D,
                    const int32_t* zero_cols, int32_t n_zero_cols, uint64_t offset_mask, float* noise, int32_t T, uint64_t offset_noise,
                    uint64_t seed, float* onehot_target, float* onehot_y, pcg_stream_t stream);
/* The same launch with the Philox offsets taken from a DEVICE counter: counter[0] = the running offset (the three draws take
 * consecutive ranges of (B+3)/4, (B*D+3)/4, (B*T+3)/4 counter values from it, as the host-side bookkeeping of the call above hands
 * them out), counter[1] = a ticket (zero before first use, left zero); the launch advances counter[0] itself.  Captured in a HIP
 * graph it draws fresh numbers on every replay — the values successive pcg_house_draws calls would produce. */
int pcg_house_draws_counter(int64_t* target_y, int32_t B, int32_t num_classes, const int64_t* y, float* mask, int32_t D,
                            const int32_t* zero_cols, int32_t n_zero_cols, float* noise, int32_t T, uint64_t seed, float* onehot_target,
                            float* onehot_y, uint64_t* counter, pcg_stream_t stream);
/* pcg_house_draws_counter + the BATCH: the training set X [n_rows][D] / Y [n_rows] and the epoch's permutation perm [n_perm] are
 * resident in HBM; the launch copies rows perm[cursor .. cursor+B) into x_out / y_out (the static inputs of the captured step),
 * writes the source rows to src_out (nullable) and draws target / mask / noise / one-hot rows for them.  counter: uint64[4] =
 * [Philox offset, ticket, row cursor, unused]; the launch advances offset and cursor (+= B) itself, so a replayed graph walks
 * through the epoch with no host-side copy — DataLoader(shuffle=True, drop_last=True) of house_sales_kc_usa/trainer.py:198. */
int pcg_house_batch_draws_counter(int64_t* target_y, int32_t B, int32_t num_classes, const float* X, const int64_t* Y,
                                  const int64_t* perm, int64_t n_perm, int64_t n_rows, float* x_out, int64_t* y_out,
                                  int64_t* src_out /*nullable*/, float* mask, int32_t D, const int32_t* zero_cols, int32_t n_zero_cols,
                                  float* noise, int32_t T, uint64_t seed, float* onehot_target /*nullable*/, float* onehot_y /*nullable*/,
                                  uint64_t* counter, pcg_stream_t stream);
/* The four per-iteration diagnostics of house_sales_kc_usa/trainer.py:318-343 in one single-block launch:
 *   out4 = { pred_gain, sparsity (|masked residual| > eps), reg_loss_l2, class_flip_rate }
 * logits_orig: the frozen classifier's logits of the ORIGINAL rows (it does not change during GAN training: evaluated once for the
 * training set), gathered through src_rows (nullable: row b).  acc (nullable, double[8]): epoch accumulators —
 * acc[2..5] += out4 (pcg_house_residual_bwd_losses_diag also adds D_loss, G_loss and the iteration count to acc[0], acc[1], acc[6]),
 * read once per epoch instead of six .item() calls per iteration (trainer.py:327-353).                                          */
int pcg_house_diag(const float* logits_cf, const float* logits_orig, const int64_t* src_rows /*nullable*/, const int64_t* target_y,
                   const float* masked, int32_t B, int32_t nc, int32_t D, float eps, float* out4, double* acc /*nullable*/,
                   pcg_stream_t stream);
/* pcg_house_residual_bwd_losses with the diagnostics riding in the same launch (one more block) and the epoch accumulators. */
int pcg_house_residual_bwd_losses_diag(const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b,
                                       float w_pen, float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev,
                                       int32_t S, int32_t T, const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B,
                                       float* dcont, float* dsamples, const float* d_real, const float* d_fake, const float* d_fake_g,
                                       int32_t n, const float* g_cls, const float* am, const float* pen, float lambda_cls, float w_reg,
                                       float lambda_mask, float w_reg_log, const float* ce_row_loss, int32_t n_ce, float* out6,
                                       const float* logits_cf, const float* logits_orig, const int64_t* src_rows /*nullable*/,
                                       const int64_t* target_y, int32_t nc, float eps, float* diag_out4, double* acc /*nullable*/,
                                       pcg_stream_t stream);

/* A barrier of the ranks on the library's own communicator (a 4-byte all-reduce in stream order; synchronise `stream` afterwards):
 * bench.py brackets its timed region with it, so that region uses ONE communicator — the one that carries the gradient exchange.
 * pcg_dp_rccl_version: ncclGetVersion of the RCCL the library bound (22105 = 2.21.5; 0 = unknown).                              */
int pcg_dp_barrier(pcg_stream_t stream);
int32_t pcg_dp_rccl_version(void);

/* ---- data-parallel exchange (RCCL over xGMI) --------------------------------------------------------------------------------
 * The reference is single-process (mnist_dcgan.py:140-175, mnist/trainer.py:89-123); data-parallel replicas add ONE exchange per
 * optimizer step: the average of the net's flat fp32 gradient bucket.  The library owns the RCCL communicator (bound at run time,
 * no link-time dependency), a side HIP stream and the ordering events.  Bootstrap: rank 0 makes an id (pcg_dp_unique_id), the
 * host layer ships the 128 bytes to the other ranks by any means (the Python layer: torch.distributed's store), every rank calls
 * pcg_dp_init on ITS GPU (the current HIP device).  None of these calls may be captured in a HIP graph: the step is captured as
 * graph segments cut at the exchange points (nn.GraphedStep).
 *   pcg_dp_allreduce          buf <- mean over ranks, in stream order on `stream`
 *   pcg_dp_allreduce_begin    the same on the library's side stream, after everything queued on producer_stream so far;
 *                             `slot` (0..7) names the reduction
 *   pcg_dp_side_stream        that stream: the caller may queue work behind the reduction (Adam), then pcg_dp_record(slot)
 *   pcg_dp_allreduce_wait     consumer_stream waits on the GPU for slot (the reduction and whatever pcg_dp_record covered)
 *   pcg_dp_allreduce_sum_f64  sum of doubles in stream order (exact-BatchNorm statistic sums, 2*C values per layer)
 *   pcg_dp_broadcast          bytes from rank `root` (initial weights, BatchNorm buffers)
 *   pcg_dp_sync_batchnorm     exact global-batch BatchNorm (SURVEY.md §8e option ii; the reference's BatchNorm spans the whole batch,
 *                             mnist_dcgan.py:77-87): while enabled, every BatchNorm-family entry point (pcg_bn_train_stats*,
 *                             pcg_conv2d_*_bn*, pcg_bn_act_bwd*, pcg_bn_bwd_partial) sums its per-channel partial rows locally in
 *                             the usual fixed order, all-reduces the 2*C fp64 sums (sum x, sum x^2 | sum dy, sum dy*xhat) over the
 *                             ranks in stream order, and finalises with rows*world rows — N ranks x B/N images compute what one
 *                             process computes on B (equal shards assumed).  dgamma / dbeta stay local sums (they are averaged with
 *                             the gradient bucket).  These calls then contain an RCCL collective: do not capture them in a graph. */
#define PCG_DP_UNIQUE_ID_BYTES 128
int pcg_dp_unique_id(void* id_out /*[PCG_DP_UNIQUE_ID_BYTES]*/);
int pcg_dp_init(const void* id /*[PCG_DP_UNIQUE_ID_BYTES]*/, int32_t rank, int32_t world);
int32_t pcg_dp_world(void);   /* 0 before pcg_dp_init */
int32_t pcg_dp_rank(void);
pcg_stream_t pcg_dp_side_stream(void);
int pcg_dp_allreduce(float* buf, int64_t n, pcg_stream_t stream);
int pcg_dp_allreduce_begin(float* buf, int64_t n, int32_t slot, pcg_stream_t producer_stream);
int pcg_dp_record(int32_t slot);
int pcg_dp_allreduce_wait(int32_t slot, pcg_stream_t consumer_stream);
int pcg_dp_allreduce_sum_f64(double* buf, int64_t n, pcg_stream_t stream);
int pcg_dp_broadcast(void* buf, int64_t nbytes, int32_t root, pcg_stream_t stream);
int pcg_dp_sync_batchnorm(int32_t enable);
int pcg_dp_shutdown(void);

/* ---- tuning switches (diagnostics): A/B of launch-planning choices inside ONE process, e.g. scripts/conv_microbench.py --ab.
 *   "korder"            order of the k-tiles of the forward / grad-input kernels: 0 (tap, channel chunk), 1 L2-friendly (default)
 *   "wgrad_order"       weight-gradient block order: 0 tile-major, 1 slice-major, -1 built-in rule
 *   "dgrad_interleave"  sub-pixel phases of a grad-input tile neighbours in launch order: 0 / 1, -1 built-in rule
 *   "stream_k"          hybrid stream-K launches: 0 off, 1 where the launch plan's model gains > 10 % (default), 2 wherever valid
 *   "sk_blocks"         number of stream-K workgroups (512 / 256 / 128 / 64), -1 built-in choice
 *   "dgrad_gemm"        grad-input of a kernel size that is not a multiple of the stride: 0 sub-pixel-phase kernel, 1 one GEMM +
 *                       col2im, -1 whichever has fewer multiply-adds for the geometry (built-in)
 *   "t64"               forward launches with at most 256 tiles of 128x128: 64x128 tiles instead (default 1), 0 keeps 128x128
 *   "fwd_splits", "persistent", "persist_tiles", "dma"   forward K-slices; experiments that are compiled out of / off in the shipped library
 * value -1 restores the built-in choice.  Results stay correct under every setting (the order of a sum changes, not its terms). */
int pcg_tune_set(const char* name, int32_t value);

/* ---- scratch of the hybrid stream-K launches (conv_igemm.hip: conv_fwd_sk_kernel / conv_dgrad_sk_kernel) --------------------------
 * A forward / grad-input launch whose tile count is not a multiple of the chip's 512 workgroup slots gives its remainder tiles to
 * workgroups that each take an equal share of those tiles' K loop; they exchange partial accumulator tiles through `parts` and
 * count arrivals per tile in `arrivals`.  Both buffers are the caller's, registered for ONE stream (launches on that stream use
 * them in stream order; two graphs that bake the same scratch must not replay concurrently).  `arrivals` must be zero when it is
 * registered; the kernels leave it zero.  A stream without scratch runs the plain launches (same results up to the order of the
 * K sum).  parts == NULL forgets the stream.  The reference has no counterpart: ATen picks its conv algorithm inside
 * torch.nn.functional.conv2d (mnist_wgan_conditional.py:61-70,87-95).                                                          */
/* Which launch form a convolution takes, as text ("128x128 tiles: 512", "64x128 tiles: 512", "stream-K: 512 whole tiles + 352 tiles x
 * 72 k-tiles over 512 ranges", "GEMM + col2im: ...", "4 phases: stream-K over unequal phases: ..."): the host-side launch planning,
 * no device needed.  op: 0 forward, 1 grad-input, 2 grad-weight; assume_scratch != 0: plan as if the stream had stream-K scratch. */
int pcg_conv_plan_describe(const pcg_conv_geom* g, int32_t op, int32_t assume_scratch, char* out, size_t out_bytes);
size_t pcg_conv_scratch_parts_bytes(void);
size_t pcg_conv_scratch_arrivals_bytes(void);
int pcg_conv_set_scratch(pcg_stream_t stream, void* parts, size_t parts_bytes, void* arrivals, size_t arrivals_bytes);
/* Zero-fill the arrival counters of `stream`'s scratch in stream order — after a launch on that stream failed or was aborted (the
 * kernels leave the counters zero only when every range of a launch ran).  The registry is keyed by (current device, stream): the
 * default stream has the same handle on every device.  No-op for a stream without scratch.                                      */
int pcg_conv_reset_scratch(pcg_stream_t stream);
/* Diagnostic builds only (`make -C csrc stamp`, -DPCG_CLOCK_STAMP): every conv kernel block leaves {shader-clock ticks, 100 MHz
 * ticks} of its main loop at buf[2*block], buf[2*block+1] (uint64) — the clock the chip holds inside the kernel.  Returns 1 when
 * this build stamps, 0 for the shipped library (which compiles no stamp code).  buf = NULL turns it off.                        */
int pcg_debug_stamp_buffer(void* buf, int64_t bytes);

/* ---- calibration (diagnostics; not on the step's path) -----------------------------------------------------------------------
 * What THIS box's fp32 matrix pipe and HBM sustain right now — bench.py prints it next to the step (`calib`) so that a run on a
 * slower-clocked box can be told from a slower kernel (the reference has nothing comparable: it publishes no performance numbers,
 * BASELINE.md §2 "re-derive from the box").
 *   pcg_calib_mfma   `rounds` blocks per CU of 4 waves, each issuing iters*16 back-to-back v_mfma_f32_32x32x2_f32 on four
 *                    independent accumulators.  workspace: [2048 floats of operands — caller-initialised, any finite values]
 *                    [blocks*256 floats sink][blocks*2 uint64: shader-clock ticks, 100 MHz reference ticks of the block's loop].
 *                    *flop_out = FLOP of the launch; *stamps_out = device address of the stamps.  Time it with events on `stream`.
 *   pcg_calib_copy   dst <- src (16 bytes per lane, grid-stride): 2*nbytes of HBM traffic.                                       */
int32_t pcg_calib_mfma_blocks(int32_t rounds);
size_t pcg_calib_mfma_workspace_bytes(int32_t rounds);
int pcg_calib_mfma(int32_t iters, int32_t rounds, void* workspace, size_t workspace_bytes, double* flop_out /*nullable*/,
                   uint64_t** stamps_out /*nullable*/, pcg_stream_t stream);
int pcg_calib_copy(const void* src, void* dst, int64_t nbytes, pcg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCGAN_HIP_H */
