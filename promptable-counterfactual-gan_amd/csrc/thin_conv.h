// thin_conv.h — internal entry points for convolutions with a 1..4-channel side (HBM-bound; vector ALU).
#pragma once
#include "pcg_common.h"

namespace pcg {

// true if the geometry is served by the thin kernels rather than the MFMA implicit GEMM
inline bool thin_is_cin(const pcg_conv_geom* g) { return g->Cin <= 3; }
inline bool thin_is_cout(const pcg_conv_geom* g) { return g->Cout <= 3 && !thin_is_cin(g); }

// x of a full-window Cout = 1 layer given as the PRE-BatchNorm output of the layer below plus that layer's batch statistics (r04):
// mean / invstd [groups][Cin]; the kernels read act(bn(x))
struct ThinBnIn { const float* mean; const float* invstd; const float* gamma; const float* beta; int act; float slope; int groups; };
bool thin_conv_bnin_full_ok(const pcg_conv_geom* g, int groups);
int thin_conv_fwd(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, void* ws,
                  size_t ws_bytes, hipStream_t s, int act = PCG_ACT_NONE, float slope = 0.f, const ThinBnIn* bnin = nullptr);
// input transform on the WIDE operand of a Cin-thin layer (r04): it is a pre-BatchNorm tensor read as act(z * scale[c] + shift[c]) — a
// one-channel ConvTranspose2d behind BatchNorm + ReLU (DCGAN's G5 behind G4) then needs no BatchNorm-apply pass and no activated copy
struct ThinXf { const float* scale; const float* shift; float neg; };
bool thin_conv_xf_ok(const pcg_conv_geom* g);
int thin_conv_dgrad(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, float* dx, void* ws,
                    size_t ws_bytes, hipStream_t s, int act = PCG_ACT_NONE, float slope = 0.f, const ThinXf* xf = nullptr);
size_t thin_conv_fwd_workspace_bytes(const pcg_conv_geom* g);   // optional scratch enabling the two-stage reduce path
size_t thin_conv_dgrad_workspace_bytes(const pcg_conv_geom* g);
size_t thin_conv_wgrad_workspace_bytes(const pcg_conv_geom* g);
int thin_conv_wgrad(const pcg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, void* ws,
                    size_t ws_bytes, hipStream_t s, const ThinXf* xf = nullptr, const ThinBnIn* bnin = nullptr);

// Cin-thin forward fused with the BatchNorm + ReLU / LeakyReLU backward of the layer whose activated output it is a gradient of (r04)
bool thin_conv_fwd_bnbwd_ok(const pcg_conv_geom* g);
size_t thin_conv_fwd_bnbwd_workspace_bytes(const pcg_conv_geom* g);
int thin_conv_fwd_bnbwd(const pcg_conv_geom* g, const float* x, const float* w, const float* z, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, int act, float slope, float* dz, float* dgamma, float* dbeta, int accumulate,
                        void* ws, size_t ws_bytes, hipStream_t s);

// grad-input of a Cout-thin convolution times act'(a_below) in the same pass (row-block forms only)
bool thin_conv_dgrad_mask_ok(const pcg_conv_geom* g);
int thin_conv_dgrad_mask(const pcg_conv_geom* g, const float* dy, const float* w, const float* a_below, int act, float slope, float* dx,
                         hipStream_t s);

// grad-input of a full-window Cout = 1 convolution pushed through the BatchNorm + ReLU / LeakyReLU backward of the layer below (r04)
bool thin_conv_dgrad_bnbwd_full_ok(const pcg_conv_geom* g, int groups);
size_t thin_conv_dgrad_bnbwd_full_workspace_bytes(const pcg_conv_geom* g, int groups);
int thin_conv_dgrad_bnbwd_full(const pcg_conv_geom* g, const float* dy, const float* w, const float* z, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, int act, float slope, float* dz, float* dgamma, float* dbeta, int accumulate,
                               int groups, void* ws, size_t ws_bytes, hipStream_t s);

// shared with conv_igemm.hip: dw[i] = (acc ? dw[i] : 0) + sum_z slab[z*stride + i]
// deferrable: a weight gradient's reduction — recorded instead of launched while slab_defer_begin .. slab_defer_flush is open on this thread
int launch_slab_reduce(const float* slab, float* dw, size_t n, size_t slab_stride, int nslabs, int accumulate, hipStream_t s, bool deferrable = false);

// deferred slab reductions: between begin and flush launch_slab_reduce records its arguments; the flush reduces all entries in one launch
int slab_defer_begin(hipStream_t s);
int slab_defer_flush(hipStream_t s);
int slab_defer_pending();   // entries recorded and not yet launched; -1 when not deferring

}  // namespace pcg
