// thin_conv.hip — convolutions with a 1..3-channel side: DCGAN D's first conv (1->64, mnist_dcgan.py:100),
// D's last conv (512->1, :111) and G's last ConvTranspose (64->1, :88); counteRGAN conv_in (3->64),
// conv_out (64->1) and the 2-channel discriminator entry (models/generator.py:39,50; discriminator.py:14).
// With N (or K) of 1..3 these are HBM-bound streaming kernels, not MFMA work: one "wide" NHWC tensor
// [B][WH][WW][C] is read or written exactly once with 16-byte lane accesses, the "thin" tensor
// [B][TH][TW][Cs] and the weights come from L1/LDS.
//
// Three access patterns, each usable on the conv's output grid (direct tap map  q = p*s - pad + k) or its
// input grid (transposed tap map  q = (p + pad - k)/s when divisible):
//   expand : wide[P][c]  = bias[c] + sum_{tap,cs} thin[map(P,tap)][cs] * W[tap,cs][c]
//   reduce : thin[P][cs] = bias[cs] + sum_{tap} sum_c wide[map(P,tap)][c] * W[tap,cs][c]
//   wgrad  : dW[tap,cs][c] = sum_P wide[P][c] * thin[map(P,tap)][cs]
#include "thin_conv.h"
#include <cstdlib>

namespace pcg {
namespace {

struct ThinP {
  const float* thin;
  const float* wide;
  const float* w;
  const float* bias;
  float* out;
  int act; float slope;   // activation fused into the output write (PCG_ACT_NONE: none)
  const float* xf_scale; const float* xf_shift; float xf_neg;   // input transform on the WIDE operand (r04, tap-dot on the matrix cores and row-block weight
                                                               // gradient only): wide := act(wide * scale[c] + shift[c]) — the producer's BatchNorm + ReLU / LeakyReLU
  const float* mask_src; float mask_neg;   // row-block expand only (r04): out *= (mask_src > 0 ? 1 : mask_neg), mask_src of the output's shape —
                                           // the ReLU / LeakyReLU backward of the layer whose activated output this gradient belongs to
  int B, TH, TW, Cs, WH, WW, C;
  int KH, KW, stride, pad, transposed;
  int wsS, wsT, wsC;       // weight element (cs, tap, c) lives at cs*wsS + tap*wsT + c*wsC
  int IH_, IW_;            // iteration grid
  int QH, QW;              // other grid
  int npix;                // B * IH_ * IW_
  FastDiv dIW, dIH, dCQ;
};

__device__ __forceinline__ bool tap_map(const ThinP& p, int ph, int pw, int kh, int kw, int& qh, int& qw) {
  if (!p.transposed) {
    qh = ph * p.stride - p.pad + kh;
    qw = pw * p.stride - p.pad + kw;
  } else {
    const int th = ph + p.pad - kh, tw = pw + p.pad - kw;
    if (th < 0 || tw < 0) return false;
    if (p.stride == 1) { qh = th; qw = tw; }
    else if (p.stride == 2) { if ((th | tw) & 1) return false; qh = th >> 1; qw = tw >> 1; }
    else { if (th % p.stride || tw % p.stride) return false; qh = th / p.stride; qw = tw / p.stride; }
  }
  return (unsigned)qh < (unsigned)p.QH && (unsigned)qw < (unsigned)p.QW;
}

// LDS weight image Wl[(tap*Cs+cs)][C]
__device__ __forceinline__ void stage_weights(const ThinP& p, float* Wl) {
  const int TT = p.KH * p.KW * p.Cs;
  for (int i = threadIdx.x; i < TT * p.C; i += blockDim.x) {
    const int t = i / p.C, c = i - t * p.C;
    const int tap = t / p.Cs, cs = t - tap * p.Cs;
    Wl[i] = p.w[(size_t)cs * p.wsS + (size_t)tap * p.wsT + (size_t)c * p.wsC];
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) thin_expand_kernel(ThinP p) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];
  stage_weights(p, Wl);
  const int CQ = p.C >> 2;
  const uint32_t total = (uint32_t)p.npix * (uint32_t)CQ;
  const float4* bias4 = reinterpret_cast<const float4*>(p.bias);
  float4* out4 = reinterpret_cast<float4*>(p.out);
  for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    uint32_t pix, cq, t, pw, b, ph;
    p.dCQ.divmod(idx, pix, cq);
    p.dIW.divmod(pix, t, pw);
    p.dIH.divmod(t, b, ph);
    float4 acc = p.bias ? bias4[cq] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw) {
        int qh, qw;
        if (!tap_map(p, (int)ph, (int)pw, kh, kw, qh, qw)) continue;
        const float* sp = p.thin + (size_t)(((int)b * p.TH + qh) * p.TW + qw) * p.Cs;
        const float* wl = Wl + (size_t)((kh * p.KW + kw) * p.Cs) * p.C + 4 * cq;
        for (int cs = 0; cs < p.Cs; ++cs) {
          const float sv = sp[cs];
          const float4 w4 = *reinterpret_cast<const float4*>(wl + (size_t)cs * p.C);
          acc.x = fmaf(sv, w4.x, acc.x); acc.y = fmaf(sv, w4.y, acc.y);
          acc.z = fmaf(sv, w4.z, acc.z); acc.w = fmaf(sv, w4.w, acc.w);
        }
      }
    if (p.act != PCG_ACT_NONE) {   // wide outputs fuse ReLU / LeakyReLU only (host falls back otherwise): neg = 0 / slope
      acc.x = act_neg_scale(acc.x, p.slope); acc.y = act_neg_scale(acc.y, p.slope);
      acc.z = act_neg_scale(acc.z, p.slope); acc.w = act_neg_scale(acc.w, p.slope);
    }
    out4[idx] = acc;
  }
}

// 16 lanes per output pixel, each lane strides over the channel quads; xor-shuffle tree inside the 16-lane group
__global__ void __launch_bounds__(256) thin_reduce_kernel(ThinP p) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];
  stage_weights(p, Wl);
  const int CQ = p.C >> 2;
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (uint32_t pix = blockIdx.x * 16u + grp; pix < (uint32_t)p.npix; pix += gridDim.x * 16u) {
    uint32_t t, pw, b, ph;
    p.dIW.divmod(pix, t, pw);
    p.dIH.divmod(t, b, ph);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw) {
        int qh, qw;
        if (!tap_map(p, (int)ph, (int)pw, kh, kw, qh, qw)) continue;
        const float4* vp = reinterpret_cast<const float4*>(p.wide + (size_t)(((int)b * p.WH + qh) * p.WW + qw) * p.C);
        const float* wl = Wl + (size_t)((kh * p.KW + kw) * p.Cs) * p.C;
        for (int cq = l16; cq < CQ; cq += 16) {
          const float4 v = vp[cq];
          {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + 4 * cq);
            a0 = fmaf(v.x, w4.x, a0); a0 = fmaf(v.y, w4.y, a0); a0 = fmaf(v.z, w4.z, a0); a0 = fmaf(v.w, w4.w, a0);
          }
          if (p.Cs > 1) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + p.C + 4 * cq);
            a1 = fmaf(v.x, w4.x, a1); a1 = fmaf(v.y, w4.y, a1); a1 = fmaf(v.z, w4.z, a1); a1 = fmaf(v.w, w4.w, a1);
          }
          if (p.Cs > 2) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + 2 * p.C + 4 * cq);
            a2 = fmaf(v.x, w4.x, a2); a2 = fmaf(v.y, w4.y, a2); a2 = fmaf(v.z, w4.z, a2); a2 = fmaf(v.w, w4.w, a2);
          }
        }
      }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      a0 += __shfl_xor(a0, off);
      a1 += __shfl_xor(a1, off);
      a2 += __shfl_xor(a2, off);
    }
    if (l16 == 0) {
      float* o = p.out + (size_t)pix * p.Cs;
      o[0] = act_apply(a0 + (p.bias ? p.bias[0] : 0.f), p.act, p.slope);
      if (p.Cs > 1) o[1] = act_apply(a1 + (p.bias ? p.bias[1] : 0.f), p.act, p.slope);
      if (p.Cs > 2) o[2] = act_apply(a2 + (p.bias ? p.bias[2] : 0.f), p.act, p.slope);
    }
  }
}

#include "thin_fast.inc"
#include "thin_rows.inc"

// ---- full-window forms (r04) ------------------------------------------------------------------------------------------------
// A Cout = 1 convolution whose window is the whole input map (DCGAN D's last layer, mnist_dcgan.py:110: Conv2d(512, 1, 4, 1, 0) on a
// 4x4 map) is one dot product per sample over L = KH*KW*Cin contiguous floats: the NHWC row of x[b] and the OHWI filter run in the
// same order.  The tap-map kernels above spend 16 taps' worth of loads and FMAs per element on it, 15 of them on padding zeros.
//   forward     y[b]  = act(bias + <x[b], w>)        one block per sample, lanes stride the row, fixed-order block sum
//   grad-input  dx[b] = dy[b] * w                    outer product: one thread per float4 column, write-bound
//   weight grad dw    = sum_b dy[b] * x[b]           column threads over a chunk of samples -> slab rows -> launch_slab_reduce
constexpr int FULL_ROWS_DX = 8;    // samples per block, grad-input
constexpr int FULL_ROWS_DW = 16;   // samples per slab row, weight gradient

bool full_window(const pcg_conv_geom* g) {
  const int64_t L = (int64_t)g->KH * g->KW * g->Cin;
  return g->Cout == 1 && g->Cin > 3 && g->pad == 0 && g->KH == g->IH && g->KW == g->IW && g->OH == 1 && g->OW == 1 && L % 4 == 0 &&
         (int64_t)g->B * L < (1ll << 31);
}

// x as the PRE-BatchNorm output z of the layer below (r04): the kernels evaluate act(bn(z)) on their loads with that layer's batch
// statistics (mean / invstd [G][C], group = sample / Bg) — the BatchNorm-apply pass and the activated copy of D4's output disappear
// (mnist_dcgan.py:108-110).  A thread's channel quad is fixed (256 % (C/4) == 0, rows are [hw][C]); same bn_fold expression as every
// BatchNorm pass, so the values are the ones pcg_bn_apply_act would have written.
struct FullXf { const float* mean; const float* invstd; const float* gamma; const float* beta; float neg; int C, Bg; };   // mean == nullptr: x is used as it is
__device__ __forceinline__ void full_xf_coef(const FullXf& xf, int grp, int cq, float (&sc)[4], float (&sh)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * cq + e;
    bn_fold(xf.gamma[c], xf.beta[c], xf.mean[grp * xf.C + c], xf.invstd[grp * xf.C + c], sc[e], sh[e]);
  }
}
__device__ __forceinline__ float4 full_xf_apply(float4 v, const float (&sc)[4], const float (&sh)[4], float neg) {
  return make_float4(act_neg_scale(fmaf(v.x, sc[0], sh[0]), neg), act_neg_scale(fmaf(v.y, sc[1], sh[1]), neg),
                     act_neg_scale(fmaf(v.z, sc[2], sh[2]), neg), act_neg_scale(fmaf(v.w, sc[3], sh[3]), neg));
}
__global__ void __launch_bounds__(256) thin_full_dot_kernel(const float4* __restrict__ x, const float4* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int B, int L4, int act, float slope, FullXf xf) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool has_xf = xf.mean != nullptr;
  const int cq = has_xf ? threadIdx.x % (xf.C >> 2) : 0;
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  int last_grp = -1;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float4* xr = x + (size_t)b * L4;
    if (has_xf && b / xf.Bg != last_grp) { last_grp = b / xf.Bg; full_xf_coef(xf, last_grp, cq, sc, sh); }   // (block-uniform)
    float a = 0.f;
    int i = threadIdx.x;
    for (; i + 7 * 256 < L4; i += 8 * 256) {   // 8 independent 16-byte loads of the row in flight
      float4 v[8], u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = xr[i + 256 * j]; u[j] = w[i + 256 * j]; }
      if (has_xf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = full_xf_apply(v[j], sc, sh, xf.neg);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a = fmaf(v[j].x, u[j].x, a); a = fmaf(v[j].y, u[j].y, a); a = fmaf(v[j].z, u[j].z, a); a = fmaf(v[j].w, u[j].w, a);
      }
    }
    for (; i < L4; i += 256) {
      float4 v = xr[i];
      const float4 u = w[i];
      if (has_xf) v = full_xf_apply(v, sc, sh, xf.neg);
      a = fmaf(v.x, u.x, a); a = fmaf(v.y, u.y, a); a = fmaf(v.z, u.z, a); a = fmaf(v.w, u.w, a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) red[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) y[b] = act_apply(((red[0] + red[1]) + (red[2] + red[3])) + (bias ? bias[0] : 0.f), act, slope);
    __syncthreads();
  }
}

// grid (ceil(L4/256), ceil(B/FULL_ROWS_DX)); neg = 1: no activation, 0 / slope: ReLU / LeakyReLU on the written value
__global__ void __launch_bounds__(256) thin_full_outer_kernel(const float* __restrict__ dy, const float4* __restrict__ w, float4* __restrict__ dx, int B,
                                                              int L4, float neg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L4) return;
  const float4 u = w[i];
  const int b0 = blockIdx.y * FULL_ROWS_DX;
#pragma unroll
  for (int j = 0; j < FULL_ROWS_DX; ++j) {
    const int b = b0 + j;
    if (b >= B) break;
    const float d = dy[b];
    float4 o = make_float4(d * u.x, d * u.y, d * u.z, d * u.w);
    if (neg != 1.f) { o.x = act_neg_scale(o.x, neg); o.y = act_neg_scale(o.y, neg); o.z = act_neg_scale(o.z, neg); o.w = act_neg_scale(o.w, neg); }
    dx[(size_t)b * L4 + i] = o;
  }
}

// grid (ceil(L4/256), ceil(B/FULL_ROWS_DW)); slab row blockIdx.y holds the chunk's sums, samples added in order
__global__ void __launch_bounds__(256) thin_full_wgrad_kernel(const float4* __restrict__ x, const float* __restrict__ dy, float4* __restrict__ slab, int B,
                                                              int L4, FullXf xf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L4) return;
  const int b0 = blockIdx.y * FULL_ROWS_DW;
  const bool has_xf = xf.mean != nullptr;       // (a chunk of FULL_ROWS_DW samples lies inside one group: checked on the host)
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_xf) full_xf_coef(xf, b0 / xf.Bg, i % (xf.C >> 2), sc, sh);
  float4 v[FULL_ROWS_DW];
  float d[FULL_ROWS_DW];
#pragma unroll
  for (int j = 0; j < FULL_ROWS_DW; ++j) {
    const int b = b0 + j;
    const bool ok = b < B;
    v[j] = ok ? x[(size_t)b * L4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_xf) v[j] = full_xf_apply(v[j], sc, sh, xf.neg);       // (a missing sample's dy is 0: its transformed value does not count)
    d[j] = ok ? dy[b] : 0.f;
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < FULL_ROWS_DW; ++j) {
    a.x = fmaf(d[j], v[j].x, a.x); a.y = fmaf(d[j], v[j].y, a.y); a.z = fmaf(d[j], v[j].z, a.z); a.w = fmaf(d[j], v[j].w, a.w);
  }
  slab[(size_t)blockIdx.y * L4 + i] = a;
}

// Full-window grad-input pushed through the BatchNorm(train) + ReLU / LeakyReLU BACKWARD of the layer below without being written (r04):
// d[b][hw][c] = dy[b] * w[hw][c] is one multiply to recompute, so the chain  outer product (write d) -> column sums (read d, z) -> apply
// (read d, z; write dz)  becomes  sums (read z) -> finalize -> apply (read z, write dz).  DCGAN: D's last conv above D4's BatchNorm
// (mnist_dcgan.py:108-110 backward), in the D step with two groups of samples (real | fake: their own statistics, §3.5).
// Same expressions for the mask (fma(z, sc, sh) > 0), xhat and dz as bn_bwd_apply / FnBnBwd / thin_rows_expand_bn_kernel.
//   sums : block (cb, chunk) owns FULLBN_CHUNK samples and 256/HW channel quads at ALL HW positions: thread (hw, cq); fp64 tallies per
//          thread, the HW positions of a channel added through LDS in position order -> one partial row [2][C] per chunk
//   apply: thread = one float4 column of the [B][L] gradient (fixed filter quad and channel quad), FULLBN_ROWS samples per block
constexpr int FULLBN_CHUNK = 16, FULLBN_ROWS = 8;
struct FullBn {
  const float* dy; const float4* w; const float4* z; const float* mean; const float* invstd; const float* gamma; const float* beta;
  const float* coef;     // [G][3][C]
  float neg; int C, HW, Bg;   // Bg: samples per group
};
__global__ void __launch_bounds__(256) thin_full_bn_sums_kernel(FullBn q, double* __restrict__ partial) {
  __shared__ double red[256];
  const int CQ = q.C >> 2, CQB = 256 / q.HW;
  const int hw = threadIdx.x / CQB, cq = blockIdx.x * CQB + threadIdx.x % CQB;
  const int b0 = blockIdx.y * FULLBN_CHUNK, grp = b0 / q.Bg;
  const float4 w4 = q.w[hw * CQ + cq];
  const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
  float sc[4], sh[4], mu[4], is[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = grp * q.C + 4 * cq + e;
    mu[e] = q.mean[c]; is[e] = q.invstd[c];
    bn_fold(q.gamma[4 * cq + e], q.beta[4 * cq + e], mu[e], is[e], sc[e], sh[e]);
  }
  float4 zv[FULLBN_CHUNK];
  float dv[FULLBN_CHUNK];
#pragma unroll
  for (int j = 0; j < FULLBN_CHUNK; ++j) {
    zv[j] = q.z[((size_t)(b0 + j) * q.HW + hw) * CQ + cq];
    dv[j] = q.dy[b0 + j];
  }
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < FULLBN_CHUNK; ++j) {
    const float zz[4] = {zv[j].x, zv[j].y, zv[j].z, zv[j].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pre = fmaf(zz[e], sc[e], sh[e]);
      const float dm = (dv[j] * ww[e]) * (pre > 0.f ? 1.f : q.neg);
      const float xh = (zz[e] - mu[e]) * is[e];
      s1[e] += (double)dm; s2[e] += (double)dm * (double)xh;
    }
  }
  double* row = partial + (size_t)blockIdx.y * 2 * q.C;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __syncthreads();
      red[threadIdx.x] = k ? s2[e] : s1[e];
      __syncthreads();
      if (hw == 0) {
        double t = red[threadIdx.x];
        for (int j = 1; j < q.HW; ++j) t += red[j * CQB + threadIdx.x];
        row[(size_t)k * q.C + 4 * cq + e] = t;
      }
    }
}
__global__ void __launch_bounds__(256) thin_full_bn_apply_kernel(FullBn q, float4* __restrict__ dz, int B, int L4) {
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  if (i4 >= L4) return;
  const int CQ = q.C >> 2, cq = i4 % CQ;
  const int b0 = blockIdx.y * FULLBN_ROWS, grp = b0 / q.Bg;
  const float4 w4 = q.w[i4];
  const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
  float sc[4], sh[4], mu[4], is[4], k0[4], k1[4], k2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * cq + e;
    mu[e] = q.mean[grp * q.C + c]; is[e] = q.invstd[grp * q.C + c];
    bn_fold(q.gamma[c], q.beta[c], mu[e], is[e], sc[e], sh[e]);
    const float* cg = q.coef + (size_t)grp * 3 * q.C;
    k0[e] = cg[c]; k1[e] = cg[q.C + c]; k2[e] = cg[2 * q.C + c];
  }
#pragma unroll
  for (int j = 0; j < FULLBN_ROWS; ++j) {
    const int b = b0 + j;
    if (b >= B) break;
    const float4 zv = q.z[(size_t)b * L4 + i4];
    const float d = q.dy[b];
    const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pre = fmaf(zz[e], sc[e], sh[e]);
      const float dm = (d * ww[e]) * (pre > 0.f ? 1.f : q.neg);
      const float xh = (zz[e] - mu[e]) * is[e];
      o[e] = k0[e] * (dm - k1[e] - xh * k2[e]);
    }
    dz[(size_t)b * L4 + i4] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Each block owns `ppb` iteration pixels; thread (cq, pl) accumulates NT float4 sums over pixels pl, pl+PL, ...
// then the PL pixel-lanes are summed through LDS in a fixed order and the block writes its slab in dw layout.
template <int KH, int KW, int CS>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(ThinP p, float* slab, int ppb, int wn, uint32_t thin_bytes) {
  constexpr int NT = KH * KW * CS;
  __shared__ float4 red[256];
  const thin_rsrc_t rs = thin_rsrc(p.thin, thin_bytes);
  const int CQ = p.C >> 2;           // power of two <= 256 (checked on the host)
  const int PL = 256 / CQ;
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  float4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int p0 = blockIdx.x * ppb;
  int p1 = p0 + ppb; if (p1 > p.npix) p1 = p.npix;
  for (int pix = p0 + pl; pix < p1; pix += PL) {
    uint32_t t, pw, b, ph;
    p.dIW.divmod((uint32_t)pix, t, pw);
    p.dIH.divmod(t, b, ph);
    const float4 v = *reinterpret_cast<const float4*>(p.wide + (size_t)pix * p.C + 4 * cq);
    float sv[NT];  // all tap scalars gathered branch-free first (invalid tap -> out-of-range offset -> 0)
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int q = tap_map_idx(p, (int)b, (int)ph, (int)pw, kh, kw);
#pragma unroll
        for (int cs = 0; cs < CS; ++cs)
          sv[(kh * KW + kw) * CS + cs] = thin_load1(rs, q >= 0 ? (uint32_t)((q * CS + cs) * 4) : THIN_OOB);
      }
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
      acc[tp].x = fmaf(sv[tp], v.x, acc[tp].x); acc[tp].y = fmaf(sv[tp], v.y, acc[tp].y);
      acc[tp].z = fmaf(sv[tp], v.z, acc[tp].z); acc[tp].w = fmaf(sv[tp], v.w, acc[tp].w);
    }
  }
  float* myslab = slab + (size_t)blockIdx.x * wn;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    red[threadIdx.x] = acc[t];
    __syncthreads();
    if (pl == 0) {
      float4 s = red[cq];
      for (int j = 1; j < PL; ++j) {
        const float4 o = red[j * CQ + cq];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      const int tap = t / CS, cs = t % CS;
      float* d = myslab + (size_t)cs * p.wsS + (size_t)tap * p.wsT + (size_t)(4 * cq) * p.wsC;
      d[0] = s.x; d[p.wsC] = s.y; d[2 * p.wsC] = s.z; d[3 * p.wsC] = s.w;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) slab_reduce4_kernel(const float4* __restrict__ s, float4* __restrict__ o, size_t n4,
                                                           size_t stride4, int nslabs, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = accumulate ? o[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 8 <= nslabs; z += 8) {  // 8 independent 16-byte loads in flight; summed in slab order
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s[(size_t)(z + j) * stride4 + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    for (; z < nslabs; ++z) {
      const float4 v = s[(size_t)z * stride4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    o[i] = a;
  }
}
// Few outputs, many slabs (the thin layers' 1024-float gradients arrive as 512 slabs): one thread per output would walk all slabs
// alone (4 blocks, 64 dependent load batches: 24 us).  Here 16 threads share an output: thread (i, g) adds slabs g, g+16, ... in
// order, the 16 group sums are added in group order through LDS — a fixed association, reproducible run to run.
__global__ void __launch_bounds__(256) slab_reduce4_wide_kernel(const float4* __restrict__ s, float4* __restrict__ o, size_t n4,
                                                                size_t stride4, int nslabs, int accumulate) {
  __shared__ float4 red[16][16];
  const int il = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + il;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    int z = g;
    for (; z + 7 * 16 < nslabs; z += 8 * 16) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s[(size_t)(z + 16 * j) * stride4 + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    for (; z < nslabs; z += 16) {
      const float4 v = s[(size_t)z * stride4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[g][il] = a;
  __syncthreads();
  if (g == 0 && i < n4) {
    float4 t = accumulate ? o[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float4 v = red[j][il]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    o[i] = t;
  }
}
__global__ void __launch_bounds__(256) slab_reduce1_kernel(const float* __restrict__ s, float* __restrict__ o, size_t n,
                                                           size_t stride, int nslabs, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float a = accumulate ? o[i] : 0.f;
    for (int z = 0; z < nslabs; ++z) a += s[(size_t)z * stride + i];
    o[i] = a;
  }
}

// Deferred slab reductions (r04).  A backward sweep ends every split-K weight gradient with a 5-6 us reduction launch (DCGAN: 9 per
// step, CounteRGAN: 29) that nothing in the sweep reads — the gradients are consumed after it, by Adam or the gradient exchange.
// Between pcg_slab_defer_begin and pcg_slab_defer_flush launch_slab_reduce only records (slab, dw, ...); the flush reduces all of
// them in ONE launch, blockIdx.y = entry, each entry summed exactly as the kernel it replaces sums it (plain: one thread per float4
// in slab order; wide: 16 threads per float4, groups added in group order) — bit-identical gradients.  The caller keeps every recorded
// slab buffer alive and distinct until the flush (ops.conv2d_wgrad takes one scratch slot per deferred call).
struct SlabEntry { const float4* slab; float4* dw; uint32_t n4, stride4; int nslabs, accumulate, wide; };
constexpr int SLAB_MANY_MAX = 24;
struct SlabMany { SlabEntry e[SLAB_MANY_MAX]; };
__global__ void __launch_bounds__(256) slab_reduce_many_kernel(SlabMany m) {
  __shared__ float4 red[16][16];
  const SlabEntry e = m.e[blockIdx.y];
  if (e.wide) {
    if ((size_t)blockIdx.x * 16 >= e.n4) return;                 // (block-uniform)
    const int il = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + il;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < e.n4) {
      int z = g;
      for (; z + 7 * 16 < e.nslabs; z += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = e.slab[(size_t)(z + 16 * j) * e.stride4 + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
      }
      for (; z < e.nslabs; z += 16) {
        const float4 v = e.slab[(size_t)z * e.stride4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    red[g][il] = a;
    __syncthreads();
    if (g == 0 && i < e.n4) {
      float4 t = e.accumulate ? e.dw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 16; ++j) { const float4 v = red[j][il]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      e.dw[i] = t;
    }
    return;
  }
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= e.n4) return;
  float4 a = e.accumulate ? e.dw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  int z = 0;
  for (; z + 8 <= e.nslabs; z += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = e.slab[(size_t)(z + j) * e.stride4 + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
  }
  for (; z < e.nslabs; ++z) {
    const float4 v = e.slab[(size_t)z * e.stride4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  e.dw[i] = a;
}
struct SlabDefer { bool on = false; hipStream_t stream = nullptr; int n = 0; unsigned max_blocks = 0; SlabMany m; };
thread_local SlabDefer g_slab_defer;
int slab_defer_launch() {
  SlabDefer& d = g_slab_defer;
  if (d.n == 0) return PCG_OK;
  hipLaunchKernelGGL(slab_reduce_many_kernel, dim3(d.max_blocks, (unsigned)d.n), dim3(256), 0, d.stream, d.m);
  d.n = 0; d.max_blocks = 0;
  return launch_status("slab_reduce_many_kernel");
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// row-block plan (thin_rows.inc): direct tap map with any stride, or the transposed map with stride 1; the wide grid must be
// a real image (>= 64 pixels) and the thin tensor addressable with 32-bit byte offsets
bool rows_plan(const ThinP& p, RowsP& rp, size_t* patch_bytes, int px_per_thread = 16, size_t max_patch_bytes = (size_t)ROWS_MAXP * 256 * sizeof(float)) {
  const bool direct = !p.transposed;
  if (!direct && p.stride != 1) return false;
  const int CQ = p.C / 4;
  if (p.C % 4 || CQ > 256 || !is_pow2(CQ)) return false;
  if (p.IH_ * p.IW_ < 64) return false;
  if ((int64_t)p.B * p.TH * p.TW * p.Cs * 4 >= (1ll << 31)) return false;
  const int PL = 256 / CQ;
  int R;
  if (px_per_thread == 16) {               // expand: about 16 pixels per thread, at least 256 per unit
    int target = PL * 16;
    if (target < 256) target = 256;
    R = (target + p.IW_ - 1) / p.IW_;
  } else {                                 // wgrad: at most px_per_thread pixels per thread (they live in registers)
    R = PL * px_per_thread / p.IW_;
    if (R < 1) return false;
  }
  if (R > p.IH_) R = p.IH_;
  rp.s = direct ? p.stride : 1;
  rp.flip = direct ? 0 : 1;
  rp.qh_off = direct ? -p.pad : p.pad - p.KH + 1;
  rp.qw_off = direct ? -p.pad : p.pad - p.KW + 1;
  rp.R = R;
  rp.upi = (p.IH_ + R - 1) / R;
  rp.nunits = p.B * rp.upi;
  rp.PH = (R - 1) * rp.s + p.KH;
  rp.PW = (p.IW_ - 1) * rp.s + p.KW;
  rp.dUPI = FastDiv((uint32_t)rp.upi);
  rp.dPWC = FastDiv((uint32_t)(rp.PW * p.Cs));
  rp.dCS = FastDiv((uint32_t)p.Cs);
  *patch_bytes = (size_t)rp.PH * rp.PW * p.Cs * sizeof(float);
  return *patch_bytes <= max_patch_bytes;
}

// geometry -> ThinP for the two thin families
int fill_common(ThinP& p, const pcg_conv_geom* g, bool cin_thin, bool iter_on_output) {
  p.B = g->B; p.KH = g->KH; p.KW = g->KW; p.stride = g->stride; p.pad = g->pad;
  const int T = g->KH * g->KW;
  if (cin_thin) {  // thin = x side (input grid), wide = y side (output grid)
    p.TH = g->IH; p.TW = g->IW; p.Cs = g->Cin; p.WH = g->OH; p.WW = g->OW; p.C = g->Cout;
    p.wsS = 1; p.wsT = g->Cin; p.wsC = T * g->Cin;
  } else {         // thin = y side (output grid), wide = x side (input grid)
    p.TH = g->OH; p.TW = g->OW; p.Cs = g->Cout; p.WH = g->IH; p.WW = g->IW; p.C = g->Cin;
    p.wsS = T * g->Cin; p.wsT = g->Cin; p.wsC = 1;
  }
  p.transposed = iter_on_output ? 0 : 1;
  if (iter_on_output) { p.IH_ = g->OH; p.IW_ = g->OW; p.QH = g->IH; p.QW = g->IW; }
  else { p.IH_ = g->IH; p.IW_ = g->IW; p.QH = g->OH; p.QW = g->OW; }
  p.npix = g->B * p.IH_ * p.IW_;
  p.dIW = FastDiv((uint32_t)p.IW_); p.dIH = FastDiv((uint32_t)p.IH_); p.dCQ = FastDiv((uint32_t)(p.C / 4));
  PCG_REQUIRE(p.C % 4 == 0, "thin conv: wide channel count %d must be a multiple of 4", p.C);
  PCG_REQUIRE(p.Cs >= 1 && p.Cs <= 3, "thin conv: thin channel count %d not in 1..3", p.Cs);
  PCG_REQUIRE((int64_t)p.npix * (p.C / 4) < (1ll << 31), "thin conv: problem too large for 32-bit indexing");
  return PCG_OK;
}

// blocks of the row-block expand kernels: every block walks units blockIdx.x, + gridDim.x, ... with the next patch in flight
unsigned rows_expand_blocks(int nunits) {
  static const int cap = [] { const char* e = getenv("PCG_ROWS_EXPAND_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
  return (unsigned)(nunits < cap ? nunits : cap);
}

size_t lds_weight_bytes(const ThinP& p) { return (size_t)p.KH * p.KW * p.Cs * p.C * sizeof(float); }

bool fast_ok(const ThinP& p) {
  return p.Cs == 1 && p.stride <= 2 && p.KH * p.KW <= 16 && p.C % 16 == 0 && lds_weight_bytes(p) <= 64 * 1024;
}

int launch_expand(ThinP& p, hipStream_t s) {
  const size_t smem = lds_weight_bytes(p);
  PCG_REQUIRE(smem <= 64 * 1024, "thin conv: weight image %zu B exceeds 64 KB of LDS", smem);
  const bool k44 = p.KH == 4 && p.KW == 4, k33 = p.KH == 3 && p.KW == 3, k11 = p.KH == 1 && p.KW == 1;
  RowsP rp{};
  size_t patch_bytes = 0;
  if ((k44 || k33) && (k33 || p.Cs == 1) && rows_plan(p, rp, &patch_bytes)) {
    const size_t two = 2 * ((patch_bytes + 15) & ~(size_t)15), sm = smem > two ? smem : two;
    const uint32_t thin_bytes = (uint32_t)((int64_t)p.B * p.TH * p.TW * p.Cs * 4);
    const unsigned blocks = rows_expand_blocks(rp.nunits);
    static const int mfma_on = [] { const char* e = getenv("PCG_EXPAND_MFMA"); return e ? atoi(e) : 1; }();      // A/B switch
    if (mfma_on && p.C == 64 && (((uintptr_t)p.out | (uintptr_t)p.mask_src) & 15) == 0) {     // 64 wide channels, at most 32 (tap, thin channel) pairs: the matrix-core form
      const size_t sm2 = 2 * ((patch_bytes + 15) & ~(size_t)15);
      if (k44) hipLaunchKernelGGL((thin_rows_expand_mfma_kernel<4, 4, 1>), dim3(blocks), dim3(256), sm2, s, p, rp, thin_bytes);
      else if (p.Cs == 1) hipLaunchKernelGGL((thin_rows_expand_mfma_kernel<3, 3, 1>), dim3(blocks), dim3(256), sm2, s, p, rp, thin_bytes);
      else if (p.Cs == 2) hipLaunchKernelGGL((thin_rows_expand_mfma_kernel<3, 3, 2>), dim3(blocks), dim3(256), sm2, s, p, rp, thin_bytes);
      else hipLaunchKernelGGL((thin_rows_expand_mfma_kernel<3, 3, 3>), dim3(blocks), dim3(256), sm2, s, p, rp, thin_bytes);
      return launch_status("thin_rows_expand_mfma_kernel");
    }
#define PCG_ROWS_EXPAND_CASE(KH_, KW_, CS_)                                                                                  \
    if (p.KH == KH_ && p.Cs == CS_) {                                                                                         \
      hipLaunchKernelGGL((thin_rows_expand_kernel<KH_, KW_, CS_>), dim3(blocks), dim3(256), sm, s, p, rp, thin_bytes);      \
      return launch_status("thin_rows_expand_kernel");                                                                        \
    }
    PCG_ROWS_EXPAND_CASE(4, 4, 1)
    PCG_ROWS_EXPAND_CASE(3, 3, 1)
    PCG_ROWS_EXPAND_CASE(3, 3, 2)
    PCG_ROWS_EXPAND_CASE(3, 3, 3)
#undef PCG_ROWS_EXPAND_CASE
  }
  if (fast_ok(p) && (k44 || k33 || k11) && (int64_t)p.npix * (p.C / 16) < (1ll << 31)) {
    ThinP q = p;
    q.dCQ = FastDiv((uint32_t)(p.C / 16));
    const uint32_t thin_bytes = (uint32_t)((int64_t)p.B * p.TH * p.TW * 4);
    const uint64_t total = (uint64_t)p.npix * (p.C / 16);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (k44) hipLaunchKernelGGL((thin_expand16_kernel<4, 4>), dim3(blocks), dim3(256), smem, s, q, thin_bytes);
    else if (k33) hipLaunchKernelGGL((thin_expand16_kernel<3, 3>), dim3(blocks), dim3(256), smem, s, q, thin_bytes);
    else hipLaunchKernelGGL((thin_expand16_kernel<1, 1>), dim3(blocks), dim3(256), smem, s, q, thin_bytes);
    return launch_status("thin_expand16_kernel");
  }
  const uint64_t total = (uint64_t)p.npix * (p.C / 4);
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(thin_expand_kernel, dim3(blocks), dim3(256), smem, s, p);
  return launch_status("thin_expand_kernel");
}

// scratch the two-stage reduce wants: T[wide pixels][16]
// 2..3 thin channels, 3x3 taps: 18 / 27 dot products per wide pixel, T rows of 32 floats
bool fast32_ok(const ThinP& p) {
  return p.Cs >= 2 && p.Cs <= 3 && p.KH == 3 && p.KW == 3 && p.stride <= 2 && p.C % 16 == 0 && lds_weight_bytes(p) <= 64 * 1024;
}

size_t reduce_scratch_bytes(const ThinP& p) {
  if (fast32_ok(p)) {
    const int64_t b = (int64_t)p.B * p.WH * p.WW * 32 * 4;
    return b < (1ll << 31) ? (size_t)b : 0;
  }
  if (!fast_ok(p)) return 0;
  const int64_t b = (int64_t)p.B * p.WH * p.WW * 16 * 4;
  return b < (1ll << 31) ? (size_t)b : 0;
}

int launch_reduce(ThinP& p, void* ws, size_t ws_bytes, hipStream_t s) {
  const size_t smem = lds_weight_bytes(p);
  PCG_REQUIRE(smem <= 64 * 1024, "thin conv: weight image %zu B exceeds 64 KB of LDS", smem);
  const size_t need = reduce_scratch_bytes(p);
  PCG_REQUIRE(!p.xf_scale || !fast32_ok(p), "thin conv: input transforms only with one thin channel");
  if (need && ws && ws_bytes >= need && (((uintptr_t)ws) & 15) == 0 && fast32_ok(p)) {
    float* T = (float*)ws;
    ThinP q = p;
    q.npix = p.B * p.WH * p.WW;  // stage 1 walks the wide tensor
    unsigned blocks = (unsigned)((q.npix + 15) / 16);
    if (blocks > 8192) blocks = 8192;
    uint32_t plane = 0;      // tap-major T (planes of q.npix floats) behind the matrix-core form, pixel-major rows behind the vector kernels
    if (p.C == 64 && (((uintptr_t)p.wide) & 15) == 0) {   // the matrix-core form: two 16-output column blocks
      unsigned mb = (unsigned)((q.npix + 127) / 128);
      if (mb > 4096) mb = 4096;
      hipLaunchKernelGGL((thin_tapdot64_mfma_kernel<2, false>), dim3(mb), dim3(256), 0, s, q, T);
      plane = (uint32_t)q.npix;
    } else if (p.Cs == 2) hipLaunchKernelGGL(thin_tapdot32_kernel<18>, dim3(blocks), dim3(256), smem, s, q, T);
    else hipLaunchKernelGGL(thin_tapdot32_kernel<27>, dim3(blocks), dim3(256), smem, s, q, T);
    if (int e = launch_status("thin_tapdot32_kernel")) return e;
    unsigned b2 = (unsigned)(((int64_t)p.npix * p.Cs + 255) / 256);
    if (b2 > 8192) b2 = 8192;
    hipLaunchKernelGGL(thin_col2im32_kernel, dim3(b2), dim3(256), 0, s, p, (const float*)T, (uint32_t)need, plane);
    return launch_status("thin_col2im32_kernel");
  }
  if (need && ws && ws_bytes >= need && (((uintptr_t)ws) & 15) == 0) {
    float* T = (float*)ws;
    ThinP q = p;
    q.npix = p.B * p.WH * p.WW;  // stage 1 walks the wide tensor
    unsigned blocks = (unsigned)((q.npix + 15) / 16);
    if (blocks > 8192) blocks = 8192;
    const int nt = p.KH * p.KW;
    const bool mfma1 = p.C == 64, mfmaN = p.C % 64 == 0 && p.C > 64 && p.C <= TD64_MAX_C;
    uint32_t plane = 0;
    PCG_REQUIRE(!p.xf_scale || ((mfma1 || mfmaN) && nt <= 16 && (((uintptr_t)p.wide) & 15) == 0), "thin conv: an input transform needs the matrix-core tap-dot form");
    if ((mfma1 || mfmaN) && nt <= 16 && (((uintptr_t)p.wide) & 15) == 0) {
      unsigned mb = (unsigned)((q.npix + 127) / 128);
      if (mb > 4096) mb = 4096;
      if (mfma1) hipLaunchKernelGGL((thin_tapdot64_mfma_kernel<1, false>), dim3(mb), dim3(256), 0, s, q, T);
      else {
        if (mb > 1024) mb = 1024;     // (every block stages the filter image once)
        hipLaunchKernelGGL((thin_tapdot64_mfma_kernel<1, true>), dim3(mb), dim3(256), (size_t)16 * (p.C + 4) * sizeof(float), s, q, T);
      }
      plane = (uint32_t)q.npix;
    } else if (nt == 16) hipLaunchKernelGGL(thin_tapdot_kernel<16>, dim3(blocks), dim3(256), smem, s, q, T);
    else if (nt == 9) hipLaunchKernelGGL(thin_tapdot_kernel<9>, dim3(blocks), dim3(256), smem, s, q, T);
    else if (nt == 1) hipLaunchKernelGGL(thin_tapdot_kernel<1>, dim3(blocks), dim3(256), smem, s, q, T);
    else goto generic;
    if (int e = launch_status("thin_tapdot_kernel")) return e;
    if (p.transposed && p.stride == 2 && p.KH == 4 && p.KW == 4 && p.IW_ % 4 == 0 && (((uintptr_t)p.out) & 15) == 0) {
      unsigned b4 = (unsigned)((p.npix / 4 + 255) / 256);
      if (b4 > 8192) b4 = 8192;
      hipLaunchKernelGGL(thin_col2im_s2k4_kernel, dim3(b4), dim3(256), 0, s, p, (const float*)T, (uint32_t)need, plane);
      return launch_status("thin_col2im_s2k4_kernel");
    }
    unsigned b2 = (unsigned)((p.npix + 255) / 256);
    if (b2 > 8192) b2 = 8192;
    hipLaunchKernelGGL(thin_col2im_kernel, dim3(b2), dim3(256), 0, s, p, (const float*)T, (uint32_t)need, plane);
    return launch_status("thin_col2im_kernel");
  }
generic:
  PCG_REQUIRE(!p.xf_scale, "thin conv: an input transform needs the matrix-core tap-dot form (workspace of pcg_conv2d_dgrad_workspace_bytes)");
  unsigned blocks = (unsigned)((p.npix + 15) / 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(thin_reduce_kernel, dim3(blocks), dim3(256), smem, s, p);
  return launch_status("thin_reduce_kernel");
}

// row-block wgrad: number of slabs (= blocks); units per block = ceil(nunits / nblocks)
int rows_wgrad_blocks(const RowsP& rp) { return rp.nunits < 512 ? rp.nunits : 512; }
bool rows_wgrad_ok(const ThinP& p, const pcg_conv_geom* g, RowsP& rp, size_t* patch_bytes) {
  const bool k44 = g->KH == 4 && g->KW == 4 && p.Cs == 1, k33 = g->KH == 3 && g->KW == 3;
  return (k44 || k33) && rows_plan(p, rp, patch_bytes, ROWS_NPT, (size_t)ROWS_MAXP * 256 * sizeof(float));
}

struct ThinWgradPlan { int ppb, nblocks; };
ThinWgradPlan plan_thin_wgrad(int npix, int C) {
  ThinWgradPlan w;
  const int PL = 256 / (C / 4);
  int ppb = (npix + 1023) / 1024;           // ~1024 blocks
  const int min_ppb = PL * 8;               // at least 8 pixels per thread
  if (ppb < min_ppb) ppb = min_ppb;
  w.ppb = ppb;
  w.nblocks = (npix + ppb - 1) / ppb;
  return w;
}

}  // namespace

int launch_slab_reduce(const float* slab, float* dw, size_t n, size_t slab_stride, int nslabs, int accumulate, hipStream_t s, bool deferrable) {
  const bool vec = (n % 4 == 0) && (slab_stride % 4 == 0) && (((uintptr_t)slab | (uintptr_t)dw) & 15) == 0;
  if (deferrable && vec && g_slab_defer.on && s == g_slab_defer.stream && n / 4 < (1ull << 32) && slab_stride / 4 < (1ull << 32)) {
    SlabDefer& d = g_slab_defer;
    const size_t n4 = n / 4;
    bool clash = d.n == SLAB_MANY_MAX;
    for (int k = 0; k < d.n && !clash; ++k) clash = (const void*)d.m.e[k].dw == (const void*)dw;     // a second sum into the same gradient: in order
    if (clash)
      if (int e = slab_defer_launch()) return e;
    const int wide = nslabs >= 64 && n4 <= 16 * 1024;
    d.m.e[d.n++] = SlabEntry{reinterpret_cast<const float4*>(slab), reinterpret_cast<float4*>(dw), (uint32_t)n4, (uint32_t)(slab_stride / 4), nslabs,
                             accumulate, wide};
    const unsigned blocks = (unsigned)(wide ? (n4 + 15) / 16 : (n4 + 255) / 256);
    if (blocks > d.max_blocks) d.max_blocks = blocks;
    return PCG_OK;
  }
  if (vec) {
    const size_t n4 = n / 4;
    if (nslabs >= 64 && n4 <= 16 * 1024) {   // few outputs, many slabs: 16 threads per output
      hipLaunchKernelGGL(slab_reduce4_wide_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, reinterpret_cast<const float4*>(slab),
                         reinterpret_cast<float4*>(dw), n4, slab_stride / 4, nslabs, accumulate);
      return launch_status("slab_reduce_kernel");
    }
    unsigned blocks = (unsigned)((n4 + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(blocks), dim3(64), 0, s, reinterpret_cast<const float4*>(slab),
                       reinterpret_cast<float4*>(dw), n4, slab_stride / 4, nslabs, accumulate);
  } else {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(slab_reduce1_kernel, dim3(blocks), dim3(256), 0, s, slab, dw, n, slab_stride, nslabs, accumulate);
  }
  return launch_status("slab_reduce_kernel");
}

// the BatchNorm-input form of the full-window kernels: x is the layer below's pre-BatchNorm output
bool thin_conv_bnin_full_ok(const pcg_conv_geom* g, int groups) {
  if (!full_window(g) || groups < 1 || groups > 8 || g->B % groups || g->Cin % 4) return false;
  const int CQ = g->Cin / 4;
  return CQ <= 256 && 256 % CQ == 0 && (g->B / groups) % FULL_ROWS_DW == 0;
}
static FullXf full_xf_of(const pcg_conv_geom* g, const ThinBnIn* bi) {
  if (!bi || !bi->mean) return FullXf{nullptr, nullptr, nullptr, nullptr, 1.f, g->Cin, g->B};
  return FullXf{bi->mean, bi->invstd, bi->gamma, bi->beta, act_neg_of(bi->act, bi->slope), g->Cin, g->B / bi->groups};
}

int thin_conv_fwd(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, void* ws,
                  size_t ws_bytes, hipStream_t s, int act, float slope, const ThinBnIn* bnin) {
  PCG_REQUIRE(!(bnin && bnin->mean) || (thin_conv_bnin_full_ok(g, bnin->groups) && (((uintptr_t)x | (uintptr_t)w) & 15) == 0),
              "thin conv forward: a BatchNorm input needs the full-window form (thin_conv_bnin_full_ok) and 16-byte aligned tensors");
  ThinP p{};
  p.act = act; p.slope = slope;
  const bool cin_thin = thin_is_cin(g);
  if (int e = fill_common(p, g, cin_thin, /*iter_on_output=*/true)) return e;
  p.w = w; p.bias = bias; p.out = y;
  if (cin_thin) {                                              // y wide: fused ReLU / LeakyReLU, anything else as a second pass
    p.thin = x;
    const bool fuse = act_is_cheap(act);
    p.act = fuse ? act : PCG_ACT_NONE; p.slope = act_neg_of(p.act, slope);
    if (int e = launch_expand(p, s)) return e;
    return fuse ? PCG_OK : pcg_act_fwd(y, (int64_t)g->B * g->OH * g->OW * g->Cout, act, slope, y, (pcg_stream_t)s);
  }
  p.wide = x;                                                 // y thin
  if (full_window(g) && (((uintptr_t)x | (uintptr_t)w) & 15) == 0) {
    const int L4 = g->KH * g->KW * g->Cin / 4;
    hipLaunchKernelGGL(thin_full_dot_kernel, dim3((unsigned)(g->B < 4096 ? g->B : 4096)), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(w), bias, y, g->B, L4, act, slope, full_xf_of(g, bnin));
    return launch_status("thin_full_dot_kernel");
  }
  return launch_reduce(p, ws, ws_bytes, s);
}

// batchnorm.hip: partial[nblocks][2][C] (fp64) -> coef[3][C], dgamma / dbeta
int launch_bn_bwd_finalize(const double* partial, int nblocks, int64_t rows, int C, const float* gamma, const float* invstd, float* coef,
                           float* dgamma, float* dbeta, int accumulate, hipStream_t s);

// y-side gradient of a Cin-thin convolution (= the grad-input of a ConvTranspose2d with a thin output, DCGAN's G5) fused with the
// BatchNorm + ReLU / LeakyReLU backward of the layer below: see thin_rows_expand_bn_kernel.  workspace: thin_fwd_bnbwd_bytes.
constexpr unsigned THIN_BN_BLOCKS = 2048;
size_t thin_conv_fwd_bnbwd_workspace_bytes(const pcg_conv_geom* g) {
  return (size_t)THIN_BN_BLOCKS * 2 * g->Cout * sizeof(double) + (size_t)3 * g->Cout * sizeof(float);
}
bool thin_conv_fwd_bnbwd_ok(const pcg_conv_geom* g) {
  if (!thin_is_cin(g) || g->Cin != 1 || !(g->KH == 4 && g->KW == 4)) return false;
  ThinP p{};
  if (fill_common(p, g, true, true) != PCG_OK) return false;
  RowsP rp{};
  size_t patch_bytes = 0;
  return rows_plan(p, rp, &patch_bytes) && lds_weight_bytes(p) <= 64 * 1024;
}
int thin_conv_fwd_bnbwd(const pcg_conv_geom* g, const float* x, const float* w, const float* z, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, int act, float slope, float* dz, float* dgamma, float* dbeta, int accumulate,
                        void* ws, size_t ws_bytes, hipStream_t s) {
  PCG_REQUIRE(thin_conv_fwd_bnbwd_ok(g), "thin conv + BatchNorm backward: only the k4 one-channel row-block form");
  PCG_REQUIRE(ws && ws_bytes >= thin_conv_fwd_bnbwd_workspace_bytes(g), "thin conv + BatchNorm backward: workspace too small");
  ThinP p{};
  if (int e = fill_common(p, g, true, true)) return e;
  p.w = w; p.bias = nullptr; p.out = dz; p.thin = x; p.act = PCG_ACT_NONE; p.slope = 0.f;
  RowsP rp{};
  size_t patch_bytes = 0;
  rows_plan(p, rp, &patch_bytes);
  const size_t smem = lds_weight_bytes(p), two = 2 * ((patch_bytes + 15) & ~(size_t)15), sm = smem > two ? smem : two;
  const uint32_t thin_bytes = (uint32_t)((int64_t)p.B * p.TH * p.TW * p.Cs * 4);
  const unsigned blocks = rows_expand_blocks(rp.nunits) < THIN_BN_BLOCKS ? rows_expand_blocks(rp.nunits) : THIN_BN_BLOCKS;
  double* partial = (double*)ws;
  float* coef = reinterpret_cast<float*>(partial + (size_t)THIN_BN_BLOCKS * 2 * p.C);
  ThinBnBwd bn{z, mean, invstd, gamma, beta, coef, act_neg_of(act, slope), partial};
  static const int mfma_on = [] { const char* e = getenv("PCG_EXPAND_MFMA"); return e ? atoi(e) : 1; }();      // A/B switch (with the forward's)
  if (mfma_on && p.C == 64)       // the sums pass on the matrix cores
    hipLaunchKernelGGL((thin_rows_expand_bn_sums_mfma_kernel<4, 4>), dim3(blocks), dim3(256), two, s, p, rp, thin_bytes, bn);
  else
    hipLaunchKernelGGL((thin_rows_expand_bn_kernel<4, 4, 1, false>), dim3(blocks), dim3(256), sm, s, p, rp, thin_bytes, bn);
  if (int e = launch_status("thin_rows_expand_bn_kernel(sums)")) return e;
  if (int e = launch_bn_bwd_finalize(partial, (int)blocks, (int64_t)g->B * g->OH * g->OW, p.C, gamma, invstd, coef, dgamma, dbeta, accumulate, s)) return e;
  const unsigned ablocks = rows_expand_blocks(rp.nunits);
  hipLaunchKernelGGL((thin_rows_expand_bn_kernel<4, 4, 1, true>), dim3(ablocks), dim3(256), sm, s, p, rp, thin_bytes, bn);
  return launch_status("thin_rows_expand_bn_kernel(apply)");
}

// a Cin-thin layer whose wide (dy-side) operand can carry an input transform: grad-input through the matrix-core tap-dot, weight
// gradient through the row-block kernel
bool thin_conv_xf_ok(const pcg_conv_geom* g) {
  if (!thin_is_cin(g) || g->Cin != 1 || g->Cout % 64 || g->Cout > TD64_MAX_C) return false;
  ThinP p{};
  if (fill_common(p, g, true, false) != PCG_OK || !fast_ok(p) || reduce_scratch_bytes(p) == 0) return false;
  ThinP q{};
  RowsP rp{};
  size_t patch_bytes = 0;
  return fill_common(q, g, true, true) == PCG_OK && rows_wgrad_ok(q, g, rp, &patch_bytes);
}

int thin_conv_dgrad(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, float* dx, void* ws,
                    size_t ws_bytes, hipStream_t s, int act, float slope, const ThinXf* xf) {
  ThinP p{};
  p.act = act; p.slope = slope;
  const bool cin_thin = thin_is_cin(g);
  if (int e = fill_common(p, g, cin_thin, /*iter_on_output=*/false)) return e;
  p.w = w; p.bias = bias_x; p.out = dx;
  if (xf && xf->scale) {
    PCG_REQUIRE(thin_conv_xf_ok(g), "thin conv grad-input: this geometry takes no input transform (thin_conv_xf_ok)");
    p.xf_scale = xf->scale; p.xf_shift = xf->shift; p.xf_neg = xf->neg;
  }
  if (cin_thin) { p.wide = dy; return launch_reduce(p, ws, ws_bytes, s); }  // dx thin
  p.thin = dy;                                                              // dx wide
  const bool fuse = act_is_cheap(act);
  p.act = fuse ? act : PCG_ACT_NONE; p.slope = act_neg_of(p.act, slope);
  if (full_window(g) && bias_x == nullptr && (((uintptr_t)dx | (uintptr_t)w) & 15) == 0) {
    const int L4 = g->KH * g->KW * g->Cin / 4;
    hipLaunchKernelGGL(thin_full_outer_kernel, dim3((unsigned)((L4 + 255) / 256), (unsigned)((g->B + FULL_ROWS_DX - 1) / FULL_ROWS_DX)), dim3(256), 0, s, dy,
                       reinterpret_cast<const float4*>(w), reinterpret_cast<float4*>(dx), g->B, L4, p.slope);
    if (int e = launch_status("thin_full_outer_kernel")) return e;
    return fuse ? PCG_OK : pcg_act_fwd(dx, (int64_t)g->B * g->IH * g->IW * g->Cin, act, slope, dx, (pcg_stream_t)s);
  }
  if (int e = launch_expand(p, s)) return e;
  return fuse ? PCG_OK : pcg_act_fwd(dx, (int64_t)g->B * g->IH * g->IW * g->Cin, act, slope, dx, (pcg_stream_t)s);
}

// grad-input of a Cout-thin convolution (dx wide) times the activation derivative of the layer below, read from its activated
// output a_below (CounteRGAN: conv_out's grad-input arriving at conv_mid's LeakyReLU, models/generator.py:78-80 backward)
bool thin_conv_dgrad_mask_ok(const pcg_conv_geom* g) {
  if (!thin_is_cout(g)) return false;
  ThinP p{};
  if (fill_common(p, g, false, false) != PCG_OK) return false;
  const bool k44 = p.KH == 4 && p.KW == 4, k33 = p.KH == 3 && p.KW == 3;
  RowsP rp{};
  size_t patch_bytes = 0;
  return (k44 || k33) && (k33 || p.Cs == 1) && lds_weight_bytes(p) <= 64 * 1024 && rows_plan(p, rp, &patch_bytes);
}
int thin_conv_dgrad_mask(const pcg_conv_geom* g, const float* dy, const float* w, const float* a_below, int act, float slope, float* dx,
                         hipStream_t s) {
  PCG_REQUIRE(thin_conv_dgrad_mask_ok(g), "thin grad-input with mask: only the row-block forms (k3 / k4 with a thin output)");
  ThinP p{};
  if (int e = fill_common(p, g, false, /*iter_on_output=*/false)) return e;
  p.w = w; p.bias = nullptr; p.out = dx; p.thin = dy; p.act = PCG_ACT_NONE; p.slope = 0.f;
  p.mask_src = a_below; p.mask_neg = act_neg_of(act, slope);
  return launch_expand(p, s);
}

int launch_bn_bwd_finalize_g(const double* partial, int rows_per_group, int groups, int64_t rows_g, int C, const float* gamma, const float* invstd,
                             float* coef, float* dgamma, float* dbeta, int accumulate, hipStream_t s);

// full-window grad-input through the BatchNorm backward below it (thin_full_bn_*_kernel); groups side-by-side batches of B / groups samples
bool thin_conv_dgrad_bnbwd_full_ok(const pcg_conv_geom* g, int groups) {
  if (!full_window(g) || groups < 1 || groups > 8 || g->B % groups) return false;
  const int HW = g->KH * g->KW, CQ = g->Cin / 4;
  if (g->Cin % 4 || HW > 256 || 256 % HW) return false;
  const int CQB = 256 / HW;
  return CQ % CQB == 0 && (g->B / groups) % FULLBN_CHUNK == 0 && g->B / FULLBN_CHUNK <= 65535;
}
size_t thin_conv_dgrad_bnbwd_full_workspace_bytes(const pcg_conv_geom* g, int groups) {
  return (size_t)(g->B / FULLBN_CHUNK) * 2 * g->Cin * sizeof(double) + (size_t)groups * 3 * g->Cin * sizeof(float);
}
int thin_conv_dgrad_bnbwd_full(const pcg_conv_geom* g, const float* dy, const float* w, const float* z, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, int act, float slope, float* dz, float* dgamma, float* dbeta, int accumulate,
                               int groups, void* ws, size_t ws_bytes, hipStream_t s) {
  PCG_REQUIRE(thin_conv_dgrad_bnbwd_full_ok(g, groups), "full-window grad-input + BatchNorm backward: geometry / batch not eligible");
  PCG_REQUIRE(ws && ws_bytes >= thin_conv_dgrad_bnbwd_full_workspace_bytes(g, groups), "full-window grad-input + BatchNorm backward: workspace too small");
  const int C = g->Cin, HW = g->KH * g->KW, L4 = HW * C / 4, nchunks = g->B / FULLBN_CHUNK, Bg = g->B / groups;
  double* partial = (double*)ws;
  float* coef = reinterpret_cast<float*>(partial + (size_t)nchunks * 2 * C);
  FullBn q{dy, reinterpret_cast<const float4*>(w), reinterpret_cast<const float4*>(z), mean, invstd, gamma, beta, coef, act_neg_of(act, slope), C, HW, Bg};
  hipLaunchKernelGGL(thin_full_bn_sums_kernel, dim3((unsigned)((C / 4) / (256 / HW)), (unsigned)nchunks), dim3(256), 0, s, q, partial);
  if (int e = launch_status("thin_full_bn_sums_kernel")) return e;
  if (groups == 1) {
    if (int e = launch_bn_bwd_finalize(partial, nchunks, (int64_t)g->B * HW, C, gamma, invstd, coef, dgamma, dbeta, accumulate, s)) return e;
  } else {
    if (int e = launch_bn_bwd_finalize_g(partial, nchunks / groups, groups, (int64_t)Bg * HW, C, gamma, invstd, coef, dgamma, dbeta, accumulate, s)) return e;
  }
  hipLaunchKernelGGL(thin_full_bn_apply_kernel, dim3((unsigned)((L4 + 255) / 256), (unsigned)((g->B + FULLBN_ROWS - 1) / FULLBN_ROWS)), dim3(256), 0, s, q,
                     reinterpret_cast<float4*>(dz), g->B, L4);
  return launch_status("thin_full_bn_apply_kernel");
}

size_t thin_conv_fwd_workspace_bytes(const pcg_conv_geom* g) {
  ThinP p{};
  const bool cin_thin = thin_is_cin(g);
  if (cin_thin || fill_common(p, g, cin_thin, true) != PCG_OK) return 0;
  return reduce_scratch_bytes(p);
}
size_t thin_conv_dgrad_workspace_bytes(const pcg_conv_geom* g) {
  ThinP p{};
  const bool cin_thin = thin_is_cin(g);
  if (!cin_thin || fill_common(p, g, cin_thin, false) != PCG_OK) return 0;
  return reduce_scratch_bytes(p);
}

size_t thin_conv_wgrad_workspace_bytes(const pcg_conv_geom* g) {
  const bool cin_thin = thin_is_cin(g);
  const int C = cin_thin ? g->Cout : g->Cin;
  if (C % 4 != 0 || C / 4 > 256 || !is_pow2(C / 4)) return 0;
  const int npix = cin_thin ? g->B * g->OH * g->OW : g->B * g->IH * g->IW;
  const ThinWgradPlan wp = plan_thin_wgrad(npix, C);
  size_t blocks = (size_t)wp.nblocks;
  ThinP p{};
  RowsP rp{};
  size_t patch_bytes = 0;
  if (fill_common(p, g, cin_thin, cin_thin) == PCG_OK && rows_wgrad_ok(p, g, rp, &patch_bytes) && (size_t)rows_wgrad_blocks(rp) > blocks)
    blocks = (size_t)rows_wgrad_blocks(rp);
  if (full_window(g) && (size_t)((g->B + FULL_ROWS_DW - 1) / FULL_ROWS_DW) > blocks) blocks = (size_t)((g->B + FULL_ROWS_DW - 1) / FULL_ROWS_DW);
  return blocks * (size_t)g->Cout * g->KH * g->KW * g->Cin * sizeof(float);
}

int thin_conv_wgrad(const pcg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate, void* ws,
                    size_t ws_bytes, hipStream_t s, const ThinXf* xf, const ThinBnIn* bnin) {
  ThinP p{};
  const bool cin_thin = thin_is_cin(g);
  PCG_REQUIRE(!(bnin && bnin->mean) || (thin_conv_bnin_full_ok(g, bnin->groups) && (((uintptr_t)x | (uintptr_t)ws) & 15) == 0),
              "thin conv weight gradient: a BatchNorm input needs the full-window form (thin_conv_bnin_full_ok) and 16-byte aligned tensors");
  const bool has_xf = xf && xf->scale;
  PCG_REQUIRE(!has_xf || thin_conv_xf_ok(g), "thin conv weight gradient: this geometry takes no input transform (thin_conv_xf_ok)");
  if (has_xf) { p.xf_scale = xf->scale; p.xf_shift = xf->shift; p.xf_neg = xf->neg; }
  if (full_window(g) && (((uintptr_t)x | (uintptr_t)ws) & 15) == 0) {
    const int L = g->KH * g->KW * g->Cin, nslabs = (g->B + FULL_ROWS_DW - 1) / FULL_ROWS_DW;
    const size_t need = (size_t)nslabs * L * sizeof(float);
    if (ws == nullptr || ws_bytes < need) {
      set_error("thin conv wgrad: workspace %zu B < required %zu B", ws_bytes, need);
      return PCG_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(thin_full_wgrad_kernel, dim3((unsigned)((L / 4 + 255) / 256), (unsigned)nslabs), dim3(256), 0, s, reinterpret_cast<const float4*>(x), dy,
                       reinterpret_cast<float4*>(ws), g->B, L / 4, full_xf_of(g, bnin));
    if (int e = launch_status("thin_full_wgrad_kernel")) return e;
    return launch_slab_reduce((const float*)ws, dw, (size_t)L, (size_t)L, nslabs, accumulate, s, true);
  }
  // iterate over the wide tensor's pixels: dy (output grid) when Cin is thin, x (input grid) when Cout is thin
  if (int e = fill_common(p, g, cin_thin, /*iter_on_output=*/cin_thin)) return e;
  PCG_REQUIRE(p.C / 4 <= 256 && is_pow2(p.C / 4), "thin conv wgrad: wide channel count %d must be 4*2^k <= 1024", p.C);
  if (cin_thin) { p.wide = dy; p.thin = x; } else { p.wide = x; p.thin = dy; }
  const ThinWgradPlan wp = plan_thin_wgrad(p.npix, p.C);
  const int wn = g->Cout * g->KH * g->KW * g->Cin;
  RowsP rp{};
  size_t patch_bytes = 0;
  const bool rows = rows_wgrad_ok(p, g, rp, &patch_bytes);
  PCG_REQUIRE(!has_xf || rows, "thin conv weight gradient: an input transform needs the row-block form");
  const int nslabs = rows ? rows_wgrad_blocks(rp) : wp.nblocks;
  const size_t need = (size_t)nslabs * wn * sizeof(float);
  if (ws == nullptr || ws_bytes < need) {
    set_error("thin conv wgrad: workspace %zu B < required %zu B", ws_bytes, need);
    return PCG_ERR_WORKSPACE;
  }
  float* slab = (float*)ws;
  const uint32_t thin_bytes = (uint32_t)((int64_t)p.B * p.TH * p.TW * p.Cs * 4);
  if (rows) {
    const int upb = (rp.nunits + nslabs - 1) / nslabs;
    const int nb = (rp.nunits + upb - 1) / upb;      // blocks that own at least one unit (<= nslabs)
    static const int mfma_on = [] { const char* e = getenv("PCG_EXPAND_MFMA"); return e ? atoi(e) : 1; }();      // A/B switch (with the expand forms')
    if (mfma_on && p.C == 64 && ((g->KH == 4 && p.Cs == 1) || g->KH == 3) && (((uintptr_t)p.wide) & 3) == 0) {      // the matrix-core form
      const size_t sm2 = 2 * ((patch_bytes + 15) & ~(size_t)15);
      if (g->KH == 4) hipLaunchKernelGGL((thin_rows_wgrad_mfma_kernel<4, 4, 1>), dim3(nb), dim3(256), sm2, s, p, rp, slab, wn, upb, thin_bytes);
      else if (p.Cs == 1) hipLaunchKernelGGL((thin_rows_wgrad_mfma_kernel<3, 3, 1>), dim3(nb), dim3(256), sm2, s, p, rp, slab, wn, upb, thin_bytes);
      else if (p.Cs == 2) hipLaunchKernelGGL((thin_rows_wgrad_mfma_kernel<3, 3, 2>), dim3(nb), dim3(256), sm2, s, p, rp, slab, wn, upb, thin_bytes);
      else hipLaunchKernelGGL((thin_rows_wgrad_mfma_kernel<3, 3, 3>), dim3(nb), dim3(256), sm2, s, p, rp, slab, wn, upb, thin_bytes);
      if (int e = launch_status("thin_rows_wgrad_mfma_kernel")) return e;
      return launch_slab_reduce(slab, dw, (size_t)wn, (size_t)wn, nb, accumulate, s, true);
    }
#define PCG_ROWS_WGRAD_CASE(KH_, KW_, CS_)                                                                                       \
    if (g->KH == KH_ && p.Cs == CS_) {                                                                                            \
      hipLaunchKernelGGL((thin_rows_wgrad_kernel<KH_, KW_, CS_>), dim3(nb), dim3(256), 2 * ((patch_bytes + 15) & ~(size_t)15), s, p, rp, slab, wn, upb, \
                         thin_bytes);                                                                                             \
      if (int e = launch_status("thin_rows_wgrad_kernel")) return e;                                                              \
      return launch_slab_reduce(slab, dw, (size_t)wn, (size_t)wn, nb, accumulate, s, true);                                             \
    }
    PCG_ROWS_WGRAD_CASE(4, 4, 1)
    PCG_ROWS_WGRAD_CASE(3, 3, 1)
    PCG_ROWS_WGRAD_CASE(3, 3, 2)
    PCG_ROWS_WGRAD_CASE(3, 3, 3)
#undef PCG_ROWS_WGRAD_CASE
  }
#define PCG_THIN_WGRAD_CASE(KH_, KW_, CS_)                                                                          \
  if (g->KH == KH_ && g->KW == KW_ && p.Cs == CS_) {                                                                 \
    hipLaunchKernelGGL((thin_wgrad_kernel<KH_, KW_, CS_>), dim3(wp.nblocks), dim3(256), 0, s, p, slab, wp.ppb, wn, thin_bytes); \
  } else
  PCG_THIN_WGRAD_CASE(4, 4, 1)
  PCG_THIN_WGRAD_CASE(3, 3, 1)
  PCG_THIN_WGRAD_CASE(3, 3, 2)
  PCG_THIN_WGRAD_CASE(3, 3, 3)
  PCG_THIN_WGRAD_CASE(1, 1, 1)
  PCG_THIN_WGRAD_CASE(1, 1, 2)
  PCG_THIN_WGRAD_CASE(1, 1, 3)
  {
    set_error("thin conv wgrad: kernel %dx%d with %d thin channels has no instantiation", g->KH, g->KW, p.Cs);
    return PCG_ERR_UNSUPPORTED;
  }
#undef PCG_THIN_WGRAD_CASE
  if (int e = launch_status("thin_wgrad_kernel")) return e;
  return launch_slab_reduce(slab, dw, (size_t)wn, (size_t)wn, wp.nblocks, accumulate, s, true);
}

int slab_defer_begin(hipStream_t s) {
  SlabDefer& d = g_slab_defer;
  PCG_REQUIRE(!d.on, "pcg_slab_defer_begin: already deferring on this thread (no nesting)");
  d.on = true; d.stream = s; d.n = 0; d.max_blocks = 0;
  return PCG_OK;
}
int slab_defer_flush(hipStream_t s) {
  SlabDefer& d = g_slab_defer;
  if (!d.on) return PCG_OK;
  d.on = false;
  PCG_REQUIRE(s == d.stream, "pcg_slab_defer_flush: called on another stream than pcg_slab_defer_begin");
  return slab_defer_launch();
}
int slab_defer_pending() { return g_slab_defer.on ? g_slab_defer.n : -1; }

}  // namespace pcg
