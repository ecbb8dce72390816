// instnorm.hip — InstanceNorm2d(affine=True) of the WGAN-GP critic (conditional_gan/mnist/mnist_wgan_conditional.py:88,91,94)
// on NHWC activations [B][HW][C]: forward (+ fused LeakyReLU), backward, and the backward OF the backward that the gradient
// penalty (:147-150, autograd.grad(..., create_graph=True)) differentiates through.  Statistics are per (sample, channel)
// over HW <= 169 positions: HBM-bound elementwise passes with a tiny reduction, one block per (sample, 64-channel group),
// 4 position-lanes x 64 channels, so every load is a 256-byte row segment.
//
// With xh = (x - mean) * invstd, a = dy - mean(dy), m = mean(a*xh):
//   backward          dx  = gamma*invstd * (a - xh*m)
//   backward-backward (cotangent r on dx; rt = r - mean(r), q = mean(r*xh), p = mean(r*a)):
//                     ddy = gamma*invstd * (rt - xh*q)                                   (cotangent reaching dy)
//                     ez  = -gamma*invstd^2 * (q*a + m*rt + (p - 3*m*q)*xh)              (cotangent reaching x)
//                     dgamma += invstd * HW * (p - m*q)
// (derivation checked against torch autograd in float64: tests/test_hip_wgan.py)
#include <initializer_list>
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int IN_THREADS = 256;

// A block owns one sample and `tc` channels; a thread owns V consecutive channels (V = 4: 16-byte loads, a row of 64 channels is
// 16 threads and the block's 16 position-lanes cover 16 rows per step; V = 1 for channel counts that are not multiples of 4).
// r03: the scalar form (4 bytes per lane, 4 position-lanes) moved 2.2 TB/s on the B = 768 first stage; these passes carry the
// LeakyReLU' / add / bias-partial work of their neighbours now, so their rate is what the critic's element-wise time is made of.
template <int V> struct Vec { float v[V]; };
template <int V> __device__ __forceinline__ Vec<V> ldv(const float* p) {
  Vec<V> r;
  if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
  else r.v[0] = *p;
  return r;
}
template <int V> __device__ __forceinline__ void stv(float* p, const Vec<V>& a) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
  else *p = a.v[0];
}

struct Lanes { int tr, th, ch, lane, c; bool on; };   // tr: threads per row, th: position-lanes, c: first channel of this thread

template <int V>
__device__ __forceinline__ Lanes lanes_of(int C, int tc) {
  Lanes l;
  l.tr = tc / V; l.th = IN_THREADS / l.tr;
  l.ch = threadIdx.x % l.tr; l.lane = threadIdx.x / l.tr;
  l.c = blockIdx.y * tc + l.ch * V;
  l.on = l.c < C;
  return l;
}

// sum K per-thread values over the position-lanes of each channel, in lane order; every thread gets the totals
template <int K>
__device__ __forceinline__ void lane_sum(float (&v)[K], const Lanes& l, float* smem /* [K][256] */) {
#pragma unroll
  for (int k = 0; k < K; ++k) smem[k * IN_THREADS + threadIdx.x] = v[k];
  __syncthreads();
  if (l.lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
      for (int j = 0; j < l.th; ++j) s += smem[k * IN_THREADS + j * l.tr + l.ch];
      smem[k * IN_THREADS + l.ch] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = smem[k * IN_THREADS + l.ch];
  __syncthreads();
}

template <int V>
__global__ void __launch_bounds__(IN_THREADS) instnorm_fwd_kernel(const float* __restrict__ x, int HW, int C, int tc,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, int act, float slope, float* __restrict__ y,
                                                                 float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  __shared__ float smem[V * IN_THREADS];
  const Lanes l = lanes_of<V>(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  float s[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s[e] = 0.f;
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const Vec<V> xv = ldv<V>(x + base + (size_t)p * C + l.c);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += xv.v[e];
    }
  lane_sum<V>(s, l, smem);
  float mean[V], q[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { mean[e] = s[e] / (float)HW; q[e] = 0.f; }
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const Vec<V> xv = ldv<V>(x + base + (size_t)p * C + l.c);
#pragma unroll
      for (int e = 0; e < V; ++e) { const float d = xv.v[e] - mean[e]; q[e] = fmaf(d, d, q[e]); }
    }
  lane_sum<V>(q, l, smem);
  if (!l.on) return;
  float g[V], b[V];
  const Vec<V> gv = ldv<V>(gamma + l.c), bv = ldv<V>(beta + l.c);
  Vec<V> mo, io;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float invstd = rsqrtf(q[e] / (float)HW + eps);
    mo.v[e] = mean[e]; io.v[e] = invstd;
    g[e] = gv.v[e] * invstd; b[e] = bv.v[e];
  }
  if (l.lane == 0) { stv<V>(mean_out + (size_t)blockIdx.x * C + l.c, mo); stv<V>(invstd_out + (size_t)blockIdx.x * C + l.c, io); }
  for (int p = l.lane; p < HW; p += l.th) {
    const size_t i = base + (size_t)p * C + l.c;
    const Vec<V> xv = ldv<V>(x + i);
    Vec<V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = act_apply(fmaf(xv.v[e] - mean[e], g[e], b[e]), act, slope);
    stv<V>(y + i, o);
  }
}

// One launch for the chain LeakyReLU' -> InstanceNorm' of a critic stage (mnist_wgan_conditional.py:88-95 backward):
//   act_y  (nullable) the activation's OUTPUT: dy is first multiplied by lrelu'(.) (sign of the output = sign of the input), the
//          separate act_bwd pass over the tensor disappears; dn_out (nullable) keeps that masked gradient for the double backward
//   addend (nullable) added to the stored dx (the cotangent ez that the gradient penalty's second pass adds at this tensor)
//   dxsum_part (nullable) [B][C]: sum over HW of the stored dx — the conv bias gradient's per-sample partial, so the bias
//          gradient needs no pass of its own over dx (pcg_rowsum3 reduces the three partial arrays over B in one launch)
template <int V>
__global__ void __launch_bounds__(IN_THREADS) instnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act_y, float neg,
                                                                 const float* __restrict__ x, int HW, int C,
                                                                 int tc, const float* __restrict__ mean_in,
                                                                 const float* __restrict__ invstd_in, const float* __restrict__ gamma,
                                                                 float* __restrict__ dn_out, const float* __restrict__ addend,
                                                                 float* __restrict__ dx, float* __restrict__ dgamma_part,
                                                                 float* __restrict__ dbeta_part, float* __restrict__ dxsum_part) {
  __shared__ float smem[2 * V * IN_THREADS];
  const Lanes l = lanes_of<V>(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  Vec<V> mean, invstd;
#pragma unroll
  for (int e = 0; e < V; ++e) { mean.v[e] = 0.f; invstd.v[e] = 0.f; }
  if (l.on) { mean = ldv<V>(mean_in + (size_t)blockIdx.x * C + l.c); invstd = ldv<V>(invstd_in + (size_t)blockIdx.x * C + l.c); }
  float s[2 * V];                          // [0..V): sum d, [V..2V): sum d*xh
#pragma unroll
  for (int e = 0; e < 2 * V; ++e) s[e] = 0.f;
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      Vec<V> d = ldv<V>(dy + i);
      const Vec<V> xv = ldv<V>(x + i);
      if (act_y) {
        const Vec<V> yv = ldv<V>(act_y + i);
#pragma unroll
        for (int e = 0; e < V; ++e) d.v[e] *= yv.v[e] > 0.f ? 1.f : neg;
        if (dn_out) stv<V>(dn_out + i, d);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) { s[e] += d.v[e]; s[V + e] = fmaf(d.v[e], (xv.v[e] - mean.v[e]) * invstd.v[e], s[V + e]); }
    }
  lane_sum<2 * V>(s, l, smem);
  if (l.on && l.lane == 0) {
    Vec<V> a, b;
#pragma unroll
    for (int e = 0; e < V; ++e) { a.v[e] = s[V + e]; b.v[e] = s[e]; }
    if (dgamma_part) stv<V>(dgamma_part + (size_t)blockIdx.x * C + l.c, a);
    if (dbeta_part) stv<V>(dbeta_part + (size_t)blockIdx.x * C + l.c, b);
  }
  if (!dx) return;                         // (uniform)
  const float inv_n = 1.f / (float)HW;
  float m1[V], m2[V], g[V], t[V];
  Vec<V> gv;
#pragma unroll
  for (int e = 0; e < V; ++e) gv.v[e] = 0.f;
  if (l.on) gv = ldv<V>(gamma + l.c);
#pragma unroll
  for (int e = 0; e < V; ++e) { m1[e] = s[e] * inv_n; m2[e] = s[V + e] * inv_n; g[e] = gv.v[e] * invstd.v[e]; t[e] = 0.f; }
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      // (dn_out, when kept, already holds the masked gradient of this thread's own elements)
      Vec<V> d = ldv<V>((act_y && dn_out) ? dn_out + i : dy + i);
      const Vec<V> xv = ldv<V>(x + i);
      if (act_y && !dn_out) {
        const Vec<V> yv = ldv<V>(act_y + i);
#pragma unroll
        for (int e = 0; e < V; ++e) d.v[e] *= yv.v[e] > 0.f ? 1.f : neg;
      }
      Vec<V> o;
#pragma unroll
      for (int e = 0; e < V; ++e) o.v[e] = g[e] * (d.v[e] - m1[e] - (xv.v[e] - mean.v[e]) * invstd.v[e] * m2[e]);
      if (addend) {
        const Vec<V> av = ldv<V>(addend + i);
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] += av.v[e];
      }
      stv<V>(dx + i, o);
#pragma unroll
      for (int e = 0; e < V; ++e) t[e] += o.v[e];
    }
  if (!dxsum_part) return;                 // (uniform)
  lane_sum<V>(t, l, smem);
  if (l.on && l.lane == 0) {
    Vec<V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = t[e];
    stv<V>(dxsum_part + (size_t)blockIdx.x * C + l.c, o);
  }
}

template <int V>
__global__ void __launch_bounds__(IN_THREADS) instnorm_bwd_bwd_kernel(const float* __restrict__ r, const float* __restrict__ dy,
                                                                     const float* __restrict__ x, int HW, int C, int tc,
                                                                     const float* __restrict__ mean_in, const float* __restrict__ invstd_in,
                                                                     const float* __restrict__ gamma, float* __restrict__ ddy,
                                                                     float* __restrict__ ez, float* __restrict__ dgamma_part,
                                                                     const float* __restrict__ act_y, float neg) {
  __shared__ float smem[4 * V * IN_THREADS];
  const Lanes l = lanes_of<V>(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  Vec<V> mean, invstd;
#pragma unroll
  for (int e = 0; e < V; ++e) { mean.v[e] = 0.f; invstd.v[e] = 0.f; }
  if (l.on) { mean = ldv<V>(mean_in + (size_t)blockIdx.x * C + l.c); invstd = ldv<V>(invstd_in + (size_t)blockIdx.x * C + l.c); }
  float s[4 * V];   // per channel e: [e] sum r, [V+e] sum r*xh, [2V+e] sum dy, [3V+e] sum dy*xh
#pragma unroll
  for (int e = 0; e < 4 * V; ++e) s[e] = 0.f;
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      const Vec<V> rv = ldv<V>(r + i), dv = ldv<V>(dy + i), xv = ldv<V>(x + i);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (xv.v[e] - mean.v[e]) * invstd.v[e];
        s[e] += rv.v[e]; s[V + e] = fmaf(rv.v[e], xh, s[V + e]); s[2 * V + e] += dv.v[e]; s[3 * V + e] = fmaf(dv.v[e], xh, s[3 * V + e]);
      }
    }
  lane_sum<4 * V>(s, l, smem);
  const float inv_n = 1.f / (float)HW;
  float mr[V], q[V], md[V], m[V], c[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { mr[e] = s[e] * inv_n; q[e] = s[V + e] * inv_n; md[e] = s[2 * V + e] * inv_n; m[e] = s[3 * V + e] * inv_n; c[e] = 0.f; }
  // sum (r - mean r)*(dy - mean dy): centred second pass, no cancellation
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      const Vec<V> rv = ldv<V>(r + i), dv = ldv<V>(dy + i);
#pragma unroll
      for (int e = 0; e < V; ++e) c[e] = fmaf(rv.v[e] - mr[e], dv.v[e] - md[e], c[e]);
    }
  lane_sum<V>(c, l, smem);
  if (!l.on) return;
  const Vec<V> gam = ldv<V>(gamma + l.c);
  float g1[V], g2[V], k3[V];
  Vec<V> dg;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float pp = c[e] * inv_n;                                             // mean(r * a)
    dg.v[e] = invstd.v[e] * (float)HW * (pp - m[e] * q[e]);
    g1[e] = gam.v[e] * invstd.v[e]; g2[e] = -gam.v[e] * invstd.v[e] * invstd.v[e]; k3[e] = pp - 3.f * m[e] * q[e];
  }
  if (l.lane == 0 && dgamma_part) stv<V>(dgamma_part + (size_t)blockIdx.x * C + l.c, dg);
  for (int p = l.lane; p < HW; p += l.th) {
    const size_t i = base + (size_t)p * C + l.c;
    const Vec<V> rv = ldv<V>(r + i), dv = ldv<V>(dy + i), xv = ldv<V>(x + i);
    Vec<V> o1, o2;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float xh = (xv.v[e] - mean.v[e]) * invstd.v[e], a = dv.v[e] - md[e], rt = rv.v[e] - mr[e];
      o1.v[e] = g1[e] * (rt - xh * q[e]);
      o2.v[e] = g2[e] * (q[e] * a + m[e] * rt + k3[e] * xh);
    }
    if (ddy) {      // act_y: the cotangent continues through the stage's LeakyReLU (its mask is the forward's), no act_bwd pass
      if (act_y) {
        const Vec<V> yv = ldv<V>(act_y + i);
#pragma unroll
        for (int e = 0; e < V; ++e) o1.v[e] *= yv.v[e] > 0.f ? 1.f : neg;
      }
      stv<V>(ddy + i, o1);
    }
    if (ez) stv<V>(ez + i, o2);
  }
}

// torch Flatten of an NCHW [B,C,H,W] tensor from / to our NHWC activation: flat[b][c*HW + p] <-> act[b][p*C + c]
__global__ void __launch_bounds__(256) nhwc_to_nchw_flat_kernel(const float* __restrict__ src, float* __restrict__ dst, int HW, int C,
                                                                size_t n, int inverse) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t b = i / ((size_t)HW * C);
    const int rem = (int)(i - b * (size_t)HW * C);
    const int p = rem / C, c = rem - p * C;                      // i indexes the NHWC side (coalesced there)
    const size_t j = b * (size_t)HW * C + (size_t)c * HW + p;    // NCHW-flat side
    if (inverse) dst[i] = src[j]; else dst[j] = src[i];
  }
}

// WGAN-GP: interpolates = alpha*real + (1-alpha)*fake (:147), alpha per sample
__global__ void __launch_bounds__(256) interpolate_kernel(const float* __restrict__ alpha, const float* __restrict__ real,
                                                          const float* __restrict__ fake, float* __restrict__ out, int per_sample,
                                                          size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float a = alpha[i / per_sample];
    out[i] = a * real[i] + (1.f - a) * fake[i];
  }
}

// the batched critic pass's input in one launch (r04): x3 = [real | fake | alpha*real + (1-alpha)*fake], B samples each — the
// interpolation and the two staging copies of torch.cat
__global__ void __launch_bounds__(256) interpolate_stack_kernel(const float* __restrict__ alpha, const float* __restrict__ real,
                                                                const float* __restrict__ fake, float* __restrict__ x3, int per_sample,
                                                                size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float a = alpha[i / per_sample], r = real[i], f = fake[i];
    x3[i] = r; x3[n + i] = f; x3[2 * n + i] = a * r + (1.f - a) * f;
  }
}

// gradient penalty (:150): pen = lambda * mean_b (||g_b|| - 1)^2; one block per sample computes the norm; dpen/dg_b =
// grad_out * 2*lambda/B * (||g_b|| - 1) * g_b / ||g_b||
__global__ void __launch_bounds__(256) row_norm_kernel(const float* __restrict__ g, int per_sample, float* __restrict__ norms) {
  __shared__ float red[256];
  const float* row = g + (size_t)blockIdx.x * per_sample;
  float s = 0.f;
  for (int i = threadIdx.x; i < per_sample; i += 256) s = fmaf(row[i], row[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) norms[blockIdx.x] = sqrtf(red[0]);
}
__global__ void gp_finish_kernel(const float* __restrict__ norms, int B, float lambda, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) { const double d = (double)norms[b] - 1.0; s += d * d; }
    out[0] = (float)((double)lambda * s / (double)B);
  }
}
__global__ void __launch_bounds__(256) gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ norms,
                                                     const float* __restrict__ grad_out, float scale, int per_sample, size_t n,
                                                     float* __restrict__ dg) {
  const float go = (grad_out ? grad_out[0] : 1.f) * scale;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float nb = norms[i / per_sample];
    dg[i] = nb > 0.f ? go * (nb - 1.f) * g[i] / nb : 0.f;     // torch's norm backward gives 0 at ||g|| = 0 as well
  }
}

// ---- register-resident forms (r03): HW <= 16 * NP positions per (sample, 64-channel group) -------------------------------------
// A block's slice of one sample is HW x 64 channels x 4 B <= 48 KB (HW = 169): each thread keeps its <= NP float4 of every tensor it
// needs twice in registers, so x / dy / r are read from HBM ONCE instead of two or three times.  Same lanes, same order of every sum
// as the streaming kernels above: bit-identical results.  (r03 critic trace: the three InstanceNorm kernels are 0.6 ms of a 6 ms
// update, all of it memory passes.)
template <int NP>
__global__ void __launch_bounds__(IN_THREADS) instnorm_fwd_reg_kernel(const float* __restrict__ x, int HW, int C, int tc,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float eps, int act, float slope, float* __restrict__ y,
                                                                     float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  constexpr int V = 4;
  __shared__ float smem[V * IN_THREADS];
  const Lanes l = lanes_of<V>(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  Vec<V> xr[NP];
  float s[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (l.on && p < HW) {
      xr[k] = ldv<V>(x + base + (size_t)p * C + l.c);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += xr[k].v[e];
    }
  }
  lane_sum<V>(s, l, smem);
  float mean[V], q[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { mean[e] = s[e] / (float)HW; q[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (l.on && p < HW) {
#pragma unroll
      for (int e = 0; e < V; ++e) { const float d = xr[k].v[e] - mean[e]; q[e] = fmaf(d, d, q[e]); }
    }
  }
  lane_sum<V>(q, l, smem);
  if (!l.on) return;
  float g[V], b[V];
  const Vec<V> gv = ldv<V>(gamma + l.c), bv = ldv<V>(beta + l.c);
  Vec<V> mo, io;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float invstd = rsqrtf(q[e] / (float)HW + eps);
    mo.v[e] = mean[e]; io.v[e] = invstd;
    g[e] = gv.v[e] * invstd; b[e] = bv.v[e];
  }
  if (l.lane == 0) { stv<V>(mean_out + (size_t)blockIdx.x * C + l.c, mo); stv<V>(invstd_out + (size_t)blockIdx.x * C + l.c, io); }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (p < HW) {
      Vec<V> o;
#pragma unroll
      for (int e = 0; e < V; ++e) o.v[e] = act_apply(fmaf(xr[k].v[e] - mean[e], g[e], b[e]), act, slope);
      stv<V>(y + base + (size_t)p * C + l.c, o);
    }
  }
}

template <int NP>
__global__ void __launch_bounds__(IN_THREADS) instnorm_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ act_y, float neg,
                                                                     const float* __restrict__ x, int HW, int C,
                                                                     int tc, const float* __restrict__ mean_in,
                                                                     const float* __restrict__ invstd_in, const float* __restrict__ gamma,
                                                                     float* __restrict__ dn_out, const float* __restrict__ addend,
                                                                     float* __restrict__ dx, float* __restrict__ dgamma_part,
                                                                     float* __restrict__ dbeta_part, float* __restrict__ dxsum_part) {
  constexpr int V = 4;
  __shared__ float smem[2 * V * IN_THREADS];
  const Lanes l = lanes_of<V>(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  Vec<V> mean, invstd;
#pragma unroll
  for (int e = 0; e < V; ++e) { mean.v[e] = 0.f; invstd.v[e] = 0.f; }
  if (l.on) { mean = ldv<V>(mean_in + (size_t)blockIdx.x * C + l.c); invstd = ldv<V>(invstd_in + (size_t)blockIdx.x * C + l.c); }
  Vec<V> dr[NP], xh[NP];                   // the masked gradient and the normalised activation of this thread's elements
  float s[2 * V];
#pragma unroll
  for (int e = 0; e < 2 * V; ++e) s[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (l.on && p < HW) {
      const size_t i = base + (size_t)p * C + l.c;
      Vec<V> d = ldv<V>(dy + i);
      const Vec<V> xv = ldv<V>(x + i);
      if (act_y) {
        const Vec<V> yv = ldv<V>(act_y + i);
#pragma unroll
        for (int e = 0; e < V; ++e) d.v[e] *= yv.v[e] > 0.f ? 1.f : neg;
        if (dn_out) stv<V>(dn_out + i, d);
      }
      dr[k] = d;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        xh[k].v[e] = (xv.v[e] - mean.v[e]) * invstd.v[e];
        s[e] += d.v[e]; s[V + e] = fmaf(d.v[e], xh[k].v[e], s[V + e]);
      }
    }
  }
  lane_sum<2 * V>(s, l, smem);
  if (l.on && l.lane == 0) {
    Vec<V> a, b;
#pragma unroll
    for (int e = 0; e < V; ++e) { a.v[e] = s[V + e]; b.v[e] = s[e]; }
    if (dgamma_part) stv<V>(dgamma_part + (size_t)blockIdx.x * C + l.c, a);
    if (dbeta_part) stv<V>(dbeta_part + (size_t)blockIdx.x * C + l.c, b);
  }
  if (!dx) return;                         // (uniform)
  const float inv_n = 1.f / (float)HW;
  float m1[V], m2[V], g[V], t[V];
  Vec<V> gv;
#pragma unroll
  for (int e = 0; e < V; ++e) gv.v[e] = 0.f;
  if (l.on) gv = ldv<V>(gamma + l.c);
#pragma unroll
  for (int e = 0; e < V; ++e) { m1[e] = s[e] * inv_n; m2[e] = s[V + e] * inv_n; g[e] = gv.v[e] * invstd.v[e]; t[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (l.on && p < HW) {
      const size_t i = base + (size_t)p * C + l.c;
      Vec<V> o;
#pragma unroll
      for (int e = 0; e < V; ++e) o.v[e] = g[e] * (dr[k].v[e] - m1[e] - xh[k].v[e] * m2[e]);
      if (addend) {
        const Vec<V> av = ldv<V>(addend + i);
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] += av.v[e];
      }
      stv<V>(dx + i, o);
#pragma unroll
      for (int e = 0; e < V; ++e) t[e] += o.v[e];
    }
  }
  if (!dxsum_part) return;                 // (uniform)
  lane_sum<V>(t, l, smem);
  if (l.on && l.lane == 0) {
    Vec<V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = t[e];
    stv<V>(dxsum_part + (size_t)blockIdx.x * C + l.c, o);
  }
}

template <int NP>
__global__ void __launch_bounds__(IN_THREADS) instnorm_bwd_bwd_reg_kernel(const float* __restrict__ r, const float* __restrict__ dy,
                                                                         const float* __restrict__ x, int HW, int C, int tc,
                                                                         const float* __restrict__ mean_in, const float* __restrict__ invstd_in,
                                                                         const float* __restrict__ gamma, float* __restrict__ ddy,
                                                                         float* __restrict__ ez, float* __restrict__ dgamma_part,
                                                                         const float* __restrict__ act_y, float neg) {
  constexpr int V = 4;
  __shared__ float smem[4 * V * IN_THREADS];
  const Lanes l = lanes_of<V>(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  Vec<V> mean, invstd;
#pragma unroll
  for (int e = 0; e < V; ++e) { mean.v[e] = 0.f; invstd.v[e] = 0.f; }
  if (l.on) { mean = ldv<V>(mean_in + (size_t)blockIdx.x * C + l.c); invstd = ldv<V>(invstd_in + (size_t)blockIdx.x * C + l.c); }
  Vec<V> rr[NP], dd[NP], xh[NP];
  float s[4 * V];
#pragma unroll
  for (int e = 0; e < 4 * V; ++e) s[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (l.on && p < HW) {
      const size_t i = base + (size_t)p * C + l.c;
      rr[k] = ldv<V>(r + i); dd[k] = ldv<V>(dy + i);
      const Vec<V> xv = ldv<V>(x + i);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        xh[k].v[e] = (xv.v[e] - mean.v[e]) * invstd.v[e];
        s[e] += rr[k].v[e]; s[V + e] = fmaf(rr[k].v[e], xh[k].v[e], s[V + e]);
        s[2 * V + e] += dd[k].v[e]; s[3 * V + e] = fmaf(dd[k].v[e], xh[k].v[e], s[3 * V + e]);
      }
    }
  }
  lane_sum<4 * V>(s, l, smem);
  const float inv_n = 1.f / (float)HW;
  float mr[V], q[V], md[V], m[V], c[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { mr[e] = s[e] * inv_n; q[e] = s[V + e] * inv_n; md[e] = s[2 * V + e] * inv_n; m[e] = s[3 * V + e] * inv_n; c[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (l.on && p < HW) {
#pragma unroll
      for (int e = 0; e < V; ++e) c[e] = fmaf(rr[k].v[e] - mr[e], dd[k].v[e] - md[e], c[e]);
    }
  }
  lane_sum<V>(c, l, smem);
  if (!l.on) return;
  const Vec<V> gam = ldv<V>(gamma + l.c);
  float g1[V], g2[V], k3[V];
  Vec<V> dg;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float pp = c[e] * inv_n;
    dg.v[e] = invstd.v[e] * (float)HW * (pp - m[e] * q[e]);
    g1[e] = gam.v[e] * invstd.v[e]; g2[e] = -gam.v[e] * invstd.v[e] * invstd.v[e]; k3[e] = pp - 3.f * m[e] * q[e];
  }
  if (l.lane == 0 && dgamma_part) stv<V>(dgamma_part + (size_t)blockIdx.x * C + l.c, dg);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = l.lane + k * l.th;
    if (p < HW) {
      const size_t i = base + (size_t)p * C + l.c;
      Vec<V> o1, o2;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float a = dd[k].v[e] - md[e], rt = rr[k].v[e] - mr[e];
        o1.v[e] = g1[e] * (rt - xh[k].v[e] * q[e]);
        o2.v[e] = g2[e] * (q[e] * a + m[e] * rt + k3[e] * xh[k].v[e]);
      }
      if (ddy) {
        if (act_y) {
          const Vec<V> yv = ldv<V>(act_y + i);
#pragma unroll
          for (int e = 0; e < V; ++e) o1.v[e] *= yv.v[e] > 0.f ? 1.f : neg;
        }
        stv<V>(ddy + i, o1);
      }
      if (ez) stv<V>(ez + i, o2);
    }
  }
}

// Sums of up to three [rows][C] partial arrays over their rows, one launch: block = 32 columns x 8 row groups of one array, fp64
// accumulation in row order (deterministic), accumulate[k] adds into dst[k] (.grad accumulation).
struct RowSum3 { const float* src[3]; float* dst[3]; int acc[3]; };
__global__ void __launch_bounds__(256) rowsum3_kernel(RowSum3 a, int rows, int C) {
  __shared__ double sm[256];
  const int k = blockIdx.y;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;    // 32 columns x 8 row groups
  const float* src = a.src[k];
  double s = 0.0;
  if (c < C) {
    int r = rg;
    for (; r + 24 < rows; r += 32) {     // four independent loads in flight per thread; the adds stay in row order
      const float v0 = src[(size_t)r * C + c], v1 = src[(size_t)(r + 8) * C + c], v2 = src[(size_t)(r + 16) * C + c], v3 = src[(size_t)(r + 24) * C + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; r < rows; r += 8) s += (double)src[(size_t)r * C + c];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    double t = sm[threadIdx.x];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += sm[threadIdx.x + 32 * j];
    float* dst = a.dst[k];
    dst[c] = a.acc[k] ? dst[c] + (float)t : (float)t;
  }
}

int pick_tc(int C) {                      // channels per block: 64, or the next power of two >= C below that (>= 4 when vectorised)
  int tc = C % 4 == 0 ? 4 : 1;
  while (tc < C && tc < 64) tc <<= 1;
  return tc;
}
bool vec4(int C, std::initializer_list<const void*> ptrs) {
  if (C % 4) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}
unsigned ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// One dispatcher for the three InstanceNorm passes (forward, backward, backward of the backward).  A pass has a register-resident
// form `Reg<NP>` (a block's slice of one sample lives in registers: NP positions per lane, every tensor read from HBM once), a
// streaming form `Stream<4>` for maps too large for that and `Stream<1>` for channel counts that are not a multiple of 4.
// Channels per block of the register-resident form are chosen so that the position-lanes match the map: 4 lanes x 256 channels for a
// 2x2 map, 16 x 64 for 6x6, 32 x 32 for 13x13 (`wide13`: the backward-of-backward keeps 16 x 64 there — with three tensors in
// registers the 32-lane form measured 80 against 61 us).  `launch(kernel, grid, tc)` issues the chosen instantiation.
template <template <int> class Reg, template <int> class Stream, class Launch>
void instnorm_dispatch(bool vec, int B, int HW, int C, bool wide13, Launch launch) {
  if (!vec) {
    int tc = 1;
    while (tc < C && tc < 64) tc <<= 1;
    launch(Stream<1>::fn(), dim3(B, (C + tc - 1) / tc), tc);
    return;
  }
  const int tc = pick_tc(C);
  int tcr = tc;
  if (HW <= 4 && C % 256 == 0) tcr = 256;
  else if (HW <= 8 && C % 128 == 0) tcr = 128;
  else if (wide13 && HW > 48 && C % 32 == 0 && tc >= 32) tcr = 32;
  const int th = IN_THREADS / (tcr / 4), np = (HW + th - 1) / th;     // positions per lane
  const dim3 grid(B, (C + tcr - 1) / tcr);
  if (np <= 1) launch(Reg<1>::fn(), grid, tcr);
  else if (np <= 2) launch(Reg<2>::fn(), grid, tcr);
  else if (np <= 3) launch(Reg<3>::fn(), grid, tcr);
  else if (np <= 6) launch(Reg<6>::fn(), grid, tcr);
  else if (np <= 12) launch(Reg<12>::fn(), grid, tcr);
  else launch(Stream<4>::fn(), dim3(B, (C + tc - 1) / tc), tc);
}
template <int NP> struct FwdReg { static auto fn() { return instnorm_fwd_reg_kernel<NP>; } };
template <int V> struct FwdStream { static auto fn() { return instnorm_fwd_kernel<V>; } };
template <int NP> struct BwdReg { static auto fn() { return instnorm_bwd_reg_kernel<NP>; } };
template <int V> struct BwdStream { static auto fn() { return instnorm_bwd_kernel<V>; } };
template <int NP> struct BwdBwdReg { static auto fn() { return instnorm_bwd_bwd_reg_kernel<NP>; } };
template <int V> struct BwdBwdStream { static auto fn() { return instnorm_bwd_bwd_kernel<V>; } };

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_instnorm_fwd(const float* x, int32_t B, int32_t HW, int32_t C, const float* gamma, const float* beta, float eps, int act,
                                float slope, float* y, float* mean, float* invstd, pcg_stream_t stream) {
  PCG_REQUIRE(x && gamma && beta && y && mean && invstd && B > 0 && HW > 0 && C > 0, "pcg_instnorm_fwd: bad arguments");
  instnorm_dispatch<FwdReg, FwdStream>(vec4(C, {x, gamma, beta, y, mean, invstd}), B, HW, C, true, [&](auto kernel, dim3 grid, int tc) {
    hipLaunchKernelGGL(kernel, grid, dim3(IN_THREADS), 0, (hipStream_t)stream, x, HW, C, tc, gamma, beta, eps, act, slope, y, mean, invstd);
  });
  return launch_status("instnorm_fwd_kernel");
}

extern "C" int pcg_instnorm_bwd_fused(const float* dy, const float* act_y, float neg_slope, const float* x, int32_t B, int32_t HW, int32_t C,
                                      const float* mean, const float* invstd, const float* gamma, float* dn_out, const float* addend, float* dx,
                                      float* dgamma_partial, float* dbeta_partial, float* dxsum_partial, pcg_stream_t stream) {
  PCG_REQUIRE(dy && x && mean && invstd && gamma && (dx || dgamma_partial) && B > 0 && HW > 0 && C > 0, "pcg_instnorm_bwd: bad arguments");
  PCG_REQUIRE((!dn_out || act_y) && (!addend || dx) && (!dxsum_partial || dx), "pcg_instnorm_bwd_fused: dn_out needs act_y; addend / dxsum_partial need dx");
  instnorm_dispatch<BwdReg, BwdStream>(vec4(C, {dy, act_y, x, mean, invstd, gamma, dn_out, addend, dx, dgamma_partial, dbeta_partial, dxsum_partial}),
                                       B, HW, C, true, [&](auto kernel, dim3 grid, int tc) {
    hipLaunchKernelGGL(kernel, grid, dim3(IN_THREADS), 0, (hipStream_t)stream, dy, act_y, neg_slope, x, HW, C, tc, mean, invstd, gamma, dn_out, addend, dx,
                       dgamma_partial, dbeta_partial, dxsum_partial);
  });
  return launch_status("instnorm_bwd_kernel");
}
extern "C" int pcg_instnorm_bwd(const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean, const float* invstd,
                                const float* gamma, float* dx, float* dgamma_partial, float* dbeta_partial, pcg_stream_t stream) {
  return pcg_instnorm_bwd_fused(dy, nullptr, 0.f, x, B, HW, C, mean, invstd, gamma, nullptr, nullptr, dx, dgamma_partial, dbeta_partial, nullptr, stream);
}

extern "C" int pcg_instnorm_bwd_bwd_act(const float* r, const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean,
                                        const float* invstd, const float* gamma, const float* act_y, float neg_slope, float* ddy, float* ez,
                                        float* dgamma_partial, pcg_stream_t stream) {
  PCG_REQUIRE(r && dy && x && mean && invstd && gamma && (ddy || ez || dgamma_partial) && B > 0 && HW > 0 && C > 0,
              "pcg_instnorm_bwd_bwd: bad arguments");
  instnorm_dispatch<BwdBwdReg, BwdBwdStream>(vec4(C, {r, dy, x, mean, invstd, gamma, ddy, ez, dgamma_partial, act_y}), B, HW, C, false,
                                             [&](auto kernel, dim3 grid, int tc) {
    hipLaunchKernelGGL(kernel, grid, dim3(IN_THREADS), 0, (hipStream_t)stream, r, dy, x, HW, C, tc, mean, invstd, gamma, ddy, ez, dgamma_partial, act_y,
                       neg_slope);
  });
  return launch_status("instnorm_bwd_bwd_kernel");
}
extern "C" int pcg_instnorm_bwd_bwd(const float* r, const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean,
                                    const float* invstd, const float* gamma, float* ddy, float* ez, float* dgamma_partial,
                                    pcg_stream_t stream) {
  return pcg_instnorm_bwd_bwd_act(r, dy, x, B, HW, C, mean, invstd, gamma, nullptr, 0.f, ddy, ez, dgamma_partial, stream);
}

extern "C" int pcg_rowsum3(int32_t n, const float* src0, float* dst0, int acc0, const float* src1, float* dst1, int acc1, const float* src2,
                           float* dst2, int acc2, int32_t rows, int32_t C, pcg_stream_t stream) {
  PCG_REQUIRE(n >= 1 && n <= 3 && rows > 0 && C > 0 && src0 && dst0 && (n < 2 || (src1 && dst1)) && (n < 3 || (src2 && dst2)), "pcg_rowsum3: bad arguments");
  RowSum3 a{{src0, src1, src2}, {dst0, dst1, dst2}, {acc0, acc1, acc2}};
  hipLaunchKernelGGL(rowsum3_kernel, dim3((C + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, a, rows, C);
  return launch_status("rowsum3_kernel");
}

extern "C" int pcg_nhwc_to_nchw_flat(const float* src, float* dst, int32_t B, int32_t HW, int32_t C, int inverse, pcg_stream_t stream) {
  PCG_REQUIRE(src && dst && src != dst && B > 0 && HW > 0 && C > 0, "pcg_nhwc_to_nchw_flat: bad arguments");
  const size_t n = (size_t)B * HW * C;
  hipLaunchKernelGGL(nhwc_to_nchw_flat_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, dst, HW, C, n, inverse);
  return launch_status("nhwc_to_nchw_flat_kernel");
}

extern "C" int pcg_interpolate_stack(const float* alpha, const float* real, const float* fake, float* x3, int32_t B, int32_t per_sample,
                                     pcg_stream_t stream) {
  PCG_REQUIRE(alpha && real && fake && x3 && B > 0 && per_sample > 0, "pcg_interpolate_stack: bad arguments");
  const size_t n = (size_t)B * per_sample;
  hipLaunchKernelGGL(interpolate_stack_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, alpha, real, fake, x3, per_sample, n);
  return launch_status("interpolate_stack_kernel");
}
extern "C" int pcg_interpolate(const float* alpha, const float* real, const float* fake, float* out, int32_t B, int32_t per_sample,
                               pcg_stream_t stream) {
  PCG_REQUIRE(alpha && real && fake && out && B > 0 && per_sample > 0, "pcg_interpolate: bad arguments");
  const size_t n = (size_t)B * per_sample;
  hipLaunchKernelGGL(interpolate_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, alpha, real, fake, out, per_sample, n);
  return launch_status("interpolate_kernel");
}

extern "C" int pcg_gradient_penalty_fwd(const float* grads, int32_t B, int32_t per_sample, float lambda, float* norms, float* penalty,
                                        pcg_stream_t stream) {
  PCG_REQUIRE(grads && norms && penalty && B > 0 && per_sample > 0, "pcg_gradient_penalty_fwd: bad arguments");
  hipLaunchKernelGGL(row_norm_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, grads, per_sample, norms);
  if (int e = launch_status("row_norm_kernel")) return e;
  hipLaunchKernelGGL(gp_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, norms, B, lambda, penalty);
  return launch_status("gp_finish_kernel");
}

extern "C" int pcg_gradient_penalty_bwd(const float* grads, const float* norms, const float* grad_out_dev, int32_t B, int32_t per_sample,
                                        float lambda, float* dgrads, pcg_stream_t stream) {
  PCG_REQUIRE(grads && norms && dgrads && B > 0 && per_sample > 0, "pcg_gradient_penalty_bwd: bad arguments");
  const size_t n = (size_t)B * per_sample;
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, grads, norms, grad_out_dev, 2.f * lambda / (float)B,
                     per_sample, n, dgrads);
  return launch_status("gp_bwd_kernel");
}
