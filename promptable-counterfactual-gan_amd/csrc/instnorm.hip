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
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int IN_THREADS = 256;

struct Lanes { int tc, th, ch, lane, c; bool on; };

__device__ __forceinline__ Lanes lanes_of(int C, int tc) {
  Lanes l;
  l.tc = tc; l.th = IN_THREADS / tc;
  l.ch = threadIdx.x % tc; l.lane = threadIdx.x / tc;
  l.c = blockIdx.y * tc + l.ch;
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
      for (int j = 0; j < l.th; ++j) s += smem[k * IN_THREADS + j * l.tc + l.ch];
      smem[k * IN_THREADS + l.ch] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = smem[k * IN_THREADS + l.ch];
  __syncthreads();
}

__global__ void __launch_bounds__(IN_THREADS) instnorm_fwd_kernel(const float* __restrict__ x, int HW, int C, int tc,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, int act, float slope, float* __restrict__ y,
                                                                 float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  __shared__ float smem[IN_THREADS];
  const Lanes l = lanes_of(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  float s[1] = {0.f};
  if (l.on) for (int p = l.lane; p < HW; p += l.th) s[0] += x[base + (size_t)p * C + l.c];
  lane_sum<1>(s, l, smem);
  const float mean = s[0] / (float)HW;
  float q[1] = {0.f};
  if (l.on) for (int p = l.lane; p < HW; p += l.th) { const float d = x[base + (size_t)p * C + l.c] - mean; q[0] = fmaf(d, d, q[0]); }
  lane_sum<1>(q, l, smem);
  const float invstd = rsqrtf(q[0] / (float)HW + eps);
  if (!l.on) return;
  if (l.lane == 0) { mean_out[(size_t)blockIdx.x * C + l.c] = mean; invstd_out[(size_t)blockIdx.x * C + l.c] = invstd; }
  const float g = gamma[l.c] * invstd, b = beta[l.c];
  for (int p = l.lane; p < HW; p += l.th) {
    const size_t i = base + (size_t)p * C + l.c;
    y[i] = act_apply(fmaf(x[i] - mean, g, b), act, slope);
  }
}

__global__ void __launch_bounds__(IN_THREADS) instnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, int HW, int C,
                                                                 int tc, const float* __restrict__ mean_in,
                                                                 const float* __restrict__ invstd_in, const float* __restrict__ gamma,
                                                                 float* __restrict__ dx, float* __restrict__ dgamma_part,
                                                                 float* __restrict__ dbeta_part) {
  __shared__ float smem[2 * IN_THREADS];
  const Lanes l = lanes_of(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  const float mean = l.on ? mean_in[(size_t)blockIdx.x * C + l.c] : 0.f, invstd = l.on ? invstd_in[(size_t)blockIdx.x * C + l.c] : 0.f;
  float s[2] = {0.f, 0.f};
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      const float d = dy[i];
      s[0] += d;
      s[1] = fmaf(d, (x[i] - mean) * invstd, s[1]);
    }
  lane_sum<2>(s, l, smem);
  if (!l.on) return;
  if (l.lane == 0) {
    if (dgamma_part) dgamma_part[(size_t)blockIdx.x * C + l.c] = s[1];
    if (dbeta_part) dbeta_part[(size_t)blockIdx.x * C + l.c] = s[0];
  }
  if (!dx) return;
  const float inv_n = 1.f / (float)HW;
  const float m1 = s[0] * inv_n, m2 = s[1] * inv_n, g = gamma[l.c] * invstd;
  for (int p = l.lane; p < HW; p += l.th) {
    const size_t i = base + (size_t)p * C + l.c;
    const float xh = (x[i] - mean) * invstd;
    dx[i] = g * (dy[i] - m1 - xh * m2);
  }
}

__global__ void __launch_bounds__(IN_THREADS) instnorm_bwd_bwd_kernel(const float* __restrict__ r, const float* __restrict__ dy,
                                                                     const float* __restrict__ x, int HW, int C, int tc,
                                                                     const float* __restrict__ mean_in, const float* __restrict__ invstd_in,
                                                                     const float* __restrict__ gamma, float* __restrict__ ddy,
                                                                     float* __restrict__ ez, float* __restrict__ dgamma_part) {
  __shared__ float smem[4 * IN_THREADS];
  const Lanes l = lanes_of(C, tc);
  const size_t base = (size_t)blockIdx.x * HW * C;
  const float mean = l.on ? mean_in[(size_t)blockIdx.x * C + l.c] : 0.f, invstd = l.on ? invstd_in[(size_t)blockIdx.x * C + l.c] : 0.f;
  float s[4] = {0.f, 0.f, 0.f, 0.f};   // sum r, sum r*xh, sum dy, sum dy*xh
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      const float rv = r[i], dv = dy[i], xh = (x[i] - mean) * invstd;
      s[0] += rv; s[1] = fmaf(rv, xh, s[1]); s[2] += dv; s[3] = fmaf(dv, xh, s[3]);
    }
  lane_sum<4>(s, l, smem);
  const float inv_n = 1.f / (float)HW;
  const float mr = s[0] * inv_n, q = s[1] * inv_n, md = s[2] * inv_n, m = s[3] * inv_n;
  float c[1] = {0.f};                    // sum (r - mean r)*(dy - mean dy): centred second pass, no cancellation
  if (l.on)
    for (int p = l.lane; p < HW; p += l.th) {
      const size_t i = base + (size_t)p * C + l.c;
      c[0] = fmaf(r[i] - mr, dy[i] - md, c[0]);
    }
  lane_sum<1>(c, l, smem);
  if (!l.on) return;
  const float pp = c[0] * inv_n;                                               // mean(r * a)
  const float gam = gamma[l.c];
  if (l.lane == 0 && dgamma_part) dgamma_part[(size_t)blockIdx.x * C + l.c] = invstd * (float)HW * (pp - m * q);
  const float g1 = gam * invstd, g2 = -gam * invstd * invstd, k3 = pp - 3.f * m * q;
  for (int p = l.lane; p < HW; p += l.th) {
    const size_t i = base + (size_t)p * C + l.c;
    const float xh = (x[i] - mean) * invstd, a = dy[i] - md, rt = r[i] - mr;
    if (ddy) ddy[i] = g1 * (rt - xh * q);
    if (ez) ez[i] = g2 * (q * a + m * rt + k3 * xh);
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

int pick_tc(int C) {
  int tc = 1;
  while (tc < C && tc < 64) tc <<= 1;
  return tc;
}
unsigned ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_instnorm_fwd(const float* x, int32_t B, int32_t HW, int32_t C, const float* gamma, const float* beta, float eps, int act,
                                float slope, float* y, float* mean, float* invstd, pcg_stream_t stream) {
  PCG_REQUIRE(x && gamma && beta && y && mean && invstd && B > 0 && HW > 0 && C > 0, "pcg_instnorm_fwd: bad arguments");
  const int tc = pick_tc(C);
  hipLaunchKernelGGL(instnorm_fwd_kernel, dim3(B, (C + tc - 1) / tc), dim3(IN_THREADS), 0, (hipStream_t)stream, x, HW, C, tc, gamma, beta, eps,
                     act, slope, y, mean, invstd);
  return launch_status("instnorm_fwd_kernel");
}

extern "C" int pcg_instnorm_bwd(const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean, const float* invstd,
                                const float* gamma, float* dx, float* dgamma_partial, float* dbeta_partial, pcg_stream_t stream) {
  PCG_REQUIRE(dy && x && mean && invstd && gamma && (dx || dgamma_partial) && B > 0 && HW > 0 && C > 0, "pcg_instnorm_bwd: bad arguments");
  const int tc = pick_tc(C);
  hipLaunchKernelGGL(instnorm_bwd_kernel, dim3(B, (C + tc - 1) / tc), dim3(IN_THREADS), 0, (hipStream_t)stream, dy, x, HW, C, tc, mean, invstd,
                     gamma, dx, dgamma_partial, dbeta_partial);
  return launch_status("instnorm_bwd_kernel");
}

extern "C" int pcg_instnorm_bwd_bwd(const float* r, const float* dy, const float* x, int32_t B, int32_t HW, int32_t C, const float* mean,
                                    const float* invstd, const float* gamma, float* ddy, float* ez, float* dgamma_partial,
                                    pcg_stream_t stream) {
  PCG_REQUIRE(r && dy && x && mean && invstd && gamma && (ddy || ez || dgamma_partial) && B > 0 && HW > 0 && C > 0,
              "pcg_instnorm_bwd_bwd: bad arguments");
  const int tc = pick_tc(C);
  hipLaunchKernelGGL(instnorm_bwd_bwd_kernel, dim3(B, (C + tc - 1) / tc), dim3(IN_THREADS), 0, (hipStream_t)stream, r, dy, x, HW, C, tc, mean,
                     invstd, gamma, ddy, ez, dgamma_partial);
  return launch_status("instnorm_bwd_bwd_kernel");
}

extern "C" int pcg_nhwc_to_nchw_flat(const float* src, float* dst, int32_t B, int32_t HW, int32_t C, int inverse, pcg_stream_t stream) {
  PCG_REQUIRE(src && dst && src != dst && B > 0 && HW > 0 && C > 0, "pcg_nhwc_to_nchw_flat: bad arguments");
  const size_t n = (size_t)B * HW * C;
  hipLaunchKernelGGL(nhwc_to_nchw_flat_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, dst, HW, C, n, inverse);
  return launch_status("nhwc_to_nchw_flat_kernel");
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
