// pointwise.hip — activations without BatchNorm, the two BCE losses, fused flat Adam, and small helpers.
// All HBM-bound streaming kernels (float4 lane accesses, grid-stride).  Reference call sites are listed
// next to each prototype in include/pcgan_hip.h.
#include <cstdlib>
#include "pcg_common.h"

namespace pcg {
namespace {

unsigned ew_blocks(size_t nv) {
  size_t b = (nv + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}
bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int VEC>
__global__ void __launch_bounds__(256) act_fwd_kernel(const float* __restrict__ x, size_t n, int act, float slope,
                                                      float* __restrict__ y) {
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (VEC == 4) {
      float4 q = reinterpret_cast<const float4*>(x)[i];
      q.x = act_apply(q.x, act, slope); q.y = act_apply(q.y, act, slope);
      q.z = act_apply(q.z, act, slope); q.w = act_apply(q.w, act, slope);
      reinterpret_cast<float4*>(y)[i] = q;
    } else {
      y[i] = act_apply(x[i], act, slope);
    }
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, size_t n,
                                                      int act, float slope, float* __restrict__ dx) {
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (VEC == 4) {
      float4 g = reinterpret_cast<const float4*>(dy)[i];
      const float4 o = reinterpret_cast<const float4*>(y)[i];
      g.x *= act_grad_from_out(o.x, act, slope); g.y *= act_grad_from_out(o.y, act, slope);
      g.z *= act_grad_from_out(o.z, act, slope); g.w *= act_grad_from_out(o.w, act, slope);
      reinterpret_cast<float4*>(dx)[i] = g;
    } else {
      dx[i] = dy[i] * act_grad_from_out(y[i], act, slope);
    }
  }
}

// one block: n is a batch size (hundreds..thousands); fixed-order tree => reproducible loss
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}

__global__ void __launch_bounds__(256) bce_kernel(const float* __restrict__ p, const float* __restrict__ target, float tconst,
                                                  int64_t n, float grad_scale, const float* grad_out, float* loss,
                                                  float* __restrict__ dp) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv_n = 1.f / (float)n;
  if (grad_out) grad_scale *= grad_out[0];
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float pi = p[i], t = target ? target[i] : tconst;
    // [torch] binary_cross_entropy: log terms clamped at -100
    const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
    acc += (t - 1.f) * lq - t * lp;
    if (dp) dp[i] = grad_scale * inv_n * (pi - t) / fmaxf((1.f - pi) * pi, 1e-12f);
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0 && loss) loss[0] = s * inv_n;
}

// Two BCE(mean) losses over the two halves of p in one launch — the discriminator's real and fake outputs side by side
// (mnist_dcgan.py:152,160; errD = errD_real + errD_fake :163).  Each half is summed exactly as bce_kernel sums it (same strided
// per-thread order, same block sum), so loss[0] / loss[1] carry the bits of two separate launches; loss[2] = loss[0] + loss[1] in
// fp32.  Backward: g0, g1, g2 are the (nullable, one-element, device) cotangents of the three outputs; dp of half h =
// (g_h + g2) / n * dBCE/dp, a missing cotangent counts as 0 (all three missing: 1).
__global__ void __launch_bounds__(256) bce_pair_kernel(const float* __restrict__ p, int64_t n, float t0, float t1, const float* g0,
                                                       const float* g1, const float* g2, float* loss, float* __restrict__ dp) {
  __shared__ float red[4];
  const float inv_n = 1.f / (float)n;
  const bool any = g0 || g1 || g2;
  float l[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float t = h ? t1 : t0;
    const float* gh = h ? g1 : g0;
    const float gs = any ? (gh ? gh[0] : 0.f) + (g2 ? g2[0] : 0.f) : 1.f;
    const float* ph = p + (size_t)h * n;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const float pi = ph[i];
      const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
      acc += (t - 1.f) * lq - t * lp;
      if (dp) dp[(size_t)h * n + i] = gs * inv_n * (pi - t) / fmaxf((1.f - pi) * pi, 1e-12f);
    }
    l[h] = block_sum_256(acc, red) * inv_n;
  }
  if (threadIdx.x == 0 && loss) { loss[0] = l[0]; loss[1] = l[1]; loss[2] = l[0] + l[1]; }
}

__global__ void __launch_bounds__(256) bce_logits_kernel(const float* __restrict__ z, float t, int64_t n, float grad_scale,
                                                         const float* grad_out, float* loss, float* __restrict__ dz) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv_n = 1.f / (float)n;
  if (grad_out) grad_scale *= grad_out[0];
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float x = z[i];
    // [torch] binary_cross_entropy_with_logits: (1-t)*x + max(-x,0) + log(exp(-max) + exp(-x-max))
    const float mx = fmaxf(-x, 0.f);
    acc += (1.f - t) * x + mx + logf(expf(-mx) + expf(-x - mx));
    if (dz) dz[i] = grad_scale * inv_n * (1.f / (1.f + expf(-x)) - t);
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0 && loss) loss[0] = s * inv_n;
}

// hyper[0] = lr / (1 - beta1^step) ; hyper[1] = sqrt(1 - beta2^step)
struct AdamHyper { float step_size, bc2_sqrt; };

struct AdamConst { float lr, w1, one_minus_w1, beta2, one_minus_beta2, eps, wd; int decoupled; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamConst& k, float step_size,
                                         float bc2_sqrt) {
  if (k.wd != 0.f) {
    if (k.decoupled) p *= (1.f - k.lr * k.wd);   // AdamW
    else g = fmaf(k.wd, p, g);                   // Adam L2
  }
  // [torch] exp_avg.lerp_(grad, 1-beta1): weight < 0.5 ? a + w(b-a) : b - (b-a)(1-w); the weights are formed in
  // double on the host (as Python does) and rounded to fp32 once
  m = (k.w1 < 0.5f) ? fmaf(k.w1, g - m, m) : g - (g - m) * k.one_minus_w1;
  v = fmaf(v, k.beta2, k.one_minus_beta2 * g * g);
  const float eps = k.eps;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

// Capturable stepping without a separate "tick" launch: the step counter lives on the device; every thread reads it at entry and
// forms the bias corrections for step t itself (two pow() in double: noise next to the memory pass), and the LAST block to finish
// (atomic ticket) writes t back — every block has read the old value by then, whatever the grid size.
//   mode 0: hy is given by value;  1: t = step + 1, the last block stores it;  2: t = step (a preceding launch already ticked)
// The two pow() cost every WAVE ~1 us of straight-line fp64 code (measured: the DCGAN update went from 18 to 44 us per launch), so
// the last block also leaves the corrections of the NEXT step in the scratch block, tagged with the step and the betas they are
// for; a launch that finds its own step there just loads them.  First step, a changed beta, a restored counter: the tag misses and
// the launch computes them itself.
struct AdamCache { int ticket; float bc2_sqrt; double bc1, beta1, beta2; long long t_for; };   // 40 bytes of the caller's scratch, zero at first use
struct AdamTick { int64_t* step; AdamCache* cache; double lr, beta1, beta2; int mode; };

template <int VEC>
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, int tail, AdamConst k,
                                                   AdamHyper hy, AdamTick tk) {
  int64_t t = 0;
  if (tk.mode) {
    t = tk.step[0] + (tk.mode == 1 ? 1 : 0);
    const AdamCache c = *tk.cache;
    double bc1;
    if (c.t_for == t && c.beta1 == tk.beta1 && c.beta2 == tk.beta2) {       // kernel-uniform
      bc1 = c.bc1; hy.bc2_sqrt = c.bc2_sqrt;
    } else {
      bc1 = 1.0 - pow(tk.beta1, (double)t);
      hy.bc2_sqrt = (float)sqrt(1.0 - pow(tk.beta2, (double)t));
    }
    hy.step_size = (float)(tk.lr / bc1);
  }
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (VEC == 4) {
      float4 p4 = reinterpret_cast<float4*>(param)[i];
      const float4 g4 = reinterpret_cast<const float4*>(grad)[i];
      float4 m4 = reinterpret_cast<float4*>(m)[i];
      float4 v4 = reinterpret_cast<float4*>(v)[i];
      adam_one(p4.x, g4.x, m4.x, v4.x, k, hy.step_size, hy.bc2_sqrt);
      adam_one(p4.y, g4.y, m4.y, v4.y, k, hy.step_size, hy.bc2_sqrt);
      adam_one(p4.z, g4.z, m4.z, v4.z, k, hy.step_size, hy.bc2_sqrt);
      adam_one(p4.w, g4.w, m4.w, v4.w, k, hy.step_size, hy.bc2_sqrt);
      reinterpret_cast<float4*>(param)[i] = p4;
      reinterpret_cast<float4*>(m)[i] = m4;
      reinterpret_cast<float4*>(v)[i] = v4;
    } else {
      float p = param[i], mm = m[i], vv = v[i];
      adam_one(p, grad[i], mm, vv, k, hy.step_size, hy.bc2_sqrt);
      param[i] = p; m[i] = mm; v[i] = vv;
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) {       // the 1..3 elements behind the 16-byte lanes
    const size_t i = n + threadIdx.x;
    float p = param[i], mm = m[i], vv = v[i];
    adam_one(p, grad[i], mm, vv, k, hy.step_size, hy.bc2_sqrt);
    param[i] = p; m[i] = mm; v[i] = vv;
  }
  if (tk.mode == 1) {          // kernel-uniform
    __syncthreads();           // every thread of this block has read the counter
    if (threadIdx.x == 0) {             // (no fence: nothing but the counter itself travels between blocks, and it is only written here)
      if (atomicAdd(&tk.cache->ticket, 1) == (int)gridDim.x - 1) {      // every block has read the counter and the cache by now
        tk.step[0] = t;
        tk.cache->ticket = 0;
        tk.cache->bc1 = 1.0 - pow(tk.beta1, (double)(t + 1));
        tk.cache->bc2_sqrt = (float)sqrt(1.0 - pow(tk.beta2, (double)(t + 1)));
        tk.cache->beta1 = tk.beta1; tk.cache->beta2 = tk.beta2; tk.cache->t_for = t + 1;
      }
    }
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ p, size_t n, float value) {
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (VEC == 4) reinterpret_cast<float4*>(p)[i] = make_float4(value, value, value, value);
    else p[i] = value;
  }
}

// single block, fixed order: diagnostic only (mnist/trainer.py:41-42)
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ p, int64_t n, float* out, int accumulate) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc = fmaf(p[i], p[i], acc);
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s;
}

// sum over segments of ||segment||_2 in ONE launch (house_sales_kc_usa/trainer.py:182-183: this trainer's grad_norm is the SUM of
// the parameters' gradient norms).  One block of 16 waves; wave w owns segments w, w+16, ...: lanes stride the segment, fp64
// butterfly over the wave, sqrt, running sum per wave; the 16 wave sums are added in wave order.  Fixed order: reproducible.
__global__ void __launch_bounds__(1024) norm_sum_kernel(const float* __restrict__ flat, const int64_t* __restrict__ seg, int nseg, float* out) {
  __shared__ double wsum[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double tot = 0.0;
  for (int sgm = wave; sgm < nseg; sgm += 16) {
    const int64_t off = seg[2 * sgm], n = seg[2 * sgm + 1];
    double acc = 0.0;
    for (int64_t i = lane; i < n; i += 64) { const double v = (double)flat[off + i]; acc = fma(v, v, acc); }
    for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh, 64);
    tot += sqrt(acc);
  }
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += wsum[w];
    out[0] = (float)t;
  }
}

int adam_launch(float* param, const float* grad, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                double wd, int decoupled, AdamHyper hy, int64_t* step_dev, AdamCache* cache_dev, hipStream_t s) {
  const bool al = al16(param) && al16(grad) && al16(m) && al16(v);
  const AdamConst k{(float)lr, (float)(1.0 - beta1), (float)(1.0 - (1.0 - beta1)), (float)beta2, (float)(1.0 - beta2), (float)eps,
                    (float)wd, decoupled};
  // 16-byte lanes over the bulk, the 1..3 trailing elements on three threads of block 0 (a flat buffer that ends in a bias of one
  // element — the WGAN-GP critic's Linear(1024, 1) — used to send all 25 M parameters down the scalar kernel)
  const int64_t bulk = al ? (n & ~(int64_t)3) : 0;
  AdamTick tk{step_dev, cache_dev, lr, beta1, beta2, step_dev ? 1 : 0};
  if (bulk) {
    unsigned blocks = ew_blocks((size_t)bulk / 4);
    if (tk.mode) {
      // every block takes a ticket at the SAME address, and device-scope atomics on one address complete ~9 ns apart (measured: 3494
      // blocks made an 16 us update 48 us): a grid-stride loop over at most 256 blocks instead (21 us)
      static const unsigned cap = [] { const char* e = getenv("PCG_ADAM_TICK_BLOCKS"); return e ? (unsigned)atoi(e) : 256u; }();
      if (blocks > cap) blocks = cap;
    }
    hipLaunchKernelGGL(adam_kernel<4>, dim3(blocks), dim3(256), 0, s, param, grad, m, v, (size_t)bulk, (int)(n - bulk), k, hy, tk);
  } else {
    unsigned blocks = ew_blocks((size_t)n);
    if (tk.mode && blocks > 256u) blocks = 256u;
    hipLaunchKernelGGL(adam_kernel<1>, dim3(blocks), dim3(256), 0, s, param, grad, m, v, (size_t)n, 0, k, hy, tk);
  }
  return launch_status("adam_kernel");
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_act_fwd(const float* x, int64_t n, int act, float slope, float* y, pcg_stream_t stream) {
  PCG_REQUIRE(x && y && n > 0, "pcg_act_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && al16(x) && al16(y))
    hipLaunchKernelGGL(act_fwd_kernel<4>, dim3(ew_blocks((size_t)n / 4)), dim3(256), 0, s, x, (size_t)n, act, slope, y);
  else
    hipLaunchKernelGGL(act_fwd_kernel<1>, dim3(ew_blocks((size_t)n)), dim3(256), 0, s, x, (size_t)n, act, slope, y);
  return launch_status("act_fwd_kernel");
}

extern "C" int pcg_act_bwd(const float* dy, const float* y, int64_t n, int act, float slope, float* dx, pcg_stream_t stream) {
  PCG_REQUIRE(dy && y && dx && n > 0, "pcg_act_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && al16(dy) && al16(y) && al16(dx))
    hipLaunchKernelGGL(act_bwd_kernel<4>, dim3(ew_blocks((size_t)n / 4)), dim3(256), 0, s, dy, y, (size_t)n, act, slope, dx);
  else
    hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(ew_blocks((size_t)n)), dim3(256), 0, s, dy, y, (size_t)n, act, slope, dx);
  return launch_status("act_bwd_kernel");
}

extern "C" int pcg_bce_fwd_bwd(const float* p, const float* target, float target_const, int64_t n, float grad_scale,
                               const float* grad_out_dev, float* loss, float* dp, pcg_stream_t stream) {
  PCG_REQUIRE(p && n > 0 && (loss || dp), "pcg_bce_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, target, target_const, n, grad_scale, grad_out_dev,
                     loss, dp);
  return launch_status("bce_kernel");
}

extern "C" int pcg_bce_pair(const float* p, int64_t n_half, float target0, float target1, const float* g0_dev, const float* g1_dev,
                            const float* g2_dev, float* loss3, float* dp, pcg_stream_t stream) {
  PCG_REQUIRE(p && n_half > 0 && (loss3 || dp), "pcg_bce_pair: bad arguments");
  hipLaunchKernelGGL(bce_pair_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, n_half, target0, target1, g0_dev, g1_dev, g2_dev, loss3, dp);
  return launch_status("bce_pair_kernel");
}

extern "C" int pcg_bce_logits_fwd_bwd(const float* z, float target_const, int64_t n, float grad_scale,
                                      const float* grad_out_dev, float* loss, float* dz, pcg_stream_t stream) {
  PCG_REQUIRE(z && n > 0 && (loss || dz), "pcg_bce_logits_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, z, target_const, n, grad_scale, grad_out_dev,
                     loss, dz);
  return launch_status("bce_logits_kernel");
}

extern "C" int pcg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int decoupled_wd, int64_t step,
                             pcg_stream_t stream) {
  PCG_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "pcg_adam_step: bad arguments");
  AdamHyper hy;
  hy.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  hy.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, hy, nullptr,
                     nullptr, (hipStream_t)stream);
}

extern "C" int pcg_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                                        double beta1, double beta2, double eps, double weight_decay, int decoupled_wd,
                                        int64_t* step_counter_dev, float* hyper_scratch2_dev, pcg_stream_t stream) {
  PCG_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step_counter_dev && hyper_scratch2_dev,
              "pcg_adam_step_capturable: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // hyper_scratch2_dev: 48 bytes, zero at first use: the last-block ticket and the cached corrections of the next step
  PCG_REQUIRE((reinterpret_cast<uintptr_t>(hyper_scratch2_dev) & 7) == 0, "pcg_adam_step_capturable: the scratch block must be 8-byte aligned");
  AdamHyper hy{0.f, 1.f};
  return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, hy, step_counter_dev,
                     reinterpret_cast<AdamCache*>(hyper_scratch2_dev), s);
}

extern "C" int pcg_fill(float* p, int64_t n, float value, pcg_stream_t stream) {
  PCG_REQUIRE(p && n > 0, "pcg_fill: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t bulk = al16(p) ? (n & ~(int64_t)3) : 0;    // 16-byte stores over the bulk, scalar stores for the 1..3 trailing elements
  if (bulk) hipLaunchKernelGGL(fill_kernel<4>, dim3(ew_blocks((size_t)bulk / 4)), dim3(256), 0, s, p, (size_t)bulk, value);
  if (n > bulk) hipLaunchKernelGGL(fill_kernel<1>, dim3(ew_blocks((size_t)(n - bulk))), dim3(256), 0, s, p + bulk, (size_t)(n - bulk), value);
  return launch_status("fill_kernel");
}

namespace pcg { namespace {
__global__ void __launch_bounds__(256) add_bias_rows_kernel(float* __restrict__ x, size_t n, int C, const float* __restrict__ bias) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] += bias[i % C];
}
} }

extern "C" int pcg_add_bias_rows(float* x, int64_t rows, int32_t C, const float* bias, pcg_stream_t stream) {
  PCG_REQUIRE(x && bias && rows > 0 && C > 0, "pcg_add_bias_rows: bad arguments");
  const size_t n = (size_t)rows * C;
  hipLaunchKernelGGL(add_bias_rows_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n, C, bias);
  return launch_status("add_bias_rows_kernel");
}

extern "C" int pcg_sumsq(const float* p, int64_t n, float* out, int accumulate, pcg_stream_t stream) {
  PCG_REQUIRE(p && out && n > 0, "pcg_sumsq: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, n, out, accumulate);
  return launch_status("sumsq_kernel");
}

extern "C" int pcg_norm_sum(const float* flat, const int64_t* seg_dev, int32_t nseg, float* out, pcg_stream_t stream) {
  PCG_REQUIRE(flat && seg_dev && out && nseg > 0, "pcg_norm_sum: bad arguments");
  hipLaunchKernelGGL(norm_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flat, seg_dev, nseg, out);
  return launch_status("norm_sum_kernel");
}
