// house_critic_fused.hip — the tabular spectral-norm critic (house_sales_kc_usa/models/discriminator.py:5-20:
// Linear 21->32, 32->64, 64->128 with LeakyReLU(0.2), Linear 128->1) as ONE forward and ONE backward kernel, one thread per
// batch row (same scheme as house_fused.hip: rolled loops over the input index, transposed weights staged in LDS, the lane's
// input vector parked in LDS, accumulators in registers).  The weights are the spectral-normalised W / sigma that
// pcg_spectral_norm_fwd_batched produced; the weight gradients are reduced afterwards by pcg_linear_wgrad_grouped from the
// per-layer pre-activation gradients this backward writes.  Widths are the reference configuration (input 17 + 4 classes,
// hidden 32): compile-time.
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int CT = 64;                                  // rows per block = one wave
constexpr int C0 = 21, C1 = 32, C2 = 64, C3 = 128;      // layer widths (input_dim + num_classes, hidden, 2*hidden, 4*hidden)
constexpr int C_LDS = C2 * C3 + C3 + C3 * CT;           // floats: largest [K][N] weight image + bias + parked vectors

// out[j<N] = b[j] + sum_{i<K} W[j][i] in[i]      (W row-major [N][K] in global memory)
template <int K, int N>
__device__ __forceinline__ void lin_kn(float* lds, const float* __restrict__ W, const float* __restrict__ b, const float (&in)[K],
                                       float (&out)[N]) {
  float* Wt = lds; float* bl = lds + K * N; float* V = bl + N;
  __syncthreads();
  for (int e = threadIdx.x; e < K * N; e += CT) { const int j = e / K, i = e - j * K; Wt[i * N + j] = W[e]; }
  for (int j = threadIdx.x; j < N; j += CT) bl[j] = b[j];
#pragma unroll
  for (int i = 0; i < K; ++i) V[i * CT + threadIdx.x] = in[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) out[j] = bl[j];
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const float a = V[i * CT + threadIdx.x];
    const float* w = Wt + i * N;
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = fmaf(w[j], a, out[j]);
  }
}
// out[i<K] = sum_{j<N} W[j][i] v[j]              (gradient with respect to the input of the layer)
template <int K, int N>
__device__ __forceinline__ void lin_t_kn(float* lds, const float* __restrict__ W, const float (&v)[N], float (&out)[K]) {
  float* Wl = lds; float* V = lds + K * N;
  __syncthreads();
  for (int e = threadIdx.x; e < K * N; e += CT) Wl[e] = W[e];
#pragma unroll
  for (int j = 0; j < N; ++j) V[j * CT + threadIdx.x] = v[j];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < K; ++i) out[i] = 0.f;
#pragma unroll 1
  for (int j = 0; j < N; ++j) {
    const float a = V[j * CT + threadIdx.x];
    const float* w = Wl + j * K;
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] = fmaf(w[i], a, out[i]);
  }
}

template <int N>
__device__ __forceinline__ void lrelu(float (&v)[N], float slope) {
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
}
template <int N>
__device__ __forceinline__ void store_row(float* p, size_t row, bool on, const float (&v)[N]) {
  if (!on) return;
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int j = 0; j < N; j += 4) *reinterpret_cast<float4*>(p + row * N + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) p[row * N + j] = v[j];
  }
}
template <int N>
__device__ __forceinline__ void load_row(const float* p, size_t row, bool on, float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    const float4 q = on ? *reinterpret_cast<const float4*>(p + row * N + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
  }
}

struct CW { const float* w[4]; const float* b[4]; };

__global__ void __launch_bounds__(CT) critic_fwd_kernel(const float* __restrict__ x, int D, const float* __restrict__ onehot, int NC, int B,
                                                        CW p, float slope, float* __restrict__ a0, float* __restrict__ a1,
                                                        float* __restrict__ a2, float* __restrict__ a3, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row = blockIdx.x * CT + threadIdx.x;
  const bool on = row < B;
  float in0[C0];
#pragma unroll
  for (int i = 0; i < C0; ++i) in0[i] = !on ? 0.f : (i < D ? x[(size_t)row * D + i] : onehot[(size_t)row * NC + (i - D)]);   // D + NC == C0 (host-checked)
  store_row<C0>(a0, row, on, in0);                                       // torch.cat([x, target_onehot], 1)  (:19)
  float h1[C1];
  lin_kn<C0, C1>(lds, p.w[0], p.b[0], in0, h1);
  lrelu<C1>(h1, slope);
  store_row<C1>(a1, row, on, h1);
  float h2[C2];
  lin_kn<C1, C2>(lds, p.w[1], p.b[1], h1, h2);
  lrelu<C2>(h2, slope);
  store_row<C2>(a2, row, on, h2);
  float h3[C3];
  lin_kn<C2, C3>(lds, p.w[2], p.b[2], h2, h3);
  lrelu<C3>(h3, slope);
  store_row<C3>(a3, row, on, h3);
  float acc = p.b[3][0];
#pragma unroll 8
  for (int j = 0; j < C3; ++j) acc = fmaf(p.w[3][j], h3[j], acc);
  if (on) out[row] = acc;
}

// pre-activation gradients d3, d2, d1 (operands of the weight gradients) and, optionally, the gradient of the first D inputs
__global__ void __launch_bounds__(CT) critic_bwd_kernel(const float* __restrict__ dout, int B, CW p, float slope, const float* __restrict__ a1,
                                                        const float* __restrict__ a2, const float* __restrict__ a3,
                                                        float* __restrict__ d3o, float* __restrict__ d2o, float* __restrict__ d1o,
                                                        float* __restrict__ dx, int D) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row = blockIdx.x * CT + threadIdx.x;
  const bool on = row < B;
  const float g = on ? dout[row] : 0.f;
  float d3[C3], act[C3];
  load_row<C3>(a3, row, on, act);
#pragma unroll
  for (int j = 0; j < C3; ++j) d3[j] = g * p.w[3][j] * (act[j] > 0.f ? 1.f : slope);
  store_row<C3>(d3o, row, on, d3);
  float d2[C2];
  lin_t_kn<C2, C3>(lds, p.w[2], d3, d2);
  {
    float a[C2];
    load_row<C2>(a2, row, on, a);
#pragma unroll
    for (int j = 0; j < C2; ++j) d2[j] *= (a[j] > 0.f ? 1.f : slope);
  }
  store_row<C2>(d2o, row, on, d2);
  float d1[C1];
  lin_t_kn<C1, C2>(lds, p.w[1], d2, d1);
  {
    float a[C1];
    load_row<C1>(a1, row, on, a);
#pragma unroll
    for (int j = 0; j < C1; ++j) d1[j] *= (a[j] > 0.f ? 1.f : slope);
  }
  store_row<C1>(d1o, row, on, d1);
  if (dx) {
    float d0[C0];
    lin_t_kn<C0, C1>(lds, p.w[0], d1, d0);
    if (on)
      for (int i = 0; i < D; ++i) dx[(size_t)row * D + i] = d0[i];
  }
}

int set_lds(const void* fn) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C_LDS * sizeof(float)));
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e)); return PCG_ERR_LAUNCH; }
  return PCG_OK;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_house_critic_fwd(const float* x, const float* onehot, int32_t B, int32_t D, int32_t NC, const float* const* w_bar,
                                    const float* const* bias, float slope, float* a0, float* a1, float* a2, float* a3, float* out,
                                    pcg_stream_t stream) {
  PCG_REQUIRE(x && onehot && w_bar && bias && a0 && a1 && a2 && a3 && out && B > 0, "pcg_house_critic_fwd: bad arguments");
  PCG_REQUIRE(D + NC == C0 && D > 0 && NC > 0, "pcg_house_critic_fwd: built for input_dim + num_classes = %d and hidden width %d", C0, C1);
  CW p{};
  for (int l = 0; l < 4; ++l) { PCG_REQUIRE(w_bar[l] && bias[l], "pcg_house_critic_fwd: null layer %d", l); p.w[l] = w_bar[l]; p.b[l] = bias[l]; }
  static int once = set_lds(reinterpret_cast<const void*>(critic_fwd_kernel));
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL(critic_fwd_kernel, dim3((B + CT - 1) / CT), dim3(CT), C_LDS * sizeof(float), (hipStream_t)stream, x, D, onehot, NC, B, p,
                     slope, a0, a1, a2, a3, out);
  return launch_status("critic_fwd_kernel");
}

extern "C" int pcg_house_critic_bwd(const float* dout, int32_t B, int32_t D, const float* const* w_bar, float slope, const float* a1,
                                    const float* a2, const float* a3, float* d3, float* d2, float* d1, float* dx, pcg_stream_t stream) {
  PCG_REQUIRE(dout && w_bar && a1 && a2 && a3 && d3 && d2 && d1 && B > 0 && D > 0 && D <= C0, "pcg_house_critic_bwd: bad arguments");
  CW p{};
  for (int l = 0; l < 4; ++l) { PCG_REQUIRE(w_bar[l], "pcg_house_critic_bwd: null layer %d", l); p.w[l] = w_bar[l]; }
  static int once = set_lds(reinterpret_cast<const void*>(critic_bwd_kernel));
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL(critic_bwd_kernel, dim3((B + CT - 1) / CT), dim3(CT), C_LDS * sizeof(float), (hipStream_t)stream, dout, B, p, slope, a1, a2,
                     a3, d3, d2, d1, dx, D);
  return launch_status("critic_bwd_kernel");
}
