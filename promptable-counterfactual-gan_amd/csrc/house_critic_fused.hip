// house_critic_fused.hip — the tabular spectral-norm critic (house_sales_kc_usa/models/discriminator.py:5-20:
// Linear 21->32, 32->64, 64->128 with LeakyReLU(0.2), Linear 128->1) as ONE forward and ONE backward kernel: a block owns 64 batch
// rows (lane = row) and four waves that each compute a quarter of every layer's output columns (rolled loops over the input index,
// weights staged in LDS and read as broadcasts, the rows' input vectors parked in LDS, accumulators in registers).  The weights are the spectral-normalised W / sigma that
// pcg_spectral_norm_fwd_batched produced; the weight gradients are reduced afterwards by pcg_linear_wgrad_grouped from the
// per-layer pre-activation gradients this backward writes.  Widths are the reference configuration (input 17 + 4 classes,
// hidden 32): compile-time.
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int CT = 64;                                  // rows per block (lane = row)
constexpr int NW = 4;                                   // waves per block: wave q owns a quarter of every layer's output columns
constexpr int C0 = 21, C1 = 32, C2 = 64, C3 = 128;      // layer widths (input_dim + num_classes, hidden, 2*hidden, 4*hidden)
// LDS (floats): weight image of the largest layer + bias + two parked-vector buffers (ping-pong between layers) + 4 x 64 partials
constexpr int V_FLOATS = C3 * CT;
constexpr int C_LDS = C2 * C3 + C3 + 2 * V_FLOATS + NW * CT;

struct CW { const float* w[4]; const float* b[4]; };
// Up to two passes of the critic in one launch (blockIdx.y): D(real) and D(fake) of the critic step share nothing but the module —
// each has its own spectral-norm weights (two successive power iterations), inputs and outputs.
constexpr int CP = 2;
struct CFwdPass { const float* x; const float* onehot; CW p; float* a0; float* a1; float* a2; float* a3; float* out; };
struct CFwdArgs { CFwdPass ps[CP]; };
struct CBwdPass { const float* dout; CW p; const float* a1; const float* a2; const float* a3; float* d3o; float* d2o; float* d1o; float* dx; };
struct CBwdArgs { CBwdPass ps[CP]; };

// One thread per (row, quarter): a block is 64 rows x 4 waves.  A row per THREAD alone (the first version) ran 4096 rows as 64
// waves on a chip with 1024 SIMDs, each a serial chain over all 32..128 outputs of a layer; splitting the output columns over
// four waves quadruples the wave count at the same total work, and the weights are staged by 256 threads instead of 64.
// Layer inputs travel through LDS ("parked" [K][CT]: conflict-free, lane = row), weights are broadcast reads.

// stage W ([N][K] row-major in global) transposed as Wt[i][j] (forward) / as is (backward), and the bias
template <int K, int N, bool TRANSPOSE>
__device__ __forceinline__ void stage_w(float* Wl, float* bl, const float* __restrict__ W, const float* __restrict__ b) {
  for (int e = threadIdx.x; e < K * N; e += CT * NW) {
    if (TRANSPOSE) { const int j = e / K, i = e - j * K; Wl[i * N + j] = W[e]; }
    else Wl[e] = W[e];
  }
  if (b) for (int j = threadIdx.x; j < N; j += CT * NW) bl[j] = b[j];
}
// out[NO] = b[j0..] + sum_{i<K} Wt[i][j0 + .] * V[i][row]
template <int K, int N, int NO>
__device__ __forceinline__ void lin_cols(const float* Wt, const float* bl, const float* V, int row, int j0, float (&out)[NO]) {
#pragma unroll
  for (int j = 0; j < NO; ++j) out[j] = bl[j0 + j];
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const float a = V[i * CT + row];
    const float* w = Wt + i * N + j0;
#pragma unroll
    for (int j = 0; j < NO; ++j) out[j] = fmaf(w[j], a, out[j]);
  }
}
// out[KO] = sum_{j<N} Wl[j][i0 + .] * V[j][row]            (gradient with respect to the layer's inputs i0..i0+KO)
template <int K, int N, int KO>
__device__ __forceinline__ void lin_rows_t(const float* Wl, const float* V, int row, int i0, float (&out)[KO]) {
#pragma unroll
  for (int i = 0; i < KO; ++i) out[i] = 0.f;
#pragma unroll 1
  for (int j = 0; j < N; ++j) {
    const float a = V[j * CT + row];
    const float* w = Wl + j * K + i0;
#pragma unroll
    for (int i = 0; i < KO; ++i) out[i] = fmaf(w[i], a, out[i]);
  }
}
// LeakyReLU, store the thread's NO columns of its row, park them for the next layer
template <int N, int NO>
__device__ __forceinline__ void finish_cols(float (&v)[NO], float slope, float* __restrict__ gl, size_t grow, bool on, float* Vnext, int row,
                                            int j0) {
#pragma unroll
  for (int j = 0; j < NO; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
  if (on) {
#pragma unroll
    for (int j = 0; j < NO; j += 4) *reinterpret_cast<float4*>(gl + grow * N + j0 + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
  }
#pragma unroll
  for (int j = 0; j < NO; ++j) Vnext[(j0 + j) * CT + row] = v[j];
}

__global__ void __launch_bounds__(CT * NW) critic_fwd_kernel(CFwdArgs args, int D, int NC, int B, float slope) {
  const CFwdPass& ps = args.ps[blockIdx.y];
  const float* __restrict__ x = ps.x; const float* __restrict__ onehot = ps.onehot;
  const CW& p = ps.p;
  float* __restrict__ a0 = ps.a0; float* __restrict__ a1 = ps.a1; float* __restrict__ a2 = ps.a2; float* __restrict__ a3 = ps.a3;
  float* __restrict__ out = ps.out;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wl = lds; float* bl = lds + C2 * C3; float* Va = bl + C3; float* Vb = Va + V_FLOATS; float* part = Vb + V_FLOATS;
  const int lane = threadIdx.x & (CT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * CT + lane;
  const bool on = row < (size_t)B;
  // torch.cat([x, target_onehot], 1) (:19): wave q brings in inputs q, q+4, ... of its rows, stores them to a0 and parks them
  for (int i = q; i < C0; i += NW) {
    const float v = !on ? 0.f : (i < D ? x[row * D + i] : onehot[row * NC + (i - D)]);   // D + NC == C0 (host-checked)
    if (on) a0[row * C0 + i] = v;
    Va[i * CT + lane] = v;
  }
  stage_w<C0, C1, true>(Wl, bl, p.w[0], p.b[0]);
  __syncthreads();
  {
    float h[C1 / NW];
    lin_cols<C0, C1, C1 / NW>(Wl, bl, Va, lane, q * (C1 / NW), h);
    finish_cols<C1, C1 / NW>(h, slope, a1, row, on, Vb, lane, q * (C1 / NW));
  }
  __syncthreads();
  stage_w<C1, C2, true>(Wl, bl, p.w[1], p.b[1]);
  __syncthreads();
  {
    float h[C2 / NW];
    lin_cols<C1, C2, C2 / NW>(Wl, bl, Vb, lane, q * (C2 / NW), h);
    finish_cols<C2, C2 / NW>(h, slope, a2, row, on, Va, lane, q * (C2 / NW));
  }
  __syncthreads();
  stage_w<C2, C3, true>(Wl, bl, p.w[2], p.b[2]);
  __syncthreads();
  float acc = 0.f;
  {
    constexpr int NO = C3 / NW;
    float h[NO];
    lin_cols<C2, C3, NO>(Wl, bl, Va, lane, q * NO, h);
#pragma unroll
    for (int j = 0; j < NO; ++j) h[j] = h[j] > 0.f ? h[j] : h[j] * slope;
    if (on) {
#pragma unroll
      for (int j = 0; j < NO; j += 4) *reinterpret_cast<float4*>(a3 + row * C3 + q * NO + j) = make_float4(h[j], h[j + 1], h[j + 2], h[j + 3]);
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) acc = fmaf(p.w[3][q * NO + j], h[j], acc);     // Linear(128 -> 1): this wave's quarter of the dot
  }
  part[q * CT + lane] = acc;
  __syncthreads();
  if (q == 0 && on) out[row] = p.b[3][0] + ((part[lane] + part[CT + lane]) + (part[2 * CT + lane] + part[3 * CT + lane]));
}

// pre-activation gradients d3, d2, d1 (operands of the weight gradients) and, optionally, the gradient of the first D inputs
__global__ void __launch_bounds__(CT * NW) critic_bwd_kernel(CBwdArgs args, int B, float slope, int D) {
  const CBwdPass& ps = args.ps[blockIdx.y];
  const float* __restrict__ dout = ps.dout;
  const CW& p = ps.p;
  const float* __restrict__ a1 = ps.a1; const float* __restrict__ a2 = ps.a2; const float* __restrict__ a3 = ps.a3;
  float* __restrict__ d3o = ps.d3o; float* __restrict__ d2o = ps.d2o; float* __restrict__ d1o = ps.d1o; float* __restrict__ dx = ps.dx;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wl = lds; float* Va = lds + C2 * C3 + C3; float* Vb = Va + V_FLOATS;
  const int lane = threadIdx.x & (CT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * CT + lane;
  const bool on = row < (size_t)B;
  const float g = on ? dout[row] : 0.f;
  {
    constexpr int NO = C3 / NW;        // d3 = dout * w4 * LeakyReLU'(a3): this wave's quarter
    float d[NO];
#pragma unroll
    for (int j = 0; j < NO; j += 4) {
      const float4 a = on ? *reinterpret_cast<const float4*>(a3 + row * C3 + q * NO + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      d[j] = g * p.w[3][q * NO + j] * (a.x > 0.f ? 1.f : slope);         d[j + 1] = g * p.w[3][q * NO + j + 1] * (a.y > 0.f ? 1.f : slope);
      d[j + 2] = g * p.w[3][q * NO + j + 2] * (a.z > 0.f ? 1.f : slope); d[j + 3] = g * p.w[3][q * NO + j + 3] * (a.w > 0.f ? 1.f : slope);
    }
    if (on) {
#pragma unroll
      for (int j = 0; j < NO; j += 4) *reinterpret_cast<float4*>(d3o + row * C3 + q * NO + j) = make_float4(d[j], d[j + 1], d[j + 2], d[j + 3]);
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) Va[(q * NO + j) * CT + lane] = d[j];
  }
  stage_w<C2, C3, false>(Wl, nullptr, p.w[2], nullptr);
  __syncthreads();
  {
    constexpr int KO = C2 / NW;
    float d[KO];
    lin_rows_t<C2, C3, KO>(Wl, Va, lane, q * KO, d);
#pragma unroll
    for (int j = 0; j < KO; j += 4) {
      const float4 a = on ? *reinterpret_cast<const float4*>(a2 + row * C2 + q * KO + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      d[j] *= a.x > 0.f ? 1.f : slope; d[j + 1] *= a.y > 0.f ? 1.f : slope; d[j + 2] *= a.z > 0.f ? 1.f : slope; d[j + 3] *= a.w > 0.f ? 1.f : slope;
    }
    if (on) {
#pragma unroll
      for (int j = 0; j < KO; j += 4) *reinterpret_cast<float4*>(d2o + row * C2 + q * KO + j) = make_float4(d[j], d[j + 1], d[j + 2], d[j + 3]);
    }
#pragma unroll
    for (int j = 0; j < KO; ++j) Vb[(q * KO + j) * CT + lane] = d[j];
  }
  __syncthreads();
  stage_w<C1, C2, false>(Wl, nullptr, p.w[1], nullptr);
  __syncthreads();
  {
    constexpr int KO = C1 / NW;
    float d[KO];
    lin_rows_t<C1, C2, KO>(Wl, Vb, lane, q * KO, d);
#pragma unroll
    for (int j = 0; j < KO; j += 4) {
      const float4 a = on ? *reinterpret_cast<const float4*>(a1 + row * C1 + q * KO + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      d[j] *= a.x > 0.f ? 1.f : slope; d[j + 1] *= a.y > 0.f ? 1.f : slope; d[j + 2] *= a.z > 0.f ? 1.f : slope; d[j + 3] *= a.w > 0.f ? 1.f : slope;
    }
    if (on) {
#pragma unroll
      for (int j = 0; j < KO; j += 4) *reinterpret_cast<float4*>(d1o + row * C1 + q * KO + j) = make_float4(d[j], d[j + 1], d[j + 2], d[j + 3]);
    }
#pragma unroll
    for (int j = 0; j < KO; ++j) Va[(q * KO + j) * CT + lane] = d[j];
  }
  if (dx) {   // block-uniform
    __syncthreads();
    stage_w<C0, C1, false>(Wl, nullptr, p.w[0], nullptr);
    __syncthreads();
    constexpr int KO = (C0 + NW - 1) / NW;            // 6, 6, 6, 3 of the 21 input columns
    float d[KO];
    // the last wave's window is clipped: it reads weights of columns < C0 only through the guarded store below, LDS reads stay in
    // the staged image (K*N floats) because i0 + KO <= C0 + 3 < 2*C0 and j*K + i0 + i < N*K for j < N-1; the last row is guarded
    const int i0 = q * KO;
#pragma unroll
    for (int i = 0; i < KO; ++i) d[i] = 0.f;
#pragma unroll 1
    for (int j = 0; j < C1; ++j) {
      const float a = Va[j * CT + lane];
      const float* w = Wl + j * C0 + i0;
#pragma unroll
      for (int i = 0; i < KO; ++i) d[i] = fmaf(i0 + i < C0 ? w[i] : 0.f, a, d[i]);
    }
    if (on) {
#pragma unroll
      for (int i = 0; i < KO; ++i)
        if (i0 + i < D) dx[row * D + i0 + i] = d[i];
    }
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

extern "C" int pcg_house_critic_fwd_n(int32_t n_pass, const float* const* x, const float* const* onehot, int32_t B, int32_t D, int32_t NC,
                                      const float* const* w_bar /*[n_pass*4]*/, const float* const* bias /*[4]*/, float slope, float* const* a0,
                                      float* const* a1, float* const* a2, float* const* a3, float* const* out, pcg_stream_t stream) {
  PCG_REQUIRE(n_pass >= 1 && n_pass <= CP && x && onehot && w_bar && bias && a0 && a1 && a2 && a3 && out && B > 0,
              "pcg_house_critic_fwd_n: bad arguments (1 or 2 passes)");
  PCG_REQUIRE(D + NC == C0 && D > 0 && NC > 0, "pcg_house_critic_fwd_n: built for input_dim + num_classes = %d and hidden width %d", C0, C1);
  CFwdArgs args{};
  for (int q = 0; q < n_pass; ++q) {
    CFwdPass& ps = args.ps[q];
    PCG_REQUIRE(x[q] && onehot[q] && a0[q] && a1[q] && a2[q] && a3[q] && out[q], "pcg_house_critic_fwd_n: pass %d: null buffer", q);
    ps.x = x[q]; ps.onehot = onehot[q]; ps.a0 = a0[q]; ps.a1 = a1[q]; ps.a2 = a2[q]; ps.a3 = a3[q]; ps.out = out[q];
    for (int l = 0; l < 4; ++l) {
      PCG_REQUIRE(w_bar[q * 4 + l] && bias[l], "pcg_house_critic_fwd_n: pass %d: null layer %d", q, l);
      ps.p.w[l] = w_bar[q * 4 + l]; ps.p.b[l] = bias[l];
    }
  }
  static int once = set_lds(reinterpret_cast<const void*>(critic_fwd_kernel));
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL(critic_fwd_kernel, dim3((B + CT - 1) / CT, n_pass), dim3(CT * NW), C_LDS * sizeof(float), (hipStream_t)stream, args, D, NC, B, slope);
  return launch_status("critic_fwd_kernel");
}

extern "C" int pcg_house_critic_fwd(const float* x, const float* onehot, int32_t B, int32_t D, int32_t NC, const float* const* w_bar,
                                    const float* const* bias, float slope, float* a0, float* a1, float* a2, float* a3, float* out,
                                    pcg_stream_t stream) {
  PCG_REQUIRE(w_bar && bias, "pcg_house_critic_fwd: bad arguments");
  return pcg_house_critic_fwd_n(1, &x, &onehot, B, D, NC, w_bar, bias, slope, &a0, &a1, &a2, &a3, &out, stream);
}

extern "C" int pcg_house_critic_bwd_n(int32_t n_pass, const float* const* dout, int32_t B, int32_t D, const float* const* w_bar /*[n_pass*4]*/,
                                      float slope, const float* const* a1, const float* const* a2, const float* const* a3, float* const* d3,
                                      float* const* d2, float* const* d1, float* const* dx /*entries nullable*/, pcg_stream_t stream) {
  PCG_REQUIRE(n_pass >= 1 && n_pass <= CP && dout && w_bar && a1 && a2 && a3 && d3 && d2 && d1 && dx && B > 0 && D > 0 && D <= C0,
              "pcg_house_critic_bwd_n: bad arguments (1 or 2 passes)");
  CBwdArgs args{};
  for (int q = 0; q < n_pass; ++q) {
    CBwdPass& ps = args.ps[q];
    PCG_REQUIRE(dout[q] && a1[q] && a2[q] && a3[q] && d3[q] && d2[q] && d1[q], "pcg_house_critic_bwd_n: pass %d: null buffer", q);
    ps.dout = dout[q]; ps.a1 = a1[q]; ps.a2 = a2[q]; ps.a3 = a3[q]; ps.d3o = d3[q]; ps.d2o = d2[q]; ps.d1o = d1[q]; ps.dx = dx[q];
    for (int l = 0; l < 4; ++l) { PCG_REQUIRE(w_bar[q * 4 + l], "pcg_house_critic_bwd_n: pass %d: null layer %d", q, l); ps.p.w[l] = w_bar[q * 4 + l]; }
  }
  static int once = set_lds(reinterpret_cast<const void*>(critic_bwd_kernel));
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL(critic_bwd_kernel, dim3((B + CT - 1) / CT, n_pass), dim3(CT * NW), C_LDS * sizeof(float), (hipStream_t)stream, args, B, slope, D);
  return launch_status("critic_bwd_kernel");
}

extern "C" int pcg_house_critic_bwd(const float* dout, int32_t B, int32_t D, const float* const* w_bar, float slope, const float* a1,
                                    const float* a2, const float* a3, float* d3, float* d2, float* d1, float* dx, pcg_stream_t stream) {
  PCG_REQUIRE(w_bar, "pcg_house_critic_bwd: bad arguments");
  return pcg_house_critic_bwd_n(1, &dout, B, D, w_bar, slope, &a1, &a2, &a3, &d3, &d2, &d1, &dx, stream);
}
