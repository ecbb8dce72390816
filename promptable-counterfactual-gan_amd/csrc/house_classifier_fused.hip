// house_classifier_fused.hip — the frozen tabular classifier of the counterfactual loss (house_sales_kc_usa/models/nn_classifier.py:4-32
// in eval mode, every BatchNorm1d folded into the following Linear at pack time: Linear 17->256, 256->256, 256->128, 128->64 with
// LeakyReLU(0.1), Linear 64->4) as ONE forward launch and ONE backward launch (gradient with respect to the input rows only:
// trainer.py:301-302, main.py:27-30 — the classifier's parameters are frozen).
//
// A block owns 32 batch rows and runs the whole chain on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32), the activations
// never leaving LDS between layers:
//   * activations sit k-major in LDS, X[k][row] with a pitch of 33 floats: the A operand of a k-step is two rows of 32 consecutive
//     floats (conflict-free), and an output tile is written back one column per lane (bank = column + row: conflict-free);
//   * weights are NOT staged: lane (li, lh) needs B[k0 + lh][n0 + li], i.e. 32 consecutive floats per half-wave of a k-major weight
//     image — a coalesced 128-byte read straight from L2 (the whole net is 444 KB; every block walks it once).  The forward uses
//     k-major (transposed, zero-padded to an even K) copies made at pack time, the backward the matrices as stored ([out][in] is
//     k-major for dX = dY W).  The B operands of the next group of k-steps are requested before the MFMAs of the current one;
//   * the four waves split the 32-column output tiles of a layer (two each at width 256, one each at 128); the last backward layer
//     (256 -> 17: one tile) is split over the reduction index instead and the four partial tiles are added through LDS in wave order.
// The op chain this replaces is 5 GEMM + 4 LeakyReLU-backward + 5 GEMM launches of 64-128 blocks each (~170 us at batch 4096).
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int CL_R = 32;                 // rows per block
constexpr int CL_P = CL_R + 1;           // LDS pitch of a k-row
constexpr int CL_W = 256;                // widest layer
constexpr int CL_IN = 17, CL_INP = 18;   // input width, padded to an even reduction length
constexpr int CL_H1 = 256, CL_H2 = 256, CL_H3 = 128, CL_H4 = 64, CL_OUT = 4;
constexpr float CL_SLOPE = 0.1f;
typedef float cl_acc_t __attribute__((ext_vector_type(16)));

struct ClsFwdW { const float* wt[5]; const float* b[5]; };   // wt[l]: [K_l (padded)][N_l] k-major; layer 4 (64 -> 4): as stored [4][64]
struct ClsBwdW { const float* w[5]; };                       // as stored [N_l][K_l]

__device__ __forceinline__ int cl_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// One dense layer on the matrix cores: Y[32 rows][N] = act(X[32][K] Wt[K][N] + bias), X / Y k-major in LDS.  Waves take the column
// tiles wave * TPW .. ; G k-steps (2 G reduction indices) form a group whose B operands are prefetched one group ahead.
template <int K, int N, bool LEAKY>
__device__ __forceinline__ void cl_dense_fwd(const float* __restrict__ Xin, float* __restrict__ Xout, const float* __restrict__ Wt,
                                             const float* __restrict__ bias, float* __restrict__ gsave, size_t row0, int rows, int wave,
                                             int li, int lh) {
  constexpr int NT = N / 32, TPW = NT >= 4 ? NT / 4 : 1, STEPS = K / 2, G = STEPS % 16 == 0 ? 16 : STEPS, NG = STEPS / G;
  static_assert(N % 32 == 0 && K % 2 == 0 && STEPS % G == 0, "tile shapes");
  if (wave * TPW >= NT) return;                        // (N = 64: waves 2, 3 have no tile; no barrier inside this function)
  const int n0 = wave * TPW * 32;
  cl_acc_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const float bv = bias[n0 + t * 32 + li];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = bv;
  }
  float bcur[TPW][G], bnxt[TPW][G];
  const float* wp = Wt + (size_t)lh * N + n0 + li;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g) bcur[t][g] = wp[(size_t)(2 * g) * N + t * 32];
#pragma unroll 1
  for (int grp = 0; grp < NG; ++grp) {
    const int k0 = grp * 2 * G;
    if (grp + 1 < NG) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) bnxt[t][g] = wp[(size_t)(k0 + 2 * G + 2 * g) * N + t * 32];
    }
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] = Xin[(k0 + 2 * g + lh) * CL_P + li];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g], bcur[t][g], acc[t], 0, 0, 0);
    if (grp + 1 < NG) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) bcur[t][g] = bnxt[t][g];
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = cl_row(r, lh), n = n0 + t * 32 + li;
      float v = acc[t][r];
      if (LEAKY) v = v > 0.f ? v : v * CL_SLOPE;
      Xout[n * CL_P + m] = v;
      if (gsave && m < rows) gsave[(row0 + m) * N + n] = v;
    }
}

// dX[32 rows][KO] = (dY[32][N] W[N][KO]) * LeakyReLU'(a), dY k-major in LDS (index n), W as stored; the result replaces nothing in
// LDS: it goes to Dout (k-major, index = output column).  a: the layer's saved post-activation, row-major [B][KO] in global memory.
template <int N, int KO>
__device__ __forceinline__ void cl_dense_bwd(const float* __restrict__ Din, float* __restrict__ Dout, const float* __restrict__ W,
                                             const float* __restrict__ act, size_t row0, int rows, int wave, int li, int lh) {
  constexpr int NT = KO / 32, TPW = NT >= 4 ? NT / 4 : 1, STEPS = N / 2, G = STEPS % 16 == 0 ? 16 : STEPS, NG = STEPS / G;
  static_assert(KO % 32 == 0 && N % 2 == 0 && STEPS % G == 0, "tile shapes");
  if (wave * TPW >= NT) return;
  const int j0 = wave * TPW * 32;
  cl_acc_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // the activation signs of this wave's tiles: requested now, used after the MFMAs
  float av[TPW][16];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(cl_row(r, lh), rows - 1);
      av[t][r] = act[(row0 + m) * KO + j0 + t * 32 + li];
    }
  float bcur[TPW][G], bnxt[TPW][G];
  const float* wp = W + (size_t)lh * KO + j0 + li;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g) bcur[t][g] = wp[(size_t)(2 * g) * KO + t * 32];
#pragma unroll 1
  for (int grp = 0; grp < NG; ++grp) {
    const int n0 = grp * 2 * G;
    if (grp + 1 < NG) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) bnxt[t][g] = wp[(size_t)(n0 + 2 * G + 2 * g) * KO + t * 32];
    }
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] = Din[(n0 + 2 * g + lh) * CL_P + li];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g], bcur[t][g], acc[t], 0, 0, 0);
    if (grp + 1 < NG) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) bcur[t][g] = bnxt[t][g];
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = cl_row(r, lh), j = j0 + t * 32 + li;
      Dout[j * CL_P + m] = acc[t][r] * (av[t][r] > 0.f ? 1.f : CL_SLOPE);
    }
}

struct alignas(16) ClsSmem {
  float X[2][CL_W * CL_P];                 // activation ping-pong, k-major
  float part[4][CL_R * CL_P];              // backward tail: the four waves' partial input-gradient tiles
};

__global__ void __launch_bounds__(256) classifier_fwd_kernel(const float* __restrict__ x, int B, ClsFwdW w, float* __restrict__ a1,
                                                             float* __restrict__ a2, float* __restrict__ a3, float* __restrict__ a4,
                                                             float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cls_lds[];
  ClsSmem& s = *reinterpret_cast<ClsSmem*>(cls_lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const size_t row0 = (size_t)blockIdx.x * CL_R;
  const int rows = min(CL_R, B - (int)row0);
  // the block's input rows, k-major, with the zero row that pads the reduction to 18
  {
    constexpr int PER = (CL_R * CL_INP + 255) / 256;
    float xv[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int e = threadIdx.x + t * 256, m = min(e / CL_INP, rows - 1), k = min(e - (e / CL_INP) * CL_INP, CL_IN - 1);
      xv[t] = x[(row0 + m) * CL_IN + k];
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int e = threadIdx.x + t * 256;
      if (e < CL_R * CL_INP) { const int m = e / CL_INP, k = e - m * CL_INP; s.X[0][k * CL_P + m] = (m < rows && k < CL_IN) ? xv[t] : 0.f; }
    }
  }
  __syncthreads();
  cl_dense_fwd<CL_INP, CL_H1, true>(s.X[0], s.X[1], w.wt[0], w.b[0], a1, row0, rows, wave, li, lh);
  __syncthreads();
  cl_dense_fwd<CL_H1, CL_H2, true>(s.X[1], s.X[0], w.wt[1], w.b[1], a2, row0, rows, wave, li, lh);
  __syncthreads();
  cl_dense_fwd<CL_H2, CL_H3, true>(s.X[0], s.X[1], w.wt[2], w.b[2], a3, row0, rows, wave, li, lh);
  __syncthreads();
  cl_dense_fwd<CL_H3, CL_H4, true>(s.X[1], s.X[0], w.wt[3], w.b[3], a4, row0, rows, wave, li, lh);
  __syncthreads();
  // Linear(64 -> 4): 128 outputs, one per thread of the first two waves
  if (threadIdx.x < CL_R * CL_OUT) {
    const int m = threadIdx.x >> 2, c = threadIdx.x & 3;
    float acc = w.b[4][c];
    const float* wr = w.wt[4] + c * CL_H4;
#pragma unroll 8
    for (int k = 0; k < CL_H4; ++k) acc = fmaf(s.X[0][k * CL_P + m], wr[k], acc);
    if (m < rows) logits[(row0 + m) * CL_OUT + c] = acc;
  }
}

__global__ void __launch_bounds__(256) classifier_bwd_kernel(const float* __restrict__ dlogits, int B, ClsBwdW w, const float* __restrict__ a1,
                                                             const float* __restrict__ a2, const float* __restrict__ a3,
                                                             const float* __restrict__ a4, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cls_lds[];
  ClsSmem& s = *reinterpret_cast<ClsSmem*>(cls_lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const size_t row0 = (size_t)blockIdx.x * CL_R;
  const int rows = min(CL_R, B - (int)row0);
  // d4[m][k] = (sum_c dlogits[m][c] W5[c][k]) * LeakyReLU'(a4[m][k]): 2048 outputs, eight per thread
  {
    constexpr int PER = CL_R * CL_H4 / 256;                 // element tid + 256 t of [32][64]: column tid % 64 for every t
    const int k = threadIdx.x & (CL_H4 - 1);
    float w5[CL_OUT], dl[PER][CL_OUT], av[PER];
#pragma unroll
    for (int c = 0; c < CL_OUT; ++c) w5[c] = w.w[4][c * CL_H4 + k];
#pragma unroll
    for (int t = 0; t < PER; ++t) {                          // all loads first: a load-use loop pays one memory latency per trip
      const int m = min((int)(threadIdx.x + t * 256) / CL_H4, rows - 1);
#pragma unroll
      for (int c = 0; c < CL_OUT; ++c) dl[t][c] = dlogits[(row0 + m) * CL_OUT + c];
      av[t] = a4[(row0 + m) * CL_H4 + k];
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int m = (threadIdx.x + t * 256) / CL_H4;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < CL_OUT; ++c) acc = fmaf(dl[t][c], w5[c], acc);
      s.X[0][k * CL_P + m] = m < rows ? acc * (av[t] > 0.f ? 1.f : CL_SLOPE) : 0.f;
    }
  }
  __syncthreads();
  cl_dense_bwd<CL_H4, CL_H3>(s.X[0], s.X[1], w.w[3], a3, row0, rows, wave, li, lh);
  __syncthreads();
  cl_dense_bwd<CL_H3, CL_H2>(s.X[1], s.X[0], w.w[2], a2, row0, rows, wave, li, lh);
  __syncthreads();
  cl_dense_bwd<CL_H2, CL_H1>(s.X[0], s.X[1], w.w[1], a1, row0, rows, wave, li, lh);
  __syncthreads();
  // dx[32][17] = d1[32][256] W1[256][17]: one output tile; each wave reduces a quarter of the 256 and the partial tiles are added in
  // wave order.  Lanes li >= 17 feed zeros (clamped address, dropped value).
  {
    constexpr int G = 16, NQ = CL_H1 / 4;                 // 64 reduction indices = 32 k-steps per wave, two groups
    cl_acc_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* wp = w.w[0] + (size_t)(wave * NQ + lh) * CL_IN + min(li, CL_IN - 1);
#pragma unroll 1
    for (int grp = 0; grp < NQ / (2 * G); ++grp) {
      float a[G], b[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int n = grp * 2 * G + 2 * g;
        const float bv = wp[(size_t)n * CL_IN];
        b[g] = li < CL_IN ? bv : 0.f;
        a[g] = s.X[1][(wave * NQ + n + lh) * CL_P + li];
      }
#pragma unroll
      for (int g = 0; g < G; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g], b[g], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s.part[wave][li * CL_P + cl_row(r, lh)] = acc[r];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < CL_R * CL_IN; e += 256) {
    const int m = e / CL_IN, i = e - m * CL_IN;
    if (m < rows) dx[(row0 + m) * CL_IN + i] = ((s.part[0][i * CL_P + m] + s.part[1][i * CL_P + m]) + s.part[2][i * CL_P + m]) + s.part[3][i * CL_P + m];
  }
}

int cls_set_lds(const void* fn) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ClsSmem));
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e)); return PCG_ERR_LAUNCH; }
  return PCG_OK;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_house_classifier_fwd(const float* x, int32_t B, const float* const* w_kmajor, const float* const* bias, float* a1, float* a2,
                                        float* a3, float* a4, float* logits, pcg_stream_t stream) {
  PCG_REQUIRE(x && B > 0 && w_kmajor && bias && a1 && a2 && a3 && a4 && logits, "pcg_house_classifier_fwd: bad arguments");
  ClsFwdW w{};
  for (int l = 0; l < 5; ++l) { PCG_REQUIRE(w_kmajor[l] && bias[l], "pcg_house_classifier_fwd: null layer %d", l); w.wt[l] = w_kmajor[l]; w.b[l] = bias[l]; }
  static int once = cls_set_lds(reinterpret_cast<const void*>(classifier_fwd_kernel));
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL(classifier_fwd_kernel, dim3((B + CL_R - 1) / CL_R), dim3(256), sizeof(ClsSmem), (hipStream_t)stream, x, B, w, a1, a2, a3, a4, logits);
  return launch_status("classifier_fwd_kernel");
}

extern "C" int pcg_house_classifier_bwd(const float* dlogits, int32_t B, const float* const* w_stored, const float* a1, const float* a2,
                                        const float* a3, const float* a4, float* dx, pcg_stream_t stream) {
  PCG_REQUIRE(dlogits && B > 0 && w_stored && a1 && a2 && a3 && a4 && dx, "pcg_house_classifier_bwd: bad arguments");
  ClsBwdW w{};
  for (int l = 0; l < 5; ++l) { PCG_REQUIRE(w_stored[l], "pcg_house_classifier_bwd: null layer %d", l); w.w[l] = w_stored[l]; }
  static int once = cls_set_lds(reinterpret_cast<const void*>(classifier_bwd_kernel));
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL(classifier_bwd_kernel, dim3((B + CL_R - 1) / CL_R), dim3(256), sizeof(ClsSmem), (hipStream_t)stream, dlogits, B, w, a1, a2, a3, a4, dx);
  return launch_status("classifier_bwd_kernel");
}
