// house_classifier_fused.hip — the frozen tabular classifier of the counterfactual loss (house_sales_kc_usa/models/nn_classifier.py:4-32
// in eval mode, every BatchNorm1d folded into the following Linear at pack time: Linear 17->256, 256->256, 256->128, 128->64 with
// LeakyReLU(0.1), Linear 64->4) as ONE forward launch and ONE backward launch (gradient with respect to the input rows only:
// trainer.py:301-302, main.py:27-30 — the classifier's parameters are frozen).
//
// A block owns 16 batch rows and runs the whole chain on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32), the activations
// never leaving LDS between layers:
//   * activations sit k-major in LDS, X[k][row] with a pitch of 17 floats: the A operand of a step is four rows of 16 consecutive
//     floats, and an output tile is written back one column per lane;
//   * weights are NOT staged: lane (li, lq) needs B[k0 + lq][n0 + li], i.e. 16 consecutive floats per quarter-wave of a k-major weight
//     image — coalesced 64-byte reads straight from L2 (the whole net is 444 KB; every block walks it once).  The forward uses
//     k-major (transposed, K zero-padded to a multiple of 4) copies made at pack time, the backward the matrices as stored ([out][in] is
//     k-major for dX = dY W).  The B operands of the next group of k-steps are requested before the MFMAs of the current one;
//   * the eight waves split the 32-column output tiles of a layer (one each at width 256); layers with fewer tiles (128, 64 wide, the
//     256 -> 17 tail) split their reduction over the spare waves and the partial tiles are added through LDS in a fixed order.
// The op chain this replaces is 5 GEMM + 4 LeakyReLU-backward + 5 GEMM launches of 64-128 blocks each (~170 us at batch 4096).
#include <cstdint>
#include "pcg_common.h"
#include "spectral_norm_body.h"

namespace pcg {
namespace {

constexpr int CL_R = 16;                 // rows per block: 4096 rows are 256 blocks, one per CU (32-row tiles left half the chip idle
                                         // and each CU MFMA-bound: 128 blocks x 1024 MFMAs for the 256 -> 256 layer alone)
constexpr int CL_P = CL_R + 1;           // LDS pitch of a k-row
constexpr int CL_W = 256;                // widest layer
constexpr int CL_IN = 17, CL_INP = 20;   // input width, padded to a multiple of the MFMA's reduction depth (4)
constexpr int CL_H1 = 256, CL_H2 = 256, CL_H3 = 128, CL_H4 = 64, CL_OUT = 4;
constexpr float CL_SLOPE = 0.1f;
constexpr int CL_NW = 8, CL_NT = CL_NW * 64;     // waves / threads per block
typedef float cl_acc_t __attribute__((ext_vector_type(4)));

struct ClsFwdW { const float* wt[5]; const float* b[5]; };   // wt[l]: [K_l (padded)][N_l] k-major; layer 4 (64 -> 4): as stored [4][64]
struct ClsBwdW { const float* w[5]; };                       // as stored [N_l][K_l]

// v_mfma_f32_16x16x4_f32: lane (li = lane % 16, lq = lane / 16) supplies A[m = li][k = lq] and B[k = lq][n = li]; accumulator
// register r holds D[m = 4 lq + r][n = li].
//
// One dense layer: Y[16 rows][N] = act(X[16][K] Wt[K][N] + bias), X / Y k-major in LDS.  The eight waves take the N / 16 column
// tiles; a layer with fewer tiles than waves splits its reduction instead (KS waves per tile, a contiguous share of K each) and the
// partial tiles are added in share order through LDS.  G MFMA steps (4 G reduction indices) form a group whose B operands are
// prefetched two groups ahead.  Contains block barriers in the split form: all waves call it.
template <int K, int N, bool LEAKY>
__device__ __forceinline__ void cl_dense_fwd(const float* __restrict__ Xin, float* __restrict__ Xout, const float* __restrict__ Wt,
                                             const float* __restrict__ bias, float* __restrict__ gsave, float* part, size_t row0, int rows,
                                             int wave, int li, int lq) {
  constexpr int NT = N / 16, TPW = NT >= CL_NW ? NT / CL_NW : 1, KS = NT >= CL_NW ? 1 : CL_NW / NT;
  constexpr int KW = K / KS, STEPS = KW / 4, G = STEPS % 16 == 0 ? 16 : STEPS, NG = STEPS / G;
  static_assert(N % 16 == 0 && K % (4 * KS) == 0 && STEPS % G == 0 && (NT >= CL_NW ? NT % CL_NW == 0 : CL_NW % NT == 0), "tile shapes");
  const int tile0 = KS == 1 ? wave * TPW : wave % NT, ks = KS == 1 ? 0 : wave / NT;
  const int n0 = tile0 * 16, kb = ks * KW;
  cl_acc_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const float bv = KS == 1 ? bias[n0 + t * 16 + li] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = bv;
  }
  float bcur[TPW][G], bnxt[TPW][G], bnx2[TPW][G];
  const float* wp = Wt + (size_t)(kb + lq) * N + n0 + li;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bcur[t][g] = wp[(size_t)(4 * g) * N + t * 16];
      bnxt[t][g] = NG > 1 ? wp[(size_t)(4 * G + 4 * g) * N + t * 16] : 0.f;
    }
#pragma unroll 1
  for (int grp = 0; grp < NG; ++grp) {
    const int k0 = grp * 4 * G;
    if (grp + 2 < NG) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) bnx2[t][g] = wp[(size_t)(k0 + 8 * G + 4 * g) * N + t * 16];
    }
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] = Xin[(kb + k0 + 4 * g + lq) * CL_P + li];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], bcur[t][g], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int g = 0; g < G; ++g) { bcur[t][g] = bnxt[t][g]; bnxt[t][g] = bnx2[t][g]; }
  }
  if (KS == 1) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * lq + r, n = n0 + t * 16 + li;
        float v = acc[t][r];
        if (LEAKY) v = v > 0.f ? v : v * CL_SLOPE;
        Xout[n * CL_P + m] = v;
        if (gsave && m < rows) gsave[(row0 + m) * N + n] = v;
      }
    return;
  }
  // split reduction: partial tiles [wave][column in tile][row] -> LDS, then every thread finishes N * 16 / 512 outputs (column
  // fastest: coalesced global rows, conflict-free LDS)
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave * (16 * CL_P) + li * CL_P + 4 * lq + r] = acc[0][r];
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < N * CL_R / CL_NT; ++tt) {
    const int e = threadIdx.x + tt * CL_NT;
    const int m = e / N, n = e - m * N, t = n >> 4, nl = n & 15;
    float v = bias[n];
#pragma unroll
    for (int q = 0; q < KS; ++q) v += part[(q * NT + t) * (16 * CL_P) + nl * CL_P + m];
    if (LEAKY) v = v > 0.f ? v : v * CL_SLOPE;
    Xout[n * CL_P + m] = v;
    if (gsave && m < rows) gsave[(row0 + m) * N + n] = v;
  }
}

// dX[16 rows][KO] = (dY[16][N] W[N][KO]) * LeakyReLU'(a), dY k-major in LDS (index n), W as stored; Dout k-major (index = output
// column).  a: the layer's saved post-activation, row-major [B][KO] in global memory.  Same wave / split scheme as the forward.
template <int N, int KO>
__device__ __forceinline__ void cl_dense_bwd(const float* __restrict__ Din, float* __restrict__ Dout, const float* __restrict__ W,
                                             const float* __restrict__ act, float* part, size_t row0, int rows, int wave, int li, int lq) {
  constexpr int NT = KO / 16, TPW = NT >= CL_NW ? NT / CL_NW : 1, KS = NT >= CL_NW ? 1 : CL_NW / NT;
  constexpr int NWD = N / KS, STEPS = NWD / 4, G = STEPS % 16 == 0 ? 16 : STEPS, NG = STEPS / G;
  static_assert(KO % 16 == 0 && N % (4 * KS) == 0 && STEPS % G == 0 && (NT >= CL_NW ? NT % CL_NW == 0 : CL_NW % NT == 0), "tile shapes");
  const int tile0 = KS == 1 ? wave * TPW : wave % NT, ks = KS == 1 ? 0 : wave / NT;
  const int j0 = tile0 * 16, nb = ks * NWD;
  cl_acc_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  // the activation signs this thread will need (its tiles' outputs, or its share of the combine pass): requested now
  float av[TPW][4];
  constexpr int CPT = KS == 1 ? 1 : KO * CL_R / CL_NT;
  float ac[CPT];
  if (KS == 1) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) av[t][r] = act[(row0 + min(4 * lq + r, rows - 1)) * KO + j0 + t * 16 + li];
  } else {
#pragma unroll
    for (int t = 0; t < CPT; ++t) { const int e = threadIdx.x + t * CL_NT, m = e / KO; ac[t] = act[(row0 + min(m, rows - 1)) * KO + (e - m * KO)]; }
  }
  float bcur[TPW][G], bnxt[TPW][G], bnx2[TPW][G];
  const float* wp = W + (size_t)(nb + lq) * KO + j0 + li;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bcur[t][g] = wp[(size_t)(4 * g) * KO + t * 16];
      bnxt[t][g] = NG > 1 ? wp[(size_t)(4 * G + 4 * g) * KO + t * 16] : 0.f;
    }
#pragma unroll 1
  for (int grp = 0; grp < NG; ++grp) {
    const int n0 = grp * 4 * G;
    if (grp + 2 < NG) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) bnx2[t][g] = wp[(size_t)(n0 + 8 * G + 4 * g) * KO + t * 16];
    }
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] = Din[(nb + n0 + 4 * g + lq) * CL_P + li];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], bcur[t][g], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int g = 0; g < G; ++g) { bcur[t][g] = bnxt[t][g]; bnxt[t][g] = bnx2[t][g]; }
  }
  if (KS == 1) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Dout[(j0 + t * 16 + li) * CL_P + 4 * lq + r] = acc[t][r] * (av[t][r] > 0.f ? 1.f : CL_SLOPE);
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave * (16 * CL_P) + li * CL_P + 4 * lq + r] = acc[0][r];
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < CPT; ++tt) {
    const int e = threadIdx.x + tt * CL_NT;
    const int m = e / KO, j = e - m * KO, t = j >> 4, jl = j & 15;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < KS; ++q) v += part[(q * NT + t) * (16 * CL_P) + jl * CL_P + m];
    Dout[j * CL_P + m] = v * (ac[tt] > 0.f ? 1.f : CL_SLOPE);
  }
}

struct alignas(16) ClsSmem {
  float X[2][CL_W * CL_P];                 // activation ping-pong, k-major
  float part[CL_NW * CL_R * CL_P];         // partial tiles of the split reductions, [wave][column][row]
};

struct ClsFwdArgs {
  const float* x; int B; ClsFwdW w; float* a1; float* a2; float* a3; float* a4; float* logits;
  const int64_t* target; float ce_scale; float* dlogits; float* rowloss;       // target != nullptr: the cross-entropy tail (below)
};
struct ClsBwdArgs { const float* dlogits; int B; ClsBwdW w; const float* a1; const float* a2; const float* a3; const float* a4; float* dx; };

__device__ __forceinline__ void classifier_fwd_body(const ClsFwdArgs& c, ClsSmem& s, int bid) {
  const float* __restrict__ x = c.x; float* __restrict__ a1 = c.a1; float* __restrict__ a2 = c.a2; float* __restrict__ a3 = c.a3;
  float* __restrict__ a4 = c.a4; float* __restrict__ logits = c.logits;
  const int B = c.B; const ClsFwdW& w = c.w;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const size_t row0 = (size_t)bid * CL_R;
  const int rows = min(CL_R, B - (int)row0);
  PCG_T(0);
  // the block's input rows, k-major, with the zero rows that pad the reduction to 20
  if (threadIdx.x < CL_R * CL_INP) {
    const int m = threadIdx.x / CL_INP, k = threadIdx.x - m * CL_INP;
    const float v = x[(row0 + min(m, rows - 1)) * CL_IN + min(k, CL_IN - 1)];
    s.X[0][k * CL_P + m] = (m < rows && k < CL_IN) ? v : 0.f;
  }
  __syncthreads();
  PCG_T(1);
  cl_dense_fwd<CL_INP, CL_H1, true>(s.X[0], s.X[1], w.wt[0], w.b[0], a1, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  PCG_T(2);
  cl_dense_fwd<CL_H1, CL_H2, true>(s.X[1], s.X[0], w.wt[1], w.b[1], a2, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  PCG_T(3);
  cl_dense_fwd<CL_H2, CL_H3, true>(s.X[0], s.X[1], w.wt[2], w.b[2], a3, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  PCG_T(4);
  cl_dense_fwd<CL_H3, CL_H4, true>(s.X[1], s.X[0], w.wt[3], w.b[3], a4, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  PCG_T(5);
  // Linear(64 -> 4): 64 outputs, one per thread of the first wave
  if (threadIdx.x < CL_R * CL_OUT) {
    const int m = threadIdx.x >> 2, col = threadIdx.x & 3;
    float acc = w.b[4][col];
    const float* wr = w.wt[4] + col * CL_H4;
#pragma unroll 8
    for (int k = 0; k < CL_H4; ++k) acc = fmaf(s.X[0][k * CL_P + m], wr[k], acc);
    if (m < rows) logits[(row0 + m) * CL_OUT + col] = acc;
    if (c.target) s.part[threadIdx.x] = acc;
  }
  PCG_T(6);
  // Cross-entropy tail (trainer.py:302, reduction mean): one thread per row with cross_entropy_kernel's expressions — the row's loss
  // term lse - z[t] goes to rowloss (the launch that logs the scalars adds them in that kernel's order) and the gradient
  // ce_scale / B * (softmax - onehot) to dlogits.  The same bits as pcg_cross_entropy_fwd_bwd on these logits.
  if (c.target) {                                      // kernel-uniform
    __syncthreads();
    if (threadIdx.x < CL_R && (int)threadIdx.x < rows) {
      const int m = threadIdx.x;
      const float r[4] = {s.part[4 * m], s.part[4 * m + 1], s.part[4 * m + 2], s.part[4 * m + 3]};
      const float g = c.ce_scale * 1.f / (float)B;
      float mx = r[0];
      for (int k = 1; k < 4; ++k) mx = fmaxf(mx, r[k]);
      float se = 0.f;
      for (int k = 0; k < 4; ++k) se += expf(r[k] - mx);
      const float lse = mx + logf(se);
      const int64_t tv = c.target[row0 + m];
      const int t = tv < 0 ? 0 : (tv >= 4 ? 3 : (int)tv);
      c.rowloss[row0 + m] = lse - r[t];
      float o[4];
      for (int k = 0; k < 4; ++k) o[k] = g * (expf(r[k] - lse) - (k == t ? 1.f : 0.f));
      *reinterpret_cast<float4*>(c.dlogits + (row0 + m) * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

__device__ __forceinline__ void classifier_bwd_body(const ClsBwdArgs& c, ClsSmem& s, int bid) {
  const float* __restrict__ dlogits = c.dlogits; const float* __restrict__ a1 = c.a1; const float* __restrict__ a2 = c.a2;
  const float* __restrict__ a3 = c.a3; const float* __restrict__ a4 = c.a4; float* __restrict__ dx = c.dx;
  const int B = c.B; const ClsBwdW& w = c.w;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const size_t row0 = (size_t)bid * CL_R;
  const int rows = min(CL_R, B - (int)row0);
  // d4[m][k] = (sum_c dlogits[m][c] W5[c][k]) * LeakyReLU'(a4[m][k]): 1024 outputs, two per thread
  {
    constexpr int PER = CL_R * CL_H4 / CL_NT;               // element tid + 512 t of [16][64]: column tid % 64 for every t
    const int k = threadIdx.x & (CL_H4 - 1);
    float w5[CL_OUT], dl[PER][CL_OUT], av[PER];
#pragma unroll
    for (int c = 0; c < CL_OUT; ++c) w5[c] = w.w[4][c * CL_H4 + k];
#pragma unroll
    for (int t = 0; t < PER; ++t) {                          // all loads first: a load-use loop pays one memory latency per trip
      const int m = min((int)(threadIdx.x + t * CL_NT) / CL_H4, rows - 1);
#pragma unroll
      for (int c = 0; c < CL_OUT; ++c) dl[t][c] = dlogits[(row0 + m) * CL_OUT + c];
      av[t] = a4[(row0 + m) * CL_H4 + k];
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int m = (threadIdx.x + t * CL_NT) / CL_H4;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < CL_OUT; ++c) acc = fmaf(dl[t][c], w5[c], acc);
      s.X[0][k * CL_P + m] = m < rows ? acc * (av[t] > 0.f ? 1.f : CL_SLOPE) : 0.f;
    }
  }
  __syncthreads();
  cl_dense_bwd<CL_H4, CL_H3>(s.X[0], s.X[1], w.w[3], a3, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  cl_dense_bwd<CL_H3, CL_H2>(s.X[1], s.X[0], w.w[2], a2, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  cl_dense_bwd<CL_H2, CL_H1>(s.X[0], s.X[1], w.w[1], a1, s.part, row0, rows, wave, li, lq);
  __syncthreads();
  // dx[16][17] = d1[16][256] W1[256][17]: two 16-column output tiles (the second holds column 16 alone); wave = (tile, quarter of the
  // 256): 16 MFMA steps each, the partial tiles added in quarter order.  Lanes past column 16 feed zeros (clamped address, dropped value).
  {
    constexpr int G = 16, NQ = CL_H1 / 4;                 // 64 reduction indices = 16 steps per wave: one group
    const int tile = wave & 1, qt = wave >> 1, j = tile * 16 + li;
    cl_acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.f;
    const float* wp = w.w[0] + (size_t)(qt * NQ + lq) * CL_IN + min(j, CL_IN - 1);
    float a[G], b[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float bv = wp[(size_t)(4 * g) * CL_IN];
      b[g] = j < CL_IN ? bv : 0.f;
      a[g] = s.X[1][(qt * NQ + 4 * g + lq) * CL_P + li];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b[g], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) s.part[wave * (16 * CL_P) + li * CL_P + 4 * lq + r] = acc[r];
  }
  __syncthreads();
  if (threadIdx.x < CL_R * CL_IN) {
    const int m = threadIdx.x / CL_IN, i = threadIdx.x - m * CL_IN, tile = i >> 4, il = i & 15;
    if (m < rows) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) v += s.part[(q * 2 + tile) * (16 * CL_P) + il * CL_P + m];
      dx[(row0 + m) * CL_IN + i] = v;
    }
  }
}

__global__ void __launch_bounds__(CL_NT) classifier_fwd_kernel(ClsFwdArgs c) {
  __shared__ ClsSmem s;
  classifier_fwd_body(c, s, blockIdx.x);
}
__global__ void __launch_bounds__(CL_NT) classifier_bwd_kernel(ClsBwdArgs c) {
  __shared__ ClsSmem s;
  classifier_bwd_body(c, s, blockIdx.x);
}

// Riders (see pcg_house_residual_fwd_sn): the frozen classifier's term of the generator loss does not depend on the critic update, so
// its two launches carry the critic's spectral-norm work that sits on the chain at the same time — the first n blocks one matrix
// each (256 of the 512 threads; the other four waves end at once, and a block barrier only waits for the waves that are left), the
// blocks behind them the classifier body.  The two bodies share the block's LDS.  Same bodies, same bits as the separate launches.
// The rider's blocks come FIRST in the grid: a classifier block is eight waves of 138 registers, so a CU holds one, and blocks behind
// the 256 of a full batch would only start when one of those ends (measured: the launch then takes the SUM of the two bodies); a
// rider block that started first leaves room for a classifier block beside its four live waves.
constexpr size_t CL_RIDER_LDS = sizeof(ClsSmem) > sizeof(SnFwdLds) ? sizeof(ClsSmem) : sizeof(SnFwdLds);
__global__ void __launch_bounds__(CL_NT) classifier_fwd_snbwd_kernel(ClsFwdArgs c, SnBwdBatch b, SnBwdExtra x, int n, int passes) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CL_RIDER_LDS];
  if ((int)blockIdx.x >= n) { classifier_fwd_body(c, *reinterpret_cast<ClsSmem*>(lds), blockIdx.x - n); return; }
  if (threadIdx.x >= 256) return;
  spectral_norm_bwd_seq_body<true>(b, x, n, passes, blockIdx.x, *reinterpret_cast<SnBwdLds*>(lds));
}
__global__ void __launch_bounds__(CL_NT) classifier_bwd_snfwd_kernel(ClsBwdArgs c, SnFwdBatch b, float eps, int n, int reps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CL_RIDER_LDS];
  if ((int)blockIdx.x >= n) { classifier_bwd_body(c, *reinterpret_cast<ClsSmem*>(lds), blockIdx.x - n); return; }
  if (threadIdx.x >= 256) return;
  const int l = blockIdx.x;
  spectral_norm_fwd_body(b.W[l], b.O[l], b.I[l], b.u[l], b.v[l], eps, 1, b.Wbar + l, b.sigma + l, b.uu + l, b.vu + l, reps, n,
                         *reinterpret_cast<SnFwdLds*>(lds));
}

}  // namespace
}  // namespace pcg

using namespace pcg;

namespace {
int fill_cls_fwd(ClsFwdArgs& c, const float* x, int32_t B, const float* const* w_kmajor, const float* const* bias, float* a1, float* a2, float* a3,
                 float* a4, float* logits) {
  PCG_REQUIRE(x && B > 0 && w_kmajor && bias && a1 && a2 && a3 && a4 && logits, "pcg_house_classifier_fwd: bad arguments");
  for (int l = 0; l < 5; ++l) { PCG_REQUIRE(w_kmajor[l] && bias[l], "pcg_house_classifier_fwd: null layer %d", l); c.w.wt[l] = w_kmajor[l]; c.w.b[l] = bias[l]; }
  c.x = x; c.B = B; c.a1 = a1; c.a2 = a2; c.a3 = a3; c.a4 = a4; c.logits = logits;
  return PCG_OK;
}
int fill_cls_bwd(ClsBwdArgs& c, const float* dlogits, int32_t B, const float* const* w_stored, const float* a1, const float* a2, const float* a3,
                 const float* a4, float* dx) {
  PCG_REQUIRE(dlogits && B > 0 && w_stored && a1 && a2 && a3 && a4 && dx, "pcg_house_classifier_bwd: bad arguments");
  for (int l = 0; l < 5; ++l) { PCG_REQUIRE(w_stored[l], "pcg_house_classifier_bwd: null layer %d", l); c.w.w[l] = w_stored[l]; }
  c.dlogits = dlogits; c.B = B; c.a1 = a1; c.a2 = a2; c.a3 = a3; c.a4 = a4; c.dx = dx;
  return PCG_OK;
}
}  // namespace

extern "C" int pcg_house_classifier_fwd(const float* x, int32_t B, const float* const* w_kmajor, const float* const* bias, float* a1, float* a2,
                                        float* a3, float* a4, float* logits, pcg_stream_t stream) {
  ClsFwdArgs c{};
  if (int e = fill_cls_fwd(c, x, B, w_kmajor, bias, a1, a2, a3, a4, logits)) return e;
  hipLaunchKernelGGL(classifier_fwd_kernel, dim3((B + CL_R - 1) / CL_R), dim3(CL_NT), 0, (hipStream_t)stream, c);
  return launch_status("classifier_fwd_kernel");
}

extern "C" int pcg_house_classifier_bwd(const float* dlogits, int32_t B, const float* const* w_stored, const float* a1, const float* a2,
                                        const float* a3, const float* a4, float* dx, pcg_stream_t stream) {
  ClsBwdArgs c{};
  if (int e = fill_cls_bwd(c, dlogits, B, w_stored, a1, a2, a3, a4, dx)) return e;
  hipLaunchKernelGGL(classifier_bwd_kernel, dim3((B + CL_R - 1) / CL_R), dim3(CL_NT), 0, (hipStream_t)stream, c);
  return launch_status("classifier_bwd_kernel");
}

// pcg_house_classifier_fwd + pcg_spectral_norm_bwd_batched_seq as ONE launch
extern "C" int pcg_house_classifier_fwd_snbwd(const float* x, int32_t B, const float* const* w_kmajor, const float* const* bias, float* a1, float* a2,
                                              float* a3, float* a4, float* logits, int32_t n, int32_t passes, const float* const* dw_bar,
                                              const float* const* w_bar, const int32_t* out_features, const int32_t* in_features,
                                              const float* const* u, const float* const* v, const float* const* sigma, float* const* dw_orig,
                                              const int32_t* accumulate, float* const* db_dst, const float* const* db_src,
                                              const int64_t* ce_target, float ce_grad_scale, float* ce_dlogits, float* ce_row_loss,
                                              pcg_stream_t stream) {
  ClsFwdArgs c{};
  if (int e = fill_cls_fwd(c, x, B, w_kmajor, bias, a1, a2, a3, a4, logits)) return e;
  if (ce_target) {
    PCG_REQUIRE(ce_dlogits && ce_row_loss && (reinterpret_cast<uintptr_t>(ce_dlogits) & 15) == 0,
                "pcg_house_classifier_fwd_snbwd: the cross-entropy tail needs dlogits (16-byte aligned) and row_loss");
    c.target = ce_target; c.ce_scale = ce_grad_scale; c.dlogits = ce_dlogits; c.rowloss = ce_row_loss;
  }
  SnBwdBatch b{};
  SnBwdExtra xx{};
  if (int e = fill_sn_bwd_batch(b, xx, n, passes, dw_bar, w_bar, out_features, in_features, u, v, sigma, dw_orig, accumulate, db_dst, db_src)) return e;
  const int ncls = (B + CL_R - 1) / CL_R;
  hipLaunchKernelGGL(classifier_fwd_snbwd_kernel, dim3(ncls + n), dim3(CL_NT), 0, (hipStream_t)stream, c, b, xx, n, passes);
  return launch_status("classifier_fwd_snbwd_kernel");
}

// pcg_house_classifier_bwd + pcg_spectral_norm_fwd_batched_reps (training mode) as ONE launch
extern "C" int pcg_house_classifier_bwd_snfwd(const float* dlogits, int32_t B, const float* const* w_stored, const float* a1, const float* a2,
                                              const float* a3, const float* a4, float* dx, int32_t n, int32_t reps, const float* const* w_orig,
                                              const int32_t* out_features, const int32_t* in_features, float* const* u, float* const* v, float eps,
                                              float* const* w_bar, float* const* sigma, float* const* u_used, float* const* v_used,
                                              pcg_stream_t stream) {
  ClsBwdArgs c{};
  if (int e = fill_cls_bwd(c, dlogits, B, w_stored, a1, a2, a3, a4, dx)) return e;
  SnFwdBatch b{};
  if (int e = fill_sn_fwd_batch(b, n, reps, w_orig, out_features, in_features, u, v, 1, w_bar, sigma, u_used, v_used)) return e;
  const int ncls = (B + CL_R - 1) / CL_R;
  hipLaunchKernelGGL(classifier_bwd_snfwd_kernel, dim3(ncls + n), dim3(CL_NT), 0, (hipStream_t)stream, c, b, eps, n, reps);
  return launch_status("classifier_bwd_snfwd_kernel");
}
