// house_critic_fused.hip — the tabular spectral-norm critic (house_sales_kc_usa/models/discriminator.py:5-20:
// Linear 21->32, 32->64, 64->128 with LeakyReLU(0.2), Linear 128->1) as ONE forward and ONE backward kernel on the matrix cores: a
// block owns 32 batch rows, the four waves split the 32-column output tiles of a layer (or, where a layer is a single tile, its
// reduction).  The weights are the spectral-normalised W / sigma that pcg_spectral_norm_fwd_batched produced; the weight gradients
// are reduced afterwards by pcg_linear_wgrad_grouped from the per-layer pre-activation gradients this backward writes.  Widths are
// the reference configuration (input 17 + 4 classes, hidden 32): compile-time.
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int C0 = 21, C1 = 32, C2 = 64, C3 = 128;      // layer widths (input_dim + num_classes, hidden, 2*hidden, 4*hidden)

struct CW { const float* w[4]; const float* b[4]; };
// Up to two passes of the critic in one launch (blockIdx.y): D(real) and D(fake) of the critic step share nothing but the module —
// each has its own spectral-norm weights (two successive power iterations), inputs and outputs.
constexpr int CP = 2;
struct CFwdPass { const float* x; const float* onehot; CW p; float* a0; float* a1; float* a2; float* a3; float* out; };
struct CFwdArgs { CFwdPass ps[CP]; };
struct CBwdPass { const float* dout; CW p; const float* a1; const float* a2; const float* a3; float* d3o; float* d2o; float* d1o; float* dx; };
struct CBwdArgs { CBwdPass ps[CP]; };

// ---- the layers on the matrix cores ------------------------------------------------------------------------------------------------
// A block owns 32 rows and runs the net on v_mfma_f32_32x32x2_f32 (exact fp32), everything in LDS:
//   * all three weight matrices are staged once, AS STORED ([out][in]) with a row pitch of in + 1 floats (odd): the forward's B
//     operand B[k][n] = W[n][k] is then a column walk (bank = n * pitch + k: conflict-free), the backward's B[n][j] = W[n][j] a row
//     walk — one image serves both directions, and the staging is a coalesced burst at kernel entry;
//   * activations / gradients sit k-major, X[k][row] with pitch 33: the A operand of a k-step is two rows of 32 floats.
// The first version gave a thread (row, quarter of the output columns) and read the weights as LDS broadcasts: every k-step cost each
// wave 8 ds_read_b128 for 32 FMAs per lane, and four waves shared one LDS — 24 us per pass, LDS-bandwidth bound.
constexpr int MR = 32, MP = MR + 1;                       // rows per block, k-row pitch
constexpr int C0P = 22;                                   // input width padded to an even reduction length (column 21 is zero)
constexpr int P1 = C0P + 1, P2 = C1 + 1, P3 = C2 + 1;     // weight row pitches (odd)
typedef float cm_acc_t __attribute__((ext_vector_type(16)));
struct alignas(16) CritSmem {
  float W1[C1 * P1], W2[C2 * P2], W3[C3 * P3];            // 736 + 2112 + 8320 floats
  float w4[C3], b1[C1], b2[C2], b3[C3];
  float X[2][C3 * MP];                                    // activation ping-pong, k-major
  float part[4][MR * MP];                                 // split-reduction partial tiles
};
__device__ __forceinline__ int cm_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
// global [N][K] -> LDS [N][pitch] (pad columns zeroed), coalesced.  Two steps — every load of the kernel's staging is requested
// before the first LDS store (a load-then-store loop waits out one memory latency per trip: 33 trips for the widest matrix).
template <int N, int K>
struct CmRegs { float v[(N * K + 255) / 256]; };
template <int N, int K>
__device__ __forceinline__ void cm_load(CmRegs<N, K>& r, const float* __restrict__ W) {
#pragma unroll
  for (int t = 0; t < (N * K + 255) / 256; ++t) r.v[t] = W[min((int)threadIdx.x + t * 256, N * K - 1)];
}
template <int N, int K, int PITCH>
__device__ __forceinline__ void cm_store(float* Wl, const CmRegs<N, K>& r) {
#pragma unroll
  for (int t = 0; t < (N * K + 255) / 256; ++t) {
    const int e = threadIdx.x + t * 256;
    if (e < N * K) { const int n = e / K, k = e - n * K; Wl[n * PITCH + k] = r.v[t]; }
  }
  for (int e = threadIdx.x; e < N * (PITCH - K); e += 256) { const int n = e / (PITCH - K), k = K + (e - n * (PITCH - K)); Wl[n * PITCH + k] = 0.f; }
}
// acc (+)= X[32 rows][k in k0 .. k0 + 2*steps) * B, B[k][n] = Wl[(n0 + n) * PITCH + k]      (forward: W as stored is [n][k])
template <int PITCH>
__device__ __forceinline__ void cm_mma_fwd(cm_acc_t& acc, const float* X, const float* Wl, int n0, int k0, int steps, int li, int lh) {
#pragma unroll 4
  for (int st = 0; st < steps; ++st) {
    const int k = k0 + 2 * st + lh;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[k * MP + li], Wl[(n0 + li) * PITCH + k], acc, 0, 0, 0);
  }
}
// acc (+)= D[32 rows][n in n0 .. n0 + 2*steps) * B, B[n][j] = Wl[n * PITCH + j0 + j]           (backward: dX = dY W)
template <int PITCH>
__device__ __forceinline__ void cm_mma_bwd(cm_acc_t& acc, const float* D, const float* Wl, int j0, int n0, int steps, int li, int lh) {
#pragma unroll 4
  for (int st = 0; st < steps; ++st) {
    const int n = n0 + 2 * st + lh;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(D[n * MP + li], Wl[n * PITCH + j0 + li], acc, 0, 0, 0);
  }
}
__device__ __forceinline__ cm_acc_t cm_zero() { cm_acc_t z; for (int r = 0; r < 16; ++r) z[r] = 0.f; return z; }

__global__ void __launch_bounds__(256) critic_fwd_kernel(CFwdArgs args, int D, int NC, int B, float slope) {
  const CFwdPass& ps = args.ps[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) unsigned char crit_lds[];
  CritSmem& s = *reinterpret_cast<CritSmem*>(crit_lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const size_t row0 = (size_t)blockIdx.x * MR;
  const int rows = min(MR, B - (int)row0);
  // ---- one burst: the three weight images, biases, the last layer's row, the block's input rows
  CmRegs<C1, C0> r1; CmRegs<C2, C1> r2; CmRegs<C3, C2> r3;
  cm_load<C1, C0>(r1, ps.p.w[0]); cm_load<C2, C1>(r2, ps.p.w[1]); cm_load<C3, C2>(r3, ps.p.w[2]);
  const int tq = threadIdx.x & (C3 - 1);
  const float w4v = ps.p.w[3][tq], b3v = ps.p.b[2][tq], b2v = ps.p.b[1][tq & (C2 - 1)], b1v = ps.p.b[0][tq & (C1 - 1)];
  float xin[(MR * C0P + 255) / 256];                       // torch.cat([x, target_onehot], 1) (:19): elements tid, tid + 256, ...
#pragma unroll
  for (int t = 0; t < (MR * C0P + 255) / 256; ++t) {
    const int e = threadIdx.x + t * 256, m = min(e / C0P, rows - 1), k = min(e - (e / C0P) * C0P, C0 - 1);
    xin[t] = k < D ? ps.x[(row0 + m) * D + k] : ps.onehot[(row0 + m) * NC + (k - D)];
  }
  cm_store<C1, C0, P1>(s.W1, r1); cm_store<C2, C1, P2>(s.W2, r2); cm_store<C3, C2, P3>(s.W3, r3);
  if (threadIdx.x < C3) { s.w4[threadIdx.x] = w4v; s.b3[threadIdx.x] = b3v; }
  if (threadIdx.x < C2) s.b2[threadIdx.x] = b2v;
  if (threadIdx.x < C1) s.b1[threadIdx.x] = b1v;
#pragma unroll
  for (int t = 0; t < (MR * C0P + 255) / 256; ++t) {
    const int e = threadIdx.x + t * 256;
    if (e < MR * C0P) {
      const int m = e / C0P, k = e - m * C0P;
      const float v = (m < rows && k < C0) ? xin[t] : 0.f;   // + the zero pad row k = 21
      if (m < rows && k < C0) ps.a0[(row0 + m) * C0 + k] = v;
      s.X[0][k * MP + m] = v;
    }
  }
  __syncthreads();
  // layer 1: 22 -> 32, one tile: wave 0
  if (wave == 0) {
    cm_acc_t acc; const float bv = s.b1[li];
    for (int r = 0; r < 16; ++r) acc[r] = bv;
    cm_mma_fwd<P1>(acc, s.X[0], s.W1, 0, 0, C0P / 2, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = cm_row(r, lh);
      const float v = acc[r] > 0.f ? acc[r] : acc[r] * slope;
      s.X[1][li * MP + m] = v;
      if (m < rows) ps.a1[(row0 + m) * C1 + li] = v;
    }
  }
  __syncthreads();
  // layer 2: 32 -> 64, two tiles: waves 0, 1
  if (wave < 2) {
    const int n0 = wave * 32;
    cm_acc_t acc; const float bv = s.b2[n0 + li];
    for (int r = 0; r < 16; ++r) acc[r] = bv;
    cm_mma_fwd<P2>(acc, s.X[1], s.W2, n0, 0, C1 / 2, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = cm_row(r, lh);
      const float v = acc[r] > 0.f ? acc[r] : acc[r] * slope;
      s.X[0][(n0 + li) * MP + m] = v;
      if (m < rows) ps.a2[(row0 + m) * C2 + n0 + li] = v;
    }
  }
  __syncthreads();
  // layer 3: 64 -> 128, four tiles: one per wave
  {
    const int n0 = wave * 32;
    cm_acc_t acc; const float bv = s.b3[n0 + li];
    for (int r = 0; r < 16; ++r) acc[r] = bv;
    cm_mma_fwd<P3>(acc, s.X[0], s.W3, n0, 0, C2 / 2, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = cm_row(r, lh);
      const float v = acc[r] > 0.f ? acc[r] : acc[r] * slope;
      s.X[1][(n0 + li) * MP + m] = v;
      if (m < rows) ps.a3[(row0 + m) * C3 + n0 + li] = v;
    }
  }
  __syncthreads();
  // Linear(128 -> 1): eight threads per row take 16 inputs each (ascending), then a fixed-order sum of the eight
  {
    const int m = threadIdx.x >> 3, sgm = threadIdx.x & 7;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(s.w4[sgm * 16 + k], s.X[1][(sgm * 16 + k) * MP + m], acc);
    s.part[0][m * 8 + sgm] = acc;
  }
  __syncthreads();
  if (threadIdx.x < MR && threadIdx.x < rows) {
    const float* q = s.part[0] + threadIdx.x * 8;
    ps.out[row0 + threadIdx.x] = ps.p.b[3][0] + (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])));
  }
}

// pre-activation gradients d3, d2, d1 (operands of the weight gradients) and, optionally, the gradient of the first D inputs
__global__ void __launch_bounds__(256) critic_bwd_kernel(CBwdArgs args, int B, float slope, int D) {
  const CBwdPass& ps = args.ps[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) unsigned char crit_lds[];
  CritSmem& s = *reinterpret_cast<CritSmem*>(crit_lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const size_t row0 = (size_t)blockIdx.x * MR;
  const int rows = min(MR, B - (int)row0);
  // ---- one burst: the weight images, the last layer's row, and the operands of d3 (a3 rows, dout)
  CmRegs<C1, C0> r1; CmRegs<C2, C1> r2; CmRegs<C3, C2> r3;
  cm_load<C1, C0>(r1, ps.p.w[0]); cm_load<C2, C1>(r2, ps.p.w[1]); cm_load<C3, C2>(r3, ps.p.w[2]);
  const int k3 = threadIdx.x & (C3 - 1);                   // this thread's column of every d3 element it makes (256 % 128 == 0)
  const float w4v = ps.p.w[3][k3];
  float a3v[MR * C3 / 256], dov[MR * C3 / 256];
#pragma unroll
  for (int t = 0; t < MR * C3 / 256; ++t) {
    const int m = min((int)(threadIdx.x + t * 256) / C3, rows - 1);
    a3v[t] = ps.a3[(row0 + m) * C3 + k3]; dov[t] = ps.dout[row0 + m];
  }
  float a2v[MR * C2 / 256], a1v[MR * C1 / 256];              // the masks of the later layers: element tid + 256 t of [32][64], [32][32]
#pragma unroll
  for (int t = 0; t < MR * C2 / 256; ++t) a2v[t] = ps.a2[(row0 + min((int)(threadIdx.x + t * 256) / C2, rows - 1)) * C2 + (threadIdx.x & (C2 - 1))];
#pragma unroll
  for (int t = 0; t < MR * C1 / 256; ++t) a1v[t] = ps.a1[(row0 + min((int)(threadIdx.x + t * 256) / C1, rows - 1)) * C1 + (threadIdx.x & (C1 - 1))];
  cm_store<C1, C0, P1>(s.W1, r1); cm_store<C2, C1, P2>(s.W2, r2); cm_store<C3, C2, P3>(s.W3, r3);
  // d3 = dout * w4 * LeakyReLU'(a3): 32 x 128, sixteen per thread; coalesced rows of a3 in, d3 out
#pragma unroll
  for (int t = 0; t < MR * C3 / 256; ++t) {
    const int m = (threadIdx.x + t * 256) / C3;
    float v = 0.f;
    if (m < rows) {
      v = dov[t] * w4v * (a3v[t] > 0.f ? 1.f : slope);
      ps.d3o[(row0 + m) * C3 + k3] = v;
    }
    s.X[0][k3 * MP + m] = v;
  }
  __syncthreads();
  // d2 = (d3 W3) * LeakyReLU'(a2): reduction 128, two output tiles: wave (tile, half of the reduction); halves added in order
  {
    const int tile = wave & 1, half = wave >> 1, j0 = tile * 32;
    cm_acc_t acc = cm_zero();
    cm_mma_bwd<P3>(acc, s.X[0], s.W3, j0, half * (C3 / 2), C3 / 4, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) s.part[wave][li * MP + cm_row(r, lh)] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MR * C2 / 256; ++t) {
    const int e = threadIdx.x + t * 256;
    const int m = e / C2, j = e - m * C2, tile = j >> 5, jj = j & 31;
    float v = 0.f;
    if (m < rows) {
      v = (s.part[tile][jj * MP + m] + s.part[2 + tile][jj * MP + m]) * (a2v[t] > 0.f ? 1.f : slope);
      ps.d2o[(row0 + m) * C2 + j] = v;
    }
    s.X[1][j * MP + m] = v;
  }
  __syncthreads();
  // d1 = (d2 W2) * LeakyReLU'(a1): reduction 64, one tile: a quarter of the reduction per wave, added in wave order
  {
    cm_acc_t acc = cm_zero();
    cm_mma_bwd<P2>(acc, s.X[1], s.W2, 0, wave * (C2 / 4), C2 / 8, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) s.part[wave][li * MP + cm_row(r, lh)] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MR * C1 / 256; ++t) {
    const int e = threadIdx.x + t * 256;
    const int m = e / C1, j = e - m * C1;
    float v = 0.f;
    if (m < rows) {
      v = (((s.part[0][j * MP + m] + s.part[1][j * MP + m]) + s.part[2][j * MP + m]) + s.part[3][j * MP + m]) * (a1v[t] > 0.f ? 1.f : slope);
      ps.d1o[(row0 + m) * C1 + j] = v;
    }
    s.X[0][j * MP + m] = v;
  }
  if (!ps.dx) return;                                        // block-uniform
  __syncthreads();
  // dx = d1 W1 (first D of the 21 input columns): reduction 32, one tile, a quarter per wave
  {
    cm_acc_t acc = cm_zero();
    cm_mma_bwd<P1>(acc, s.X[0], s.W1, 0, wave * (C1 / 4), C1 / 8, li, lh);      // columns >= 22 of the tile read pad / the next row: dropped below
#pragma unroll
    for (int r = 0; r < 16; ++r) s.part[wave][li * MP + cm_row(r, lh)] = acc[r];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < MR * D; e += 256) {
    const int m = e / D, i = e - m * D;
    if (m < rows) ps.dx[(row0 + m) * D + i] = ((s.part[0][i * MP + m] + s.part[1][i * MP + m]) + s.part[2][i * MP + m]) + s.part[3][i * MP + m];
  }
}

int set_lds(const void* fn) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CritSmem));
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
  hipLaunchKernelGGL(critic_fwd_kernel, dim3((B + MR - 1) / MR, n_pass), dim3(256), sizeof(CritSmem), (hipStream_t)stream, args, D, NC, B, slope);
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
  hipLaunchKernelGGL(critic_bwd_kernel, dim3((B + MR - 1) / MR, n_pass), dim3(256), sizeof(CritSmem), (hipStream_t)stream, args, B, slope, D);
  return launch_status("critic_bwd_kernel");
}

extern "C" int pcg_house_critic_bwd(const float* dout, int32_t B, int32_t D, const float* const* w_bar, float slope, const float* a1,
                                    const float* a2, const float* a3, float* d3, float* d2, float* d1, float* dx, pcg_stream_t stream) {
  PCG_REQUIRE(w_bar, "pcg_house_critic_bwd: bad arguments");
  return pcg_house_critic_bwd_n(1, &dout, B, D, w_bar, slope, &a1, &a2, &a3, &d3, &d2, &d1, &dx, stream);
}
