// tabular.hip — building blocks of the tabular ("prompted counterfactual") CounteRGAN step,
// conditional_counteRGAN/house_sales_kc_usa: every tensor is [batch][<=256 features], every weight fits in LDS.
// Generic small GEMM (any M, N, K — the layer widths 38, 21, 17, 10, 9, 30, 6, 2, 5, 13 are not multiples of 4, so these
// layers do not go through the MFMA implicit-GEMM path), FiLM, Gumbel-softmax heads, residual assembly, spectral
// normalisation, one-hot, column concat/split, mean.  Round-1 status: correct and graph-capturable, but one launch per
// op — the layer chain is launch-latency bound; the MI355X-shaped answer is one persistent fused kernel.
// Reference call sites are cited next to each prototype in include/pcgan_hip.h.
#include <algorithm>
#include "pcg_common.h"
#include "spectral_norm_body.h"

namespace pcg {
namespace {

unsigned ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// C[m][n] (+)= sum_k opA[m][k] * opB[k][n] (+ bias[n]);  opA = A (lda: row stride of [M][K]) or A^T (A stored [K][M]);
// opB = B ([K][N]) or B^T (B stored [N][K]).  64x64 tile per block, 4x4 outputs per thread, K step 16.
constexpr int GT = 64, GK = 16;
__global__ void __launch_bounds__(256) gemm_kernel(int transA, int transB, int M, int N, int K, const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                   const float* __restrict__ bias, int accumulate, int act_on, float neg) {
  __shared__ float As[GK][GT + 1], Bs[GK][GT + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += GK) {
    // consecutive threads walk the contiguous direction of each operand (k for row-major A / transposed B, else m / n)
    for (int e = threadIdx.x; e < GK * GT; e += 256) {
      {
        const int kk = transA ? e / GT : e % GK, r = transA ? e % GT : e / GK;
        const int k = k0 + kk, m = m0 + r;
        As[kk][r] = (k < K && m < M) ? (transA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k]) : 0.f;
      }
      {
        const int kk = transB ? e % GK : e / GT, r = transB ? e / GK : e % GT;
        const int k = k0 + kk, n = n0 + r;
        Bs[kk][r] = (k < K && n < N) ? (transB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j] + (bias ? bias[n] : 0.f);
      float* c = C + (size_t)m * ldc + n;
      if (accumulate) v += *c;
      if (act_on) v = act_neg_scale(v, neg);       // fused ReLU / LeakyReLU of the layer (after the accumulation, if any)
      *c = v;
    }
  }
}

// Skinny-N GEMM: C[M][N <= 32] (+)= A[M][K] opB (+ bias).  The 64x64 tile above gives such a problem one block per 64 rows (M = 4096,
// N = 17, K = 256 — the frozen classifier's grad-input to the 17 features: 64 blocks on 256 CUs, 16 latency-bound k-steps, 48 us).
// Here opB (K x N) sits in LDS whole, a block owns 16 rows (256 blocks at M = 4096) and a thread two columns of one row.
constexpr int SK_R = 16, SK_LDB = 33;
__global__ void __launch_bounds__(256) gemm_skinny_kernel(int transB, int M, int N, int K, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                          const float* __restrict__ bias, int accumulate, int act_on, float neg) {
  extern __shared__ float sk_lds[];
  float* Bs = sk_lds;                 // [K][33]
  float* As = sk_lds + K * SK_LDB;    // [16][K + 1]
  for (int e = threadIdx.x; e < K * N; e += 256) {
    const int k = transB ? e % K : e / N, n = transB ? e / K : e % N;
    Bs[k * SK_LDB + n] = transB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
  }
  const int m0 = blockIdx.x * SK_R;
  for (int e = threadIdx.x; e < SK_R * K; e += 256) {
    const int r = e / K, k = e - r * K, m = m0 + r;
    As[r * (K + 1) + k] = m < M ? A[(size_t)m * lda + k] : 0.f;
  }
  __syncthreads();
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const float* a = As + r * (K + 1);
  float acc0 = 0.f, acc1 = 0.f;
  const bool two = c + 16 < N;
  for (int k = 0; k < K; ++k) {
    const float av = a[k];
    acc0 = fmaf(av, Bs[k * SK_LDB + c], acc0);
    acc1 = fmaf(av, Bs[k * SK_LDB + c + 16], acc1);      // columns >= N of Bs are never written: the product is discarded below
  }
  const int m = m0 + r;
  if (m >= M) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = c + 16 * j;
    if (n >= N || (j == 1 && !two)) continue;
    float v = (j ? acc1 : acc0) + (bias ? bias[n] : 0.f);
    float* cp = C + (size_t)m * ldc + n;
    if (accumulate) v += *cp;
    if (act_on) v = act_neg_scale(v, neg);
    *cp = v;
  }
}
static bool launch_skinny(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          const float* bias, int accumulate, int act_on, float neg, hipStream_t s) {
  if (transA || N > 32 || N < 2 || K > 320 || M < 512) return false;
  const size_t lds = ((size_t)K * SK_LDB + (size_t)SK_R * (K + 1)) * sizeof(float);     // <= 63 KB at K = 320
  hipLaunchKernelGGL(gemm_skinny_kernel, dim3((M + SK_R - 1) / SK_R), dim3(256), lds, s, transB, M, N, K, A, lda, B, ldb, C, ldc, bias,
                     accumulate, act_on, neg);
  return true;
}

// dW[O][I] (+)= dy^T x, db[O] (+)= column sums of dy, reducing over the B rows: the batch is split into S slabs across
// blockIdx.z; every slab writes its partial tile, and the block that takes the last ticket of a tile adds the S partials in
// slab order (deterministic, no float atomics).  x is seen with a virtual column of ones at index I, whose "weight gradient"
// is the bias gradient.  tickets: caller-owned int32[tiles], zero before first use, left zero.
__device__ __forceinline__ void linear_wgrad_body(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx, int B, int O,
                                                  int I, float* __restrict__ dW, float* __restrict__ db, int accW, int accB, int S,
                                                  int chunk, float* __restrict__ partial, int* __restrict__ ticket, int tile_x, int tile_y,
                                                  int slab) {
  __shared__ float As[GK][GT + 1], Bs[GK][GT + 1];
  __shared__ int s_last;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = tile_y * GT, n0 = tile_x * GT;
  const int kbeg = slab * chunk, kend = min(B, kbeg + chunk);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // The slab is walked in groups of WG_NS k-steps whose operands are ALL requested before the first one is used: the loop is
  // latency-bound (a 32x33 output tile is a few hundred FMAs per step), so one exposed memory latency per 128 rows instead of
  // one per 16 rows is what matters.
  constexpr int WG_NS = 8, PER = GK * GT / 256;
  for (int kc = kbeg; kc < kend; kc += GK * WG_NS) {
    float ra[WG_NS][PER], rb[WG_NS][PER];
#pragma unroll
    for (int st = 0; st < WG_NS; ++st)
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int kk = e / GT, r = e - kk * GT;
        const int k = kc + st * GK + kk, m = m0 + r, n = n0 + r;
        const bool kin = k < kend;
        ra[st][q] = (kin && m < O) ? dy[(size_t)k * ldy + m] : 0.f;
        rb[st][q] = !kin ? 0.f : (n < I ? x[(size_t)k * ldx + n] : (n == I ? 1.f : 0.f));
      }
#pragma unroll
    for (int st = 0; st < WG_NS; ++st) {
      if (kc + st * GK >= kend) break;            // block-uniform
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int kk = e / GT, r = e - kk * GT;
        As[kk][r] = ra[st][q];
        Bs[kk][r] = rb[st][q];
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < GK; ++kk) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
      }
      __syncthreads();
    }
  }
  const int NP = I + 1;
  if (S > 1) {
    float* mine = partial + (size_t)slab * O * NP;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
        if (m < O && n < NP) __hip_atomic_store(mine + (size_t)m * NP + n, acc[i][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    // The partials travel between blocks (other XCDs, other L2s) as agent-scope relaxed atomics: stores that write through, loads
    // that do not hit a stale line.  No agent-scope fence: a release / acquire pair here is a whole-L2 write-back and invalidate PER
    // BLOCK, and a launch has ~1000 blocks taking tickets — measured, that serialised to 43-65 us per launch for ~10 us of work.
    // Order: every thread's stores are complete (vmcnt 0) before the barrier, the ticket is taken behind the barrier.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = t == S - 1;
      if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // the last block of the tile adds the S slabs in slab order; the tile's valid outputs are spread over all 256 threads
    // (a 32x33 layer keeps only 72 threads busy in the 4x4 register blocking), eight slabs in flight per output
    const int th = min(GT, O - m0), tw = min(GT, NP - n0);
    for (int e = threadIdx.x; e < th * tw; e += 256) {
      const int m = m0 + e / tw, n = n0 + e % tw;
      const float* src = partial + (size_t)m * NP + n;
      float sum = 0.f;
      int z = 0;
      for (; z + 16 <= S; z += 16) {       // 16 slabs in flight per output (the tail of the launch is this one block's latency chain)
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __hip_atomic_load(src + (size_t)(z + j) * O * NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += v[j];
      }
      for (; z < S; ++z) sum += __hip_atomic_load(src + (size_t)z * O * NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n < I) {
        float* p = dW + (size_t)m * I + n;
        *p = accW ? *p + sum : sum;
      } else if (db) {
        db[m] = accB ? db[m] + sum : sum;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m >= O) continue;
      if (n < I) {
        float* p = dW + (size_t)m * I + n;
        *p = accW ? *p + acc[i][j] : acc[i][j];
      } else if (n == I && db) {
        db[m] = accB ? db[m] + acc[i][j] : acc[i][j];
      }
    }
}

__global__ void __launch_bounds__(256) linear_wgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                           int B, int O, int I, float* __restrict__ dW, float* __restrict__ db,
                                                           int accW, int accB, int S, int chunk, float* __restrict__ partial,
                                                           int* __restrict__ tickets) {
  linear_wgrad_body(dy, ldy, x, ldx, B, O, I, dW, db, accW, accB, S, chunk, partial, tickets + blockIdx.y * gridDim.x + blockIdx.x, blockIdx.x,
                    blockIdx.y, blockIdx.z);
}

// Many small layers of one backward pass in ONE launch (a step of the tabular generator has 29 of them, 35 tiles), on the matrix
// cores: dW[O][I] = dy^T x is a [O x rows] x [rows x I] product whose operands are ALREADY in MFMA layout in global memory — for
// v_mfma_f32_32x32x2_f32 lane (li, lh) supplies A[m = li][k = lh] = dy[row + lh][m0 + li] and B[k = lh][n = li] = x[row + lh][n0 + li]:
// two coalesced 128-byte reads per half-wave, no LDS staging.  A WAVE owns (32x32 output tile, slab of rows); the four waves of a
// block add their tiles through LDS in wave order, blocks leave one partial each, and the block that takes the last ticket of a
// tile adds the partials in block order (deterministic, no float atomics; the partials travel as agent-scope relaxed atomics, see
// linear_wgrad_body).  The bias gradient is the column sum of dy: accumulated on the side by the tile_x = 0 waves.
// (The first version ran these layers through the 64x64 VALU tile above: a 32x33 layer keeps 72 of 256 threads busy, and the launch
// was VALU-bound on padding — 31-65 us for 0.3 GFLOP.)
constexpr int WM_MAX_ITEMS = 40, WM_MAX_TILES = 64, WM_T = 32, WM_NP = WM_T + 1;
struct WgradMfmaGroup {
  const float* dy[WM_MAX_ITEMS]; const float* x[WM_MAX_ITEMS]; float* dW[WM_MAX_ITEMS]; float* db[WM_MAX_ITEMS];
  int ldy[WM_MAX_ITEMS], ldx[WM_MAX_ITEMS], O[WM_MAX_ITEMS], I[WM_MAX_ITEMS], poff[WM_MAX_ITEMS];
  unsigned char accW[WM_MAX_ITEMS], accB[WM_MAX_ITEMS];
  unsigned char t_item[WM_MAX_TILES], t_x[WM_MAX_TILES], t_y[WM_MAX_TILES];
};
typedef float wm_acc_t __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) linear_wgrad_mfma_kernel(WgradMfmaGroup g, int B, int Sb, int chunk, float* __restrict__ partial,
                                                                int* __restrict__ tickets) {
  __shared__ float red[4][WM_T * WM_NP];
  __shared__ int s_last;
  PCG_T(0);
  const int tile = blockIdx.y, sb = blockIdx.x;
  const int it = g.t_item[tile], m0 = g.t_y[tile] * WM_T, n0 = g.t_x[tile] * WM_T;
  const int O = g.O[it], I = g.I[it], ldy = g.ldy[it], ldx = g.ldx[it];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int kbeg = (sb * 4 + wave) * chunk, kend = min(B, kbeg + chunk);
  const bool va = m0 + li < O, vb = n0 + li < I;
  const float* pa = g.dy[it] + min(m0 + li, O - 1);          // clamped, not guarded: out-of-tile lanes read a valid element and drop it
  const float* pb = g.x[it] + min(n0 + li, I - 1);
  wm_acc_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  PCG_T(1);
  constexpr int GS = 32;                                     // MFMA steps (of two rows) per group: all 64 loads of a group in flight together
  for (int r = kbeg; r < kend; r += 2 * GS) {
    float av[GS], bv[GS];
#pragma unroll
    for (int st = 0; st < GS; ++st) {
      const int row = r + 2 * st + lh, rc = min(row, kend - 1);
      const float a = pa[(size_t)rc * ldy], b = pb[(size_t)rc * ldx];
      av[st] = (row < kend && va) ? a : 0.f;
      bv[st] = (row < kend && vb) ? b : 0.f;
    }
#pragma unroll
    for (int st = 0; st < GS; ++st) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bv[st], acc, 0, 0, 0);
      bsum += av[st];
    }
  }
  PCG_T(2);
  bsum += __shfl_xor(bsum, 32);                              // rows of both k-halves: lanes lh = 0 hold column m0 + li of db
  // the four waves' tiles -> LDS, added in wave order
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * WM_NP + li] = acc[r];
  if (lh == 0) red[wave][li * WM_NP + WM_T] = bsum;
  __syncthreads();
  PCG_T(3);
  const int NP = I + 1;
  const bool bias_tile = n0 == 0 && g.db[it] != nullptr;
  float* mine = partial + g.poff[it] + (size_t)sb * O * NP;
  for (int e = threadIdx.x; e < WM_T * WM_NP; e += 256) {
    const int m = e / WM_NP, n = e - m * WM_NP;
    const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    const int gm = m0 + m, gn = n < WM_T ? n0 + n : I;       // n == 32: the bias column, stored at column I of the partial rows
    const bool ok = gm < O && (n < WM_T ? gn < I : bias_tile);
    if (!ok) continue;
    if (Sb == 1) {
      if (n < WM_T) { float* q = g.dW[it] + (size_t)gm * I + gn; *q = g.accW[it] ? *q + v : v; }
      else { float* q = g.db[it] + gm; *q = g.accB[it] ? *q + v : v; }
    } else {
      __hip_atomic_store(mine + (size_t)gm * NP + gn, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (Sb == 1) return;                                       // kernel-uniform
  PCG_T(4);
  __builtin_amdgcn_s_waitcnt(0);                             // this thread's partial stores are complete ...
  __syncthreads();                                           // ... every thread's are
  PCG_T(5);
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == Sb - 1;
    if (s_last) __hip_atomic_store(tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  PCG_T(6);
  if (!s_last) return;
  const float* base = partial + g.poff[it];
  for (int e = threadIdx.x; e < WM_T * WM_NP; e += 256) {
    const int m = e / WM_NP, n = e - m * WM_NP;
    const int gm = m0 + m, gn = n < WM_T ? n0 + n : I;
    const bool ok = gm < O && (n < WM_T ? gn < I : bias_tile);
    if (!ok) continue;
    const float* src = base + (size_t)gm * NP + gn;
    float sum = 0.f;
    int z = 0;
    for (; z + 8 <= Sb; z += 8) {                            // eight partials in flight per output, added in block order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __hip_atomic_load(src + (size_t)(z + j) * O * NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; z < Sb; ++z) sum += __hip_atomic_load(src + (size_t)z * O * NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n < WM_T) { float* q = g.dW[it] + (size_t)gm * I + gn; *q = g.accW[it] ? *q + sum : sum; }
    else { float* q = g.db[it] + gm; *q = g.accB[it] ? *q + sum : sum; }
  }
  PCG_T(7);
}

__global__ void __launch_bounds__(256) onehot_kernel(const int64_t* __restrict__ idx, int B, int K, float* __restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * K; i += gridDim.x * 256) {
    const int b = i / K, k = i - b * K;
    out[i] = idx[b] == (int64_t)k ? 1.f : 0.f;
  }
}

__global__ void __launch_bounds__(256) concat_cols_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                                          int rows, float* __restrict__ out) {
  const int C = ca + cb;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * C; i += gridDim.x * 256) {
    const int r = i / C, c = i - r * C;
    out[i] = c < ca ? a[(size_t)r * ca + c] : b[(size_t)r * cb + (c - ca)];
  }
}
__global__ void __launch_bounds__(256) split_cols_kernel(const float* __restrict__ d, int ca, int cb, int rows, float* __restrict__ da,
                                                         float* __restrict__ db) {
  const int C = ca + cb;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * C; i += gridDim.x * 256) {
    const int r = i / C, c = i - r * C;
    if (c < ca) { if (da) da[(size_t)r * ca + c] = d[i]; }
    else if (db) db[(size_t)r * cb + (c - ca)] = d[i];
  }
}

// FiLM: y = g*h + b (generator.py:13-16)
__global__ void __launch_bounds__(256) film_fwd_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                       const float* __restrict__ b, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = fmaf(g[i], h[i], b[i]);
}
__global__ void __launch_bounds__(256) film_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ g,
                                                       const float* __restrict__ h, float* __restrict__ dg, float* __restrict__ dh,
                                                       size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = dy[i];
    dg[i] = d * h[i];
    dh[i] = d * g[i];
  }
}

// Gumbel-softmax heads ([torch] F.gumbel_softmax, hard=False): within segment s of a row,
// y = softmax((logits + noise)/tau).  One thread per (row, segment); segments are <= 30 wide.
__global__ void __launch_bounds__(256) gumbel_softmax_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ noise,
                                                                 const int* __restrict__ seg, int S, int T, int B, float inv_tau,
                                                                 float* __restrict__ y, float* __restrict__ y_hard) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * S; i += gridDim.x * 256) {
    const int b = i / S, s = i - b * S;
    const int c0 = seg[s], c1 = seg[s + 1];
    const float* l = logits + (size_t)b * T;
    const float* g = noise + (size_t)b * T;
    float mx = -INFINITY;
    for (int c = c0; c < c1; ++c) mx = fmaxf(mx, (l[c] + g[c]) * inv_tau);
    float se = 0.f;
    for (int c = c0; c < c1; ++c) se += expf((l[c] + g[c]) * inv_tau - mx);
    const float inv = 1.f / se;
    float best = -1.f;
    int arg = c0;
    for (int c = c0; c < c1; ++c) {
      const float p = expf((l[c] + g[c]) * inv_tau - mx) * inv;
      y[(size_t)b * T + c] = p;
      if (p > best) { best = p; arg = c; }   // first maximum, as y_soft.max(dim)[1]
    }
    if (y_hard) for (int c = c0; c < c1; ++c) y_hard[(size_t)b * T + c] = c == arg ? 1.f : 0.f;
  }
}
// dlogits = y * (dy - sum_seg(dy*y)) / tau
__global__ void __launch_bounds__(256) gumbel_softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const int* __restrict__ seg, int S, int T, int B, float inv_tau,
                                                                 float* __restrict__ dl) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * S; i += gridDim.x * 256) {
    const int b = i / S, s = i - b * S;
    const int c0 = seg[s], c1 = seg[s + 1];
    const size_t o = (size_t)b * T;
    float dot = 0.f;
    for (int c = c0; c < c1; ++c) dot = fmaf(dy[o + c], y[o + c], dot);
    for (int c = c0; c < c1; ++c) dl[o + c] = y[o + c] * (dy[o + c] - dot) * inv_tau;
  }
}

// residual_full (trainer.py:266-279): continuous columns copy cont[:, i]; categorical column f = samples[:, seg s] . norm[seg s] - x[:, f]
__global__ void __launch_bounds__(256) assemble_fwd_kernel(const float* __restrict__ cont, int ncont, const int* __restrict__ cont_idx,
                                                           const float* __restrict__ samples, const int* __restrict__ seg, int S,
                                                           int T, const int* __restrict__ cat_idx, const float* __restrict__ norm,
                                                           const float* __restrict__ x, int D, int B, float* __restrict__ res) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * (ncont + S); i += gridDim.x * 256) {
    const int b = i / (ncont + S), j = i - b * (ncont + S);
    if (j < ncont) {
      res[(size_t)b * D + cont_idx[j]] = cont[(size_t)b * ncont + j];
    } else {
      const int s = j - ncont, f = cat_idx[s];
      float acc = 0.f;
      for (int c = seg[s]; c < seg[s + 1]; ++c) acc = fmaf(samples[(size_t)b * T + c], norm[c], acc);
      res[(size_t)b * D + f] = acc - x[(size_t)b * D + f];
    }
  }
}
__global__ void __launch_bounds__(256) assemble_bwd_kernel(const float* __restrict__ dres, int ncont, const int* __restrict__ cont_idx,
                                                           const int* __restrict__ seg, int S, int T, const int* __restrict__ cat_idx,
                                                           const float* __restrict__ norm, int D, int B, float* __restrict__ dcont,
                                                           float* __restrict__ dsamples) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * (ncont + S); i += gridDim.x * 256) {
    const int b = i / (ncont + S), j = i - b * (ncont + S);
    if (j < ncont) {
      dcont[(size_t)b * ncont + j] = dres[(size_t)b * D + cont_idx[j]];
    } else {
      const int s = j - ncont;
      const float d = dres[(size_t)b * D + cat_idx[s]];
      for (int c = seg[s]; c < seg[s + 1]; ++c) dsamples[(size_t)b * T + c] = d * norm[c];
    }
  }
}

// ---- the residual block of the tabular step (trainer.py:266-287, :305) in two launches -------------------------------------------
// forward: residual_full (assemble), masked_residual = residual_full * mask, x_cf = x + masked_residual, and the two penalties
// mean|residual_full * (1 - mask)|, mean|masked_residual| — what pcg_assemble_residual_fwd + pcg_scale_mask_fwd + pcg_axpby +
// 2 x pcg_abs_mean_fwd compute in seven launches, with the same element arithmetic, the same partition of the sums over threads and
// the same reduction trees (so the same bits).  Contraction is off: every product and sum below was a rounded result in its own
// kernel.  col_src[col] >= 0: continuous column (index into cont); < 0: categorical head -(s + 1).
struct ResCols { int src[32]; };
struct ResFwdArgs {
  const float* cont; int ncont; const float* samples; const int* seg; int nseg, T; const float* norm; const float* x; const float* mask;
  ResCols cols; int D; size_t n; double inv_n; float* res; float* masked; float* x_cf; float* partial; int* ticket; float* pen_out; float* am_out;
};
// SMALL: one block of 1024 threads (n <= 16 K, as pcg_abs_mean_fwd); else nb blocks of 256 (block index bid) + last-block finish
template <bool SMALL>
__device__ __forceinline__ void house_residual_fwd_body(const ResFwdArgs& a, int bid, int nb) {
#pragma clang fp contract(off)
  const float* __restrict__ cont = a.cont; const float* __restrict__ samples = a.samples; const int* __restrict__ seg = a.seg;
  const float* __restrict__ norm = a.norm; const float* __restrict__ x = a.x; const float* __restrict__ mask = a.mask;
  float* __restrict__ res = a.res; float* __restrict__ masked = a.masked; float* __restrict__ x_cf = a.x_cf; float* __restrict__ partial = a.partial;
  int* __restrict__ ticket = a.ticket; float* __restrict__ pen_out = a.pen_out; float* __restrict__ am_out = a.am_out;
  const int ncont = a.ncont, nseg = a.nseg, T = a.T, D = a.D;
  const size_t n = a.n; const double inv_n = a.inv_n;
  const ResCols& cols = a.cols;
  constexpr int NTH = SMALL ? 1024 : 256;
  __shared__ double redd[2][NTH];
  __shared__ float redf[2][SMALL ? 1 : 256];
  __shared__ int s_last;
  __shared__ int s_seg[34];                                // head boundaries: read once, not per element (a dependent load in front of the dot)
  if (threadIdx.x <= (unsigned)nseg) s_seg[threadIdx.x] = seg[threadIdx.x];
  __syncthreads();
  float acc_pen = 0.f, acc_am = 0.f;
  for (size_t i = (size_t)bid * NTH + threadIdx.x; i < n; i += (size_t)nb * NTH) {
    const size_t b = i / (size_t)D;
    const int col = (int)(i - b * (size_t)D), src = cols.src[col];
    float r;
    if (src >= 0) {
      r = cont[b * ncont + src];
    } else {
      const int sg = -src - 1;
      float acc = 0.f;
      int c = s_seg[sg];
      const int ce = s_seg[sg + 1];
      const float xv = x[i];
      for (; c + 8 <= ce; c += 8) {                      // eight (sample, value) pairs requested together, added in column order
        float sv[8], nv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sv[j] = samples[b * T + c + j]; nv[j] = norm[c + j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(sv[j], nv[j], acc);
      }
      for (; c < ce; ++c) acc = fmaf(samples[b * T + c], norm[c], acc);
      r = acc - xv;
    }
    const float mk = mask[i];
    const float msk = r * mk;                          // (1.0 * r) * mask
    res[i] = r; masked[i] = msk;
    x_cf[i] = x[i] + msk;                              // 1.0 * x + 1.0 * masked
    acc_pen += fabsf(r * (1.f - mk));
    acc_am += fabsf(msk);                              // |masked * 1.0|
  }
  if (SMALL) {
    redd[0][threadIdx.x] = (double)acc_pen; redd[1][threadIdx.x] = (double)acc_am;
    __syncthreads();
    for (int k = NTH / 2; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) { redd[0][threadIdx.x] += redd[0][threadIdx.x + k]; redd[1][threadIdx.x] += redd[1][threadIdx.x + k]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) { pen_out[0] = (float)(redd[0][0] * inv_n); am_out[0] = (float)(redd[1][0] * inv_n); }
    return;
  } else {
    redf[0][threadIdx.x] = acc_pen; redf[1][threadIdx.x] = acc_am;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) { redf[0][threadIdx.x] += redf[0][threadIdx.x + k]; redf[1][threadIdx.x] += redf[1][threadIdx.x + k]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store(partial + bid, redf[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(partial + 256 + bid, redf[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = t == nb - 1;
      if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // the finish of abs_mean_finish_kernel (256 partials, one per thread, fp64 tree) for both sums
    redd[0][threadIdx.x] = 0.0 + (double)__hip_atomic_load(partial + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    redd[1][threadIdx.x] = 0.0 + (double)__hip_atomic_load(partial + 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) { redd[0][threadIdx.x] += redd[0][threadIdx.x + k]; redd[1][threadIdx.x] += redd[1][threadIdx.x + k]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) { pen_out[0] = (float)(redd[0][0] * inv_n); am_out[0] = (float)(redd[1][0] * inv_n); }
  }
}
template <bool SMALL>
__global__ void __launch_bounds__(SMALL ? 1024 : 256) house_residual_fwd_kernel(ResFwdArgs a) {
  house_residual_fwd_body<SMALL>(a, blockIdx.x, gridDim.x);
}
// backward of the same block, given the gradient of the G loss with respect to x_cf as two addends (critic, classifier):
//   d_res = lambda_mask * d mean|res (1-mask)| + mask * (w_reg * d mean|masked| + (gx_a + gx_b)),   then assemble's backward —
// pcg_axpby + pcg_weighted_sum_bwd + 2 x pcg_abs_mean_bwd + pcg_axpby + pcg_scale_mask_bwd + autograd's add + pcg_assemble_residual_bwd
// in one launch, one rounded operation per step as there.  One thread per (row, source column) like assemble_bwd.
struct ResBwdArgs {
  const float* res; const float* masked; const float* mask; const float* gx_a; const float* gx_b; float w_pen, w_am; size_t n; int ncont;
  const int* cont_idx; const int* seg; int S, T; const int* cat_idx; const float* norm; int D, B; float* dcont; float* dsamples;
};
// (elementwise: any partition of the (row, source column) pairs over threads gives the same bits) block bid of nb blocks of nth threads
__device__ __forceinline__ void house_residual_bwd_body(const ResBwdArgs& a, int bid, int nb, int nth) {
#pragma clang fp contract(off)
  const float* __restrict__ res = a.res; const float* __restrict__ masked = a.masked; const float* __restrict__ mask = a.mask;
  const float* __restrict__ gx_a = a.gx_a; const float* __restrict__ gx_b = a.gx_b; const int* __restrict__ cont_idx = a.cont_idx;
  const int* __restrict__ seg = a.seg; const int* __restrict__ cat_idx = a.cat_idx; const float* __restrict__ norm = a.norm;
  float* __restrict__ dcont = a.dcont; float* __restrict__ dsamples = a.dsamples;
  const float w_pen = a.w_pen, w_am = a.w_am; const size_t n = a.n; const int ncont = a.ncont, S = a.S, T = a.T, D = a.D, B = a.B;
  const float g_pen = (w_pen * 1.f) * 1.f / (float)n, g_am = (w_am * 1.f) * 1.f / (float)n;   // weighted_sum_bwd (grad_out = 1), then abs_mean_bwd's g
  for (int i = bid * nth + threadIdx.x; i < B * (ncont + S); i += nb * nth) {
    const int b = i / (ncont + S), j = i - b * (ncont + S);
    const int col = j < ncont ? cont_idx[j] : cat_idx[j - ncont];
    const size_t e = (size_t)b * D + col;
    const float mk = mask[e];
    const float w1 = 1.f - mk, v1 = res[e] * w1;
    const float sg1 = v1 > 0.f ? 1.f : (v1 < 0.f ? -1.f : 0.f);
    const float d_pen = 0.f + g_pen * sg1 * w1;
    const float v2 = masked[e] * 1.f;
    const float sg2 = v2 > 0.f ? 1.f : (v2 < 0.f ? -1.f : 0.f);
    const float d_am = 0.f + g_am * sg2 * 1.f;
    const float gx = 1.f * gx_a[e] + 1.f * gx_b[e];
    const float dsum = 1.f * d_am + 1.f * gx;
    const float d_mm = 1.f * (0.f + dsum * mk);
    const float d = d_pen + d_mm;
    if (j < ncont) {
      dcont[(size_t)b * ncont + j] = d;
    } else {
      const int sgm = j - ncont;
      for (int c = seg[sgm]; c < seg[sgm + 1]; ++c) dsamples[(size_t)b * T + c] = d * norm[c];
    }
  }
}

__global__ void __launch_bounds__(256) house_residual_bwd_kernel(ResBwdArgs a) { house_residual_bwd_body(a, blockIdx.x, gridDim.x, 256); }

// mean(x): per-block partials + fixed-order finish; bwd: dx = g * scale / n
constexpr int MEAN_BLOCKS = 64;
__global__ void __launch_bounds__(256) mean_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void mean_finish_kernel(const float* __restrict__ partial, int nparts, double inv_n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)partial[i];
    out[0] = (float)(s * inv_n);
  }
}
// small tensors (a critic output of 4096 rows): the whole mean in ONE block — one launch instead of two, fixed tree order
__global__ void __launch_bounds__(1024) mean_small_kernel(const float* __restrict__ x, size_t n, double inv_n, float* out) {
  __shared__ double red[1024];
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) acc += x[i];
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}
// The logged scalars of the tabular step in ONE launch: the three critic-output means (each with mean_small_kernel's exact reduction
// tree) and the weighted combinations the trainer forms from them (each with weighted_sum_fwd_kernel's fma chain, in its term order):
//   out[0] = D_loss = mean(d_fake) - mean(d_real)                                  (trainer.py:292)
//   out[1] = G_loss = -mean(d_fake_g) + l_cls*g_cls + l_reg*am + l_mask*pen        (:299, :307-312)
//   out[2] = g_adv  = -mean(d_fake_g)       out[3] = g_reg = w_reg_log * am        out[4] = mean(d_fake_g)
struct LossArgs {
  const float* d_real; const float* d_fake; const float* d_fake_g; size_t n; double inv_n; const float* g_cls; const float* am; const float* pen;
  float l_cls, l_reg, l_mask, w_reg_log; float* out;
  const float* rowloss; int n_ce;       // rowloss != nullptr: g_cls = the mean of these cross-entropy row terms, summed as cross_entropy_kernel does (out[5])
  double* acc;                          // nullable: epoch accumulators (see DiagArgs): acc[0] += D_loss, acc[1] += G_loss, acc[6] += 1
};
__device__ __forceinline__ void house_losses_body(const LossArgs& a) {       // one block of 1024 threads
  const float* __restrict__ d_real = a.d_real; const float* __restrict__ d_fake = a.d_fake; const float* __restrict__ d_fake_g = a.d_fake_g;
  const float* __restrict__ g_cls = a.g_cls; const float* __restrict__ am = a.am; const float* __restrict__ pen = a.pen; float* __restrict__ out = a.out;
  const size_t n = a.n; const double inv_n = a.inv_n; const float l_cls = a.l_cls, l_reg = a.l_reg, l_mask = a.l_mask, w_reg_log = a.w_reg_log;
  __shared__ double red[1024];
  __shared__ float means[3];
  const float* vec[3] = {d_real, d_fake, d_fake_g};
#pragma unroll 1
  for (int v = 0; v < 3; ++v) {
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) acc += vec[v][i];
    red[threadIdx.x] = (double)acc;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
      __syncthreads();
    }
    if (threadIdx.x == 0) means[v] = (float)(red[0] * inv_n);
    __syncthreads();
  }
  __shared__ float ce_red[1024];
  float gcls = 0.f;
  if (a.rowloss) {                                     // kernel-uniform: cross_entropy_kernel's partition (1024 threads above 1024 rows, else 256) and tree
    const int nt = a.n_ce > 1024 ? 1024 : 256;
    float acc = 0.f;
    if ((int)threadIdx.x < nt)
      for (int b = threadIdx.x; b < a.n_ce; b += nt) acc += a.rowloss[b];
    ce_red[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = nt >> 1; sft > 0; sft >>= 1) {
      if ((int)threadIdx.x < sft) ce_red[threadIdx.x] += ce_red[threadIdx.x + sft];
      __syncthreads();
    }
    gcls = ce_red[0] / (float)a.n_ce;
  }
  if (threadIdx.x == 0) {
    const float m_real = means[0], m_fake = means[1], m_g = means[2];
    if (!a.rowloss) gcls = g_cls[0];
    else out[5] = gcls;
    out[0] = fmaf(-1.f, m_real, fmaf(1.f, m_fake, 0.f));
    out[1] = fmaf(l_mask, pen[0], fmaf(l_reg, am[0], fmaf(l_cls, gcls, fmaf(-1.f, m_g, 0.f))));
    out[2] = fmaf(-1.f, m_g, 0.f);
    out[3] = fmaf(w_reg_log, am[0], 0.f);
    out[4] = m_g;
    if (a.acc) {                        // the trainer's per-epoch means (trainer.py:349-355) without a host read per iteration
      a.acc[0] += (double)out[0];
      a.acc[1] += (double)out[1];
      a.acc[6] += 1.0;
    }
  }
}

// The four per-iteration diagnostics of the tabular trainer (house_sales_kc_usa/trainer.py:318-343) as ONE block-wide reduction:
//   out[0] pred_gain       = mean_b softmax(logits_cf)[b, t_b] - softmax(logits_orig)[src_b, t_b]          (:320-327)
//   out[1] sparsity        = 1 - mean_{b,f} [ |masked_residual| > eps ]                                    (:330-333)
//   out[2] reg_loss_l2     = mean_b ||masked_residual_b||_2                                                (:335)
//   out[3] class_flip_rate = mean_b [ argmax logits_cf[b] == t_b ]                                         (:337-338)
// logits_orig: the frozen classifier on the ORIGINAL rows — it never changes during GAN training, so the trainer evaluates it once
// for the whole training set and every iteration gathers its batch's rows through src (nullptr: row b).  Fixed partition (thread t
// owns rows t, t+1024, ...) and a fixed fp64 tree: bit-reproducible.  acc (nullable): acc[2..5] += out[0..3].
struct DiagArgs {
  const float* logits_cf; const float* logits_orig; const int64_t* src; const int64_t* target; const float* masked;
  int B, nc, D; float eps; float* out; double* acc;
};
__device__ __forceinline__ void house_diag_body(const DiagArgs& a) {          // one block of 1024 threads
  __shared__ double dred[4][1024];
  double s_gain = 0.0, s_chg = 0.0, s_l2 = 0.0, s_flip = 0.0;
  for (int b = threadIdx.x; b < a.B; b += 1024) {
    const int t = (int)a.target[b];
    const float* lc = a.logits_cf + (size_t)b * a.nc;
    const float* lo = a.logits_orig + (size_t)(a.src ? a.src[b] : (int64_t)b) * a.nc;
    float mc = lc[0], mo = lo[0];
    int arg = 0;
    for (int q = 1; q < a.nc; ++q) {
      if (lc[q] > mc) { mc = lc[q]; arg = q; }      // first maximum, as torch.argmax on the CPU
      mo = fmaxf(mo, lo[q]);
    }
    float zc = 0.f, zo = 0.f;
    for (int q = 0; q < a.nc; ++q) { zc += expf(lc[q] - mc); zo += expf(lo[q] - mo); }
    s_gain += (double)(expf(lc[t] - mc) / zc - expf(lo[t] - mo) / zo);
    s_flip += arg == t ? 1.0 : 0.0;
    const float* m = a.masked + (size_t)b * a.D;
    float sq = 0.f;
    int chg = 0;
    for (int f = 0; f < a.D; ++f) { sq = fmaf(m[f], m[f], sq); chg += fabsf(m[f]) > a.eps; }
    s_l2 += (double)sqrtf(sq);
    s_chg += (double)chg;
  }
  dred[0][threadIdx.x] = s_gain; dred[1][threadIdx.x] = s_chg; dred[2][threadIdx.x] = s_l2; dred[3][threadIdx.x] = s_flip;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k)
      for (int v = 0; v < 4; ++v) dred[v][threadIdx.x] += dred[v][threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double inv_b = 1.0 / (double)a.B;
    a.out[0] = (float)(dred[0][0] * inv_b);
    a.out[1] = (float)(1.0 - dred[1][0] * inv_b / (double)a.D);
    a.out[2] = (float)(dred[2][0] * inv_b);
    a.out[3] = (float)(dred[3][0] * inv_b);
    if (a.acc)
      for (int v = 0; v < 4; ++v) a.acc[2 + v] += (double)a.out[v];
  }
}
__global__ void __launch_bounds__(1024) house_diag_kernel(DiagArgs a) { house_diag_body(a); }

__global__ void __launch_bounds__(1024) house_losses_kernel(LossArgs a) { house_losses_body(a); }
// Riders: two launches of the tabular step that do not depend on each other as ONE launch whose blocks split between the two bodies
// (a graph with parallel branches is launched node by node by the host; one launch is not).  Block 0: the logged scalars (needs the
// last critic forward); blocks 1 ..: the way back from dLoss/dx_cf to the generator's outputs.  Same bodies, same bits.
__global__ void __launch_bounds__(1024) house_residual_bwd_losses_kernel(ResBwdArgs r, LossArgs l) {
  if (blockIdx.x == 0) house_losses_body(l);
  else house_residual_bwd_body(r, blockIdx.x - 1, gridDim.x - 1, 1024);
}
// the same with a second rider block: the trainer's per-iteration diagnostics (they read the classifier's logits, the masked
// residual and the targets — nothing this launch writes)
__global__ void __launch_bounds__(1024) house_residual_bwd_losses_diag_kernel(ResBwdArgs r, LossArgs l, DiagArgs d) {
  if (blockIdx.x == 0) house_losses_body(l);
  else if (blockIdx.x == 1) house_diag_body(d);
  else house_residual_bwd_body(r, blockIdx.x - 2, gridDim.x - 2, 1024);
}
__global__ void __launch_bounds__(256) mean_bwd_kernel(const float* __restrict__ gout, float scale, size_t n, float* __restrict__ dx) {
  const float g = (gout ? gout[0] : 1.f) * scale / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = g;
}

__global__ void __launch_bounds__(256) spectral_norm_fwd_kernel(const float* __restrict__ W, int O, int I, float* __restrict__ u,
                                                                float* __restrict__ v, float eps, int power_iter,
                                                                float* __restrict__ Wbar, float* __restrict__ sigma_out,
                                                                float* __restrict__ u_used, float* __restrict__ v_used) {
  __shared__ SnFwdLds lds;
  float* wb[1] = {Wbar}; float* sg[1] = {sigma_out}; float* uu[1] = {u_used}; float* vu[1] = {v_used};
  spectral_norm_fwd_body(W, O, I, u, v, eps, power_iter, wb, sg, uu, vu, 1, 0, lds);
}
__global__ void __launch_bounds__(256) spectral_norm_bwd_kernel(const float* __restrict__ dWbar, const float* __restrict__ Wbar, int O,
                                                                int I, const float* __restrict__ u, const float* __restrict__ v,
                                                                const float* __restrict__ sigma, float* __restrict__ dW, int accumulate) {
  __shared__ SnBwdLds lds;
  spectral_norm_bwd_body<false>(dWbar, Wbar, O, I, u, v, sigma, dW, accumulate, lds);
}

__global__ void __launch_bounds__(256) spectral_norm_fwd_batched_kernel(SnFwdBatch b, float eps, int power_iter, int n, int reps) {
  __shared__ SnFwdLds lds;
  const int l = blockIdx.x;       // outputs of call r of layer l: entry r * n + l
  spectral_norm_fwd_body(b.W[l], b.O[l], b.I[l], b.u[l], b.v[l], eps, power_iter, b.Wbar + l, b.sigma + l, b.uu + l, b.vu + l, reps, n, lds);
}
// Rider: the critic step's power iterations only need the critic's weights, so they ride with the residual block's forward —
// the first n blocks one matrix each (they are the long pole of the launch: in front, they start with the launch), the 256 blocks
// behind them the residual block (its ticket counts those 256).  Same bodies, same bits.
constexpr int RES_BLOCKS = 256;
__global__ void __launch_bounds__(256) house_residual_fwd_sn_kernel(ResFwdArgs a, SnFwdBatch b, float eps, int power_iter, int n, int reps) {
  if ((int)blockIdx.x >= n) { house_residual_fwd_body<false>(a, blockIdx.x - n, RES_BLOCKS); return; }
  __shared__ SnFwdLds lds;
  const int l = blockIdx.x;
  spectral_norm_fwd_body(b.W[l], b.O[l], b.I[l], b.u[l], b.v[l], eps, power_iter, b.Wbar + l, b.sigma + l, b.uu + l, b.vu + l, reps, n, lds);
}
__global__ void __launch_bounds__(256) spectral_norm_bwd_batched_kernel(SnBwdBatch b, SnBwdExtra x, int n, int passes) {
  __shared__ SnBwdLds lds;
  spectral_norm_bwd_seq_body<false>(b, x, n, passes, blockIdx.x, lds);
}

}  // namespace
}  // namespace pcg

using namespace pcg;

namespace {
// C[m][n] = act(sum_k A[m][k] * W[n][k] + bias[n]) for 1..4 output columns and a long K (the critic's Linear(1024 -> 1) head,
// mnist_wgan_conditional.py:101): one wave per row, lanes stride over k with 16-byte loads, butterfly sum.  The 64x64-tile kernel
// above runs such a shape as M/64 blocks of K/16 dependent load -> LDS -> barrier steps (140 us for 768 x 1024 -> 1; this: ~5 us).
template <int NN>
__global__ void __launch_bounds__(256) rowdot_kernel(int M, int K, const float* __restrict__ A, int lda, const float* __restrict__ W, int ldb,
                                                     float* __restrict__ C, int ldc, const float* __restrict__ bias, int accumulate,
                                                     int act_on, float neg) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* a = A + (size_t)row * lda;
  float acc[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) acc[n] = 0.f;
  const bool vec = ((((uintptr_t)a | (uintptr_t)W) & 15) == 0) && (lda % 4 == 0) && (ldb % 4 == 0);
  int k = 0;
  if (vec) {
    for (k = lane * 4; k + 3 < K; k += 256) {
      const float4 av = *reinterpret_cast<const float4*>(a + k);
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        const float4 wv = *reinterpret_cast<const float4*>(W + (size_t)n * ldb + k);
        acc[n] = fmaf(av.x, wv.x, acc[n]); acc[n] = fmaf(av.y, wv.y, acc[n]); acc[n] = fmaf(av.z, wv.z, acc[n]); acc[n] = fmaf(av.w, wv.w, acc[n]);
      }
    }
    k = K & ~3;                         // scalar tail below
    k += lane;
  } else {
    k = lane;
  }
  for (; k < K; k += 64) {
    const float av = a[k];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = fmaf(av, W[(size_t)n * ldb + k], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    float v = acc[n];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) {
      v += bias ? bias[n] : 0.f;
      float* c = C + (size_t)row * ldc + n;
      if (accumulate) v += *c;
      if (act_on) v = act_neg_scale(v, neg);
      *c = v;
    }
  }
}

bool launch_rowdot(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, int accumulate, int act_on, float neg, hipStream_t s) {
  if (transA || !transB || N > 4 || K < 256) return false;
  const dim3 grid((M + 3) / 4), block(256);
  switch (N) {
    case 1: hipLaunchKernelGGL(rowdot_kernel<1>, grid, block, 0, s, M, K, A, lda, B, ldb, C, ldc, bias, accumulate, act_on, neg); break;
    case 2: hipLaunchKernelGGL(rowdot_kernel<2>, grid, block, 0, s, M, K, A, lda, B, ldb, C, ldc, bias, accumulate, act_on, neg); break;
    case 3: hipLaunchKernelGGL(rowdot_kernel<3>, grid, block, 0, s, M, K, A, lda, B, ldb, C, ldc, bias, accumulate, act_on, neg); break;
    default: hipLaunchKernelGGL(rowdot_kernel<4>, grid, block, 0, s, M, K, A, lda, B, ldb, C, ldc, bias, accumulate, act_on, neg); break;
  }
  return true;
}
}  // namespace

extern "C" int pcg_gemm(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                        int32_t ldb, float* C, int32_t ldc, const float* bias, int accumulate, pcg_stream_t stream) {
  PCG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N, "pcg_gemm: bad arguments");
  if (launch_rowdot(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, 0, 1.f, (hipStream_t)stream)) return launch_status("rowdot_kernel");
  hipLaunchKernelGGL(gemm_kernel, dim3((N + GT - 1) / GT, (M + GT - 1) / GT), dim3(256), 0, (hipStream_t)stream, transA, transB, M, N, K,
                     A, lda, B, ldb, C, ldc, bias, accumulate, 0, 1.f);
  return launch_status("gemm_kernel");
}

extern "C" int pcg_gemm_act(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                            int32_t ldb, float* C, int32_t ldc, const float* bias, int accumulate, int act, float slope, pcg_stream_t stream) {
  PCG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N, "pcg_gemm_act: bad arguments");
  PCG_REQUIRE(act_is_cheap(act), "pcg_gemm_act: only ReLU / LeakyReLU are fused (activation %d)", act);
  if (launch_rowdot(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, act != PCG_ACT_NONE, act_neg_of(act, slope), (hipStream_t)stream))
    return launch_status("rowdot_kernel");
  if (launch_skinny(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, act != PCG_ACT_NONE, act_neg_of(act, slope), (hipStream_t)stream))
    return launch_status("gemm_skinny_kernel");
  hipLaunchKernelGGL(gemm_kernel, dim3((N + GT - 1) / GT, (M + GT - 1) / GT), dim3(256), 0, (hipStream_t)stream, transA, transB, M, N, K,
                     A, lda, B, ldb, C, ldc, bias, accumulate, act != PCG_ACT_NONE, act_neg_of(act, slope));
  return launch_status("gemm_kernel");
}

extern "C" int pcg_spectral_norm_fwd_batched_reps(int32_t n, int32_t reps, const float* const* w_orig, const int32_t* out_features,
                                                  const int32_t* in_features, float* const* u, float* const* v, float eps, int power_iteration,
                                                  float* const* w_bar, float* const* sigma, float* const* u_used, float* const* v_used,
                                                  pcg_stream_t stream) {
  SnFwdBatch b{};
  if (int e = fill_sn_fwd_batch(b, n, reps, w_orig, out_features, in_features, u, v, power_iteration, w_bar, sigma, u_used, v_used)) return e;
  hipLaunchKernelGGL(spectral_norm_fwd_batched_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, b, eps, power_iteration, n, reps);
  return launch_status("spectral_norm_fwd_batched_kernel");
}

extern "C" int pcg_spectral_norm_fwd_batched(int32_t n, const float* const* w_orig, const int32_t* out_features, const int32_t* in_features,
                                             float* const* u, float* const* v, float eps, int power_iteration, float* const* w_bar,
                                             float* const* sigma, float* const* u_used, float* const* v_used, pcg_stream_t stream) {
  return pcg_spectral_norm_fwd_batched_reps(n, 1, w_orig, out_features, in_features, u, v, eps, power_iteration, w_bar, sigma, u_used, v_used, stream);
}

extern "C" int pcg_spectral_norm_bwd_batched_seq(int32_t n, int32_t passes, const float* const* dw_bar, const float* const* w_bar,
                                                 const int32_t* out_features, const int32_t* in_features, const float* const* u,
                                                 const float* const* v, const float* const* sigma, float* const* dw_orig, const int32_t* accumulate,
                                                 float* const* db_dst, const float* const* db_src, pcg_stream_t stream) {
  SnBwdBatch b{};
  SnBwdExtra x{};
  if (int e = fill_sn_bwd_batch(b, x, n, passes, dw_bar, w_bar, out_features, in_features, u, v, sigma, dw_orig, accumulate, db_dst, db_src)) return e;
  hipLaunchKernelGGL(spectral_norm_bwd_batched_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, b, x, n, passes);
  return launch_status("spectral_norm_bwd_batched_kernel");
}

extern "C" int pcg_spectral_norm_bwd_batched(int32_t n, const float* const* dw_bar, const float* const* w_bar, const int32_t* out_features,
                                             const int32_t* in_features, const float* const* u, const float* const* v,
                                             const float* const* sigma, float* const* dw_orig, const int32_t* accumulate, pcg_stream_t stream) {
  return pcg_spectral_norm_bwd_batched_seq(n, 1, dw_bar, w_bar, out_features, in_features, u, v, sigma, dw_orig, accumulate, nullptr, nullptr, stream);
}

namespace {
struct WgradPlan { int S, chunk, tiles; };
WgradPlan plan_linear_wgrad(int B, int O, int I) {
  const int tiles = ((I + 1 + GT - 1) / GT) * ((O + GT - 1) / GT);
  int S = (256 + tiles - 1) / tiles;                 // aim at ~256 blocks (one per CU) ...
  const int maxS = (B + 8 * GK - 1) / (8 * GK);      // ... of at least 128 rows (one preloaded group of k-steps) each
  if (S > maxS) S = maxS;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  int chunk = (B + S - 1) / S;
  chunk = (chunk + GK - 1) / GK * GK;
  S = (B + chunk - 1) / chunk;
  return {S, chunk, tiles};
}
}  // namespace

extern "C" size_t pcg_linear_wgrad_workspace_bytes(int32_t B, int32_t O, int32_t I) {
  if (B <= 0 || O <= 0 || I <= 0) return 0;
  const WgradPlan p = plan_linear_wgrad(B, O, I);
  return p.S > 1 ? (size_t)p.S * O * (I + 1) * sizeof(float) : 0;
}
extern "C" int32_t pcg_linear_wgrad_ticket_count(void) { return 4096; }

extern "C" int pcg_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int32_t B, int32_t O, int32_t I, float* dW,
                                float* db, int accumulate_w, int accumulate_b, void* workspace, size_t workspace_bytes, int32_t* tickets,
                                pcg_stream_t stream) {
  PCG_REQUIRE(dy && x && dW && B > 0 && O > 0 && I > 0 && ldy >= O && ldx >= I, "pcg_linear_wgrad: bad arguments");
  const WgradPlan p = plan_linear_wgrad(B, O, I);
  PCG_REQUIRE(p.tiles <= 4096, "pcg_linear_wgrad: layer %dx%d needs %d tiles (limit 4096)", O, I, p.tiles);
  if (p.S > 1) {
    PCG_REQUIRE(tickets, "pcg_linear_wgrad: tickets buffer required");
    if (!workspace || workspace_bytes < pcg_linear_wgrad_workspace_bytes(B, O, I)) { set_error("pcg_linear_wgrad: workspace too small"); return PCG_ERR_WORKSPACE; }
  }
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((I + 1 + GT - 1) / GT, (O + GT - 1) / GT, p.S), dim3(256), 0, (hipStream_t)stream, dy, ldy, x,
                     ldx, B, O, I, dW, db, accumulate_w, accumulate_b, p.S, p.chunk, (float*)workspace, tickets);
  return launch_status("linear_wgrad_kernel");
}

// grouped plan: rows per wave ~64 (32 MFMA steps, all 64 loads requested before the first MFMA: what a wave costs is the latency
// of its load groups, so more, shorter waves win until the last block's sum over the partials grows — step time with 128 rows per
// wave in four groups of 32 loads 0.380 ms, 64 rows in two groups 0.367, 32 rows 0.374, 64 rows in ONE group 0.361), slabs in
// blocks of four waves, at most 16 blocks per tile
namespace {
struct WgradMfmaPlan { int Sb, chunk; };
WgradMfmaPlan plan_wgrad_mfma(int B) {
  int S = (B + 63) / 64;
  S = (S + 3) / 4 * 4;
  if (S > 64) S = 64;
  int chunk = (B + S - 1) / S;
  chunk = (chunk + 31) / 32 * 32;
  return {S / 4, chunk};
}
}  // namespace

extern "C" int32_t pcg_linear_wgrad_grouped_slabs(int32_t B) { return B > 0 ? plan_wgrad_mfma(B).Sb : 0; }

extern "C" size_t pcg_linear_wgrad_grouped_workspace_bytes(int32_t B, const pcg_wgrad_item* items, int32_t n_items) {
  if (B <= 0 || n_items <= 0 || !items) return 0;
  const WgradMfmaPlan p = plan_wgrad_mfma(B);
  size_t total = 0;
  for (int i = 0; i < n_items; ++i) total += (size_t)p.Sb * items[i].O * (items[i].I + 1);
  return total * sizeof(float) + 16;
}

extern "C" int pcg_linear_wgrad_grouped(const pcg_wgrad_item* items, int32_t n_items, int32_t B, void* workspace, size_t workspace_bytes,
                                        int32_t* tickets, pcg_stream_t stream) {
  PCG_REQUIRE(items && n_items > 0 && n_items <= WM_MAX_ITEMS && B > 0 && tickets, "pcg_linear_wgrad_grouped: bad arguments (at most %d layers)", WM_MAX_ITEMS);
  const WgradMfmaPlan p = plan_wgrad_mfma(B);
  if (!workspace || workspace_bytes < pcg_linear_wgrad_grouped_workspace_bytes(B, items, n_items)) {
    set_error("pcg_linear_wgrad_grouped: workspace too small"); return PCG_ERR_WORKSPACE;
  }
  WgradMfmaGroup g{};
  size_t off = 0;
  int nt = 0;
  for (int i = 0; i < n_items; ++i) {
    const pcg_wgrad_item& it = items[i];
    PCG_REQUIRE(it.dy && it.x && it.dW && it.O > 0 && it.I > 0 && it.ldy >= it.O && it.ldx >= it.I, "pcg_linear_wgrad_grouped: layer %d: bad arguments", i);
    g.dy[i] = it.dy; g.x[i] = it.x; g.dW[i] = it.dW; g.db[i] = it.db; g.ldy[i] = it.ldy; g.ldx[i] = it.ldx; g.O[i] = it.O; g.I[i] = it.I;
    g.accW[i] = it.accumulate_w != 0; g.accB[i] = it.accumulate_b != 0;
    PCG_REQUIRE(off < (1u << 30), "pcg_linear_wgrad_grouped: workspace offset overflow");
    g.poff[i] = (int)off;
    off += (size_t)p.Sb * it.O * (it.I + 1);
    for (int ty = 0; ty < (it.O + WM_T - 1) / WM_T; ++ty)
      for (int tx = 0; tx < (it.I + WM_T - 1) / WM_T; ++tx) {
        PCG_REQUIRE(nt < WM_MAX_TILES, "pcg_linear_wgrad_grouped: more than %d 32x32 output tiles in one call", WM_MAX_TILES);
        g.t_item[nt] = (unsigned char)i; g.t_x[nt] = (unsigned char)tx; g.t_y[nt] = (unsigned char)ty; ++nt;
      }
  }
  hipLaunchKernelGGL(linear_wgrad_mfma_kernel, dim3(p.Sb, nt), dim3(256), 0, (hipStream_t)stream, g, B, p.Sb, p.chunk, (float*)workspace, tickets);
  return launch_status("linear_wgrad_mfma_kernel");
}

extern "C" int pcg_onehot(const int64_t* idx, int32_t B, int32_t K, float* out, pcg_stream_t stream) {
  PCG_REQUIRE(idx && out && B > 0 && K > 0, "pcg_onehot: bad arguments");
  hipLaunchKernelGGL(onehot_kernel, dim3(ew_blocks((size_t)B * K)), dim3(256), 0, (hipStream_t)stream, idx, B, K, out);
  return launch_status("onehot_kernel");
}

extern "C" int pcg_concat_cols(const float* a, int32_t ca, const float* b, int32_t cb, int32_t rows, float* out, pcg_stream_t stream) {
  PCG_REQUIRE(a && b && out && ca > 0 && cb > 0 && rows > 0, "pcg_concat_cols: bad arguments");
  hipLaunchKernelGGL(concat_cols_kernel, dim3(ew_blocks((size_t)rows * (ca + cb))), dim3(256), 0, (hipStream_t)stream, a, ca, b, cb,
                     rows, out);
  return launch_status("concat_cols_kernel");
}

extern "C" int pcg_split_cols(const float* d, int32_t ca, int32_t cb, int32_t rows, float* da, float* db, pcg_stream_t stream) {
  PCG_REQUIRE(d && (da || db) && ca > 0 && cb > 0 && rows > 0, "pcg_split_cols: bad arguments");
  hipLaunchKernelGGL(split_cols_kernel, dim3(ew_blocks((size_t)rows * (ca + cb))), dim3(256), 0, (hipStream_t)stream, d, ca, cb, rows,
                     da, db);
  return launch_status("split_cols_kernel");
}

extern "C" int pcg_film_fwd(const float* g, const float* h, const float* b, float* y, int64_t n, pcg_stream_t stream) {
  PCG_REQUIRE(g && h && b && y && n > 0, "pcg_film_fwd: bad arguments");
  hipLaunchKernelGGL(film_fwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, g, h, b, y, (size_t)n);
  return launch_status("film_fwd_kernel");
}

extern "C" int pcg_film_bwd(const float* dy, const float* g, const float* h, float* dg, float* dh, int64_t n, pcg_stream_t stream) {
  PCG_REQUIRE(dy && g && h && dg && dh && n > 0, "pcg_film_bwd: bad arguments");
  hipLaunchKernelGGL(film_bwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, dy, g, h, dg, dh, (size_t)n);
  return launch_status("film_bwd_kernel");
}

extern "C" int pcg_gumbel_softmax_fwd(const float* logits, const float* noise, const int32_t* seg_offsets, int32_t S, int32_t T,
                                      int32_t B, float tau, float* y, float* y_hard, pcg_stream_t stream) {
  PCG_REQUIRE(logits && noise && seg_offsets && y && S > 0 && T > 0 && B > 0 && tau > 0.f, "pcg_gumbel_softmax_fwd: bad arguments");
  hipLaunchKernelGGL(gumbel_softmax_fwd_kernel, dim3(ew_blocks((size_t)B * S)), dim3(256), 0, (hipStream_t)stream, logits, noise,
                     seg_offsets, S, T, B, 1.f / tau, y, y_hard);
  return launch_status("gumbel_softmax_fwd_kernel");
}

extern "C" int pcg_gumbel_softmax_bwd(const float* dy, const float* y, const int32_t* seg_offsets, int32_t S, int32_t T, int32_t B,
                                      float tau, float* dlogits, pcg_stream_t stream) {
  PCG_REQUIRE(dy && y && seg_offsets && dlogits && S > 0 && T > 0 && B > 0 && tau > 0.f, "pcg_gumbel_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(gumbel_softmax_bwd_kernel, dim3(ew_blocks((size_t)B * S)), dim3(256), 0, (hipStream_t)stream, dy, y, seg_offsets, S,
                     T, B, 1.f / tau, dlogits);
  return launch_status("gumbel_softmax_bwd_kernel");
}

extern "C" int pcg_assemble_residual_fwd(const float* cont, int32_t ncont, const int32_t* cont_idx, const float* samples,
                                         const int32_t* seg_offsets, int32_t S, int32_t T, const int32_t* cat_idx, const float* norm_vals,
                                         const float* x, int32_t D, int32_t B, float* residual, pcg_stream_t stream) {
  PCG_REQUIRE(cont && cont_idx && samples && seg_offsets && cat_idx && norm_vals && x && residual && B > 0 && ncont + S == D,
              "pcg_assemble_residual_fwd: bad arguments (every column must be continuous or categorical)");
  hipLaunchKernelGGL(assemble_fwd_kernel, dim3(ew_blocks((size_t)B * D)), dim3(256), 0, (hipStream_t)stream, cont, ncont, cont_idx,
                     samples, seg_offsets, S, T, cat_idx, norm_vals, x, D, B, residual);
  return launch_status("assemble_fwd_kernel");
}

extern "C" int pcg_assemble_residual_bwd(const float* dres, int32_t ncont, const int32_t* cont_idx, const int32_t* seg_offsets, int32_t S,
                                         int32_t T, const int32_t* cat_idx, const float* norm_vals, int32_t D, int32_t B, float* dcont,
                                         float* dsamples, pcg_stream_t stream) {
  PCG_REQUIRE(dres && cont_idx && seg_offsets && cat_idx && norm_vals && dcont && dsamples && B > 0 && ncont + S == D,
              "pcg_assemble_residual_bwd: bad arguments");
  hipLaunchKernelGGL(assemble_bwd_kernel, dim3(ew_blocks((size_t)B * D)), dim3(256), 0, (hipStream_t)stream, dres, ncont, cont_idx,
                     seg_offsets, S, T, cat_idx, norm_vals, D, B, dcont, dsamples);
  return launch_status("assemble_bwd_kernel");
}

extern "C" size_t pcg_mean_workspace_bytes(void) { return MEAN_BLOCKS * sizeof(float); }

extern "C" int pcg_mean_fwd(const float* x, int64_t n, float* out, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  PCG_REQUIRE(x && out && n > 0, "pcg_mean_fwd: bad arguments");
  if (!workspace || workspace_bytes < pcg_mean_workspace_bytes()) { set_error("pcg_mean_fwd: workspace too small"); return PCG_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  if (n <= 16 * 1024) {
    hipLaunchKernelGGL(mean_small_kernel, dim3(1), dim3(1024), 0, s, x, (size_t)n, 1.0 / (double)n, out);
    return launch_status("mean_small_kernel");
  }
  hipLaunchKernelGGL(mean_partial_kernel, dim3(MEAN_BLOCKS), dim3(256), 0, s, x, (size_t)n, (float*)workspace);
  if (int e = launch_status("mean_partial_kernel")) return e;
  hipLaunchKernelGGL(mean_finish_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, MEAN_BLOCKS, 1.0 / (double)n, out);
  return launch_status("mean_finish_kernel");
}

extern "C" int pcg_house_losses(const float* d_real, const float* d_fake, const float* d_fake_g, int64_t n, const float* g_cls,
                                const float* am, const float* pen, float lambda_cls, float w_reg, float lambda_mask, float w_reg_log,
                                float* out5, pcg_stream_t stream) {
  PCG_REQUIRE(d_real && d_fake && d_fake_g && g_cls && am && pen && out5 && n > 0 && n <= 16 * 1024,
              "pcg_house_losses: bad arguments (critic outputs of at most 16384 rows)");
  const LossArgs la{d_real, d_fake, d_fake_g, (size_t)n, 1.0 / (double)n, g_cls, am, pen, lambda_cls, w_reg, lambda_mask, w_reg_log, out5, nullptr, 0, nullptr};
  hipLaunchKernelGGL(house_losses_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, la);
  return launch_status("house_losses_kernel");
}

extern "C" int pcg_mean_bwd(const float* grad_out_dev, float grad_scale, int64_t n, float* dx, pcg_stream_t stream) {
  PCG_REQUIRE(dx && n > 0, "pcg_mean_bwd: bad arguments");
  hipLaunchKernelGGL(mean_bwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, grad_out_dev, grad_scale, (size_t)n, dx);
  return launch_status("mean_bwd_kernel");
}

extern "C" int pcg_spectral_norm_fwd(const float* w_orig, int32_t out_features, int32_t in_features, float* u, float* v, float eps,
                                     int power_iteration, float* w_bar, float* sigma, float* u_used, float* v_used,
                                     pcg_stream_t stream) {
  PCG_REQUIRE(w_orig && u && v && w_bar && sigma && out_features > 0 && in_features > 0 && out_features <= 256 && in_features <= 256,
              "pcg_spectral_norm_fwd: bad arguments (dimensions must be 1..256)");
  hipLaunchKernelGGL(spectral_norm_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_orig, out_features, in_features, u, v, eps,
                     power_iteration, w_bar, sigma, u_used, v_used);
  return launch_status("spectral_norm_fwd_kernel");
}

extern "C" int pcg_spectral_norm_bwd(const float* dw_bar, const float* w_bar, int32_t out_features, int32_t in_features, const float* u,
                                     const float* v, const float* sigma, float* dw_orig, int accumulate, pcg_stream_t stream) {
  PCG_REQUIRE(dw_bar && w_bar && u && v && sigma && dw_orig && out_features > 0 && in_features > 0 && out_features <= 256 && in_features <= 256,
              "pcg_spectral_norm_bwd: bad arguments");
  hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dw_bar, w_bar, out_features, in_features, u, v,
                     sigma, dw_orig, accumulate);
  return launch_status("spectral_norm_bwd_kernel");
}

namespace {
int fill_res_fwd_args(ResFwdArgs& a, const float* cont, int32_t ncont, const float* samples, const int32_t* seg_dev, int32_t T, const float* norm,
                      const float* x, const float* mask, const int32_t* col_src, int32_t D, int32_t B, float* res, float* masked, float* x_cf,
                      float* partial512, int32_t* ticket, float* pen_out, float* am_out) {
  PCG_REQUIRE(cont && samples && seg_dev && norm && x && mask && col_src && res && masked && x_cf && partial512 && ticket && pen_out && am_out &&
                  B > 0 && D > 0 && D <= 32 && T > 0 && ncont >= 0, "pcg_house_residual_fwd: bad arguments (at most 32 feature columns)");
  int nseg = 0;                                            // number of categorical heads = the largest -(src) among the columns
  for (int c = 0; c < D; ++c) { a.cols.src[c] = col_src[c]; if (col_src[c] < 0 && -col_src[c] > nseg) nseg = -col_src[c]; }
  PCG_REQUIRE(nseg <= 32, "pcg_house_residual_fwd: at most 32 categorical heads");
  a.cont = cont; a.ncont = ncont; a.samples = samples; a.seg = seg_dev; a.nseg = nseg; a.T = T; a.norm = norm; a.x = x; a.mask = mask; a.D = D;
  a.n = (size_t)B * D; a.inv_n = 1.0 / (double)a.n; a.res = res; a.masked = masked; a.x_cf = x_cf; a.partial = partial512; a.ticket = ticket;
  a.pen_out = pen_out; a.am_out = am_out;
  return PCG_OK;
}
int fill_res_bwd_args(ResBwdArgs& a, const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b, float w_pen,
                      float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev, int32_t S, int32_t T,
                      const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B, float* dcont, float* dsamples) {
  PCG_REQUIRE(res && masked && mask && gx_a && gx_b && cont_idx_dev && seg_dev && cat_idx_dev && norm && dcont && dsamples && B > 0 && D > 0 &&
                  S >= 0 && T > 0 && ncont >= 0 && ncont + S > 0, "pcg_house_residual_bwd: bad arguments");
  a = ResBwdArgs{res, masked, mask, gx_a, gx_b, w_pen, w_am, (size_t)B * D, ncont, cont_idx_dev, seg_dev, S, T, cat_idx_dev, norm, D, B, dcont, dsamples};
  return PCG_OK;
}
}  // namespace

extern "C" int pcg_house_residual_fwd(const float* cont, int32_t ncont, const float* samples, const int32_t* seg_dev, int32_t T, const float* norm,
                                      const float* x, const float* mask, const int32_t* col_src, int32_t D, int32_t B, float* res, float* masked,
                                      float* x_cf, float* partial512, int32_t* ticket, float* pen_out, float* am_out, pcg_stream_t stream) {
  ResFwdArgs a{};
  if (int e = fill_res_fwd_args(a, cont, ncont, samples, seg_dev, T, norm, x, mask, col_src, D, B, res, masked, x_cf, partial512, ticket, pen_out, am_out)) return e;
  hipStream_t s = (hipStream_t)stream;
  if (a.n <= 16 * 1024) hipLaunchKernelGGL(house_residual_fwd_kernel<true>, dim3(1), dim3(1024), 0, s, a);
  else hipLaunchKernelGGL(house_residual_fwd_kernel<false>, dim3(RES_BLOCKS), dim3(256), 0, s, a);
  return launch_status("house_residual_fwd_kernel");
}

// pcg_house_residual_fwd + pcg_spectral_norm_fwd_batched_reps (training mode) as ONE launch: the two do not depend on each other
extern "C" int pcg_house_residual_fwd_sn(const float* cont, int32_t ncont, const float* samples, const int32_t* seg_dev, int32_t T, const float* norm,
                                         const float* x, const float* mask, const int32_t* col_src, int32_t D, int32_t B, float* res, float* masked,
                                         float* x_cf, float* partial512, int32_t* ticket, float* pen_out, float* am_out,
                                         int32_t n_layers, int32_t reps, const float* const* w_orig, const int32_t* out_features,
                                         const int32_t* in_features, float* const* u, float* const* v, float eps, float* const* w_bar,
                                         float* const* sigma, float* const* u_used, float* const* v_used, pcg_stream_t stream) {
  ResFwdArgs a{};
  if (int e = fill_res_fwd_args(a, cont, ncont, samples, seg_dev, T, norm, x, mask, col_src, D, B, res, masked, x_cf, partial512, ticket, pen_out, am_out)) return e;
  SnFwdBatch b{};
  if (int e = fill_sn_fwd_batch(b, n_layers, reps, w_orig, out_features, in_features, u, v, 1, w_bar, sigma, u_used, v_used)) return e;
  hipStream_t s = (hipStream_t)stream;
  if (a.n <= 16 * 1024) {                                  // the one-block form of the residual kernel: two launches
    hipLaunchKernelGGL(house_residual_fwd_kernel<true>, dim3(1), dim3(1024), 0, s, a);
    if (int e = launch_status("house_residual_fwd_kernel")) return e;
    hipLaunchKernelGGL(spectral_norm_fwd_batched_kernel, dim3(n_layers), dim3(256), 0, s, b, eps, 1, n_layers, reps);
    return launch_status("spectral_norm_fwd_batched_kernel");
  }
  hipLaunchKernelGGL(house_residual_fwd_sn_kernel, dim3(RES_BLOCKS + n_layers), dim3(256), 0, s, a, b, eps, 1, n_layers, reps);
  return launch_status("house_residual_fwd_sn_kernel");
}

extern "C" int pcg_house_residual_bwd(const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b, float w_pen,
                                      float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev, int32_t S, int32_t T,
                                      const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B, float* dcont, float* dsamples,
                                      pcg_stream_t stream) {
  ResBwdArgs a{};
  if (int e = fill_res_bwd_args(a, res, masked, mask, gx_a, gx_b, w_pen, w_am, ncont, cont_idx_dev, seg_dev, S, T, cat_idx_dev, norm, D, B, dcont, dsamples)) return e;
  const size_t work = (size_t)B * (ncont + S);
  hipLaunchKernelGGL(house_residual_bwd_kernel, dim3((unsigned)std::min<size_t>((work + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status("house_residual_bwd_kernel");
}

// pcg_house_residual_bwd + pcg_house_losses as ONE launch (the logged scalars do not feed the backward)
extern "C" int pcg_house_residual_bwd_losses(const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b,
                                             float w_pen, float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev, int32_t S,
                                             int32_t T, const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B, float* dcont,
                                             float* dsamples, const float* d_real, const float* d_fake, const float* d_fake_g, int32_t n,
                                             const float* g_cls, const float* am, const float* pen, float lambda_cls, float w_reg, float lambda_mask,
                                             float w_reg_log, const float* ce_row_loss, int32_t n_ce, float* out6, pcg_stream_t stream) {
  ResBwdArgs a{};
  if (int e = fill_res_bwd_args(a, res, masked, mask, gx_a, gx_b, w_pen, w_am, ncont, cont_idx_dev, seg_dev, S, T, cat_idx_dev, norm, D, B, dcont, dsamples)) return e;
  PCG_REQUIRE(d_real && d_fake && d_fake_g && (g_cls || ce_row_loss) && am && pen && out6 && n > 0 && n <= 16 * 1024 && (!ce_row_loss || n_ce > 0),
              "pcg_house_residual_bwd_losses: bad arguments (critic outputs of at most 16384 rows)");
  const LossArgs la{d_real, d_fake, d_fake_g, (size_t)n, 1.0 / (double)n, g_cls, am, pen, lambda_cls, w_reg, lambda_mask, w_reg_log, out6, ce_row_loss, n_ce, nullptr};
  const size_t work = (size_t)B * (ncont + S);
  hipLaunchKernelGGL(house_residual_bwd_losses_kernel, dim3(1 + (unsigned)std::min<size_t>((work + 1023) / 1024, 4096)), dim3(1024), 0,
                     (hipStream_t)stream, a, la);
  return launch_status("house_residual_bwd_losses_kernel");
}

namespace {
int fill_diag_args(pcg::DiagArgs& d, const float* logits_cf, const float* logits_orig, const int64_t* src_rows, const int64_t* target_y,
                   const float* masked, int32_t B, int32_t nc, int32_t D, float eps, float* out4, double* acc) {
  PCG_REQUIRE(logits_cf && logits_orig && target_y && masked && out4 && B > 0 && nc > 1 && nc <= 64 && D > 0,
              "pcg_house_diag: bad arguments (B %d, classes %d, features %d)", B, nc, D);
  d = pcg::DiagArgs{logits_cf, logits_orig, src_rows, target_y, masked, B, nc, D, eps, out4, acc};
  return PCG_OK;
}
}  // namespace

extern "C" int pcg_house_diag(const float* logits_cf, const float* logits_orig, const int64_t* src_rows, const int64_t* target_y,
                              const float* masked, int32_t B, int32_t nc, int32_t D, float eps, float* out4, double* acc,
                              pcg_stream_t stream) {
  DiagArgs d{};
  if (int e = fill_diag_args(d, logits_cf, logits_orig, src_rows, target_y, masked, B, nc, D, eps, out4, acc)) return e;
  hipLaunchKernelGGL(house_diag_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d);
  return launch_status("house_diag_kernel");
}

// pcg_house_residual_bwd_losses with the diagnostics block riding along and the epoch accumulators (acc[8] doubles:
// sum D_loss, sum G_loss, sum pred_gain, sum sparsity, sum l2, sum class_flip_rate, iterations, unused)
extern "C" int pcg_house_residual_bwd_losses_diag(const float* res, const float* masked, const float* mask, const float* gx_a, const float* gx_b,
                                                  float w_pen, float w_am, int32_t ncont, const int32_t* cont_idx_dev, const int32_t* seg_dev,
                                                  int32_t S, int32_t T, const int32_t* cat_idx_dev, const float* norm, int32_t D, int32_t B,
                                                  float* dcont, float* dsamples, const float* d_real, const float* d_fake, const float* d_fake_g,
                                                  int32_t n, const float* g_cls, const float* am, const float* pen, float lambda_cls, float w_reg,
                                                  float lambda_mask, float w_reg_log, const float* ce_row_loss, int32_t n_ce, float* out6,
                                                  const float* logits_cf, const float* logits_orig, const int64_t* src_rows,
                                                  const int64_t* target_y, int32_t nc, float eps, float* diag_out4, double* acc,
                                                  pcg_stream_t stream) {
  ResBwdArgs a{};
  if (int e = fill_res_bwd_args(a, res, masked, mask, gx_a, gx_b, w_pen, w_am, ncont, cont_idx_dev, seg_dev, S, T, cat_idx_dev, norm, D, B, dcont, dsamples)) return e;
  PCG_REQUIRE(d_real && d_fake && d_fake_g && (g_cls || ce_row_loss) && am && pen && out6 && n > 0 && n <= 16 * 1024 && (!ce_row_loss || n_ce > 0),
              "pcg_house_residual_bwd_losses_diag: bad arguments (critic outputs of at most 16384 rows)");
  const LossArgs la{d_real, d_fake, d_fake_g, (size_t)n, 1.0 / (double)n, g_cls, am, pen, lambda_cls, w_reg, lambda_mask, w_reg_log, out6, ce_row_loss, n_ce, acc};
  DiagArgs d{};
  if (int e = fill_diag_args(d, logits_cf, logits_orig, src_rows, target_y, masked, B, nc, D, eps, diag_out4, acc)) return e;
  const size_t work = (size_t)B * (ncont + S);
  hipLaunchKernelGGL(house_residual_bwd_losses_diag_kernel, dim3(2 + (unsigned)std::min<size_t>((work + 1023) / 1024, 4096)), dim3(1024), 0,
                     (hipStream_t)stream, a, la, d);
  return launch_status("house_residual_bwd_losses_diag_kernel");
}
