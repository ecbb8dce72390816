// batchnorm.hip — training-mode BatchNorm (+ fused ReLU / LeakyReLU) forward and backward, and the
// per-channel column sums they are built from.  All HBM-bound: x[rows][C] (NHWC flattened) is streamed
// with 16-byte lane accesses; per-channel sums are reduced per thread -> through LDS per block -> as
// per-block partial rows that a finalize kernel adds in a fixed order in fp64 (bitwise reproducible, no
// float atomics).  Replaces nn.BatchNorm2d + nn.ReLU/nn.LeakyReLU at mnist_dcgan.py:77-87,103-110 and
// conditional_counteRGAN/mnist/models/generator.py:12-20; [torch] semantics cited in pcgan_hip.h.
#include "pcg_common.h"
#include <cstdlib>

namespace pcg {
// dp_rccl.hip: exact global-batch BatchNorm under data parallelism
bool dp_sync_bn();
int dp_world();
int dp_allreduce_f64(double* buf, int64_t n, hipStream_t s);

namespace {

constexpr int CR_THREADS = 256;
constexpr int CR_MAX_BLOCKS = 512;
constexpr int FIN_CH = 8, FIN_SL = 128;  // finalize: 8 channels x (32 | 128) partial-row slices per (256 | 1024)-thread block: the loop over the
                                         // partial rows is latency-bound, so many rows get many slices; few rows the cheaper small block
static inline int fin_threads(int nparts) { return nparts <= 1024 ? FIN_CH * 32 : FIN_CH * FIN_SL; }

struct ColPlan { int vec, CG, TX, TY, nblocks, rows_per_block; };

ColPlan plan_cols(int64_t rows, int C, bool aligned16) {
  ColPlan p;
  p.vec = (C % 4 == 0 && aligned16) ? 4 : 1;
  p.CG = C / p.vec;
  int tx = 1;
  while (tx * 2 <= p.CG && tx * 2 <= CR_THREADS) tx *= 2;
  p.TX = tx;
  p.TY = CR_THREADS / tx;
  // >= 4 rows per thread, <= CR_MAX_BLOCKS blocks
  int64_t rpb = ceil_div64(rows, CR_MAX_BLOCKS);
  const int64_t min_rpb = (int64_t)p.TY * 4;
  if (rpb < min_rpb) rpb = min_rpb;
  p.rows_per_block = (int)rpb;
  p.nblocks = (int)ceil_div64(rows, rpb);
  if (p.nblocks < 1) p.nblocks = 1;
  return p;
}

// ---- element functors: NVAL per-channel quantities for VEC consecutive channels of one row --------
template <int VEC>
__device__ __forceinline__ void ldv(const float* p, size_t idx, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p + idx);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
    v[0] = p[idx];
  }
}
struct FnStats {  // sum x, sum x^2
  static constexpr int NVAL = 2;
  const float* x;
  template <int VEC>
  __device__ __forceinline__ void eval(size_t idx, int c0, double (&o)[2][VEC]) const {
    (void)c0;
    float v[VEC];
    ldv<VEC>(x, idx, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { o[0][e] = (double)v[e]; o[1][e] = (double)v[e] * (double)v[e]; }   // exact in fp64
  }
};
struct FnSum {  // sum dy
  static constexpr int NVAL = 1;
  const float* x;
  template <int VEC>
  __device__ __forceinline__ void eval(size_t idx, int c0, double (&o)[1][VEC]) const {
    (void)c0;
    float v[VEC];
    ldv<VEC>(x, idx, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[0][e] = (double)v[e];
  }
};
struct FnBnBwd {  // sum dz, sum dz*xhat with dz = dy_scale*dy*act'(y)
  static constexpr int NVAL = 2;
  const float* dy; const float* x; const float* y; const float* mean; const float* invstd;
  int act; float slope; float dy_scale;
  const float* gamma; const float* beta;   // y == nullptr: the ReLU / LeakyReLU mask is the sign of the recomputed BatchNorm output
  template <int VEC>
  __device__ __forceinline__ void eval(size_t idx, int c0, double (&o)[2][VEC]) const {
    float g[VEC], xv[VEC], yv[VEC];
    ldv<VEC>(dy, idx, g); ldv<VEC>(x, idx, xv);
    if (act != PCG_ACT_NONE) {
      if (y) ldv<VEC>(y, idx, yv);
      else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float sc, sh;
          bn_fold(gamma[c0 + e], beta[c0 + e], mean[c0 + e], invstd[c0 + e], sc, sh);
          yv[e] = fmaf(xv[e], sc, sh);      // same expression as bn_apply_act
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float dz = dy_scale * g[e] * (act != PCG_ACT_NONE ? act_grad_from_out(yv[e], act, slope) : 1.f);
      const float xh = (xv[e] - mean[c0 + e]) * invstd[c0 + e];
      o[0][e] = (double)dz; o[1][e] = (double)dz * (double)xh;
    }
  }
};

// partial[blk][k][C] (fp64: the reference's CPU path accumulates these sums in double — [torch] at::acc_type<float, false>)
template <int VEC, class Fn>
__global__ void __launch_bounds__(CR_THREADS) colreduce_kernel(Fn fn, int64_t rows, int C, int CG, int TX, int TY,
                                                               int rows_per_block, double* __restrict__ partial) {
  constexpr int NVAL = Fn::NVAL;
  __shared__ double red[NVAL * VEC * CR_THREADS];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  const int passes = (CG + TX - 1) / TX;
  for (int ps = 0; ps < passes; ++ps) {
    const int cg = ps * TX + tx;
    const bool cok = cg < CG;
    double acc[NVAL][VEC];
#pragma unroll
    for (int k = 0; k < NVAL; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[k][e] = 0.0;
    if (cok) {
      for (int64_t r = r0 + ty; r < r1; r += TY) {
        double o[NVAL][VEC];
        fn.template eval<VEC>((size_t)r * C + (size_t)cg * VEC, cg * VEC, o);
#pragma unroll
        for (int k = 0; k < NVAL; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[k][e] += o[k][e];
      }
    }
#pragma unroll
    for (int k = 0; k < NVAL; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[(k * VEC + e) * CR_THREADS + threadIdx.x] = acc[k][e];
    __syncthreads();
    if (ty == 0 && cok) {
#pragma unroll
      for (int k = 0; k < NVAL; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          double s = 0.0;
          for (int j = 0; j < TY; ++j) s += red[(k * VEC + e) * CR_THREADS + j * TX + tx];
          partial[((size_t)blockIdx.x * NVAL + k) * C + cg * VEC + e] = s;
        }
    }
    __syncthreads();
  }
}

// Sum NVAL per-channel partial rows in fp64: thread (c, slice) adds partial blocks slice, slice+FIN_SL, ... and the FIN_SL
// slices are combined in a fixed order through LDS (bitwise reproducible; ~nblocks/(8 FIN_SL) dependent load batches per thread).
// Which physical partial row the i-th row of a sum is.  Identity for the plain finalizes; RowMapSeg for the GROUPED ones (several
// independent batches side by side along the rows of one launch, r04): group g's rows are `seg_len` consecutive rows out of every
// `seg_stride` (one segment per sub-pixel phase of a grad-input launch, a single segment for a forward launch), starting at `off`.
struct RowMapId { __device__ __forceinline__ int operator()(int i) const { return i; } };
struct RowMapSeg {
  int seg_len, seg_stride, off;
  __device__ __forceinline__ int operator()(int i) const { const int q = i / seg_len; return q * seg_stride + off + (i - q * seg_len); }
};
template <int NVAL, class SrcT, class Map = RowMapId>
__device__ __forceinline__ bool finalize_sums(const SrcT* __restrict__ partial, int nblocks, int C, int& c_out,
                                              double (&sum)[NVAL], const Map map = Map()) {
  __shared__ double red[NVAL][FIN_SL][FIN_CH];
  const int SL = blockDim.x / FIN_CH;          // 32 or 128 slices (fin_threads)
  const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double acc[NVAL];
#pragma unroll
  for (int k = 0; k < NVAL; ++k) acc[k] = 0.0;
  if (c < C) {
    int b = sl;
    for (; b + 7 * SL < nblocks; b += 8 * SL) {   // 8 independent loads in flight, added in block order
      SrcT v[8][NVAL];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < NVAL; ++k) v[j][k] = partial[((size_t)map(b + j * SL) * NVAL + k) * C + c];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < NVAL; ++k) acc[k] += (double)v[j][k];
    }
    for (; b < nblocks; b += SL)
#pragma unroll
      for (int k = 0; k < NVAL; ++k) acc[k] += (double)partial[((size_t)map(b) * NVAL + k) * C + c];
  }
#pragma unroll
  for (int k = 0; k < NVAL; ++k) red[k][sl][cl] = acc[k];
  __syncthreads();
  c_out = c;
  if (sl != 0 || c >= C) return false;
#pragma unroll
  for (int k = 0; k < NVAL; ++k) {
    double s = 0.0;
#pragma unroll 8
    for (int j = 0; j < SL; ++j) s += red[k][j][cl];   // (a full unroll of the LDS reads spills to scratch: slow dispatch)
    sum[k] = s;
  }
  return true;
}

template <class SrcT>
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_stats_finalize_kernel(
    const SrcT* __restrict__ partial, int nblocks, int C, double inv_rows, double unbias, float eps, float momentum,
    float* save_mean, float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
    const float* gamma, const float* beta, float* coef /* nullable: [2][C] scale / shift of y = x*sc + sh */) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
  int c;
  double sm[2];
  if (!finalize_sums<2, SrcT>(partial, nblocks, C, c, sm)) return;
  const double mean = sm[0] * inv_rows;
  double var = sm[1] * inv_rows - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (coef) bn_fold(gamma[c], beta[c], (float)mean, (float)(1.0 / sqrt(var + (double)eps)), coef[c], coef[C + c]);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
}

__global__ void __launch_bounds__(FIN_CH * FIN_SL) colsum_finalize_kernel(const double* __restrict__ partial, int nblocks,
                                                                         int C, float* out, int accumulate) {
  int c;
  double sm[1];
  if (!finalize_sums<1, double>(partial, nblocks, C, c, sm)) return;
  out[c] = (accumulate ? out[c] : 0.f) + (float)sm[0];
}

// exact-BatchNorm mode, step 1: this rank's [2][C] sums of the partial rows (same fixed order as the finalize kernels), written
// twice — `local` stays, `global` is all-reduced over the ranks and then finalised as ONE partial row
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_sums_kernel(const double* __restrict__ partial, int nblocks, int C,
                                                                 double* __restrict__ local, double* __restrict__ global) {
  int c;
  double sm[2];
  if (!finalize_sums<2, double>(partial, nblocks, C, c, sm)) return;
  local[c] = sm[0]; local[C + c] = sm[1];
  global[c] = sm[0]; global[C + c] = sm[1];
}

// coef[0][c] = gamma*invstd ; coef[1][c] = mean(dz) ; coef[2][c] = mean(dz*xhat)
template <class SrcT>
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_bwd_finalize_kernel(
    const SrcT* __restrict__ partial, int nblocks, int C, double inv_rows, const float* gamma, const float* invstd,
    float* coef, float* dgamma, float* dbeta, int accumulate, const double* __restrict__ local /* nullable [2][C]: this rank's sums */) {
  int c;
  double sm[2];
  if (!finalize_sums<2, SrcT>(partial, nblocks, C, c, sm)) return;
  // exact-BatchNorm mode: `partial` holds the sums over ALL ranks (they make the two means below); the parameter gradients are
  // this rank's own sums — they are averaged across ranks with the rest of the gradient bucket
  const double l0 = local ? local[c] : sm[0], l1 = local ? local[C + c] : sm[1];
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)l0;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)l1;
  coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
  coef[C + c] = (float)(sm[0] * inv_rows);
  coef[2 * C + c] = (float)(sm[1] * inv_rows);
}

// ---- grouped forms (r04): G independent batches side by side along the rows of ONE launch (the discriminator's real and fake
// passes of mnist_dcgan.py:151-161 as one 2B-row pass).  Each group has its own batch statistics — save_mean / save_invstd /
// coef are [G][..] — taken from its own partial rows in the same fixed order the one-batch finalize uses; the running statistics
// are updated group after group, which is what G successive training-mode forwards do ([torch] momentum update per call).
template <class SrcT>
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_stats_finalize_g_kernel(
    const SrcT* __restrict__ partial, int rows_per_group, int groups, int seg_len, int seg_stride, int C, double inv_rows, double unbias,
    float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += groups;
  for (int g = 0; g < groups; ++g) {
    __syncthreads();                       // the previous group's reads of the LDS slices are done
    int c;
    double sm[2];
    if (!finalize_sums<2, SrcT, RowMapSeg>(partial, rows_per_group, C, c, sm, RowMapSeg{seg_len, seg_stride, g * seg_len})) continue;
    const double mean = sm[0] * inv_rows;
    double var = sm[1] * inv_rows - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[(size_t)g * C + c] = (float)mean;
    save_invstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
  }
}
// coef[g][0][c] = gamma*invstd_g ; coef[g][1][c] = mean_g(dz) ; coef[g][2][c] = mean_g(dz*xhat); dgamma / dbeta: the groups' sums added
// in group order (what G successive backward passes accumulate into .grad)
template <class SrcT>
__global__ void __launch_bounds__(FIN_CH * FIN_SL) bn_bwd_finalize_g_kernel(
    const SrcT* __restrict__ partial, int rows_per_group, int groups, int seg_len, int seg_stride, int C, double inv_rows,
    const float* gamma, const float* invstd, float* coef, float* dgamma, float* dbeta, int accumulate) {
  for (int g = 0; g < groups; ++g) {
    __syncthreads();
    int c;
    double sm[2];
    if (!finalize_sums<2, SrcT, RowMapSeg>(partial, rows_per_group, C, c, sm, RowMapSeg{seg_len, seg_stride, g * seg_len})) continue;
    const bool acc = accumulate || g > 0;
    if (dbeta) dbeta[c] = (acc ? dbeta[c] : 0.f) + (float)sm[0];
    if (dgamma) dgamma[c] = (acc ? dgamma[c] : 0.f) + (float)sm[1];
    float* cg = coef + (size_t)g * 3 * C;
    cg[c] = (gamma ? gamma[c] : 1.f) * invstd[(size_t)g * C + c];
    cg[C + c] = (float)(sm[0] * inv_rows);
    cg[2 * C + c] = (float)(sm[1] * inv_rows);
  }
}

// Many partial rows (the conv epilogues leave one per 64 output rows: 12544 for a 1024x28x28 activation) make the finalize a
// latency-bound loop in a handful of blocks.  Level 1 below sums groups of `rpc` consecutive rows, full rows coalesced, in fp64
// into out[chunk][cols] (cols = NVAL*C); the finalize kernels then read <= PRE_CHUNKS rows of doubles.  Fixed orders throughout.
constexpr int PRE_CHUNKS = 256, PRE_MIN_ROWS = 2048;   // a direct finalize of 2048 rows x 128 channels costs 13 us, the two levels 5 + 5 (r04: DCGAN step -0.025 ms); at 1024 rows no gain
__global__ void __launch_bounds__(256) partial_presum_kernel(const double* __restrict__ partial, int nparts, int cols, int rpc,
                                                             double* __restrict__ out) {
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= cols) return;
  const int r0 = blockIdx.x * rpc;
  int r1 = r0 + rpc; if (r1 > nparts) r1 = nparts;
  const double* src = partial + (size_t)r0 * cols + col;
  double acc = 0.0;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * cols];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
    src += (size_t)8 * cols;
  }
  for (; r < r1; ++r) { acc += *src; src += cols; }
  out[(size_t)blockIdx.x * cols + col] = acc;
}

struct PrePlan { int rpc, nchunks; };
PrePlan plan_presum(int nparts) {
  PrePlan q{0, 0};
  if (nparts < PRE_MIN_ROWS) return q;
  q.rpc = (nparts + PRE_CHUNKS - 1) / PRE_CHUNKS;
  q.nchunks = (nparts + q.rpc - 1) / q.rpc;
  return q;
}
size_t presum_offset(int nparts, int cols) { return (size_t)nparts * cols; }   // the chunk sums follow the partial rows
// returns the fp64 chunk sums (and their count) if the partial rows were pre-summed, nullptr otherwise
const double* launch_presum(const double* partial, int nparts, int cols, int* nchunks, hipStream_t s) {
  const PrePlan q = plan_presum(nparts);
  if (q.nchunks == 0) return nullptr;
  double* out = const_cast<double*>(partial) + presum_offset(nparts, cols);
  hipLaunchKernelGGL(partial_presum_kernel, dim3(q.nchunks, (cols + 255) / 256), dim3(256), 0, s, partial, nparts, cols, q.rpc, out);
  *nchunks = q.nchunks;
  return out;
}

// grouped launches: the same level-1 sums, taken only when a chunk never straddles a group's (or phase's) segment of `*seg` rows;
// on success *partial points at the chunk sums and *per_phase / *seg count chunks
struct GroupPre { int rpc; };
GroupPre launch_presum_g(const double** partial, int nparts, int cols, int* per_phase, int* seg, hipStream_t s) {
  const PrePlan q = plan_presum(nparts);
  if (q.nchunks == 0 || (*seg) % q.rpc != 0 || nparts % q.rpc != 0) return GroupPre{0};
  int nchunks = 0;
  const double* pre = launch_presum(*partial, nparts, cols, &nchunks, s);
  if (!pre) return GroupPre{0};
  *partial = pre; *per_phase /= q.rpc; *seg /= q.rpc;
  return GroupPre{q.rpc};
}

// exact-BatchNorm mode: reduce `src` (partial rows or chunk sums) to this rank's sums, all-reduce them, and hand back what the
// finalize kernel should read instead: one row of global sums.  scratch: 4*C doubles ([2][C] local | [2][C] global).
int sync_exchange(const double* src, int nrows, int C, double* scratch, hipStream_t s) {
  hipLaunchKernelGGL(bn_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nrows)), 0, s, src, nrows, C, scratch, scratch + 2 * C);
  if (int e = launch_status("bn_sums_kernel")) return e;
  return dp_allreduce_f64(scratch + 2 * C, 2 * (int64_t)C, s);
}

// ---- elementwise passes (float4 when C % 4 == 0) -------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) bn_apply_act_kernel(const float* __restrict__ x, size_t n, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float var_eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int act, float slope, const float* __restrict__ residual,
                                                           float alpha, float* __restrict__ y) {
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)((i * VEC) % (size_t)C);
    float v[VEC], rs[VEC];
    if constexpr (VEC == 4) {
      const float4 q = reinterpret_cast<const float4*>(x)[i];
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      const float4 r = residual ? reinterpret_cast<const float4*>(residual)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      rs[0] = r.x; rs[1] = r.y; rs[2] = r.z; rs[3] = r.w;
    } else {
      v[0] = x[i];
      rs[0] = residual ? residual[i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = c0 + e;
      const float is = var_eps >= 0.f ? 1.f / sqrtf(invstd[c] + var_eps) : invstd[c];
      float sc, sh;
      bn_fold(gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f, mean[c], is, sc, sh);
      v[e] = fmaf(alpha, act_apply(fmaf(v[e], sc, sh), act, slope), rs[e]);
    }
    if constexpr (VEC == 4) reinterpret_cast<float4*>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
    else y[i] = v[0];
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, size_t n, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, int act, float slope,
                                                           float dy_scale, float* __restrict__ dx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta) {
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)((i * VEC) % (size_t)C);
    float g[VEC], xv[VEC], yv[VEC];
    if constexpr (VEC == 4) {
      const float4 a = reinterpret_cast<const float4*>(dy)[i];
      const float4 b = reinterpret_cast<const float4*>(x)[i];
      const float4 c = (act != PCG_ACT_NONE && y) ? reinterpret_cast<const float4*>(y)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w;
      xv[0] = b.x; xv[1] = b.y; xv[2] = b.z; xv[3] = b.w;
      yv[0] = c.x; yv[1] = c.y; yv[2] = c.z; yv[3] = c.w;
    } else {
      g[0] = dy[i]; xv[0] = x[i]; yv[0] = (act != PCG_ACT_NONE && y) ? y[i] : 1.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = c0 + e;
      if (act != PCG_ACT_NONE && !y) { float sc, sh; bn_fold(gamma[c], beta[c], mean[c], invstd[c], sc, sh); yv[e] = fmaf(xv[e], sc, sh); }
      const float dz = dy_scale * g[e] * (act != PCG_ACT_NONE ? act_grad_from_out(yv[e], act, slope) : 1.f);
      const float xh = (xv[e] - mean[c]) * invstd[c];
      g[e] = coef[c] * (dz - coef[C + c] - xh * coef[2 * C + c]);
    }
    if constexpr (VEC == 4) reinterpret_cast<float4*>(dx)[i] = make_float4(g[0], g[1], g[2], g[3]);
    else dx[i] = g[0];
  }
}

// ---- fast paths: (C/4) is a power of two <= 256, so a thread's channel quad is the same in every grid-stride step:
// the per-channel coefficients live in registers and the loop body is pure streaming (two 16-byte accesses per tensor in
// flight per thread).
// A/B knobs of the streaming passes (environment, read once): grid cap, loads in flight per thread, non-temporal output stores.
// PCG_BN_NT: 0 never, 1 always, unset: when the tensor is at least PCG_BN_NT_MB (default 160) MB — measured r04 (scripts/probes/bn_pass_probe.py):
// a 205 MB CounteRGAN activation applies in 63.7 us with non-temporal stores against 82.4 without (its input and output together
// exceed the 256 MB Infinity Cache, and ordinary stores evict the input lines ahead of their reads); 134 MB DCGAN tensors: no change.
struct BnTune { int blocks, depth, nt; size_t nt_bytes; };
static const BnTune& bn_tune() {
  static const BnTune t = [] {
    auto env = [](const char* k, int d) { const char* e = getenv(k); return e ? atoi(e) : d; };
    return BnTune{env("PCG_BN_BLOCKS", 4096), env("PCG_BN_DEPTH", 2), env("PCG_BN_NT", -1), (size_t)env("PCG_BN_NT_MB", 160) * 1000000u};
  }();
  return t;
}
static bool bn_use_nt(size_t n4) { const BnTune& t = bn_tune(); return t.nt < 0 ? n4 * 16 >= t.nt_bytes : t.nt != 0; }
template <bool NT>
__device__ __forceinline__ void bn_store4(float4* at, const float4& v) {
  typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
  if constexpr (NT) __builtin_nontemporal_store(nt_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f32x4*>(at)); else *at = v;
}

template <int DEPTH, bool NT>
__global__ void __launch_bounds__(256) bn_apply_act_fast_kernel(const float4* __restrict__ x, size_t n4, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                float var_eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int act, float slope,
                                                                const float4* __restrict__ residual, float alpha,
                                                                float4* __restrict__ y, size_t group_n4) {
  if (group_n4) {      // grouped form: blockIdx.y = group; its rows are contiguous, its statistics the y-th [C] row of mean / invstd
    const size_t g = blockIdx.y;
    x += g * group_n4; y += g * group_n4; if (residual) residual += g * group_n4;
    mean += g * C; invstd += g * C; n4 = group_n4;
  }
  const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  const int c0 = (int)(gtid % (size_t)(C >> 2)) * 4;
  float sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    const float is = var_eps >= 0.f ? 1.f / sqrtf(invstd[c] + var_eps) : invstd[c];
    bn_fold(gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f, mean[c], is, sc[e], sh[e]);
  }
  auto one = [&](float4 q, float4 r) {
    q.x = fmaf(alpha, act_apply(fmaf(q.x, sc[0], sh[0]), act, slope), r.x);
    q.y = fmaf(alpha, act_apply(fmaf(q.y, sc[1], sh[1]), act, slope), r.y);
    q.z = fmaf(alpha, act_apply(fmaf(q.z, sc[2], sh[2]), act, slope), r.z);
    q.w = fmaf(alpha, act_apply(fmaf(q.w, sc[3], sh[3]), act, slope), r.w);
    return q;
  };
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  auto put = [&](size_t at, const float4& v) { bn_store4<NT>(y + at, v); };
  size_t i = gtid;
  for (; i + (DEPTH - 1) * stride < n4; i += DEPTH * stride) {
    float4 q[DEPTH], r[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) { q[j] = x[i + j * stride]; r[j] = residual ? residual[i + j * stride] : zero; }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) put(i + j * stride, one(q[j], r[j]));
  }
  for (; i < n4; i += stride) put(i, one(x[i], residual ? residual[i] : zero));
}
template <class... A>
static void launch_bn_apply_fast(dim3 grid, hipStream_t s, size_t n4_all, A... a) {
  const BnTune& t = bn_tune();
  const bool nt = bn_use_nt(n4_all);
  if (t.depth == 4 && nt) hipLaunchKernelGGL((bn_apply_act_fast_kernel<4, true>), grid, dim3(256), 0, s, a...);
  else if (t.depth == 4) hipLaunchKernelGGL((bn_apply_act_fast_kernel<4, false>), grid, dim3(256), 0, s, a...);
  else if (nt) hipLaunchKernelGGL((bn_apply_act_fast_kernel<2, true>), grid, dim3(256), 0, s, a...);
  else hipLaunchKernelGGL((bn_apply_act_fast_kernel<2, false>), grid, dim3(256), 0, s, a...);
}

template <bool NT>
__global__ void __launch_bounds__(256) bn_bwd_apply_fast_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                                const float4* __restrict__ y, size_t n4, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ coef, int act, float slope,
                                                                float dy_scale, float4* __restrict__ dx, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, double* __restrict__ colpart,
                                                                size_t group_n4) {
  if (group_n4) {      // grouped form (no fused column sums): blockIdx.y = group, coef is [G][3][C]
    const size_t g = blockIdx.y;
    dy += g * group_n4; x += g * group_n4; dx += g * group_n4; if (y) y += g * group_n4;
    mean += g * C; invstd += g * C; coef += g * 3 * C; n4 = group_n4;
  }
  const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  const int c0 = (int)(gtid % (size_t)(C >> 2)) * 4;
  float mu[4], is[4], k0[4], k1[4], k2[4], msc[4], msh[4];
  // colpart != nullptr: also the column sums of the dx written here — the bias gradient of the convolution in front of this
  // BatchNorm (analytically zero, numerically rounding residue: the reference computes it, so it is computed) without a
  // separate pass over dx.  fp64 per thread (a thread's channel quad is fixed), fixed-order block sum, one partial row per block.
  double cs[4] = {0.0, 0.0, 0.0, 0.0};
  const bool premask = act != PCG_ACT_NONE && y == nullptr;     // mask from the recomputed BatchNorm output: y is not read
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    mu[e] = mean[c]; is[e] = invstd[c]; k0[e] = coef[c]; k1[e] = coef[C + c]; k2[e] = coef[2 * C + c];
    msc[e] = 0.f; msh[e] = 0.f;
    if (premask) bn_fold(gamma[c], beta[c], mean[c], invstd[c], msc[e], msh[e]);
  }
  const bool has_act = act != PCG_ACT_NONE;
  auto one = [&](float4 g, float4 xv, float4 yv) {
    float gg[4] = {g.x, g.y, g.z, g.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (premask) yy[e] = fmaf(xx[e], msc[e], msh[e]);
      const float dz = dy_scale * gg[e] * (has_act ? act_grad_from_out(yy[e], act, slope) : 1.f);
      const float xh = (xx[e] - mu[e]) * is[e];
      gg[e] = k0[e] * (dz - k1[e] - xh * k2[e]);
    }
    return make_float4(gg[0], gg[1], gg[2], gg[3]);
  };
  const float4 ones = make_float4(1.f, 1.f, 1.f, 1.f);
  auto tally = [&](const float4& v) { cs[0] += (double)v.x; cs[1] += (double)v.y; cs[2] += (double)v.z; cs[3] += (double)v.w; };
  size_t i = gtid;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 g0 = dy[i], g1 = dy[i + stride], x0 = x[i], x1 = x[i + stride];
    const float4 y0 = (has_act && !premask) ? y[i] : ones, y1 = (has_act && !premask) ? y[i + stride] : ones;
    const float4 o0 = one(g0, x0, y0), o1 = one(g1, x1, y1);
    bn_store4<NT>(dx + i, o0);
    bn_store4<NT>(dx + i + stride, o1);
    if (colpart) { tally(o0); tally(o1); }
  }
  if (i < n4) {
    const float4 o = one(dy[i], x[i], (has_act && !premask) ? y[i] : ones);
    bn_store4<NT>(dx + i, o);
    if (colpart) tally(o);
  }
  if (colpart) {   // kernel-uniform
    __shared__ double red[4][256];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[e][threadIdx.x] = cs[e];
    __syncthreads();
    const int nq = C >> 2;                    // threads q, q + nq, q + 2 nq, ... of a block own channel quad q (256 % nq == 0)
    if ((int)threadIdx.x < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double sum = 0.0;
        for (int t = threadIdx.x; t < 256; t += nq) sum += red[e][t];
        colpart[(size_t)blockIdx.x * C + 4 * threadIdx.x + e] = sum;
      }
    }
  }
}

template <class... A>
static void launch_bn_bwd_apply_fast(dim3 grid, hipStream_t s, size_t n4_all, A... a) {
  if (bn_use_nt(n4_all)) hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<true>), grid, dim3(256), 0, s, a...);
  else hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<false>), grid, dim3(256), 0, s, a...);
}

bool fast_channels(int C) { return C % 4 == 0 && C / 4 <= 256 && ((C / 4) & (C / 4 - 1)) == 0; }

template <class Fn>
int launch_colreduce(const Fn& fn, int64_t rows, int C, const ColPlan& cp, double* partial, hipStream_t s) {
  if (cp.vec == 4)
    hipLaunchKernelGGL((colreduce_kernel<4, Fn>), dim3(cp.nblocks), dim3(CR_THREADS), 0, s, fn, rows, C, cp.CG, cp.TX, cp.TY,
                       cp.rows_per_block, partial);
  else
    hipLaunchKernelGGL((colreduce_kernel<1, Fn>), dim3(cp.nblocks), dim3(CR_THREADS), 0, s, fn, rows, C, cp.CG, cp.TX, cp.TY,
                       cp.rows_per_block, partial);
  return launch_status("colreduce_kernel");
}

unsigned ew_blocks(size_t nv) {
  size_t b = (nv + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace
}  // namespace pcg

namespace pcg {
// shared with conv_igemm.hip (BatchNorm statistics fused into the conv epilogue): partial[nparts][2][C] -> stats
size_t bn_partial_buffer_bytes(int nparts, int C) {
  // partial[nparts][2][C] doubles (+ the chunk sums of the two-level finalize when there are many rows)
  const PrePlan q = plan_presum(nparts);
  return (presum_offset(nparts, 2 * C) + (size_t)q.nchunks * 2 * C + 4 * (size_t)C) * sizeof(double);   // + exact-BatchNorm scratch
}
// partial[nblocks][2][C] (fp64 column sums of dm and dm*xhat) -> coef[3][C], dgamma / dbeta (thin_conv.hip's fused BatchNorm backward)
int launch_bn_bwd_finalize(const double* partial, int nblocks, int64_t rows, int C, const float* gamma, const float* invstd, float* coef,
                           float* dgamma, float* dbeta, int accumulate, hipStream_t s) {
  PCG_REQUIRE(!dp_sync_bn(), "fused thin BatchNorm backward: not available in the exact global-batch mode");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nblocks)), 0, s, partial, nblocks, C,
                     1.0 / (double)rows, gamma, invstd, coef, dgamma, dbeta, accumulate, (const double*)nullptr);
  return launch_status("bn_bwd_finalize_kernel");
}
// grouped form: group g's sums are partial rows [g*rows_per_group, (g+1)*rows_per_group); coef [G][3][C], invstd [G][C]
int launch_bn_bwd_finalize_g(const double* partial, int rows_per_group, int groups, int64_t rows_g, int C, const float* gamma, const float* invstd,
                             float* coef, float* dgamma, float* dbeta, int accumulate, hipStream_t s) {
  PCG_REQUIRE(!dp_sync_bn(), "fused thin BatchNorm backward: not available in the exact global-batch mode");
  hipLaunchKernelGGL(bn_bwd_finalize_g_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(rows_per_group)), 0, s, partial,
                     rows_per_group, groups, rows_per_group, rows_per_group * groups, C, 1.0 / (double)rows_g, gamma, invstd, coef, dgamma, dbeta,
                     accumulate);
  return launch_status("bn_bwd_finalize_g_kernel");
}
int launch_bn_stats_finalize(const double* partial, int nparts, int64_t rows, int C, float eps, float momentum, float* save_mean,
                             float* save_invstd, float* running_mean, float* running_var, int64_t* nbt, hipStream_t s,
                             bool has_presum_tail, const float* gamma, const float* beta, float* coef) {
  int nchunks = 0;
  const double* pre = has_presum_tail ? launch_presum(partial, nparts, 2 * C, &nchunks, s) : nullptr;
  if (dp_sync_bn()) {
    PCG_REQUIRE(has_presum_tail, "exact BatchNorm: buffer without scratch tail");
    double* scratch = const_cast<double*>(partial) + presum_offset(nparts, 2 * C) + (size_t)plan_presum(nparts).nchunks * 2 * C;
    if (int e = sync_exchange(pre ? pre : partial, pre ? nchunks : nparts, C, scratch, s)) return e;
    partial = scratch + 2 * C; nparts = 1; pre = nullptr;
    rows *= dp_world();
  }
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  if (pre)
    hipLaunchKernelGGL(bn_stats_finalize_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nchunks)), 0, s, pre, nchunks, C,
                       1.0 / (double)rows, unbias, eps, momentum, save_mean, save_invstd, running_mean, running_var, nbt, gamma, beta, coef);
  else
    hipLaunchKernelGGL(bn_stats_finalize_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nparts)), 0, s, partial, nparts, C,
                       1.0 / (double)rows, unbias, eps, momentum, save_mean, save_invstd, running_mean, running_var, nbt, gamma, beta, coef);
  return launch_status("bn_stats_finalize_kernel");
}
}  // namespace pcg

using namespace pcg;

extern "C" size_t pcg_bn_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  // fp64 partial[nblocks][2][C] + fp64 exact-BatchNorm scratch [4][C] + fp32 coef[3][C]; the plan with vec=1 never has more
  // blocks than vec=4
  const ColPlan a = plan_cols(rows, C, true), b = plan_cols(rows, C, false);
  const int nb = a.nblocks > b.nblocks ? a.nblocks : b.nblocks;
  return ((size_t)nb * 2 * C + 4 * (size_t)C) * sizeof(double) + 3 * (size_t)C * sizeof(float);
}
extern "C" size_t pcg_colsum_workspace_bytes(int64_t rows, int32_t C) { return pcg_bn_workspace_bytes(rows, C); }

extern "C" int pcg_bn_train_stats(const float* x, int64_t rows, int32_t C, float eps, float momentum, float* save_mean,
                                  float* save_invstd, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                                  pcg_stream_t stream) {
  return pcg_bn_train_stats_coef(x, rows, C, eps, momentum, save_mean, save_invstd, running_mean, running_var, num_batches_tracked,
                                 nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int pcg_bn_train_stats_coef(const float* x, int64_t rows, int32_t C, float eps, float momentum, float* save_mean,
                                       float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                       const float* gamma, const float* beta, float* coef_out, void* workspace,
                                       size_t workspace_bytes, pcg_stream_t stream) {
  PCG_REQUIRE(x && save_mean && save_invstd && rows > 0 && C > 0, "pcg_bn_train_stats: bad arguments");
  PCG_REQUIRE(!coef_out || (gamma && beta), "pcg_bn_train_stats_coef: coef_out needs gamma and beta");
  if (!workspace || workspace_bytes < pcg_bn_workspace_bytes(rows, C)) {
    set_error("pcg_bn_train_stats: workspace %zu B < required %zu B", workspace_bytes, pcg_bn_workspace_bytes(rows, C));
    return PCG_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const ColPlan cp = plan_cols(rows, C, al16(x));
  double* partial = (double*)workspace;
  FnStats fn{x};
  if (int e = launch_colreduce(fn, rows, C, cp, partial, s)) return e;
  const double* src = partial;
  int nsrc = cp.nblocks;
  if (dp_sync_bn()) {   // exact global-batch statistics: all-reduce this rank's sums, finalise them as one row over rows*world
    double* scratch = partial + (size_t)cp.nblocks * 2 * C;
    if (int e = sync_exchange(partial, cp.nblocks, C, scratch, s)) return e;
    src = scratch + 2 * C; nsrc = 1;
    rows *= dp_world();
  }
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipLaunchKernelGGL(bn_stats_finalize_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nsrc)), 0, s, src, nsrc, C,
                     1.0 / (double)rows, unbias, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                     num_batches_tracked, gamma, beta, coef_out);
  return launch_status("bn_stats_finalize_kernel");
}

extern "C" int pcg_bn_apply_act(const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                float var_eps, const float* gamma, const float* beta, int act, float slope,
                                const float* residual, float alpha, float* y, pcg_stream_t stream) {
  PCG_REQUIRE(x && y && mean && invstd && rows > 0 && C > 0, "pcg_bn_apply_act: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)rows * C;
  if (fast_channels(C) && al16(x) && al16(y) && al16(residual)) {
    unsigned blocks = (unsigned)((n / 4 + 511) / 512);
    if (blocks > (unsigned)bn_tune().blocks) blocks = (unsigned)bn_tune().blocks;
    if (blocks < 1) blocks = 1;
    launch_bn_apply_fast(dim3(blocks), s, n / 4, reinterpret_cast<const float4*>(x), n / 4, C, mean,
                         invstd, var_eps, gamma, beta, act, slope, reinterpret_cast<const float4*>(residual), alpha,
                         reinterpret_cast<float4*>(y), (size_t)0);
  } else if (C % 4 == 0 && al16(x) && al16(y) && al16(residual))
    hipLaunchKernelGGL(bn_apply_act_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), 0, s, x, n, C, mean, invstd, var_eps, gamma, beta, act, slope, residual, alpha, y);
  else
    hipLaunchKernelGGL(bn_apply_act_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, s, x, n, C, mean, invstd, var_eps, gamma, beta, act, slope, residual, alpha, y);
  return launch_status("bn_apply_act_kernel");
}

constexpr int COL_MAX_BLOCKS = 4096;   // grid cap of the fast apply kernels
constexpr int COL_SUM_BLOCKS = 1024;   // ... when the pass also leaves column sums (one fp64 partial row per block): the finalize behind it
                                       // walks the rows in a latency-bound loop (12 us at 4096 rows, r04 census of the CounteRGAN step)
static size_t colpart_bytes(int32_t C) { return (size_t)COL_MAX_BLOCKS * C * sizeof(double); }

static int bn_act_bwd_impl(const float* dy, const float* x, const float* y, int64_t rows, int32_t C, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, int act, float slope, float dy_scale, float* dx,
                           float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, pcg_stream_t stream,
                           float* dcol = nullptr, int accumulate_col = 0) {
  PCG_REQUIRE(dy && x && mean && invstd && dx && rows > 0 && C > 0, "pcg_bn_act_bwd: bad arguments");
  PCG_REQUIRE(y || act == PCG_ACT_NONE || (gamma && beta && (act == PCG_ACT_RELU || act == PCG_ACT_LRELU)),
              "pcg_bn_act_bwd: without y the activation must be ReLU / LeakyReLU and gamma, beta must be given");
  if (!workspace || workspace_bytes < pcg_bn_workspace_bytes(rows, C)) {
    set_error("pcg_bn_act_bwd: workspace %zu B < required %zu B", workspace_bytes, pcg_bn_workspace_bytes(rows, C));
    return PCG_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = al16(dy) && al16(x) && al16(y) && al16(dx);  // al16(nullptr) is true
  const ColPlan cp = plan_cols(rows, C, aligned);
  double* partial = (double*)workspace;
  double* scratch = partial + (size_t)cp.nblocks * 2 * C;
  float* coef = reinterpret_cast<float*>(scratch + 4 * (size_t)C);
  // fused column sums of dx: partial rows behind everything else (only when the caller sized the workspace for them)
  double* colpart = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + ((pcg_bn_workspace_bytes(rows, C) + 7) & ~(size_t)7));
  PCG_REQUIRE(!dcol || workspace_bytes >= ((pcg_bn_workspace_bytes(rows, C) + 7) & ~(size_t)7) + colpart_bytes(C),
              "pcg_bn_act_bwd_db: workspace smaller than pcg_bn_db_workspace_bytes");
  FnBnBwd fn{dy, x, y, mean, invstd, act, slope, dy_scale, gamma, beta};
  if (int e = launch_colreduce(fn, rows, C, cp, partial, s)) return e;
  const double* src = partial;
  const double* local = nullptr;
  int nsrc = cp.nblocks;
  int64_t rows_all = rows;
  if (dp_sync_bn()) {
    if (int e = sync_exchange(partial, cp.nblocks, C, scratch, s)) return e;
    src = scratch + 2 * C; nsrc = 1; local = scratch;
    rows_all = rows * dp_world();
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nsrc)), 0, s, src, nsrc, C,
                     1.0 / (double)rows_all, gamma, invstd, coef, dgamma, dbeta, accumulate, local);
  if (int e = launch_status("bn_bwd_finalize_kernel")) return e;
  const size_t n = (size_t)rows * C;
  const bool fused_col = dcol != nullptr;
  if (fast_channels(C) && aligned) {
    unsigned blocks = (unsigned)((n / 4 + 511) / 512);
    if (blocks > (fused_col ? COL_SUM_BLOCKS : COL_MAX_BLOCKS)) blocks = fused_col ? COL_SUM_BLOCKS : COL_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    launch_bn_bwd_apply_fast(dim3(blocks), s, n / 4, reinterpret_cast<const float4*>(dy),
                             reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y), n / 4, C, mean, invstd,
                             (const float*)coef, act, slope, dy_scale, reinterpret_cast<float4*>(dx), gamma, beta, fused_col ? colpart : (double*)nullptr,
                             (size_t)0);
    if (fused_col) {
      if (int e = launch_status("bn_bwd_apply_kernel")) return e;
      hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads((int)blocks)), 0, s, (const double*)colpart,
                         (int)blocks, C, dcol, accumulate_col);
      return launch_status("colsum_finalize_kernel");
    }
  } else if (C % 4 == 0 && aligned)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), 0, s, dy, x, y, n, C, mean, invstd,
                       (const float*)coef, act, slope, dy_scale, dx, gamma, beta);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, s, dy, x, y, n, C, mean, invstd,
                       (const float*)coef, act, slope, dy_scale, dx, gamma, beta);
  if (int e = launch_status("bn_bwd_apply_kernel")) return e;
  if (dcol) {   // generic shapes: the separate reduction over the dx just written (the partial rows are free again)
    const ColPlan cq = plan_cols(rows, C, al16(dx));
    FnSum fs{dx};
    if (int e = launch_colreduce(fs, rows, C, cq, partial, s)) return e;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(cq.nblocks)), 0, s, (const double*)partial, cq.nblocks, C,
                       dcol, accumulate_col);
    return launch_status("colsum_finalize_kernel");
  }
  return PCG_OK;
}

extern "C" int pcg_bn_act_bwd(const float* dy, const float* x, const float* y, int64_t rows, int32_t C, const float* mean,
                              const float* invstd, const float* gamma, int act, float slope, float dy_scale, float* dx,
                              float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                              pcg_stream_t stream) {
  PCG_REQUIRE(y || act == PCG_ACT_NONE, "pcg_bn_act_bwd: y is required with an activation (or use pcg_bn_act_bwd_premask)");
  return bn_act_bwd_impl(dy, x, y, rows, C, mean, invstd, gamma, nullptr, act, slope, dy_scale, dx, dgamma, dbeta, accumulate, workspace,
                         workspace_bytes, stream);
}

extern "C" int pcg_bn_act_bwd_premask(const float* dy, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int act, float slope, float dy_scale, float* dx,
                                      float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                                      pcg_stream_t stream) {
  return bn_act_bwd_impl(dy, x, nullptr, rows, C, mean, invstd, gamma, beta, act, slope, dy_scale, dx, dgamma, dbeta, accumulate, workspace,
                         workspace_bytes, stream);
}

extern "C" size_t pcg_bn_bwd_partial_workspace_bytes(int32_t C) { return (size_t)3 * C * sizeof(float); }
extern "C" size_t pcg_bn_bwd_partial_db_workspace_bytes(int32_t C) {
  return C > 0 ? (((size_t)3 * C * sizeof(float) + 7) & ~(size_t)7) + (size_t)4096 * C * sizeof(double) : 0;
}

static int bn_bwd_partial_impl(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                               const float* gamma, const void* partial_, int32_t nparts, float* dx, float* dgamma, float* dbeta,
                               int accumulate, void* workspace, size_t workspace_bytes, pcg_stream_t stream, float* dcol,
                               int accumulate_col, float dm_scale = 1.f) {
  const double* partial = static_cast<const double*>(partial_);
  PCG_REQUIRE(!dcol || workspace_bytes >= pcg_bn_bwd_partial_db_workspace_bytes(C),
              "pcg_bn_bwd_partial_db: workspace smaller than pcg_bn_bwd_partial_db_workspace_bytes");
  PCG_REQUIRE(dm && x && mean && invstd && partial && dx && rows > 0 && C > 0 && nparts > 0, "pcg_bn_bwd_partial: bad arguments");
  PCG_REQUIRE(((uintptr_t)partial & 7) == 0, "pcg_bn_bwd_partial: the partial-sum buffer must be 8-byte aligned");
  if (!workspace || workspace_bytes < pcg_bn_bwd_partial_workspace_bytes(C)) {
    set_error("pcg_bn_bwd_partial: workspace %zu B < required %zu B", workspace_bytes, pcg_bn_bwd_partial_workspace_bytes(C));
    return PCG_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* coef = (float*)workspace;
  int nchunks = 0;
  const double* pre = launch_presum(partial, nparts, 2 * C, &nchunks, s);   // the buffer of pcg_conv2d_*_bn_workspace_bytes has the tail
  const double* src = pre ? pre : partial;
  const double* local = nullptr;
  int nsrc = pre ? nchunks : nparts;
  int64_t rows_all = rows;
  if (dp_sync_bn()) {
    double* scratch = const_cast<double*>(partial) + presum_offset(nparts, 2 * C) + (size_t)plan_presum(nparts).nchunks * 2 * C;
    if (int e = sync_exchange(src, nsrc, C, scratch, s)) return e;
    src = scratch + 2 * C; nsrc = 1; local = scratch;
    rows_all = rows * dp_world();
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(nsrc)), 0, s, src, nsrc, C,
                     1.0 / (double)rows_all, gamma, invstd, coef, dgamma, dbeta, accumulate, local);
  if (int e = launch_status("bn_bwd_finalize_kernel")) return e;
  const bool aligned = al16(dm) && al16(x) && al16(dx);
  const size_t n = (size_t)rows * C;
  double* colpart = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + (((size_t)3 * C * sizeof(float) + 7) & ~(size_t)7));
  if (fast_channels(C) && aligned) {
    unsigned blocks = (unsigned)((n / 4 + 511) / 512);
    if (blocks > (dcol ? COL_SUM_BLOCKS : COL_MAX_BLOCKS)) blocks = dcol ? COL_SUM_BLOCKS : COL_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    launch_bn_bwd_apply_fast(dim3(blocks), s, n / 4, reinterpret_cast<const float4*>(dm),
                             reinterpret_cast<const float4*>(x), (const float4*)nullptr, n / 4, C, mean, invstd, (const float*)coef, (int)PCG_ACT_NONE,
                             0.f, dm_scale, reinterpret_cast<float4*>(dx), gamma, (const float*)nullptr, dcol ? colpart : (double*)nullptr, (size_t)0);
    if (int e = launch_status("bn_bwd_apply_kernel")) return e;
    if (dcol) {
      hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads((int)blocks)), 0, s, (const double*)colpart,
                         (int)blocks, C, dcol, accumulate_col);
      return launch_status("colsum_finalize_kernel");
    }
    return PCG_OK;
  } else if (C % 4 == 0 && aligned)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), 0, s, dm, x, (const float*)nullptr, n, C, mean, invstd,
                       (const float*)coef, PCG_ACT_NONE, 0.f, dm_scale, dx, gamma, (const float*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, s, dm, x, (const float*)nullptr, n, C, mean, invstd,
                       (const float*)coef, PCG_ACT_NONE, 0.f, dm_scale, dx, gamma, (const float*)nullptr);
  if (int e = launch_status("bn_bwd_apply_kernel")) return e;
  if (dcol) {
    const ColPlan cq = plan_cols(rows, C, al16(dx));
    FnSum fs{dx};
    if (int e = launch_colreduce(fs, rows, C, cq, colpart, s)) return e;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(cq.nblocks)), 0, s, (const double*)colpart, cq.nblocks, C,
                       dcol, accumulate_col);
    return launch_status("colsum_finalize_kernel");
  }
  return PCG_OK;
}

extern "C" int pcg_bn_bwd_partial(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                  const float* gamma, const void* partial, int32_t nparts, float* dx, float* dgamma, float* dbeta,
                                  int accumulate, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return bn_bwd_partial_impl(dm, x, rows, C, mean, invstd, gamma, partial, nparts, dx, dgamma, dbeta, accumulate, workspace, workspace_bytes,
                             stream, nullptr, 0);
}
extern "C" int pcg_bn_bwd_partial_db(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                     const float* gamma, const void* partial, int32_t nparts, float dm_scale, float* dx, float* dgamma,
                                     float* dbeta, int accumulate, float* dcol, int accumulate_col, void* workspace,
                                     size_t workspace_bytes, pcg_stream_t stream) {
  return bn_bwd_partial_impl(dm, x, rows, C, mean, invstd, gamma, partial, nparts, dx, dgamma, dbeta, accumulate, workspace, workspace_bytes,
                             stream, dcol, accumulate_col, dm_scale);
}
extern "C" size_t pcg_bn_db_workspace_bytes(int64_t rows, int32_t C) {
  return rows > 0 && C > 0 ? ((pcg_bn_workspace_bytes(rows, C) + 7) & ~(size_t)7) + colpart_bytes(C) : 0;
}
extern "C" int pcg_bn_act_bwd_db(const float* dy, const float* x, const float* y, int64_t rows, int32_t C, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, int act, float slope, float dy_scale, float* dx,
                                 float* dgamma, float* dbeta, int accumulate, float* dcol, int accumulate_col, void* workspace,
                                 size_t workspace_bytes, pcg_stream_t stream) {
  PCG_REQUIRE(dcol != nullptr, "pcg_bn_act_bwd_db: null column-sum output");
  return bn_act_bwd_impl(dy, x, y, rows, C, mean, invstd, gamma, beta, act, slope, dy_scale, dx, dgamma, dbeta, accumulate, workspace,
                         workspace_bytes, stream, dcol, accumulate_col);
}

// ---- grouped BatchNorm (G independent batches along the rows of one tensor; see bn_stats_finalize_g_kernel) --------------------
namespace pcg {
// partial rows of a grouped conv launch -> per-group statistics.  nphases: sub-pixel phases of a grad-input launch (1 for a forward
// launch); every phase holds nparts / nphases rows, the g-th G-th of them belong to group g.
int launch_bn_stats_finalize_g(const double* partial, int nparts, int nphases, int groups, int64_t rows_per_group, int C, float eps,
                               float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                               int64_t* nbt, hipStream_t s) {
  PCG_REQUIRE(!dp_sync_bn(), "grouped BatchNorm: not available in the exact global-batch mode (pcg_dp_sync_batchnorm)");
  PCG_REQUIRE(groups >= 1 && nphases >= 1 && nparts % (nphases * groups) == 0, "grouped BatchNorm: %d partial rows do not split into %d phases x %d groups", nparts, nphases, groups);
  int per_phase = nparts / nphases, seg = per_phase / groups;
  // many partial rows (the two passes of D2 at batch 512 leave 4096): the one-block-per-8-channels finalize walks them in a
  // latency-bound loop, once per group (33 us measured r04).  Level 1 of the two-level finalize first (partial_presum_kernel: every
  // CU adds `rpc` consecutive rows) whenever its chunks respect the group / phase boundaries; the grouped finalize then reads chunks.
  const GroupPre gp = launch_presum_g(&partial, nparts, 2 * C, &per_phase, &seg, s);
  (void)gp;
  const int rpg = seg * nphases;
  const double unbias = rows_per_group > 1 ? (double)rows_per_group / (double)(rows_per_group - 1) : 1.0;
  hipLaunchKernelGGL(bn_stats_finalize_g_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(rpg)), 0, s, partial, rpg, groups,
                     seg, per_phase, C, 1.0 / (double)rows_per_group, unbias, eps, momentum, save_mean, save_invstd, running_mean, running_var, nbt);
  return launch_status("bn_stats_finalize_g_kernel");
}
}  // namespace pcg

static bool grouped_shape_ok(int64_t rows, int32_t C, int32_t groups) {
  return groups >= 1 && groups <= 8 && rows > 0 && rows % groups == 0 && fast_channels(C);
}
static unsigned grouped_apply_blocks(size_t n4_group) {
  unsigned blocks = (unsigned)((n4_group + 511) / 512);
  if (blocks > (unsigned)bn_tune().blocks) blocks = (unsigned)bn_tune().blocks;
  return blocks < 1 ? 1 : blocks;
}

extern "C" int pcg_bn_apply_act_g(const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int act, float slope, float* y, int32_t groups, pcg_stream_t stream) {
  PCG_REQUIRE(x && y && mean && invstd, "pcg_bn_apply_act_g: null pointer");
  PCG_REQUIRE(grouped_shape_ok(rows, C, groups) && al16(x) && al16(y),
              "pcg_bn_apply_act_g: needs rows %% groups == 0, C / 4 a power of two <= 256 and 16-byte aligned tensors (rows %lld, C %d, groups %d)",
              (long long)rows, C, groups);
  const size_t n4g = (size_t)(rows / groups) * C / 4;
  launch_bn_apply_fast(dim3(grouped_apply_blocks(n4g), groups), (hipStream_t)stream, n4g * groups,
                       reinterpret_cast<const float4*>(x), n4g, C, mean, invstd, -1.f, gamma, beta, act, slope, (const float4*)nullptr, 1.f,
                       reinterpret_cast<float4*>(y), n4g);
  return launch_status("bn_apply_act_fast_kernel");
}

extern "C" size_t pcg_bn_bwd_partial_g_workspace_bytes(int32_t C, int32_t groups) { return (size_t)3 * C * groups * sizeof(float); }

static int launch_bwd_finalize_apply_g(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, int act, float slope, const double* partial, int rpg, int seg,
                                       int seg_stride, float* dx, float* dgamma, float* dbeta, int accumulate, int32_t groups, float* coef,
                                       hipStream_t s) {
  const int64_t rows_g = rows / groups;
  hipLaunchKernelGGL(bn_bwd_finalize_g_kernel<double>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(rpg)), 0, s, partial, rpg, groups, seg,
                     seg_stride, C, 1.0 / (double)rows_g, gamma, invstd, coef, dgamma, dbeta, accumulate);
  if (int e = launch_status("bn_bwd_finalize_g_kernel")) return e;
  const size_t n4g = (size_t)rows_g * C / 4;
  launch_bn_bwd_apply_fast(dim3(grouped_apply_blocks(n4g), groups), s, n4g * groups, reinterpret_cast<const float4*>(dm),
                           reinterpret_cast<const float4*>(x), (const float4*)nullptr, n4g, C, mean, invstd, (const float*)coef, act, slope, 1.f,
                           reinterpret_cast<float4*>(dx), gamma, beta, (double*)nullptr, n4g);
  return launch_status("bn_bwd_apply_fast_kernel");
}

// BatchNorm backward of G side-by-side batches from the column sums a grouped fused grad-input epilogue left in `partial`
// (pcg_conv2d_dgrad_bnbwd_g; dm is already masked).  mean / invstd: [G][C]; nphases: sub-pixel phases of the launch that wrote the
// partial rows (pcg_conv2d_dgrad_bn_phases), 1 for a forward-kernel launch.
extern "C" int pcg_bn_bwd_partial_g(const float* dm, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                    const float* gamma, const void* partial_, int32_t nparts, int32_t nphases, float* dx, float* dgamma,
                                    float* dbeta, int accumulate, int32_t groups, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  const double* partial = static_cast<const double*>(partial_);
  PCG_REQUIRE(dm && x && mean && invstd && partial && dx && nparts > 0 && nphases > 0, "pcg_bn_bwd_partial_g: bad arguments");
  PCG_REQUIRE(grouped_shape_ok(rows, C, groups) && al16(dm) && al16(x) && al16(dx) && ((uintptr_t)partial & 7) == 0,
              "pcg_bn_bwd_partial_g: needs rows %% groups == 0, C / 4 a power of two <= 256 and aligned tensors (rows %lld, C %d, groups %d)",
              (long long)rows, C, groups);
  PCG_REQUIRE(nparts % (nphases * groups) == 0, "pcg_bn_bwd_partial_g: %d partial rows do not split into %d phases x %d groups", nparts, nphases, groups);
  PCG_REQUIRE(!dp_sync_bn(), "pcg_bn_bwd_partial_g: not available in the exact global-batch mode");
  if (!workspace || workspace_bytes < pcg_bn_bwd_partial_g_workspace_bytes(C, groups)) {
    set_error("pcg_bn_bwd_partial_g: workspace %zu B < required %zu B", workspace_bytes, pcg_bn_bwd_partial_g_workspace_bytes(C, groups));
    return PCG_ERR_WORKSPACE;
  }
  int per_phase = nparts / nphases, seg = per_phase / groups;
  (void)launch_presum_g(&partial, nparts, 2 * C, &per_phase, &seg, (hipStream_t)stream);   // the buffer of pcg_conv2d_*_bn_workspace_bytes has the tail
  return launch_bwd_finalize_apply_g(dm, x, rows, C, mean, invstd, gamma, nullptr, PCG_ACT_NONE, 0.f, partial, seg * nphases, seg, per_phase, dx,
                                     dgamma, dbeta, accumulate, groups, (float*)workspace, (hipStream_t)stream);
}

// BatchNorm(train) + ReLU / LeakyReLU backward of G side-by-side batches without a fused producer (the layer above is a thin
// layer): one column reduction per group, then the grouped finalize and apply.  The mask is recomputed from x (premask form).
extern "C" size_t pcg_bn_act_bwd_g_workspace_bytes(int64_t rows, int32_t C, int32_t groups) {
  if (rows <= 0 || C <= 0 || groups <= 0 || rows % groups) return 0;
  const ColPlan a = plan_cols(rows / groups, C, true);
  return ((size_t)groups * a.nblocks * 2 * C) * sizeof(double) + (size_t)3 * C * groups * sizeof(float);
}
extern "C" int pcg_bn_act_bwd_premask_g(const float* dy, const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, int act, float slope, float* dx, float* dgamma, float* dbeta,
                                        int accumulate, int32_t groups, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  PCG_REQUIRE(dy && x && mean && invstd && gamma && beta && dx, "pcg_bn_act_bwd_premask_g: null pointer");
  PCG_REQUIRE(act == PCG_ACT_NONE || act == PCG_ACT_RELU || act == PCG_ACT_LRELU, "pcg_bn_act_bwd_premask_g: activation %d is not none / ReLU / LeakyReLU", act);
  PCG_REQUIRE(grouped_shape_ok(rows, C, groups) && al16(dy) && al16(x) && al16(dx) && ((rows / groups) * C) % 4 == 0,
              "pcg_bn_act_bwd_premask_g: needs rows %% groups == 0, C / 4 a power of two <= 256 and aligned tensors (rows %lld, C %d, groups %d)",
              (long long)rows, C, groups);
  PCG_REQUIRE(!dp_sync_bn(), "pcg_bn_act_bwd_premask_g: not available in the exact global-batch mode");
  const size_t need = pcg_bn_act_bwd_g_workspace_bytes(rows, C, groups);
  if (!workspace || workspace_bytes < need) { set_error("pcg_bn_act_bwd_premask_g: workspace %zu B < required %zu B", workspace_bytes, need); return PCG_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows_g = rows / groups;
  const ColPlan cp = plan_cols(rows_g, C, true);
  double* partial = (double*)workspace;
  float* coef = reinterpret_cast<float*>(partial + (size_t)groups * cp.nblocks * 2 * C);
  for (int g = 0; g < groups; ++g) {
    const size_t off = (size_t)g * rows_g * C;
    FnBnBwd fn{dy + off, x + off, nullptr, mean + (size_t)g * C, invstd + (size_t)g * C, act, slope, 1.f, gamma, beta};
    if (int e = launch_colreduce(fn, rows_g, C, cp, partial + (size_t)g * cp.nblocks * 2 * C, s)) return e;
  }
  return launch_bwd_finalize_apply_g(dy, x, rows, C, mean, invstd, gamma, beta, act, slope, partial, cp.nblocks, cp.nblocks, cp.nblocks * groups,
                                     dx, dgamma, dbeta, accumulate, groups, coef, s);
}

extern "C" int pcg_colsum(const float* dy, int64_t rows, int32_t C, float* db, int accumulate, void* workspace,
                          size_t workspace_bytes, pcg_stream_t stream) {
  PCG_REQUIRE(dy && db && rows > 0 && C > 0, "pcg_colsum: bad arguments");
  if (!workspace || workspace_bytes < pcg_colsum_workspace_bytes(rows, C)) {
    set_error("pcg_colsum: workspace %zu B < required %zu B", workspace_bytes, pcg_colsum_workspace_bytes(rows, C));
    return PCG_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const ColPlan cp = plan_cols(rows, C, al16(dy));
  double* partial = (double*)workspace;
  FnSum fn{dy};
  if (int e = launch_colreduce(fn, rows, C, cp, partial, s)) return e;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(fin_threads(cp.nblocks)), 0, s, (const double*)partial, cp.nblocks, C, db, accumulate);
  return launch_status("colsum_finalize_kernel");
}
