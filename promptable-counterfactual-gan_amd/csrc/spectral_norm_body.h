// spectral_norm_body.h — the device bodies of spectral normalisation (one 256-thread block per matrix) and their batched argument
// blocks, shared by the launches of tabular.hip and by the "rider" launches that carry them beside another kernel body
// (house_classifier_fused.hip).  The bodies work on an LDS block handed in by the kernel, so that a rider kernel can overlay it with
// the other body's LDS.
#pragma once
#include "pcg_common.h"

namespace pcg {
namespace {

// ---- spectral normalisation ([torch] nn.utils.spectral_norm, n_power_iterations=1, eps=1e-12), one block per matrix ----
__device__ __forceinline__ float block_sum(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}
// sum over the block's 256 threads: wave butterflies + four partials in a fixed order (two barriers instead of nine)
__device__ __forceinline__ float block_sum4(float v, float* red4) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (red4[0] + red4[1]) + (red4[2] + red4[3]);
  __syncthreads();
  return r;
}
// y[j] = sum_k A(j, k) x[k] for j < NJ with ALL 256 threads: 256 / P2 threads per output (P2 = NJ rounded up to a power of two), each
// a contiguous share of k, the shares added in order through LDS.  TRANS: A(j, k) = M[k][j] (W^T u), else M[j][k] (W v).  The result
// is valid in threads < NJ.  (One thread per output left 64-128 threads walking 64-128 dependent FMAs each, three times per call.)
template <bool TRANS>
__device__ __forceinline__ float sn_matvec(const float* M, int lm, const float* x, int NJ, int NK, float* ps) {
  int p2 = 1;
  while (p2 < NJ) p2 <<= 1;                               // <= 256 (host-checked)
  const int parts = 256 / p2, j = threadIdx.x & (p2 - 1), part = threadIdx.x / p2;
  const int chunk = (NK + parts - 1) / parts, k0 = part * chunk, k1 = min(NK, k0 + chunk);
  float acc = 0.f;
  if (j < NJ) {
#pragma unroll 4
    for (int k = k0; k < k1; ++k) acc = fmaf(TRANS ? M[k * lm + j] : M[j * lm + k], x[k], acc);
  }
  ps[threadIdx.x] = acc;
  __syncthreads();
  float r = 0.f;
  if ((int)threadIdx.x < p2)
    for (int q = 0; q < parts; ++q) r += ps[q * p2 + threadIdx.x];
  __syncthreads();
  return r;
}
// training: v <- normalize(W^T u); u <- normalize(W v)  (in place);  sigma = u . (W v);  Wbar = W / sigma
// The matrix is staged once in LDS (row stride made odd: the row-wise products then hit 32 different banks) and the
// matrix-vector products read it from there; matrices beyond 48 KB take the global-memory form.
constexpr int SN_LDS_FLOATS = 12288;
struct SnFwdLds { float red[256]; float su[256], sv[256]; float sW[SN_LDS_FLOATS]; };     // 51 KB: a block's LDS for the forward body
struct SnBwdLds { float red[256]; float su[256], sv[256]; };
__device__ __forceinline__ void spectral_norm_fwd_body(const float* __restrict__ W, int O, int I, float* __restrict__ u,
                                                       float* __restrict__ v, float eps, int power_iter, float* const* Wbar_r,
                                                       float* const* sigma_r, float* const* uu_r, float* const* vu_r, int reps, int rstride,
                                                       SnFwdLds& lds) {
  float* red = lds.red; float* su = lds.su; float* sv = lds.sv; float* sW = lds.sW;      // O, I <= 256 (host-checked)
  const int ld = (I & 1) ? I : I + 1;
  const bool in_lds = O * ld <= SN_LDS_FLOATS;      // block-uniform
  if (in_lds) {
    // one burst: u, v and the whole matrix are requested before the first LDS store (a load-then-store loop pays one memory
    // latency per trip: 32 trips for the 128 x 64 layer); the register copy of the matrix also feeds the W / sigma writes
    constexpr int PER = SN_LDS_FLOATS / 256;                // 48 elements per thread at most
    const float uv = u[min((int)threadIdx.x, O - 1)], vv = v[min((int)threadIdx.x, I - 1)];      // O, I <= 256
    float wr[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) wr[t] = W[min((int)threadIdx.x + t * 256, O * I - 1)];
    if ((int)threadIdx.x < O) su[threadIdx.x] = uv;
    if ((int)threadIdx.x < I) sv[threadIdx.x] = vv;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int e = threadIdx.x + t * 256;
      if (e < O * I) { const int o = e / I, i = e - o * I; sW[o * ld + i] = wr[t]; }
    }
    __syncthreads();
    // reps > 1: that many successive training-mode calls (each one power iteration from the previous call's u, v, as the module's
    // forward does) in one launch — the matrix is staged once; call r writes its W / sigma, sigma and the u, v it used to set r.
    for (int rp = 0; rp < reps; ++rp) {
      float* __restrict__ Wbar = Wbar_r[rp * rstride]; float* __restrict__ sigma_out = sigma_r[rp * rstride];
      float* __restrict__ u_used = uu_r[rp * rstride]; float* __restrict__ v_used = vu_r[rp * rstride];
      float w;
      if (power_iter) {
        const float t = sn_matvec<true>(sW, ld, su, I, O, red);
        const float nv = sqrtf(block_sum4((int)threadIdx.x < I ? t * t : 0.f, red));
        if ((int)threadIdx.x < I) sv[threadIdx.x] = t / fmaxf(nv, eps);
        __syncthreads();
        w = sn_matvec<false>(sW, ld, sv, O, I, red);
        const float nu = sqrtf(block_sum4((int)threadIdx.x < O ? w * w : 0.f, red));
        if ((int)threadIdx.x < O) su[threadIdx.x] = w / fmaxf(nu, eps);
        __syncthreads();
        if ((int)threadIdx.x < O) u[threadIdx.x] = su[threadIdx.x];
        if ((int)threadIdx.x < I) v[threadIdx.x] = sv[threadIdx.x];
      } else {
        w = sn_matvec<false>(sW, ld, sv, O, I, red);
      }
      if (u_used && (int)threadIdx.x < O) u_used[threadIdx.x] = su[threadIdx.x];
      if (v_used && (int)threadIdx.x < I) v_used[threadIdx.x] = sv[threadIdx.x];
      const float sigma = block_sum4((int)threadIdx.x < O ? w * su[threadIdx.x] : 0.f, red);     // u . (W v): W v is `w` (v unchanged since)
      if (threadIdx.x == 0) sigma_out[0] = sigma;
      const float inv = 1.f / sigma;
#pragma unroll
      for (int t = 0; t < PER; ++t) {
        const int e = threadIdx.x + t * 256;
        if (e < O * I) Wbar[e] = wr[t] * inv;
      }
    }
    return;
  }
  // matrices beyond the LDS image: the global-memory form, one thread per output
  for (int i = threadIdx.x; i < O; i += 256) su[i] = u[i];
  for (int i = threadIdx.x; i < I; i += 256) sv[i] = v[i];
  __syncthreads();
  const float* M = W;
  const int lm = I;
  for (int rp = 0; rp < reps; ++rp) {
  float* __restrict__ Wbar = Wbar_r[rp * rstride]; float* __restrict__ sigma_out = sigma_r[rp * rstride];
  float* __restrict__ u_used = uu_r[rp * rstride]; float* __restrict__ v_used = vu_r[rp * rstride];
  if (power_iter) {
    float t = 0.f;
    if ((int)threadIdx.x < I) { for (int o = 0; o < O; ++o) t = fmaf(M[(size_t)o * lm + threadIdx.x], su[o], t); }
    const float nv = sqrtf(block_sum((int)threadIdx.x < I ? t * t : 0.f, red));
    if ((int)threadIdx.x < I) sv[threadIdx.x] = t / fmaxf(nv, eps);
    __syncthreads();
    float w = 0.f;
    if ((int)threadIdx.x < O) { for (int i = 0; i < I; ++i) w = fmaf(M[(size_t)threadIdx.x * lm + i], sv[i], w); }
    const float nu = sqrtf(block_sum((int)threadIdx.x < O ? w * w : 0.f, red));
    if ((int)threadIdx.x < O) su[threadIdx.x] = w / fmaxf(nu, eps);
    __syncthreads();
    for (int i = threadIdx.x; i < O; i += 256) u[i] = su[i];
    for (int i = threadIdx.x; i < I; i += 256) v[i] = sv[i];
  }
  if (u_used) for (int i = threadIdx.x; i < O; i += 256) u_used[i] = su[i];
  if (v_used) for (int i = threadIdx.x; i < I; i += 256) v_used[i] = sv[i];
  float wv = 0.f;
  if ((int)threadIdx.x < O) { for (int i = 0; i < I; ++i) wv = fmaf(M[(size_t)threadIdx.x * lm + i], sv[i], wv); wv *= su[threadIdx.x]; }
  const float sigma = block_sum((int)threadIdx.x < O ? wv : 0.f, red);
  if (threadIdx.x == 0) sigma_out[0] = sigma;
  const float inv = 1.f / sigma;
  for (int e = threadIdx.x; e < O * I; e += 256) Wbar[e] = W[e] * inv;
  }
}
// dW (+)= (dWbar - (sum dWbar*Wbar) u v^T) / sigma
// LOWREG: the burst in chunks of 16 elements per thread, the operands of the second sweep loaded again (96 loads in flight with their
// addresses are 360 registers; a rider inside a 512-thread launch has 256 per wave) — the same expressions in the same order: the
// same bits (the plain loops at the bottom are the same arithmetic too, but one memory latency per trip: 36 us for two passes)
template <bool LOWREG>
__device__ __forceinline__ void spectral_norm_bwd_body(const float* __restrict__ dWbar, const float* __restrict__ Wbar, int O, int I,
                                                       const float* __restrict__ u, const float* __restrict__ v,
                                                       const float* __restrict__ sigma, float* __restrict__ dW, int accumulate, SnBwdLds& lds) {
  // no contraction: "dW + g" must round g first, so that accumulating here and adding a separately written g later (the two-buffer
  // schedule of the tabular step) give the same bits
#pragma clang fp contract(off)
  float* red = lds.red;
  if (!LOWREG && O * I <= 32 * 256) {       // (block-uniform) every operand of the layer in ONE burst: 32 elements per thread at most
    float* su = lds.su; float* sv = lds.sv;
    constexpr int PER = 32;
    float a[PER], b[PER], w0[PER];
    const float uv = u[min((int)threadIdx.x, O - 1)], vv = v[min((int)threadIdx.x, I - 1)], sg = sigma[0];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int e = min((int)threadIdx.x + 256 * j, O * I - 1); a[j] = dWbar[e]; b[j] = Wbar[e]; w0[j] = accumulate ? dW[e] : 0.f; }
    su[threadIdx.x] = uv; sv[threadIdx.x] = vv;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) t = (int)threadIdx.x + 256 * j < O * I ? fmaf(a[j], b[j], t) : t;      // element order, as the loops below
    const float dot = block_sum(t, red);
    const float inv = 1.f / sg;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = threadIdx.x + 256 * j;
      if (e < O * I) {
        const int o = e / I, i = e - o * I;
        const float g = (a[j] - dot * su[o] * sv[i]) * inv;
        dW[e] = accumulate ? w0[j] + g : g;
      }
    }
    __syncthreads();             // su / sv / red are reused by the next pass of a sequence
    return;
  }
  if (LOWREG && O * I <= 32 * 256) {
    float* su = lds.su; float* sv = lds.sv;
    constexpr int CH = 16, NCH = 32 / CH;
    const float uv = u[min((int)threadIdx.x, O - 1)], vv = v[min((int)threadIdx.x, I - 1)], sg = sigma[0];
    float t = 0.f;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      float a[CH], b[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) { const int e = min((int)threadIdx.x + 256 * (c * CH + j), O * I - 1); a[j] = dWbar[e]; b[j] = Wbar[e]; }
#pragma unroll
      for (int j = 0; j < CH; ++j) t = (int)threadIdx.x + 256 * (c * CH + j) < O * I ? fmaf(a[j], b[j], t) : t;
    }
    su[threadIdx.x] = uv; sv[threadIdx.x] = vv;
    const float dot = block_sum(t, red);
    const float inv = 1.f / sg;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      float a[CH], w0[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) { const int e = min((int)threadIdx.x + 256 * (c * CH + j), O * I - 1); a[j] = dWbar[e]; w0[j] = accumulate ? dW[e] : 0.f; }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int e = threadIdx.x + 256 * (c * CH + j);
        if (e < O * I) {
          const int o = e / I, i = e - o * I;
          const float g = (a[j] - dot * su[o] * sv[i]) * inv;
          dW[e] = accumulate ? w0[j] + g : g;
        }
      }
    }
    __syncthreads();
    return;
  }
  float t = 0.f;
  int e0 = threadIdx.x;
  for (; e0 + 7 * 256 < O * I; e0 += 8 * 256) {      // 16 independent loads in flight, products added in element order
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = dWbar[e0 + 256 * j]; b[j] = Wbar[e0 + 256 * j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) t = fmaf(a[j], b[j], t);
  }
  for (; e0 < O * I; e0 += 256) t = fmaf(dWbar[e0], Wbar[e0], t);
  const float dot = block_sum(t, red);
  const float inv = 1.f / sigma[0];
  for (int e = threadIdx.x; e < O * I; e += 256) {
    const int o = e / I, i = e - o * I;
    const float g = (dWbar[e] - dot * u[o] * v[i]) * inv;
    dW[e] = accumulate ? dW[e] + g : g;
  }
}
// all spectral-norm layers of a net in one launch (one block per layer): the critic has four
constexpr int SN_MAX = 8;      // entries: layers x (reps | passes)
struct SnFwdBatch { const float* W[SN_MAX]; float* u[SN_MAX]; float* v[SN_MAX]; float* Wbar[SN_MAX]; float* sigma[SN_MAX]; float* uu[SN_MAX]; float* vu[SN_MAX]; int O[SN_MAX], I[SN_MAX]; };
struct SnBwdBatch { const float* dWbar[SN_MAX]; const float* Wbar[SN_MAX]; const float* u[SN_MAX]; const float* v[SN_MAX]; const float* sigma[SN_MAX]; float* dW[SN_MAX]; int O[SN_MAX], I[SN_MAX], acc[SN_MAX]; };
// passes > 1: the backward of that many calls of the same layer (entry q * n + l), one after the other into the same dW — the first
// writes or accumulates as its flag says, the others add, exactly as chained launches would; then db_dst[l] += db_src[l] (the bias
// gradient of a later pass, reduced into its own buffer by the grouped weight-gradient launch, which cannot order two writers)
struct SnBwdExtra { float* db_dst[SN_MAX]; const float* db_src[SN_MAX]; };
template <bool LOWREG>
__device__ __forceinline__ void spectral_norm_bwd_seq_body(const SnBwdBatch& b, const SnBwdExtra& x, int n, int passes, int l, SnBwdLds& lds) {
  for (int q = 0; q < passes; ++q) {
    const int e = q * n + l;
    spectral_norm_bwd_body<LOWREG>(b.dWbar[e], b.Wbar[e], b.O[l], b.I[l], b.u[e], b.v[e], b.sigma[e], b.dW[l], q == 0 ? b.acc[l] : 1, lds);
  }
  if (x.db_dst[l])
    for (int i = threadIdx.x; i < b.O[l]; i += 256) x.db_dst[l][i] += x.db_src[l][i];
}
// host side: argument blocks of the batched launches from the C-ABI arrays
inline int fill_sn_fwd_batch(SnFwdBatch& b, int32_t n, int32_t reps, const float* const* w_orig, const int32_t* out_features, const int32_t* in_features,
                             float* const* u, float* const* v, int power_iteration, float* const* w_bar, float* const* sigma, float* const* u_used,
                             float* const* v_used) {
  PCG_REQUIRE(n > 0 && reps >= 1 && n * reps <= SN_MAX && w_orig && out_features && in_features && u && v && w_bar && sigma && u_used && v_used,
              "pcg_spectral_norm_fwd_batched: bad arguments (at most %d layers x calls)", SN_MAX);
  PCG_REQUIRE(reps == 1 || power_iteration, "pcg_spectral_norm_fwd_batched_reps: several calls only differ in training mode");
  for (int l = 0; l < n; ++l) {
    PCG_REQUIRE(w_orig[l] && u[l] && v[l] && out_features[l] > 0 && out_features[l] <= 256 && in_features[l] > 0 && in_features[l] <= 256,
                "pcg_spectral_norm_fwd_batched: layer %d: bad arguments", l);
    b.W[l] = w_orig[l]; b.u[l] = u[l]; b.v[l] = v[l]; b.O[l] = out_features[l]; b.I[l] = in_features[l];
  }
  for (int e = 0; e < n * reps; ++e) {
    PCG_REQUIRE(w_bar[e] && sigma[e], "pcg_spectral_norm_fwd_batched: output set %d: null buffer", e);
    b.Wbar[e] = w_bar[e]; b.sigma[e] = sigma[e]; b.uu[e] = u_used[e]; b.vu[e] = v_used[e];
  }
  return PCG_OK;
}
inline int fill_sn_bwd_batch(SnBwdBatch& b, SnBwdExtra& x, int32_t n, int32_t passes, const float* const* dw_bar, const float* const* w_bar,
                             const int32_t* out_features, const int32_t* in_features, const float* const* u, const float* const* v,
                             const float* const* sigma, float* const* dw_orig, const int32_t* accumulate, float* const* db_dst,
                             const float* const* db_src) {
  PCG_REQUIRE(n > 0 && passes >= 1 && n * passes <= SN_MAX && dw_bar && w_bar && out_features && in_features && u && v && sigma && dw_orig && accumulate,
              "pcg_spectral_norm_bwd_batched: bad arguments (at most %d layers x passes)", SN_MAX);
  for (int l = 0; l < n; ++l) {
    PCG_REQUIRE(dw_orig[l] && out_features[l] > 0 && out_features[l] <= 256 && in_features[l] > 0 && in_features[l] <= 256,
                "pcg_spectral_norm_bwd_batched: layer %d: bad arguments", l);
    b.dW[l] = dw_orig[l]; b.O[l] = out_features[l]; b.I[l] = in_features[l]; b.acc[l] = accumulate[l];
    if (db_dst && db_dst[l]) { PCG_REQUIRE(db_src && db_src[l], "pcg_spectral_norm_bwd_batched_seq: layer %d: db_src missing", l); x.db_dst[l] = db_dst[l]; x.db_src[l] = db_src[l]; }
  }
  for (int e = 0; e < n * passes; ++e) {
    PCG_REQUIRE(dw_bar[e] && w_bar[e] && u[e] && v[e] && sigma[e], "pcg_spectral_norm_bwd_batched: entry %d: null buffer", e);
    b.dWbar[e] = dw_bar[e]; b.Wbar[e] = w_bar[e]; b.u[e] = u[e]; b.v[e] = v[e]; b.sigma[e] = sigma[e];
  }
  return PCG_OK;
}

}  // namespace
}  // namespace pcg
