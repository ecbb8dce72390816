// house_fused.hip — the tabular ResidualGenerator (house_sales_kc_usa/models/generator.py:38-92) as a chain of "segment"
// kernels.  Every tensor of this net is [B][32] and every weight matrix is at most 38 wide: a block owns 64 batch rows (lane = row)
// and four waves that each carry 8 of the 32 channels in registers (the entry segment still gives a thread the whole row); the
// weights of a segment sit in LDS and are read as broadcasts, the rows' input vectors are parked in LDS between layers.  The only
// cross-row dependencies are the ten BatchNorm1d batch statistics, so the net is cut there: a segment ends by writing its
// pre-BatchNorm activations plus per-block column sums, and the next segment starts by turning those sums into mean / invstd
// (fixed order, fp64) — the kernel boundary is the grid barrier.  Forward = 12 launches (op chain: ~95), backward = 11 launches
// + the weight-gradient reductions (op chain: ~190).  Hidden width 32, 5 residual blocks (the reference's configuration)
// are compile-time; other configurations use the op-chain path.
#include <cstring>
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int HH = 32;            // hidden width
constexpr int NBLK = 5;           // residual blocks
constexpr int DIN = 17;           // input_dim   (config.py:14) — compile-time: every per-row loop unrolls into straight FMAs
constexpr int NCLS = 4;           // num_classes (4 price quartiles)
constexpr int MAXCOND = NCLS + DIN;          // cond = (one-hot target, mask)
constexpr int MAXIN = DIN + MAXCOND;         // fc_in input = (x, cond)
constexpr int MAXT = 96;          // packed categorical columns <= 96
constexpr int MAXHEADS = 8;
constexpr int FT = 64;            // threads per block = rows per block: one wave, so 4096 rows spread over 64 CUs

struct GDesc {                    // element offsets into the flat parameter (and gradient) buffer + dimensions
  int fc_in_w, fc_in_b;
  int fc1_w[NBLK], fc1_b[NBLK], bn1_g[NBLK], bn1_b[NBLK], fc2_w[NBLK], fc2_b[NBLK], bn2_g[NBLK], bn2_b[NBLK];
  int fg_w[NBLK], fg_b[NBLK], fb_w[NBLK], fb_b[NBLK];
  int cont_w, cont_b;
  int head_w[MAXHEADS], head_b[MAXHEADS], seg[MAXHEADS + 1];
  int nheads, ncont, D, NC;     // D = input_dim, NC = num_classes; cond = NC + D, inp = D + cond
  int hidden, nblocks;          // checked on the host (32, 5)
};

struct GBufs {
  const float* x; const float* onehot; const float* mask; const float* noise;
  float* inp;                     // [B][D + NC + D]   (x, onehot, mask): also the cond operand of the FiLM weight gradients
  float* H;                       // [NBLK+1][B][HH]  block inputs h_0..h_5
  float* Z1; float* Z2;           // [NBLK][B][HH]    pre-BatchNorm activations
  float* P;                       // [2*NBLK][nblocks][2][HH]  partial column sums (sum, sum of squares) per BatchNorm
  float* SM;                      // [2*NBLK][2][HH]  saved mean / invstd
  float* running;                 // BatchNorm buffers are separate tensors: pointers per layer come in RS
  float* cont; float* logits; float* soft; float* hard;   // outputs ([B][ncont], [B][T] x3; hard nullable)
  int B, nblocks;
  float eps, momentum, tau, res_scale;
};

struct BNState { float* running_mean[2 * NBLK]; float* running_var[2 * NBLK]; int64_t* nbt[2 * NBLK]; };

// ---- small helpers ---------------------------------------------------------------------------------------------------------
// Weight matrices are NOT staged: every lane of a wave needs the same element at the same time, so they are read from
// global memory at wave-uniform addresses — scalar loads into SGPRs that feed v_fma directly (no LDS traffic, no VGPRs).
// Staging them in LDS and reading broadcasts was measured slower (60-86 us vs 34-64 us per segment at B = 4096).
__device__ __forceinline__ void stage(float* dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += FT) dst[i] = src[i];
}

// Matrix-vector products of one segment, per lane (= per batch row).  A fully unrolled 32x32 product is 1024 FMAs of straight-
// line code per layer — measured, the segment kernels then spend their time FETCHING INSTRUCTIONS (every line of a 50-100 KB
// kernel is a cold miss executed once).  Instead the loop over the input index stays rolled: the weight matrix is staged
// transposed in LDS ([i][j], so one input index needs 32 contiguous weights = 8 broadcast ds_read_b128), the lane's input
// vector is parked in LDS ([i][lane], conflict-free), and the 32 accumulators live in registers.  ~40 instructions per
// input index, a few hundred bytes of code per layer.
//   out[j] = b[j] + sum_i W[j][i] in[i]          W row-major [HH][K] in global memory
template <int K>
__device__ __forceinline__ void lin_to32(float* lds, const float* __restrict__ W, const float* __restrict__ b, const float (&in)[K],
                                         float (&out)[HH]) {
  float* Wt = lds; float* bl = lds + K * HH; float* V = bl + HH;
  __syncthreads();                                           // the scratch area is free again
  for (int e = threadIdx.x; e < K * HH; e += FT) { const int j = e / K, i = e - j * K; Wt[i * HH + j] = W[e]; }
  if (threadIdx.x < HH) bl[threadIdx.x] = b[threadIdx.x];
#pragma unroll
  for (int i = 0; i < K; ++i) V[i * FT + threadIdx.x] = in[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < HH; ++j) out[j] = bl[j];
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const float a = V[i * FT + threadIdx.x];
    const float* w = Wt + i * HH;
#pragma unroll
    for (int j = 0; j < HH; ++j) out[j] = fmaf(w[j], a, out[j]);
  }
}
template <int KMAX>
__device__ __forceinline__ void lin_to32_rt(float* lds, const float* __restrict__ W, const float* __restrict__ b, const float (&in)[KMAX], int,
                                            float (&out)[HH]) {
  lin_to32<KMAX>(lds, W, b, in, out);
}
// per-block column sums of v[0..31] and w[0..31] over the block's rows -> part[2][HH] of this block (fixed order)
__device__ __forceinline__ void block_colsums(const float (&v)[HH], const float (&w)[HH], float* red /* [FT][HH+1] */,
                                              float* red2 /* [8][HH] */, float* part) {
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < HH; ++j) red[threadIdx.x * (HH + 1) + j] = pass == 0 ? v[j] : w[j];
    __syncthreads();
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += red[(g * 32 + r) * (HH + 1) + c];     // FT / 32 row groups
    red2[g * HH + c] = s;
    __syncthreads();
    if (threadIdx.x < HH) {
      float t = 0.f;
      for (int q = 0; q < FT / 32; ++q) t += red2[q * HH + threadIdx.x];
      part[pass * HH + threadIdx.x] = t;
    }
  }
  __syncthreads();
}

// mean / invstd of a BatchNorm from the per-block partials (fp64, block order); block 0 also maintains the module's buffers
__device__ __forceinline__ void bn_finalize(const float* P, int nblocks, int B, float eps, float momentum, float* s_mean, float* s_inv,
                                            float* save, float* rmean, float* rvar, int64_t* nbt) {
  // all 64 lanes: lane (c, half) adds the partials of blocks half, half+2, ... of column c (8 loads in flight), then the two
  // halves are combined in a fixed order
  __shared__ double fin[2][2][HH];
  if (threadIdx.x < FT) {   // (the four-wave kernels call this with 256 threads: the first wave does the work)
    const int c = threadIdx.x & 31, half = threadIdx.x >> 5;
    double s0 = 0.0, q0 = 0.0;
#pragma unroll 8
    for (int b = half; b < nblocks; b += 2) { s0 += (double)P[(size_t)b * 2 * HH + c]; q0 += (double)P[(size_t)b * 2 * HH + HH + c]; }
    fin[half][0][c] = s0; fin[half][1][c] = q0;
  }
  __syncthreads();
  if (threadIdx.x < HH) {
    const double s = fin[0][0][threadIdx.x] + fin[1][0][threadIdx.x], q = fin[0][1][threadIdx.x] + fin[1][1][threadIdx.x];
    const double mean = s / B;
    double var = q / B - mean * mean;
    if (var < 0.0) var = 0.0;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    s_mean[threadIdx.x] = (float)mean; s_inv[threadIdx.x] = inv;
    if (blockIdx.x == 0) {
      save[threadIdx.x] = (float)mean; save[HH + threadIdx.x] = inv;
      if (rmean) {
        const double unb = B > 1 ? var * (double)B / (double)(B - 1) : var;
        rmean[threadIdx.x] = (float)((1.0 - momentum) * rmean[threadIdx.x] + momentum * mean);
        rvar[threadIdx.x] = (float)((1.0 - momentum) * rvar[threadIdx.x] + momentum * unb);
        if (threadIdx.x == 0 && nbt) nbt[0] += 1;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void load32(const float* p, size_t row, bool on, float (&v)[HH]) {
#pragma unroll
  for (int j = 0; j < HH; j += 4) {
    float4 q = on ? *reinterpret_cast<const float4*>(p + row * HH + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
  }
}
__device__ __forceinline__ void store32(float* p, size_t row, bool on, const float (&v)[HH]) {
  if (!on) return;
#pragma unroll
  for (int j = 0; j < HH; j += 4) *reinterpret_cast<float4*>(p + row * HH + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
}

// LDS layout shared by the kernels (floats)
struct alignas(16) Smem {
  float red[FT * (HH + 1)];
  float red2[(FT / 32) * HH];
  float gamma[HH], beta[HH], mean[HH], inv[HH];
  float sums[2 * HH];
  float lin[MAXIN * HH + HH + MAXIN * FT];     // scratch of lin_to32: transposed weights, bias, the lanes' input vectors
};

// ---- forward ---------------------------------------------------------------------------------------------------------------
// kind 0: fc_in + ReLU -> h0; z1_0 = fc1_0(h0), partial statistics
__global__ void __launch_bounds__(FT) g_fwd_first_kernel(const float* __restrict__ PRM, GBufs a, GDesc d) {
  __shared__ Smem s;
  constexpr int K = MAXIN;
  const int row = blockIdx.x * FT + threadIdx.x;
  const bool on = row < a.B;
  float inp[MAXIN];
#pragma unroll
  for (int i = 0; i < DIN; ++i) inp[i] = on ? a.x[(size_t)row * DIN + i] : 0.f;
#pragma unroll
  for (int i = 0; i < NCLS; ++i) inp[DIN + i] = on ? a.onehot[(size_t)row * NCLS + i] : 0.f;
#pragma unroll
  for (int i = 0; i < DIN; ++i) inp[DIN + NCLS + i] = on ? a.mask[(size_t)row * DIN + i] : 0.f;
  if (on) {
#pragma unroll
    for (int i = 0; i < K; ++i) a.inp[(size_t)row * K + i] = inp[i];
  }
  float h[HH], z[HH], zz[HH];
  lin_to32_rt<MAXIN>(s.lin, PRM + d.fc_in_w, PRM + d.fc_in_b, inp, K, h);
#pragma unroll
  for (int j = 0; j < HH; ++j) h[j] = h[j] > 0.f ? h[j] : 0.f;
  store32(a.H, row, on, h);
  lin_to32<HH>(s.lin, PRM + d.fc1_w[0], PRM + d.fc1_b[0], h, z);
  store32(a.Z1, row, on, z);
#pragma unroll
  for (int j = 0; j < HH; ++j) { z[j] = on ? z[j] : 0.f; zz[j] = z[j] * z[j]; }
  block_colsums(z, zz, s.red, s.red2, a.P + (size_t)blockIdx.x * 2 * HH);
}

// ---- forward, four waves per 64 rows --------------------------------------------------------------------------------------------
// One thread per row (above) runs 4096 rows as 64 waves on 1024 SIMDs.  Here a block is the same 64 rows (lane = row) and four
// waves, each owning 8 of the 32 channels: its quarter of every matrix-vector product (weights staged transposed in LDS by all 256
// threads and read as broadcasts, the rows' input vectors parked in LDS), of the BatchNorm / FiLM / residual arithmetic and of the
// per-block column sums (wave butterfly, fixed order).  Used for kinds A and B of the forward except the last block's heads.
constexpr int NQ = 4, HQ = HH / NQ;
struct alignas(16) Smem4 {
  float gamma[HH], beta[HH], mean[HH], inv[HH];
  float Wt[2][HH * HH];        // transposed weight images [i][j]: FiLM gamma + beta (21 x 32 each), or one 32 x 32 Linear
  float bl[2][HH];
  float V[HH * FT], V2[HH * FT];   // parked input vectors [i][row]
  float sums[2 * HH];              // backward: mean(dz), mean(dz * xhat) of the BatchNorm in flight
};
template <int K>
__device__ __forceinline__ void stage_t4(float* Wt, float* bl, const float* __restrict__ W, const float* __restrict__ b) {
  for (int e = threadIdx.x; e < K * HH; e += FT * NQ) { const int j = e / K, i = e - j * K; Wt[i * HH + j] = W[e]; }
  if (threadIdx.x < HH) bl[threadIdx.x] = b[threadIdx.x];
}
template <int K>
__device__ __forceinline__ void lin_q(const float* Wt, const float* bl, const float* V, int lane, int q, float (&out)[HQ]) {
#pragma unroll
  for (int j = 0; j < HQ; ++j) out[j] = bl[q * HQ + j];
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const float a = V[i * FT + lane];
    const float* w = Wt + i * HH + q * HQ;
#pragma unroll
    for (int j = 0; j < HQ; ++j) out[j] = fmaf(w[j], a, out[j]);
  }
}
__device__ __forceinline__ void load8(const float* p, size_t row, int q, bool on, float (&v)[HQ]) {
#pragma unroll
  for (int j = 0; j < HQ; j += 4) {
    const float4 t = on ? *reinterpret_cast<const float4*>(p + row * HH + q * HQ + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
  }
}
__device__ __forceinline__ void store8(float* p, size_t row, int q, bool on, const float (&v)[HQ]) {
  if (!on) return;
#pragma unroll
  for (int j = 0; j < HQ; j += 4) *reinterpret_cast<float4*>(p + row * HH + q * HQ + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
}
__device__ __forceinline__ void park8(float* V, int lane, int q, const float (&v)[HQ]) {
#pragma unroll
  for (int j = 0; j < HQ; ++j) V[(q * HQ + j) * FT + lane] = v[j];
}
// cond = (one-hot target, mask): wave q brings in elements q, q+4, ... of its rows
__device__ __forceinline__ void park_cond(float* V, const float* __restrict__ onehot, const float* __restrict__ mask, size_t row, bool on,
                                          int lane, int q) {
  for (int i = q; i < MAXCOND; i += NQ)
    V[i * FT + lane] = !on ? 0.f : (i < NCLS ? onehot[row * NCLS + i] : mask[row * DIN + (i - NCLS)]);
}
// column sums of v and v*v over the block's 64 rows for this wave's 8 channels -> part[2][HH] (butterfly: a fixed order)
__device__ __forceinline__ void wave_colsums(const float (&v)[HQ], bool on, int lane, int q, float* part) {
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    float s1 = on ? v[j] : 0.f, s2 = s1 * s1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (lane == 0) { part[q * HQ + j] = s1; part[HH + q * HQ + j] = s2; }
  }
}

// kind A (block k): bn1 statistics -> a1 = relu(film(bn1(z1))) ; z2 = fc2(a1), partial statistics
__global__ void __launch_bounds__(FT * NQ) g_fwd_a4_kernel(const float* __restrict__ PRM, GBufs a, GDesc d, BNState bs, int k) {
  __shared__ Smem4 s;
  const int li = 2 * k;
  bn_finalize(a.P + (size_t)li * a.nblocks * 2 * HH, a.nblocks, a.B, a.eps, a.momentum, s.mean, s.inv, a.SM + (size_t)li * 2 * HH,
              bs.running_mean[li], bs.running_var[li], bs.nbt[li]);
  if (threadIdx.x < HH) { s.gamma[threadIdx.x] = PRM[d.bn1_g[k] + threadIdx.x]; s.beta[threadIdx.x] = PRM[d.bn1_b[k] + threadIdx.x]; }
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * FT + lane;
  const bool on = row < (size_t)a.B;
  park_cond(s.V, a.onehot, a.mask, row, on, lane, q);
  stage_t4<MAXCOND>(s.Wt[0], s.bl[0], PRM + d.fg_w[k], PRM + d.fg_b[k]);
  stage_t4<MAXCOND>(s.Wt[1], s.bl[1], PRM + d.fb_w[k], PRM + d.fb_b[k]);
  __syncthreads();
  float gam[HQ], bet[HQ], z[HQ], a1[HQ];
  lin_q<MAXCOND>(s.Wt[0], s.bl[0], s.V, lane, q, gam);
  lin_q<MAXCOND>(s.Wt[1], s.bl[1], s.V, lane, q, bet);
  load8(a.Z1 + (size_t)k * a.B * HH, row, q, on, z);
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    const float n = fmaf((z[j] - s.mean[c]) * s.inv[c], s.gamma[c], s.beta[c]);
    const float f = fmaf(gam[j], n, bet[j]);
    a1[j] = f > 0.f ? f : 0.f;
  }
  park8(s.V2, lane, q, a1);
  __syncthreads();                                   // FiLM images read by every wave; a1 complete
  stage_t4<HH>(s.Wt[0], s.bl[0], PRM + d.fc2_w[k], PRM + d.fc2_b[k]);
  __syncthreads();
  float z2[HQ];
  lin_q<HH>(s.Wt[0], s.bl[0], s.V2, lane, q, z2);
  store8(a.Z2 + (size_t)k * a.B * HH, row, q, on, z2);
  wave_colsums(z2, on, lane, q, a.P + ((size_t)(li + 1) * a.nblocks + blockIdx.x) * 2 * HH);
}

// kind B (block k): bn2 statistics -> h_{k+1} = h_k + film(bn2(z2)); z1_{k+1} = fc1_{k+1}(h), partial statistics (after the last
// block the output heads follow instead: g_heads4_kernel)
__global__ void __launch_bounds__(FT * NQ) g_fwd_b4_kernel(const float* __restrict__ PRM, GBufs a, GDesc d, BNState bs, int k) {
  __shared__ Smem4 s;
  const int li = 2 * k + 1;
  bn_finalize(a.P + (size_t)li * a.nblocks * 2 * HH, a.nblocks, a.B, a.eps, a.momentum, s.mean, s.inv, a.SM + (size_t)li * 2 * HH,
              bs.running_mean[li], bs.running_var[li], bs.nbt[li]);
  if (threadIdx.x < HH) { s.gamma[threadIdx.x] = PRM[d.bn2_g[k] + threadIdx.x]; s.beta[threadIdx.x] = PRM[d.bn2_b[k] + threadIdx.x]; }
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * FT + lane;
  const bool on = row < (size_t)a.B;
  park_cond(s.V, a.onehot, a.mask, row, on, lane, q);
  stage_t4<MAXCOND>(s.Wt[0], s.bl[0], PRM + d.fg_w[k], PRM + d.fg_b[k]);
  stage_t4<MAXCOND>(s.Wt[1], s.bl[1], PRM + d.fb_w[k], PRM + d.fb_b[k]);
  __syncthreads();
  float gam[HQ], bet[HQ], z[HQ], h[HQ];
  lin_q<MAXCOND>(s.Wt[0], s.bl[0], s.V, lane, q, gam);
  lin_q<MAXCOND>(s.Wt[1], s.bl[1], s.V, lane, q, bet);
  load8(a.Z2 + (size_t)k * a.B * HH, row, q, on, z);
  load8(a.H + (size_t)k * a.B * HH, row, q, on, h);
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    const float n = fmaf((z[j] - s.mean[c]) * s.inv[c], s.gamma[c], s.beta[c]);
    h[j] += fmaf(gam[j], n, bet[j]);
  }
  store8(a.H + (size_t)(k + 1) * a.B * HH, row, q, on, h);
  if (k == NBLK - 1) return;                         // block-uniform
  park8(s.V2, lane, q, h);
  __syncthreads();
  stage_t4<HH>(s.Wt[0], s.bl[0], PRM + d.fc1_w[k + 1], PRM + d.fc1_b[k + 1]);
  __syncthreads();
  float z1[HQ];
  lin_q<HH>(s.Wt[0], s.bl[0], s.V2, lane, q, z1);
  store8(a.Z1 + (size_t)(k + 1) * a.B * HH, row, q, on, z1);
  wave_colsums(z1, on, lane, q, a.P + ((size_t)(li + 1) * a.nblocks + blockIdx.x) * 2 * HH);
}

// Output heads on four waves: the continuous residual head and the categorical heads (logits, Gumbel-softmax samples) are dealt to
// the waves by the host (largest first onto the least loaded wave); every wave reads the 32-vector of its rows and walks its heads
// exactly as the one-wave kernel does.
struct HeadOwner { signed char owner[MAXHEADS + 1]; };   // [nheads] = the continuous head
__global__ void __launch_bounds__(FT * NQ) g_heads4_kernel(const float* __restrict__ PRM, GBufs a, GDesc d, HeadOwner ho) {
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * FT + lane;
  if (row >= (size_t)a.B) return;
  float h[HH];
  load32(a.H + (size_t)NBLK * a.B * HH, row, true, h);
  const int T = d.seg[d.nheads];
  if (ho.owner[d.nheads] == q) {
    for (int c = 0; c < d.ncont; ++c) {
      float acc = PRM[d.cont_b + c];
#pragma unroll
      for (int i = 0; i < HH; ++i) acc = fmaf(PRM[d.cont_w + c * HH + i], h[i], acc);
      a.cont[row * d.ncont + c] = acc * a.res_scale;
    }
  }
  const float inv_tau = 1.f / a.tau;
  for (int hd = 0; hd < d.nheads; ++hd) {
    if (ho.owner[hd] != q) continue;               // wave-uniform
    const int c0 = d.seg[hd], c1 = d.seg[hd + 1];
    const float* __restrict__ Wh = PRM + d.head_w[hd] - c0 * HH;
    const float* __restrict__ bh = PRM + d.head_b[hd] - c0;
    float mx = -INFINITY;
    for (int c = c0; c < c1; ++c) {
      float acc = bh[c];
#pragma unroll
      for (int i = 0; i < HH; ++i) acc = fmaf(Wh[c * HH + i], h[i], acc);
      a.logits[row * T + c] = acc;
      mx = fmaxf(mx, (acc + a.noise[row * T + c]) * inv_tau);
    }
    float se = 0.f;
    for (int c = c0; c < c1; ++c) se += expf((a.logits[row * T + c] + a.noise[row * T + c]) * inv_tau - mx);
    const float inv = 1.f / se;
    float best = -1.f; int arg = c0;
    for (int c = c0; c < c1; ++c) {
      const float p = expf((a.logits[row * T + c] + a.noise[row * T + c]) * inv_tau - mx) * inv;
      a.soft[row * T + c] = p;
      if (p > best) { best = p; arg = c; }
    }
    if (a.hard) for (int c = c0; c < c1; ++c) a.hard[row * T + c] = c == arg ? 1.f : 0.f;
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------------
struct GBwd {
  float* grads;                           // flat gradient buffer (BatchNorm gamma/beta gradients are written here)
  const float* onehot; const float* mask;
  const float* H; const float* Z1; const float* Z2; const float* SM; const float* soft;
  const float* d_cont; const float* d_logits; const float* d_samples;   // nullable
  float* DH;                              // [NBLK][B][HH]  gradient entering block k from above (k = NBLK-1 .. 0)
  float* DZ1; float* DZ2; float* A1;      // [NBLK][B][HH]
  float* DN1;                             // [B][HH] scratch: gradient at bn1's output of the block in flight
  float* DG; float* DB;                   // [NBLK][B][HH]  FiLM gamma / beta output gradients
  float* DZIN;                            // [B][HH]
  float* DL; float* DC;                   // [B][T], [B][ncont]
  float* Q;                               // [2*NBLK][nblocks][2][HH] partial sums of the BatchNorm backward
  int B, nblocks, accumulate;
  float tau, res_scale;
};

// sums of one BatchNorm backward from the partials; block 0 writes dgamma / dbeta
__device__ __forceinline__ void bnb_finalize(float* sums, const GBwd& a, int li, int g_off, int b_off) {
  __shared__ double fin[2][2][HH];
  if (threadIdx.x < FT) {   // (the four-wave kernels call this with 256 threads: the first wave does the work)
    const float* Q = a.Q + (size_t)li * a.nblocks * 2 * HH;
    const int c = threadIdx.x & 31, half = threadIdx.x >> 5;
    double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
    for (int b = half; b < a.nblocks; b += 2) { t1 += (double)Q[(size_t)b * 2 * HH + c]; t2 += (double)Q[(size_t)b * 2 * HH + HH + c]; }
    fin[half][0][c] = t1; fin[half][1][c] = t2;
  }
  __syncthreads();
  if (threadIdx.x < HH) {
    const double s1 = fin[0][0][threadIdx.x] + fin[1][0][threadIdx.x], s2 = fin[0][1][threadIdx.x] + fin[1][1][threadIdx.x];
    sums[threadIdx.x] = (float)(s1 / a.B); sums[HH + threadIdx.x] = (float)(s2 / a.B);
    if (blockIdx.x == 0) {
      float* gg = a.grads + g_off + threadIdx.x; float* gb = a.grads + b_off + threadIdx.x;
      *gg = a.accumulate ? *gg + (float)s2 : (float)s2;
      *gb = a.accumulate ? *gb + (float)s1 : (float)s1;
    }
  }
  __syncthreads();
}

// ---- backward, four waves per 64 rows (same split as the forward: wave q owns channels 8q .. 8q+7) -------------------------------
// out[8] = sum_j Wl[j][8q + .] * V[j][row]      (Wl = the Linear's weight as stored, [out j][in i]: gradient w.r.t. its input)
__device__ __forceinline__ void lin_tq(const float* Wl, const float* V, int lane, int q, float (&out)[HQ]) {
#pragma unroll
  for (int i = 0; i < HQ; ++i) out[i] = 0.f;
#pragma unroll 1
  for (int j = 0; j < HH; ++j) {
    const float a = V[j * FT + lane];
    const float* w = Wl + j * HH + q * HQ;
#pragma unroll
    for (int i = 0; i < HQ; ++i) out[i] = fmaf(w[i], a, out[i]);
  }
}
__device__ __forceinline__ void stage_asis4(float* Wl, const float* __restrict__ W) {
  for (int e = threadIdx.x; e < HH * HH; e += FT * NQ) Wl[e] = W[e];
}
// column sums of v and w over the block's 64 rows for this wave's 8 channels -> part[2][HH]
__device__ __forceinline__ void wave_colsums2(const float (&v)[HQ], const float (&w)[HQ], int lane, int q, float* part) {
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    float s1 = v[j], s2 = w[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (lane == 0) { part[q * HQ + j] = s1; part[HH + q * HQ + j] = s2; }
  }
}
// part "a" of block k: dn2 = dh * gam; partial sums (dn2, dn2 * xhat2).  Uses s.V / s.Wt[1] (free at every call site).
__device__ __forceinline__ void bwd_part_a4(const float* __restrict__ PRM, Smem4& s, const GBwd& a, const GDesc& d, int k, size_t row, bool on,
                                            int lane, int q, const float (&dh)[HQ]) {
  park_cond(s.V, a.onehot, a.mask, row, on, lane, q);
  stage_t4<MAXCOND>(s.Wt[1], s.bl[1], PRM + d.fg_w[k], PRM + d.fg_b[k]);
  __syncthreads();
  float gam[HQ], z[HQ], v[HQ], w[HQ];
  lin_q<MAXCOND>(s.Wt[1], s.bl[1], s.V, lane, q, gam);
  load8(a.Z2 + (size_t)k * a.B * HH, row, q, on, z);
  const float* sm = a.SM + (size_t)(2 * k + 1) * 2 * HH;
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    const float xh = (z[j] - sm[c]) * sm[HH + c];
    v[j] = on ? dh[j] * gam[j] : 0.f;
    w[j] = v[j] * xh;
  }
  wave_colsums2(v, w, lane, q, a.Q + ((size_t)(2 * k + 1) * a.nblocks + blockIdx.x) * 2 * HH);
}

// first backward kernel: gradients of the heads (dealt to the waves like the forward) -> dh entering the last block; part a
__global__ void __launch_bounds__(FT * NQ) g_bwd_first4_kernel(const float* __restrict__ PRM, GBwd a, GDesc d, HeadOwner ho) {
  __shared__ Smem4 s;
  __shared__ float part[NQ][HH][FT];        // per-wave partial dh
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * FT + lane;
  const bool on = row < (size_t)a.B;
  const int T = d.seg[d.nheads];
  float dh[HH];
#pragma unroll
  for (int i = 0; i < HH; ++i) dh[i] = 0.f;
  if (on) {
    if (ho.owner[d.nheads] == q) {
      for (int c = 0; c < d.ncont; ++c) {
        const float dc = a.d_cont ? a.d_cont[row * d.ncont + c] * a.res_scale : 0.f;
        a.DC[row * d.ncont + c] = dc;
#pragma unroll
        for (int i = 0; i < HH; ++i) dh[i] = fmaf(PRM[d.cont_w + c * HH + i], dc, dh[i]);
      }
    }
    const float inv_tau = 1.f / a.tau;
    for (int hd = 0; hd < d.nheads; ++hd) {
      if (ho.owner[hd] != q) continue;             // wave-uniform
      const int c0 = d.seg[hd], c1 = d.seg[hd + 1];
      const float* __restrict__ Wh = PRM + d.head_w[hd] - c0 * HH;
      float dot = 0.f;
      if (a.d_samples)
        for (int c = c0; c < c1; ++c) dot = fmaf(a.d_samples[row * T + c], a.soft[row * T + c], dot);
      for (int c = c0; c < c1; ++c) {
        float dl = a.d_logits ? a.d_logits[row * T + c] : 0.f;
        if (a.d_samples) { const float y = a.soft[row * T + c]; dl += y * (a.d_samples[row * T + c] - dot) * inv_tau; }
        a.DL[row * T + c] = dl;
#pragma unroll
        for (int i = 0; i < HH; ++i) dh[i] = fmaf(Wh[c * HH + i], dl, dh[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < HH; ++i) part[q][i][lane] = dh[i];
  __syncthreads();
  float d8[HQ];
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    d8[j] = (part[0][c][lane] + part[1][c][lane]) + (part[2][c][lane] + part[3][c][lane]);
  }
  store8(a.DH + (size_t)(NBLK - 1) * a.B * HH, row, q, on, d8);
  bwd_part_a4(PRM, s, a, d, NBLK - 1, row, on, lane, q, d8);
}

// kind B (block k): bn2 backward -> dz2; through fc2 and the ReLU / FiLM -> dn1 and its partial sums; FiLM output gradients
__global__ void __launch_bounds__(FT * NQ) g_bwd_b4_kernel(const float* __restrict__ PRM, GBwd a, GDesc d, int k) {
  __shared__ Smem4 s;
  bnb_finalize(s.sums, a, 2 * k + 1, d.bn2_g[k], d.bn2_b[k]);
  if (threadIdx.x < HH) {
    s.gamma[threadIdx.x] = PRM[d.bn2_g[k] + threadIdx.x]; s.beta[threadIdx.x] = PRM[d.bn2_b[k] + threadIdx.x];
    s.mean[threadIdx.x] = PRM[d.bn1_g[k] + threadIdx.x]; s.inv[threadIdx.x] = PRM[d.bn1_b[k] + threadIdx.x];   // bn1's gamma / beta (names reused)
  }
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * FT + lane;
  const bool on = row < (size_t)a.B;
  park_cond(s.V, a.onehot, a.mask, row, on, lane, q);
  stage_t4<MAXCOND>(s.Wt[0], s.bl[0], PRM + d.fg_w[k], PRM + d.fg_b[k]);
  stage_t4<MAXCOND>(s.Wt[1], s.bl[1], PRM + d.fb_w[k], PRM + d.fb_b[k]);
  __syncthreads();
  float gam[HQ], bet[HQ], z[HQ], dh[HQ], dz2[HQ], da1[HQ], dgam[HQ];
  lin_q<MAXCOND>(s.Wt[0], s.bl[0], s.V, lane, q, gam);
  lin_q<MAXCOND>(s.Wt[1], s.bl[1], s.V, lane, q, bet);
  load8(a.DH + (size_t)k * a.B * HH, row, q, on, dh);
  load8(a.Z2 + (size_t)k * a.B * HH, row, q, on, z);
  const float* sm2 = a.SM + (size_t)(2 * k + 1) * 2 * HH;
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    const float xh = (z[j] - sm2[c]) * sm2[HH + c];
    const float n2 = fmaf(xh, s.gamma[c], s.beta[c]);
    const float dn2 = dh[j] * gam[j];
    dz2[j] = s.gamma[c] * sm2[HH + c] * (dn2 - s.sums[c] - xh * s.sums[HH + c]);
    dgam[j] = dh[j] * n2;
  }
  store8(a.DZ2 + (size_t)k * a.B * HH, row, q, on, dz2);
  park8(s.V2, lane, q, dz2);
  __syncthreads();                                   // FiLM images read by every wave; dz2 complete
  stage_asis4(s.Wt[0], PRM + d.fc2_w[k]);
  __syncthreads();
  lin_tq(s.Wt[0], s.V2, lane, q, da1);
  load8(a.Z1 + (size_t)k * a.B * HH, row, q, on, z);
  const float* sm1 = a.SM + (size_t)(2 * k) * 2 * HH;
  float a1[HQ], v[HQ], w[HQ], dbet[HQ];
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    const float xh = (z[j] - sm1[c]) * sm1[HH + c];
    const float n1 = fmaf(xh, s.mean[c], s.inv[c]);
    const float f = fmaf(gam[j], n1, bet[j]);
    a1[j] = f > 0.f ? f : 0.f;
    const float df1 = f > 0.f ? da1[j] : 0.f;
    dgam[j] = fmaf(df1, n1, dgam[j]);
    dbet[j] = dh[j] + df1;
    v[j] = on ? df1 * gam[j] : 0.f;          // dn1
    w[j] = v[j] * xh;
  }
  store8(a.A1 + (size_t)k * a.B * HH, row, q, on, a1);
  store8(a.DG + (size_t)k * a.B * HH, row, q, on, dgam);
  store8(a.DB + (size_t)k * a.B * HH, row, q, on, dbet);
  store8(a.DN1, row, q, on, v);
  wave_colsums2(v, w, lane, q, a.Q + ((size_t)(2 * k) * a.nblocks + blockIdx.x) * 2 * HH);
}

// kind C (block k): bn1 backward -> dz1; dh_{k-1} = dh_k + fc1^T dz1; then part a of block k-1, or the fc_in ReLU for k = 0
__global__ void __launch_bounds__(FT * NQ) g_bwd_c4_kernel(const float* __restrict__ PRM, GBwd a, GDesc d, int k) {
  __shared__ Smem4 s;
  bnb_finalize(s.sums, a, 2 * k, d.bn1_g[k], d.bn1_b[k]);
  if (threadIdx.x < HH) s.gamma[threadIdx.x] = PRM[d.bn1_g[k] + threadIdx.x];
  stage_asis4(s.Wt[0], PRM + d.fc1_w[k]);
  __syncthreads();
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * FT + lane;
  const bool on = row < (size_t)a.B;
  float z[HQ], dn1[HQ], dz1[HQ], dh[HQ], t[HQ];
  load8(a.Z1 + (size_t)k * a.B * HH, row, q, on, z);
  load8(a.DN1, row, q, on, dn1);
  const float* sm1 = a.SM + (size_t)(2 * k) * 2 * HH;
#pragma unroll
  for (int j = 0; j < HQ; ++j) {
    const int c = q * HQ + j;
    const float xh = (z[j] - sm1[c]) * sm1[HH + c];
    dz1[j] = s.gamma[c] * sm1[HH + c] * (dn1[j] - s.sums[c] - xh * s.sums[HH + c]);
  }
  store8(a.DZ1 + (size_t)k * a.B * HH, row, q, on, dz1);
  park8(s.V2, lane, q, dz1);
  __syncthreads();
  lin_tq(s.Wt[0], s.V2, lane, q, t);
  load8(a.DH + (size_t)k * a.B * HH, row, q, on, dh);
#pragma unroll
  for (int j = 0; j < HQ; ++j) dh[j] += t[j];
  if (k > 0) {                                       // block-uniform
    store8(a.DH + (size_t)(k - 1) * a.B * HH, row, q, on, dh);
    bwd_part_a4(PRM, s, a, d, k - 1, row, on, lane, q, dh);
    return;
  }
  float h0[HQ];
  load8(a.H, row, q, on, h0);
#pragma unroll
  for (int j = 0; j < HQ; ++j) dh[j] = h0[j] > 0.f ? dh[j] : 0.f;
  store8(a.DZIN, row, q, on, dh);
}

}  // namespace
}  // namespace pcg

using namespace pcg;

// output heads (and the continuous head, index nheads) dealt to the four waves: largest first onto the least loaded wave
static HeadOwner deal_heads(const GDesc& d) {
  HeadOwner ho{};
  int size[MAXHEADS + 1], load[NQ] = {0, 0, 0, 0};
  bool done[MAXHEADS + 1] = {};
  for (int h = 0; h < d.nheads; ++h) size[h] = d.seg[h + 1] - d.seg[h];
  size[d.nheads] = d.ncont;
  for (int it = 0; it <= d.nheads; ++it) {
    int best = -1;
    for (int j = 0; j <= d.nheads; ++j)
      if (!done[j] && (best < 0 || size[j] > size[best])) best = j;
    int w = 0;
    for (int j = 1; j < NQ; ++j)
      if (load[j] < load[w]) w = j;
    done[best] = true; ho.owner[best] = (signed char)w; load[w] += size[best];
  }
  return ho;
}

// C-side mirrors of the argument blocks (plain arrays of offsets / pointers: see include/pcgan_hip.h)
extern "C" int pcg_house_g_fwd(const pcg_house_g_desc* desc, const pcg_house_g_fwd_args* args, pcg_stream_t stream) {
  PCG_REQUIRE(desc && args, "pcg_house_g_fwd: null argument block");
  static_assert(sizeof(pcg_house_g_desc) == sizeof(GDesc), "descriptor layouts differ");
  GDesc d;
  std::memcpy(&d, desc, sizeof(d));
  PCG_REQUIRE(desc->hidden == 32 && desc->nblocks == 5, "pcg_house_g_fwd: built for hidden width 32 and 5 residual blocks");
  PCG_REQUIRE(d.D == DIN && d.NC == NCLS && d.nheads >= 0 && d.nheads <= MAXHEADS && d.ncont >= 0 &&
                  d.ncont <= HH && d.seg[d.nheads] <= MAXT,
              "pcg_house_g_fwd: built for input_dim 17, 4 classes, <= 8 heads / 96 packed categories");
  PCG_REQUIRE(args->B > 0 && args->params && args->x && args->onehot && args->mask && args->noise && args->inp && args->H && args->Z1 &&
                  args->Z2 && args->P && args->SM && args->cont && args->logits && args->soft,
              "pcg_house_g_fwd: null buffer");
  GBufs a{};
  a.x = args->x; a.onehot = args->onehot; a.mask = args->mask; a.noise = args->noise; a.inp = args->inp;
  a.H = args->H; a.Z1 = args->Z1; a.Z2 = args->Z2; a.P = args->P; a.SM = args->SM; a.cont = args->cont; a.logits = args->logits;
  a.soft = args->soft; a.hard = args->hard; a.B = args->B; a.nblocks = (args->B + FT - 1) / FT; a.eps = args->eps;
  a.momentum = args->momentum; a.tau = args->tau; a.res_scale = args->res_scale;
  BNState bs{};
  for (int i = 0; i < 2 * NBLK; ++i) { bs.running_mean[i] = args->running_mean[i]; bs.running_var[i] = args->running_var[i]; bs.nbt[i] = args->num_batches_tracked[i]; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a.nblocks), block(FT);
  hipLaunchKernelGGL(g_fwd_first_kernel, grid, block, 0, s, args->params, a, d);
  if (int e = launch_status("g_fwd_first_kernel")) return e;
  const dim3 block4(FT * NQ);
  for (int k = 0; k < NBLK; ++k) {
    hipLaunchKernelGGL(g_fwd_a4_kernel, grid, block4, 0, s, args->params, a, d, bs, k);
    if (int e = launch_status("g_fwd_a4_kernel")) return e;
    hipLaunchKernelGGL(g_fwd_b4_kernel, grid, block4, 0, s, args->params, a, d, bs, k);
    if (int e = launch_status("g_fwd_b4_kernel")) return e;
  }
  hipLaunchKernelGGL(g_heads4_kernel, grid, block4, 0, s, args->params, a, d, deal_heads(d));
  if (int e = launch_status("g_heads4_kernel")) return e;
  return PCG_OK;
}

extern "C" int pcg_house_g_bwd(const pcg_house_g_desc* desc, const pcg_house_g_bwd_args* args, pcg_stream_t stream) {
  PCG_REQUIRE(desc && args, "pcg_house_g_bwd: null argument block");
  GDesc d;
  std::memcpy(&d, desc, sizeof(d));
  PCG_REQUIRE(desc->hidden == 32 && desc->nblocks == 5 && d.D == DIN && d.NC == NCLS, "pcg_house_g_bwd: built for hidden width 32, 5 residual blocks, input_dim 17, 4 classes");
  PCG_REQUIRE(args->B > 0 && args->params && args->grads && args->onehot && args->mask && args->H && args->Z1 && args->Z2 && args->SM &&
                  args->soft && args->DH && args->DZ1 && args->DZ2 && args->A1 && args->DN1 && args->DG && args->DB && args->DZIN &&
                  args->DL && args->DC && args->Q,
              "pcg_house_g_bwd: null buffer");
  GBwd a{};
  a.grads = args->grads; a.onehot = args->onehot; a.mask = args->mask; a.H = args->H; a.Z1 = args->Z1; a.Z2 = args->Z2;
  a.SM = args->SM; a.soft = args->soft; a.d_cont = args->d_cont; a.d_logits = args->d_logits; a.d_samples = args->d_samples; a.DH = args->DH;
  a.DZ1 = args->DZ1; a.DZ2 = args->DZ2; a.A1 = args->A1; a.DN1 = args->DN1; a.DG = args->DG; a.DB = args->DB; a.DZIN = args->DZIN;
  a.DL = args->DL; a.DC = args->DC; a.Q = args->Q; a.B = args->B; a.nblocks = (args->B + FT - 1) / FT; a.accumulate = args->accumulate;
  a.tau = args->tau; a.res_scale = args->res_scale;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a.nblocks), block4(FT * NQ);
  hipLaunchKernelGGL(g_bwd_first4_kernel, grid, block4, 0, s, args->params, a, d, deal_heads(d));
  if (int e = launch_status("g_bwd_first4_kernel")) return e;
  for (int k = NBLK - 1; k >= 0; --k) {
    hipLaunchKernelGGL(g_bwd_b4_kernel, grid, block4, 0, s, args->params, a, d, k);
    if (int e = launch_status("g_bwd_b4_kernel")) return e;
    hipLaunchKernelGGL(g_bwd_c4_kernel, grid, block4, 0, s, args->params, a, d, k);
    if (int e = launch_status("g_bwd_c4_kernel")) return e;
  }
  return PCG_OK;
}
