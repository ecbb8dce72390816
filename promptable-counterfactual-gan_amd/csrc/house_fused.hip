// house_fused.hip — the tabular ResidualGenerator (house_sales_kc_usa/models/generator.py:38-92) as a chain of "segment"
// kernels.  Every tensor of this net is [B][32] and every weight matrix is at most 38 wide: a block owns 64 batch rows and four
// waves, every product runs on v_mfma_f32_16x16x4_f32 (exact fp32) with the activations parked k-major in LDS and the weights
// staged as stored, and the BatchNorm / FiLM / ReLU arithmetic is done on the accumulator layout.  The only
// cross-row dependencies are the ten BatchNorm1d batch statistics, so the net is cut there: a segment ends by writing its
// pre-BatchNorm activations plus per-block column sums, and the next segment starts by turning those sums into mean / invstd
// (fixed order, fp64) — the kernel boundary is the grid barrier.  (A persistent launch whose blocks meet at an arrival counter in
// global memory was built and measured in round 2: a dependent launch inside a single-stream HIP graph costs 1.8 us on this part,
// a 64-block grid barrier with agent-scope release / acquire 3.9 us, 9.5 us at 256 blocks — scripts/probes/grid_barrier_probe.hip —
// and the persistent forward ran 181 us against 157 us.  What a segment costs is its chain of dependent cold global loads, so
// every segment issues ALL its global loads before its first barrier.)  Forward = 12 launches (op chain: ~95), backward = 11
// launches + the weight-gradient reductions (op chain: ~190).  Hidden width 32, 5 residual blocks (the reference's configuration)
// are compile-time; other configurations use the op-chain path.
#include <cstddef>
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
constexpr int FT = 64;            // rows per block (lane = row); a block is four such waves: 4096 rows = 64 blocks of 256 threads

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

// ---- layout of a block ---------------------------------------------------------------------------------------------------------
// A segment block is 64 rows and four waves (256 threads); wave w owns rows 16 w .. 16 w + 15 and both 16-channel tiles of every
// product (v_mfma_f32_16x16x4_f32, see "the forward segments on the matrix cores" below).  The two head kernels use 16-row blocks.
//
// What a segment costs is latency, not work: a launch boundary is ~2 us, but every DEPENDENT global load behind it is a cold miss
// (~1-2 us), and the first version had four or five of them in a row (statistics -> barrier -> weights -> barrier -> rows ->
// barrier -> next weights).  So each segment starts with ONE burst: every weight image, the rows' operands, the BatchNorm
// partials and whatever block 0 will update are requested into registers before the first barrier; the rest runs out of LDS.
constexpr int NQ = 4, NT = FT * NQ;
constexpr int NPART = NT / HH;              // 8 threads per column add the per-block partial statistics

// -- the burst: global -> registers ------------------------------------------------------------------------------------------
template <int K>
struct WRegs { float v[(K * HH + NT - 1) / NT]; float b; };
template <int K>
__device__ __forceinline__ void wload(WRegs<K>& r, const float* __restrict__ W, const float* __restrict__ b) {
#pragma unroll
  for (int t = 0; t < (K * HH + NT - 1) / NT; ++t) r.v[t] = W[min((int)threadIdx.x + t * NT, K * HH - 1)];   // clamped, not guarded: a guard is a branch per load
  r.b = b ? b[threadIdx.x & (HH - 1)] : 0.f;
}
// per-block partial sums [nblocks][2][HH] -> this thread's share (column c = tid & 31, blocks part, part + 8, ...: fixed order, fp64)
struct PRegs { double s, q; };
__device__ __forceinline__ void pload(PRegs& r, const float* __restrict__ P, int nblocks) {
  const int c = threadIdx.x & (HH - 1), part = threadIdx.x >> 5;
  double s = 0.0, q = 0.0;
#pragma unroll 8
  for (int b = part; b < nblocks; b += NPART) { s += (double)P[(size_t)b * 2 * HH + c]; q += (double)P[(size_t)b * 2 * HH + HH + c]; }
  r.s = s; r.q = q;
}
struct CondRegs { float v[(MAXCOND + NQ - 1) / NQ]; };
// cond = (one-hot target, mask): wave q brings in elements q, q+4, ... of its rows
__device__ __forceinline__ void cload(CondRegs& r, const float* __restrict__ onehot, const float* __restrict__ mask, size_t row, bool on, int q) {
#pragma unroll
  for (int t = 0; t < (MAXCOND + NQ - 1) / NQ; ++t) {
    const int i = min(q + t * NQ, MAXCOND - 1);     // (row is clamped by the caller: rows past the batch read the last row and are masked later)
    r.v[t] = i < NCLS ? onehot[row * NCLS + i] : mask[row * DIN + (i - NCLS)];
  }
}
// ---- forward -------------------------------------------------------------------------------------------------------------------
// entry segment: inp = (x, onehot, mask); h0 = relu(fc_in(inp)); z1_0 = fc1_0(h0), partial statistics
struct FSeg { int fg_w, fg_b, fb_w, fb_b, fc_w, fc_b, bn_g, bn_b; float* rmean; float* rvar; int64_t* nbt; int k, li, more; };

// ---- the forward segments on the matrix cores -------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32: lane (li = lane % 16, lq = lane / 16) supplies A[m = li][k = lq], B[k = lq][n = li]; accumulator register r
// holds D[m = 4 lq + r][n = li].  Wave w owns rows 16 w .. 16 w + 15 of the block and BOTH 16-channel tiles: its FiLM products
// (cond x gamma / beta weights, K = 21 padded to 24), the BatchNorm / FiLM / ReLU arithmetic on the accumulator layout, and the next
// Linear (K = 32) — no cross-wave dependency until the column sums.  The lane-per-row form read the weights as LDS broadcasts: 8
// ds_read_b128 per 32 FMAs per lane, four waves on one LDS — 2.3 us of a 7 us segment in the three products alone.
constexpr int MK_C = 24;                    // cond width padded to the MFMA's reduction depth
constexpr int PWM = 36;                     // weight row pitch in LDS (as stored [out][in]): banks 4 li + lq
constexpr int PRM_ = 68;                    // pitch of a k-major [k][64 rows] image
typedef float m16_t __attribute__((ext_vector_type(4)));
struct alignas(16) SmemM {
  float gamma[HH], beta[HH], mean[HH], inv[HH];
  float Wg[HH * PWM], Wb[HH * PWM], Wf[HH * PWM];   // FiLM gamma / beta weights (21 columns used), the segment's Linear (32)
  float bg[HH], bb[HH], bf[HH];
  float C[MK_C * PRM_];                     // cond, k-major (rows 21 .. 23 zero)
  float A[HH * PRM_];                       // the Linear's input, k-major
  float ws[NQ][2][HH];                      // per-wave column sums
  double fin[NPART][2][HH];
};
// registers -> LDS as stored with pitch PWM, pad columns zero
template <int K>
__device__ __forceinline__ void wstore_m(float* Wl, float* bl, const WRegs<K>& r) {
#pragma unroll
  for (int t = 0; t < (K * HH + NT - 1) / NT; ++t) {
    const int e = threadIdx.x + t * NT;
    if (e < K * HH) { const int j = e / K, i = e - j * K; Wl[j * PWM + i] = r.v[t]; }
  }
  for (int e = threadIdx.x; e < HH * (PWM - K); e += NT) { const int j = e / (PWM - K); Wl[j * PWM + K + (e - j * (PWM - K))] = 0.f; }
  if (threadIdx.x < HH) bl[threadIdx.x] = r.b;
}
// acc[ct] += A[16 rows of this wave][k] * W[ct * 16 + n][k] over KSTEPS steps of 4; A k-major in LDS (pitch PRM_), W as stored (pitch PWM)
template <int KSTEPS>
__device__ __forceinline__ void mma16(m16_t (&acc)[2], const float* A, const float* W, int wave, int li, int lq) {
#pragma unroll
  for (int st = 0; st < KSTEPS; ++st) {
    const float av = A[(4 * st + lq) * PRM_ + wave * 16 + li];
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, W[li * PWM + 4 * st + lq], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, W[(16 + li) * PWM + 4 * st + lq], acc[1], 0, 0, 0);
  }
}
__device__ __forceinline__ void bn_finish_m(SmemM& s, int B, float eps, float momentum, float* save, float* rmean, float* rvar, int64_t* nbt,
                                            float rm_old, float rv_old) {
  __syncthreads();                                   // s.fin and the LDS images of the burst are complete
  if (threadIdx.x < HH) {
    double sm = 0.0, q = 0.0;
#pragma unroll
    for (int p = 0; p < NPART; ++p) { sm += s.fin[p][0][threadIdx.x]; q += s.fin[p][1][threadIdx.x]; }
    const double mean = sm / B;
    double var = q / B - mean * mean;
    if (var < 0.0) var = 0.0;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    s.mean[threadIdx.x] = (float)mean; s.inv[threadIdx.x] = inv;
    if (blockIdx.x == 0) {
      save[threadIdx.x] = (float)mean; save[HH + threadIdx.x] = inv;
      if (rmean) {
        const double unb = B > 1 ? var * (double)B / (double)(B - 1) : var;
        rmean[threadIdx.x] = (float)((1.0 - momentum) * rm_old + momentum * mean);
        rvar[threadIdx.x] = (float)((1.0 - momentum) * rv_old + momentum * unb);
        if (threadIdx.x == 0 && nbt) nbt[0] += 1;
      }
    }
  }
  __syncthreads();
}
// column sums of v and v*v over the block's 64 rows from the accumulator layout (lane: rows 4 lq + r of its wave, column ct * 16 + li):
// in-lane over r, across lq at distances 16 and 32, the four waves through LDS in wave order
__device__ __forceinline__ void colsums_m(SmemM& s, const m16_t (&v)[2], const bool (&ok)[4], int wave, int li, int lq, float* part) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float x = ok[r] ? v[ct][r] : 0.f; s1 += x; s2 += x * x; }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lq == 0) { s.ws[wave][0][ct * 16 + li] = s1; s.ws[wave][1][ct * 16 + li] = s2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * HH) {
    const int st = threadIdx.x >> 5, c = threadIdx.x & (HH - 1);
    part[st * HH + c] = (s.ws[0][st][c] + s.ws[1][st][c]) + (s.ws[2][st][c] + s.ws[3][st][c]);
  }
}

// entry segment: inp = (x, onehot, mask); h0 = relu(fc_in(inp)); z1_0 = fc1_0(h0), partial statistics — on the matrix cores like the
// other segments (K = 38 padded to 40 for fc_in)
constexpr int MK_IN = 40, PWI = 44;         // padded input width, fc_in's weight row pitch in LDS
struct alignas(16) SmemF {
  float Wi[HH * PWI], Wf[HH * PWM], bi[HH], bf[HH];
  float IN[MK_IN * PRM_];
  float A[HH * PRM_];
  float ws[NQ][2][HH];
};
__global__ void __launch_bounds__(NT) g_fwd_first4_kernel(const float* __restrict__ PRM, GBufs a, GDesc d) {
  __shared__ SmemF s;
  const int lane = threadIdx.x & (FT - 1), wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t rowL = (size_t)blockIdx.x * FT + lane;
  const bool onL = rowL < (size_t)a.B;
  const size_t rowA = (size_t)blockIdx.x * FT + wave * 16 + 4 * lq;
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ok[r] = rowA + r < (size_t)a.B;
  WRegs<MAXIN> w_in; WRegs<HH> w_fc1;
  wload<MAXIN>(w_in, PRM + d.fc_in_w, PRM + d.fc_in_b);
  wload<HH>(w_fc1, PRM + d.fc1_w[0], PRM + d.fc1_b[0]);
  float in[(MAXIN + NQ - 1) / NQ];                   // wave q brings in inputs q, q+4, ... of its rows (lane = row)
  const size_t rc = min(rowL, (size_t)a.B - 1);
#pragma unroll
  for (int t = 0; t < (MAXIN + NQ - 1) / NQ; ++t) {
    const int i = min(wave + t * NQ, MAXIN - 1);
    in[t] = i < DIN ? a.x[rc * DIN + i] : (i < DIN + NCLS ? a.onehot[rc * NCLS + (i - DIN)] : a.mask[rc * DIN + (i - DIN - NCLS)]);
  }
#pragma unroll
  for (int t = 0; t < (MAXIN * HH + NT - 1) / NT; ++t) {
    const int e = threadIdx.x + t * NT;
    if (e < MAXIN * HH) { const int j = e / MAXIN, i = e - j * MAXIN; s.Wi[j * PWI + i] = w_in.v[t]; }
  }
  for (int e = threadIdx.x; e < HH * (PWI - MAXIN); e += NT) { const int j = e / (PWI - MAXIN); s.Wi[j * PWI + MAXIN + (e - j * (PWI - MAXIN))] = 0.f; }
  if (threadIdx.x < HH) s.bi[threadIdx.x] = w_in.b;
  wstore_m<HH>(s.Wf, s.bf, w_fc1);
#pragma unroll
  for (int t = 0; t < (MAXIN + NQ - 1) / NQ; ++t) {
    const int i = wave + t * NQ;
    if (i < MAXIN) { const float v = onL ? in[t] : 0.f; s.IN[i * PRM_ + lane] = v; if (onL) a.inp[rowL * MAXIN + i] = v; }
  }
  if (wave < MK_IN - MAXIN) s.IN[(MAXIN + wave) * PRM_ + lane] = 0.f;
  __syncthreads();
  m16_t h[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[ct][r] = s.bi[ct * 16 + li];
#pragma unroll
  for (int st = 0; st < MK_IN / 4; ++st) {
    const float av = s.IN[(4 * st + lq) * PRM_ + wave * 16 + li];
    h[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s.Wi[li * PWI + 4 * st + lq], h[0], 0, 0, 0);
    h[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s.Wi[(16 + li) * PWI + 4 * st + lq], h[1], 0, 0, 0);
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h[ct][r] = h[ct][r] > 0.f ? h[ct][r] : 0.f;
      if (ok[r]) a.H[(rowA + r) * HH + ct * 16 + li] = h[ct][r];
      s.A[(ct * 16 + li) * PRM_ + wave * 16 + 4 * lq + r] = h[ct][r];
    }
  __syncthreads();
  m16_t z[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) z[ct][r] = s.bf[ct * 16 + li];
  mma16<HH / 4>(z, s.A, s.Wf, wave, li, lq);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r]) a.Z1[(rowA + r) * HH + ct * 16 + li] = z[ct][r];
  // column sums (as colsums_m, on this kernel's own LDS block)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float x = ok[r] ? z[ct][r] : 0.f; s1 += x; s2 += x * x; }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lq == 0) { s.ws[wave][0][ct * 16 + li] = s1; s.ws[wave][1][ct * 16 + li] = s2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * HH) {
    const int st = threadIdx.x >> 5, c = threadIdx.x & (HH - 1);
    a.P[(size_t)blockIdx.x * 2 * HH + st * HH + c] = (s.ws[0][st][c] + s.ws[1][st][c]) + (s.ws[2][st][c] + s.ws[3][st][c]);
  }
}

// kind A (block k): bn1 statistics -> a1 = relu(film(bn1(z1))) ; z2 = fc2(a1), partial statistics
__global__ void __launch_bounds__(NT) g_fwd_a4_kernel(const float* __restrict__ PRM, GBufs a, FSeg f) {
  __shared__ SmemM s;
  const int lyr = f.li, k = f.k;
  const int lane = threadIdx.x & (FT - 1), wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t rowL = (size_t)blockIdx.x * FT + lane;            // the lane-per-row view (cond loads)
  const size_t rowA = (size_t)blockIdx.x * FT + wave * 16 + 4 * lq;   // this lane's first accumulator row
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ok[r] = rowA + r < (size_t)a.B;
  // ---- the burst
  WRegs<MAXCOND> w_g, w_b; WRegs<HH> w_fc; CondRegs cr; PRegs pr;
  float z[2][4], g_bn = 0.f, b_bn = 0.f, rm_old = 0.f, rv_old = 0.f;
  PCG_T(0);
  wload<MAXCOND>(w_g, PRM + f.fg_w, PRM + f.fg_b);
  wload<MAXCOND>(w_b, PRM + f.fb_w, PRM + f.fb_b);
  wload<HH>(w_fc, PRM + f.fc_w, PRM + f.fc_b);
  cload(cr, a.onehot, a.mask, min(rowL, (size_t)a.B - 1), rowL < (size_t)a.B, wave);
  const float* Zk = a.Z1 + (size_t)k * a.B * HH;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) z[ct][r] = Zk[min(rowA + r, (size_t)a.B - 1) * HH + ct * 16 + li];
  if (threadIdx.x < HH) {
    g_bn = PRM[f.bn_g + threadIdx.x]; b_bn = PRM[f.bn_b + threadIdx.x];
    if (f.rmean) { rm_old = f.rmean[threadIdx.x]; rv_old = f.rvar[threadIdx.x]; }     // every block: no block-0 detour in the burst
  }
  pload(pr, a.P + (size_t)lyr * a.nblocks * 2 * HH, a.nblocks);
  PCG_T(1);
  // ---- into LDS
  wstore_m<MAXCOND>(s.Wg, s.bg, w_g);
  wstore_m<MAXCOND>(s.Wb, s.bb, w_b);
  wstore_m<HH>(s.Wf, s.bf, w_fc);
#pragma unroll
  for (int t = 0; t < (MAXCOND + NQ - 1) / NQ; ++t) { const int i = wave + t * NQ; if (i < MAXCOND) s.C[i * PRM_ + lane] = cr.v[t]; }
  if (wave < MK_C - MAXCOND) s.C[(MAXCOND + wave) * PRM_ + lane] = 0.f;
  if (threadIdx.x < HH) { s.gamma[threadIdx.x] = g_bn; s.beta[threadIdx.x] = b_bn; }
  { const int c = threadIdx.x & (HH - 1), part = threadIdx.x >> 5; s.fin[part][0][c] = pr.s; s.fin[part][1][c] = pr.q; }
  PCG_T(2);
  bn_finish_m(s, a.B, a.eps, a.momentum, a.SM + (size_t)lyr * 2 * HH, f.rmean, f.rvar, f.nbt, rm_old, rv_old);
  PCG_T(3);
  // FiLM: gamma and beta of this wave's 16 rows x 32 channels
  m16_t gam[2], bet[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gam[ct][r] = s.bg[ct * 16 + li]; bet[ct][r] = s.bb[ct * 16 + li]; }
  mma16<MK_C / 4>(gam, s.C, s.Wg, wave, li, lq);
  mma16<MK_C / 4>(bet, s.C, s.Wb, wave, li, lq);
  PCG_T(4);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = ct * 16 + li;
    const float mu = s.mean[c], iv = s.inv[c], ga = s.gamma[c], be = s.beta[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float n = fmaf((z[ct][r] - mu) * iv, ga, be);
      const float fv = fmaf(gam[ct][r], n, bet[ct][r]);
      s.A[c * PRM_ + wave * 16 + 4 * lq + r] = fv > 0.f ? fv : 0.f;
    }
  }
  __syncthreads();                                   // a1 parked (each wave reads back only its own rows; the barrier orders the LDS traffic)
  PCG_T(5);
  m16_t z2[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) z2[ct][r] = s.bf[ct * 16 + li];
  mma16<HH / 4>(z2, s.A, s.Wf, wave, li, lq);
  PCG_T(6);
  float* Z2k = a.Z2 + (size_t)k * a.B * HH;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r]) Z2k[(rowA + r) * HH + ct * 16 + li] = z2[ct][r];
  colsums_m(s, z2, ok, wave, li, lq, a.P + ((size_t)(lyr + 1) * a.nblocks + blockIdx.x) * 2 * HH);
  PCG_T(7);
}

// kind B (block k): bn2 statistics -> h_{k+1} = h_k + film(bn2(z2)); z1_{k+1} = fc1_{k+1}(h), partial statistics (after the last
// block the output heads follow instead: g_heads4_kernel)
__global__ void __launch_bounds__(NT) g_fwd_b4_kernel(const float* __restrict__ PRM, GBufs a, FSeg f) {
  __shared__ SmemM s;
  const int lyr = f.li, k = f.k;
  const int lane = threadIdx.x & (FT - 1), wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t rowL = (size_t)blockIdx.x * FT + lane;
  const size_t rowA = (size_t)blockIdx.x * FT + wave * 16 + 4 * lq;
  const bool more = f.more != 0;                     // kernel-uniform: a next block follows (f.fc_* = its fc1)
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ok[r] = rowA + r < (size_t)a.B;
  WRegs<MAXCOND> w_g, w_b; WRegs<HH> w_fc; CondRegs cr; PRegs pr;
  float z[2][4], h[2][4], g_bn = 0.f, b_bn = 0.f, rm_old = 0.f, rv_old = 0.f;
  wload<MAXCOND>(w_g, PRM + f.fg_w, PRM + f.fg_b);
  wload<MAXCOND>(w_b, PRM + f.fb_w, PRM + f.fb_b);
  if (more) wload<HH>(w_fc, PRM + f.fc_w, PRM + f.fc_b);
  cload(cr, a.onehot, a.mask, min(rowL, (size_t)a.B - 1), rowL < (size_t)a.B, wave);
  const float* Zk = a.Z2 + (size_t)k * a.B * HH;
  const float* Hk = a.H + (size_t)k * a.B * HH;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = min(rowA + r, (size_t)a.B - 1) * HH + ct * 16 + li;
      z[ct][r] = Zk[o]; h[ct][r] = Hk[o];
    }
  if (threadIdx.x < HH) {
    g_bn = PRM[f.bn_g + threadIdx.x]; b_bn = PRM[f.bn_b + threadIdx.x];
    if (f.rmean) { rm_old = f.rmean[threadIdx.x]; rv_old = f.rvar[threadIdx.x]; }
  }
  pload(pr, a.P + (size_t)lyr * a.nblocks * 2 * HH, a.nblocks);
  wstore_m<MAXCOND>(s.Wg, s.bg, w_g);
  wstore_m<MAXCOND>(s.Wb, s.bb, w_b);
  if (more) wstore_m<HH>(s.Wf, s.bf, w_fc);
#pragma unroll
  for (int t = 0; t < (MAXCOND + NQ - 1) / NQ; ++t) { const int i = wave + t * NQ; if (i < MAXCOND) s.C[i * PRM_ + lane] = cr.v[t]; }
  if (wave < MK_C - MAXCOND) s.C[(MAXCOND + wave) * PRM_ + lane] = 0.f;
  if (threadIdx.x < HH) { s.gamma[threadIdx.x] = g_bn; s.beta[threadIdx.x] = b_bn; }
  { const int c = threadIdx.x & (HH - 1), part = threadIdx.x >> 5; s.fin[part][0][c] = pr.s; s.fin[part][1][c] = pr.q; }
  bn_finish_m(s, a.B, a.eps, a.momentum, a.SM + (size_t)lyr * 2 * HH, f.rmean, f.rvar, f.nbt, rm_old, rv_old);
  m16_t gam[2], bet[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gam[ct][r] = s.bg[ct * 16 + li]; bet[ct][r] = s.bb[ct * 16 + li]; }
  mma16<MK_C / 4>(gam, s.C, s.Wg, wave, li, lq);
  mma16<MK_C / 4>(bet, s.C, s.Wb, wave, li, lq);
  float* Hn = a.H + (size_t)(k + 1) * a.B * HH;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = ct * 16 + li;
    const float mu = s.mean[c], iv = s.inv[c], ga = s.gamma[c], be = s.beta[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float n = fmaf((z[ct][r] - mu) * iv, ga, be);
      const float hv = h[ct][r] + fmaf(gam[ct][r], n, bet[ct][r]);
      if (ok[r]) Hn[(rowA + r) * HH + c] = hv;
      if (more) s.A[c * PRM_ + wave * 16 + 4 * lq + r] = hv;
    }
  }
  if (!more) return;
  __syncthreads();
  m16_t z1[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) z1[ct][r] = s.bf[ct * 16 + li];
  mma16<HH / 4>(z1, s.A, s.Wf, wave, li, lq);
  float* Z1n = a.Z1 + (size_t)(k + 1) * a.B * HH;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r]) Z1n[(rowA + r) * HH + ct * 16 + li] = z1[ct][r];
  colsums_m(s, z1, ok, wave, li, lq, a.P + ((size_t)(lyr + 1) * a.nblocks + blockIdx.x) * 2 * HH);
}

// ---- output heads ----------------------------------------------------------------------------------------------------------------
// The continuous residual head and the categorical heads as ONE product on the matrix cores — out[rows][T + ncont columns] =
// h_5 W^T + b over the packed weight rows of all heads — followed by the per-(row, head) Gumbel-softmax.  A block owns HR = 16 rows
// (4096 rows = 256 blocks: the 64-row form of this kernel kept 64 of the 256 CUs busy for 19 us — 3.5 us of products, 5 us of
// softmax rounds, 3.5 us of tile stores per block — where a 16-row block does a quarter of each):
//   * the weight rows are staged as stored ([column][32], pitch 36: B[k][n] = W[n][k], banks 4 li + lq); h_5 comes in k-major
//     ([k][16 rows]); logits, noise and the backward's cotangents sit in [16][T] tiles of pitch 100 (36 mod 64: a 16-row column
//     walk of the MFMA's A operand is conflict-free);
//   * wave q owns the 16-column tiles q and q + 4 (v_mfma_f32_16x16x4_f32, 8 steps per tile);
//   * the softmax of a (row, head) is shared by TWO lanes (the head's columns split in two, max and sum exchanged at distance 32):
//     16 rows x 7 heads x 2 lanes fit the block's 256 threads in one round, the widest head (30 columns) costs 15 column visits.
// (The first version — a lane per row walking its heads, weights as LDS broadcasts — was bound by LDS bandwidth in the dot products
// and by the 30-wide head's serial softmax on one wave: 24 us per launch.)
constexpr int HR = 16;                      // rows per block of the two head kernels
constexpr int MAXCOLS = 96;                 // packed categorical + continuous columns (host-checked)
constexpr int TP = 100;                     // row pitch of the [HR][T] LDS tiles
struct alignas(16) SmemHeads {
  float W[MAXCOLS * PWM];                   // head weight rows as stored; the backward reads the same image as B[k = column][n = input]
  float b[MAXCOLS];
  float Hk[HH * HR];                        // forward: h_5 k-major
  float lg[HR * TP];                        // logits; backward: d_logits in, dl out (+ the continuous head's gradient in columns T ..)
  float ns[HR * TP];                        // noise -> (logit + noise) / tau -> soft sample; backward: soft
  float ds[HR * TP];                        // backward: d_samples
  float ct[HR * (HH + 1)];                  // continuous head
  float Wg[HH * PWM], bg[HH];               // backward: FiLM gamma of the last block (part a)
  float C[MK_C * HR];                       // backward: cond, k-major
  float sm[2][HH];                          // backward: saved mean / invstd of the last bn2
  int seg[MAXHEADS + 1];
};
// flat-parameter offsets of output column c (categorical columns 0 .. T-1, then the continuous head); -1 past the last column
struct ColSrc { int w, b; };
__device__ __forceinline__ ColSrc col_src(const GDesc& d, int c, int T) {
  ColSrc r;
  if (c >= T) { const bool ok = c < T + d.ncont; r.w = ok ? d.cont_w + (c - T) * HH : -1; r.b = ok ? d.cont_b + (c - T) : -1; return r; }
  int w = d.head_w[0], b = d.head_b[0], c0 = 0;
#pragma unroll
  for (int k = 1; k < MAXHEADS; ++k)
    if (k < d.nheads && c >= d.seg[k]) { w = d.head_w[k]; b = d.head_b[k]; c0 = d.seg[k]; }
  r.w = w + (c - c0) * HH; r.b = b + (c - c0);
  return r;
}
constexpr int HW_PER = MAXCOLS * HH / NT;   // 12 weight elements per thread
constexpr int TILE_PER = (HR * MAXT + NT - 1) / NT;   // 6 tile elements per thread
// coalesced [rows][T] global tile <-> LDS tile [HR][TP]; element e = tid + 256 t sits at (e / T, e % T), walked incrementally (a
// division by the runtime T per element was a sixth of the kernel)
struct TileWalk { int rr, cc, dr, dc; };
__device__ __forceinline__ TileWalk tile_walk(int T) {
  TileWalk w; w.rr = (int)threadIdx.x / T; w.cc = (int)threadIdx.x - w.rr * T; w.dr = NT / T; w.dc = NT - w.dr * T; return w;
}
__device__ __forceinline__ void tile_step(TileWalk& w, int T) { w.rr += w.dr; w.cc += w.dc; if (w.cc >= T) { w.cc -= T; w.rr += 1; } }
__device__ __forceinline__ void tile_load(float (&r)[TILE_PER], const float* __restrict__ g, size_t row0, int rows, int T) {
#pragma unroll
  for (int t = 0; t < TILE_PER; ++t) r[t] = g ? g[row0 * T + min((int)threadIdx.x + t * NT, rows * T - 1)] : 0.f;
}
__device__ __forceinline__ void tile_park(float* L, const float (&r)[TILE_PER], int rows, int T) {
  TileWalk w = tile_walk(T);
#pragma unroll
  for (int t = 0; t < TILE_PER; ++t) { if (w.rr < rows) L[w.rr * TP + w.cc] = r[t]; tile_step(w, T); }
}
__device__ __forceinline__ void tile_out(float* __restrict__ g, const float* L, size_t row0, int rows, int T) {
  TileWalk w = tile_walk(T);
#pragma unroll
  for (int t = 0; t < TILE_PER; ++t) { if (w.rr < rows) g[row0 * T + threadIdx.x + t * NT] = L[w.rr * TP + w.cc]; tile_step(w, T); }
}
// the packed weight rows (and biases) of all heads: global -> registers -> LDS [column][36]
__device__ __forceinline__ void heads_wload(float (&wr)[HW_PER], float& br, const float* __restrict__ PRM, const GDesc& d, int T) {
#pragma unroll
  for (int t = 0; t < HW_PER; ++t) {
    const int e = threadIdx.x + t * NT;
    const ColSrc cs = col_src(d, e >> 5, T);
    const float v = PRM[max(cs.w, 0) + (e & 31)];
    wr[t] = cs.w >= 0 ? v : 0.f;
  }
  br = 0.f;
  if (threadIdx.x < MAXCOLS) { const ColSrc cs = col_src(d, threadIdx.x, T); const float v = PRM[max(cs.b, 0)]; br = cs.b >= 0 ? v : 0.f; }
}
__device__ __forceinline__ void heads_wstore(SmemHeads& s, const float (&wr)[HW_PER], float br, const GDesc& d) {
#pragma unroll
  for (int t = 0; t < HW_PER; ++t) { const int e = threadIdx.x + t * NT; s.W[(e >> 5) * PWM + (e & 31)] = wr[t]; }
  if (threadIdx.x < MAXCOLS) s.b[threadIdx.x] = br;
  if (threadIdx.x <= MAXHEADS) s.seg[threadIdx.x] = d.seg[min((int)threadIdx.x, d.nheads)];
}
// the (row, head) pair a lane works on in a softmax round: pair = base + lane % 32 -> head = pair / 16, row = pair % 16; lanes l and
// l + 32 share the pair and take the lower / upper half of the head's columns [cb, ce).  Past the last pair (or past the batch): no columns.
struct PairCols { int m, cb, ce; };
__device__ __forceinline__ PairCols pair_cols(const SmemHeads& s, int base, int lane, int nheads, int rows) {
  const int p = base + (lane & 31), hd = min(p >> 4, max(nheads - 1, 0)), lh = lane >> 5;
  PairCols r; r.m = p & (HR - 1);
  const int c0 = s.seg[hd], c1 = s.seg[hd + 1], cm = c0 + (c1 - c0 + 1) / 2;
  const bool live = p < HR * nheads && r.m < rows;
  r.cb = live ? (lh ? cm : c0) : 0; r.ce = live ? (lh ? c1 : cm) : 0;
  return r;
}

__global__ void __launch_bounds__(NT) g_heads16_kernel(const float* __restrict__ PRM, GBufs a, GDesc d) {
  __shared__ SmemHeads s;
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t row0 = (size_t)blockIdx.x * HR;
  const int rows = min(HR, a.B - (int)row0);
  const int T = d.seg[d.nheads], ncols = T + d.ncont;
  // ---- the burst: weight rows, biases, the noise tile, h_5 (element tid + 256 t of the block's [16][32] slice)
  float wr[HW_PER], br, nr[TILE_PER], hv[HR * HH / NT];
  PCG_T(8);
  heads_wload(wr, br, PRM, d, T);
  tile_load(nr, a.noise, row0, rows, T);
  const float* h5 = a.H + (size_t)NBLK * a.B * HH + row0 * HH;
#pragma unroll
  for (int t = 0; t < HR * HH / NT; ++t) hv[t] = h5[min((int)threadIdx.x + t * NT, rows * HH - 1)];
  PCG_T(9);
  heads_wstore(s, wr, br, d);
  tile_park(s.ns, nr, rows, T);
#pragma unroll
  for (int t = 0; t < HR * HH / NT; ++t) { const int e = threadIdx.x + t * NT; s.Hk[(e & 31) * HR + (e >> 5)] = hv[t]; }
  __syncthreads();
  PCG_T(10);
  // ---- out = h_5 W^T + b: wave q owns the 16-column tiles q, q + 4
  const float inv_tau = 1.f / a.tau;
  for (int ct = q; ct * 16 < ncols; ct += NQ) {      // wave-uniform
    const int col = ct * 16 + li;
    const float bv = s.b[col];
    m16_t acc = {bv, bv, bv, bv};
#pragma unroll
    for (int st = 0; st < HH / 4; ++st)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s.Hk[(4 * st + lq) * HR + li], s.W[col * PWM + 4 * st + lq], acc, 0, 0, 0);
    const int colc = min(col, max(T - 1, 0));
    float nz[4];                                       // the noise under this lane's outputs: all reads first, then the writes
#pragma unroll
    for (int r = 0; r < 4; ++r) nz[r] = s.ns[(4 * lq + r) * TP + colc];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * lq + r;
      const float v = acc[r];
      if (col < T) { s.lg[m * TP + col] = v; s.ns[m * TP + col] = (v + nz[r]) * inv_tau; }
      else if (col < ncols) s.ct[m * (HH + 1) + (col - T)] = v * a.res_scale;
    }
  }
  __syncthreads();
  PCG_T(11);
  // ---- Gumbel-softmax per (row, head): two lanes per pair, 128 pairs per round
  for (int base = q * 32; base < HR * d.nheads; base += NQ * 32) {     // wave-uniform
    const PairCols pc = pair_cols(s, base, lane, d.nheads, rows);
    const int cb = pc.cb, ce = pc.ce, m = pc.m;
    float* ns = s.ns + m * TP;
    // this lane's columns (at most 16) come into registers once, independent LDS reads
    constexpr int MAXHALF = 16;
    float tv[MAXHALF];
#pragma unroll
    for (int j = 0; j < MAXHALF; ++j) tv[j] = cb + j < ce ? ns[cb + j] : -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXHALF; ++j) mx = fmaxf(mx, tv[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < MAXHALF; ++j) { tv[j] = cb + j < ce ? expf(tv[j] - mx) : 0.f; se += tv[j]; }
    se += __shfl_xor(se, 32);
    const float inv = 1.f / se;
    float best = -1.f; int arg = cb;
#pragma unroll
    for (int j = 0; j < MAXHALF; ++j) {
      if (cb + j < ce) {
        const float p = tv[j] * inv;
        ns[cb + j] = p;
        if (p > best) { best = p; arg = cb + j; }
      }
    }
    if (a.hard) {                                    // kernel-uniform; first maximum, as y_soft.max(dim)[1]
      const float ob = __shfl_xor(best, 32); const int oa = __shfl_xor(arg, 32);
      if (ob > best || (ob == best && oa < arg)) arg = oa;
      if (m < rows) for (int c = cb; c < ce; ++c) a.hard[(row0 + m) * T + c] = c == arg ? 1.f : 0.f;
    }
  }
  PCG_T(12);
  __syncthreads();
  PCG_T(13);
  tile_out(a.logits, s.lg, row0, rows, T);
  tile_out(a.soft, s.ns, row0, rows, T);
  for (int e = threadIdx.x; e < rows * d.ncont; e += NT) { const int rr = e / d.ncont; a.cont[row0 * d.ncont + e] = s.ct[rr * (HH + 1) + (e - rr * d.ncont)]; }
  PCG_T(14);
}

// ---- backward ------------------------------------------------------------------------------------------------------------------
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

// ---- backward segments on the matrix cores (same ownership as the forward: wave w has rows 16 w .. 16 w + 15, both channel tiles) ---
struct alignas(16) SmemMB {
  float aff[4][HH];                         // bn2 gamma, bn2 beta, bn1 gamma, bn1 beta
  float sm[4][HH];                          // saved mean / invstd: [0..1] bn1 of the block, [2..3] bn2 (kind C: bn2 of block k - 1)
  float sums[2 * HH];                       // mean(dz), mean(dz * xhat) of the BatchNorm in flight
  float Wg[HH * PWM], Wb[HH * PWM], Wf[HH * PWM];
  float bg[HH], bb[HH];
  float C[MK_C * PRM_];
  float A[HH * PRM_];
  float ws[NQ][2][HH];
  double fin[NPART][2][HH];
};
// acc[ct][row][i = ct * 16 + n] += D[row][j] * W[j][i] over j = 0 .. 31: D k-major (index j) in LDS, W as stored [j][i]
__device__ __forceinline__ void mma16_t(m16_t (&acc)[2], const float* D, const float* W, int wave, int li, int lq) {
#pragma unroll
  for (int st = 0; st < HH / 4; ++st) {
    const float dv = D[(4 * st + lq) * PRM_ + wave * 16 + li];
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, W[(4 * st + lq) * PWM + li], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, W[(4 * st + lq) * PWM + 16 + li], acc[1], 0, 0, 0);
  }
}
__device__ __forceinline__ void mma16_c(m16_t (&acc)[2], const float* C, const float* W, int wave, int li, int lq) {    // cond x FiLM weights
#pragma unroll
  for (int st = 0; st < MK_C / 4; ++st) {
    const float av = C[(4 * st + lq) * PRM_ + wave * 16 + li];
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, W[li * PWM + 4 * st + lq], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, W[(16 + li) * PWM + 4 * st + lq], acc[1], 0, 0, 0);
  }
}
// sums of one BatchNorm backward from the partials parked in s.fin -> s.sums (means); block 0 writes dgamma / dbeta
__device__ __forceinline__ void bnb_finish_m(SmemMB& s, const GBwd& a, int g_off, int b_off, float gg_old, float gb_old) {
  __syncthreads();
  if (threadIdx.x < HH) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int p = 0; p < NPART; ++p) { s1 += s.fin[p][0][threadIdx.x]; s2 += s.fin[p][1][threadIdx.x]; }
    s.sums[threadIdx.x] = (float)(s1 / a.B); s.sums[HH + threadIdx.x] = (float)(s2 / a.B);
    if (blockIdx.x == 0) {
      a.grads[g_off + threadIdx.x] = a.accumulate ? gg_old + (float)s2 : (float)s2;
      a.grads[b_off + threadIdx.x] = a.accumulate ? gb_old + (float)s1 : (float)s1;
    }
  }
  __syncthreads();
}
// column sums of v and w over the block's 64 rows from the accumulator layout -> part[2][HH] (callers zero the rows past the batch)
__device__ __forceinline__ void colsums2_m(SmemMB& s, const m16_t (&v)[2], const m16_t (&w)[2], int wave, int li, int lq, float* part) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float s1 = (v[ct][0] + v[ct][1]) + (v[ct][2] + v[ct][3]), s2 = (w[ct][0] + w[ct][1]) + (w[ct][2] + w[ct][3]);
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lq == 0) { s.ws[wave][0][ct * 16 + li] = s1; s.ws[wave][1][ct * 16 + li] = s2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * HH) {
    const int st = threadIdx.x >> 5, c = threadIdx.x & (HH - 1);
    part[st * HH + c] = (s.ws[0][st][c] + s.ws[1][st][c]) + (s.ws[2][st][c] + s.ws[3][st][c]);
  }
}
template <int K>
__device__ __forceinline__ void wstore_mb(float* Wl, float* bl, const WRegs<K>& r) {
#pragma unroll
  for (int t = 0; t < (K * HH + NT - 1) / NT; ++t) {
    const int e = threadIdx.x + t * NT;
    if (e < K * HH) { const int j = e / K, i = e - j * K; Wl[j * PWM + i] = r.v[t]; }
  }
  for (int e = threadIdx.x; e < HH * (PWM - K); e += NT) { const int j = e / (PWM - K); Wl[j * PWM + K + (e - j * (PWM - K))] = 0.f; }
  if (bl && threadIdx.x < HH) bl[threadIdx.x] = r.b;
}
__device__ __forceinline__ void acc_rows_load(float (&v)[2][4], const float* __restrict__ p, size_t rowA, int B, int li) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[ct][r] = p[min(rowA + r, (size_t)B - 1) * HH + ct * 16 + li];
}
__device__ __forceinline__ void acc_rows_store(float* __restrict__ p, const m16_t (&v)[2], size_t rowA, const bool (&ok)[4], int li) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r]) p[(rowA + r) * HH + ct * 16 + li] = v[ct][r];
}

// first backward kernel (16 rows per block, like the forward's head kernel): the Gumbel-softmax backward of every (row, head) — two
// lanes per pair —, then dh = G W on the matrix cores (G: the rows' gradients at all T + ncont output columns, K = 96; W: the packed
// weight rows as stored: the gradient entering the last block; waves 0 / 1 own one 16-channel tile each), then part "a" of that
// block: dn2 = dh * gam (FiLM gamma product on the matrix cores), partial sums (dn2, dn2 * xhat2) per 16-ROW block — the last
// bn2's slot of Q holds ceil(B / 16) partial rows (BSeg.qrows)
__global__ void __launch_bounds__(NT) g_bwd_first16_kernel(const float* __restrict__ PRM, GBwd a, GDesc d) {
  __shared__ SmemHeads s;
  const int lane = threadIdx.x & (FT - 1), q = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t row0 = (size_t)blockIdx.x * HR, rowA = row0 + 4 * lq;
  const int rows = min(HR, a.B - (int)row0);
  const int T = d.seg[d.nheads];
  constexpr int k = NBLK - 1;
  const int cch = (q & 1) * 16 + li;                 // this lane's channel in the dh product (waves 0 / 1)
  // ---- the burst: head weight rows, the three [rows][T] tiles, d_cont; for part a: FiLM gamma of the last block, cond, z2, saved statistics
  float wr[HW_PER], br, t_dl[TILE_PER], t_ds[TILE_PER], t_y[TILE_PER], z[4], smr = 0.f;
  WRegs<MAXCOND> w_g;
  constexpr int DC_PER = (HR * HH + NT - 1) / NT, C_PER = (MK_C * HR + NT - 1) / NT;
  float dcr[DC_PER], cv[C_PER];
  heads_wload(wr, br, PRM, d, T);
  tile_load(t_dl, a.d_logits, row0, rows, T);
  tile_load(t_ds, a.d_samples, row0, rows, T);
  tile_load(t_y, a.soft, row0, rows, T);
#pragma unroll
  for (int t = 0; t < DC_PER; ++t) dcr[t] = a.d_cont ? a.d_cont[row0 * d.ncont + min((int)threadIdx.x + t * NT, max(rows * d.ncont - 1, 0))] : 0.f;
  wload<MAXCOND>(w_g, PRM + d.fg_w[k], PRM + d.fg_b[k]);
#pragma unroll
  for (int t = 0; t < C_PER; ++t) {                    // cond = (one-hot target, mask), element (i, row) = (e / 16, e % 16)
    const int e = threadIdx.x + t * NT, i = min(e >> 4, MAXCOND - 1);
    const size_t rw = min(row0 + (e & (HR - 1)), (size_t)a.B - 1);
    cv[t] = i < NCLS ? a.onehot[rw * NCLS + i] : a.mask[rw * DIN + (i - NCLS)];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) z[r] = a.Z2[(size_t)k * a.B * HH + min(rowA + r, (size_t)a.B - 1) * HH + cch];
  if (threadIdx.x < 2 * HH) smr = a.SM[(size_t)(2 * k + 1) * 2 * HH + threadIdx.x];
  // ---- into LDS.  G (in s.lg) must be finite and zero outside the live rows / columns: it is an MFMA operand
  heads_wstore(s, wr, br, d);
  for (int e = threadIdx.x; e < HR * TP; e += NT) s.lg[e] = 0.f;
  __syncthreads();
  tile_park(s.lg, t_dl, rows, T);
  tile_park(s.ds, t_ds, rows, T);
  tile_park(s.ns, t_y, rows, T);
#pragma unroll
  for (int t = 0; t < DC_PER; ++t) {
    const int e = threadIdx.x + t * NT;
    if (e < rows * d.ncont) {
      const int rr = e / d.ncont;
      const float dc = dcr[t] * a.res_scale;           // 0 without a cotangent
      s.lg[rr * TP + T + (e - rr * d.ncont)] = dc;     // the continuous head's columns of G
      a.DC[row0 * d.ncont + e] = dc;
    }
  }
  wstore_mb<MAXCOND>(s.Wg, s.bg, w_g);
#pragma unroll
  for (int t = 0; t < C_PER; ++t) { const int e = threadIdx.x + t * NT; if (e < MK_C * HR) s.C[e] = e < MAXCOND * HR ? cv[t] : 0.f; }
  if (threadIdx.x < 2 * HH) s.sm[threadIdx.x >> 5][threadIdx.x & (HH - 1)] = smr;
  __syncthreads();
  // ---- dl of every (row, head), left in G
  {
    const float inv_tau = 1.f / a.tau;
    for (int base = q * 32; base < HR * d.nheads; base += NQ * 32) {   // wave-uniform
      const PairCols pc = pair_cols(s, base, lane, d.nheads, rows);
      const int cb = pc.cb, ce = pc.ce;
      float* dlr = s.lg + pc.m * TP; const float* dsr = s.ds + pc.m * TP; const float* yr = s.ns + pc.m * TP;
      constexpr int MAXHALF = 16;                      // this lane's columns in registers: independent LDS reads, then the arithmetic
      float dsv[MAXHALF], yv[MAXHALF], dlv[MAXHALF];
#pragma unroll
      for (int j = 0; j < MAXHALF; ++j) {
        const bool in = cb + j < ce;
        dsv[j] = in ? dsr[cb + j] : 0.f; yv[j] = in ? yr[cb + j] : 0.f; dlv[j] = in ? dlr[cb + j] : 0.f;
      }
      float dot = 0.f;
      if (a.d_samples) {
#pragma unroll
        for (int j = 0; j < MAXHALF; ++j) dot = fmaf(dsv[j], yv[j], dot);
        dot += __shfl_xor(dot, 32);
      }
#pragma unroll
      for (int j = 0; j < MAXHALF; ++j) {
        if (cb + j < ce) {
          float dl = a.d_logits ? dlv[j] : 0.f;
          if (a.d_samples) dl += yv[j] * (dsv[j] - dot) * inv_tau;
          dlr[cb + j] = dl;
        }
      }
    }
  }
  __syncthreads();
  if (q < 2) {                                         // wave-uniform
    // ---- dh[16 rows][16 channels of this wave] = G[16][96] W[96][32]
    m16_t dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int st = 0; st < MAXCOLS / 4; ++st)
      dh = __builtin_amdgcn_mfma_f32_16x16x4f32(s.lg[li * TP + 4 * st + lq], s.W[(4 * st + lq) * PWM + cch], dh, 0, 0, 0);
    // ---- part a: gam = cond Wg^T + bg; dn2 = dh * gam; column sums of (dn2, dn2 * xhat2) over the block's rows
    const float bgv = s.bg[cch];
    m16_t gam = {bgv, bgv, bgv, bgv};
#pragma unroll
    for (int st = 0; st < MK_C / 4; ++st)
      gam = __builtin_amdgcn_mfma_f32_16x16x4f32(s.C[(4 * st + lq) * HR + li], s.Wg[cch * PWM + 4 * st + lq], gam, 0, 0, 0);
    const float m2 = s.sm[0][cch], i2 = s.sm[1][cch];
    float v[4], w[4];
    float* DHk = a.DH + (size_t)k * a.B * HH;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = rowA + r < (size_t)a.B;
      if (ok) DHk[(rowA + r) * HH + cch] = dh[r];
      const float xh = (z[r] - m2) * i2;
      v[r] = ok ? dh[r] * gam[r] : 0.f;
      w[r] = v[r] * xh;
    }
    float s1 = (v[0] + v[1]) + (v[2] + v[3]), s2 = (w[0] + w[1]) + (w[2] + w[3]);
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    float* part = a.Q + ((size_t)(2 * k + 1) * a.nblocks + blockIdx.x) * 2 * HH;
    if (lq == 0) { part[cch] = s1; part[HH + cch] = s2; }
  }
  // DL = the first T columns of G
  tile_out(a.DL, s.lg, row0, rows, T);
}

// kind B (block k): bn2 backward -> dz2; through fc2 and the ReLU / FiLM -> dn1 and its partial sums; FiLM output gradients
struct BSeg { int fg_w, fg_b, fb_w, fb_b, fc2_w, bn2_g, bn2_b, bn1_g, bn1_b, k, qrows; };   // qrows: partial rows in bn2's slot of Q
__global__ void __launch_bounds__(NT) g_bwd_b4_kernel(const float* __restrict__ PRM, GBwd a, BSeg f) {
  __shared__ SmemMB s;
  const int k = f.k;
  const int lane = threadIdx.x & (FT - 1), wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t rowL = (size_t)blockIdx.x * FT + lane;
  const size_t rowA = (size_t)blockIdx.x * FT + wave * 16 + 4 * lq;
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ok[r] = rowA + r < (size_t)a.B;
  // ---- the burst
  WRegs<MAXCOND> w_g, w_b; WRegs<HH> w_fc2; CondRegs cr; PRegs pr;
  float dh[2][4], z2[2][4], z1[2][4], affv = 0.f, smr = 0.f, gg_old = 0.f, gb_old = 0.f;
  wload<MAXCOND>(w_g, PRM + f.fg_w, PRM + f.fg_b);
  wload<MAXCOND>(w_b, PRM + f.fb_w, PRM + f.fb_b);
  wload<HH>(w_fc2, PRM + f.fc2_w, nullptr);
  cload(cr, a.onehot, a.mask, min(rowL, (size_t)a.B - 1), rowL < (size_t)a.B, wave);
  acc_rows_load(dh, a.DH + (size_t)k * a.B * HH, rowA, a.B, li);
  acc_rows_load(z2, a.Z2 + (size_t)k * a.B * HH, rowA, a.B, li);
  acc_rows_load(z1, a.Z1 + (size_t)k * a.B * HH, rowA, a.B, li);
  if (threadIdx.x < 4 * HH) {
    const int w = threadIdx.x >> 5, c = threadIdx.x & (HH - 1);
    affv = PRM[(w == 0 ? f.bn2_g : w == 1 ? f.bn2_b : w == 2 ? f.bn1_g : f.bn1_b) + c];
    smr = a.SM[(size_t)(2 * k) * 2 * HH + threadIdx.x];        // sm[0..1] = bn1's mean / invstd, sm[2..3] = bn2's
  }
  if (threadIdx.x < HH && a.accumulate) { gg_old = a.grads[f.bn2_g + threadIdx.x]; gb_old = a.grads[f.bn2_b + threadIdx.x]; }
  pload(pr, a.Q + (size_t)(2 * k + 1) * a.nblocks * 2 * HH, f.qrows);
  // ---- into LDS
  wstore_mb<MAXCOND>(s.Wg, s.bg, w_g);
  wstore_mb<MAXCOND>(s.Wb, s.bb, w_b);
  wstore_mb<HH>(s.Wf, nullptr, w_fc2);
#pragma unroll
  for (int t = 0; t < (MAXCOND + NQ - 1) / NQ; ++t) { const int i = wave + t * NQ; if (i < MAXCOND) s.C[i * PRM_ + lane] = cr.v[t]; }
  if (wave < MK_C - MAXCOND) s.C[(MAXCOND + wave) * PRM_ + lane] = 0.f;
  if (threadIdx.x < 4 * HH) { s.aff[threadIdx.x >> 5][threadIdx.x & (HH - 1)] = affv; s.sm[threadIdx.x >> 5][threadIdx.x & (HH - 1)] = smr; }
  { const int c = threadIdx.x & (HH - 1), part = threadIdx.x >> 5; s.fin[part][0][c] = pr.s; s.fin[part][1][c] = pr.q; }
  bnb_finish_m(s, a, f.bn2_g, f.bn2_b, gg_old, gb_old);
  m16_t gam[2], bet[2], dz2[2], dgam[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gam[ct][r] = s.bg[ct * 16 + li]; bet[ct][r] = s.bb[ct * 16 + li]; }
  mma16_c(gam, s.C, s.Wg, wave, li, lq);
  mma16_c(bet, s.C, s.Wb, wave, li, lq);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = ct * 16 + li;
    const float g2 = s.aff[0][c], b2 = s.aff[1][c], m2 = s.sm[2][c], i2 = s.sm[3][c], su = s.sums[c], sx = s.sums[HH + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (z2[ct][r] - m2) * i2;
      const float n2 = fmaf(xh, g2, b2);
      const float dn2 = dh[ct][r] * gam[ct][r];
      dz2[ct][r] = g2 * i2 * (dn2 - su - xh * sx);
      dgam[ct][r] = dh[ct][r] * n2;
      s.A[c * PRM_ + wave * 16 + 4 * lq + r] = dz2[ct][r];
    }
  }
  acc_rows_store(a.DZ2 + (size_t)k * a.B * HH, dz2, rowA, ok, li);
  __syncthreads();                                   // dz2 parked
  m16_t da1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  mma16_t(da1, s.A, s.Wf, wave, li, lq);
  m16_t a1[2], v[2], w[2], dbet[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = ct * 16 + li;
    const float g1 = s.aff[2][c], b1 = s.aff[3][c], m1 = s.sm[0][c], i1 = s.sm[1][c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (z1[ct][r] - m1) * i1;
      const float n1 = fmaf(xh, g1, b1);
      const float fv = fmaf(gam[ct][r], n1, bet[ct][r]);
      a1[ct][r] = fv > 0.f ? fv : 0.f;
      const float df1 = fv > 0.f ? da1[ct][r] : 0.f;
      dgam[ct][r] = fmaf(df1, n1, dgam[ct][r]);
      dbet[ct][r] = dh[ct][r] + df1;
      v[ct][r] = ok[r] ? df1 * gam[ct][r] : 0.f;          // dn1
      w[ct][r] = v[ct][r] * xh;
    }
  }
  acc_rows_store(a.A1 + (size_t)k * a.B * HH, a1, rowA, ok, li);
  acc_rows_store(a.DG + (size_t)k * a.B * HH, dgam, rowA, ok, li);
  acc_rows_store(a.DB + (size_t)k * a.B * HH, dbet, rowA, ok, li);
  acc_rows_store(a.DN1, v, rowA, ok, li);
  colsums2_m(s, v, w, wave, li, lq, a.Q + ((size_t)(2 * k) * a.nblocks + blockIdx.x) * 2 * HH);
}

// kind C (block k): bn1 backward -> dz1; dh_{k-1} = dh_k + fc1^T dz1; then part a of block k-1, or the fc_in ReLU for k = 0
struct CSeg { int fc1_w, bn1_g, bn1_b, fgp_w, fgp_b, k; };     // fgp_*: FiLM gamma of block k-1
__global__ void __launch_bounds__(NT) g_bwd_c4_kernel(const float* __restrict__ PRM, GBwd a, CSeg f) {
  __shared__ SmemMB s;
  const int k = f.k;
  const int lane = threadIdx.x & (FT - 1), wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const size_t rowL = (size_t)blockIdx.x * FT + lane;
  const size_t rowA = (size_t)blockIdx.x * FT + wave * 16 + 4 * lq;
  const bool more = k > 0;                           // kernel-uniform
  const int kp = more ? k - 1 : 0;
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ok[r] = rowA + r < (size_t)a.B;
  // ---- the burst (with the operands of part a of block k-1, or h0 for the fc_in ReLU)
  WRegs<HH> w_fc1; WRegs<MAXCOND> w_g; CondRegs cr; PRegs pr;
  float z1[2][4], dn1[2][4], dhv[2][4], zp[2][4], g1 = 0.f, smr = 0.f, gg_old = 0.f, gb_old = 0.f;
  wload<HH>(w_fc1, PRM + f.fc1_w, nullptr);
  acc_rows_load(z1, a.Z1 + (size_t)k * a.B * HH, rowA, a.B, li);
  acc_rows_load(dn1, a.DN1, rowA, a.B, li);
  acc_rows_load(dhv, a.DH + (size_t)k * a.B * HH, rowA, a.B, li);
  acc_rows_load(zp, more ? a.Z2 + (size_t)kp * a.B * HH : a.H, rowA, a.B, li);     // z2 of block k-1, or h0
  if (more) {
    wload<MAXCOND>(w_g, PRM + f.fgp_w, PRM + f.fgp_b);
    cload(cr, a.onehot, a.mask, min(rowL, (size_t)a.B - 1), rowL < (size_t)a.B, wave);
  }
  if (threadIdx.x < HH) g1 = PRM[f.bn1_g + threadIdx.x];
  if (threadIdx.x < 4 * HH) {                        // sm[0..1]: bn1_k; sm[2..3]: bn2_{k-1} (unused for k = 0)
    const int w = threadIdx.x >> 5, c = threadIdx.x & (HH - 1);
    const int lyr = w < 2 ? 2 * k : 2 * kp + 1;
    smr = a.SM[(size_t)lyr * 2 * HH + (w & 1) * HH + c];
  }
  if (threadIdx.x < HH && a.accumulate) { gg_old = a.grads[f.bn1_g + threadIdx.x]; gb_old = a.grads[f.bn1_b + threadIdx.x]; }
  pload(pr, a.Q + (size_t)(2 * k) * a.nblocks * 2 * HH, a.nblocks);
  // ---- into LDS
  wstore_mb<HH>(s.Wf, nullptr, w_fc1);
  if (more) {
    wstore_mb<MAXCOND>(s.Wg, s.bg, w_g);
#pragma unroll
    for (int t = 0; t < (MAXCOND + NQ - 1) / NQ; ++t) { const int i = wave + t * NQ; if (i < MAXCOND) s.C[i * PRM_ + lane] = cr.v[t]; }
    if (wave < MK_C - MAXCOND) s.C[(MAXCOND + wave) * PRM_ + lane] = 0.f;
  }
  if (threadIdx.x < HH) s.aff[2][threadIdx.x] = g1;
  if (threadIdx.x < 4 * HH) s.sm[threadIdx.x >> 5][threadIdx.x & (HH - 1)] = smr;
  { const int c = threadIdx.x & (HH - 1), part = threadIdx.x >> 5; s.fin[part][0][c] = pr.s; s.fin[part][1][c] = pr.q; }
  bnb_finish_m(s, a, f.bn1_g, f.bn1_b, gg_old, gb_old);
  m16_t dz1[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = ct * 16 + li;
    const float ga = s.aff[2][c], m1 = s.sm[0][c], i1 = s.sm[1][c], su = s.sums[c], sx = s.sums[HH + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (z1[ct][r] - m1) * i1;
      dz1[ct][r] = ga * i1 * (dn1[ct][r] - su - xh * sx);
      s.A[c * PRM_ + wave * 16 + 4 * lq + r] = dz1[ct][r];
    }
  }
  acc_rows_store(a.DZ1 + (size_t)k * a.B * HH, dz1, rowA, ok, li);
  __syncthreads();
  m16_t t[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  mma16_t(t, s.A, s.Wf, wave, li, lq);
  m16_t dh[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[ct][r] = dhv[ct][r] + t[ct][r];
  if (more) {                                        // kernel-uniform
    acc_rows_store(a.DH + (size_t)kp * a.B * HH, dh, rowA, ok, li);
    // part "a" of block k-1: dn2 = dh * gam; partial sums (dn2, dn2 * xhat2)
    m16_t gam[2], v[2], w[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) gam[ct][r] = s.bg[ct * 16 + li];
    mma16_c(gam, s.C, s.Wg, wave, li, lq);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int c = ct * 16 + li;
      const float m2 = s.sm[2][c], i2 = s.sm[3][c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (zp[ct][r] - m2) * i2;
        v[ct][r] = ok[r] ? dh[ct][r] * gam[ct][r] : 0.f;
        w[ct][r] = v[ct][r] * xh;
      }
    }
    colsums2_m(s, v, w, wave, li, lq, a.Q + ((size_t)(2 * kp + 1) * a.nblocks + blockIdx.x) * 2 * HH);
    return;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[ct][r] = zp[ct][r] > 0.f ? dh[ct][r] : 0.f;
  acc_rows_store(a.DZIN, dh, rowA, ok, li);
}

}  // namespace
}  // namespace pcg

using namespace pcg;

static bool heads_at_most_32_wide(const GDesc& d) {      // a lane keeps its half of a head's columns (<= 16) in registers
  for (int h = 0; h < d.nheads; ++h)
    if (d.seg[h + 1] - d.seg[h] > 32) return false;
  return true;
}

// C-side mirrors of the argument blocks (plain arrays of offsets / pointers: see include/pcgan_hip.h)
extern "C" int pcg_house_g_fwd(const pcg_house_g_desc* desc, const pcg_house_g_fwd_args* args, pcg_stream_t stream) {
  PCG_REQUIRE(desc && args, "pcg_house_g_fwd: null argument block");
  static_assert(sizeof(pcg_house_g_desc) == sizeof(GDesc), "descriptor layouts differ");
  GDesc d;
  std::memcpy(&d, desc, sizeof(d));
  PCG_REQUIRE(desc->hidden == 32 && desc->nblocks == 5, "pcg_house_g_fwd: built for hidden width 32 and 5 residual blocks");
  PCG_REQUIRE(d.D == DIN && d.NC == NCLS && d.nheads >= 0 && d.nheads <= MAXHEADS && d.ncont >= 0 &&
                  d.ncont <= HH && d.seg[d.nheads] <= MAXT && d.seg[d.nheads] + d.ncont <= MAXCOLS && heads_at_most_32_wide(d),
              "pcg_house_g_fwd: built for input_dim 17, 4 classes, <= 8 heads, <= 96 packed categorical + continuous output columns");
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
  const dim3 grid(a.nblocks), block4(FT * NQ);
  hipLaunchKernelGGL(g_fwd_first4_kernel, grid, block4, 0, s, args->params, a, d);
  if (int e = launch_status("g_fwd_first4_kernel")) return e;
  for (int k = 0; k < NBLK; ++k) {
    const bool more = k < NBLK - 1;
    const FSeg fa{d.fg_w[k], d.fg_b[k], d.fb_w[k], d.fb_b[k], d.fc2_w[k], d.fc2_b[k], d.bn1_g[k], d.bn1_b[k],
                  bs.running_mean[2 * k], bs.running_var[2 * k], bs.nbt[2 * k], k, 2 * k, 1};
    const FSeg fb{d.fg_w[k], d.fg_b[k], d.fb_w[k], d.fb_b[k], more ? d.fc1_w[k + 1] : 0, more ? d.fc1_b[k + 1] : 0, d.bn2_g[k], d.bn2_b[k],
                  bs.running_mean[2 * k + 1], bs.running_var[2 * k + 1], bs.nbt[2 * k + 1], k, 2 * k + 1, more ? 1 : 0};
    hipLaunchKernelGGL(g_fwd_a4_kernel, grid, block4, 0, s, args->params, a, fa);
    if (int e = launch_status("g_fwd_a4_kernel")) return e;
    hipLaunchKernelGGL(g_fwd_b4_kernel, grid, block4, 0, s, args->params, a, fb);
    if (int e = launch_status("g_fwd_b4_kernel")) return e;
  }
  hipLaunchKernelGGL(g_heads16_kernel, dim3((a.B + HR - 1) / HR), block4, 0, s, args->params, a, d);
  if (int e = launch_status("g_heads16_kernel")) return e;
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
  const int nblk16 = (a.B + HR - 1) / HR;
  hipLaunchKernelGGL(g_bwd_first16_kernel, dim3(nblk16), block4, 0, s, args->params, a, d);
  if (int e = launch_status("g_bwd_first16_kernel")) return e;
  for (int k = NBLK - 1; k >= 0; --k) {
    const BSeg fb{d.fg_w[k], d.fg_b[k], d.fb_w[k], d.fb_b[k], d.fc2_w[k], d.bn2_g[k], d.bn2_b[k], d.bn1_g[k], d.bn1_b[k], k, k == NBLK - 1 ? nblk16 : a.nblocks};
    const CSeg fc{d.fc1_w[k], d.bn1_g[k], d.bn1_b[k], k > 0 ? d.fg_w[k - 1] : 0, k > 0 ? d.fg_b[k - 1] : 0, k};
    hipLaunchKernelGGL(g_bwd_b4_kernel, grid, block4, 0, s, args->params, a, fb);
    if (int e = launch_status("g_bwd_b4_kernel")) return e;
    hipLaunchKernelGGL(g_bwd_c4_kernel, grid, block4, 0, s, args->params, a, fc);
    if (int e = launch_status("g_bwd_c4_kernel")) return e;
  }
  return PCG_OK;
}
