// igemm_core.h — the one MFMA main loop behind every dense contraction of the GAN step.
//
// C[M][N] = sum_k A[m][k] * B[k][n] in exact fp32 on v_mfma_f32_32x32x2_f32 (gfx950).  A workgroup owns a BM x BN
// tile; operand tiles of depth BK=32 are gathered global -> registers by "loader" functors (which know the
// convolution geometry: im2col rows, sub-pixel phases, zero padding — conv_loaders.h), written to LDS and
// double-buffered.
//
// Wave specialisation.  fp32 MFMA is 64 cycles per instruction and a wave issues in order, so every instruction a
// wave spends on gathering (address VALU, buffer_load, ds_write, waits) between two MFMAs is a hole in the matrix
// pipe unless another wave fills it.  Measured on MI355X: a bare MFMA loop sustains 155 TFLOP/s (64.0 cycles per
// MFMA at 2.4 GHz); the same MFMAs with the gather in the same instruction stream reach 67-75 %.  The workgroup is
// therefore split by role:
//   waves 0..3  "consumers": ds_read fragments + MFMA only (one per SIMD, 64x64 of the tile each, 4 accumulators)
//   waves 4..7  "producers": gather tile t+2 into registers, ds_write tile t+1, nothing else
// VALU/VMEM/LDS-write work of the producers issues on the same SIMDs in the shadow of the consumers' MFMAs (the
// matrix and vector pipes are separate).  One s_barrier per k-tile hands stage (t+1)%2 to the consumers and stage
// t%2 back to the producers:
//   RAW  consumers read stage (t+1)%2 only after barrier t+1, which every producer reaches after its ds_writes retired
//   WAR  producers overwrite stage (t+1)%2 during iteration t; its last readers finished before barrier t
// Two workgroups fit per CU (73.7 KB LDS, <=128 VGPR+AGPR), so each SIMD hosts two consumers whose barrier / first-
// fragment latencies cover each other.
//
// LDS images (floats):
//   K-major  [rows][BK+4]  : source rows are k-contiguous (NHWC im2col rows, OHWI weight rows).
//                            Fragment = one ds_read_b128 : lane (i,h) gets k = k0+4h .. k0+4h+3 of row i.
//   MN-major [BK][rows+4]  : source is contiguous along m/n for a fixed k (transposed operands).
//                            Fragment = four ds_read_b32 : same (i,h,t) -> k = k0+4h+t mapping.
// Both give lane (i = lane&31, h = lane>>5) the values a[t] = A[i][k0+4h+t], t=0..3; MFMA number t of a group
// consumes k-pair {k0+t, k0+4+t} — the order of k inside the sum is free as long as A and B agree.
#pragma once
#include "pcg_common.h"

namespace pcg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IG_LOADERS = 256;            // threads that gather one k-tile (4 waves)
constexpr int IG_THREADS = 512;            // 4 consumer + 4 producer waves
constexpr int IG_BK = 32;
#ifndef PCG_PREFETCH_DEPTH
#define PCG_PREFETCH_DEPTH 2               // k-tiles of operand gathers in flight per producer thread (1: the r01 pipeline)
#endif
constexpr int IG_LDK = IG_BK + 4;          // K-major row stride: 144 B = 9*16 (aligned for b128, conflict-free)

// SWZ_: K-major LDS images without row padding, 16-byte chunks XOR-swizzled by the row (LdsImage): 128x64 tiles then need 49 KB
//       instead of 55 KB and THREE workgroups fit a CU's 160 KB — while one of them is in its prologue / epilogue the SIMD still
//       hosts two consumer waves (one wave alone does not keep the MFMA pipe full).  MINW_: waves per SIMD the register budget
//       must allow (launch bounds); PF_: k-tiles of gathers in flight per producer thread (register sets).
template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, bool SWZ_ = false, int MINW_ = 4, int PF_ = PCG_PREFETCH_DEPTH>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr bool SWZ = SWZ_;
  static constexpr int MINW = MINW_, PF = PF_;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 consumer waves per block");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
};

template <int ROWS, bool KMAJOR, bool SWZ = false>
struct LdsImage {
  static constexpr int LDM = ROWS + 4;
  // K-major row stride: 36 floats (padded, conflict-free as is) or 32 floats with the row's eight 16-byte chunks permuted by
  // chunk ^ ((row >> 1) & 7): sixteen consecutive rows reading the same logical chunk (one quarter-wave of a ds_read_b128) then
  // touch all 64 banks once (even rows: bank base 0, chunks 0..7 each once; odd rows: bank base 32, likewise)
  static constexpr int LDKS = (KMAJOR && SWZ) ? IG_BK : IG_LDK;
  static constexpr int FLOATS = KMAJOR ? ROWS * LDKS : IG_BK * LDM;
  static constexpr int NV = ROWS / 32;  // float4 per loader thread per k-tile (256 loader threads)

  // loader thread tid (0..255) -> (row, k-quad) for K-major; (k-row, column-quad) for MN-major
  __device__ static __forceinline__ void store(float* lds, const float4 (&v)[NV], int tid) {
    if constexpr (KMAJOR && SWZ) {
      const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
      for (int p = 0; p < NV; ++p)     // (row >> 1) & 7 == (r0 >> 1) & 7: adding 32 * p does not touch bits 1..3 of the row
        *reinterpret_cast<float4*>(lds + (r0 + 32 * p) * IG_BK + 4 * (kq ^ ((r0 >> 1) & 7))) = v[p];
    } else if constexpr (KMAJOR) {
      const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
      for (int p = 0; p < NV; ++p)
        *reinterpret_cast<float4*>(lds + (r0 + 32 * p) * IG_LDK + 4 * kq) = v[p];
    } else if constexpr (IG_LOADERS % (ROWS / 4) == 0) {
      constexpr int C4 = ROWS / 4;          // float4 per k-row
      constexpr int KR = IG_LOADERS / C4;   // k-rows per pass
      const int c4 = tid % C4, kr0 = tid / C4;
#pragma unroll
      for (int p = 0; p < NV; ++p)
        *reinterpret_cast<float4*>(lds + (kr0 + KR * p) * LDM + 4 * c4) = v[p];
    } else {
      // ROWS = 192 (48 float4 per k-row do not divide the 256 loader threads): three 64-column sub-images side by side, each
      // loaded like a 64-row image (16 float4 per k-row, 16 k-rows per pass, 2 passes); v[2*s + h] = sub-image s, k-rows 16h..16h+15
      static_assert(ROWS % 64 == 0, "MN-major image: ROWS must divide the loader threads or be a multiple of 64");
      const int c4 = tid & 15, kr0 = tid >> 4;
#pragma unroll
      for (int s = 0; s < ROWS / 64; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          *reinterpret_cast<float4*>(lds + (kr0 + 16 * h) * LDM + 64 * s + 4 * c4) = v[2 * s + h];
    }
  }
  // fragment for MFMA tile rows [row0, row0+32), k-group ks (8 k's): f[t] = T[row0+i][8ks+4h+t]
  __device__ static __forceinline__ void frag(const float* lds, int row0, int ks, int li, int lh, float (&f)[4]) {
    if constexpr (KMAJOR && SWZ) {
      const int row = row0 + li;
      const float4 q = *reinterpret_cast<const float4*>(lds + row * IG_BK + 4 * ((2 * ks + lh) ^ ((row >> 1) & 7)));
      f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
    } else if constexpr (KMAJOR) {
      const float4 q = *reinterpret_cast<const float4*>(lds + (row0 + li) * IG_LDK + 8 * ks + 4 * lh);
      f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
    } else {
      const float* p = lds + (8 * ks + 4 * lh) * LDM + row0 + li;
      f[0] = p[0]; f[1] = p[LDM]; f[2] = p[2 * LDM]; f[3] = p[3 * LDM];
    }
  }
};

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// raw barrier behind an explicit lgkmcnt(0): __syncthreads() would also drain vmcnt, i.e. stall the producers on the
// gathers they have just issued for tile t+2.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <class Cfg, bool AK, bool BK_>
constexpr int igemm_smem_floats() {
  return 2 * (LdsImage<Cfg::BM, AK, Cfg::SWZ>::FLOATS + LdsImage<Cfg::BN, BK_, Cfg::SWZ>::FLOATS);
}

// Loader concept (per-thread state of a producer thread, constructed with its loader-thread id 0..255):
//   static constexpr bool KMAJOR; static constexpr int ROWS;
//   __device__ void load_next(float4 (&v)[ROWS/32]);   // gathers the next k-tile (sequential) into registers
//   __device__ void transform(float4 (&v)[ROWS/32]);   // applied to the registers of the last load_next before the ds_write
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void igemm_produce(LA& la, LB& lb, int ktiles, float* smem, int tid) {
  using IA = LdsImage<Cfg::BM, LA::KMAJOR, Cfg::SWZ>;
  using IB = LdsImage<Cfg::BN, LB::KMAJOR, Cfg::SWZ>;
  static_assert(LA::ROWS == Cfg::BM && LB::ROWS == Cfg::BN, "loader/tile mismatch");
  float* As = smem;
  float* Bs = smem + 2 * IA::FLOATS;
  // loaders that carry an input transform keep ONE pending tile of transform state: they run the depth-1 pipeline
  if constexpr (Cfg::PF == 2 && !LA::XFORM && !LB::XFORM) {
  // Two k-tiles of gathers in flight.  One k-tile of MFMAs is 2048 (128x64 tile) .. 4096 cycles (128x128) = 1 .. 2 us, which is no
  // more than a loaded HBM / L2 round trip: with a single tile in flight the producers reach the hand-over barrier late and the
  // consumers wait.  (Measured r02: +2..4 % per kernel in isolation, nothing in the back-to-back step — the larger part of what idle
  // producers gain in the ablation, +12 % on 128x128 tiles and +19 % on 128x64, is not gather latency.)  Register sets 0/1 alternate: at
  // iteration kt the set holding tile kt+1 is written to LDS and immediately refilled with the gathers of tile kt+3.
  float4 ra[2][IA::NV], rb[2][IB::NV];
  if (ktiles > 0) {
    la.load_next(ra[0]); lb.load_next(rb[0]);                       // tile 0
    if (ktiles > 1) { la.load_next(ra[1]); lb.load_next(rb[1]); }   // tile 1
    la.transform(ra[0]); lb.transform(rb[0]);
    IA::store(As, ra[0], tid);
    IB::store(Bs, rb[0], tid);
    if (ktiles > 2) { la.load_next(ra[0]); lb.load_next(rb[0]); }   // tile 2
  }
  lds_barrier();  // barrier 0: stage 0 is ready
  // invariant at the top of iteration kt: set (kt+1)&1 holds tile kt+1 (in flight or landed), set kt&1 holds tile kt+2
  int kt = 0;
  for (; kt + 1 < ktiles; kt += 2) {
    // kt even: tile kt+1 is in set 1 -> stage 1; refill set 1 with tile kt+3
    la.transform(ra[1]); lb.transform(rb[1]);
    IA::store(As + IA::FLOATS, ra[1], tid);
    IB::store(Bs + IB::FLOATS, rb[1], tid);
    if (kt + 3 < ktiles) { la.load_next(ra[1]); lb.load_next(rb[1]); }
    lds_barrier();
    // kt+1 odd: tile kt+2 is in set 0 -> stage 0; refill set 0 with tile kt+4
    if (kt + 2 < ktiles) {
      la.transform(ra[0]); lb.transform(rb[0]);
      IA::store(As, ra[0], tid);
      IB::store(Bs, rb[0], tid);
    }
    if (kt + 4 < ktiles) { la.load_next(ra[0]); lb.load_next(rb[0]); }
    lds_barrier();
  }
  if (kt < ktiles) lds_barrier();   // odd ktiles: the last iteration has nothing left to stage
  } else {
  float4 ra[IA::NV], rb[IB::NV];
  if (ktiles > 0) {
    la.load_next(ra);
    lb.load_next(rb);
    la.transform(ra); lb.transform(rb);     // input transform of the tile just fetched (no-op for plain loaders)
    IA::store(As, ra, tid);
    IB::store(Bs, rb, tid);
    if (ktiles > 1) {
      la.load_next(ra);
      lb.load_next(rb);
    }
  }
  lds_barrier();  // barrier 0: stage 0 is ready
  int nxt = 1;
  for (int kt = 0; kt < ktiles; ++kt) {
#ifndef PCG_ABL_PRODUCER_IDLE   // timing-only ablation: producers just keep the barrier protocol
    if (kt + 1 < ktiles) {
      la.transform(ra); lb.transform(rb);
      IA::store(As + nxt * IA::FLOATS, ra, tid);
      IB::store(Bs + nxt * IB::FLOATS, rb, tid);
    }
    if (kt + 2 < ktiles) {
      la.load_next(ra);
      lb.load_next(rb);
    }
#endif
    lds_barrier();  // barrier kt+1: stage nxt handed to the consumers, stage nxt^1 handed back
    nxt ^= 1;
  }
  }
}

// Consumer: fragments are double-buffered in registers so that the LDS read of k-group g+1 is in flight under the
// 16 MFMAs of k-group g; the hand-over barrier of the k-tile sits BEFORE its last k-group (whose operands are already
// in registers), and the first fragments of the next tile are fetched right behind it — no LDS latency is exposed at
// the tile boundary.  s_setprio keeps MFMA issue ahead of the co-resident producers' vector instructions.
// Diagnostic build (-DPCG_CLOCK_STAMP): the clock the chip holds INSIDE the main loop = delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6), one pair per block, written to a buffer nothing else reads.
struct ClockStamp {
  unsigned long long* out; int slots;
#ifdef PCG_CLOCK_STAMP
  unsigned long long t0, r0;
  __device__ __forceinline__ void begin() { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  __device__ __forceinline__ void end() {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    if (out && threadIdx.x == 0 && b < slots) { out[2 * b] = t1 - t0; out[2 * b + 1] = r1 - r0; }
  }
#else
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void end() {}
#endif
};

template <class Cfg, bool AK, bool BK_>
__device__ __forceinline__ void igemm_consume(int ktiles, f32x16 (&acc)[Cfg::TM][Cfg::TN], const float* smem, ClockStamp cs = ClockStamp{nullptr, 0}) {
  using IA = LdsImage<Cfg::BM, AK, Cfg::SWZ>;
  using IB = LdsImage<Cfg::BN, BK_, Cfg::SWZ>;
  constexpr int KG = IG_BK / 8;  // k-groups per tile
  const float* As = smem;
  const float* Bs = smem + 2 * IA::FLOATS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;
  const int arow = wm * Cfg::WTM, brow = wn * Cfg::WTN;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float a[2][Cfg::TM][4], b[2][Cfg::TN][4];
  auto fetch = [&](const float* as, const float* bs, int ks, int buf) {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) IA::frag(as, arow + 32 * i, ks, li, lh, a[buf][i]);
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) IB::frag(bs, brow + 32 * j, ks, li, lh, b[buf][j]);
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][i][t], b[buf][j][t], acc[i][j], 0, 0, 0);
  };

  lds_barrier();  // barrier 0: stage 0 is ready
  if (ktiles <= 0) return;
  __builtin_amdgcn_s_setprio(2);
  cs.begin();
  fetch(As, Bs, 0, 0);
  int cur = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    const float* as = As + cur * IA::FLOATS;
    const float* bs = Bs + cur * IB::FLOATS;
#pragma unroll
    for (int ks = 0; ks < KG - 1; ++ks) {
      fetch(as, bs, ks + 1, (ks + 1) & 1);
      mma(ks & 1);
    }
    lds_barrier();  // barrier kt+1: all reads of stage cur have retired (lgkmcnt(0)); stage cur^1 is ready
    cur ^= 1;
    if (kt + 1 < ktiles) fetch(As + cur * IA::FLOATS, Bs + cur * IB::FLOATS, 0, KG & 1);
    mma((KG - 1) & 1);
  }
  cs.end();
  __builtin_amdgcn_s_setprio(0);
}

// ---- epilogue: accumulators -> LDS (wave-private region) -> 16-byte row-contiguous global stores ------------------
// A 32x32 MFMA accumulator has its column on the lane and its rows in the registers, so a direct store is 64 4-byte-
// per-lane instructions per wave.  Staging the wave's WTM x WTN tile through LDS turns that into WTM*WTN/256 dwordx4
// stores whose lanes cover whole rows.  Every LDS read of the main loop retired before the final barrier and each
// wave only touches its own region, so no further barrier is needed.  `row_base(row)` returns the output pointer of
// tile row `row` (0..BM) at column 0 of the tile's N range, or nullptr for a row outside the problem.
template <class Cfg>
constexpr int epilogue_smem_floats() { return 4 * Cfg::WTM * (Cfg::WTN + 4); }

// Backward-pass epilogues (the tile is a gradient w.r.t. the OUTPUT a = act(..) of the layer below; `aux` is a tensor of the
// same shape and addressing as the output):
//   EPI_MASK   aux = a (post-activation).  v *= (aux > 0 ? 1 : neg)            — ReLU / LeakyReLU backward, no extra pass
//   EPI_BNBWD  aux = z (the layer below's pre-BatchNorm conv output).  pre = z*sc + sh (sc = gamma*invstd, sh = beta - mean*sc),
//              v *= (pre > 0 ? 1 : neg), xhat = (z - mean)*invstd; column sums of v and v*xhat go to the partial rows
//              (-> dbeta, dgamma and the two means BatchNorm's backward needs): no separate reduction pass over (dy, z).
//   EPI_ADD    aux = an addend of the output's shape (may BE the output: in-place accumulation).  v += aux   — skip connections
//   EPI_ADDSUM EPI_ADD + the BatchNorm-backward column sums of the SUM for the next BatchNorm down the skip chain: aux2 = that layer's
//              pre-BatchNorm output z (delta2_bytes), mean / invstd its statistics; no activation in between (x + 0.1*bn2(...), models/
//              generator.py:20).  The partial rows get sum(scale*v) and sum(scale*v*xhat) (scale = neg: the 0.1 of the residual branch);
//              the stored tile is the unscaled sum — the skip path needs it as it is.
enum { EPI_NONE = 0, EPI_MASK = 1, EPI_BNBWD = 2, EPI_ADD = 3, EPI_ADDSUM = 4 };
struct EpiAux {
  int mode;                 // EPI_*
  float neg;                // slope of the negative side (0 ReLU, 0.2 LeakyReLU, 1 = no activation); EPI_ADDSUM: the sum scale
  int64_t delta_bytes;      // (char*)aux - (char*)out
  int64_t delta2_bytes;     // EPI_ADDSUM: (char*)z_next - (char*)out
  const float* mean; const float* invstd; const float* gamma; const float* beta;   // EPI_BNBWD, per output column
};

template <class Cfg, class RowBase>
__device__ __forceinline__ void igemm_store_tile(f32x16 (&acc)[Cfg::TM][Cfg::TN], float* smem, int n_block, int N,
                                                 const float* bias, RowBase row_base, double* stat_row = nullptr, int act = PCG_ACT_NONE,
                                                 float slope = 0.f, const EpiAux* epi = nullptr) {
  constexpr int LDW = Cfg::WTN + 4;
  constexpr int Q = Cfg::WTN / 4;          // float4 per row of the wave tile
  constexpr int RPI = 64 / Q;              // rows per store instruction (Q = 24, the 64x192 tile: 2 rows, lanes 48..63 idle)
  static_assert(Cfg::WTM % RPI == 0, "wave tile rows must be a multiple of the rows stored per instruction");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N, li = lane & 31, lh = lane >> 5;
  float* reg = smem + wave * (Cfg::WTM * LDW);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) reg[(32 * i + acc_row(r, lh)) * LDW + 32 * j + li] = acc[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int cq = lane % Q, r0 = lane / Q;
  const int n = n_block + wn * Cfg::WTN + 4 * cq;
  const bool nok = n < N && r0 < RPI;  // N % 4 == 0: a quad is entirely inside or outside; r0 >= RPI: surplus lanes (Q not a power of two)
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && nok) bv = *reinterpret_cast<const float4*>(bias + n);
  // column sums for the fused BatchNorm statistics / BatchNorm-backward sums: accumulated in fp64 (the reference's CPU path sums
  // in double — [torch] at::acc_type<float, false> — and BatchNorm's backward subtracts these means from strongly correlated
  // gradients: an fp32 chain of 16+ same-sign adds is 1e-6 off, which the cancellation amplifies to 1e-3 in the weight gradients)
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  const bool want_sums = stat_row != nullptr;   // wave-uniform
  const int emode = epi ? epi->mode : EPI_NONE;   // wave-uniform
  if (emode == EPI_NONE) {
#pragma unroll
    for (int k = 0; k < Cfg::WTM / RPI; ++k) {
      const int row = r0 + RPI * k;
      float* dst = row_base(wm * Cfg::WTM + row);
      if (dst && nok) {
        float4 v = *reinterpret_cast<const float4*>(reg + row * LDW + 4 * cq);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (act != PCG_ACT_NONE) {   // wave-uniform; ReLU / LeakyReLU only (slope = 0 / negative slope), others are applied by the host wrapper
          v.x = act_neg_scale(v.x, slope); v.y = act_neg_scale(v.y, slope); v.z = act_neg_scale(v.z, slope); v.w = act_neg_scale(v.w, slope);
        }
        *reinterpret_cast<float4*>(dst + wn * Cfg::WTN + 4 * cq) = v;
        if (want_sums) {
          const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
          s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
          s2[0] = fma(d0, d0, s2[0]); s2[1] = fma(d1, d1, s2[1]); s2[2] = fma(d2, d2, s2[2]); s2[3] = fma(d3, d3, s2[3]);
        }
      }
    }
  } else {
    const float neg = epi->neg;
    const int64_t delta = epi->delta_bytes;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), mu = sh, is = sc;
    if (emode == EPI_BNBWD && nok) {
      mu = *reinterpret_cast<const float4*>(epi->mean + n);
      is = *reinterpret_cast<const float4*>(epi->invstd + n);
      const float4 ga = *reinterpret_cast<const float4*>(epi->gamma + n), be = *reinterpret_cast<const float4*>(epi->beta + n);
      bn_fold(ga.x, be.x, mu.x, is.x, sc.x, sh.x); bn_fold(ga.y, be.y, mu.y, is.y, sc.y, sh.y);
      bn_fold(ga.z, be.z, mu.z, is.z, sc.z, sh.z); bn_fold(ga.w, be.w, mu.w, is.w, sc.w, sh.w);
    }
    if (emode == EPI_ADDSUM && nok) {
      mu = *reinterpret_cast<const float4*>(epi->mean + n);
      is = *reinterpret_cast<const float4*>(epi->invstd + n);
    }
    // all aux loads of the wave tile first (independent of the LDS reads), then the arithmetic
    float4 u[Cfg::WTM / RPI];
#pragma unroll
    for (int k = 0; k < Cfg::WTM / RPI; ++k) {
      float* dst = row_base(wm * Cfg::WTM + r0 + RPI * k);
      u[k] = (dst && nok) ? *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(dst + wn * Cfg::WTN + 4 * cq) + delta)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (emode == EPI_ADDSUM) {   // wave-uniform: its own loop (the second aux tensor is read next to each row's arithmetic)
      const int64_t delta2 = epi->delta2_bytes;
      const double sc2 = (double)neg;
#pragma unroll
      for (int k = 0; k < Cfg::WTM / RPI; ++k) {
        const int row = r0 + RPI * k;
        float* dst = row_base(wm * Cfg::WTM + row);
        if (dst && nok) {
          float4 v = *reinterpret_cast<const float4*>(reg + row * LDW + 4 * cq);
          const float4 z = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(dst + wn * Cfg::WTN + 4 * cq) + delta2);
          v.x += u[k].x; v.y += u[k].y; v.z += u[k].z; v.w += u[k].w;
          *reinterpret_cast<float4*>(dst + wn * Cfg::WTN + 4 * cq) = v;
          if (want_sums) {
            const double d0 = sc2 * (double)v.x, d1 = sc2 * (double)v.y, d2 = sc2 * (double)v.z, d3 = sc2 * (double)v.w;
            s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
            s2[0] = fma(d0, (double)((z.x - mu.x) * is.x), s2[0]); s2[1] = fma(d1, (double)((z.y - mu.y) * is.y), s2[1]);
            s2[2] = fma(d2, (double)((z.z - mu.z) * is.z), s2[2]); s2[3] = fma(d3, (double)((z.w - mu.w) * is.w), s2[3]);
          }
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < Cfg::WTM / RPI; ++k) {
      const int row = r0 + RPI * k;
      float* dst = row_base(wm * Cfg::WTM + row);
      if (dst && nok) {
        float4 v = *reinterpret_cast<const float4*>(reg + row * LDW + 4 * cq);
        // same expression as bn_bwd_apply / FnBnBwd use for the recomputed BatchNorm output (sc = 1, sh = 0 for EPI_MASK)
        if (emode == EPI_ADD) {   // wave-uniform
          v.x += u[k].x; v.y += u[k].y; v.z += u[k].z; v.w += u[k].w;
        } else {
          const float4 pre = make_float4(fmaf(u[k].x, sc.x, sh.x), fmaf(u[k].y, sc.y, sh.y), fmaf(u[k].z, sc.z, sh.z), fmaf(u[k].w, sc.w, sh.w));
          v.x *= pre.x > 0.f ? 1.f : neg; v.y *= pre.y > 0.f ? 1.f : neg; v.z *= pre.z > 0.f ? 1.f : neg; v.w *= pre.w > 0.f ? 1.f : neg;
        }
        *reinterpret_cast<float4*>(dst + wn * Cfg::WTN + 4 * cq) = v;
        if (want_sums) {
          s1[0] += (double)v.x; s1[1] += (double)v.y; s1[2] += (double)v.z; s1[3] += (double)v.w;
          s2[0] = fma((double)v.x, (double)((u[k].x - mu.x) * is.x), s2[0]); s2[1] = fma((double)v.y, (double)((u[k].y - mu.y) * is.y), s2[1]);
          s2[2] = fma((double)v.z, (double)((u[k].z - mu.z) * is.z), s2[2]); s2[3] = fma((double)v.w, (double)((u[k].w - mu.w) * is.w), s2[3]);
        }
      }
    }
    }
  }
  // fused BatchNorm statistics: per-column sum / sum of squares over this wave's rows -> one fp64 partial row per
  // (tile row, wave row); a finalize kernel adds the partial rows in a fixed order (bitwise reproducible)
  if (want_sums) {
#pragma unroll
    for (int off = Q; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off); s2[e] += __shfl_xor(s2[e], off); }
    }
    if (r0 == 0 && nok) {
      double* pr = stat_row + (size_t)wm * 2 * N;      // [wm][2][N] inside this tile row's slot
#pragma unroll
      for (int e = 0; e < 4; ++e) { pr[n + e] = s1[e]; pr[N + n + e] = s2[e]; }
    }
  }
}

// Accumulator element (tile i,j ; register r) of lane (li,lh) sits at
//   row = 32*i + (r&3) + 8*(r>>2) + 4*lh   col = 32*j + li      (within the wave tile)

}  // namespace pcg
