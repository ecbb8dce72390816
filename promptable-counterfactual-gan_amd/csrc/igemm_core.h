// igemm_core.h — the one MFMA main loop behind every dense contraction of the GAN step.
//
// C[M][N] = sum_k A[m][k] * B[k][n] in exact fp32 on v_mfma_f32_32x32x2_f32 (gfx950).  A block of 256
// threads (4 waves, one per SIMD) owns a BM x BN tile; each wave owns a WTM x WTN sub-tile made of
// 32x32 MFMA tiles.  Operand tiles of depth BK=32 are gathered global -> registers by "loader"
// functors (which know the convolution geometry: im2col rows, sub-pixel phases, zero padding), written
// to LDS, and double-buffered so the gather of k-tile t+1 is in flight while the MFMAs of k-tile t
// issue.  fp32 MFMA is 64 cycles per instruction, so one ds_read feeds many matrix cycles: LDS
// bandwidth is never the limiter here; what matters is coalesced 16-byte global loads, conflict-free
// LDS images and enough independent accumulators (TM*TN >= 2) to keep the matrix pipe issuing.
//
// LDS images (floats):
//   K-major  [rows][BK+4]  : source rows are k-contiguous (NHWC im2col rows, OHWI weight rows).
//                            Fragment = one ds_read_b128 : lane (i,h) gets k = k0+4h .. k0+4h+3 of row i.
//   MN-major [BK][rows+4]  : source is contiguous along m/n for a fixed k (transposed operands).
//                            Fragment = four ds_read_b32 : same (i,h,t) -> k = k0+4h+t mapping.
// Both give lane (i = lane&31, h = lane>>5) the values a[t] = A[i][k0+4h+t], t=0..3; MFMA number t of
// a group consumes k-pair {k0+t, k0+4+t} — the order of k inside the sum is free as long as A and B
// agree, which they do by construction.
#pragma once
#include "pcg_common.h"

namespace pcg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IG_THREADS = 256;
constexpr int IG_BK = 32;
constexpr int IG_LDK = IG_BK + 4;  // K-major row stride: 144 B = 9*16 (aligned for b128, conflict-free)

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
};

// floats of LDS one stage of an operand tile needs
template <int ROWS, bool KMAJOR>
struct LdsImage {
  static constexpr int LDM = ROWS + 4;
  static constexpr int FLOATS = KMAJOR ? ROWS * IG_LDK : IG_BK * LDM;
  static constexpr int NV = ROWS / 32;  // float4 per thread per k-tile (256 threads)

  // thread -> (row, k-quad) for K-major; (k-row, column-quad) for MN-major
  __device__ static __forceinline__ void store(float* lds, const float4 (&v)[NV]) {
    const int tid = threadIdx.x;
    if constexpr (KMAJOR) {
      const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
      for (int p = 0; p < NV; ++p)
        *reinterpret_cast<float4*>(lds + (r0 + 32 * p) * IG_LDK + 4 * kq) = v[p];
    } else {
      constexpr int C4 = ROWS / 4;          // float4 per k-row
      constexpr int KR = IG_THREADS / C4;   // k-rows per pass
      const int c4 = tid % C4, kr0 = tid / C4;
#pragma unroll
      for (int p = 0; p < NV; ++p)
        *reinterpret_cast<float4*>(lds + (kr0 + KR * p) * LDM + 4 * c4) = v[p];
    }
  }
  // fragment for MFMA tile rows [row0, row0+32), k-group ks (8 k's): f[t] = T[row0+i][8ks+4h+t]
  __device__ static __forceinline__ void frag(const float* lds, int row0, int ks, int li, int lh, float (&f)[4]) {
    if constexpr (KMAJOR) {
      const float4 q = *reinterpret_cast<const float4*>(lds + (row0 + li) * IG_LDK + 8 * ks + 4 * lh);
      f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
    } else {
      const float* p = lds + (8 * ks + 4 * lh) * LDM + row0 + li;
      f[0] = p[0]; f[1] = p[LDM]; f[2] = p[2 * LDM]; f[3] = p[3 * LDM];
    }
  }
};

// Loader concept (per-thread state, constructed once per block):
//   static constexpr bool KMAJOR; static constexpr int ROWS;
//   __device__ void load_next(float4 (&v)[ROWS/32]);   // gathers the next k-tile (sequential) into registers
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void igemm_mainloop(LA& la, LB& lb, int ktiles, f32x16 (&acc)[Cfg::TM][Cfg::TN],
                                               float* smem) {
  using IA = LdsImage<Cfg::BM, LA::KMAJOR>;
  using IB = LdsImage<Cfg::BN, LB::KMAJOR>;
  static_assert(LA::ROWS == Cfg::BM && LB::ROWS == Cfg::BN, "loader/tile mismatch");
  float* As = smem;
  float* Bs = smem + 2 * IA::FLOATS;

  float4 ra[IA::NV], rb[IB::NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (ktiles <= 0) return;
  la.load_next(ra);
  lb.load_next(rb);
  IA::store(As, ra);
  IB::store(Bs, rb);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    const bool has_next = (kt + 1 < ktiles);
    if (has_next) {  // issue the next tile's global gathers before touching the matrix pipe
      la.load_next(ra);
      lb.load_next(rb);
    }
    const float* as = As + cur * IA::FLOATS;
    const float* bs = Bs + cur * IB::FLOATS;
#pragma unroll
    for (int ks = 0; ks < IG_BK / 8; ++ks) {
      float a[Cfg::TM][4], b[Cfg::TN][4];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) IA::frag(as, wm * Cfg::WTM + 32 * i, ks, li, lh, a[i]);
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) IB::frag(bs, wn * Cfg::WTN + 32 * j, ks, li, lh, b[j]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
          for (int j = 0; j < Cfg::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    if (has_next) {
      IA::store(As + (cur ^ 1) * IA::FLOATS, ra);
      IB::store(Bs + (cur ^ 1) * IB::FLOATS, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
}

template <class Cfg, class LA, class LB>
constexpr int igemm_smem_floats() {
  return 2 * LdsImage<Cfg::BM, LA::KMAJOR>::FLOATS + 2 * LdsImage<Cfg::BN, LB::KMAJOR>::FLOATS;
}

// Accumulator element (tile i,j ; register r) of lane (li,lh) sits at
//   row = 32*i + (r&3) + 8*(r>>2) + 4*lh   col = 32*j + li      (within the wave tile)
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

}  // namespace pcg
