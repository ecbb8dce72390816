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

// Raw buffer resources: num_records = the tensor's size in bytes, so an out-of-range byte offset reads as zeros and a store to it
// is dropped — padding, ragged edges and K tails without branches (conv_loaders.h), and the register epilogue's edge handling.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t OOB_OFF = 0x80000000u;  // >= num_records of any accepted tensor

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(rsrc_t r, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, uint32_t off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ void buf_store1(rsrc_t r, uint32_t off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0); }

constexpr int IG_LOADERS = 256;            // threads that gather one k-tile (4 waves)
constexpr int IG_THREADS = 512;            // 4 consumer + 4 producer waves
constexpr int IG_BK = 32;
#ifndef PCG_PREFETCH_DEPTH
#define PCG_PREFETCH_DEPTH 2               // k-tiles of operand gathers in flight per producer thread (1: the r01 pipeline)
#endif
#ifndef PCG_CONSUMER_PRIO
#define PCG_CONSUMER_PRIO 2                // s_setprio of the consumer waves inside their main loop (A/B builds: 0, 1, 2)
#endif
#ifndef PCG_CONSUMER_ALTERNATE
#define PCG_CONSUMER_ALTERNATE 256
#endif
#ifndef PCG_PRODUCER_PRIO
#define PCG_PRODUCER_PRIO 3                // s_setprio of the producer waves: ABOVE the consumers (r03, see igemm_produce)
#endif
constexpr int IG_LDK = IG_BK + 4;          // K-major row stride: 144 B = 9*16 (aligned for b128, conflict-free)

// SWZ_: K-major LDS images without row padding, 16-byte chunks XOR-swizzled by the row (LdsImage): 128x64 tiles then need 49 KB
//       instead of 55 KB and THREE workgroups fit a CU's 160 KB — while one of them is in its prologue / epilogue the SIMD still
//       hosts two consumer waves (one wave alone does not keep the MFMA pipe full).  MINW_: waves per SIMD the register budget
//       must allow (launch bounds); PF_: k-tiles of gathers in flight per producer thread (register sets).
// MN-major operands (both operands of the weight gradient, the weight operand of the grad-input): k-row (0..31) of a k-tile that a
// loader thread of row group kr0 fetches as its i-th of NV loads.  Consecutive rows (r03; r01/r02: kr0 + (32 / NV) * i): the weight
// gradient's x operand is gathered per output PIXEL = k-row, and consecutive pixels share their (image, row) decomposition up to
// a carry — one FastDiv pair per k-tile and thread instead of NV (WgradBLoader).  PCG_MN_CONSEC=0 builds the strided mapping.
#ifndef PCG_MN_CONSEC
#define PCG_MN_CONSEC 1
#endif
template <int NV>
__device__ __forceinline__ constexpr int mn_krow(int kr0, int i) { return PCG_MN_CONSEC ? NV * kr0 + i : kr0 + (32 / NV) * i; }

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, bool SWZ_ = false, int MINW_ = 4, int PF_ = PCG_PREFETCH_DEPTH, bool DMA_ = false>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr bool SWZ = SWZ_;
  static constexpr bool DMA = DMA_;          // operand tiles go global -> LDS directly (buffer_load ... lds), see igemm_produce_dma
  static_assert(!DMA_ || SWZ_, "LDS-DMA needs the lane-linear (unpadded, swizzled) K-major images");
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
      static_assert(KR * NV == IG_BK, "MN-major image: loader threads x loads must cover the k-tile");
#pragma unroll
      for (int p = 0; p < NV; ++p)
        *reinterpret_cast<float4*>(lds + mn_krow<NV>(kr0, p) * LDM + 4 * c4) = v[p];
    } else {
      // ROWS = 192 (48 float4 per k-row do not divide the 256 loader threads): three 64-column sub-images side by side, each
      // loaded like a 64-row image (16 float4 per k-row, 16 k-rows per pass, 2 passes); v[2*s + h] = sub-image s, k-rows 16h..16h+15
      static_assert(ROWS % 64 == 0, "MN-major image: ROWS must divide the loader threads or be a multiple of 64");
      const int c4 = tid & 15, kr0 = tid >> 4;
#pragma unroll
      for (int s = 0; s < ROWS / 64; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          *reinterpret_cast<float4*>(lds + mn_krow<2>(kr0, h) * LDM + 64 * s + 4 * c4) = v[2 * s + h];
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
  // phase stamps of the one-tile-per-workgroup kernels (third quarter of the buffer, 4 words per block): absolute 100 MHz ticks at
  // kernel entry, main-loop begin, main-loop end, epilogue end — where a launch's time outside its main loops goes
  __device__ __forceinline__ void phase(int k) {
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    if (out && threadIdx.x == 0 && b < slots / 8) out[(size_t)slots + (size_t)b * 4 + k] = __builtin_amdgcn_s_memrealtime();
  }
  // barrier-wait accounting (fourth quarter of the buffer, 4 words per block): shader cycles consumer wave 0 / producer wave 4
  // spent inside the per-k-tile hand-over barrier (lgkmcnt wait + s_barrier), and their loop cycles — who waits for whom
  unsigned long long bar_acc, bar_t, sec_acc[3], sec_t;
  __device__ __forceinline__ void sec_begin() { sec_t = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void sec_end(int k) { const unsigned long long t = __builtin_amdgcn_s_memtime(); sec_acc[k] += t - sec_t; sec_t = t; }
  __device__ __forceinline__ void bar_begin() { bar_t = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void bar_end() { bar_acc += __builtin_amdgcn_s_memtime() - bar_t; }
  __device__ __forceinline__ void bar_init() { bar_acc = 0; sec_acc[0] = sec_acc[1] = sec_acc[2] = 0; }
  __device__ __forceinline__ void bar_flush(int who, unsigned long long loop_cycles) {    // who: 0 consumer, 1 producer
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    if (out && (threadIdx.x & 63) == 0 && b < slots / 8) {
      out[(size_t)slots + (size_t)slots / 2 + (size_t)b * 4 + 2 * who] = bar_acc;
      out[(size_t)slots + (size_t)slots / 2 + (size_t)b * 4 + 2 * who + 1] = loop_cycles;
      if (who == 1 && b < slots / 16) {       // producer sections: [0] waiting for gathers + ds_write issue, [1] gather issue
        out[(size_t)slots + (size_t)slots / 2 + (size_t)slots / 4 + (size_t)b * 2] = sec_acc[0] | (sec_acc[2] << 32);   // [2]: pure wait for the gathers
        out[(size_t)slots + (size_t)slots / 2 + (size_t)slots / 4 + (size_t)b * 2 + 1] = sec_acc[1];
      }
    }
  }
  // per-tile timeline of the persistent kernels (second half of the buffer: 2 + 32 words per block): mark(k) = 100 MHz ticks since begin();
  // word 0 = begin() on the chip-wide 100 MHz clock (start skew between blocks), word 1 = hardware id
  __device__ __forceinline__ void mark(int k) {
    const int b = blockIdx.x;
    if (out && threadIdx.x == 0 && b < slots / 36 && k < 32) {
      unsigned long long* tl = out + 2 * (size_t)slots / 2 + (size_t)b * 34;     // second half of the buffer
      tl[2 + k] = __builtin_amdgcn_s_memrealtime() - r0;
      if (k == 0) { tl[0] = r0; tl[1] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }   // HW_REG_HW_ID, 32 bits
    }
  }
#else
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void end() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void phase(int) {}
  __device__ __forceinline__ void bar_begin() {}
  __device__ __forceinline__ void bar_end() {}
  __device__ __forceinline__ void bar_init() {}
  __device__ __forceinline__ void bar_flush(int, unsigned long long) {}
  __device__ __forceinline__ void sec_begin() {}
  __device__ __forceinline__ void sec_end(int) {}
#endif
};

// Loader concept (per-thread state of a producer thread, constructed with its loader-thread id 0..255):
//   static constexpr bool KMAJOR; static constexpr int ROWS;
//   __device__ void load_next(float4 (&v)[ROWS/32]);   // gathers the next k-tile (sequential) into registers
//   __device__ void transform(float4 (&v)[ROWS/32]);   // applied to the registers of the last load_next before the ds_write
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void igemm_produce(LA& la, LB& lb, int ktiles, float* smem, int tid, ClockStamp cs = ClockStamp{nullptr, 0}) {
  using IA = LdsImage<Cfg::BM, LA::KMAJOR, Cfg::SWZ>;
  using IB = LdsImage<Cfg::BN, LB::KMAJOR, Cfg::SWZ>;
  static_assert(LA::ROWS == Cfg::BM && LB::ROWS == Cfg::BN, "loader/tile mismatch");
  float* As = smem;
  float* Bs = smem + 2 * IA::FLOATS;
  // The producers outrank the consumers in the SIMD's issue arbitration (r03).  They issue ~100 vector instructions per k-tile
  // (addresses, selects, eight gathers, eight ds_writes); at priority 0 under priority-2 consumers, whose next MFMA is pending
  // almost every cycle, each of those got about one issue slot per MFMA interval: stamps showed a producer wave 72-81 % of its
  // k-tile inside its eight ds_writes (0.1 % waiting for its gathers) and the consumers 16-30 % of theirs in the hand-over barrier,
  // waiting for it.  With the producers on top their instructions go out at once and cost the consumers next to nothing: D3
  // forward 130.8 -> 140.2 TFLOP/s, D4 grad-input 133.5 -> 139.5, weight gradients +2 %; consumers at 0 / 1 / 2 make no
  // difference, equal priorities lose the gain (scripts/conv_microbench.py on builds with -DPCG_CONSUMER_PRIO / -DPCG_PRODUCER_PRIO).
  if (PCG_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(PCG_PRODUCER_PRIO);
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
  cs.bar_init();
#ifdef PCG_CLOCK_STAMP
  const unsigned long long loop_t0 = __builtin_amdgcn_s_memtime();
#endif
  // invariant at the top of iteration kt: set (kt+1)&1 holds tile kt+1 (in flight or landed), set kt&1 holds tile kt+2
  int kt = 0;
  // Steady state WITHOUT conditionals (r03).  With the refills behind `if (kt + 3 < ktiles)` the compiler's wait-count pass lost
  // track of what is in flight at the control-flow joins and put `s_waitcnt vmcnt(7) ... vmcnt(0)` in front of the eight ds_writes
  // of a tile: the write of tile t+1 then waited for the gathers of tile t+2 as well — one k-tile of prefetch distance, not
  // two, and in-kernel stamps showed the consumers 26-28 % of their loop in the hand-over barrier while the producers waited
  // there 3-9 %.  Straight-line code gives the pass exact counts: vmcnt(8+..) leaves the younger tile's eight gathers in flight.
#ifndef PCG_OLD_PRODUCER_LOOP   // (A/B builds)
  for (; kt + 4 < ktiles; kt += 2) {
    cs.sec_begin();
#ifdef PCG_CLOCK_STAMP
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (stamp build only) the older tile has landed: separates waiting from storing
    cs.sec_end(2);
#endif
    IA::store(As + IA::FLOATS, ra[1], tid);
    IB::store(Bs + IB::FLOATS, rb[1], tid);
    cs.sec_end(0);
    la.load_next(ra[1]); lb.load_next(rb[1]);
    cs.sec_end(1);
    cs.bar_begin(); lds_barrier(); cs.bar_end();
    cs.sec_begin();
#ifdef PCG_CLOCK_STAMP
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    cs.sec_end(2);
#endif
    IA::store(As, ra[0], tid);
    IB::store(Bs, rb[0], tid);
    cs.sec_end(0);
    la.load_next(ra[0]); lb.load_next(rb[0]);
    cs.sec_end(1);
    cs.bar_begin(); lds_barrier(); cs.bar_end();
  }
#endif
  for (; kt + 1 < ktiles; kt += 2) {          // the last (up to four) k-tiles: the same steps, each behind its bound
    // kt even: tile kt+1 is in set 1 -> stage 1; refill set 1 with tile kt+3
    la.transform(ra[1]); lb.transform(rb[1]);
    IA::store(As + IA::FLOATS, ra[1], tid);
    IB::store(Bs + IB::FLOATS, rb[1], tid);
    if (kt + 3 < ktiles) { la.load_next(ra[1]); lb.load_next(rb[1]); }
    cs.bar_begin(); lds_barrier(); cs.bar_end();
    // kt+1 odd: tile kt+2 is in set 0 -> stage 0; refill set 0 with tile kt+4
    if (kt + 2 < ktiles) {
      la.transform(ra[0]); lb.transform(rb[0]);
      IA::store(As, ra[0], tid);
      IB::store(Bs, rb[0], tid);
    }
    if (kt + 4 < ktiles) { la.load_next(ra[0]); lb.load_next(rb[0]); }
    cs.bar_begin(); lds_barrier(); cs.bar_end();
  }
  if (kt < ktiles) lds_barrier();   // odd ktiles: the last iteration has nothing left to stage
#ifdef PCG_CLOCK_STAMP
  cs.bar_flush(1, __builtin_amdgcn_s_memtime() - loop_t0);
#endif
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

// Producer by LDS-DMA (r03).  In-kernel stamps of the register-staged producers (scripts/conv_microbench.py --clock) showed what they
// spend their k-tile on: 0.1 % waiting for their gathers, 11-14 % issuing the next ones — and 72-81 % inside their eight
// ds_write_b128, behind consumers whose fp32 MFMAs own the register file's read ports almost every cycle; the consumers in turn
// sat 16-30 % of their loop in the hand-over barrier waiting for those writes.  `buffer_load_dwordx4 ... lds` moves a gathered
// 16-byte chunk from memory straight into LDS (lane l of the wave lands at M0 + 16*l; a lane whose offset is out of range lands
// as zeros — scripts/probes/ldsdma_probe.hip): no VGPR staging, no ds_write, no producer-side register traffic at all.
//   * the image must be lane-linear: the unpadded K-major image (32 floats per row); one wave-instruction fills eight rows.  Its
//     XOR swizzle (conflict-free ds_read_b128 fragments) moves to the SOURCE side: the loaders are built with src_swz, the lane at
//     chunk position c of row r fetches logical chunk c ^ ((r >> 1) & 7) — the involution LdsImage::frag applies when reading.
//   * two stages; tile t+1 is issued into stage (t+1)&1 right after barrier t (its last readers retired their ds_reads before
//     that barrier) and must have landed — the issuing wave's vmcnt(0) — before barrier t+1, after which the consumers read it.
//     One k-tile (1.7-3.5 us of MFMAs) is several loaded L2 round trips.
// Loaders need next_offsets() (conv_loaders.h); K-major operands only (forward: both; grad-input: the dy operand).
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void igemm_produce_dma(LA& la, LB& lb, int ktiles, float* smem, int tid, ClockStamp cs = ClockStamp{nullptr, 0}) {
  using IA = LdsImage<Cfg::BM, true, true>;
  using IB = LdsImage<Cfg::BN, true, true>;
  static_assert(LA::KMAJOR && LB::KMAJOR && !LA::XFORM && !LB::XFORM, "LDS-DMA staging: plain K-major operands");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* As = smem;
  float* Bs = smem + 2 * IA::FLOATS;
  if (PCG_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(PCG_PRODUCER_PRIO);
  auto issue = [&](int stage) {          // rows 8*wave + 32*p .. +7 of each image: one 1 KB wave-instruction per p
    uint32_t oa[IA::NV], ob[IB::NV];
    la.next_offsets(oa); lb.next_offsets(ob);
#pragma unroll
    for (int p = 0; p < IA::NV; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(la.rs, (lds_ptr_t)(As + stage * IA::FLOATS + (8 * wave + 32 * p) * IG_BK), 16, oa[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < IB::NV; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lb.rs, (lds_ptr_t)(Bs + stage * IB::FLOATS + (8 * wave + 32 * p) * IG_BK), 16, ob[p], 0, 0, 0);
  };
  if (ktiles > 0) issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();                           // barrier 0: stage 0 is ready
  cs.bar_init();
#ifdef PCG_CLOCK_STAMP
  const unsigned long long loop_t0 = __builtin_amdgcn_s_memtime();
#endif
  for (int kt = 0; kt < ktiles; ++kt) {
    cs.sec_begin();
    if (kt + 1 < ktiles) issue((kt + 1) & 1);
    cs.sec_end(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt+1 has landed in LDS
    cs.sec_end(2);
    cs.bar_begin(); lds_barrier(); cs.bar_end();          // barrier kt+1: stage (kt+1)&1 handed to the consumers, stage kt&1 handed back
  }
#ifdef PCG_CLOCK_STAMP
  cs.bar_flush(1, __builtin_amdgcn_s_memtime() - loop_t0);
#endif
}

// Consumer: fragments are double-buffered in registers so that the LDS read of k-group g+1 is in flight under the
// 16 MFMAs of k-group g; the hand-over barrier of the k-tile sits BEFORE its last k-group (whose operands are already
// in registers), and the first fragments of the next tile are fetched right behind it — no LDS latency is exposed at
// the tile boundary.  (s_setprio 2 here dates from r01; what matters is that the PRODUCERS rank above it — igemm_produce.)
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
  __builtin_amdgcn_s_setprio(PCG_CONSUMER_PRIO);
  cs.begin(); cs.phase(1); cs.bar_init();
#ifdef PCG_CLOCK_STAMP
  const unsigned long long loop_t0 = __builtin_amdgcn_s_memtime();
#endif
  fetch(As, Bs, 0, 0);
  int cur = 0;
#if PCG_CONSUMER_ALTERNATE
  // fairness between the two workgroups of a CU: the arbiter breaks priority ties by age, so the older workgroup's
  // consumers win every contested MFMA slot and finish early, and the younger one finishes alone at a fraction of the pipe.
  // Alternate the priority k-tile by k-tile, in opposite phase for the two workgroups (block b and b + #CUs usually share a CU).
  // Measured (r03): median block of D3 fwd finishes at 212 us instead of 200 us, the last one at 234 instead of 237; whole DCGAN step
  // -0.4 % (10.86 -> 10.80 ms).  PCG_CONSUMER_ALTERNATE=0 builds the fixed-priority loop.
  const int alt_phase = (int)((blockIdx.x / PCG_CONSUMER_ALTERNATE) & 1u);
#endif
  for (int kt = 0; kt < ktiles; ++kt) {
#if PCG_CONSUMER_ALTERNATE
    if ((kt + alt_phase) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
#endif
    const float* as = As + cur * IA::FLOATS;
    const float* bs = Bs + cur * IB::FLOATS;
#pragma unroll
    for (int ks = 0; ks < KG - 1; ++ks) {
      fetch(as, bs, ks + 1, (ks + 1) & 1);
      mma(ks & 1);
    }
    cs.bar_begin();
    lds_barrier();  // barrier kt+1: all reads of stage cur have retired (lgkmcnt(0)); stage cur^1 is ready
    cs.bar_end();
    cur ^= 1;
    if (kt + 1 < ktiles) fetch(As + cur * IA::FLOATS, Bs + cur * IB::FLOATS, 0, KG & 1);
    mma((KG - 1) & 1);
  }
  cs.end(); cs.phase(2);
#ifdef PCG_CLOCK_STAMP
  cs.bar_flush(0, __builtin_amdgcn_s_memtime() - loop_t0);
#endif
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
//   EPI_ADD    aux = an addend of the output's shape (may BE the output: in-place accumulation).  v += aux   — skip connections.
//              With delta2_bytes != 0 (r04): aux2 = the activated output a of the layer below, v = (v + aux) * (aux2 > 0 ? 1 : neg) — the
//              last skip-add of a residual chain followed by the entry convolution's LeakyReLU backward (models/generator.py:76)
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
  int no_addend;            // EPI_ADDSUM without an addend (r04): the aux resource is empty, its loads return 0 — the column sums of a plain
                            // grad-input for a BatchNorm whose gradient arrives scaled (CounteRGAN: conv_mid's grad-input above the last block's bn2)
  int group_rows;           // > 0: grouped launch (r04) — G independent batches side by side along M, `group_rows` rows each (per sub-pixel
                            // phase for a grad-input launch; a multiple of the tile height): mean / invstd are [G][N], the tile's group
                            // picks the row (the caller passes the offset to igemm_store_tile)
};

// the tile's output store: plain write-back stores.  Measured r03 (whole DCGAN step, two rounds each): non-temporal stores 11.13 vs
// 11.11 ms, write-through (sc1) 11.15 — the dirty lines the kernel boundary has to write back are not what a launch's tail costs.
__device__ __forceinline__ void epi_store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <class Cfg, class RowBase>
__device__ __forceinline__ void igemm_store_tile(f32x16 (&acc)[Cfg::TM][Cfg::TN], float* smem, int n_block, int N,
                                                 const float* bias, RowBase row_base, double* stat_row = nullptr, int act = PCG_ACT_NONE,
                                                 float slope = 0.f, const EpiAux* epi = nullptr, int stat_goff = 0) {
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
        epi_store4(dst + wn * Cfg::WTN + 4 * cq, v);
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
    if (emode == EPI_BNBWD && nok) {     // stat_goff: this tile's group's row of the [G][N] statistics (0 for ordinary launches)
      mu = *reinterpret_cast<const float4*>(epi->mean + stat_goff + n);
      is = *reinterpret_cast<const float4*>(epi->invstd + stat_goff + n);
      const float4 ga = *reinterpret_cast<const float4*>(epi->gamma + n), be = *reinterpret_cast<const float4*>(epi->beta + n);
      bn_fold(ga.x, be.x, mu.x, is.x, sc.x, sh.x); bn_fold(ga.y, be.y, mu.y, is.y, sc.y, sh.y);
      bn_fold(ga.z, be.z, mu.z, is.z, sc.z, sh.z); bn_fold(ga.w, be.w, mu.w, is.w, sc.w, sh.w);
    }
    if (emode == EPI_ADDSUM && nok) {
      mu = *reinterpret_cast<const float4*>(epi->mean + n);
      is = *reinterpret_cast<const float4*>(epi->invstd + n);
    }
    // The aux reads of a CHUNK of row groups first (independent of the LDS reads), then the arithmetic; a scheduling fence between
    // chunks.  Reading all 16 row groups of the 128x128 tile at once (r02) took 64 registers: 7 VGPRs spilled in its grad-input
    // kernel; two chunks of 8: none (scripts/kernel_resources.py).  The three-per-CU 128x64 configuration (80 registers, 8 row
    // groups) keeps ONE chunk — halving it made the allocator spill more, not less (57 / 65 against 11 / 17 now, all outside the
    // main loop; r02: 37 / 18).
    constexpr int NK = Cfg::WTM / RPI, CH = (NK >= 8 && Cfg::MINW < 6) ? NK / 2 : NK;
#pragma unroll
    for (int k0 = 0; k0 < NK; k0 += CH) {
    float4 u[CH];
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      float* dst = row_base(wm * Cfg::WTM + r0 + RPI * (k0 + kk));
      u[kk] = (dst && nok && !epi->no_addend) ? *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(dst + wn * Cfg::WTN + 4 * cq) + delta)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (emode == EPI_ADDSUM) {   // wave-uniform: its own loop (the second aux tensor is read next to each row's arithmetic)
      const int64_t delta2 = epi->delta2_bytes;
      const double sc2 = (double)neg;
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) {
        const int row = r0 + RPI * (k0 + kk);
        float* dst = row_base(wm * Cfg::WTM + row);
        if (dst && nok) {
          float4 v = *reinterpret_cast<const float4*>(reg + row * LDW + 4 * cq);
          const float4 z = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(dst + wn * Cfg::WTN + 4 * cq) + delta2);
          v.x += u[kk].x; v.y += u[kk].y; v.z += u[kk].z; v.w += u[kk].w;
          epi_store4(dst + wn * Cfg::WTN + 4 * cq, v);
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
    for (int kk = 0; kk < CH; ++kk) {
      const int row = r0 + RPI * (k0 + kk);
      float* dst = row_base(wm * Cfg::WTM + row);
      if (dst && nok) {
        float4 v = *reinterpret_cast<const float4*>(reg + row * LDW + 4 * cq);
        // same expression as bn_bwd_apply / FnBnBwd use for the recomputed BatchNorm output (sc = 1, sh = 0 for EPI_MASK)
        if (emode == EPI_ADD) {   // wave-uniform
          v.x += u[kk].x; v.y += u[kk].y; v.z += u[kk].z; v.w += u[kk].w;
          if (epi->delta2_bytes != 0) {   // wave-uniform: the sum times the activation derivative of the layer below
            const float4 a2 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(dst + wn * Cfg::WTN + 4 * cq) + epi->delta2_bytes);
            v.x *= a2.x > 0.f ? 1.f : neg; v.y *= a2.y > 0.f ? 1.f : neg; v.z *= a2.z > 0.f ? 1.f : neg; v.w *= a2.w > 0.f ? 1.f : neg;
          }
        } else {
          const float4 pre = make_float4(fmaf(u[kk].x, sc.x, sh.x), fmaf(u[kk].y, sc.y, sh.y), fmaf(u[kk].z, sc.z, sh.z), fmaf(u[kk].w, sc.w, sh.w));
          v.x *= pre.x > 0.f ? 1.f : neg; v.y *= pre.y > 0.f ? 1.f : neg; v.z *= pre.z > 0.f ? 1.f : neg; v.w *= pre.w > 0.f ? 1.f : neg;
        }
        epi_store4(dst + wn * Cfg::WTN + 4 * cq, v);
        if (want_sums) {
          s1[0] += (double)v.x; s1[1] += (double)v.y; s1[2] += (double)v.z; s1[3] += (double)v.w;
          s2[0] = fma((double)v.x, (double)((u[kk].x - mu.x) * is.x), s2[0]); s2[1] = fma((double)v.y, (double)((u[kk].y - mu.y) * is.y), s2[1]);
          s2[2] = fma((double)v.z, (double)((u[kk].z - mu.z) * is.z), s2[2]); s2[3] = fma((double)v.w, (double)((u[kk].w - mu.w) * is.w), s2[3]);
        }
      }
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    }   // chunk
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

// ======================================================================================================================
// Persistent, tile-pipelined form (r03).  In-kernel stamps (scripts/conv_microbench.py --clock) showed where the time of the
// one-tile-per-workgroup kernels goes: the chip holds 2.37-2.40 GHz inside them and the main loops run at 93-96 % of the MFMA rate,
// but every tile pays 16-19 us OUTSIDE its main loop — workgroup dispatch, loader set-up, the first gathers' round trip, the LDS-
// staged epilogue, the final write burst — which is 7 % of a 128x128 tile with K = 2048 and as long as the main loop itself for a
// 128x64 tile with K = 512.  Here a workgroup is resident for the whole launch and walks a strided list of tiles; the k-tiles of
// all its tiles form ONE stream through the same two LDS stages and the same barrier protocol:
//   producers  run ahead across tile boundaries (the next tile's first k-tiles are gathered and staged while the consumers finish
//              the current one), re-creating their loader state when the stream crosses into the next tile;
//   consumers  store a finished tile STRAIGHT from the accumulator registers (column on the lane, rows in the registers: one
//              128-byte row segment per half-wave per store) — no LDS staging, so the stages stay with the producers — zero the
//              accumulators and continue with the fragments that are already waiting.
// MEASURED (MI355X, scripts/conv_microbench.py --ab persistent=0,1 and --clock --timeline): NOT faster.  With the
// plain epilogue it ties the one-tile-per-workgroup kernels on the forward shapes (D2 260.6 vs 261.1 us, D4 +4.6 %) and loses 3-8 %
// on the grad-input shapes; with the full epilogue compiled in, the whole DCGAN step is 11.20 vs 10.96 ms.  Why: resident
// workgroups that all walk equal tiles stay in LOCKSTEP — every tile boundary is chip-wide, all consumers leave the matrix pipe
// together and all write their tiles together (a 32 MB burst: 8-9 us per 128x128 tile boundary), whereas the dispatcher's own
// round-robin of short-lived workgroups desynchronises after the first round and hides one workgroup's prologue / epilogue
// behind its CU neighbour's main loop; and a static tile walk ends with its slowest workgroup (main-loop spread 200-252 us).
// Kept as an opt-in experiment (-DPCG_PERSISTENT_KERNELS: `make -C csrc lean`, pcg_tune_set("persistent", 1)); the shipped
// library does not compile it.
// ======================================================================================================================
#ifdef PCG_PERSISTENT_KERNELS

// Walk of one workgroup over the launch's work items (tiles, or (tile, K-slice) pairs): the `total` items are cut into 8 contiguous
// runs, one per XCD (blocks b and b + 8 share an XCD and its L2), and the G/8 workgroups of an XCD take the items of their run
// round-robin — at any time the workgroups of one XCD work on neighbouring tiles (shared halo rows / weight panels in that L2).
struct TileWalk {
  uint32_t base, count, q, step;   // this workgroup visits base + q, base + q + step, ... < base + count
  __device__ __forceinline__ TileWalk(uint32_t total) {
    const uint32_t w = blockIdx.x, G = gridDim.x;
    if (G & 7u) {                              // not a multiple of 8 (tiny launches): plain striding
      base = 0; count = total; q = w; step = G;
    } else {
      const uint32_t x = w & 7u, qq = total >> 3, r = total & 7u;
      base = x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq;
      count = qq + (x < r ? 1u : 0u);
      q = w >> 3; step = G >> 3;
    }
  }
  __device__ __forceinline__ uint32_t ntiles() const { return q < count ? (count - q + step - 1) / step : 0u; }
  __device__ __forceinline__ uint32_t item(uint32_t i) const { return base + q + i * step; }
};

// Producer side.  Src concept: `la`, `lb` (loaders of the CURRENT tile), `int n` (its k-tiles, >= 1), `bool next_tile()` (advance to
// this workgroup's next tile and rebuild la / lb / n; false when there is none).  S = k-tiles of the whole stream.
template <class Cfg, class Src>
__device__ __forceinline__ void igemm_produce_stream(Src& src, int S, float* smem, int tid) {
  using LA = decltype(src.la);
  using LB = decltype(src.lb);
  using IA = LdsImage<Cfg::BM, LA::KMAJOR, Cfg::SWZ>;
  using IB = LdsImage<Cfg::BN, LB::KMAJOR, Cfg::SWZ>;
  static_assert(LA::ROWS == Cfg::BM && LB::ROWS == Cfg::BN, "loader/tile mismatch");
  float* As = smem;
  float* Bs = smem + 2 * IA::FLOATS;
  int left = src.n;                            // k-tiles of the current tile not yet gathered
  auto issue = [&](float4 (&ra)[IA::NV], float4 (&rb)[IB::NV]) {
    if (left == 0) { src.next_tile(); left = src.n; }        // only called while the stream has elements left
    src.la.load_next(ra); src.lb.load_next(rb);
    --left;
  };
  if constexpr (Cfg::PF == 2 && !LA::XFORM && !LB::XFORM) {
    float4 ra[2][IA::NV], rb[2][IB::NV];
    if (S > 0) {
      issue(ra[0], rb[0]);
      if (S > 1) issue(ra[1], rb[1]);
      IA::store(As, ra[0], tid);
      IB::store(Bs, rb[0], tid);
      if (S > 2) issue(ra[0], rb[0]);
    }
    lds_barrier();
    int g = 0;
    for (; g + 1 < S; g += 2) {
      IA::store(As + IA::FLOATS, ra[1], tid);
      IB::store(Bs + IB::FLOATS, rb[1], tid);
      if (g + 3 < S) issue(ra[1], rb[1]);
      lds_barrier();
      if (g + 2 < S) {
        IA::store(As, ra[0], tid);
        IB::store(Bs, rb[0], tid);
      }
      if (g + 4 < S) issue(ra[0], rb[0]);
      lds_barrier();
    }
    if (g < S) lds_barrier();
  } else {
    float4 ra[IA::NV], rb[IB::NV];
    if (S > 0) {
      issue(ra, rb);
      src.la.transform(ra); src.lb.transform(rb);
      IA::store(As, ra, tid);
      IB::store(Bs, rb, tid);
      if (S > 1) issue(ra, rb);
    }
    lds_barrier();
    int nxt = 1;
    for (int g = 0; g < S; ++g) {
      if (g + 1 < S) {
        src.la.transform(ra); src.lb.transform(rb);      // the state of the loader that issued this k-tile: transform precedes the next issue
        IA::store(As + nxt * IA::FLOATS, ra, tid);
        IB::store(Bs + nxt * IB::FLOATS, rb, tid);
      }
      if (g + 2 < S) issue(ra, rb);
      lds_barrier();
      nxt ^= 1;
    }
  }
}

// Consumer side.  ktiles_of(i) = k-tiles of this workgroup's i-th tile (>= 1); epilogue(i, acc) stores it (registers -> global).
template <class Cfg, bool AK, bool BK_, class KtOf, class Epi>
__device__ __forceinline__ void igemm_consume_stream(int ntiles, KtOf ktiles_of, Epi epilogue, const float* smem, ClockStamp cs = ClockStamp{nullptr, 0}) {
  using IA = LdsImage<Cfg::BM, AK, Cfg::SWZ>;
  using IB = LdsImage<Cfg::BN, BK_, Cfg::SWZ>;
  constexpr int KG = IG_BK / 8;
  const float* As = smem;
  const float* Bs = smem + 2 * IA::FLOATS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;
  const int arow = wm * Cfg::WTM, brow = wn * Cfg::WTN;
  f32x16 acc[Cfg::TM][Cfg::TN];
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
  lds_barrier();                                 // barrier 0: the stream's first k-tile is staged
  if (ntiles <= 0) return;
  __builtin_amdgcn_s_setprio(2);
  cs.begin();
  fetch(As, Bs, 0, 0);
  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int n = ktiles_of(t);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < n; ++kt) {
      const float* as = As + cur * IA::FLOATS;
      const float* bs = Bs + cur * IB::FLOATS;
#pragma unroll
      for (int ks = 0; ks < KG - 1; ++ks) {
        fetch(as, bs, ks + 1, (ks + 1) & 1);
        mma(ks & 1);
      }
      lds_barrier();                             // all reads of stage cur retired; stage cur^1 (the stream's next k-tile) is ready
      cur ^= 1;
      if (kt + 1 < n) fetch(As + cur * IA::FLOATS, Bs + cur * IB::FLOATS, 0, KG & 1);
      mma((KG - 1) & 1);
    }
    __builtin_amdgcn_s_setprio(0);
    cs.mark(2 * t);
    epilogue(t, acc);                            // registers -> global; the LDS stages stay with the producers
    cs.mark(2 * t + 1);
    __builtin_amdgcn_s_setprio(2);
    // the next tile's first fragments only now: they would be 16 more live registers across the epilogue (one exposed LDS read per tile)
    if (t + 1 < ntiles) fetch(As + cur * IA::FLOATS, Bs + cur * IB::FLOATS, 0, KG & 1);
  }
  cs.end();
  __builtin_amdgcn_s_setprio(0);
}

// Epilogue straight from the accumulator registers: element (i, j, r) of lane (li, lh) is row 32i + acc_row(r, lh), column 32j + li
// of the wave tile, so a store instruction writes two 128-byte row segments (one per half-wave) and every per-column constant
// (bias, BatchNorm mean / invstd / scale / shift) is ONE value per lane and j.  Same arithmetic per element as igemm_store_tile;
// the column sums (fused statistics / BatchNorm-backward sums) are taken per lane over its rows in fp64 and the two half-waves
// added by one shuffle.  Addressing is by 32-bit byte offsets into raw buffer resources of the output (and of the aux tensors of
// the backward epilogues, which have the output's shape): row_off(row) = offset of the tile row's column n_block, or OOB_OFF for a
// row outside the problem — such stores are dropped and such loads return zero, no branches.
struct EpiBufs { rsrc_t out, aux, aux2; };
__device__ __forceinline__ EpiBufs make_epi_bufs(float* out, uint32_t out_bytes, const EpiAux& e) {
  EpiBufs b;
  b.out = make_rsrc(out, out_bytes);
  b.aux = make_rsrc(reinterpret_cast<const char*>(out) + e.delta_bytes, (e.mode != EPI_NONE && !e.no_addend) ? out_bytes : 0u);
  b.aux2 = make_rsrc(reinterpret_cast<const char*>(out) + e.delta2_bytes, e.mode == EPI_ADDSUM ? out_bytes : 0u);
  return b;
}

// Row addressing policies of the register epilogue.  An element's address is (per-lane VGPR offset) + (wave-uniform SGPR offset):
// gfx950 range-checks their SUM against num_records (scripts/probes/soffset_probe.hip), so for a row-major [M][N] output the rows
// beyond M fall out of range by themselves and the sixteen row steps of an accumulator are sixteen SGPR constants — one VGPR of
// addressing per accumulator instead of sixteen.  begin(lane_row, colb) is called once per accumulator with this lane's first row
// (wm*WTM + 32*i + 4*lh) and column byte offset inside the tile row; c = (r & 3) + 8 * (r >> 2) is a compile-time constant.
struct RowsAffine {            // consecutive rows of a row-major [M][N] matrix (forward, split-K slabs)
  int M, m_block; uint32_t n4, nb4;      // n4 = 4*N, nb4 = 4*n_block
  uint32_t vbase; int left;
  __device__ __forceinline__ void begin(int lane_row, uint32_t colb) {
    const int m = m_block + lane_row;
    vbase = (uint32_t)m * n4 + nb4 + colb;
    left = M - m;
  }
  __device__ __forceinline__ uint32_t voff(int) const { return vbase; }
  __device__ __forceinline__ uint32_t soff(int c) const { return (uint32_t)c * n4; }
  __device__ __forceinline__ bool valid(int c) const { return c < left; }
};
struct RowsTable {             // rows map to arbitrary pixels: a table in LDS of byte offsets (or OOB_OFF), one per tile row (grad-input)
  const uint32_t* table; uint32_t nb4;
  const uint32_t* p; uint32_t colb;
  __device__ __forceinline__ void begin(int lane_row, uint32_t colb_) { p = table + lane_row; colb = nb4 + colb_; }
  __device__ __forceinline__ uint32_t voff(int c) const { return p[c] + colb; }      // OOB_OFF + (< 2^31) stays out of range
  __device__ __forceinline__ uint32_t soff(int) const { return 0u; }
  __device__ __forceinline__ bool valid(int c) const { return p[c] < OOB_OFF; }
};
__device__ __forceinline__ float buf_load1s(rsrc_t r, uint32_t voff, uint32_t soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ void buf_store1s(rsrc_t r, uint32_t voff, uint32_t soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0); }
// fence between the accumulators of the epilogue: nothing is scheduled or kept alive across it (the compiler otherwise hoists the
// address arithmetic and the aux reads of all accumulators to the front — seen: 250+ spilled VGPRs)
__device__ __forceinline__ void epi_fence() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// What makes this epilogue fast or slow is the in-order vmcnt: waiting for ANY load means waiting for every store issued before
// it, and a tile's stores complete only after a trip through a memory system that every workgroup is writing into.  So: all
// per-column constants are loaded before the first store, and the aux reads of accumulator k+1 are issued BEFORE the stores of
// accumulator k (the counted wait then leaves exactly those stores in flight).  The fp64 column sums sit behind a wave-uniform
// branch; a lane whose column lies outside N is masked off for the whole accumulator.
template <class Cfg, class Rows>
__device__ __forceinline__ void igemm_store_regs(f32x16 (&acc)[Cfg::TM][Cfg::TN], int n_block, int N, const float* bias, Rows rows,
                                                 const EpiBufs& eb, double* stat_row, int act, float slope, const EpiAux* epi) {
  // Opaque copy of the thread index: everything below that depends only on it is loop-invariant over the workgroup's tiles, and the
  // compiler would hoist it out of the tile loop and keep it alive — i.e. spill it — across the main loop.
  int tix = threadIdx.x;
  asm volatile("" : "+v"(tix));
  const int lane = tix & 63, wave = tix >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N, li = lane & 31, lh = lane >> 5;
  const int emode = epi ? epi->mode : EPI_NONE;       // wave-uniform
  const bool want_sums = stat_row != nullptr;         // wave-uniform
  const float neg = epi ? epi->neg : 1.f;
  constexpr int NA = Cfg::TM * Cfg::TN;               // accumulators of the wave tile, visited j-major: k = j * TM + i
  int ncol[Cfg::TN];
  float bv[Cfg::TN], mu[Cfg::TN], is[Cfg::TN], sc[Cfg::TN], sh[Cfg::TN];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {                 // every per-column load before the first store
    const int n = n_block + wn * Cfg::WTN + 32 * j + li;
    const bool ok = n < N;
    ncol[j] = ok ? n : -1;
    bv[j] = (bias && ok) ? bias[n] : 0.f;
    mu[j] = 0.f; is[j] = 1.f; sc[j] = 1.f; sh[j] = 0.f;
    if ((emode == EPI_BNBWD || emode == EPI_ADDSUM) && ok) { mu[j] = epi->mean[n]; is[j] = epi->invstd[n]; }
    if (emode == EPI_BNBWD && ok) bn_fold(epi->gamma[n], epi->beta[n], mu[j], is[j], sc[j], sh[j]);
  }
  auto rows_of = [&](int k) {
    Rows rw = rows;
    rw.begin(wm * Cfg::WTM + 32 * (k % Cfg::TM) + 4 * lh, (uint32_t)(4 * (wn * Cfg::WTN + 32 * (k / Cfg::TM) + li)));
    return rw;
  };
#ifdef PCG_EPI_LEAN   // measurement-only build: the plain epilogue without statistics, nothing else compiled in
  {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int j = k / Cfg::TM, i = k % Cfg::TM;
      if (ncol[j] >= 0) {
        const Rows rw = rows_of(k);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r] + bv[j];
          if (act != PCG_ACT_NONE) v = act_neg_scale(v, slope);
          buf_store1s(eb.out, rw.voff(c), rw.soff(c), v);
        }
      }
      epi_fence();
    }
    return;
  }
#endif
  double s1 = 0.0, s2 = 0.0;
  auto flush_sums = [&](int j) {
    double* pr = stat_row + (size_t)wm * 2 * N;       // [wm][2][N] inside this tile row's slot
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lh == 0 && ncol[j] >= 0) { pr[ncol[j]] = s1; pr[N + ncol[j]] = s2; }
    s1 = 0.0; s2 = 0.0;
  };
  if (emode == EPI_NONE) {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int j = k / Cfg::TM, i = k % Cfg::TM;
      if (ncol[j] >= 0) {
        const Rows rw = rows_of(k);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          v[r] = acc[i][j][r] + bv[j];
          if (act != PCG_ACT_NONE) v[r] = act_neg_scale(v[r], slope);
          buf_store1s(eb.out, rw.voff(c), rw.soff(c), v[r]);
        }
        if (want_sums) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { const double d = rw.valid((r & 3) + 8 * (r >> 2)) ? (double)v[r] : 0.0; s1 += d; s2 = fma(d, d, s2); }
        }
      }
      if (want_sums && i == Cfg::TM - 1) flush_sums(j);
      epi_fence();
    }
    return;
  }
  // backward-pass epilogues: aux reads software-pipelined one accumulator ahead of the stores
  float u[2][16];
  auto load_aux = [&](int k, float (&dst)[16], rsrc_t rs) {
    if (ncol[k / Cfg::TM] >= 0) {
      const Rows rw = rows_of(k);
#pragma unroll
      for (int r = 0; r < 16; ++r) { const int c = (r & 3) + 8 * (r >> 2); dst[r] = buf_load1s(rs, rw.voff(c), rw.soff(c)); }
    }
  };
  load_aux(0, u[0], eb.aux);
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int j = k / Cfg::TM, i = k % Cfg::TM;
    if (k + 1 < NA) load_aux(k + 1, u[(k + 1) & 1], eb.aux);
    float (&uu)[16] = u[k & 1];
    if (ncol[j] >= 0) {
      const Rows rw = rows_of(k);
      if (emode == EPI_ADD || emode == EPI_ADDSUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float v = acc[i][j][r] + uu[r];
          buf_store1s(eb.out, rw.voff(c), rw.soff(c), v);
          acc[i][j][r] = v;                          // EPI_ADDSUM needs the sum once more
        }
        if (emode == EPI_ADDSUM && want_sums) {       // the second aux tensor (the next BatchNorm's pre-normalisation output)
          float z[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) { const int c = (r & 3) + 8 * (r >> 2); z[r] = buf_load1s(eb.aux2, rw.voff(c), rw.soff(c)); }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const double d = rw.valid((r & 3) + 8 * (r >> 2)) ? (double)neg * (double)acc[i][j][r] : 0.0;
            s1 += d; s2 = fma(d, (double)((z[r] - mu[j]) * is[j]), s2);
          }
        }
      } else {                                        // EPI_MASK (sc = 1, sh = 0) / EPI_BNBWD: the forward's own expression for the sign
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float pre = fmaf(uu[r], sc[j], sh[j]);
          v[r] = acc[i][j][r] * (pre > 0.f ? 1.f : neg);
          buf_store1s(eb.out, rw.voff(c), rw.soff(c), v[r]);
        }
        if (want_sums) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const double d = rw.valid((r & 3) + 8 * (r >> 2)) ? (double)v[r] : 0.0;
            s1 += d; s2 = fma(d, (double)((uu[r] - mu[j]) * is[j]), s2);
          }
        }
      }
    }
    if (want_sums && i == Cfg::TM - 1) flush_sums(j);
    epi_fence();
  }
}

#endif  // PCG_PERSISTENT_KERNELS

// Accumulator element (tile i,j ; register r) of lane (li,lh) sits at
//   row = 32*i + (r&3) + 8*(r>>2) + 4*lh   col = 32*j + li      (within the wave tile)

}  // namespace pcg
