// conv_igemm.hip — Conv2d / ConvTranspose2d forward, grad-input and grad-weight as fp32-MFMA implicit
// GEMMs over NHWC activations and OHWI weights (see include/pcgan_hip.h for the layer <-> op mapping).
//
//   fwd   : M = B*OH*OW        N = Cout          K = (kh,kw,ci)     A = im2col(x) gathered, B = w rows
//   dgrad : per sub-pixel phase (ih%s, iw%s):  M = B*PHh*PHw  N = Cin  K = (taps of the phase, co)
//           A = dy gathered, B = w^T slices.  For k4 s2 p1 every phase sees exactly 2x2 taps: no MAC is
//           spent on the zeros a "dilate then convolve" formulation would insert.
//   wgrad : M = Cout  N = (kh,kw,ci)  K = B*OH*OW split over gridDim.y; partial slabs reduced in a
//           fixed order by a second kernel (deterministic; also implements .grad accumulation).
//
// Replaces ATen conv forward/backward behind nn.Conv2d / nn.ConvTranspose2d at
// dconv_gan/mnist/mnist_dcgan.py:76-88,100-111 and conditional_counteRGAN/mnist/models/*.py.
#include <string.h>
#include <mutex>
#include "conv_loaders.h"
#include "thin_conv.h"

namespace pcg {
namespace {

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// role of this wave, provably wave-uniform for the compiler (scalar branch, no exec masking)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Issue priority at the edges of a workgroup's life (ConvP::edge_prio).  The main loops run producers at 3 and consumers at 1-2
// (igemm_core.h); everything outside them — loader set-up with its divisions, the first gathers, the LDS-staged epilogue — used to
// run at priority 0 and was handed an issue slot only when no wave of the CU's other workgroups wanted one: in-kernel stamps
// (r04, 3x3 64->64 forward, three workgroups per CU) read 4.0 us from entry to the first k-tile and 7.1 us for a ~150-instruction
// epilogue, 17 % of the workgroup's life.
__device__ __forceinline__ void prio_entry(const ConvP& p) { if (p.edge_prio & 1) __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ void prio_epilogue(const ConvP& p) { if (p.edge_prio & 2) __builtin_amdgcn_s_setprio(3); }

// grouped launches (EpiAux::group_rows): float offset of the tile's group inside the [G][N] BatchNorm statistics
__device__ __forceinline__ int epi_group_off(const ConvP& p, int m_block) {
  return p.epi.group_rows > 0 ? __builtin_amdgcn_readfirstlane(m_block / p.epi.group_rows) * p.N : 0;
}

// XF (here and below): the activation operand carries an input transform (ConvP::in_sc) — a separate instantiation, so the
// plain kernels pay nothing for it.
template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_fwd_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  prio_entry(p);
  const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / p.tilesN, nt = tile % p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  // split-K (few output tiles, long K): slab blockIdx.y covers k-tiles [kt_begin, kt_begin + ktiles)
  const int kt_begin = blockIdx.y * p.ktiles_per_split;
  int ktiles = p.ktiles - kt_begin;
  if (ktiles > p.ktiles_per_split) ktiles = p.ktiles_per_split;

  if (wave_id() >= 4) {  // producers
    const int tid = threadIdx.x - IG_LOADERS;
    constexpr bool DMA = Cfg::DMA && !XF;
    FwdALoader<Cfg::BM, XF> la(p, m_block, tid, DMA);
    FwdBLoader<Cfg::BN> lb(p, n_block, tid, DMA);
    if (kt_begin) { la.seek(kt_begin); lb.seek(kt_begin); }
    if constexpr (DMA) igemm_produce_dma<Cfg>(la, lb, ktiles, smem, tid, ClockStamp{p.stamps, p.stamp_slots});
    else igemm_produce<Cfg>(la, lb, ktiles, smem, tid, ClockStamp{p.stamps, p.stamp_slots});
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  ClockStamp cs{p.stamps, p.stamp_slots};
  cs.phase(0);
  igemm_consume<Cfg, true, true>(ktiles, acc, smem, cs);
  prio_epilogue(p);
  float* out = p.out + (size_t)blockIdx.y * (size_t)p.M * (size_t)p.N;
#ifdef PCG_ABL_NO_EPILOGUE   // timing-only ablation: keep one store so the accumulators stay live
  if (acc[0][0][0] == 12345.678f) out[0] = acc[0][0][1];
  return;
#endif
  igemm_store_tile<Cfg>(acc, smem, n_block, p.N, blockIdx.y == 0 ? p.bias : nullptr, [&](int row) -> float* {
    const int m = m_block + row;
    return m < p.M ? out + (size_t)m * p.N + n_block : nullptr;
  }, p.stat_partial ? p.stat_partial + (size_t)mt * Cfg::WAVES_M * 2 * p.N : nullptr, p.act, p.slope, &p.epi, epi_group_off(p, m_block));
  cs.phase(3);
}

template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_dgrad_kernel(ConvP p, DgradPhases phases) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int rowpix[Cfg::BM];
  prio_entry(p);
  // phases.interleave (all phases the same size, 1-D grid): the sub-pixel phases of one tile are neighbours in launch order and
  // on one XCD — they gather the same dy rows, which then come from HBM once and from that XCD's L2 for the other phases
  uint32_t bx = blockIdx.x, py = blockIdx.y;
  if (phases.interleave) {
    const uint32_t l = xcd_remap(blockIdx.x, gridDim.x);
    py = l % (uint32_t)phases.interleave; bx = l / (uint32_t)phases.interleave;
  }
  const PhaseInfo& f = phases.p[py];
  const int tilesM = (f.Mp + Cfg::BM - 1) / Cfg::BM;
  const uint32_t ntiles = (uint32_t)tilesM * p.tilesN;
  if (bx >= ntiles) return;  // phases can differ in size (odd IH/IW); uniform per block
  const uint32_t tile = phases.interleave ? bx : xcd_remap(bx, ntiles);
  const int mt = tile / p.tilesN, nt = tile % p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  const int ktiles = f.nth * f.ntw * ((p.Cout + IG_BK - 1) / IG_BK);

  if (wave_id() >= 4) {  // producers: also publish the output pixel of every tile row (visible after barrier 0)
    const int tid = threadIdx.x - IG_LOADERS;
    for (int r = tid; r < Cfg::BM; r += IG_LOADERS) {
      const int m = m_block + r;
      int pix = -1;
      if (m < f.Mp) {
        uint32_t t, cc, b, aa;
        f.dPHw.divmod((uint32_t)m, t, cc);
        f.dPHh.divmod(t, b, aa);
        pix = ((int)b * p.IH + (int)aa * p.stride + f.ph) * p.IW + (int)cc * p.stride + f.pw;
      }
      rowpix[r] = pix;
    }
    DgradALoader<Cfg::BM, XF> la(p, f, m_block, tid);
    DgradBLoader<Cfg::BN> lb(p, f, n_block, tid);
    igemm_produce<Cfg>(la, lb, ktiles, smem, tid, ClockStamp{p.stamps, p.stamp_slots});
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  ClockStamp cs{p.stamps, p.stamp_slots};
  cs.phase(0);
  igemm_consume<Cfg, true, false>(ktiles, acc, smem, cs);
  prio_epilogue(p);
  igemm_store_tile<Cfg>(acc, smem, n_block, p.N, p.bias, [&](int row) -> float* {
    const int pix = rowpix[row];
    return pix >= 0 ? p.out + (size_t)pix * p.Cin + n_block : nullptr;
  }, p.stat_partial ? p.stat_partial + (size_t)(f.prow0 + mt * Cfg::WAVES_M) * 2 * p.N : nullptr, p.act, p.slope, &p.epi, epi_group_off(p, m_block));
  cs.phase(3);
}

// ======================================================================================================================
// Hybrid stream-K launches (r03).  The one-tile-per-workgroup kernels above run in rounds of 512 workgroups (2 per CU); a launch
// whose tile count is not a multiple of that pays a whole extra round for the remainder — the in-kernel stamps of WGAN-GP's
// 288-tile grad-input GEMM (1024 x 4608, K = 1024) show 224 CUs finishing their single tile at 69 us and 32 CUs running two tiles
// side by side until 124 us: 66 TFLOP/s.  Here the first `dp_tiles` tiles (whole rounds) keep one workgroup each, and the k-tiles
// of the remaining `sk_tiles` tiles form ONE iteration space that is cut into `sk_blocks` equal contiguous ranges, one per
// workgroup; a range covers the tail of one tile and the head of the next (at most two segments: sk_blocks >= sk_tiles).
// A segment that is not a whole tile leaves its accumulators as a partial tile in the scratch buffer (register layout, coalesced
// 16-byte stores) and bumps the arrival counter of (tile, consumer wave); the wave that arrives LAST adds the partial tiles of its
// quadrant in K order — its own included, from memory: the order of the sum never depends on who arrives when, results are
// bit-reproducible — and runs the ordinary epilogue (bias, activation, statistics, backward riders: nothing is lost to the
// split, unlike the slab form of plan_fwd).  Counters return to zero behind the last arrival; visibility across the XCDs' L2s
// comes from the agent-scope release / acquire fences around the counter update.
// ======================================================================================================================
struct SkPlan {
  int dp_tiles, sk_tiles, sk_blocks, ktiles;   // ktiles: per tile
  float* parts;                                // [2 * sk_blocks][BM * BN]
  int* arrivals;                               // [sk_tiles][4], zero between launches
};
struct SkSeg { int tile, kt_begin, nkt, nparts, first_block; };   // tile: index among ALL tiles of the launch

// (wave-uniform values; the divisions run on the vector unit, readfirstlane puts the results back into scalar registers.
//  32-bit: the host keeps sk_tiles * ktiles * sk_blocks below 2^31)
__device__ __forceinline__ int sk_start(int g, const SkPlan& sk) {
  return __builtin_amdgcn_readfirstlane((int)((uint32_t)g * (uint32_t)(sk.sk_tiles * sk.ktiles) / (uint32_t)sk.sk_blocks));
}
__device__ __forceinline__ int sk_owner(int x, const SkPlan& sk) {     // the block whose range holds iteration x
  return __builtin_amdgcn_readfirstlane((int)((((uint32_t)x + 1u) * (uint32_t)sk.sk_blocks - 1u) / (uint32_t)(sk.sk_tiles * sk.ktiles)));
}
__device__ __forceinline__ int sk_segments(const SkPlan& sk, SkSeg& s0, SkSeg& s1) {
  const int b = blockIdx.x;
  if (b < sk.dp_tiles) { s0 = SkSeg{(int)xcd_remap((uint32_t)b, (uint32_t)sk.dp_tiles), 0, sk.ktiles, 1, 0}; return 1; }
  // neighbouring ranges (neighbouring tiles: shared operand rows) on ONE XCD, like the tiles of the data-parallel part
  const int g = (int)xcd_remap((uint32_t)(b - sk.dp_tiles), (uint32_t)sk.sk_blocks), KT = sk.ktiles;
  const int it0 = sk_start(g, sk), it1 = sk_start(g + 1, sk);
  const int j0 = __builtin_amdgcn_readfirstlane(it0 / KT), e0 = it1 < (j0 + 1) * KT ? it1 : (j0 + 1) * KT;
  auto fill = [&](SkSeg& sg, int j, int from, int to) {
    const int f = sk_owner(j * KT, sk), l = sk_owner((j + 1) * KT - 1, sk);
    sg = SkSeg{sk.dp_tiles + j, from - j * KT, to - from, l - f + 1, f};
  };
  fill(s0, j0, it0, e0);
  if (it1 <= e0) return 1;
  fill(s1, j0 + 1, e0, it1);
  return 2;
}
// Partial tiles cross XCDs (each has its own L2).  Their stores and loads carry sc1 (agent scope: written through / read past
// the non-coherent lines; raw-buffer builtins with aux 16, so the compiler counts them) instead of an agent-scope release /
// acquire fence pair around the counter, which writes back and invalidates whole caches once per wave and arrival (measured r03:
// with the fences the 288-tile GEMM went from 137 to 203 us).  Each storing wave drains its stores (vmcnt(0)) before ITS add to
// the (tile, wave) counter; the wave whose add returns nparts - 1 loads only behind that return.
// PER-ARCHITECTURE REQUIREMENT: this protocol (sc1 stores drained by vmcnt(0), then a relaxed agent-scope atomic, no fences) is
// what gfx950's memory model guarantees for agent-scope visibility; another target has to re-derive it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "stream-K partial-tile exchange (sc1 stores + relaxed agent-scope counters) is written for gfx950 only"
#endif
__device__ __forceinline__ void sk_store4(rsrc_t r, uint32_t off, float a, float b, float c, float d) {
  const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}
__device__ __forceinline__ float4 sk_load4(rsrc_t r, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// consumer waves, after the main loop of a segment: true when this wave holds the finished sum of its quadrant in `acc`.
//   g: this workgroup's range index, j: arrival-counter index of the tile, nparts / first_block: the ranges that touch the tile,
//   first_of(blk): does range blk START inside this tile (its partial is then in the range's slot 0, else in slot 1)
template <class Cfg, class FirstOf>
__device__ __forceinline__ bool sk_combine_core(float* parts, int* arrivals, int nblocks, int g, int j, int nparts, int first_block,
                                                FirstOf first_of, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  if (nparts == 1) return true;
  constexpr uint32_t TILE_BYTES = Cfg::BM * Cfg::BN * 4;
  auto slot_of = [&](int blk) { return (uint32_t)(2 * blk + (first_of(blk) ? 0 : 1)); };
  const rsrc_t rs = make_rsrc(parts, (uint32_t)(2 * nblocks) * TILE_BYTES);
  const uint32_t toff = threadIdx.x * 16u;                 // consumer threads 0..255: one 16-byte column of each 4 KB row of the partial tile
  const uint32_t mine = slot_of(g) * TILE_BYTES + toff;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        sk_store4(rs, mine + (uint32_t)(((i * Cfg::TN + jj) * 4 + q) * IG_LOADERS * 16),
                  acc[i][jj][4 * q], acc[i][jj][4 * q + 1], acc[i][jj][4 * q + 2], acc[i][jj][4 * q + 3]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's partial quadrant has left for memory
  int* cnt = arrivals + j * 4 + (int)(threadIdx.x >> 6);
  int old = 0;
  if ((threadIdx.x & 63) == 0) old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != nparts - 1) return false;
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  for (int k = 0; k < nparts; ++k) {
    const uint32_t src = slot_of(first_block + k) * TILE_BYTES + toff;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int jj = 0; jj < Cfg::TN; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = sk_load4(rs, src + (uint32_t)(((i * Cfg::TN + jj) * 4 + q) * IG_LOADERS * 16));
          acc[i][jj][4 * q] += v.x; acc[i][jj][4 * q + 1] += v.y; acc[i][jj][4 * q + 2] += v.z; acc[i][jj][4 * q + 3] += v.w;
        }
  }
  return true;
}
template <class Cfg>
__device__ __forceinline__ bool sk_combine(const SkPlan& sk, const SkSeg& sg, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  const int g = (int)xcd_remap((uint32_t)((int)blockIdx.x - sk.dp_tiles), (uint32_t)sk.sk_blocks), j = sg.tile - sk.dp_tiles, KT = sk.ktiles;
  return sk_combine_core<Cfg>(sk.parts, sk.arrivals, sk.sk_blocks, g, j, sg.nparts, sg.first_block,
                              [&](int blk) { return sk_start(blk, sk) >= j * KT; }, acc);
}

// One segment = the base kernel's body over k-tiles [kt_begin, kt_begin + nkt) of one tile.  A workgroup runs it once or twice;
// the two runs are two inlined copies, not a loop (as a loop the compiler kept both segments' state live across the body: 60-80
// spilled VGPRs and a scratch access inside the k-tile loop; the all-data-parallel cost of that kernel was +13..21 % on grad-input
// launches, scripts/probes/streamk_dp_cost.py).
template <class Cfg, bool XF>
__device__ __forceinline__ void fwd_sk_segment(const ConvP& p, const SkPlan& sk, const SkSeg& sg, float* smem) {
  const int mt = __builtin_amdgcn_readfirstlane(sg.tile / p.tilesN), nt = sg.tile - mt * p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  if (wave_id() >= 4) {  // producers
    const int tid = threadIdx.x - IG_LOADERS;
    FwdALoader<Cfg::BM, XF> la(p, m_block, tid);
    FwdBLoader<Cfg::BN> lb(p, n_block, tid);
    if (sg.kt_begin) { la.seek(sg.kt_begin); lb.seek(sg.kt_begin); }
    igemm_produce<Cfg>(la, lb, sg.nkt, smem, tid, ClockStamp{nullptr, 0});
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  igemm_consume<Cfg, true, true>(sg.nkt, acc, smem);
  if (sk_combine<Cfg>(sk, sg, acc))
    igemm_store_tile<Cfg>(acc, smem, n_block, p.N, p.bias, [&](int row) -> float* {
      const int m = m_block + row;
      return m < p.M ? p.out + (size_t)m * p.N + n_block : nullptr;
    }, p.stat_partial ? p.stat_partial + (size_t)mt * Cfg::WAVES_M * 2 * p.N : nullptr, p.act, p.slope, &p.epi, epi_group_off(p, m_block));
}
template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_fwd_sk_kernel(ConvP p, SkPlan sk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  SkSeg s0, s1{};
  const int nseg = sk_segments(sk, s0, s1);
  fwd_sk_segment<Cfg, XF>(p, sk, s0, smem);
  if (nseg == 1) return;
  lds_barrier();   // the epilogue staged through the LDS stages: the second segment's producers wait for it
  fwd_sk_segment<Cfg, XF>(p, sk, s1, smem);
}

// grad-input: the tiles of all sub-pixel phases are numbered phase-major (phase = tile / tiles_per_phase); the host only takes
// this form when every phase has the same number of rows and taps (one iteration space)
template <class Cfg, bool XF>
__device__ __forceinline__ void dgrad_sk_segment(const ConvP& p, const DgradPhases& phases, const SkPlan& sk, const SkSeg& sg, int tiles_per_phase,
                                                 float* smem, int* rowpix) {
  const int py = __builtin_amdgcn_readfirstlane(sg.tile / tiles_per_phase), tile = sg.tile - py * tiles_per_phase;
  const PhaseInfo& f = phases.p[py];
  const int mt = __builtin_amdgcn_readfirstlane(tile / p.tilesN), nt = tile - mt * p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
    for (int r = tid; r < Cfg::BM; r += IG_LOADERS) {
      const int m = m_block + r;
      int pix = -1;
      if (m < f.Mp) {
        uint32_t t, cc, b, aa;
        f.dPHw.divmod((uint32_t)m, t, cc);
        f.dPHh.divmod(t, b, aa);
        pix = ((int)b * p.IH + (int)aa * p.stride + f.ph) * p.IW + (int)cc * p.stride + f.pw;
      }
      rowpix[r] = pix;
    }
    DgradALoader<Cfg::BM, XF> la(p, f, m_block, tid);
    DgradBLoader<Cfg::BN> lb(p, f, n_block, tid);
    if (sg.kt_begin) { la.seek(sg.kt_begin); lb.seek(sg.kt_begin); }
    igemm_produce<Cfg>(la, lb, sg.nkt, smem, tid, ClockStamp{nullptr, 0});
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  igemm_consume<Cfg, true, false>(sg.nkt, acc, smem);
  if (sk_combine<Cfg>(sk, sg, acc))
    igemm_store_tile<Cfg>(acc, smem, n_block, p.N, p.bias, [&](int row) -> float* {
      const int pix = rowpix[row];
      return pix >= 0 ? p.out + (size_t)pix * p.Cin + n_block : nullptr;
    }, p.stat_partial ? p.stat_partial + (size_t)(f.prow0 + mt * Cfg::WAVES_M) * 2 * p.N : nullptr, p.act, p.slope, &p.epi, epi_group_off(p, m_block));
}
template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_dgrad_sk_kernel(ConvP p, DgradPhases phases, SkPlan sk, int tiles_per_phase) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int rowpix[Cfg::BM];
  SkSeg s0, s1{};
  const int nseg = sk_segments(sk, s0, s1);
  dgrad_sk_segment<Cfg, XF>(p, phases, sk, s0, tiles_per_phase, smem, rowpix);
  if (nseg == 1) return;
  lds_barrier();
  dgrad_sk_segment<Cfg, XF>(p, phases, sk, s1, tiles_per_phase, smem, rowpix);
}

template <class Cfg, bool XFA, bool XFB>   // XFA: the dy operand is a transformed activation (ConvTranspose2d layers); XFB: x is
__global__ void __launch_bounds__(IG_THREADS, 4) conv_wgrad_kernel(ConvP p, int ktiles_total, int ktiles_per_split, int tiles,
                                                                   int slice_major) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  prio_entry(p);
  // slice_major: 1-D grid, the output tiles of one K-slice are consecutive on ONE XCD, so the slice of dy / x they all read
  // comes from HBM once and from that XCD's L2 afterwards (layers with few output tiles re-read their operands per tile)
  uint32_t tile, split;
  if (slice_major) {
    const uint32_t l = xcd_remap(blockIdx.x, gridDim.x);
    split = l / (uint32_t)tiles; tile = l - split * (uint32_t)tiles;
  } else {
    tile = xcd_remap(blockIdx.x, gridDim.x); split = blockIdx.y;
  }
  const int mt = tile / p.tilesN, nt = tile % p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  const int kt_begin = split * ktiles_per_split;
  int ktiles = ktiles_total - kt_begin;
  if (ktiles > ktiles_per_split) ktiles = ktiles_per_split;

  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
    WgradALoader<Cfg::BM, XFA> la(p, m_block, kt_begin, tid);
    WgradBLoader<Cfg::BN, XFB> lb(p, n_block, kt_begin, tid);
    igemm_produce<Cfg>(la, lb, ktiles, smem, tid, ClockStamp{p.stamps, p.stamp_slots});
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  igemm_consume<Cfg, false, false>(ktiles, acc, smem, ClockStamp{p.stamps, p.stamp_slots});
  prio_epilogue(p);
  float* slab = p.out + (size_t)split * (size_t)p.M * (size_t)p.N;
  igemm_store_tile<Cfg>(acc, smem, n_block, p.N, nullptr, [&](int row) -> float* {
    const int m = m_block + row;
    return m < p.M ? slab + (size_t)m * p.N + n_block : nullptr;
  });
}


// Grad-input whose sub-pixel phases differ in rows AND taps (k3 s2: 4 : 2 : 2 : 1 taps; the longest phase's tiles alone are as
// long as the whole launch should be): ALL tiles are stream-K.  The k-tiles of every tile of every phase, phase-major, form one
// iteration space cut into `blocks` equal ranges; a range covers any number of tiles — a loop over segments, whose descriptors are
// derived from the iteration index inside the loop (nothing but the index is carried: the loop form costs ~40 spilled registers in
// the segments' prologue / epilogue and none in the k-tile loops).  Only a range's first and last segment can be partial tiles:
// two partial slots per range as before.  Replaces the GEMM + col2im detour where it was only there for balance.
struct SkNPlan {
  int nph, blocks, total;        // sub-pixel phases, workgroups, k-tile iterations of the launch
  int tile0[5], it0[5], kt[4];   // per phase: first tile (arrival-counter index), first iteration, k-tiles per tile; [nph]: totals
  float* parts; int* arrivals;
};
__device__ __forceinline__ int skn_start(int g, const SkNPlan& sk) {
  return __builtin_amdgcn_readfirstlane((int)((uint32_t)g * (uint32_t)sk.total / (uint32_t)sk.blocks));
}
__device__ __forceinline__ int skn_owner(int x, const SkNPlan& sk) {
  return __builtin_amdgcn_readfirstlane((int)((((uint32_t)x + 1u) * (uint32_t)sk.blocks - 1u) / (uint32_t)sk.total));
}
// one segment of a range: which tile, which of its k-tiles, who shares the tile (all wave-uniform, derived from the iteration index)
struct SkNSeg { int py, tin, t_begin, e, kt_begin, nkt, fb, nparts, mt, nt; };
__device__ __forceinline__ SkNSeg skn_segment(int it, int it1, const SkNPlan& sk, int tilesN) {
  SkNSeg s;
  s.py = 0;
  if (sk.nph > 1 && it >= sk.it0[1]) s.py = 1;
  if (sk.nph > 2 && it >= sk.it0[2]) s.py = 2;
  if (sk.nph > 3 && it >= sk.it0[3]) s.py = 3;
  const int KT = sk.kt[s.py], local = it - sk.it0[s.py];
  s.tin = __builtin_amdgcn_readfirstlane(local / KT);                  // tile inside the phase
  s.t_begin = sk.it0[s.py] + s.tin * KT;                               // the tile's iterations: [t_begin, t_begin + KT)
  const int t_end = s.t_begin + KT;
  s.e = it1 < t_end ? it1 : t_end;
  s.kt_begin = it - s.t_begin; s.nkt = s.e - it;
  s.fb = skn_owner(s.t_begin, sk); s.nparts = skn_owner(t_end - 1, sk) - s.fb + 1;
  s.mt = __builtin_amdgcn_readfirstlane(s.tin / tilesN); s.nt = s.tin - s.mt * tilesN;
  return s;
}
// The segment loop exists TWICE, once per role (r04): as one loop holding both roles' bodies the register allocator had to carry the
// union of the producers' loader state and the consumers' accumulators / epilogue temporaries across the back edge — 61 spilled
// VGPRs and 208 bytes of scratch.  Both loops derive the same segments from the same indices and meet at the same
// barriers.
template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_dgrad_skn_kernel(ConvP p, DgradPhases phases, SkNPlan sk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int rowpix[Cfg::BM];
  const int g = (int)xcd_remap(blockIdx.x, (uint32_t)sk.blocks);
  const int it1 = skn_start(g + 1, sk);
  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
#pragma unroll 1
    for (int it = skn_start(g, sk); it < it1;) {
      const SkNSeg sg = skn_segment(it, it1, sk, p.tilesN);
      const PhaseInfo& f = phases.p[sg.py];
      const int m_block = sg.mt * Cfg::BM, n_block = sg.nt * Cfg::BN;
      for (int r = tid; r < Cfg::BM; r += IG_LOADERS) {
        const int m = m_block + r;
        int pix = -1;
        if (m < f.Mp) {
          uint32_t t, cc, b, aa;
          f.dPHw.divmod((uint32_t)m, t, cc);
          f.dPHh.divmod(t, b, aa);
          pix = ((int)b * p.IH + (int)aa * p.stride + f.ph) * p.IW + (int)cc * p.stride + f.pw;
        }
        rowpix[r] = pix;
      }
      DgradALoader<Cfg::BM, XF> la(p, f, m_block, tid);
      DgradBLoader<Cfg::BN> lb(p, f, n_block, tid);
      if (sg.kt_begin) { la.seek(sg.kt_begin); lb.seek(sg.kt_begin); }
      igemm_produce<Cfg>(la, lb, sg.nkt, smem, tid, ClockStamp{nullptr, 0});
      it = sg.e;
      if (it < it1) lds_barrier();   // the consumers' epilogue staged through the LDS stages / read rowpix: wait for it
    }
    return;
  }
#pragma unroll 1
  for (int it = skn_start(g, sk); it < it1;) {
    const SkNSeg sg = skn_segment(it, it1, sk, p.tilesN);
    const PhaseInfo& f = phases.p[sg.py];
    const int m_block = sg.mt * Cfg::BM, n_block = sg.nt * Cfg::BN;
    {
      f32x16 acc[Cfg::TM][Cfg::TN];
      igemm_consume<Cfg, true, false>(sg.nkt, acc, smem);
      const int t_begin = sg.t_begin;
      if (sk_combine_core<Cfg>(sk.parts, sk.arrivals, sk.blocks, g, sk.tile0[sg.py] + sg.tin, sg.nparts, sg.fb,
                               [&](int blk) { return skn_start(blk, sk) >= t_begin; }, acc))
        igemm_store_tile<Cfg>(acc, smem, n_block, p.N, p.bias, [&](int row) -> float* {
          const int pix = rowpix[row];
          return pix >= 0 ? p.out + (size_t)pix * p.Cin + n_block : nullptr;
        }, p.stat_partial ? p.stat_partial + (size_t)(f.prow0 + sg.mt * Cfg::WAVES_M) * 2 * p.N : nullptr, p.act, p.slope, &p.epi, epi_group_off(p, m_block));
    }
    it = sg.e;
    if (it < it1) lds_barrier();
  }
}

// weight gradient as a stream-K launch: the K = B*OH*OW loops of ALL tiles form the iteration space (a gradient has at most a few
// hundred tiles), the last arrival of a (tile, wave) writes dw itself — with the .grad accumulation as an EPI_ADD on dw — so
// neither slabs nor the slab_reduce pass exist (288-tile gradients of the WGAN-GP critic: one workgroup per tile left 224 CUs
// with one tile and 32 with two).
template <class Cfg, bool XFA, bool XFB>
__device__ __forceinline__ void wgrad_sk_segment(const ConvP& p, const SkPlan& sk, const SkSeg& sg, float* smem) {
  const int mt = __builtin_amdgcn_readfirstlane(sg.tile / p.tilesN), nt = sg.tile - mt * p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
    WgradALoader<Cfg::BM, XFA> la(p, m_block, sg.kt_begin, tid);
    WgradBLoader<Cfg::BN, XFB> lb(p, n_block, sg.kt_begin, tid);
    igemm_produce<Cfg>(la, lb, sg.nkt, smem, tid, ClockStamp{nullptr, 0});
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  igemm_consume<Cfg, false, false>(sg.nkt, acc, smem);
  if (sk_combine<Cfg>(sk, sg, acc))
    igemm_store_tile<Cfg>(acc, smem, n_block, p.N, nullptr, [&](int row) -> float* {
      const int m = m_block + row;
      return m < p.M ? p.out + (size_t)m * p.N + n_block : nullptr;
    }, nullptr, PCG_ACT_NONE, 0.f, &p.epi);
}
template <class Cfg, bool XFA, bool XFB>
__global__ void __launch_bounds__(IG_THREADS, 4) conv_wgrad_sk_kernel(ConvP p, SkPlan sk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  SkSeg s0, s1{};
  const int nseg = sk_segments(sk, s0, s1);
  wgrad_sk_segment<Cfg, XFA, XFB>(p, sk, s0, smem);
  if (nseg == 1) return;
  lds_barrier();
  wgrad_sk_segment<Cfg, XFA, XFB>(p, sk, s1, smem);
}


#ifdef PCG_PERSISTENT_KERNELS   // experiment kept out of the shipped library (measured slower: igemm_core.h); `make lean`
// ---- persistent, tile-pipelined forms (igemm_core.h: igemm_produce_stream / igemm_consume_stream / igemm_store_regs) -------------
// Work item of the forward launch: (output tile, K-slice); item = slice * tiles + tile.
template <class Cfg, bool XF>
struct FwdSrc {
  using LA = FwdALoader<Cfg::BM, XF>;
  using LB = FwdBLoader<Cfg::BN>;
  const ConvP& p; TileWalk walk; uint32_t i, tiles; int tid;
  LA la; LB lb; int n;
  __device__ __forceinline__ void decode(uint32_t item, int& m_block, int& n_block, int& kt_begin, int& kt) const {
    const uint32_t split = item / tiles, tile = item - split * tiles;
    m_block = (int)(tile / (uint32_t)p.tilesN) * Cfg::BM; n_block = (int)(tile % (uint32_t)p.tilesN) * Cfg::BN;
    kt_begin = (int)split * p.ktiles_per_split;
    kt = p.ktiles - kt_begin;
    if (kt > p.ktiles_per_split) kt = p.ktiles_per_split;
  }
  __device__ __forceinline__ FwdSrc(const ConvP& p_, const TileWalk& w, uint32_t tiles_, int tid_, int m0, int n0, int kb0, int kt0)
      : p(p_), walk(w), i(0), tiles(tiles_), tid(tid_), la(p_, m0, tid_), lb(p_, n0, tid_), n(kt0) {
    if (kb0) { la.seek(kb0); lb.seek(kb0); }
  }
  __device__ __forceinline__ void next_tile() {
    int m_block, n_block, kb;
    decode(walk.item(++i), m_block, n_block, kb, n);
    la = LA(p, m_block, tid); lb = LB(p, n_block, tid);
    if (kb) { la.seek(kb); lb.seek(kb); }
  }
};

template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_fwd_pkernel(ConvP p, int splits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t tiles = (uint32_t)((p.M + Cfg::BM - 1) / Cfg::BM) * (uint32_t)p.tilesN;
  const TileWalk walk(tiles * (uint32_t)splits);
  const int nt = (int)walk.ntiles();
  if (nt == 0) return;                           // workgroup-uniform
  auto item_ktiles = [&](uint32_t item) {
    const int kb = (int)(item / tiles) * p.ktiles_per_split;
    const int kt = p.ktiles - kb;
    return kt > p.ktiles_per_split ? p.ktiles_per_split : kt;
  };
  int S = 0;
  for (int t = 0; t < nt; ++t) S += item_ktiles(walk.item(t));
  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
    const uint32_t it0 = walk.item(0), sp0 = it0 / tiles, tl0 = it0 - sp0 * tiles;
    FwdSrc<Cfg, XF> src(p, walk, tiles, tid, (int)(tl0 / (uint32_t)p.tilesN) * Cfg::BM, (int)(tl0 % (uint32_t)p.tilesN) * Cfg::BN,
                        (int)sp0 * p.ktiles_per_split, item_ktiles(it0));
    igemm_produce_stream<Cfg>(src, S, smem, tid);
    return;
  }
  const uint32_t slab_bytes = (uint32_t)((size_t)p.M * p.N * 4);
  igemm_consume_stream<Cfg, true, true>(nt, [&](int t) { return item_ktiles(walk.item(t)); },
    [&](int t, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
      const uint32_t item = walk.item(t), split = item / tiles, tile = item - split * tiles;
      const int mt = (int)(tile / (uint32_t)p.tilesN), m_block = mt * Cfg::BM, n_block = (int)(tile % (uint32_t)p.tilesN) * Cfg::BN;
      float* out = p.out + (size_t)split * (size_t)p.M * (size_t)p.N;
      const EpiBufs eb = make_epi_bufs(out, slab_bytes, p.epi);
      RowsAffine rows{p.M, m_block, (uint32_t)p.N * 4u, (uint32_t)n_block * 4u, 0u, 0};
      igemm_store_regs<Cfg>(acc, n_block, p.N, split == 0 ? p.bias : nullptr, rows, eb,
                            p.stat_partial ? p.stat_partial + (size_t)mt * Cfg::WAVES_M * 2 * p.N : nullptr, p.act, p.slope, &p.epi);
    }, smem, ClockStamp{p.stamps, p.stamp_slots});
}

// Work item of the grad-input launch: (sub-pixel phase, tile of the phase's pixel grid).  Phases of equal size are interleaved (the
// phases of one pixel tile are neighbours: they gather the same dy rows); otherwise the phases' tile ranges follow each other.
struct DgradItems {
  int nph, interleave, cnt[4];       // cnt[p] = tiles of phase p (tilesM_p * tilesN)
  uint32_t total;
  __device__ __forceinline__ void decode(uint32_t item, int& py, uint32_t& tile) const {
    if (interleave) { py = (int)(item % (uint32_t)nph); tile = item / (uint32_t)nph; return; }
    py = 0;
    while (py + 1 < nph && item >= (uint32_t)cnt[py]) { item -= (uint32_t)cnt[py]; ++py; }
    tile = item;
  }
};

template <class Cfg, bool XF>
struct DgradSrc {
  using LA = DgradALoader<Cfg::BM, XF>;
  using LB = DgradBLoader<Cfg::BN>;
  const ConvP& p; const DgradPhases& ph; const DgradItems& items; TileWalk walk; uint32_t i; int tid; uint32_t (*rowpix)[Cfg::BM];
  LA la; LB lb; int n;
  __device__ __forceinline__ static int ktiles_of(const ConvP& p, const PhaseInfo& f) { return f.nth * f.ntw * ((p.Cout + IG_BK - 1) / IG_BK); }
  __device__ __forceinline__ void publish_rows(const PhaseInfo& f, int m_block, uint32_t seq) {   // byte offset of every tile row's output pixel
    for (int r = tid; r < Cfg::BM; r += IG_LOADERS) {
      const int m = m_block + r;
      uint32_t off = OOB_OFF;
      if (m < f.Mp) {
        uint32_t t, cc, b, aa;
        f.dPHw.divmod((uint32_t)m, t, cc);
        f.dPHh.divmod(t, b, aa);
        const int pix = ((int)b * p.IH + (int)aa * p.stride + f.ph) * p.IW + (int)cc * p.stride + f.pw;
        off = (uint32_t)pix * (uint32_t)p.Cin * 4u;
      }
      rowpix[seq & 3u][r] = off;
    }
  }
  __device__ __forceinline__ DgradSrc(const ConvP& p_, const DgradPhases& ph_, const DgradItems& it_, const TileWalk& w, int tid_,
                                      uint32_t (*rowpix_)[Cfg::BM], const PhaseInfo& f0, int m0, int n0)
      : p(p_), ph(ph_), items(it_), walk(w), i(0), tid(tid_), rowpix(rowpix_), la(p_, f0, m0, tid_), lb(p_, f0, n0, tid_), n(ktiles_of(p_, f0)) {
    publish_rows(f0, m0, 0);
  }
  __device__ __forceinline__ void next_tile() {
    int py; uint32_t tile;
    items.decode(walk.item(++i), py, tile);
    const PhaseInfo& f = ph.p[py];
    const int m_block = (int)(tile / (uint32_t)p.tilesN) * Cfg::BM, n_block = (int)(tile % (uint32_t)p.tilesN) * Cfg::BN;
    la = LA(p, f, m_block, tid); lb = LB(p, f, n_block, tid);
    n = ktiles_of(p, f);
    publish_rows(f, m_block, i);     // slot i & 3: the consumers are at most three tiles behind (the producers lead by <= 3 k-tiles)
  }
};

template <class Cfg, bool XF>
__global__ void __launch_bounds__(IG_THREADS, Cfg::MINW) conv_dgrad_pkernel(ConvP p, DgradPhases phases, DgradItems items) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ uint32_t rowpix[4][Cfg::BM];
  const TileWalk walk(items.total);
  const int nt = (int)walk.ntiles();
  if (nt == 0) return;
  auto item_ktiles = [&](uint32_t item) {
    int py; uint32_t tile;
    items.decode(item, py, tile);
    return DgradSrc<Cfg, XF>::ktiles_of(p, phases.p[py]);
  };
  int S = 0;
  for (int t = 0; t < nt; ++t) S += item_ktiles(walk.item(t));
  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
    int py; uint32_t tile;
    items.decode(walk.item(0), py, tile);
    DgradSrc<Cfg, XF> src(p, phases, items, walk, tid, rowpix, phases.p[py], (int)(tile / (uint32_t)p.tilesN) * Cfg::BM,
                          (int)(tile % (uint32_t)p.tilesN) * Cfg::BN);
    igemm_produce_stream<Cfg>(src, S, smem, tid);
    return;
  }
  const EpiBufs eb = make_epi_bufs(p.out, p.x_bytes, p.epi);
  igemm_consume_stream<Cfg, true, false>(nt, [&](int t) { return item_ktiles(walk.item(t)); },
    [&](int t, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
      int py; uint32_t tile;
      items.decode(walk.item(t), py, tile);
      const PhaseInfo& f = phases.p[py];
      const int mt = (int)(tile / (uint32_t)p.tilesN), n_block = (int)(tile % (uint32_t)p.tilesN) * Cfg::BN;
      RowsTable rows{rowpix[(uint32_t)t & 3u], (uint32_t)n_block * 4u, nullptr, 0u};
      igemm_store_regs<Cfg>(acc, n_block, p.N, p.bias, rows, eb,
                            p.stat_partial ? p.stat_partial + (size_t)(f.prow0 + mt * Cfg::WAVES_M) * 2 * p.N : nullptr, p.act, p.slope, &p.epi);
    }, smem, ClockStamp{p.stamps, p.stamp_slots});
}

#endif  // PCG_PERSISTENT_KERNELS

// 64x192 tile of the weight gradient (Cout <= 64, N = KH*KW*Cin a multiple of 192): the same pipeline with the 192-column B loader
using Cfg64x192 = TileCfg<64, 192, 2, 2>;
__global__ void __launch_bounds__(IG_THREADS, 4) conv_wgrad192_kernel(ConvP p, int ktiles_total, int ktiles_per_split, int tiles,
                                                                      int slice_major) {
  using Cfg = Cfg64x192;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  prio_entry(p);
  uint32_t tile, split;
  if (slice_major) {
    const uint32_t l = xcd_remap(blockIdx.x, gridDim.x);
    split = l / (uint32_t)tiles; tile = l - split * (uint32_t)tiles;
  } else {
    tile = xcd_remap(blockIdx.x, gridDim.x); split = blockIdx.y;
  }
  const int mt = tile / p.tilesN, nt = tile % p.tilesN;
  const int m_block = mt * Cfg::BM, n_block = nt * Cfg::BN;
  const int kt_begin = split * ktiles_per_split;
  int ktiles = ktiles_total - kt_begin;
  if (ktiles > ktiles_per_split) ktiles = ktiles_per_split;
  if (wave_id() >= 4) {
    const int tid = threadIdx.x - IG_LOADERS;
    WgradALoader<Cfg::BM, false> la(p, m_block, kt_begin, tid);
    WgradBLoader192 lb(p, n_block, kt_begin, tid);
    igemm_produce<Cfg>(la, lb, ktiles, smem, tid);
    return;
  }
  f32x16 acc[Cfg::TM][Cfg::TN];
  igemm_consume<Cfg, false, false>(ktiles, acc, smem, ClockStamp{p.stamps, p.stamp_slots});
  prio_epilogue(p);
  float* slab = p.out + (size_t)split * (size_t)p.M * (size_t)p.N;
  igemm_store_tile<Cfg>(acc, smem, n_block, p.N, nullptr, [&](int row) -> float* {
    const int m = m_block + row;
    return m < p.M ? slab + (size_t)m * p.N + n_block : nullptr;
  });
}

// col2im for the "one GEMM + scatter" form of the grad-input (see dgrad_as_gemm): dx[b,ih,iw,ci] = bias[ci] + sum over the taps
// (kh,kw) that reach (ih,iw) of dcol[b,oh,ow][(kh,kw,ci)], taps added in (kh,kw) order.  Every dcol element is read once.
__global__ void __launch_bounds__(256) col2im_kernel(const float4* __restrict__ dcol, float4* __restrict__ dx, const float* __restrict__ bias,
                                                     int B, int IH, int IW, int CQ, int OH, int OW, int KH, int KW, int stride, int pad,
                                                     FastDiv dCQ, FastDiv dIW, FastDiv dIH, int act, float neg) {
  const uint32_t total = (uint32_t)B * IH * IW * CQ;
  for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    uint32_t pix, cq, t, iw, b, ih;
    dCQ.divmod(idx, pix, cq);
    dIW.divmod(pix, t, iw);
    dIH.divmod(t, b, ih);
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < KH; ++kh) {
      const int th = (int)ih + pad - kh;
      if (th < 0 || th % stride) continue;
      const int oh = th / stride;
      if (oh >= OH) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int tw = (int)iw + pad - kw;
        if (tw < 0 || tw % stride) continue;
        const int ow = tw / stride;
        if (ow >= OW) continue;
        const float4 v = dcol[((size_t)(((int)b * OH + oh) * OW + ow) * (KH * KW) + (kh * KW + kw)) * CQ + cq];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    if (act != PCG_ACT_NONE) {
      acc.x = act_neg_scale(acc.x, neg); acc.y = act_neg_scale(acc.y, neg); acc.z = act_neg_scale(acc.z, neg); acc.w = act_neg_scale(acc.w, neg);
    }
    dx[idx] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
using Cfg128x128 = TileCfg<128, 128, 2, 2>;
using Cfg128x128D = TileCfg<128, 128, 2, 2, true, 4, PCG_PREFETCH_DEPTH, true>;   // operands by LDS-DMA into unpadded swizzled images
#ifndef PCG_TILE64_SWZ
#define PCG_TILE64_SWZ 1     // 128x64 tiles: swizzled unpadded LDS images, three workgroups per CU (0: the r01 layout, two per CU)
#endif
using Cfg128x64P = TileCfg<128, 64, 2, 2>;                 // padded images, two workgroups per CU, two k-tiles of gathers in flight
#if PCG_TILE64_SWZ
using Cfg128x64 = TileCfg<128, 64, 2, 2, true, 6, 1>;      // measured r02: 3x3 s1 64->64 forward 97.7 -> 101.1, grad-input 106.0 -> 111.5
#else                                                      // TFLOP/s; the k4 s2 grad-input (four interleaved phases, K = 512) LOSES
using Cfg128x64 = Cfg128x64P;                              // 9 % (105.7 -> 96.6) and therefore keeps the padded config
#endif

// (r04: 192x64 tiles — four consumer waves of 96x32, two workgroups per CU, 1.5x the MFMAs per workgroup — were built and measured:
//  bit-identical results, no gain: 3x3 64->64 forward 0.4867 -> 0.4943 ms, its grad-input 0.4969 -> 0.5227, D2's k4 s2 grad-input
//  0.2875 -> 0.2877.  The epilogue is per-OUTPUT work, not per-workgroup work; only the ~4 us entry is amortised.  Removed)

int check_geom(const pcg_conv_geom* g) {
  PCG_REQUIRE(g != nullptr, "conv geometry is null");
  PCG_REQUIRE(g->B > 0 && g->IH > 0 && g->IW > 0 && g->Cin > 0 && g->OH > 0 && g->OW > 0 && g->Cout > 0,
              "conv geometry: non-positive extent");
  PCG_REQUIRE(g->KH > 0 && g->KW > 0 && g->stride > 0 && g->pad >= 0, "conv geometry: bad kernel/stride/pad");
  PCG_REQUIRE(g->OH == (g->IH + 2 * g->pad - g->KH) / g->stride + 1 && g->OW == (g->IW + 2 * g->pad - g->KW) / g->stride + 1,
              "conv geometry: OH/OW inconsistent with IH/IW, kernel %dx%d stride %d pad %d", g->KH, g->KW, g->stride, g->pad);
  PCG_REQUIRE((int64_t)g->B * g->IH * g->IW * g->Cin * 4 < (1ll << 31) && (int64_t)g->B * g->OH * g->OW * g->Cout * 4 < (1ll << 31) &&
                  (int64_t)g->Cout * g->KH * g->KW * g->Cin * 4 < (1ll << 31),
              "conv geometry: an operand tensor reaches 2 GiB (32-bit buffer offsets); split the batch");
  PCG_REQUIRE(g->KH * g->KW <= 32, "conv geometry: more than 32 taps (%dx%d) unsupported", g->KH, g->KW);
  return PCG_OK;
}

// Tuning switches for A/B measurements in ONE process (pcg_tune_set; scripts/conv_microbench.py --ab): -1 = the built-in choice.
struct Tune { int edge_prio = -1, dgrad_swz3 = -1, wgrad_rounds = -1, korder = -1, wgrad_order = -1, dgrad_interleave = -1, persistent = -1, persist_tiles = -1, fwd_splits = -1, dma = -1, stream_k = -1, sk_blocks = -1, dgrad_gemm = -1, t64 = -1; unsigned long long* stamps = nullptr; int stamp_slots = 0; };
Tune g_tune;

ConvP make_params(const pcg_conv_geom* g) {
  ConvP p{};
  static const int korder_env = getenv("PCG_KORDER") ? atoi(getenv("PCG_KORDER")) : 1;
  p.korder = g_tune.korder >= 0 ? g_tune.korder : korder_env;
  // default 2: the epilogue at priority 3, the entry left alone (measured r04, CounteRGAN step, one process, interleaved:
  // 0: 27.15 ms, 2: 26.49, 3: 27.62 — raising the entry as well lets every new workgroup's set-up cut into the running loops)
  static const int edge_env = getenv("PCG_EDGE_PRIO") ? atoi(getenv("PCG_EDGE_PRIO")) : 2;
  p.edge_prio = g_tune.edge_prio >= 0 ? g_tune.edge_prio : edge_env;
  p.stamps = g_tune.stamps; p.stamp_slots = g_tune.stamp_slots;
  p.B = g->B; p.IH = g->IH; p.IW = g->IW; p.Cin = g->Cin; p.OH = g->OH; p.OW = g->OW; p.Cout = g->Cout;
  p.KH = g->KH; p.KW = g->KW; p.stride = g->stride; p.pad = g->pad;
  p.dOW = FastDiv((uint32_t)g->OW);
  p.dOH = FastDiv((uint32_t)g->OH);
  p.x_bytes = (uint32_t)((int64_t)g->B * g->IH * g->IW * g->Cin * 4);
  p.dy_bytes = (uint32_t)((int64_t)g->B * g->OH * g->OW * g->Cout * 4);
  p.w_bytes = (uint32_t)((int64_t)g->Cout * g->KH * g->KW * g->Cin * 4);
  return p;
}

// validate an input transform for an operand with C channels and put it into the kernel parameters
int set_xform(const char* who, const pcg_in_xform* xf, int C, ConvP* p) {
  if (!xf || !xf->scale) return PCG_OK;
  PCG_REQUIRE(xf->shift != nullptr, "%s: input transform without shift", who);
  PCG_REQUIRE(xf->act == PCG_ACT_NONE || xf->act == PCG_ACT_RELU || xf->act == PCG_ACT_LRELU,
              "%s: input transform activation %d is not none / ReLU / LeakyReLU", who, xf->act);
  PCG_REQUIRE(C % 4 == 0 && (((uintptr_t)xf->scale | (uintptr_t)xf->shift) & 15) == 0,
              "%s: input transform needs a channel count %% 4 == 0 and 16-byte aligned scale / shift", who);
  p->in_sc = xf->scale; p->in_sh = xf->shift; p->in_neg = act_neg_of(xf->act, xf->slope); p->in_c_bytes = (uint32_t)C * 4u;
  return PCG_OK;
}

template <class Cfg, bool AK, bool BK_>
constexpr size_t smem_bytes() {
  constexpr int a = igemm_smem_floats<Cfg, AK, BK_>(), b = epilogue_smem_floats<Cfg>();
  return sizeof(float) * (size_t)(a > b ? a : b);
}

template <class K>
int set_smem(K kernel, size_t bytes) {
  // > 64 KB of dynamic LDS needs the opt-in attribute; harmless otherwise
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", bytes, hipGetErrorString(e)); return PCG_ERR_LAUNCH; }
  return PCG_OK;
}

// persistent launches: workgroups resident per CU (what the LDS and register budgets of the tile configuration admit) x CUs
int cu_count() {
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return cus;
}
template <class Cfg>
unsigned persistent_grid(uint64_t items) {
  const unsigned slots = (unsigned)cu_count() * (Cfg::MINW >= 6 ? 3u : 2u);
  unsigned g = items < slots ? (unsigned)items : slots;
  // persist_tiles = T > 0: T tiles per workgroup instead of all-resident workgroups — more workgroups than slots, dispatched in
  // rounds, so they desynchronise like the one-tile kernels while a workgroup's second .. T-th prologue is pipelined away
  if (g_tune.persist_tiles > 0) g = (unsigned)((items + g_tune.persist_tiles - 1) / g_tune.persist_tiles);
  if (g >= 8) g &= ~7u;                          // TileWalk: a multiple of 8 keeps the per-XCD runs
  return g ? g : 1u;
}
bool use_persistent() {
#ifdef PCG_PERSISTENT_KERNELS
  static const int env = getenv("PCG_PERSISTENT") ? atoi(getenv("PCG_PERSISTENT")) : 0;
  return (g_tune.persistent >= 0 ? g_tune.persistent : env) != 0;
#else
  return false;
#endif
}
template <class Cfg, bool AK, bool BK_>
constexpr size_t stage_smem_bytes() { return sizeof(float) * (size_t)igemm_smem_floats<Cfg, AK, BK_>(); }

// ---- stream-K scratch: caller-owned, registered per stream (pcg_conv_set_scratch) --------------------------------------
constexpr int SK_MAX_BLOCKS = 512, SK_MAX_TILES = 4096;
constexpr size_t SK_PARTS_BYTES = (size_t)2 * SK_MAX_BLOCKS * 128 * 128 * sizeof(float);
constexpr size_t SK_ARRIVALS_BYTES = (size_t)SK_MAX_TILES * 4 * sizeof(int);
// keyed by (device, stream): the default stream has handle 0 on EVERY device, so the handle alone would let a second GPU's
// registration overwrite the first one's
struct SkScratch { int device; hipStream_t stream; float* parts; int* arrivals; };
SkScratch g_sk_scratch[64];
int g_sk_scratch_n = 0;
std::mutex g_sk_mutex;
int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : 0; }
bool sk_scratch_of(hipStream_t s, SkPlan* sk) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  for (int i = 0; i < g_sk_scratch_n; ++i)
    if (g_sk_scratch[i].stream == s && g_sk_scratch[i].device == dev) {
      if (sk) { sk->parts = g_sk_scratch[i].parts; sk->arrivals = g_sk_scratch[i].arrivals; }
      return true;
    }
  return false;
}
int sk_mode() {      // 0 off, 1 where the model sees > 10 % to gain (default), 2 wherever the form is valid (tests, scans)
  static const int env = getenv("PCG_STREAM_K") ? atoi(getenv("PCG_STREAM_K")) : 1;
  return g_tune.stream_k >= 0 ? g_tune.stream_k : env;
}
// Decide between whole rounds + stream-K remainder and the plain launch.  Times in us from the r03 stamps of the 128x128 kernels:
// a k-tile takes a workgroup 3.5 us when two share a CU (the matrix pipe's rate) and 2.2 us when it has the CU alone; a whole
// tile costs ~6 us outside its main loop, a stream-K segment ~12 (partial tile out, counter, the next segment's cold start); the
// last arrival reads ~1.5 us per partial tile.  Checked against scripts/probes/streamk_scan.py: the model takes the launches
// that gained 7-24 % there and leaves the K = 512 grad-input GEMMs (16 k-tiles per tile: no gain) data-parallel.
bool plan_sk_shape(int tiles, int ktiles, SkPlan* sk) {       // the decision alone (no scratch, no stream): host logic, CPU-testable
  const int mode = sk_mode();
  if (mode == 3) {                       // diagnostic: every tile data-parallel, but through the stream-K kernels (their cost as such)
    sk->dp_tiles = tiles; sk->sk_tiles = 0; sk->sk_blocks = 0; sk->ktiles = ktiles;
    return true;
  }
  if (mode == 0 || ktiles < 8) return false;
  const int dp = tiles / 512 * 512, r = tiles - dp;
  if (r == 0 || r > SK_MAX_TILES || (int64_t)r * ktiles * 512 >= (1ll << 31)) return false;
  const double t_dp = (r <= 256 ? 2.2 : 3.5) * ktiles + 6.0;
  double best = 1e30;
  int bg = 0;
  for (int G = 512; G >= 64; G >>= 1) {
    if (g_tune.sk_blocks > 0 && G != g_tune.sk_blocks) continue;
    if (G < r) break;
    const int it = ceil_div(r * ktiles, G);
    if (it < 4) continue;
    const int parts = ceil_div(ktiles, it) + 1;
    if (parts > 8) continue;
    const double t = it * (G == 512 ? 3.5 : 2.2) + 2 * 12.0 + 3.0 + 1.5 * parts;
    if (t < best) { best = t; bg = G; }
  }
  if (!bg || (mode == 1 && best > 0.9 * t_dp)) return false;
  sk->dp_tiles = dp; sk->sk_tiles = r; sk->sk_blocks = bg; sk->ktiles = ktiles;
  return true;
}
bool plan_sk(int tiles, int ktiles, hipStream_t s, SkPlan* sk) { return plan_sk_shape(tiles, ktiles, sk) && sk_scratch_of(s, sk); }
// workspace queries carry no stream: they assume scratch when ANY stream of this device has it (an upper bound on what a launch
// may need; a launch on a stream without scratch plans with stream_has_sk and falls back to the plain forms)
bool have_sk_scratch() {
  if (sk_mode() == 0) return false;
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  for (int i = 0; i < g_sk_scratch_n; ++i) if (g_sk_scratch[i].device == dev) return true;
  return false;
}
int stream_has_sk(hipStream_t s) { return sk_mode() != 0 && sk_scratch_of(s, nullptr) ? 1 : 0; }

template <class Cfg, bool XF>
int launch_fwd_x(ConvP p, int splits, hipStream_t s) {
  p.tilesN = ceil_div(p.N, Cfg::BN);
  const int tilesM = ceil_div(p.M, Cfg::BM);
#ifdef PCG_PERSISTENT_KERNELS
  if (use_persistent()) {
    constexpr size_t smem = stage_smem_bytes<Cfg, true, true>();
    static int once = set_smem(conv_fwd_pkernel<Cfg, XF>, smem);
    if (once != PCG_OK) return once;
    const uint64_t items = (uint64_t)tilesM * p.tilesN * splits;
    hipLaunchKernelGGL((conv_fwd_pkernel<Cfg, XF>), dim3(persistent_grid<Cfg>(items)), dim3(IG_THREADS), smem, s, p, splits);
    return launch_status("conv_fwd_pkernel");
  }
#endif
  constexpr size_t smem = smem_bytes<Cfg, true, true>();
  if constexpr (Cfg::BM == 128 && Cfg::BN == 128 && !Cfg::DMA) {
    SkPlan sk{};
    if (splits == 1 && plan_sk(tilesM * p.tilesN, p.ktiles, s, &sk)) {
      static int once_sk = set_smem(conv_fwd_sk_kernel<Cfg, XF>, smem);
      if (once_sk != PCG_OK) return once_sk;
      hipLaunchKernelGGL((conv_fwd_sk_kernel<Cfg, XF>), dim3((unsigned)(sk.dp_tiles + sk.sk_blocks)), dim3(IG_THREADS), smem, s, p, sk);
      return launch_status("conv_fwd_sk_kernel");
    }
  }
  static int once = set_smem(conv_fwd_kernel<Cfg, XF>, smem);
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL((conv_fwd_kernel<Cfg, XF>), dim3((unsigned)tilesM * p.tilesN, splits), dim3(IG_THREADS), smem, s, p);
  return launch_status("conv_fwd_kernel");
}
template <class Cfg>
int launch_fwd(const ConvP& p, int splits, hipStream_t s) {
  return p.in_sc ? launch_fwd_x<Cfg, true>(p, splits, s) : launch_fwd_x<Cfg, false>(p, splits, s);
}

// Forward split-K: when M*N gives far fewer tiles than the chip has CUs and K is long (small-batch layers with big weights:
// the WGAN-GP critic's conv3 / 8192->1024 Linear, the generator's 1x1 -> 4x4 ConvT backward), K is cut into slabs that are
// summed in slab order by slab_reduce — the same deterministic scheme as the weight gradient.
struct FwdPlan { int splits, ktiles_per_split; };
FwdPlan plan_fwd(const pcg_conv_geom* g, int have_sk = -1) {
  if (have_sk < 0) have_sk = have_sk_scratch() ? 1 : 0;
  const int M = g->B * g->OH * g->OW, N = g->Cout;
  const int tiles = ceil_div(M, 128) * ceil_div(N, N > 64 ? 128 : 64);
  const int ktiles = g->KH * g->KW * ceil_div(g->Cin, IG_BK);
  FwdPlan f{1, ktiles};
  if (g_tune.fwd_splits > 0) {               // A/B override (pcg_tune_set("fwd_splits", s)): s K-slices whatever the shape
    f.ktiles_per_split = ceil_div(ktiles, g_tune.fwd_splits);
    f.splits = ceil_div(ktiles, f.ktiles_per_split);
    return f;
  }
  if (tiles > 256 && tiles < 448 && ktiles >= 32 && !have_sk) {     // (stream-K takes these when it has scratch)
    // a little over one block per CU (288 tiles: 32 CUs get two full-K blocks, the others one — the critic's conv2 at batch 256
    // ran at 73 TFLOP/s): a few K-slices bring the blocks per CU to ceil(tiles*s/256)/s.  Each slice also costs a slab of M*N
    // floats written and read: measured r03 (scripts/probes/fwd_splits_scan.py, 288 tiles x 72 k-tiles) 1: 283, 2: 241, 3: 221,
    // 4: 256 us — the slab term is worth ~0.1 of a block per slice.
    int best = 1;
    double bc = 2.0;
    for (int s = 2; s <= 4; ++s) {
      const double c = (double)ceil_div(tiles * s, 256) / s + 0.1 * s;
      if (c < bc - 1e-9) { bc = c; best = s; }
    }
    f.ktiles_per_split = ceil_div(ktiles, best);
    f.splits = ceil_div(ktiles, f.ktiles_per_split);
    return f;
  }
  if (tiles > 96 || ktiles < 32) return f;
  int splits = ceil_div(tiles >= 48 ? 512 : 384, tiles);     // (64 tiles x 144 k-tiles, r03 scan: 6 slices 115 us, 8 slices 101)
  if (splits > ktiles / 8) splits = ktiles / 8;
  if (splits < 2) return f;
  f.ktiles_per_split = ceil_div(ktiles, splits);
  f.splits = ceil_div(ktiles, f.ktiles_per_split);
  return f;
}

// every tile stream-K over phases of different length (conv_dgrad_skn_kernel): the decision alone
bool plan_skn_shape(const DgradPhases& ph, int nphases, int BM, int tilesN, int Cout, SkNPlan* sn) {
  if (sk_mode() == 0 || nphases < 2) return false;
  sn->nph = nphases;
  bool ok = true;
  int tiles = 0, total = 0, ktmax = 0;
  for (int i = 0; i < nphases && ok; ++i) {
    const int kt = ph.p[i].nth * ph.p[i].ntw * ceil_div(Cout, IG_BK), t = ceil_div(ph.p[i].Mp, BM) * tilesN;
    ok = ph.p[i].nth > 0 && ph.p[i].ntw > 0 && ph.p[i].Mp > 0 && kt > 0;
    sn->tile0[i] = tiles; sn->it0[i] = total; sn->kt[i] = kt;
    tiles += t; total += t * kt;
    if (kt > ktmax) ktmax = kt;
  }
  sn->tile0[nphases] = tiles; sn->it0[nphases] = total;
  int blocks = total / 512 >= 16 ? 512 : 256;
  if (g_tune.sk_blocks > 0) blocks = g_tune.sk_blocks;
  // (measured: pays where the tiles are long — 72-79 k-tiles on average: 309 -> 201 and 363 -> 237 us — and loses where they
  //  are short — 38 on average, 13x13 -> 6x6 at 512 channels: 236 -> 261 us: ~2.7 segments per range, each with its fixed cost)
  ok = ok && tiles <= SK_MAX_TILES && blocks <= SK_MAX_BLOCKS && total / blocks >= 8 && (int64_t)total * blocks < (1ll << 31) &&
       ktmax <= 8 * (total / blocks) && (sk_mode() == 2 || g_tune.sk_blocks > 0 || total >= 48 * tiles);
  sn->blocks = blocks; sn->total = total;
  return ok;
}

template <class Cfg, bool XF>
int launch_dgrad_x(ConvP p, const DgradPhases& ph, int nphases, int maxMp, hipStream_t s) {
  p.tilesN = ceil_div(p.N, Cfg::BN);
  const int tilesM = ceil_div(maxMp, Cfg::BM);
#ifdef PCG_PERSISTENT_KERNELS
  bool all_have_taps = true;
  for (int i = 0; i < nphases; ++i) all_have_taps = all_have_taps && ph.p[i].nth > 0;
  if (use_persistent() && all_have_taps) {
    constexpr size_t smem = stage_smem_bytes<Cfg, true, false>();
    static int once = set_smem(conv_dgrad_pkernel<Cfg, XF>, smem);
    if (once != PCG_OK) return once;
    static const int il_env0 = getenv("PCG_DGRAD_INTERLEAVE") ? atoi(getenv("PCG_DGRAD_INTERLEAVE")) : 1;
    const int il_env = g_tune.dgrad_interleave >= 0 ? g_tune.dgrad_interleave : il_env0;
    DgradItems it{};
    it.nph = nphases;
    bool same = nphases > 1 && il_env && p.w_bytes <= (1u << 20);
    for (int i = 1; i < nphases; ++i) same = same && ph.p[i].Mp == ph.p[0].Mp;
    it.interleave = same ? 1 : 0;
    uint64_t total = 0;
    for (int i = 0; i < nphases; ++i) { it.cnt[i] = ceil_div(ph.p[i].Mp, Cfg::BM) * p.tilesN; total += (uint64_t)it.cnt[i]; }
    it.total = (uint32_t)total;
    hipLaunchKernelGGL((conv_dgrad_pkernel<Cfg, XF>), dim3(persistent_grid<Cfg>(total)), dim3(IG_THREADS), smem, s, p, ph, it);
    return launch_status("conv_dgrad_pkernel");
  }
#endif
  constexpr size_t smem = smem_bytes<Cfg, true, false>();
  if constexpr (Cfg::BM == 128 && Cfg::BN == 128) {
    bool uniform = true;
    for (int i = 0; i < nphases; ++i)
      uniform = uniform && ph.p[i].Mp == ph.p[0].Mp && ph.p[i].nth * ph.p[i].ntw == ph.p[0].nth * ph.p[0].ntw && ph.p[i].nth > 0 && ph.p[i].ntw > 0;
    if (!uniform && nphases > 1 && sk_mode() != 0) {
      // phases of different length: every tile stream-K (conv_dgrad_skn_kernel).  512 ranges, or 256 for small launches (>= 16
      // k-tiles per range, at most 8 ranges per tile)
      SkNPlan sn{};
      SkPlan scratch{};
      if (plan_skn_shape(ph, nphases, Cfg::BM, p.tilesN, p.Cout, &sn) && sk_scratch_of(s, &scratch)) {
        sn.parts = scratch.parts; sn.arrivals = scratch.arrivals;
        static int once_sn = set_smem(conv_dgrad_skn_kernel<Cfg, XF>, smem);
        if (once_sn != PCG_OK) return once_sn;
        DgradPhases phs = ph;
        phs.interleave = 0;
        hipLaunchKernelGGL((conv_dgrad_skn_kernel<Cfg, XF>), dim3((unsigned)sn.blocks), dim3(IG_THREADS), smem, s, p, phs, sn);
        return launch_status("conv_dgrad_skn_kernel");
      }
    }
    SkPlan sk{};
    const int tpp = tilesM * p.tilesN;
    if (uniform && plan_sk(tpp * nphases, ph.p[0].nth * ph.p[0].ntw * ceil_div(p.Cout, IG_BK), s, &sk)) {
      static int once_sk = set_smem(conv_dgrad_sk_kernel<Cfg, XF>, smem);
      if (once_sk != PCG_OK) return once_sk;
      DgradPhases phs = ph;
      phs.interleave = 0;
      hipLaunchKernelGGL((conv_dgrad_sk_kernel<Cfg, XF>), dim3((unsigned)(sk.dp_tiles + sk.sk_blocks)), dim3(IG_THREADS), smem, s, p, phs, sk, tpp);
      return launch_status("conv_dgrad_sk_kernel");
    }
  }
  static int once = set_smem(conv_dgrad_kernel<Cfg, XF>, smem);
  if (once != PCG_OK) return once;
  static const int il_env0 = getenv("PCG_DGRAD_INTERLEAVE") ? atoi(getenv("PCG_DGRAD_INTERLEAVE")) : 1;   // A/B switch
  const int il_env = g_tune.dgrad_interleave >= 0 ? g_tune.dgrad_interleave : il_env0;
  DgradPhases phl = ph;
  // only for small weight tensors: interleaved phases keep ALL phases' weight slices live in an XCD's 4 MB L2 at once (measured:
  // WGAN-GP's 8 MB ConvT weights ran 20 % slower interleaved, DCGAN's 0.5 MB D2 / G4 layers 1-3 % faster)
  bool same = nphases > 1 && il_env && p.w_bytes <= (1u << 20);
  for (int i = 1; i < nphases; ++i) same = same && ph.p[i].Mp == ph.p[0].Mp;
  phl.interleave = same ? nphases : 0;
  if (same)
    hipLaunchKernelGGL((conv_dgrad_kernel<Cfg, XF>), dim3((unsigned)tilesM * p.tilesN * nphases), dim3(IG_THREADS), smem, s, p, phl);
  else
    hipLaunchKernelGGL((conv_dgrad_kernel<Cfg, XF>), dim3((unsigned)tilesM * p.tilesN, nphases), dim3(IG_THREADS), smem, s, p, phl);
  return launch_status("conv_dgrad_kernel");
}
template <class Cfg>
int launch_dgrad(const ConvP& p, const DgradPhases& ph, int nphases, int maxMp, hipStream_t s) {
  return p.in_sc ? launch_dgrad_x<Cfg, true>(p, ph, nphases, maxMp, s) : launch_dgrad_x<Cfg, false>(p, ph, nphases, maxMp, s);
}

struct WgradPlan { int splits, ktiles_total, ktiles_per_split, tiles; bool narrow, wide192; };

WgradPlan plan_wgrad(const pcg_conv_geom* g) {
  WgradPlan w{};
  const int M = g->Cout, N = g->KH * g->KW * g->Cin;
  w.narrow = (M <= 64);
  static const int t192_env = getenv("PCG_WGRAD_192") ? atoi(getenv("PCG_WGRAD_192")) : 1;   // A/B switch
  // 64x192 tile: N = 576 (3x3, 64 channels) is 4.5 tiles of 128 columns — the fifth one is half empty, 11 % of the MFMAs
  w.wide192 = w.narrow && t192_env && N % 192 == 0 && (N % 128) != 0;
  const int BM = w.narrow ? 64 : 128;
  w.tiles = ceil_div(M, BM) * (w.wide192 ? N / 192 : ceil_div(N, 128));
  const int64_t K = (int64_t)g->B * g->OH * g->OW;
  w.ktiles_total = (int)ceil_div64(K, IG_BK);
  // fill the 512 block slots (2 per CU) in ONE round — 515 blocks would run as 512 + a second round of 3 — but keep
  // >= 8 k-tiles (256 pixels) per slice
  const int slots = 512 * (g_tune.wgrad_rounds > 0 ? g_tune.wgrad_rounds : 1);      // (A/B: more, shorter K-slices = more than one round of workgroups)
  int splits = w.tiles <= slots ? slots / w.tiles : 1;
  const int max_splits = w.ktiles_total / 8 > 0 ? w.ktiles_total / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  // When the tiles do not divide the chip (288 tiles of a 1024 x 4608 gradient: 32 CUs get two full-K blocks, the rest one) a few
  // more K-slices balance the CUs at the price of slab traffic.  Cost model per candidate s, in us: n = blocks per CU (ceil),
  // n * k-tiles per slice * time per k-tile (one block alone on a CU exposes its latencies: x1.2) + s slabs written and read
  // at ~4 TB/s.  The 512-slot rule above stands unless the model sees > 5 % to gain (it agrees on every DCGAN / counteRGAN layer).
  auto cost = [&](int s, int* s_eff) {
    const int kps = ceil_div(w.ktiles_total, s), se = ceil_div(w.ktiles_total, kps);
    const int n = ceil_div(w.tiles * se, 256);
    *s_eff = se;
    return (double)n * kps * (w.wide192 ? 1.5 : w.narrow ? 1.0 : 1.95) * (n == 1 ? 1.2 : 1.0) + (double)se * M * N * 8.0 / 4.0e6;
  };
  int se = 0;
  const double c0 = cost(splits, &se);
  double cb = c0;
  int sb = splits;
  for (int s = 1; s <= max_splits && s <= 64; ++s) {
    const double c = cost(s, &se);
    if (se == s && c < cb * 0.98) { cb = c; sb = s; }
  }
  if (cb < 0.95 * c0) splits = sb;
  w.ktiles_per_split = ceil_div(w.ktiles_total, splits);
  w.splits = ceil_div(w.ktiles_total, w.ktiles_per_split);
  return w;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" size_t pcg_conv2d_fwd_workspace_bytes(const pcg_conv_geom* g) {
  if (check_geom(g) != PCG_OK) return 0;
  if (thin_is_cin(g) || thin_is_cout(g)) return thin_conv_fwd_workspace_bytes(g);
  const FwdPlan f = plan_fwd(g);
  return f.splits > 1 ? (size_t)f.splits * g->B * g->OH * g->OW * g->Cout * sizeof(float) : 0;
}
// Grad-input as ONE balanced GEMM + col2im.  With a kernel size that is not a multiple of the stride (k3 s2: WGAN-GP critic,
// mnist_wgan_conditional.py:84-90) the sub-pixel phases see 4 : 2 : 2 : 1 taps; the phase kernel's longest blocks are then the
// critical path (measured 30-60 TFLOP/s on the critic's conv2 / conv3).  dcol[B*OH*OW][KH*KW*Cin] = dy * W is the grad-input of
// the 1x1 convolution with Cin' = KH*KW*Cin on the SAME weight bytes (OHWI rows are [Cout][KH*KW*Cin]) — same MACs as the phase
// form, every block the same K = Cout — followed by a col2im pass over dcol (HBM-bound, 2 x |dcol| extra traffic).
static bool dgrad_as_gemm(const pcg_conv_geom* g, int have_sk = -1) {
  if (have_sk < 0) have_sk = have_sk_scratch() ? 1 : 0;
  if (thin_is_cin(g) || thin_is_cout(g)) return false;
  if (g_tune.dgrad_gemm == 0) return false;       // A/B: the phase kernel instead
  if (g_tune.dgrad_gemm < 0 && have_sk && g->stride == 2) {
    // The phase form multiplies every pixel of a phase by every tap of the phase, the GEMM form every dy pixel by every tap:
    // which one wastes fewer MACs on taps that fall outside depends on the geometry (7 -> 4 with padding 1: 121 against 144 per
    // image and channel pair; 13 -> 6 without padding: 400 against 324; 6 -> 2: 81 against 36).  With its load balance repaired
    // by conv_dgrad_skn_kernel the phase form is taken where it has clearly fewer (measured r03, B = 256 ConvT 1024 -> 512:
    // 328 -> 237 us).
    int64_t phase_macs = 0;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const int PHh = a < g->IH ? (g->IH - a + 1) / 2 : 0, PHw = b < g->IW ? (g->IW - b + 1) / 2 : 0;
        const int kh0 = (a + g->pad) % 2, kw0 = (b + g->pad) % 2;
        const int nth = kh0 < g->KH ? (g->KH - kh0 + 1) / 2 : 0, ntw = kw0 < g->KW ? (g->KW - kw0 + 1) / 2 : 0;
        phase_macs += (int64_t)PHh * PHw * nth * ntw;
      }
    if (phase_macs * 20 <= (int64_t)g->OH * g->OW * g->KH * g->KW * 19) return false;
  }
  if (g->stride < 2 || (g->KH % g->stride == 0 && g->KW % g->stride == 0)) return false;
  if (g->Cout < 512 || g->Cin % 4 || g->Cout % 4) return false;          // K = Cout: >= 16 k-tiles per block
  const int64_t bytes = (int64_t)g->B * g->OH * g->OW * g->KH * g->KW * g->Cin * 4;
  return bytes < (1ll << 31);
}
static size_t dgrad_gemm_bytes(const pcg_conv_geom* g) { return (size_t)g->B * g->OH * g->OW * g->KH * g->KW * g->Cin * sizeof(float); }

// Scratch of the hybrid stream-K launches (partial accumulator tiles + arrival counters), owned by the caller and registered for
// ONE stream: launches on that stream use it in stream order.  `arrivals` must be zero-filled when it is registered; the kernels
// leave it zero.  A stream without scratch runs the plain launches.  parts == nullptr forgets the stream.
extern "C" size_t pcg_conv_scratch_parts_bytes(void) { return pcg::SK_PARTS_BYTES; }
extern "C" size_t pcg_conv_scratch_arrivals_bytes(void) { return pcg::SK_ARRIVALS_BYTES; }
extern "C" int pcg_conv_set_scratch(pcg_stream_t stream, void* parts, size_t parts_bytes, void* arrivals, size_t arrivals_bytes) {
  using namespace pcg;
  hipStream_t s = (hipStream_t)stream;
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  int at = -1;
  for (int i = 0; i < g_sk_scratch_n; ++i) if (g_sk_scratch[i].stream == s && g_sk_scratch[i].device == dev) at = i;
  if (!parts) {
    if (at >= 0) g_sk_scratch[at] = g_sk_scratch[--g_sk_scratch_n];
    return PCG_OK;
  }
  PCG_REQUIRE(arrivals != nullptr && parts_bytes >= SK_PARTS_BYTES && arrivals_bytes >= SK_ARRIVALS_BYTES,
              "pcg_conv_set_scratch: parts %zu B (need %zu), arrivals %zu B (need %zu)", parts_bytes, SK_PARTS_BYTES, arrivals_bytes, SK_ARRIVALS_BYTES);
  PCG_REQUIRE((((uintptr_t)parts) & 15) == 0 && (((uintptr_t)arrivals) & 3) == 0, "pcg_conv_set_scratch: misaligned scratch");
  if (at < 0) {
    PCG_REQUIRE(g_sk_scratch_n < 64, "pcg_conv_set_scratch: more than 64 streams with scratch");
    at = g_sk_scratch_n++;
  }
  g_sk_scratch[at] = SkScratch{dev, s, (float*)parts, (int*)arrivals};
  return PCG_OK;
}
// The stream-K kernels rely on the arrival counters being zero between launches (the last arrival of a tile resets its counters).
// After a launch on `stream` failed or was aborted they may not be: this zero-fills them in stream order.  No-op for a stream
// without scratch.
extern "C" int pcg_conv_reset_scratch(pcg_stream_t stream) {
  using namespace pcg;
  SkPlan sk{};
  if (!sk_scratch_of((hipStream_t)stream, &sk)) return PCG_OK;
  hipError_t e = hipMemsetAsync(sk.arrivals, 0, SK_ARRIVALS_BYTES, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("pcg_conv_reset_scratch: %s", hipGetErrorString(e)); return PCG_ERR_LAUNCH; }
  return PCG_OK;
}

extern "C" size_t pcg_conv2d_dgrad_workspace_bytes(const pcg_conv_geom* g) {
  if (check_geom(g) != PCG_OK) return 0;
  if (thin_is_cin(g) || thin_is_cout(g)) return thin_conv_dgrad_workspace_bytes(g);
  return dgrad_as_gemm(g) ? dgrad_gemm_bytes(g) : 0;
}

namespace pcg {
int launch_bn_stats_finalize(const double* partial, int nparts, int64_t rows, int C, float eps, float momentum, float* save_mean,
                             float* save_invstd, float* running_mean, float* running_var, int64_t* nbt, hipStream_t s,
                             bool has_presum_tail, const float* gamma = nullptr, const float* beta = nullptr, float* coef = nullptr);
size_t bn_partial_buffer_bytes(int nparts, int C);
int launch_bn_stats_finalize_g(const double* partial, int nparts, int nphases, int groups, int64_t rows_per_group, int C, float eps,
                               float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                               int64_t* nbt, hipStream_t s);
}

static bool mfma_layer(const pcg_conv_geom* g) { return !(thin_is_cin(g) || thin_is_cout(g)); }
static int fwd_stat_rows(const pcg_conv_geom* g) { return ceil_div(g->B * g->OH * g->OW, 128) * 2; }
static int dgrad_stat_rows(const pcg_conv_geom* g) {
  int rows = 0;
  const int s = g->stride;
  for (int a = 0; a < s; ++a)
    for (int b = 0; b < s; ++b) {
      const int PHh = a < g->IH ? (g->IH - a + s - 1) / s : 0, PHw = b < g->IW ? (g->IW - b + s - 1) / s : 0;
      rows += ceil_div(g->B * PHh * PHw, 128) * 2;
    }
  return rows;
}

static bool fwd_use_t64(int M, int N, int splits) {
  const int tiles128 = ceil_div(M, 128) * ceil_div(N, 128);
  return g_tune.t64 != 0 && N > 64 && splits == 1 && tiles128 > 224 && tiles128 <= 256 && M % 128 == 0;
}
static int conv2d_fwd_impl(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                           double* stat_partial, void* workspace, size_t workspace_bytes, pcg_stream_t stream, int act = PCG_ACT_NONE,
                           float slope = 0.f, const EpiAux* epi = nullptr, const pcg_in_xform* xf = nullptr) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(x && w && y, "pcg_conv2d_fwd: null pointer");
  PCG_REQUIRE(act >= PCG_ACT_NONE && act <= PCG_ACT_SIGMOID, "pcg_conv2d_fwd: unknown activation %d", act);
  if (thin_is_cin(g) || thin_is_cout(g)) {
    PCG_REQUIRE(!(xf && xf->scale), "pcg_conv2d_fwd: input transforms are only implemented on the MFMA path (Cin > 3 and Cout > 3)");
    return thin_conv_fwd(g, x, w, bias, y, workspace, workspace_bytes, (hipStream_t)stream, act, slope);
  }
  PCG_REQUIRE(g->Cin % 4 == 0, "pcg_conv2d_fwd: Cin=%d must be a multiple of 4 for the MFMA path (1..3-channel layers take the thin path)", g->Cin);
  ConvP p = make_params(g);
  p.x = x; p.w = w; p.bias = bias; p.out = y; p.stat_partial = stat_partial;
  if (epi) p.epi = *epi;
  if (int e = set_xform("pcg_conv2d_fwd", xf, g->Cin, &p)) return e;
  p.M = g->B * g->OH * g->OW; p.N = g->Cout;
  p.ktiles = g->KH * g->KW * ceil_div(g->Cin, IG_BK);
  p.ktiles_per_split = p.ktiles;
  hipStream_t s = (hipStream_t)stream;
  // split-K needs the slab workspace and cannot fuse statistics or the activation (both need the complete sum)
  FwdPlan f = plan_fwd(g, stream_has_sk(s));        // this stream's scratch decides, not "some stream has scratch"
  const size_t need = (size_t)f.splits * p.M * p.N * sizeof(float);
  if (f.splits > 1 && (stat_partial || epi || !workspace || workspace_bytes < need || ((uintptr_t)workspace & 15) || ((uintptr_t)y & 15) || (p.M * (size_t)p.N) % 4))
    f = FwdPlan{1, p.ktiles};
  const bool fuse = act_is_cheap(act) && f.splits == 1;   // the epilogue fuses ReLU / LeakyReLU; tanh / sigmoid run as a second pass
  p.act = fuse ? act : PCG_ACT_NONE; p.slope = act_neg_of(p.act, slope);
  if (f.splits > 1) { p.out = (float*)workspace; p.ktiles_per_split = f.ktiles_per_split; }
  static const int dma_env = getenv("PCG_DMA") ? atoi(getenv("PCG_DMA")) : 0;
  const bool dma = (g_tune.dma >= 0 ? g_tune.dma : dma_env) != 0 && !p.in_sc;
  // At most one 128x128 tile per CU (DCGAN D4: 8192 x 512 = 256 tiles): a workgroup alone on a CU runs at ~80 % of the matrix rate
  // (one consumer wave per SIMD exposes its own latencies).  64x128 tiles double the workgroups — two per CU again — at the price
  // of 64x32 wave tiles: measured r03, D4 forward 262 -> 249 us (131 -> 138 TFLOP/s), G2's 260 -> 249; stream-K on the 256 big
  // tiles had measured neutral.  (pcg_tune_set("t64", 0) keeps the 128x128 tiles.)
  // Only where the doubled count fills the 512 slots (225..256 big tiles); fewer tiles take stream-K / K-slices as before.
  const bool t64 = fwd_use_t64(p.M, p.N, f.splits);
  if (int e = t64 ? launch_fwd<TileCfg<64, 128, 1, 4>>(p, f.splits, s)
                  : p.N > 64 ? (dma ? launch_fwd<Cfg128x128D>(p, f.splits, s) : launch_fwd<Cfg128x128>(p, f.splits, s))
                       : launch_fwd<Cfg128x64>(p, f.splits, s)) return e;
  if (f.splits > 1) {
    const size_t n = (size_t)p.M * p.N;
    if (int e = launch_slab_reduce((const float*)workspace, y, n, n, f.splits, 0, s)) return e;
  }
  return (fuse || act == PCG_ACT_NONE) ? PCG_OK : pcg_act_fwd(y, (int64_t)p.M * p.N, act, slope, y, stream);
}

extern "C" int pcg_conv2d_fwd(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                              void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return conv2d_fwd_impl(g, x, w, bias, y, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int pcg_conv2d_fwd_act(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, int act, float slope,
                                  float* y, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return conv2d_fwd_impl(g, x, w, bias, y, nullptr, workspace, workspace_bytes, stream, act, slope);
}

extern "C" size_t pcg_conv2d_fwd_bn_workspace_bytes(const pcg_conv_geom* g) {
  if (check_geom(g) != PCG_OK || !mfma_layer(g) || g->Cout % 4) return 0;
  return bn_partial_buffer_bytes(fwd_stat_rows(g), g->Cout);
}
extern "C" size_t pcg_conv2d_dgrad_bn_workspace_bytes(const pcg_conv_geom* g) {
  if (check_geom(g) != PCG_OK || !mfma_layer(g) || g->stride > 2 || g->Cin % 4) return 0;
  return bn_partial_buffer_bytes(dgrad_stat_rows(g), g->Cin);
}

extern "C" int pcg_conv2d_fwd_bn_xf(const pcg_conv_geom* g, const float* x, const pcg_in_xform* xf, const float* w, const float* bias,
                                    float* y, float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean,
                                    float* running_var, int64_t* num_batches_tracked, const float* gamma, const float* beta,
                                    float* coef_out, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  const size_t need = pcg_conv2d_fwd_bn_workspace_bytes(g);
  PCG_REQUIRE(need > 0, "pcg_conv2d_fwd_bn: only MFMA layers (Cin > 3, Cout > 3, Cout %% 4 == 0); use pcg_conv2d_fwd + pcg_bn_train_stats");
  PCG_REQUIRE(save_mean && save_invstd, "pcg_conv2d_fwd_bn: null statistics output");
  PCG_REQUIRE(!coef_out || (gamma && beta), "pcg_conv2d_fwd_bn_xf: coef_out needs gamma and beta");
  if (!workspace || workspace_bytes < need) { set_error("pcg_conv2d_fwd_bn: workspace %zu B < required %zu B", workspace_bytes, need); return PCG_ERR_WORKSPACE; }
  if (int e = conv2d_fwd_impl(g, x, w, bias, y, (double*)workspace, nullptr, 0, stream, PCG_ACT_NONE, 0.f, nullptr, xf)) return e;
  return launch_bn_stats_finalize((const double*)workspace, fwd_stat_rows(g), (int64_t)g->B * g->OH * g->OW, g->Cout, eps, momentum,
                                  save_mean, save_invstd, running_mean, running_var, num_batches_tracked, (hipStream_t)stream, true,
                                  gamma, beta, coef_out);
}
extern "C" int pcg_conv2d_fwd_bn(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, float eps,
                                 float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return pcg_conv2d_fwd_bn_xf(g, x, nullptr, w, bias, y, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                              num_batches_tracked, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}
// Grouped form (r04): `groups` independent batches of g->B / groups images side by side in ONE launch — the discriminator's real and
// fake passes (mnist_dcgan.py:151-161) as one 2B-row convolution per layer.  The convolution itself does not care; BatchNorm does:
// each group gets its own batch statistics (save_mean / save_invstd are [groups][Cout]) from its own partial rows, the running
// statistics move once per group in group order and num_batches_tracked advances by `groups`, exactly what `groups` successive
// training-mode forwards do.  Needs (B / groups) * OH * OW % 128 == 0 (whole tiles per group).
extern "C" int pcg_conv2d_fwd_bn_g(const pcg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, float eps,
                                   float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, int32_t groups, void* workspace, size_t workspace_bytes,
                                   pcg_stream_t stream) {
  const size_t need = pcg_conv2d_fwd_bn_workspace_bytes(g);
  PCG_REQUIRE(need > 0, "pcg_conv2d_fwd_bn_g: only MFMA layers (Cin > 3, Cout > 3, Cout %% 4 == 0)");
  PCG_REQUIRE(save_mean && save_invstd, "pcg_conv2d_fwd_bn_g: null statistics output");
  PCG_REQUIRE(groups >= 1 && groups <= 8 && g->B % groups == 0 && ((int64_t)(g->B / groups) * g->OH * g->OW) % 128 == 0,
              "pcg_conv2d_fwd_bn_g: %d images do not split into %d groups of whole 128-row tiles (%d x %d output)", g->B, groups, g->OH, g->OW);
  if (!workspace || workspace_bytes < need) { set_error("pcg_conv2d_fwd_bn_g: workspace %zu B < required %zu B", workspace_bytes, need); return PCG_ERR_WORKSPACE; }
  if (int e = conv2d_fwd_impl(g, x, w, bias, y, (double*)workspace, nullptr, 0, stream)) return e;
  return launch_bn_stats_finalize_g((const double*)workspace, fwd_stat_rows(g), 1, groups, (int64_t)(g->B / groups) * g->OH * g->OW, g->Cout,
                                    eps, momentum, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, (hipStream_t)stream);
}
extern "C" int pcg_conv2d_fwd_xf(const pcg_conv_geom* g, const float* x, const pcg_in_xform* xf, const float* w, const float* bias,
                                 int act, float slope, float* y, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return conv2d_fwd_impl(g, x, w, bias, y, nullptr, workspace, workspace_bytes, stream, act, slope, nullptr, xf);
}

// the sub-pixel phases (ih % s, iw % s) of a grad-input: rows, first taps, taps per axis, partial-statistics rows
static int build_phases(const pcg_conv_geom* g, DgradPhases* php, int* maxMp_out) {
  DgradPhases& ph = *php;
  int nph = 0, maxMp = 0, prow = 0;
  const int s = g->stride;
  for (int a = 0; a < s; ++a)
    for (int b = 0; b < s; ++b) {
      PhaseInfo& f = ph.p[nph];
      f.ph = a; f.pw = b;
      f.PHh = a < g->IH ? (g->IH - a + s - 1) / s : 0;
      f.PHw = b < g->IW ? (g->IW - b + s - 1) / s : 0;
      f.Mp = g->B * f.PHh * f.PHw;
      if (f.Mp == 0) continue;
      f.kh0 = (a + g->pad) % s; f.kw0 = (b + g->pad) % s;
      f.nth = f.kh0 < g->KH ? (g->KH - f.kh0 + s - 1) / s : 0;
      f.ntw = f.kw0 < g->KW ? (g->KW - f.kw0 + s - 1) / s : 0;
      if (f.nth == 0 || f.ntw == 0) { f.nth = 0; f.ntw = 1; }
      f.dh0 = (a + g->pad - f.kh0) / s; f.dw0 = (b + g->pad - f.kw0) / s;
      f.dPHw = FastDiv((uint32_t)f.PHw); f.dPHh = FastDiv((uint32_t)f.PHh);
      if (f.Mp > maxMp) maxMp = f.Mp;
      f.prow0 = prow; prow += ceil_div(f.Mp, 128) * 2;
      ++nph;
    }
  *maxMp_out = maxMp;
  return nph;
}

static int conv2d_dgrad_impl(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, float* dx,
                             double* stat_partial, void* workspace, size_t workspace_bytes, pcg_stream_t stream, int act = PCG_ACT_NONE,
                             float slope = 0.f, const EpiAux* epi = nullptr, const pcg_in_xform* xf = nullptr) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(dy && w && dx, "pcg_conv2d_dgrad: null pointer");
  PCG_REQUIRE(act >= PCG_ACT_NONE && act <= PCG_ACT_SIGMOID, "pcg_conv2d_dgrad: unknown activation %d", act);
  const bool has_xf = xf && xf->scale;
  if (thin_is_cin(g) || thin_is_cout(g)) {
    PCG_REQUIRE(!has_xf || thin_conv_xf_ok(g), "pcg_conv2d_dgrad: input transforms on the MFMA path (Cin > 3 and Cout > 3) and on the thin forms of pcg_conv2d_xf_thin_ok only");
    PCG_REQUIRE(!has_xf || (xf->act == PCG_ACT_NONE || xf->act == PCG_ACT_RELU || xf->act == PCG_ACT_LRELU), "pcg_conv2d_dgrad: transform activation %d", has_xf ? xf->act : 0);
    const ThinXf tx{has_xf ? xf->scale : nullptr, has_xf ? xf->shift : nullptr, has_xf ? act_neg_of(xf->act, xf->slope) : 1.f};
    return thin_conv_dgrad(g, dy, w, bias_x, dx, workspace, workspace_bytes, (hipStream_t)stream, act, slope, &tx);
  }
  PCG_REQUIRE(g->Cin % 4 == 0 && g->Cout % 4 == 0, "pcg_conv2d_dgrad: Cin=%d and Cout=%d must be multiples of 4", g->Cin, g->Cout);
  PCG_REQUIRE(g->stride <= 2, "pcg_conv2d_dgrad: stride %d > 2 unsupported", g->stride);
  if (!epi && !stat_partial && !has_xf && dgrad_as_gemm(g, stream_has_sk((hipStream_t)stream)) && workspace && workspace_bytes >= dgrad_gemm_bytes(g) &&
      (((uintptr_t)workspace | (uintptr_t)dx) & 15) == 0) {
    pcg_conv_geom g1 = {g->B, g->OH, g->OW, g->KH * g->KW * g->Cin, g->OH, g->OW, g->Cout, 1, 1, 1, 0};
    float* dcol = (float*)workspace;
    if (int e = conv2d_dgrad_impl(&g1, dy, w, nullptr, dcol, nullptr, nullptr, 0, stream)) return e;
    const bool fuse_act = act_is_cheap(act);
    const int CQ = g->Cin / 4;
    const uint64_t total = (uint64_t)g->B * g->IH * g->IW * CQ;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(col2im_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(dcol),
                       reinterpret_cast<float4*>(dx), bias_x, g->B, g->IH, g->IW, CQ, g->OH, g->OW, g->KH, g->KW, g->stride, g->pad,
                       FastDiv((uint32_t)CQ), FastDiv((uint32_t)g->IW), FastDiv((uint32_t)g->IH), fuse_act ? act : PCG_ACT_NONE,
                       act_neg_of(fuse_act ? act : PCG_ACT_NONE, slope));
    if (int e = launch_status("col2im_kernel")) return e;
    return fuse_act ? PCG_OK : pcg_act_fwd(dx, (int64_t)g->B * g->IH * g->IW * g->Cin, act, slope, dx, stream);
  }
  ConvP p = make_params(g);
  const bool fuse = act_is_cheap(act);
  p.dy = dy; p.w = w; p.bias = bias_x; p.out = dx; p.stat_partial = stat_partial;
  if (epi) p.epi = *epi;
  if (int e = set_xform("pcg_conv2d_dgrad", xf, g->Cout, &p)) return e;
  p.act = fuse ? act : PCG_ACT_NONE; p.slope = act_neg_of(p.act, slope);
  p.N = g->Cin;
  DgradPhases ph{};
  int maxMp = 0;
  const int nph = build_phases(g, &ph, &maxMp);
  PCG_REQUIRE(nph > 0, "pcg_conv2d_dgrad: empty problem");
  hipStream_t st = (hipStream_t)stream;
  if (int e = p.N > 64 ? launch_dgrad<Cfg128x128>(p, ph, nph, maxMp, st)
                       : (nph > 1 && g_tune.dgrad_swz3 == 0) ? launch_dgrad<Cfg128x64P>(p, ph, nph, maxMp, st)   // (r04: with the epilogue at priority 3 the three-per-CU config also wins on the four-phase k4 s2 grad-input: D2 0.2957 -> 0.2893 ms)
                                                             : launch_dgrad<Cfg128x64>(p, ph, nph, maxMp, st)) return e;
  return fuse ? PCG_OK : pcg_act_fwd(dx, (int64_t)g->B * g->IH * g->IW * g->Cin, act, slope, dx, stream);
}

extern "C" int pcg_conv2d_dgrad(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, float* dx,
                                void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return conv2d_dgrad_impl(g, dy, w, bias_x, dx, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int pcg_conv2d_dgrad_act(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, int act, float slope,
                                    float* dx, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return conv2d_dgrad_impl(g, dy, w, bias_x, dx, nullptr, workspace, workspace_bytes, stream, act, slope);
}

extern "C" int pcg_conv2d_dgrad_bn_xf(const pcg_conv_geom* g, const float* dy, const pcg_in_xform* xf, const float* w, const float* bias_x,
                                      float* dx, float eps, float momentum, float* save_mean, float* save_invstd, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, const float* gamma, const float* beta,
                                      float* coef_out, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  const size_t need = pcg_conv2d_dgrad_bn_workspace_bytes(g);
  PCG_REQUIRE(need > 0, "pcg_conv2d_dgrad_bn: only MFMA layers with stride <= 2; use pcg_conv2d_dgrad + pcg_bn_train_stats");
  PCG_REQUIRE(save_mean && save_invstd, "pcg_conv2d_dgrad_bn: null statistics output");
  PCG_REQUIRE(!coef_out || (gamma && beta), "pcg_conv2d_dgrad_bn_xf: coef_out needs gamma and beta");
  if (!workspace || workspace_bytes < need) { set_error("pcg_conv2d_dgrad_bn: workspace %zu B < required %zu B", workspace_bytes, need); return PCG_ERR_WORKSPACE; }
  if (int e = conv2d_dgrad_impl(g, dy, w, bias_x, dx, (double*)workspace, nullptr, 0, stream, PCG_ACT_NONE, 0.f, nullptr, xf)) return e;
  return launch_bn_stats_finalize((const double*)workspace, dgrad_stat_rows(g), (int64_t)g->B * g->IH * g->IW, g->Cin, eps, momentum,
                                  save_mean, save_invstd, running_mean, running_var, num_batches_tracked, (hipStream_t)stream, true,
                                  gamma, beta, coef_out);
}
extern "C" int pcg_conv2d_dgrad_bn(const pcg_conv_geom* g, const float* dy, const float* w, const float* bias_x, float* dx, float eps,
                                   float momentum, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return pcg_conv2d_dgrad_bn_xf(g, dy, nullptr, w, bias_x, dx, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                                num_batches_tracked, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int pcg_conv2d_dgrad_xf(const pcg_conv_geom* g, const float* dy, const pcg_in_xform* xf, const float* w, const float* bias_x,
                                   int act, float slope, float* dx, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return conv2d_dgrad_impl(g, dy, w, bias_x, dx, nullptr, workspace, workspace_bytes, stream, act, slope, nullptr, xf);
}

// ---- backward-pass epilogues: the gradient w.r.t. the layer below's OUTPUT leaves the kernel already multiplied by that
// layer's ReLU / LeakyReLU derivative, optionally with BatchNorm-backward's two column sums taken on the way -------------------
static int epi_common(const char* who, const pcg_conv_geom* g, const float* out, const float* aux, int act, float slope, EpiAux* e) {
  PCG_REQUIRE(mfma_layer(g), "%s: only MFMA layers (Cin > 3 and Cout > 3); use the unfused calls", who);
  PCG_REQUIRE(aux && out, "%s: null pointer", who);
  PCG_REQUIRE(act == PCG_ACT_NONE || act == PCG_ACT_RELU || act == PCG_ACT_LRELU, "%s: activation %d is not none / ReLU / LeakyReLU", who, act);
  PCG_REQUIRE((((uintptr_t)aux | (uintptr_t)out) & 15) == 0, "%s: tensors must be 16-byte aligned", who);
  e->neg = act_neg_of(act, slope);
  e->delta_bytes = (int64_t)((intptr_t)aux - (intptr_t)out);
  return PCG_OK;
}

// a Cin = 1 layer whose dy-side operand may carry an input transform (pcg_conv2d_dgrad_xf, pcg_conv2d_wgrad_xf with xf_dy): DCGAN's last
// ConvTranspose2d(64, 1, 4, 2, 1) behind BatchNorm + ReLU (mnist_dcgan.py:85-88) — no BatchNorm-apply pass, no activated copy
extern "C" int32_t pcg_conv2d_xf_thin_ok(const pcg_conv_geom* g) { return check_geom(g) == PCG_OK && thin_conv_xf_ok(g) ? 1 : 0; }
extern "C" int32_t pcg_conv2d_dgrad_mask_thin_ok(const pcg_conv_geom* g) { return check_geom(g) == PCG_OK && thin_conv_dgrad_mask_ok(g) && g->Cin % 4 == 0 ? 1 : 0; }
extern "C" int pcg_conv2d_dgrad_mask(const pcg_conv_geom* g, const float* dy, const float* w, const float* a_below, int act, float slope,
                                     float* dx, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  EpiAux e{};
  e.mode = EPI_MASK;
  if (int rc = check_geom(g)) return rc;
  if (pcg_conv2d_dgrad_mask_thin_ok(g)) {      // a one-channel layer's grad-input (r04): the mask in the row-block expand kernel
    PCG_REQUIRE(dy && w && a_below && dx && (((uintptr_t)a_below | (uintptr_t)dx) & 15) == 0, "pcg_conv2d_dgrad_mask: null or misaligned tensor");
    PCG_REQUIRE(act == PCG_ACT_NONE || act == PCG_ACT_RELU || act == PCG_ACT_LRELU, "pcg_conv2d_dgrad_mask: activation %d is not none / ReLU / LeakyReLU", act);
    return thin_conv_dgrad_mask(g, dy, w, a_below, act, slope, dx, (hipStream_t)stream);
  }
  if (int rc = epi_common("pcg_conv2d_dgrad_mask", g, dx, a_below, act, slope, &e)) return rc;
  return conv2d_dgrad_impl(g, dy, w, nullptr, dx, nullptr, workspace, workspace_bytes, stream, PCG_ACT_NONE, 0.f, &e);
}
extern "C" int pcg_conv2d_fwd_mask(const pcg_conv_geom* g, const float* x, const float* w, const float* a_below, int act, float slope,
                                   float* y, void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  EpiAux e{};
  e.mode = EPI_MASK;
  if (int rc = check_geom(g)) return rc;
  if (int rc = epi_common("pcg_conv2d_fwd_mask", g, y, a_below, act, slope, &e)) return rc;
  return conv2d_fwd_impl(g, x, w, nullptr, y, nullptr, workspace, workspace_bytes, stream, PCG_ACT_NONE, 0.f, &e);
}

extern "C" int pcg_conv2d_dgrad_add(const pcg_conv_geom* g, const float* dy, const float* w, const float* addend, float* dx,
                                    void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  EpiAux e{};
  e.mode = EPI_ADD;
  if (int rc = check_geom(g)) return rc;
  if (int rc = epi_common("pcg_conv2d_dgrad_add", g, dx, addend, PCG_ACT_NONE, 0.f, &e)) return rc;
  return conv2d_dgrad_impl(g, dy, w, nullptr, dx, nullptr, workspace, workspace_bytes, stream, PCG_ACT_NONE, 0.f, &e);
}

extern "C" int pcg_conv2d_fwd_add(const pcg_conv_geom* g, const float* x, const float* w, const float* addend, float* y,
                                  void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  EpiAux e{};
  e.mode = EPI_ADD;
  if (int rc = check_geom(g)) return rc;
  if (int rc = epi_common("pcg_conv2d_fwd_add", g, y, addend, PCG_ACT_NONE, 0.f, &e)) return rc;
  return conv2d_fwd_impl(g, x, w, nullptr, y, nullptr, workspace, workspace_bytes, stream, PCG_ACT_NONE, 0.f, &e);
}
// skip-add followed by the ReLU / LeakyReLU backward of the layer whose activated output a_below the sum is a gradient of:
// y = (conv(x, w) + addend) * (a_below > 0 ? 1 : slope) in ONE epilogue (was pcg_conv2d_*_add + a pcg_act_bwd pass over the tensor)
static int add_mask_epi(const char* who, const pcg_conv_geom* g, const float* out, const float* addend, const float* a_below, int act,
                        float slope, EpiAux* e) {
  e->mode = EPI_ADD;
  if (int rc = check_geom(g)) return rc;
  if (int rc = epi_common(who, g, out, addend, act, slope, e)) return rc;      // e->neg = the activation's negative-side factor
  PCG_REQUIRE(a_below && a_below != out && (((uintptr_t)a_below) & 15) == 0, "%s: a_below must be a 16-byte aligned tensor other than the output", who);
  e->delta2_bytes = (int64_t)((intptr_t)a_below - (intptr_t)out);
  return PCG_OK;
}
extern "C" int pcg_conv2d_fwd_add_mask(const pcg_conv_geom* g, const float* x, const float* w, const float* addend, const float* a_below,
                                       int act, float slope, float* y, pcg_stream_t stream) {
  EpiAux e{};
  if (int rc = add_mask_epi("pcg_conv2d_fwd_add_mask", g, y, addend, a_below, act, slope, &e)) return rc;
  return conv2d_fwd_impl(g, x, w, nullptr, y, nullptr, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}
extern "C" int pcg_conv2d_dgrad_add_mask(const pcg_conv_geom* g, const float* dy, const float* w, const float* addend, const float* a_below,
                                         int act, float slope, float* dx, pcg_stream_t stream) {
  EpiAux e{};
  if (int rc = add_mask_epi("pcg_conv2d_dgrad_add_mask", g, dx, addend, a_below, act, slope, &e)) return rc;
  return conv2d_dgrad_impl(g, dy, w, nullptr, dx, nullptr, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}
extern "C" int pcg_conv2d_fwd_add_bnsum(const pcg_conv_geom* g, const float* x, const float* w, const float* addend, const float* z_next,
                                        const float* mean, const float* invstd, float sum_scale, float* y, void* partial,
                                        size_t partial_bytes, pcg_stream_t stream) {
  EpiAux e{};
  e.mode = EPI_ADDSUM;
  if (int rc = check_geom(g)) return rc;
  e.no_addend = addend == nullptr;          // (r04) nullable: the sums of the plain result
  if (int rc = epi_common("pcg_conv2d_fwd_add_bnsum", g, y, addend ? addend : y, PCG_ACT_NONE, 0.f, &e)) return rc;
  const size_t need = pcg_conv2d_fwd_bn_workspace_bytes(g);
  PCG_REQUIRE(need > 0, "pcg_conv2d_fwd_add_bnsum: layer not eligible (MFMA layers, channel count %% 4 == 0)");
  PCG_REQUIRE(z_next && mean && invstd && (((uintptr_t)z_next | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0,
              "pcg_conv2d_fwd_add_bnsum: z_next / mean / invstd must be non-null and 16-byte aligned");
  if (!partial || partial_bytes < need) { set_error("pcg_conv2d_fwd_add_bnsum: partial-sum buffer %zu B < required %zu B", partial_bytes, need); return PCG_ERR_WORKSPACE; }
  e.neg = sum_scale;
  e.delta2_bytes = (int64_t)((intptr_t)z_next - (intptr_t)y);
  e.mean = mean; e.invstd = invstd;
  return conv2d_fwd_impl(g, x, w, nullptr, y, (double*)partial, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}

namespace pcg { namespace {
// w[co][kh][kw][ci] -> w_adj[ci][KH-1-kh][KW-1-kw][co]: the OHWI weight of the adjoint (grad-input) convolution of a stride-1 layer
__global__ void __launch_bounds__(256) weight_adjoint_kernel(const float* __restrict__ w, float* __restrict__ wa, int Cout, int KHW, int Cin) {
  const int total = Cout * KHW * Cin;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    const int co = o % Cout, t = o / Cout;           // o indexes w_adj: [ci][tap'][co]
    const int tapr = t % KHW, ci = t / KHW;
    wa[o] = w[((size_t)co * KHW + (KHW - 1 - tapr)) * Cin + ci];
  }
}
} }
namespace pcg { namespace {
struct AdjMany { const float* w[16]; float* wa[16]; };
__global__ void __launch_bounds__(256) weight_adjoint_many_kernel(AdjMany a, int Cout, int KHW, int Cin) {
  const float* __restrict__ w = a.w[blockIdx.y];
  float* __restrict__ wa = a.wa[blockIdx.y];
  const int total = Cout * KHW * Cin;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    const int co = o % Cout, t = o / Cout;
    const int tapr = t % KHW, ci = t / KHW;
    wa[o] = w[((size_t)co * KHW + (KHW - 1 - tapr)) * Cin + ci];
  }
}
} }
// pcg_conv_weight_adjoint for up to 16 layers of ONE shape in a single launch (the 12 3x3 64->64 convolutions of the CounteRGAN
// generator's residual blocks, models/generator.py:11-14, whose grad-inputs all run on the forward kernel: 12 launches of 6 us -> 1)
extern "C" int pcg_conv_weight_adjoint_many(const float* const* w, float* const* w_adj, int32_t n, int32_t Cout, int32_t KH, int32_t KW,
                                            int32_t Cin, pcg_stream_t stream) {
  PCG_REQUIRE(w && w_adj && n >= 1 && n <= 16 && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "pcg_conv_weight_adjoint_many: bad arguments (1..16 layers)");
  AdjMany a{};
  for (int i = 0; i < n; ++i) {
    PCG_REQUIRE(w[i] && w_adj[i], "pcg_conv_weight_adjoint_many: null pointer");
    a.w[i] = w[i]; a.wa[i] = w_adj[i];
  }
  const int total = Cout * KH * KW * Cin;
  hipLaunchKernelGGL(weight_adjoint_many_kernel, dim3((unsigned)((total + 255) / 256 > 256 ? 256 : (total + 255) / 256), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, a, Cout, KH * KW, Cin);
  return launch_status("weight_adjoint_many_kernel");
}

// Stride-1 layers: the grad-input convolution is itself a forward convolution of dy with the 180-degree-rotated, channel-transposed
// weight and padding K-1-pad.  Running it on the FORWARD kernel makes both operands K-major (one ds_read_b128 per fragment
// instead of four ds_read_b32 for the transposed weight slices of the grad-input kernel): measured r02 on the 3x3 64->64 layers.
extern "C" int pcg_conv_weight_adjoint(const float* w, float* w_adj, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, pcg_stream_t stream) {
  PCG_REQUIRE(w && w_adj && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "pcg_conv_weight_adjoint: bad arguments");
  const int total = Cout * KH * KW * Cin;
  hipLaunchKernelGGL(weight_adjoint_kernel, dim3((unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, w_adj, Cout, KH * KW, Cin);
  return launch_status("weight_adjoint_kernel");
}

extern "C" int pcg_conv2d_dgrad_add_bnsum(const pcg_conv_geom* g, const float* dy, const float* w, const float* addend, const float* z_next,
                                          const float* mean, const float* invstd, float sum_scale, float* dx, void* partial,
                                          size_t partial_bytes, pcg_stream_t stream) {
  EpiAux e{};
  e.mode = EPI_ADDSUM;
  if (int rc = check_geom(g)) return rc;
  e.no_addend = addend == nullptr;          // (r04) nullable: the sums of the plain result
  if (int rc = epi_common("pcg_conv2d_dgrad_add_bnsum", g, dx, addend ? addend : dx, PCG_ACT_NONE, 0.f, &e)) return rc;
  const size_t need = pcg_conv2d_dgrad_bn_workspace_bytes(g);
  PCG_REQUIRE(need > 0, "pcg_conv2d_dgrad_add_bnsum: layer not eligible (MFMA layers, stride <= 2, channel count %% 4 == 0)");
  PCG_REQUIRE(z_next && mean && invstd && (((uintptr_t)z_next | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0,
              "pcg_conv2d_dgrad_add_bnsum: z_next / mean / invstd must be non-null and 16-byte aligned");
  if (!partial || partial_bytes < need) { set_error("pcg_conv2d_dgrad_add_bnsum: partial-sum buffer %zu B < required %zu B", partial_bytes, need); return PCG_ERR_WORKSPACE; }
  e.neg = sum_scale;
  e.delta2_bytes = (int64_t)((intptr_t)z_next - (intptr_t)dx);
  e.mean = mean; e.invstd = invstd;
  return conv2d_dgrad_impl(g, dy, w, nullptr, dx, (double*)partial, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}

static int bnbwd_epi(const char* who, const pcg_conv_geom* g, const float* out, const float* z_below, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int act, float slope, size_t need, void* partial, size_t partial_bytes,
                     EpiAux* e) {
  PCG_REQUIRE(need > 0, "%s: layer not eligible (MFMA layers, stride <= 2, channel count %% 4 == 0)", who);
  PCG_REQUIRE(mean && invstd && gamma && beta, "%s: null BatchNorm parameter", who);
  PCG_REQUIRE((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "%s: BatchNorm vectors must be 16-byte aligned", who);
  if (!partial || partial_bytes < need) { set_error("%s: partial-sum buffer %zu B < required %zu B", who, partial_bytes, need); return PCG_ERR_WORKSPACE; }
  e->mode = EPI_BNBWD;
  e->mean = mean; e->invstd = invstd; e->gamma = gamma; e->beta = beta;
  return epi_common(who, g, out, z_below, act, slope, e);
}

extern "C" int pcg_conv2d_dgrad_bnbwd(const pcg_conv_geom* g, const float* dy, const float* w, const float* z_below, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dx,
                                      void* partial, size_t partial_bytes, pcg_stream_t stream) {
  EpiAux e{};
  if (int rc = check_geom(g)) return rc;
  if (int rc = bnbwd_epi("pcg_conv2d_dgrad_bnbwd", g, dx, z_below, mean, invstd, gamma, beta, act, slope,
                         pcg_conv2d_dgrad_bn_workspace_bytes(g), partial, partial_bytes, &e)) return rc;
  return conv2d_dgrad_impl(g, dy, w, nullptr, dx, (double*)partial, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}
extern "C" int pcg_conv2d_fwd_bnbwd(const pcg_conv_geom* g, const float* x, const float* w, const float* z_below, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, int act, float slope, float* y,
                                    void* partial, size_t partial_bytes, pcg_stream_t stream) {
  EpiAux e{};
  if (int rc = check_geom(g)) return rc;
  if (int rc = bnbwd_epi("pcg_conv2d_fwd_bnbwd", g, y, z_below, mean, invstd, gamma, beta, act, slope,
                         pcg_conv2d_fwd_bn_workspace_bytes(g), partial, partial_bytes, &e)) return rc;
  return conv2d_fwd_impl(g, x, w, nullptr, y, (double*)partial, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}
// Grouped form of pcg_conv2d_dgrad_bnbwd: dy holds `groups` batches side by side; mean / invstd of the layer below are [groups][Cin].
// The partial rows keep the layout of the ungrouped launch ([phase][tile row]); pcg_bn_bwd_partial_g reads each group's share of
// every phase (pcg_conv2d_dgrad_bn_phases of them).  Needs equal sub-pixel phases with whole 128-row tiles per group.
extern "C" int32_t pcg_conv2d_dgrad_bn_phases(const pcg_conv_geom* g) {
  if (check_geom(g) != PCG_OK) return 0;
  DgradPhases ph{};
  int maxMp = 0;
  return build_phases(g, &ph, &maxMp);
}
extern "C" int pcg_conv2d_dgrad_bnbwd_g(const pcg_conv_geom* g, const float* dy, const float* w, const float* z_below, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dx,
                                        void* partial, size_t partial_bytes, int32_t groups, pcg_stream_t stream) {
  EpiAux e{};
  if (int rc = check_geom(g)) return rc;
  if (int rc = bnbwd_epi("pcg_conv2d_dgrad_bnbwd_g", g, dx, z_below, mean, invstd, gamma, beta, act, slope,
                         pcg_conv2d_dgrad_bn_workspace_bytes(g), partial, partial_bytes, &e)) return rc;
  DgradPhases ph{};
  int maxMp = 0;
  const int nph = build_phases(g, &ph, &maxMp);
  bool ok = groups >= 1 && groups <= 8 && g->B % groups == 0 && nph > 0;
  for (int i = 0; i < nph && ok; ++i) ok = ph.p[i].Mp == maxMp;
  ok = ok && (maxMp / groups) % 128 == 0 && maxMp % groups == 0;
  PCG_REQUIRE(ok, "pcg_conv2d_dgrad_bnbwd_g: needs equal sub-pixel phases whose rows split into %d groups of whole 128-row tiles", groups);
  e.group_rows = groups > 1 ? maxMp / groups : 0;
  return conv2d_dgrad_impl(g, dy, w, nullptr, dx, (double*)partial, nullptr, 0, stream, PCG_ACT_NONE, 0.f, &e);
}
// Thin forward (Cin = 1, k4) whose output is a gradient w.r.t. the activated output of  z -> BatchNorm(train) -> ReLU / LeakyReLU:
// returns dz directly and adds dgamma / dbeta — the output tensor itself, the reduction pass over it and the re-read in the apply pass
// disappear (thin_rows_expand_bn_kernel; DCGAN: G5's grad-input + G4's BatchNorm backward, mnist_dcgan.py:85-88 backward).
namespace pcg { bool dp_sync_bn(); }     // dp_rccl.hip: exact global-batch BatchNorm is on (its sums are all-reduced inside the BatchNorm entry points)
extern "C" int32_t pcg_conv2d_fwd_bnbwd_thin_ok(const pcg_conv_geom* g) {
  return check_geom(g) == PCG_OK && !pcg::dp_sync_bn() && thin_conv_fwd_bnbwd_ok(g) && g->Cout % 4 == 0 ? 1 : 0;
}
extern "C" size_t pcg_conv2d_fwd_bnbwd_thin_workspace_bytes(const pcg_conv_geom* g) {
  return pcg_conv2d_fwd_bnbwd_thin_ok(g) ? thin_conv_fwd_bnbwd_workspace_bytes(g) : 0;
}
extern "C" int pcg_conv2d_fwd_bnbwd_thin(const pcg_conv_geom* g, const float* x, const float* w, const float* z_below, const float* mean,
                                         const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dz,
                                         float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                                         pcg_stream_t stream) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(x && w && z_below && mean && invstd && gamma && beta && dz, "pcg_conv2d_fwd_bnbwd_thin: null pointer");
  PCG_REQUIRE(act == PCG_ACT_NONE || act == PCG_ACT_RELU || act == PCG_ACT_LRELU, "pcg_conv2d_fwd_bnbwd_thin: activation %d is not none / ReLU / LeakyReLU", act);
  PCG_REQUIRE(pcg_conv2d_fwd_bnbwd_thin_ok(g), "pcg_conv2d_fwd_bnbwd_thin: geometry not eligible (pcg_conv2d_fwd_bnbwd_thin_ok)");
  PCG_REQUIRE((((uintptr_t)z_below | (uintptr_t)dz) & 15) == 0, "pcg_conv2d_fwd_bnbwd_thin: tensors must be 16-byte aligned");
  return thin_conv_fwd_bnbwd(g, x, w, z_below, mean, invstd, gamma, beta, act, slope, dz, dgamma, dbeta, accumulate, workspace, workspace_bytes,
                             (hipStream_t)stream);
}
// Grad-input of a full-window Cout = 1 convolution (DCGAN D's last layer, mnist_dcgan.py:110) pushed through the BatchNorm + LeakyReLU backward of
// the layer below it without being written; groups side-by-side batches with their own statistics (mean / invstd [G][C]).
extern "C" int32_t pcg_conv2d_dgrad_bnbwd_full_ok(const pcg_conv_geom* g, int32_t groups) {
  return check_geom(g) == PCG_OK && !pcg::dp_sync_bn() && thin_conv_dgrad_bnbwd_full_ok(g, groups) ? 1 : 0;
}
extern "C" size_t pcg_conv2d_dgrad_bnbwd_full_workspace_bytes(const pcg_conv_geom* g, int32_t groups) {
  return pcg_conv2d_dgrad_bnbwd_full_ok(g, groups) ? thin_conv_dgrad_bnbwd_full_workspace_bytes(g, groups) : 0;
}
extern "C" int pcg_conv2d_dgrad_bnbwd_full(const pcg_conv_geom* g, const float* dy, const float* w, const float* z_below, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, int act, float slope, float* dz,
                                           float* dgamma, float* dbeta, int accumulate, int32_t groups, void* workspace, size_t workspace_bytes,
                                           pcg_stream_t stream) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(dy && w && z_below && mean && invstd && gamma && beta && dz, "pcg_conv2d_dgrad_bnbwd_full: null pointer");
  PCG_REQUIRE(act == PCG_ACT_NONE || act == PCG_ACT_RELU || act == PCG_ACT_LRELU, "pcg_conv2d_dgrad_bnbwd_full: activation %d is not none / ReLU / LeakyReLU", act);
  PCG_REQUIRE(pcg_conv2d_dgrad_bnbwd_full_ok(g, groups), "pcg_conv2d_dgrad_bnbwd_full: geometry / batch not eligible (pcg_conv2d_dgrad_bnbwd_full_ok)");
  PCG_REQUIRE((((uintptr_t)z_below | (uintptr_t)dz | (uintptr_t)w) & 15) == 0, "pcg_conv2d_dgrad_bnbwd_full: tensors must be 16-byte aligned");
  return thin_conv_dgrad_bnbwd_full(g, dy, w, z_below, mean, invstd, gamma, beta, act, slope, dz, dgamma, dbeta, accumulate, groups, workspace,
                                    workspace_bytes, (hipStream_t)stream);
}
// The full-window layer reading the PRE-BatchNorm output of the layer below (forward and weight gradient): see thin_conv.hip FullXf.
extern "C" int32_t pcg_conv2d_bnin_full_ok(const pcg_conv_geom* g, int32_t groups) {
  return check_geom(g) == PCG_OK && thin_conv_bnin_full_ok(g, groups) ? 1 : 0;
}
static int bnin_args(const char* who, const pcg_conv_geom* g, const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     int in_act, int32_t groups) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(z && mean && invstd && gamma && beta, "%s: null pointer", who);
  PCG_REQUIRE(in_act == PCG_ACT_NONE || in_act == PCG_ACT_RELU || in_act == PCG_ACT_LRELU, "%s: input activation %d is not none / ReLU / LeakyReLU", who, in_act);
  PCG_REQUIRE(pcg_conv2d_bnin_full_ok(g, groups), "%s: geometry / batch not eligible (pcg_conv2d_bnin_full_ok)", who);
  return PCG_OK;
}
extern "C" int pcg_conv2d_fwd_bnin_full(const pcg_conv_geom* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, int in_act, float in_slope, int32_t groups, const float* w, const float* bias, int act,
                                        float slope, float* y, pcg_stream_t stream) {
  if (int e = bnin_args("pcg_conv2d_fwd_bnin_full", g, z, mean, invstd, gamma, beta, in_act, groups)) return e;
  PCG_REQUIRE(w && y && act >= PCG_ACT_NONE && act <= PCG_ACT_SIGMOID, "pcg_conv2d_fwd_bnin_full: bad arguments");
  const ThinBnIn bi{mean, invstd, gamma, beta, in_act, in_slope, groups};
  return thin_conv_fwd(g, z, w, bias, y, nullptr, 0, (hipStream_t)stream, act, slope, &bi);
}
extern "C" int pcg_conv2d_wgrad_bnin_full(const pcg_conv_geom* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta, int in_act, float in_slope, int32_t groups, const float* dy, float* dw, int accumulate,
                                          void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  if (int e = bnin_args("pcg_conv2d_wgrad_bnin_full", g, z, mean, invstd, gamma, beta, in_act, groups)) return e;
  PCG_REQUIRE(dy && dw, "pcg_conv2d_wgrad_bnin_full: null pointer");
  const ThinBnIn bi{mean, invstd, gamma, beta, in_act, in_slope, groups};
  return thin_conv_wgrad(g, z, dy, dw, accumulate, workspace, workspace_bytes, (hipStream_t)stream, nullptr, &bi);
}
extern "C" int32_t pcg_conv2d_fwd_bn_partial_rows(const pcg_conv_geom* g) { return check_geom(g) == PCG_OK ? fwd_stat_rows(g) : 0; }
extern "C" int32_t pcg_conv2d_dgrad_bn_partial_rows(const pcg_conv_geom* g) { return check_geom(g) == PCG_OK ? dgrad_stat_rows(g) : 0; }

// Deferred slab reductions (thin_conv.hip): every split-K weight gradient issued on `stream` by THIS thread between begin and flush leaves its
// slabs unreduced; the flush sums all of them in one launch, bit-identical to the per-layer reductions.  The caller keeps each call's
// workspace alive and distinct until the flush.  Replaces the 5-6 us reduction launch behind every weight gradient of a backward sweep
// (mnist_dcgan.py:153,161,173 — the gradients are read by optimizer.step() at :164,176, after the sweep).
extern "C" int pcg_slab_defer_begin(pcg_stream_t stream) { return slab_defer_begin((hipStream_t)stream); }
extern "C" int pcg_slab_defer_flush(pcg_stream_t stream) { return slab_defer_flush((hipStream_t)stream); }
extern "C" int32_t pcg_slab_defer_pending(void) { return slab_defer_pending(); }

extern "C" size_t pcg_conv2d_wgrad_workspace_bytes(const pcg_conv_geom* g) {
  if (check_geom(g) != PCG_OK) return 0;
  if (thin_is_cin(g) || thin_is_cout(g)) return thin_conv_wgrad_workspace_bytes(g);
  const WgradPlan w = plan_wgrad(g);
  return (size_t)w.splits * (size_t)g->Cout * (size_t)g->KH * g->KW * g->Cin * sizeof(float);
}

template <class Cfg, bool XFA, bool XFB>
static int launch_wgrad_x(const ConvP& p, const WgradPlan& wp, int slice_major, hipStream_t s) {
  constexpr size_t smem = smem_bytes<Cfg, false, false>();
  static int once = set_smem(conv_wgrad_kernel<Cfg, XFA, XFB>, smem);
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL((conv_wgrad_kernel<Cfg, XFA, XFB>), slice_major ? dim3((unsigned)wp.tiles * wp.splits) : dim3((unsigned)wp.tiles, wp.splits),
                     dim3(IG_THREADS), smem, s, p, wp.ktiles_total, wp.ktiles_per_split, wp.tiles, slice_major);
  return launch_status("conv_wgrad_kernel");
}
template <class Cfg, bool XFA, bool XFB>
static int launch_wgrad_sk_x(const ConvP& p, const SkPlan& sk, hipStream_t s) {
  constexpr size_t smem = smem_bytes<Cfg, false, false>();
  static int once = set_smem(conv_wgrad_sk_kernel<Cfg, XFA, XFB>, smem);
  if (once != PCG_OK) return once;
  hipLaunchKernelGGL((conv_wgrad_sk_kernel<Cfg, XFA, XFB>), dim3((unsigned)(sk.dp_tiles + sk.sk_blocks)), dim3(IG_THREADS), smem, s, p, sk);
  return launch_status("conv_wgrad_sk_kernel");
}
template <class Cfg>
static int launch_wgrad(const ConvP& p, const WgradPlan& wp, int slice_major, int xf_side, hipStream_t s) {
  if (xf_side == 1) return launch_wgrad_x<Cfg, false, true>(p, wp, slice_major, s);    // x is a transformed activation
  if (xf_side == 2) return launch_wgrad_x<Cfg, true, false>(p, wp, slice_major, s);    // dy is
  return launch_wgrad_x<Cfg, false, false>(p, wp, slice_major, s);
}

extern "C" int pcg_conv2d_wgrad_xf(const pcg_conv_geom* g, const float* x, const pcg_in_xform* xf_x, const float* dy,
                                   const pcg_in_xform* xf_dy, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                                   pcg_stream_t stream) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(x && dy && dw, "pcg_conv2d_wgrad: null pointer");
  const bool hx = xf_x && xf_x->scale, hy = xf_dy && xf_dy->scale;
  PCG_REQUIRE(!(hx && hy), "pcg_conv2d_wgrad_xf: at most one operand can carry an input transform");
  if (thin_is_cin(g) || thin_is_cout(g)) {
    PCG_REQUIRE(!hx && (!hy || thin_conv_xf_ok(g)), "pcg_conv2d_wgrad: input transforms on the MFMA path (Cin > 3 and Cout > 3) and, on the dy side, on the thin forms of pcg_conv2d_xf_thin_ok only");
    PCG_REQUIRE(!hy || (xf_dy->act == PCG_ACT_NONE || xf_dy->act == PCG_ACT_RELU || xf_dy->act == PCG_ACT_LRELU), "pcg_conv2d_wgrad: transform activation %d", hy ? xf_dy->act : 0);
    const ThinXf tx{hy ? xf_dy->scale : nullptr, hy ? xf_dy->shift : nullptr, hy ? act_neg_of(xf_dy->act, xf_dy->slope) : 1.f};
    return thin_conv_wgrad(g, x, dy, dw, accumulate, workspace, workspace_bytes, (hipStream_t)stream, &tx);
  }
  PCG_REQUIRE(g->Cin % 4 == 0 && g->Cout % 4 == 0, "pcg_conv2d_wgrad: Cin=%d and Cout=%d must be multiples of 4", g->Cin, g->Cout);
  const WgradPlan wp = plan_wgrad(g);
  const size_t need = pcg_conv2d_wgrad_workspace_bytes(g);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("pcg_conv2d_wgrad: workspace %zu B < required %zu B", workspace_bytes, need);
    return PCG_ERR_WORKSPACE;
  }
  PCG_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dw & 15) == 0, "pcg_conv2d_wgrad: workspace/dw must be 16-byte aligned");
  ConvP p = make_params(g);
  p.x = x; p.dy = dy; p.out = (float*)workspace;
  p.M = g->Cout; p.N = g->KH * g->KW * g->Cin;
  p.tilesN = ceil_div(p.N, 128);
  if (int e = hx ? set_xform("pcg_conv2d_wgrad_xf", xf_x, g->Cin, &p) : set_xform("pcg_conv2d_wgrad_xf", xf_dy, g->Cout, &p)) return e;
  hipStream_t s = (hipStream_t)stream;
  static const int order_env0 = getenv("PCG_WGRAD_ORDER") ? atoi(getenv("PCG_WGRAD_ORDER")) : -1;   // A/B switch: 0 tile-major, 1 slice-major
  const int order_env = g_tune.wgrad_order >= 0 ? g_tune.wgrad_order : order_env0;
  const int slice_major = order_env >= 0 ? order_env : (wp.tiles <= 8 ? 1 : 0);
  const int side = hx ? 1 : hy ? 2 : 0;
  // a tile count that leaves the chip unevenly loaded: stream-K, straight into dw.  One full-K block per tile anyway (splits == 1:
  // the Linear 8192 -> 1024 gradients, 512 tiles x 8-16 k-tiles): the same kernel with every tile data-parallel — dw written (or
  // accumulated) by the epilogue, no slab and no slab_reduce pass (30 us behind an 84 us kernel in the r03 critic trace)
  if (!wp.narrow && !wp.wide192 && wp.tiles > 96 && sk_mode() != 0) {
    SkPlan sk{};
    bool take = plan_sk(wp.tiles, wp.ktiles_total, s, &sk);
    if (!take && wp.splits == 1) { sk = SkPlan{wp.tiles, 0, 0, wp.ktiles_total, nullptr, nullptr}; take = true; }
    if (take) {
      p.out = dw;
      if (accumulate) { p.epi.mode = EPI_ADD; p.epi.neg = 1.f; p.epi.delta_bytes = 0; }
      return side == 1 ? launch_wgrad_sk_x<Cfg128x128, false, true>(p, sk, s)
           : side == 2 ? launch_wgrad_sk_x<Cfg128x128, true, false>(p, sk, s) : launch_wgrad_sk_x<Cfg128x128, false, false>(p, sk, s);
    }
  }
  int rc;
  if (wp.wide192 && side == 0) {
    p.tilesN = p.N / 192;
    constexpr size_t smem = smem_bytes<Cfg64x192, false, false>();
    static int once = set_smem(conv_wgrad192_kernel, smem);
    if (once != PCG_OK) return once;
    hipLaunchKernelGGL(conv_wgrad192_kernel, slice_major ? dim3((unsigned)wp.tiles * wp.splits) : dim3((unsigned)wp.tiles, wp.splits),
                       dim3(IG_THREADS), smem, s, p, wp.ktiles_total, wp.ktiles_per_split, wp.tiles, slice_major);
    rc = launch_status("conv_wgrad192_kernel");
  } else {
    WgradPlan wq = wp;
    if (wp.wide192) {     // an input transform is pending on an operand: the 128-column tiles carry it (re-plan without the 192 tile)
      wq.wide192 = false;
      wq.tiles = ceil_div(p.M, 64) * ceil_div(p.N, 128);
    }
    rc = wq.narrow ? launch_wgrad<TileCfg<64, 128, 1, 4>>(p, wq, slice_major, side, s) : launch_wgrad<Cfg128x128>(p, wq, slice_major, side, s);
  }
  if (rc != PCG_OK) return rc;
  const size_t n = (size_t)p.M * p.N;
  return launch_slab_reduce((const float*)workspace, dw, n, n, wp.splits, accumulate, s, true);
}

extern "C" int pcg_conv2d_wgrad(const pcg_conv_geom* g, const float* x, const float* dy, float* dw, int accumulate,
                                void* workspace, size_t workspace_bytes, pcg_stream_t stream) {
  return pcg_conv2d_wgrad_xf(g, x, nullptr, dy, nullptr, dw, accumulate, workspace, workspace_bytes, stream);
}

// Which launch form a layer takes (the host-side planning above, no device needed): "thin", "128x128 x tiles [x K-slices]",
// "64x128 tiles", "stream-K: ...", "GEMM + col2im", ... — the decisions DESIGN.md section 3.1.1 tabulates, as text.
// op: 0 forward, 1 grad-input, 2 grad-weight.  assume_scratch: plan as if the stream had stream-K scratch registered.
extern "C" int pcg_conv_plan_describe(const pcg_conv_geom* g, int32_t op, int32_t assume_scratch, char* out, size_t out_bytes) {
  if (int e = check_geom(g)) return e;
  PCG_REQUIRE(out && out_bytes > 0 && op >= 0 && op <= 2, "pcg_conv_plan_describe: bad arguments");
  const int have = assume_scratch ? (sk_mode() != 0 ? 1 : 0) : (have_sk_scratch() ? 1 : 0);
  char buf[512];
  buf[0] = 0;
  auto say = [&](const char* fmt, auto... a) { snprintf(buf + strlen(buf), sizeof(buf) - strlen(buf), fmt, a...); };
  if (thin_is_cin(g) || thin_is_cout(g)) {
    say("%s", "thin (Cin or Cout <= 3): no matrix-core launch");
  } else if (op == 0) {
    const int M = g->B * g->OH * g->OW, N = g->Cout, kt = g->KH * g->KW * ceil_div(g->Cin, IG_BK);
    const FwdPlan f = plan_fwd(g, have);
    SkPlan sk{};
    if (N <= 64) say("128x64 tiles: %d", ceil_div(M, 128));
    else if (fwd_use_t64(M, N, f.splits)) say("64x128 tiles: %d", ceil_div(M, 64) * ceil_div(N, 128));
    else if (f.splits == 1 && have && plan_sk_shape(ceil_div(M, 128) * ceil_div(N, 128), kt, &sk))
      say("stream-K: %d whole tiles + %d tiles x %d k-tiles over %d ranges", sk.dp_tiles, sk.sk_tiles, kt, sk.sk_blocks);
    else if (f.splits > 1) say("128x128 tiles: %d x %d K-slices of %d k-tiles (slabs)", ceil_div(M, 128) * ceil_div(N, 128), f.splits, f.ktiles_per_split);
    else say("128x128 tiles: %d", ceil_div(M, 128) * ceil_div(N, 128));
  } else if (op == 1) {
    if (dgrad_as_gemm(g, have)) {
      const int M = g->B * g->OH * g->OW, N = g->KH * g->KW * g->Cin, kt = ceil_div(g->Cout, IG_BK), tiles = ceil_div(M, 128) * ceil_div(N, 128);
      SkPlan sk{};
      say("%s", "GEMM + col2im: ");
      if (have && plan_sk_shape(tiles, kt, &sk)) say("stream-K: %d whole tiles + %d tiles x %d k-tiles over %d ranges", sk.dp_tiles, sk.sk_tiles, kt, sk.sk_blocks);
      else say("128x128 tiles: %d", tiles);
    } else {
      DgradPhases ph{};
      int maxMp = 0;
      const int nph = build_phases(g, &ph, &maxMp);
      const int N = g->Cin, tilesN = ceil_div(N, N > 64 ? 128 : 64);
      bool uniform = true;
      for (int i = 0; i < nph; ++i)
        uniform = uniform && ph.p[i].Mp == ph.p[0].Mp && ph.p[i].nth * ph.p[i].ntw == ph.p[0].nth * ph.p[0].ntw && ph.p[i].nth > 0 && ph.p[i].ntw > 0;
      SkNPlan sn{};
      SkPlan sk{};
      say("%d phase%s: ", nph, nph == 1 ? "" : "s");
      if (N > 64 && have && !uniform && plan_skn_shape(ph, nph, 128, tilesN, g->Cout, &sn))
        say("stream-K over unequal phases: %d tiles, %d k-tile iterations over %d ranges", sn.tile0[nph], sn.total, sn.blocks);
      else if (N > 64 && have && uniform && nph > 0 &&
               plan_sk_shape(ceil_div(maxMp, 128) * tilesN * nph, ph.p[0].nth * ph.p[0].ntw * ceil_div(g->Cout, IG_BK), &sk))
        say("stream-K: %d whole tiles + %d tiles over %d ranges", sk.dp_tiles, sk.sk_tiles, sk.sk_blocks);
      else say("%s tiles: %d per phase", N > 64 ? "128x128" : "128x64", ceil_div(maxMp, 128) * tilesN);
    }
  } else {
    const WgradPlan w = plan_wgrad(g);
    SkPlan sk{};
    if (!w.narrow && !w.wide192 && w.tiles > 96 && have && plan_sk_shape(w.tiles, w.ktiles_total, &sk))
      say("stream-K: %d whole tiles + %d tiles x %d k-tiles over %d ranges, dw written by the epilogue", sk.dp_tiles, sk.sk_tiles, w.ktiles_total, sk.sk_blocks);
    else if (!w.narrow && !w.wide192 && w.tiles > 96 && w.splits == 1 && sk_mode() != 0) say("128x128 tiles: %d, dw written by the epilogue", w.tiles);
    else say("%s tiles: %d x %d K-slices of %d k-tiles (slabs + slab_reduce)", w.wide192 ? "64x192" : w.narrow ? "64x128" : "128x128", w.tiles, w.splits, w.ktiles_per_split);
  }
  snprintf(out, out_bytes, "%s", buf);
  return PCG_OK;
}

extern "C" int pcg_tune_set(const char* name, int32_t value) {
  PCG_REQUIRE(name != nullptr, "pcg_tune_set: null name");
  if (!strcmp(name, "korder")) g_tune.korder = value;
  else if (!strcmp(name, "edge_prio")) g_tune.edge_prio = value;
  else if (!strcmp(name, "dgrad_swz3")) g_tune.dgrad_swz3 = value;
  else if (!strcmp(name, "wgrad_rounds")) g_tune.wgrad_rounds = value;  // A/B: K-slices of the weight gradient sized for this many rounds of 512 workgroups      // A/B: multi-phase N <= 64 grad-inputs on the three-per-CU tile configuration
  else if (!strcmp(name, "wgrad_order")) g_tune.wgrad_order = value;
  else if (!strcmp(name, "dgrad_interleave")) g_tune.dgrad_interleave = value;
  else if (!strcmp(name, "persistent")) g_tune.persistent = value;
  else if (!strcmp(name, "persist_tiles")) g_tune.persist_tiles = value;
  else if (!strcmp(name, "fwd_splits")) g_tune.fwd_splits = value;
  else if (!strcmp(name, "stream_k")) g_tune.stream_k = value;
  else if (!strcmp(name, "sk_blocks")) g_tune.sk_blocks = value;
  else if (!strcmp(name, "dgrad_gemm")) g_tune.dgrad_gemm = value;
  else if (!strcmp(name, "t64")) g_tune.t64 = value;
  else if (!strcmp(name, "dma")) g_tune.dma = value;
  else {
    set_error("pcg_tune_set: unknown switch '%s' (korder, edge_prio, dgrad_swz3, wgrad_rounds, wgrad_order, dgrad_interleave, persistent, persist_tiles, fwd_splits, "
              "stream_k, sk_blocks, dgrad_gemm, t64, dma)", name);
    return PCG_ERR_INVALID;
  }
  return PCG_OK;
}

// Diagnostic builds (make stamp: -DPCG_CLOCK_STAMP): where the conv kernels leave their per-block clock stamps; the shipped library
// carries the pointer but compiles no stamp code.  Returns 1 if this build stamps, 0 if not.
extern "C" int pcg_debug_stamp_buffer(void* buf, int64_t bytes) {
  g_tune.stamps = static_cast<unsigned long long*>(buf);
  g_tune.stamp_slots = buf ? (int)(bytes / 16) : 0;
#ifdef PCG_CLOCK_STAMP
  return 1;
#else
  return 0;
#endif
}
