// conv_loaders.h — operand gathers of the implicit-GEMM convolutions.
//
// Every gather is a branch-free `buffer_load_dwordx4`: the tensor is described by a raw buffer resource whose
// num_records is its size in bytes, and an out-of-range byte offset makes the hardware return zeros — which is
// exactly what zero padding, ragged tile edges and K tails need.  Per k-tile a lane therefore spends ~4 VALU
// instructions per 16-byte load (add the tile's uniform delta to a per-row base offset, test one bit of a
// per-row tap-validity mask, select the offset or the out-of-range sentinel): the main loop stays one basic block
// the scheduler can interleave with the MFMA stream.  Offsets are 32-bit with wrap-around arithmetic (a base can
// be "negative" for a padded row; base + delta of a valid tap is the true offset), so every operand tensor must be
// smaller than 2 GiB — checked on the host.
#pragma once
#include "igemm_core.h"

namespace pcg {

// bit (a*nw + c) set for lo_h <= a < hi_h (clipped to [0,nh)) and lo_w <= c < hi_w (clipped to [0,nw)); nh*nw <= 32
__device__ __forceinline__ uint32_t tap_mask(int lo_h, int hi_h, int nh, int lo_w, int hi_w, int nw) {
  lo_h = lo_h < 0 ? 0 : lo_h; hi_h = hi_h > nh ? nh : hi_h;
  lo_w = lo_w < 0 ? 0 : lo_w; hi_w = hi_w > nw ? nw : hi_w;
  if (lo_h >= hi_h || lo_w >= hi_w) return 0u;
  const uint32_t rowbits = ((1u << (hi_w - lo_w)) - 1u) << lo_w;   // hi_w - lo_w <= nw <= 32; nw == 32 only if nh == 1
  uint32_t m = 0;
  for (int a = lo_h; a < hi_h; ++a) m |= rowbits << (a * nw);
  return (hi_w - lo_w) >= 32 ? 0xFFFFFFFFu : m;
}

// ---- parameter blocks (kernel arguments) -----------------------------------------------------------------
struct ConvP {
  const float* x;    // fwd / wgrad: input activations
  const float* w;
  const float* bias; // fwd: per-Cout       dgrad: per-Cin (nullable)
  float* out;        // fwd: y              dgrad: dx                     wgrad: slab base
  const float* dy;   // dgrad / wgrad
  double* stat_partial; // nullable: fused BatchNorm statistics of the output, fp64 [partial row][2][N]
  int act; float slope; // activation fused into the epilogue (PCG_ACT_NONE: none; never together with stat_partial)
  EpiAux epi;           // backward-pass epilogue (mode EPI_NONE: off); EPI_BNBWD writes its column sums to stat_partial
  // input transform (in_sc != nullptr): the ACTIVATION operand of this launch (fwd: x; dgrad: dy; wgrad: x or dy, see the kernel's
  // template flags) is read as act(v * in_sc[c] + in_sh[c]) — the producing layer's BatchNorm + ReLU / LeakyReLU applied inside
  // the gather, so that layer's output is never written in its activated form.  Padding / tile-edge zeros stay zeros.
  const float* in_sc; const float* in_sh; float in_neg; uint32_t in_c_bytes;
  uint32_t x_bytes, w_bytes, dy_bytes;
  int B, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int M, N;          // GEMM extents of this launch
  int tilesN;
  int ktiles;        // fwd: KH*KW*ceil(Cin/32)
  int ktiles_per_split;  // fwd split-K (gridDim.y slabs of M*N floats at `out`); == ktiles when not split
  int korder;        // order of the k-tiles of fwd / dgrad (see FwdKIter): 0 = (tap, channel chunk), 1 = L2-friendly (default)
  int edge_prio;     // issue priority outside the main loops (r04): bit 0 — every wave enters at priority 3 (loader set-up, first
                     // gathers), bit 1 — the consumers run the epilogue at priority 3 (was 0: starved by the other workgroups' loops)
  unsigned long long* stamps;  // diagnostic builds only (-DPCG_CLOCK_STAMP, pcg_debug_stamp_buffer): per block {shader-clock ticks,
  int stamp_slots;             // 100 MHz ticks} of consumer wave 0's main loop; the shipped library compiles no stamp code
  FastDiv dOW, dOH;  // fwd/wgrad pixel decomposition
};

// Order of the forward k-tiles.  A k-tile is (tap, 32-channel chunk); which order the sum over them runs in is free (A and B
// loaders share this iterator), but it decides what the XCD's 4 MB L2 sees: an input line [pixel][32 channels] is wanted by
// KH*KW/stride^2 taps of the tile (4 for k4 s2, 9 for 3x3 s1).
//   korder 0: taps in raster order, chunks inside a tap.  The re-reads of a line are 2..4 k-tiles (kw + stride) and 8..16 k-tiles
//             (kh + stride) apart — with 64 workgroups per XCD streaming 32 KB per k-tile the far ones miss L2: measured r02,
//             the forward kernels fetched 2.6x their input bytes.
//   korder 1: taps grouped by their stride-parity class (kh % s, kw % s) — the taps of a class read the SAME input pixels shifted
//             by whole output steps — then channel chunk, then the taps of the class: all re-reads of a line are 1..3 k-tiles apart.
struct FwdKIter {
  int KH, KW, Cin, S, mode;
  int ch, cw, jh, jw, ci0;
  __device__ __forceinline__ void init(const ConvP& p) {
    KH = p.KH; KW = p.KW; Cin = p.Cin; mode = p.korder; S = mode ? p.stride : 1;
    ch = cw = jh = jw = ci0 = 0;
  }
  __device__ __forceinline__ int kh() const { return ch + S * jh; }
  __device__ __forceinline__ int kw() const { return cw + S * jw; }
  __device__ __forceinline__ bool next_tap_in_class() {
    if (cw + S * (++jw) < KW) return true;
    jw = 0;
    if (ch + S * (++jh) < KH) return true;
    jh = 0;
    return false;
  }
  __device__ __forceinline__ void next_class() {
    if (++cw >= S || cw >= KW) { cw = 0; ++ch; }
  }
  __device__ __forceinline__ void advance() {
    if (mode == 0) {                       // (tap, chunk)
      ci0 += IG_BK;
      if (ci0 < Cin) return;
      ci0 = 0;
      next_tap_in_class();                 // S == 1: one class holding every tap, raster order
    } else {                               // (class, chunk, tap of the class)
      if (next_tap_in_class()) return;
      ci0 += IG_BK;
      if (ci0 < Cin) return;
      ci0 = 0;
      next_class();
    }
  }
  // split-K: start at k-tile kt.  Closed form (a loop of advance() cost 0.1 us per skipped k-tile: 29 us in front of the last
  // K-slice of WGAN-GP's Linear 8192 -> 1024, measured r03 with the phase stamps).
  __device__ __forceinline__ void seek(int kt) {
    const int nchunks = (Cin + IG_BK - 1) / IG_BK;
    if (mode == 0) {                       // (tap, chunk), taps in raster order
      const int tap = kt / nchunks;
      ci0 = (kt - tap * nchunks) * IG_BK;
      jh = tap / KW; jw = tap - jh * KW;
      return;
    }
    // (class, chunk, tap of the class): at most S*S classes of nh(ch) * nw(cw) taps each
    for (;;) {
      const int nh = (KH - ch + S - 1) / S, nw = (KW - cw + S - 1) / S, per = nh * nw;
      const int in_class = nchunks * per;
      if (kt < in_class || per <= 0) {
        const int c = per > 0 ? kt / per : 0, t = kt - c * per;
        ci0 = c * IG_BK;
        jh = t / nw; jw = t - jh * nw;
        return;
      }
      kt -= in_class;
      next_class();
    }
  }
};

struct PhaseInfo {
  int ph, pw;          // phase offsets (ih % s, iw % s)
  int PHh, PHw;        // phase grid
  int Mp;              // B*PHh*PHw
  int kh0, kw0;        // first tap of the phase
  int nth, ntw;        // taps per axis
  int dh0, dw0;        // oh = a + dh0 - jh ; ow = c + dw0 - jw
  int prow0;           // first partial-statistics row of this phase
  FastDiv dPHw, dPHh;
};
struct DgradPhases { PhaseInfo p[4]; int interleave; };   // interleave = number of equal-sized phases sharing a 1-D grid, or 0

// Input transform of a K-major activation gather (k = channel): the k-tile's channel quad changes every tile, so its scale /
// shift quads are fetched with the tile (two more 16-byte loads, L1/L2 hits on a <= 4 KB table) and applied between the arrival
// of the gathered registers and the ds_write; `ok` bits remember which rows were real (zero padding must stay zero).
struct XfK {
  rsrc_t rsc, rsh;
  float4 sc, sh;
  float neg;
  uint32_t ok;
  __device__ __forceinline__ void init(const ConvP& p) {
    rsc = make_rsrc(p.in_sc, p.in_c_bytes); rsh = make_rsrc(p.in_sh, p.in_c_bytes); neg = p.in_neg; ok = 0;
  }
  __device__ __forceinline__ void fetch(uint32_t c_off_bytes, bool kok) {
    sc = buf_load4(rsc, kok ? c_off_bytes : OOB_OFF);
    sh = buf_load4(rsh, kok ? c_off_bytes : OOB_OFF);
  }
};
__device__ __forceinline__ float4 xf_apply(float4 v, float4 sc, float4 sh, float neg, bool ok) {
  float4 t = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
  t.x = act_neg_scale(t.x, neg); t.y = act_neg_scale(t.y, neg); t.z = act_neg_scale(t.z, neg); t.w = act_neg_scale(t.w, neg);
  return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- forward: A = im2col rows of x (K-major), B = OHWI weight rows (K-major) --------------------------------
template <int ROWS_, bool XF = false>
struct FwdALoader {
  static constexpr bool KMAJOR = true, XFORM = XF;
  static constexpr int ROWS = ROWS_, NV = ROWS_ / 32;
  rsrc_t rs;
  uint32_t base[NV], mask[NV];
  int IW, Cin, KW, kq4;
  FwdKIter it;
  XfK xf;

  // src_swz: LDS-DMA staging (igemm_produce_dma) — the destination of a lane is fixed (base + lane * 16), so the XOR swizzle of the
  // unpadded K-major image is applied on the SOURCE side: the lane at chunk position tid & 7 of row r fetches the logical chunk
  // (tid & 7) ^ ((r >> 1) & 7), the same involution LdsImage<.., SWZ>::frag applies on the read side
  __device__ __forceinline__ FwdALoader(const ConvP& p, int m_block, int tid, bool src_swz = false) {
    if constexpr (XF) xf.init(p);
    rs = make_rsrc(p.x, p.x_bytes);
    IW = p.IW; Cin = p.Cin; KW = p.KW;
    it.init(p);
    kq4 = (src_swz ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7)) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int m = m_block + r0 + 32 * i;
      uint32_t mk = 0, bs = OOB_OFF;
      if (m < p.M) {
        uint32_t t, ow, b, oh;
        p.dOW.divmod((uint32_t)m, t, ow);
        p.dOH.divmod(t, b, oh);
        const int ih0 = (int)oh * p.stride - p.pad, iw0 = (int)ow * p.stride - p.pad;
        bs = (uint32_t)(((((int)b * p.IH + ih0) * IW + iw0) * Cin + kq4) * 4);
        mk = tap_mask(-ih0, p.IH - ih0, p.KH, -iw0, IW - iw0, KW);
      }
      base[i] = bs; mask[i] = mk;
    }
  }
  // start at k-tile kt (split-K)
  __device__ __forceinline__ void seek(int kt) { it.seek(kt); }
  // byte offsets of the next k-tile's NV gathers (OOB_OFF where the hardware is to deliver zeros); advances to the following k-tile
  __device__ __forceinline__ void next_offsets(uint32_t (&off)[NV]) {
    const int kh = it.kh(), kw = it.kw(), ci0 = it.ci0;
    const int tap = kh * KW + kw;
    const uint32_t delta = (uint32_t)(((kh * IW + kw) * Cin + ci0) * 4);
    const bool kok = kq4 < Cin - ci0;
#pragma unroll
    for (int i = 0; i < NV; ++i) off[i] = (kok && ((mask[i] >> tap) & 1u)) ? base[i] + delta : OOB_OFF;
    it.advance();
  }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    const int kh = it.kh(), kw = it.kw(), ci0 = it.ci0;
    const int tap = kh * KW + kw;
    const uint32_t delta = (uint32_t)(((kh * IW + kw) * Cin + ci0) * 4);
    const bool kok = kq4 < Cin - ci0;
    uint32_t okbits = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool ok = kok && ((mask[i] >> tap) & 1u);
      v[i] = buf_load4(rs, ok ? base[i] + delta : OOB_OFF);
      if constexpr (XF) okbits |= (ok ? 1u : 0u) << i;
    }
    if constexpr (XF) { xf.ok = okbits; xf.fetch((uint32_t)((ci0 + kq4) * 4), kok); }
    it.advance();
  }
  // applied to the registers of the tile fetched by the LAST load_next, right before they are written to LDS
  __device__ __forceinline__ void transform(float4 (&v)[NV]) {
    if constexpr (XF) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = xf_apply(v[i], xf.sc, xf.sh, xf.neg, (xf.ok >> i) & 1u);
    }
  }
};

template <int ROWS_>
struct FwdBLoader {
  static constexpr bool KMAJOR = true, XFORM = false;
  static constexpr int ROWS = ROWS_, NV = ROWS_ / 32;
  rsrc_t rs;
  uint32_t base[NV];
  int Cin, KW, kq4;
  FwdKIter it;

  __device__ __forceinline__ FwdBLoader(const ConvP& p, int n_block, int tid, bool src_swz = false) {
    rs = make_rsrc(p.w, p.w_bytes);
    Cin = p.Cin; KW = p.KW; kq4 = (src_swz ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7)) * 4;
    it.init(p);
    const int r0 = tid >> 3;
    const int Ktot = p.KH * p.KW * p.Cin;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int n = n_block + r0 + 32 * i;
      base[i] = n < p.N ? (uint32_t)((n * Ktot + kq4) * 4) : OOB_OFF;  // OOB_OFF + delta stays out of range
    }
  }
  __device__ __forceinline__ void seek(int kt) { it.seek(kt); }
  __device__ __forceinline__ void next_offsets(uint32_t (&off)[NV]) {
    const int ci0 = it.ci0;
    const uint32_t delta = (uint32_t)(((it.kh() * KW + it.kw()) * Cin + ci0) * 4);
    const bool kok = kq4 < Cin - ci0;
#pragma unroll
    for (int i = 0; i < NV; ++i) off[i] = kok ? base[i] + delta : OOB_OFF;
    it.advance();
  }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    const int ci0 = it.ci0;
    const uint32_t delta = (uint32_t)(((it.kh() * KW + it.kw()) * Cin + ci0) * 4);
    const bool kok = kq4 < Cin - ci0;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = buf_load4(rs, kok ? base[i] + delta : OOB_OFF);
    it.advance();
  }
  __device__ __forceinline__ void transform(float4 (&)[NV]) {}
};

// ---- dgrad: per sub-pixel phase; A = dy rows gathered (K-major, k = co), B = w[co][tap][ci] slices (MN-major) ----
// k-tiles of a grad-input phase: (jh, jw, co-chunk) with the chunk innermost (mode 0), or (co-chunk, jh, jw) with the taps innermost
// (mode 1, default): the taps of a phase read the same dy pixels shifted by one — with the taps innermost a dy line [pixel][32 co]
// is re-read in the NEXT k-tiles instead of Cout/32 k-tiles later (see FwdKIter).
struct DgradTapIter {
  int Cout, nth, ntw, jw, co0, jh, mode;
  __device__ __forceinline__ void init(int Cout_, int nth_, int ntw_, int mode_) {
    Cout = Cout_; nth = nth_; ntw = ntw_; mode = mode_; jh = 0; jw = 0; co0 = 0;
  }
  __device__ __forceinline__ void advance() {
    if (mode == 0) {
      co0 += IG_BK;
      if (co0 >= Cout) { co0 = 0; if (++jw == ntw) { jw = 0; ++jh; } }
    } else {
      if (++jw < ntw) return;
      jw = 0;
      if (++jh < nth) return;
      jh = 0;
      co0 += IG_BK;
    }
  }
  __device__ __forceinline__ void seek(int kt) {       // start at k-tile kt of the phase (stream-K segments)
    const int ntaps = nth * ntw;
    int tap;
    if (mode == 0) {
      const int nchunks = (Cout + IG_BK - 1) / IG_BK;
      tap = kt / nchunks;
      co0 = (kt - tap * nchunks) * IG_BK;
    } else {
      const int c = kt / ntaps;
      tap = kt - c * ntaps;
      co0 = c * IG_BK;
    }
    jh = tap / ntw; jw = tap - jh * ntw;
  }
};

template <int ROWS_, bool XF = false>
struct DgradALoader {
  static constexpr bool KMAJOR = true, XFORM = XF;
  static constexpr int ROWS = ROWS_, NV = ROWS_ / 32;
  rsrc_t rs;
  uint32_t base[NV], mask[NV];
  int OW, Cout, kq4;
  DgradTapIter it;
  XfK xf;

  __device__ __forceinline__ DgradALoader(const ConvP& p, const PhaseInfo& f, int m_block, int tid) {
    if constexpr (XF) xf.init(p);
    rs = make_rsrc(p.dy, p.dy_bytes);
    OW = p.OW; Cout = p.Cout;
    kq4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int m = m_block + r0 + 32 * i;
      uint32_t mk = 0, bs = OOB_OFF;
      if (m < f.Mp) {
        uint32_t t, cc, b, aa;
        f.dPHw.divmod((uint32_t)m, t, cc);
        f.dPHh.divmod(t, b, aa);
        const int oh0 = (int)aa + f.dh0, ow0 = (int)cc + f.dw0;  // tap (jh, jw) reads (oh0 - jh, ow0 - jw)
        bs = (uint32_t)(((((int)b * p.OH + oh0) * OW + ow0) * Cout + kq4) * 4);
        // tap (a, c) reads (oh0 - a, ow0 - c): valid for oh0 - OH < a <= oh0
        mk = tap_mask(oh0 - p.OH + 1, oh0 + 1, f.nth, ow0 - OW + 1, ow0 + 1, f.ntw);
      }
      base[i] = bs; mask[i] = mk;
    }
    it.init(Cout, f.nth, f.ntw, p.korder);
  }
  __device__ __forceinline__ void seek(int kt) { it.seek(kt); }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    const int tap = it.jh * it.ntw + it.jw;
    const uint32_t delta = (uint32_t)((it.co0 - (it.jh * OW + it.jw) * Cout) * 4);
    const bool kok = kq4 < Cout - it.co0;
    uint32_t okbits = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool ok = kok && ((mask[i] >> tap) & 1u);
      v[i] = buf_load4(rs, ok ? base[i] + delta : OOB_OFF);
      if constexpr (XF) okbits |= (ok ? 1u : 0u) << i;
    }
    if constexpr (XF) { xf.ok = okbits; xf.fetch((uint32_t)((it.co0 + kq4) * 4), kok); }
    it.advance();
  }
  __device__ __forceinline__ void transform(float4 (&v)[NV]) {
    if constexpr (XF) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = xf_apply(v[i], xf.sc, xf.sh, xf.neg, (xf.ok >> i) & 1u);
    }
  }
};

template <int ROWS_>
struct DgradBLoader {
  static constexpr bool KMAJOR = false, XFORM = false;
  static constexpr int ROWS = ROWS_, NV = ROWS_ / 32;
  static constexpr int C4 = ROWS_ / 4, KR = IG_LOADERS / C4;
  rsrc_t rs;
  uint32_t base[NV];
  int Cin, KHKW, KW, stride, kh0, kw0, kr0;
  DgradTapIter it;

  __device__ __forceinline__ DgradBLoader(const ConvP& p, const PhaseInfo& f, int n_block, int tid) {
    rs = make_rsrc(p.w, p.w_bytes);
    Cin = p.Cin; KHKW = p.KH * p.KW; KW = p.KW; stride = p.stride; kh0 = f.kh0; kw0 = f.kw0;
    const int c4 = tid % C4;
    kr0 = tid / C4;
    const int n = n_block + 4 * c4;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      base[i] = n < p.N ? (uint32_t)(((mn_krow<NV>(kr0, i) * KHKW) * Cin + n) * 4) : OOB_OFF;
    it.init(p.Cout, f.nth, f.ntw, p.korder);
  }
  __device__ __forceinline__ void seek(int kt) { it.seek(kt); }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    const int tap = (kh0 + stride * it.jh) * KW + kw0 + stride * it.jw;
    const uint32_t delta = (uint32_t)(((it.co0 * KHKW + tap) * Cin) * 4);
    const int krem = it.Cout - it.co0;  // rows kr < krem are real output channels
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = buf_load4(rs, mn_krow<NV>(kr0, i) < krem ? base[i] + delta : OOB_OFF);
    it.advance();
  }
  __device__ __forceinline__ void transform(float4 (&)[NV]) {}
};

// ---- wgrad: k = output pixel; A = dy[pixel][co] (MN-major), B = x gathered at the tap's shift (MN-major) -------
// MN-major activation gathers (wgrad): a thread's channel quad is fixed for the whole kernel, so the transform's scale / shift
// live in registers; XF = the operand is an activation to be read as act(v*sc + sh).
template <int ROWS_, bool XF = false>
struct WgradALoader {
  static constexpr bool KMAJOR = false, XFORM = XF;
  static constexpr int ROWS = ROWS_, NV = ROWS_ / 32;
  static constexpr int C4 = ROWS_ / 4, KR = IG_LOADERS / C4;
  rsrc_t rs;
  uint32_t base[NV];
  int Cout, K, q0, kr0;
  float4 sc, sh; float neg; uint32_t okb; bool mok;

  __device__ __forceinline__ WgradALoader(const ConvP& p, int m_block, int kt_begin, int tid) {
    rs = make_rsrc(p.dy, p.dy_bytes);
    Cout = p.Cout; K = p.B * p.OH * p.OW;
    const int c4 = tid % C4;
    kr0 = tid / C4;
    const int m = m_block + 4 * c4;
    mok = m < p.M;
#pragma unroll
    for (int i = 0; i < NV; ++i) base[i] = mok ? (uint32_t)((mn_krow<NV>(kr0, i) * Cout + m) * 4) : OOB_OFF;
    q0 = kt_begin * IG_BK;
    if constexpr (XF) {
      neg = p.in_neg; okb = 0;
      sc = mok ? *reinterpret_cast<const float4*>(p.in_sc + m) : make_float4(0.f, 0.f, 0.f, 0.f);
      sh = mok ? *reinterpret_cast<const float4*>(p.in_sh + m) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    const uint32_t delta = (uint32_t)(q0 * Cout * 4);
    const int krem = K - q0;
    uint32_t okbits = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool ok = mn_krow<NV>(kr0, i) < krem;
      v[i] = buf_load4(rs, ok ? base[i] + delta : OOB_OFF);
      if constexpr (XF) okbits |= ((ok && mok) ? 1u : 0u) << i;
    }
    if constexpr (XF) okb = okbits;
    q0 += IG_BK;
  }
  __device__ __forceinline__ void transform(float4 (&v)[NV]) {
    if constexpr (XF) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = xf_apply(v[i], sc, sh, neg, (okb >> i) & 1u);
    }
  }
};

template <int ROWS_, bool XF = false>
struct WgradBLoader {
  static constexpr bool KMAJOR = false, XFORM = XF;
  static constexpr int ROWS = ROWS_, NV = ROWS_ / 32;
  static constexpr int C4 = ROWS_ / 4, KR = IG_LOADERS / C4;
  // 128-column tiles: the 64 lanes of a wave own 32 column quads x TWO row groups (lanes 0..31 / 32..63), i.e. only two different
  // pixels per load instruction.  Their (image, oh, ow) decomposition and base offset are computed ONCE per wave on the scalar unit
  // and selected per half; a lane adds its own (tap, channel) part.  r03 stamps of the weight gradient: its producers spent 75 % of
  // the loop issuing gathers (address arithmetic) and the consumers waited 18 % of theirs at the barrier.
  static constexpr bool SCALAR_PIX = PCG_MN_CONSEC && C4 == 32;
  rsrc_t rs;
  int IH, IW, Cin, stride, K, q0, kr0, OH, OW, dh, dw, ci;  // dh = kh - pad
  int lane_off;      // SCALAR_PIX: ((dh * IW + dw) * Cin + ci) * 4, this lane's part of every offset
  bool nok;
  FastDiv dOW, dOH;
  float4 sc, sh; float neg; uint32_t okb;

  __device__ __forceinline__ WgradBLoader(const ConvP& p, int n_block, int kt_begin, int tid) {
    rs = make_rsrc(p.x, p.x_bytes);
    IH = p.IH; IW = p.IW; Cin = p.Cin; stride = p.stride; K = p.B * p.OH * p.OW;
    dOW = p.dOW; dOH = p.dOH;
    const int c4 = tid % C4;
    const int n = n_block + 4 * c4;
    nok = n < p.N;
    const int tap = nok ? n / Cin : 0;
    ci = nok ? n - tap * Cin : 0;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    dh = kh - p.pad; dw = kw - p.pad;
    lane_off = ((dh * IW + dw) * Cin + ci) * 4;
    q0 = kt_begin * IG_BK; kr0 = tid / C4; OH = p.OH; OW = p.OW;
    if constexpr (XF) {
      neg = p.in_neg; okb = 0;
      sc = nok ? *reinterpret_cast<const float4*>(p.in_sc + ci) : make_float4(0.f, 0.f, 0.f, 0.f);
      sh = nok ? *reinterpret_cast<const float4*>(p.in_sh + ci) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    uint32_t okbits = 0;
    if constexpr (SCALAR_PIX) {
      const bool hi = (threadIdx.x & 32) != 0;                                   // second row group of the wave
      const int qw = __builtin_amdgcn_readfirstlane(q0 + NV * (kr0 & ~1));      // first pixel of the wave's first row group
      uint32_t t, ow[2], b[2], oh[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {                                              // (wave-uniform: scalar multiplies and shifts)
        dOW.divmod((uint32_t)(qw + NV * h), t, ow[h]);
        dOH.divmod(t, b[h], oh[h]);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int pix[2], ihs[2], iws[2];
        bool in[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (i > 0) {
            ++ow[h];
            if (ow[h] == (uint32_t)OW) { ow[h] = 0; ++oh[h]; if (oh[h] == (uint32_t)OH) { oh[h] = 0; ++b[h]; } }
          }
          ihs[h] = (int)oh[h] * stride; iws[h] = (int)ow[h] * stride;
          pix[h] = (((int)b[h] * IH + ihs[h]) * IW + iws[h]) * Cin * 4;
          in[h] = qw + NV * h + i < K;
        }
        const int ih = (hi ? ihs[1] : ihs[0]) + dh, iw = (hi ? iws[1] : iws[0]) + dw;
        const bool ok = nok && (hi ? in[1] : in[0]) && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
        v[i] = buf_load4(rs, ok ? (uint32_t)((hi ? pix[1] : pix[0]) + lane_off) : OOB_OFF);
        if constexpr (XF) okbits |= (ok ? 1u : 0u) << i;
      }
    } else {
      uint32_t t, ow, b, oh;
      if constexpr (PCG_MN_CONSEC) {           // this thread's NV pixels are consecutive: decompose the first, carry for the others
        dOW.divmod((uint32_t)(q0 + mn_krow<NV>(kr0, 0)), t, ow);
        dOH.divmod(t, b, oh);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int q = q0 + mn_krow<NV>(kr0, i);
        if constexpr (PCG_MN_CONSEC) {
          if (i > 0) {
            ++ow;
            if (ow == (uint32_t)OW) { ow = 0; ++oh; if (oh == (uint32_t)OH) { oh = 0; ++b; } }
          }
        } else {
          dOW.divmod((uint32_t)q, t, ow);
          dOH.divmod(t, b, oh);
        }
        const int ih = (int)oh * stride + dh, iw = (int)ow * stride + dw;
        const bool ok = nok && q < K && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
        const uint32_t off = (uint32_t)(((((int)b * IH + ih) * IW + iw) * Cin + ci) * 4);
        v[i] = buf_load4(rs, ok ? off : OOB_OFF);
        if constexpr (XF) okbits |= (ok ? 1u : 0u) << i;
      }
    }
    if constexpr (XF) okb = okbits;
    q0 += IG_BK;
  }
  __device__ __forceinline__ void transform(float4 (&v)[NV]) {
    if constexpr (XF) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = xf_apply(v[i], sc, sh, neg, (okb >> i) & 1u);
    }
  }
};

// 192-column variant (the 64x192 weight-gradient tile: N = KH*KW*Cin = 576 = 3 x 192 for the 3x3 64-channel layers, where 128-column
// tiles leave the fifth tile half empty): three 64-column sub-images, see LdsImage.  A thread owns one column quad per sub-image
// (three (tap, ci) pairs) and two k-rows (pixels) per k-tile, whose pixel decomposition is shared by the sub-images.
struct WgradBLoader192 {
  static constexpr bool KMAJOR = false, XFORM = false;
  static constexpr int ROWS = 192, NV = 6;
  rsrc_t rs;
  int IH, IW, Cin, stride, K, q0, kr0, OH, OW, dh[3], dw[3], ci[3];
  bool nok[3];
  FastDiv dOW, dOH;

  __device__ __forceinline__ WgradBLoader192(const ConvP& p, int n_block, int kt_begin, int tid) {
    rs = make_rsrc(p.x, p.x_bytes);
    IH = p.IH; IW = p.IW; Cin = p.Cin; stride = p.stride; K = p.B * p.OH * p.OW;
    dOW = p.dOW; dOH = p.dOH;
    const int c4 = tid & 15;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int n = n_block + 64 * s + 4 * c4;
      nok[s] = n < p.N;
      const int tap = nok[s] ? n / Cin : 0;
      ci[s] = nok[s] ? n - tap * Cin : 0;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      dh[s] = kh - p.pad; dw[s] = kw - p.pad;
    }
    q0 = kt_begin * IG_BK; kr0 = tid >> 4; OH = p.OH; OW = p.OW;
  }
  __device__ __forceinline__ void load_next(float4 (&v)[NV]) {
    uint32_t t, ow, b, oh;
    if constexpr (PCG_MN_CONSEC) {
      dOW.divmod((uint32_t)(q0 + mn_krow<2>(kr0, 0)), t, ow);
      dOH.divmod(t, b, oh);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = q0 + mn_krow<2>(kr0, h);
      if constexpr (PCG_MN_CONSEC) {
        if (h > 0) {
          ++ow;
          if (ow == (uint32_t)OW) { ow = 0; ++oh; if (oh == (uint32_t)OH) { oh = 0; ++b; } }
        }
      } else {
        dOW.divmod((uint32_t)q, t, ow);
        dOH.divmod(t, b, oh);
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ih = (int)oh * stride + dh[s], iw = (int)ow * stride + dw[s];
        const bool ok = nok[s] && q < K && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
        const uint32_t off = (uint32_t)(((((int)b * IH + ih) * IW + iw) * Cin + ci[s]) * 4);
        v[2 * s + h] = buf_load4(rs, ok ? off : OOB_OFF);
      }
    }
    q0 += IG_BK;
  }
  __device__ __forceinline__ void transform(float4 (&)[NV]) {}
};

}  // namespace pcg
