// pcg_common.h — shared host/device helpers for libpcgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pcgan_hip.h"

// Phase timestamps inside a kernel (scripts/probes/seg_timing_probe.hip builds the kernel sources with -DPCG_SEG_TIMING and
// provides pcg_dbg_ts); compiled out of the library.
#ifdef PCG_SEG_TIMING
extern __device__ unsigned long long* pcg_dbg_ts;
#define PCG_T(i) do { if (threadIdx.x == 0 && pcg_dbg_ts) pcg_dbg_ts[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define PCG_T(i) do { } while (0)
#endif

namespace pcg {

// ---- error reporting (thread-local text behind pcg_last_error()) -------------------------------
void set_error(const char* fmt, ...);
int launch_status(const char* what);  // PCG_OK or PCG_ERR_LAUNCH after a kernel launch

#define PCG_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ::pcg::set_error(__VA_ARGS__);           \
      return PCG_ERR_INVALID;                  \
    }                                          \
  } while (0)

// ---- exact unsigned division by a runtime constant (n < 2^31), host-prepared -------------------
// q = umulhi(n, mul) >> shr ; d == 1 handled by mul == 0.
struct FastDiv {
  uint32_t d, mul, shr;
  __host__ __device__ FastDiv() : d(1), mul(0), shr(0) {}
  explicit FastDiv(uint32_t div) : d(div), mul(0), shr(0) {
    if (div > 1) {
      uint32_t l = 0;
      while ((1u << l) < div) ++l;              // ceil(log2 div)
      const unsigned p = 31 + l;
      const uint64_t m = ((1ull << p) + div - 1) / div;
      mul = (uint32_t)m;
      shr = p - 32;
    }
  }
  __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return mul ? (__umulhi(n, mul) >> shr) : n;
#else
    return mul ? (uint32_t)(((uint64_t)n * mul) >> 32) >> shr : n;
#endif
  }
  __host__ __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
    q = div(n);
    r = n - q * d;
  }
};

// ---- XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
// run of tile ids so neighbouring tiles (shared operand panels) hit the same L2.  Bijective for any n.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u;
  const uint32_t xcd = bid & 7u, idx = bid >> 3;
  const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case PCG_ACT_RELU: return v > 0.f ? v : 0.f;
    case PCG_ACT_LRELU: return v > 0.f ? v : v * slope;
    case PCG_ACT_TANH: return tanhf(v);
    case PCG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// ReLU / LeakyReLU / identity as one select (neg = 0 / slope / 1): what the conv epilogues fuse; keeps their register count
__device__ __forceinline__ float act_neg_scale(float v, float neg) { return v > 0.f ? v : v * neg; }
static inline bool act_is_cheap(int act) { return act == PCG_ACT_NONE || act == PCG_ACT_RELU || act == PCG_ACT_LRELU; }
static inline float act_neg_of(int act, float slope) { return act == PCG_ACT_RELU ? 0.f : act == PCG_ACT_LRELU ? slope : 1.f; }

// BatchNorm as one fma per element: y = x*sc + sh.  ONE definition for every place that applies it or recomputes its sign
// (bn_apply_act, the input transforms of the conv gathers, the backward epilogues' masks): they must agree bit for bit.
__device__ __forceinline__ void bn_fold(float gamma, float beta, float mean, float invstd, float& sc, float& sh) {
  sc = gamma * invstd;
  sh = fmaf(-mean, sc, beta);
}

// derivative expressed through the OUTPUT y of the activation
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
  switch (act) {
    case PCG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case PCG_ACT_LRELU: return y > 0.f ? 1.f : slope;
    case PCG_ACT_TANH: return 1.f - y * y;
    case PCG_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace pcg
