// rng.hip — device-side batch synthesis: the per-iteration random draws of the training loops,
// generated on the GPU by a counter-based Philox-4x32-10 stream instead of the reference's host loop / ATen calls:
//   * patch masks            trainer.py:45-72  (per sample: `num_modifiable_patches` of the patch grid, nearest-upsampled)
//                            — the reference's Python loop of B randperm calls costs 18-125 ms per batch on the host
//   * target classes         trainer.py:94     torch.randint(0, num_classes, (bs,))
//   * latent noise           mnist_dcgan.py:156 torch.randn(b_size, z_dim, 1, 1)
// The streams differ from torch's generators (RNG parity is by supplied tensors — SURVEY.md §7 "RNG parity"); what is
// tested is the distribution: exact patch counts, uniform marginals, N(0,1) moments, determinism in (seed, offset).
#include <algorithm>
#include "pcg_common.h"

namespace pcg {
namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = U4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
    k0 += W0; k1 += W1;
  }
  return ctr;
}
__device__ __forceinline__ U4 draw(uint64_t seed, uint64_t offset, uint64_t idx) {
  const uint64_t c = offset + idx;
  return philox4x32_10(U4{(uint32_t)c, (uint32_t)(c >> 32), 0x5eed5eedu, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// one thread per sample: partial Fisher-Yates over <= 64 patches held as a 64-bit "taken" set, then the sample's
// H*W mask is written by the whole block (nearest up-sampling: pixel (h,w) belongs to patch (h/ps, w/ps))
__global__ void __launch_bounds__(256) patch_mask_kernel(float* __restrict__ out, int B, int H, int W, int ps, int nph, int npw,
                                                         int nsel, uint64_t seed, uint64_t offset) {
  __shared__ unsigned long long sel[256];
  const int b0 = blockIdx.x * 256;
  const int b = b0 + threadIdx.x;
  const int total = nph * npw;
  unsigned long long taken = 0ull;
  if (b < B) {
    int remaining = total;
    uint64_t ctr = 0;
    U4 r = draw(seed, offset, (uint64_t)b * 16);
    int have = 4;
    for (int s = 0; s < nsel && s < total; ++s) {
      if (have == 0) { r = draw(seed, offset, (uint64_t)b * 16 + (++ctr)); have = 4; }
      const uint32_t v = have == 4 ? r.x : have == 3 ? r.y : have == 2 ? r.z : r.w;
      --have;
      int k = (int)(((uint64_t)v * (uint64_t)remaining) >> 32);   // uniform in [0, remaining)
      // the k-th patch that is not yet taken
      int pidx = 0;
      for (; pidx < total; ++pidx) {
        if (!((taken >> pidx) & 1ull)) { if (k == 0) break; --k; }
      }
      taken |= 1ull << pidx;
      --remaining;
    }
  }
  sel[threadIdx.x] = taken;
  __syncthreads();
  const int nb = (B - b0) < 256 ? (B - b0) : 256;
  const int HW = H * W;
  for (int i = threadIdx.x; i < nb * HW; i += 256) {
    const int s = i / HW, p = i - s * HW;
    const int h = p / W, w = p - h * W;
    const int ph = h / ps, pw = w / ps;
    const bool on = ph < nph && pw < npw && ((sel[s] >> (ph * npw + pw)) & 1ull);
    out[(size_t)(b0 + s) * HW + p] = on ? 1.f : 0.f;
  }
}

// (the bodies are device functions over the counter index i — four values each — so that a fused launch draws exactly what the
// separate launches draw)
__device__ __forceinline__ void randint_quad(int64_t i, int64_t* __restrict__ out, int64_t n, int32_t lo, int32_t hi,
                                             const int64_t* __restrict__ exclude, uint64_t seed, uint64_t offset,
                                             float* __restrict__ onehot = nullptr) {
  const uint32_t span = (uint32_t)(hi - lo);
  {
    const U4 r = draw(seed, offset, (uint64_t)i);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i * 4 + e;
      if (j >= n) break;
      if (exclude) {  // the reference's rule (house-sales trainer.py:248-249): k uniform over the WHOLE span, a collision with
                      // exclude[j] maps to the next class (cyclically) — P(exclude+1) = 2/span, 1/span for the others, never exclude
        const uint32_t k = (uint32_t)(((uint64_t)v[e] * (uint64_t)span) >> 32);
        const int64_t ex = exclude[j] - lo;
        const uint32_t kk = (int64_t)k == ex ? (k + 1u) % span : k;
        out[j] = lo + (int64_t)kk;
        if (onehot) for (uint32_t q = 0; q < span; ++q) onehot[j * span + q] = q == kk ? 1.f : 0.f;      // F.one_hot(target).float() (trainer.py:250)
      } else {
        out[j] = lo + (int64_t)(((uint64_t)v[e] * (uint64_t)span) >> 32);
      }
    }
  }
}
__global__ void __launch_bounds__(256) randint_kernel(int64_t* __restrict__ out, int64_t n, int32_t lo, int32_t hi,
                                                      const int64_t* __restrict__ exclude, uint64_t seed, uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 3) / 4; i += (int64_t)gridDim.x * 256)
    randint_quad(i, out, n, lo, hi, exclude, seed, offset);
}

__global__ void __launch_bounds__(256) randn_kernel(float* __restrict__ out, int64_t n, float mean, float std, uint64_t seed,
                                                    uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 3) / 4; i += (int64_t)gridDim.x * 256) {
    const U4 r = draw(seed, offset, (uint64_t)i);
    // Box-Muller on two uniform pairs
    const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
    sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
    const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i * 4 + e;
      if (j < n) out[j] = fmaf(v[e], std, mean);
    }
  }
}

// 23-bit variant whose largest value, 1 - 2^-24, is still below 1 in fp32 (u01's top value rounds to 1.0f, harmless under a
// single log but an infinity under -log(-log(u)))
__device__ __forceinline__ float u01_open(uint32_t v) { return ((float)(v >> 9) + 0.5f) * (1.0f / 8388608.0f); }

__device__ __forceinline__ void gumbel_quad(int64_t i, float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  const U4 r = draw(seed, offset, (uint64_t)i);
  const uint32_t v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t j = i * 4 + e;
    if (j < n) out[j] = -logf(-logf(u01_open(v[e])));
  }
}
__global__ void __launch_bounds__(256) gumbel_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 3) / 4; i += (int64_t)gridDim.x * 256) gumbel_quad(i, out, n, seed, offset);
}

__device__ __forceinline__ void feature_mask_quad(int64_t i, float* __restrict__ out, int64_t n, int D, const int* __restrict__ zero_cols, int nz,
                                                  uint64_t seed, uint64_t offset) {
  {
    const U4 r = draw(seed, offset, (uint64_t)i);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i * 4 + e;
      if (j >= n) break;
      const int col = (int)(j % D);
      bool on = (v[e] >> 31) != 0u;
      for (int z = 0; z < nz; ++z) on = on && zero_cols[z] != col;
      out[j] = on ? 1.f : 0.f;
    }
  }
}
__global__ void __launch_bounds__(256) feature_mask_kernel(float* __restrict__ out, int B, int D, const int* __restrict__ zero_cols, int nz,
                                                           uint64_t seed, uint64_t offset) {
  const int64_t n = (int64_t)B * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 3) / 4; i += (int64_t)gridDim.x * 256)
    feature_mask_quad(i, out, n, D, zero_cols, nz, seed, offset);
}
// the three per-iteration draws of the tabular trainer (trainer.py:248-255, generator.py:90) in one launch
// ctr != nullptr: the Philox offsets come from a DEVICE counter — ctr[0] is the running offset (the three draws take consecutive
// ranges from it, as DeviceRNG's host-side bookkeeping hands them out), ctr[1] a ticket; the block that takes the last ticket
// advances ctr[0] (every block has read it by then).  A launch captured in a HIP graph then draws fresh numbers on every replay.
__global__ void __launch_bounds__(256) house_draws_kernel(int64_t* __restrict__ target, int B, int32_t lo, int32_t hi, const int64_t* __restrict__ y,
                                                          uint64_t off_t, float* __restrict__ mask, int D, const int* __restrict__ zero_cols, int nz,
                                                          uint64_t off_m, float* __restrict__ noise, int64_t n_noise, uint64_t off_n, uint64_t seed,
                                                          float* __restrict__ onehot_t, float* __restrict__ onehot_y, unsigned long long* ctr) {
  const int64_t nm = (int64_t)B * D;
  unsigned long long base = 0;
  if (ctr) {                                                 // kernel-uniform
    base = ctr[0];
    off_t = base; off_m = off_t + (uint64_t)((B + 3) / 4); off_n = off_m + (uint64_t)((nm + 3) / 4);
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n_noise + 3) / 4; i += (int64_t)gridDim.x * 256) gumbel_quad(i, noise, n_noise, seed, off_n);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (nm + 3) / 4; i += (int64_t)gridDim.x * 256)
    feature_mask_quad(i, mask, nm, D, zero_cols, nz, seed, off_m);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ((int64_t)B + 3) / 4; i += (int64_t)gridDim.x * 256)
    randint_quad(i, target, B, lo, hi, y, seed, off_t, onehot_t);
  if (onehot_y) {                                            // F.one_hot(y).float() (trainer.py:290) rides along: y is an input
    const int nc = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * nc; i += (int64_t)gridDim.x * 256) {
      const int64_t b = i / nc; const int q = (int)(i - b * nc);
      onehot_y[i] = y[b] - lo == q ? 1.f : 0.f;
    }
  }
  if (ctr) {
    __syncthreads();                                         // every thread of this block has read the counter
    if (threadIdx.x == 0 && atomicAdd(reinterpret_cast<int*>(ctr + 1), 1) == (int)gridDim.x - 1) {
      ctr[0] = off_n + (uint64_t)((n_noise + 3) / 4);
      *reinterpret_cast<int*>(ctr + 1) = 0;
    }
  }
}

// house_draws_kernel (counter form) + the BATCH itself: the training set is resident in HBM, the epoch's permutation too; the launch
// takes rows perm[cur .. cur+B) (cur = ctr[2], advanced by B by the block that takes the last ticket) into the static x / y buffers
// of the captured step, so a replay needs no host-side copy at all (DataLoader(shuffle=True, drop_last=True), trainer.py:198,
// without per-batch collation and PCIe traffic).  y is gathered by the thread that draws that row's target class (the target rule
// reads it), its one-hot row written there too; src_out[b] = the source row (the diagnostics gather the frozen classifier's logits
// of the original rows with it).
__global__ void __launch_bounds__(256) house_batch_draws_kernel(int64_t* __restrict__ target, int B, int32_t lo, int32_t hi,
                                                                const float* __restrict__ X, const int64_t* __restrict__ Y,
                                                                const int64_t* __restrict__ perm, int64_t n_perm, int64_t n_rows,
                                                                float* __restrict__ x_out, int64_t* __restrict__ y_out, int64_t* __restrict__ src_out,
                                                                float* __restrict__ mask, int D, const int* __restrict__ zero_cols, int nz,
                                                                float* __restrict__ noise, int64_t n_noise, uint64_t seed,
                                                                float* __restrict__ onehot_t, float* __restrict__ onehot_y, unsigned long long* ctr) {
  const int64_t nm = (int64_t)B * D;
  const unsigned long long base = ctr[0];
  const int64_t cur = (int64_t)ctr[2];
  const uint64_t off_t = base, off_m = off_t + (uint64_t)((B + 3) / 4), off_n = off_m + (uint64_t)((nm + 3) / 4);
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
  auto src_row = [&](int64_t b) -> int64_t {
    int64_t p = cur + b;
    p = p < n_perm ? p : n_perm - 1;                         // host guarantees cur + B <= n_perm; never read past the buffers
    int64_t r = perm[p];
    return r < 0 ? 0 : (r < n_rows ? r : n_rows - 1);
  };
  for (int64_t i = tid; i < (n_noise + 3) / 4; i += nthr) gumbel_quad(i, noise, n_noise, seed, off_n);
  for (int64_t i = tid; i < (nm + 3) / 4; i += nthr) feature_mask_quad(i, mask, nm, D, zero_cols, nz, seed, off_m);
  const int nc = hi - lo;
  for (int64_t i = tid; i < ((int64_t)B + 3) / 4; i += nthr) {
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i * 4 + e;
      if (j >= B) break;
      const int64_t r = src_row(j);
      const int64_t yv = Y[r];
      y_out[j] = yv;
      if (src_out) src_out[j] = r;
      if (onehot_y) for (int q = 0; q < nc; ++q) onehot_y[j * nc + q] = yv - lo == q ? 1.f : 0.f;
    }
    randint_quad(i, target, B, lo, hi, y_out, seed, off_t, onehot_t);      // reads the y values this thread has just written
  }
  for (int64_t i = tid; i < nm; i += nthr) {
    const int64_t b = i / D; const int c = (int)(i - b * D);
    x_out[i] = X[src_row(b) * D + c];
  }
  __syncthreads();                                           // every thread of this block has read the counters
  if (threadIdx.x == 0 && atomicAdd(reinterpret_cast<int*>(ctr + 1), 1) == (int)gridDim.x - 1) {
    ctr[0] = off_n + (uint64_t)((n_noise + 3) / 4);
    ctr[2] = (unsigned long long)(cur + B);
    *reinterpret_cast<int*>(ctr + 1) = 0;
  }
}

__global__ void __launch_bounds__(256) uniform_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 3) / 4; i += (int64_t)gridDim.x * 256) {
    const U4 r = draw(seed, offset, (uint64_t)i);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i * 4 + e;
      if (j < n) out[j] = (float)(v[e] >> 8) * (1.0f / 16777216.0f);     // [0, 1), 24 bits, as torch.rand
    }
  }
}

__global__ void __launch_bounds__(256) bernoulli_kernel(float* __restrict__ out, int64_t n, float keep, uint64_t seed, uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 3) / 4; i += (int64_t)gridDim.x * 256) {
    const U4 r = draw(seed, offset, (uint64_t)i);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i * 4 + e;
      if (j < n) out[j] = (float)(v[e] >> 8) * (1.0f / 16777216.0f) < keep ? 1.f : 0.f;
    }
  }
}

unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_patch_mask(float* out, int32_t B, int32_t H, int32_t W, int32_t patch_size, int32_t num_selected,
                              uint64_t seed, uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && B > 0 && H > 0 && W > 0 && patch_size > 0 && num_selected >= 0, "pcg_patch_mask: bad arguments");
  const int nph = H / patch_size, npw = W / patch_size;
  PCG_REQUIRE(nph * npw >= 1 && nph * npw <= 64, "pcg_patch_mask: patch grid %dx%d must have 1..64 patches", nph, npw);
  hipLaunchKernelGGL(patch_mask_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, B, H, W, patch_size, nph, npw,
                     num_selected, seed, offset);
  return launch_status("patch_mask_kernel");
}

extern "C" int pcg_randint(int64_t* out, int64_t n, int32_t low, int32_t high, const int64_t* exclude, uint64_t seed,
                           uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && n > 0 && high > low && (!exclude || high - low > 1), "pcg_randint: bad arguments");
  hipLaunchKernelGGL(randint_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, low, high, exclude, seed,
                     offset);
  return launch_status("randint_kernel");
}

extern "C" int pcg_randn(float* out, int64_t n, float mean, float std, uint64_t seed, uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && n > 0, "pcg_randn: bad arguments");
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, mean, std, seed, offset);
  return launch_status("randn_kernel");
}

extern "C" int pcg_rand_gumbel(float* out, int64_t n, uint64_t seed, uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && n > 0, "pcg_rand_gumbel: bad arguments");
  hipLaunchKernelGGL(gumbel_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset);
  return launch_status("gumbel_kernel");
}

extern "C" int pcg_feature_mask(float* out, int32_t B, int32_t D, const int32_t* zero_cols, int32_t n_zero_cols, uint64_t seed,
                                uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && B > 0 && D > 0 && n_zero_cols >= 0 && (zero_cols || n_zero_cols == 0), "pcg_feature_mask: bad arguments");
  hipLaunchKernelGGL(feature_mask_kernel, dim3(grid_for(((int64_t)B * D + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, B, D, zero_cols,
                     n_zero_cols, seed, offset);
  return launch_status("feature_mask_kernel");
}

namespace {
int house_draws_launch(int64_t* target_y, int32_t B, int32_t num_classes, const int64_t* y, uint64_t offset_target, float* mask, int32_t D,
                       const int32_t* zero_cols, int32_t n_zero_cols, uint64_t offset_mask, float* noise, int32_t T, uint64_t offset_noise,
                       uint64_t seed, float* onehot_target, float* onehot_y, unsigned long long* counter, pcg_stream_t stream) {
  PCG_REQUIRE(target_y && y && mask && noise && B > 0 && num_classes > 1 && D > 0 && T > 0 && n_zero_cols >= 0 && (zero_cols || n_zero_cols == 0),
              "pcg_house_draws: bad arguments");
  int64_t quads = std::max(((int64_t)B * T + 3) / 4, ((int64_t)B * D + 3) / 4);
  unsigned blocks = grid_for(quads);
  if (counter && blocks > 256u) blocks = 256u;               // one same-address ticket per block (see adam_launch)
  hipLaunchKernelGGL(house_draws_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, target_y, B, 0, num_classes, y, offset_target, mask, D,
                     zero_cols, n_zero_cols, offset_mask, noise, (int64_t)B * T, offset_noise, seed, onehot_target, onehot_y, counter);
  return launch_status("house_draws_kernel");
}
}  // namespace

extern "C" int pcg_house_draws(int64_t* target_y, int32_t B, int32_t num_classes, const int64_t* y, uint64_t offset_target, float* mask, int32_t D,
                               const int32_t* zero_cols, int32_t n_zero_cols, uint64_t offset_mask, float* noise, int32_t T, uint64_t offset_noise,
                               uint64_t seed, float* onehot_target, float* onehot_y, pcg_stream_t stream) {
  return house_draws_launch(target_y, B, num_classes, y, offset_target, mask, D, zero_cols, n_zero_cols, offset_mask, noise, T, offset_noise, seed,
                            onehot_target, onehot_y, nullptr, stream);
}

extern "C" int pcg_house_draws_counter(int64_t* target_y, int32_t B, int32_t num_classes, const int64_t* y, float* mask, int32_t D,
                                       const int32_t* zero_cols, int32_t n_zero_cols, float* noise, int32_t T, uint64_t seed, float* onehot_target,
                                       float* onehot_y, uint64_t* counter, pcg_stream_t stream) {
  PCG_REQUIRE(counter, "pcg_house_draws_counter: null counter");
  return house_draws_launch(target_y, B, num_classes, y, 0, mask, D, zero_cols, n_zero_cols, 0, noise, T, 0, seed, onehot_target, onehot_y,
                            reinterpret_cast<unsigned long long*>(counter), stream);
}

extern "C" int pcg_rand_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && n > 0, "pcg_rand_uniform: bad arguments");
  hipLaunchKernelGGL(uniform_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset);
  return launch_status("uniform_kernel");
}

extern "C" int pcg_rand_bernoulli(float* out, int64_t n, float keep_prob, uint64_t seed, uint64_t offset, pcg_stream_t stream) {
  PCG_REQUIRE(out && n > 0 && keep_prob >= 0.f && keep_prob <= 1.f, "pcg_rand_bernoulli: bad arguments");
  hipLaunchKernelGGL(bernoulli_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, keep_prob, seed, offset);
  return launch_status("bernoulli_kernel");
}

// pcg_house_draws_counter + the batch gather (see house_batch_draws_kernel).  counter: uint64[4] on the device =
// [Philox offset, ticket, row cursor into perm, unused]; the launch advances offset and cursor itself.
extern "C" int pcg_house_batch_draws_counter(int64_t* target_y, int32_t B, int32_t num_classes, const float* X, const int64_t* Y,
                                             const int64_t* perm, int64_t n_perm, int64_t n_rows, float* x_out, int64_t* y_out,
                                             int64_t* src_out, float* mask, int32_t D, const int32_t* zero_cols, int32_t n_zero_cols,
                                             float* noise, int32_t T, uint64_t seed, float* onehot_target, float* onehot_y, uint64_t* counter,
                                             pcg_stream_t stream) {
  PCG_REQUIRE(target_y && X && Y && perm && x_out && y_out && mask && noise && counter && B > 0 && num_classes > 1 && D > 0 && T > 0 &&
                  n_perm >= B && n_rows > 0 && n_zero_cols >= 0 && (zero_cols || n_zero_cols == 0),
              "pcg_house_batch_draws_counter: bad arguments (B %d, permutation of %lld entries over %lld rows)", B, (long long)n_perm, (long long)n_rows);
  int64_t quads = std::max(((int64_t)B * T + 3) / 4, ((int64_t)B * D + 3) / 4);
  unsigned blocks = grid_for(quads);
  if (blocks > 256u) blocks = 256u;                           // one same-address ticket per block
  hipLaunchKernelGGL(house_batch_draws_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, target_y, B, 0, num_classes, X, Y, perm, n_perm, n_rows,
                     x_out, y_out, src_out, mask, D, zero_cols, n_zero_cols, noise, (int64_t)B * T, seed, onehot_target, onehot_y,
                     reinterpret_cast<unsigned long long*>(counter));
  return launch_status("house_batch_draws_kernel");
}
