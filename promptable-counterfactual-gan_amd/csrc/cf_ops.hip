// cf_ops.hip — the non-convolution pieces of the CounteRGAN step (conditional_counteRGAN/mnist): label-embedding
// lookup + channel concat (bit-exact row copies), residual composition, clamp, L1 penalties, global average pool,
// softmax cross-entropy.  All are small HBM-bound streaming kernels; reductions are fixed-order (reproducible).
// Reference call sites are cited next to each prototype in include/pcgan_hip.h.
#include "pcg_common.h"

namespace pcg {
namespace {

unsigned ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// out[b][p][0] = x[b][p] ; out[b][p][1] = table[idx[b]][p] ; out[b][p][2] = mask[b][p] (C == 3)
__global__ void __launch_bounds__(256) embed_concat_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                               const float* __restrict__ table, const float* __restrict__ mask,
                                                               float* __restrict__ out, int B, int HW, int C, int K) {
  const size_t n = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / HW), p = (int)(i - (size_t)b * HW);
    int64_t k = idx[b];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);  // host checks the range; never read outside the table
    float* o = out + i * C;
    o[0] = x[i];
    o[1] = table[(size_t)k * HW + p];
    if (C > 2) o[2] = mask[i];
  }
}

// dtable[k][p] (+)= sum over b with idx[b]==k of dinp[b][p][1]: thread (i = (k,p), g) adds the matching rows of batch chunk g in
// ascending order, the 16 chunk sums are added in chunk order through LDS (fixed association: reproducible).  One thread per (k,p)
// walking the whole batch alone was a 1024-step dependent chain in 31 blocks (108 us at batch 1024).
__global__ void __launch_bounds__(256) embed_table_grad_kernel(const float* __restrict__ dinp, const int64_t* __restrict__ idx,
                                                               float* __restrict__ dtable, int B, int HW, int C, int K, int accumulate,
                                                               int ch = 1) {
  __shared__ float red[16][16];
  const int il = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il, total = K * HW;
  float s = 0.f;
  if (i < total) {
    const int k = i / HW, p = i - k * HW;
    const int chunk = (B + 15) / 16, b0 = g * chunk, b1 = b0 + chunk < B ? b0 + chunk : B;
    // every row is loaded (the gradient is a few MB: cache-resident) and a non-matching one adds 0 — eight independent loads in flight
    // instead of a chain of branches; same sum (s + 0 == s), same order (r04: 25.7 -> 7.3 us at batch 1024)
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
      bool m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = dinp[((size_t)(b + j) * HW + p) * C + ch]; m[j] = idx[b + j] == (int64_t)k; }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += m[j] ? v[j] : 0.f;
    }
    for (; b < b1; ++b)
      if (idx[b] == (int64_t)k) s += dinp[((size_t)b * HW + p) * C + ch];
  }
  red[g][il] = s;
  __syncthreads();
  if (g == 0 && i < total) {
    float t = accumulate ? dtable[i] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][il];
    dtable[i] = t;
  }
}

// dtable[k][p] (+)= sum over b with idx[b]==k (ascending b) of dinp[b][p][1] ; dx[b][p] = dinp[b][p][0] (optional)
__global__ void __launch_bounds__(256) embed_concat_bwd_kernel(const float* __restrict__ dinp, const int64_t* __restrict__ idx,
                                                               float* __restrict__ dtable, float* __restrict__ dx, int B, int HW,
                                                               int C, int K, int accumulate) {
  const int total = K * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    if (!dtable) break;
    const int k = i / HW, p = i - k * HW;
    float s = accumulate ? dtable[i] : 0.f;
    for (int b = 0; b < B; ++b)
      if (idx[b] == (int64_t)k) s += dinp[((size_t)b * HW + p) * C + 1];
    dtable[i] = s;
  }
  if (dx) {
    const size_t n = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = dinp[i * C];
  }
}

__global__ void __launch_bounds__(256) axpby_kernel(float* __restrict__ out, float a, const float* __restrict__ x, float b,
                                                    const float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// raw = s*c ; masked = raw*mask
__global__ void __launch_bounds__(256) scale_mask_fwd_kernel(const float* __restrict__ c, const float* __restrict__ mask, float s,
                                                             float* __restrict__ raw, float* __restrict__ masked, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float r = s * c[i];
    raw[i] = r;
    masked[i] = mask ? r * mask[i] : r;
  }
}
// dc = s*(d_raw + d_masked*mask)
__global__ void __launch_bounds__(256) scale_mask_bwd_kernel(const float* __restrict__ d_raw, const float* __restrict__ d_masked,
                                                             const float* __restrict__ mask, float s, float* __restrict__ dc, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float g = d_raw ? d_raw[i] : 0.f;
    if (d_masked) g += d_masked[i] * (mask ? mask[i] : 1.f);
    dc[i] = s * g;
  }
}

// y = clamp(x + r, lo, hi) ; dr = dy * [lo <= x + r <= hi]   ([torch] clamp passes the gradient on the closed interval)
__global__ void __launch_bounds__(256) clamp_add_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, float lo,
                                                            float hi, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = fminf(fmaxf(x[i] + r[i], lo), hi);
}
__global__ void __launch_bounds__(256) clamp_add_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ r, float lo, float hi,
                                                            float* __restrict__ dr, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i] + r[i];
    dr[i] = (v >= lo && v <= hi) ? dy[i] : 0.f;
  }
}

// mean |a * w| (w = 1 - m if one_minus, m otherwise, 1 if null): per-block partials, then a fixed-order finish
constexpr int AM_BLOCKS = 256;
__device__ __forceinline__ float am_weight(const float* m, size_t i, int one_minus) {
  return m ? (one_minus ? 1.f - m[i] : m[i]) : 1.f;
}
__global__ void __launch_bounds__(256) abs_mean_partial_kernel(const float* __restrict__ a, const float* __restrict__ m,
                                                               int one_minus, size_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += fabsf(a[i] * am_weight(m, i, one_minus));
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// fixed-order tree over the block partials in fp64 (one thread walking all 256 partials was a 10 us dependent chain)
__global__ void __launch_bounds__(256) abs_mean_finish_kernel(const float* __restrict__ partial, int nparts, double inv_n, float* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}
// small tensors (the tabular step: 4096 x 17): the whole reduction in ONE block, one launch instead of two
__global__ void __launch_bounds__(1024) abs_mean_small_kernel(const float* __restrict__ a, const float* __restrict__ m, int one_minus,
                                                              size_t n, double inv_n, float* out) {
  __shared__ double red[1024];
  float acc = 0.f;
  size_t i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {       // 8 independent loads in flight (one dependent load per element was 23 us for 70 k)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a[i + j * 1024] * am_weight(m, i + j * 1024, one_minus);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += fabsf(v[j]);
  }
  for (; i < n; i += 1024) acc += fabsf(a[i] * am_weight(m, i, one_minus));
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}
// da (+)= g * sign(a*w) * w / n      ([torch] abs'(0) = 0)
__global__ void __launch_bounds__(256) abs_mean_bwd_kernel(const float* __restrict__ a, const float* __restrict__ m, int one_minus,
                                                           size_t n, const float* __restrict__ gout, float scale,
                                                           float* __restrict__ da, int accumulate) {
  const float g = (gout ? gout[0] : 1.f) * scale / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float w = am_weight(m, i, one_minus), v = a[i] * w;
    const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
    da[i] = (accumulate ? da[i] : 0.f) + g * sg * w;
  }
}

// global average pool over HW: y[b][c] = mean_p x[b][p][c]
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
  const int total = B * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += x[((size_t)b * HW + p) * C + c];
    y[i] = s / (float)HW;
  }
}
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int HW, int C) {
  const size_t n = (size_t)B * HW * C;
  const float inv = 1.f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t b = i / ((size_t)HW * C);
    dx[i] = dy[b * C + c] * inv;
  }
}

// softmax cross-entropy, reduction mean: one thread per row (K is the number of classes: 10), ONE block so that the loss is summed
// in a fixed order; 1024 threads for large batches (batch 4096 on 256 threads was 16 rows of expf/logf per thread: 24 us)
__global__ void __launch_bounds__(1024) cross_entropy_kernel(const float* __restrict__ z, const int64_t* __restrict__ target, int B,
                                                             int K, float grad_scale, const float* __restrict__ gout,
                                                             float* loss, float* __restrict__ dz) {
  __shared__ float red[1024];
  const int nt = blockDim.x;    // power of two
  float acc = 0.f;
  const float g = grad_scale * (gout ? gout[0] : 1.f) / (float)B;
  int bstart = threadIdx.x;
  if (K == 4 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (!dz || (reinterpret_cast<uintptr_t>(dz) & 15) == 0)) {
    // four classes (the tabular classifier): a row is one float4.  A thread's rows are requested four at a time before any of them
    // is used (one dependent load chain per row made batch 4096 a 15 us launch); the rows are still ADDED in row order.
    for (; bstart + 3 * nt < B; bstart += 4 * nt) {
      float4 rv[4]; int64_t tv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { rv[j] = *reinterpret_cast<const float4*>(z + (size_t)(bstart + j * nt) * 4); tv[j] = target[bstart + j * nt]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
        float mx = r[0];
        for (int k = 1; k < 4; ++k) mx = fmaxf(mx, r[k]);
        float se = 0.f;
        for (int k = 0; k < 4; ++k) se += expf(r[k] - mx);
        const float lse = mx + logf(se);
        const int t = tv[j] < 0 ? 0 : (tv[j] >= 4 ? 3 : (int)tv[j]);
        acc += lse - r[t];
        if (dz) {
          float o[4];
          for (int k = 0; k < 4; ++k) o[k] = g * (expf(r[k] - lse) - (k == t ? 1.f : 0.f));
          *reinterpret_cast<float4*>(dz + (size_t)(bstart + j * nt) * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
  for (int b = bstart; b < B; b += nt) {
    const float* r = z + (size_t)b * K;
    float mx = r[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, r[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(r[k] - mx);
    const float lse = mx + logf(se);
    int64_t t = target[b];
    t = t < 0 ? 0 : (t >= K ? K - 1 : t);
    acc += lse - r[t];
    if (dz)
      for (int k = 0; k < K; ++k) dz[(size_t)b * K + k] = g * (expf(r[k] - lse) - (k == (int)t ? 1.f : 0.f));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = nt >> 1; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) loss[0] = red[0] / (float)B;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_embed_concat_fwd(const float* x, const int64_t* idx, const float* table, const float* mask, float* out,
                                    int32_t B, int32_t HW, int32_t C, int32_t K, pcg_stream_t stream) {
  PCG_REQUIRE(x && idx && table && out && B > 0 && HW > 0 && K > 0 && (C == 2 || (C == 3 && mask)), "pcg_embed_concat_fwd: bad arguments");
  hipLaunchKernelGGL(embed_concat_fwd_kernel, dim3(ew_blocks((size_t)B * HW)), dim3(256), 0, (hipStream_t)stream, x, idx, table, mask,
                     out, B, HW, C, K);
  return launch_status("embed_concat_fwd_kernel");
}

extern "C" int pcg_embed_concat_bwd(const float* dinp, const int64_t* idx, float* dtable, float* dx, int32_t B, int32_t HW,
                                    int32_t C, int32_t K, int accumulate, pcg_stream_t stream) {
  PCG_REQUIRE(dinp && idx && (dtable || dx) && B > 0 && HW > 0 && K > 0 && C >= 2, "pcg_embed_concat_bwd: bad arguments");
  if (dtable && B >= 64) {     // table gradient with the batch split over 16 thread groups; the optional dx copy as its own launch
    hipLaunchKernelGGL(embed_table_grad_kernel, dim3((unsigned)(((size_t)K * HW + 15) / 16)), dim3(256), 0, (hipStream_t)stream, dinp, idx,
                       dtable, B, HW, C, K, accumulate);
    if (int e = launch_status("embed_table_grad_kernel")) return e;
    if (!dx) return PCG_OK;
    dtable = nullptr;
  }
  const size_t work = dx ? (size_t)B * HW : (size_t)K * HW;
  hipLaunchKernelGGL(embed_concat_bwd_kernel, dim3(ew_blocks(work)), dim3(256), 0, (hipStream_t)stream, dinp, idx, dtable, dx, B, HW,
                     C, K, accumulate);
  return launch_status("embed_concat_bwd_kernel");
}

namespace pcg { namespace {
__global__ void __launch_bounds__(256) gather_channel_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int C, int ch) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) dst[i] = src[(size_t)i * C + ch];
}
} }
// dst[r] = src[r][ch] for a [rows][C] tensor: the OHWI weight of ONE input channel of a convolution (conv_in's label-map channel: the
// only one of its three input channels whose gradient anybody reads, models/generator.py:73-74)
extern "C" int pcg_gather_channel(const float* src, float* dst, int32_t rows, int32_t C, int32_t ch, pcg_stream_t stream) {
  PCG_REQUIRE(src && dst && rows > 0 && C > 0 && ch >= 0 && ch < C, "pcg_gather_channel: bad arguments");
  hipLaunchKernelGGL(gather_channel_kernel, dim3((unsigned)((rows + 255) / 256 > 1024 ? 1024 : (rows + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, rows, C, ch);
  return launch_status("gather_channel_kernel");
}
// The embedding-table gradient alone, from channel `ch` of a [B][HW][C] gradient (pcg_embed_concat_bwd reads channel 1 of C >= 2):
// dtable[k][p] (+)= sum over b with idx[b] == k, ascending b, of dinp[b][p][ch]
extern "C" int pcg_embed_table_grad(const float* dinp, const int64_t* idx, float* dtable, int32_t B, int32_t HW, int32_t C, int32_t ch,
                                    int32_t K, int accumulate, pcg_stream_t stream) {
  PCG_REQUIRE(dinp && idx && dtable && B > 0 && HW > 0 && K > 0 && C >= 1 && ch >= 0 && ch < C, "pcg_embed_table_grad: bad arguments");
  hipLaunchKernelGGL(embed_table_grad_kernel, dim3((unsigned)(((size_t)K * HW + 15) / 16)), dim3(256), 0, (hipStream_t)stream, dinp, idx,
                     dtable, B, HW, C, K, accumulate, ch);
  return launch_status("embed_table_grad_kernel");
}

extern "C" int pcg_axpby(float* out, float a, const float* x, float b, const float* y, int64_t n, pcg_stream_t stream) {
  PCG_REQUIRE(out && x && n > 0, "pcg_axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, out, a, x, b, y, (size_t)n);
  return launch_status("axpby_kernel");
}

extern "C" int pcg_scale_mask_fwd(const float* c, const float* mask, float scale, float* raw, float* masked, int64_t n,
                                  pcg_stream_t stream) {
  PCG_REQUIRE(c && raw && masked && n > 0, "pcg_scale_mask_fwd: bad arguments");
  hipLaunchKernelGGL(scale_mask_fwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, c, mask, scale, raw, masked,
                     (size_t)n);
  return launch_status("scale_mask_fwd_kernel");
}

extern "C" int pcg_scale_mask_bwd(const float* d_raw, const float* d_masked, const float* mask, float scale, float* dc, int64_t n,
                                  pcg_stream_t stream) {
  PCG_REQUIRE((d_raw || d_masked) && dc && n > 0, "pcg_scale_mask_bwd: bad arguments");
  hipLaunchKernelGGL(scale_mask_bwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, d_raw, d_masked, mask,
                     scale, dc, (size_t)n);
  return launch_status("scale_mask_bwd_kernel");
}

extern "C" int pcg_clamp_add_fwd(const float* x, const float* r, float lo, float hi, float* y, int64_t n, pcg_stream_t stream) {
  PCG_REQUIRE(x && r && y && n > 0, "pcg_clamp_add_fwd: bad arguments");
  hipLaunchKernelGGL(clamp_add_fwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, r, lo, hi, y, (size_t)n);
  return launch_status("clamp_add_fwd_kernel");
}

extern "C" int pcg_clamp_add_bwd(const float* dy, const float* x, const float* r, float lo, float hi, float* dr, int64_t n,
                                 pcg_stream_t stream) {
  PCG_REQUIRE(dy && x && r && dr && n > 0, "pcg_clamp_add_bwd: bad arguments");
  hipLaunchKernelGGL(clamp_add_bwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, dy, x, r, lo, hi, dr,
                     (size_t)n);
  return launch_status("clamp_add_bwd_kernel");
}

extern "C" size_t pcg_abs_mean_workspace_bytes(void) { return AM_BLOCKS * sizeof(float); }

extern "C" int pcg_abs_mean_fwd(const float* a, const float* m, int one_minus_m, int64_t n, float* out, void* workspace,
                                size_t workspace_bytes, pcg_stream_t stream) {
  PCG_REQUIRE(a && out && n > 0, "pcg_abs_mean_fwd: bad arguments");
  if (!workspace || workspace_bytes < pcg_abs_mean_workspace_bytes()) {
    set_error("pcg_abs_mean_fwd: workspace too small");
    return PCG_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;
  if (n <= 16 * 1024) {   // one block is enough up to ~16 elements per thread; beyond, 256 blocks + the parallel finish are faster
    hipLaunchKernelGGL(abs_mean_small_kernel, dim3(1), dim3(1024), 0, s, a, m, one_minus_m, (size_t)n, 1.0 / (double)n, out);
    return launch_status("abs_mean_small_kernel");
  }
  hipLaunchKernelGGL(abs_mean_partial_kernel, dim3(AM_BLOCKS), dim3(256), 0, s, a, m, one_minus_m, (size_t)n, partial);
  if (int e = launch_status("abs_mean_partial_kernel")) return e;
  hipLaunchKernelGGL(abs_mean_finish_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, AM_BLOCKS, 1.0 / (double)n, out);
  return launch_status("abs_mean_finish_kernel");
}

extern "C" int pcg_abs_mean_bwd(const float* a, const float* m, int one_minus_m, int64_t n, const float* grad_out_dev,
                                float grad_scale, float* da, int accumulate, pcg_stream_t stream) {
  PCG_REQUIRE(a && da && n > 0, "pcg_abs_mean_bwd: bad arguments");
  hipLaunchKernelGGL(abs_mean_bwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, a, m, one_minus_m, (size_t)n,
                     grad_out_dev, grad_scale, da, accumulate);
  return launch_status("abs_mean_bwd_kernel");
}

extern "C" int pcg_avgpool_fwd(const float* x, float* y, int32_t B, int32_t HW, int32_t C, pcg_stream_t stream) {
  PCG_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, "pcg_avgpool_fwd: bad arguments");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ew_blocks((size_t)B * C)), dim3(256), 0, (hipStream_t)stream, x, y, B, HW, C);
  return launch_status("avgpool_fwd_kernel");
}

extern "C" int pcg_avgpool_bwd(const float* dy, float* dx, int32_t B, int32_t HW, int32_t C, pcg_stream_t stream) {
  PCG_REQUIRE(dy && dx && B > 0 && HW > 0 && C > 0, "pcg_avgpool_bwd: bad arguments");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks((size_t)B * HW * C)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, HW, C);
  return launch_status("avgpool_bwd_kernel");
}

extern "C" int pcg_cross_entropy_fwd_bwd(const float* logits, const int64_t* target, int32_t B, int32_t K, float grad_scale,
                                         const float* grad_out_dev, float* loss, float* dlogits, pcg_stream_t stream) {
  PCG_REQUIRE(logits && target && B > 0 && K > 0 && (loss || dlogits), "pcg_cross_entropy_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(B > 1024 ? 1024 : 256), 0, (hipStream_t)stream, logits, target, B, K, grad_scale,
                     grad_out_dev, loss, dlogits);
  return launch_status("cross_entropy_kernel");
}

namespace pcg { namespace {
// counterfactual evaluation reductions (mnist/eval_utils.py:61-66; house_sales_kc_usa/eval_utils.py:246-258):
//   flip = mean_b [argmax_k logits_cf[b] == target[b]]
//   gain = mean_b ( softmax(logits_cf[b])[target[b]] - q_b ),  q_b = softmax(logits_ref[b])[target[b]] if logits_ref
//                                                                   else softmax(logits_cf[b])[other[b]]
__global__ void __launch_bounds__(256) cf_metrics_kernel(const float* __restrict__ lcf, const float* __restrict__ lref,
                                                         const int64_t* __restrict__ target, const int64_t* __restrict__ other, int B, int K,
                                                         float* __restrict__ out) {
  __shared__ float r0[256], r1[256];
  float flips = 0.f, gain = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* row = lcf + (size_t)b * K;
    const int t = (int)target[b];
    float mx = row[0]; int arg = 0;
    for (int k = 1; k < K; ++k) if (row[k] > mx) { mx = row[k]; arg = k; }      // first maximum, as torch.argmax
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(row[k] - mx);
    const float pt = expf(row[t] - mx) / se;
    float q;
    if (lref) {
      const float* rr = lref + (size_t)b * K;
      float m2 = rr[0];
      for (int k = 1; k < K; ++k) m2 = fmaxf(m2, rr[k]);
      float s2 = 0.f;
      for (int k = 0; k < K; ++k) s2 += expf(rr[k] - m2);
      q = expf(rr[t] - m2) / s2;
    } else {
      q = expf(row[(int)other[b]] - mx) / se;
    }
    flips += arg == t ? 1.f : 0.f;
    gain += pt - q;
  }
  r0[threadIdx.x] = flips; r1[threadIdx.x] = gain;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = r0[0] / (float)B; out[1] = r1[0] / (float)B; }
}
} }

extern "C" int pcg_cf_metrics(const float* logits_cf, const float* logits_ref, const int64_t* target, const int64_t* other, int32_t B,
                              int32_t K, float* out, pcg_stream_t stream) {
  PCG_REQUIRE(logits_cf && target && out && (logits_ref || other) && B > 0 && K > 0, "pcg_cf_metrics: bad arguments");
  hipLaunchKernelGGL(cf_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits_cf, logits_ref, target, other, B, K, out);
  return launch_status("cf_metrics_kernel");
}

namespace pcg { namespace {
// nn.CrossEntropyLoss(weight=w), reduction mean:  loss = sum_b w[t_b] * (lse_b - z_b[t_b]) / sum_b w[t_b]
__global__ void __launch_bounds__(256) cross_entropy_weighted_kernel(const float* __restrict__ z, const int64_t* __restrict__ target,
                                                                     const float* __restrict__ w, int B, int K, float grad_scale,
                                                                     const float* __restrict__ gout, float* loss, float* __restrict__ dz) {
  __shared__ float red[256];
  __shared__ float s_wsum;
  float ws = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    int64_t t = target[b];
    t = t < 0 ? 0 : (t >= K ? K - 1 : t);
    ws += w[t];
  }
  red[threadIdx.x] = ws;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) s_wsum = red[0];
  __syncthreads();
  const float wsum = s_wsum;
  const float g = grad_scale * (gout ? gout[0] : 1.f) / wsum;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* r = z + (size_t)b * K;
    float mx = r[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, r[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(r[k] - mx);
    const float lse = mx + logf(se);
    int64_t t = target[b];
    t = t < 0 ? 0 : (t >= K ? K - 1 : t);
    const float wt = w[t];
    acc += wt * (lse - r[t]);
    if (dz)
      for (int k = 0; k < K; ++k) dz[(size_t)b * K + k] = g * wt * (expf(r[k] - lse) - (k == (int)t ? 1.f : 0.f));
  }
  __syncthreads();
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) loss[0] = red[0] / wsum;
}

// Dropout / Dropout2d: y[i] = x[i] * mask[m(i)] * scale, scale = 1/(1-p); mask has one entry per (row, channel) with
// `inner` positions sharing it (Dropout: inner = 1 -> m(i) = i; Dropout2d on NHWC [B][HW][C]: inner = HW)
__global__ void __launch_bounds__(256) dropout_apply_kernel(const float* __restrict__ x, const float* __restrict__ mask, size_t n, int inner,
                                                            int C, float scale, float* __restrict__ y) {
  const size_t per = (size_t)inner * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t m = inner == 1 ? i : (i / per) * C + (i % C);
    y[i] = x[i] * mask[m] * scale;
  }
}
} }

extern "C" int pcg_cross_entropy_weighted_fwd_bwd(const float* logits, const int64_t* target, const float* class_weight, int32_t B, int32_t K,
                                                  float grad_scale, const float* grad_out_dev, float* loss, float* dlogits,
                                                  pcg_stream_t stream) {
  PCG_REQUIRE(logits && target && class_weight && B > 0 && K > 0 && (loss || dlogits), "pcg_cross_entropy_weighted_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(cross_entropy_weighted_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, class_weight, B, K, grad_scale,
                     grad_out_dev, loss, dlogits);
  return launch_status("cross_entropy_weighted_kernel");
}

extern "C" int pcg_dropout_apply(const float* x, const float* mask, int64_t n, int32_t inner, int32_t C, float scale, float* y,
                                 pcg_stream_t stream) {
  PCG_REQUIRE(x && mask && y && n > 0 && inner > 0 && C > 0 && n % ((int64_t)inner * C) == 0, "pcg_dropout_apply: bad arguments");
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, mask, (size_t)n, inner, C, scale, y);
  return launch_status("dropout_apply_kernel");
}

namespace pcg { namespace {
constexpr int WS_MAX = 8;
struct WsTerms { const float* v[WS_MAX]; float w[WS_MAX]; float* g[WS_MAX]; int n; };
// total = sum_i w_i * v_i[0] (scalar losses); backward: g_i[0] = w_i * grad_out[0]
__global__ void weighted_sum_fwd_kernel(WsTerms t, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < t.n; ++i) s = fmaf(t.w[i], t.v[i][0], s);
    out[0] = s;
  }
}
__global__ void weighted_sum_bwd_kernel(WsTerms t, const float* __restrict__ grad_out) {
  if ((int)threadIdx.x < t.n && blockIdx.x == 0 && t.g[threadIdx.x]) t.g[threadIdx.x][0] = t.w[threadIdx.x] * (grad_out ? grad_out[0] : 1.f);
}
} }

extern "C" int pcg_weighted_sum_fwd(int32_t n, const float* const* terms, const float* weights, float* out, pcg_stream_t stream) {
  PCG_REQUIRE(n > 0 && n <= WS_MAX && terms && weights && out, "pcg_weighted_sum_fwd: bad arguments (at most %d terms)", WS_MAX);
  WsTerms t{};
  t.n = n;
  for (int i = 0; i < n; ++i) { PCG_REQUIRE(terms[i], "pcg_weighted_sum_fwd: null term %d", i); t.v[i] = terms[i]; t.w[i] = weights[i]; }
  hipLaunchKernelGGL(weighted_sum_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, out);
  return launch_status("weighted_sum_fwd_kernel");
}

extern "C" int pcg_weighted_sum_bwd(int32_t n, const float* weights, const float* grad_out_dev, float* const* grads, pcg_stream_t stream) {
  PCG_REQUIRE(n > 0 && n <= WS_MAX && weights && grads, "pcg_weighted_sum_bwd: bad arguments");
  WsTerms t{};
  t.n = n;
  for (int i = 0; i < n; ++i) { t.w[i] = weights[i]; t.g[i] = grads[i]; }
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, grad_out_dev);
  return launch_status("weighted_sum_bwd_kernel");
}
