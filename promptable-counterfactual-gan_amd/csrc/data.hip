// data.hip — input pipeline of the DCGAN configuration on the device:
//   transforms.Resize(64) -> ToTensor -> Normalize((0.5,), (0.5,))      dconv_gan/mnist/mnist_dcgan.py:42-46
// on a batch of 8-bit single-channel images.  The resize is Pillow's (torchvision resizes the PIL image): separable, 8-bit
// fixed point — per output coordinate a window [xmin, xmin+n) and integer coefficients (22 fractional bits) prepared on
// the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do; each pass rounds back to uint8
// (horizontal first, then vertical).  One block per image; the horizontally resized rows live in LDS.  Bit-exact
// against Pillow (tests/golden/mnist_resize.npz).
#include "pcg_common.h"

namespace pcg {
namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ void __launch_bounds__(256) resize_normalize_kernel(const uint8_t* __restrict__ src, int IH, int IW, int OH, int OW,
                                                               const int* __restrict__ xb, const int* __restrict__ xk, int xks,
                                                               const int* __restrict__ yb, const int* __restrict__ yk, int yks,
                                                               float mean, float stdv, float* __restrict__ dst) {
  extern __shared__ uint8_t tmp[];   // [IH][OW]
  const uint8_t* img = src + (size_t)blockIdx.x * IH * IW;
  for (int i = threadIdx.x; i < IH * OW; i += 256) {
    const int y = i / OW, x = i - y * OW;
    const int x0 = xb[2 * x], n = xb[2 * x + 1];
    int ss = 1 << (PRECISION_BITS - 1);
    for (int k = 0; k < n; ++k) ss += (int)img[y * IW + x0 + k] * xk[x * xks + k];
    tmp[i] = (uint8_t)clip8(ss);
  }
  __syncthreads();
  float* out = dst + (size_t)blockIdx.x * OH * OW;
  for (int i = threadIdx.x; i < OH * OW; i += 256) {
    const int y = i / OW, x = i - y * OW;
    const int y0 = yb[2 * y], n = yb[2 * y + 1];
    int ss = 1 << (PRECISION_BITS - 1);
    for (int k = 0; k < n; ++k) ss += (int)tmp[(y0 + k) * OW + x] * yk[y * yks + k];
    const float t = __fdiv_rn((float)clip8(ss), 255.f);            // ToTensor: uint8 -> float32 / 255
    out[i] = __fdiv_rn(t - mean, stdv);                            // Normalize: (t - mean) / std
  }
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_resize8_normalize(const uint8_t* src, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, const int32_t* x_bounds,
                                     const int32_t* x_coeffs, int32_t x_ksize, const int32_t* y_bounds, const int32_t* y_coeffs,
                                     int32_t y_ksize, float mean, float stdv, float* dst, pcg_stream_t stream) {
  PCG_REQUIRE(src && dst && x_bounds && x_coeffs && y_bounds && y_coeffs && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 &&
                  x_ksize > 0 && y_ksize > 0 && stdv != 0.f && (size_t)IH * OW <= 48 * 1024,
              "pcg_resize8_normalize: bad arguments (the horizontally resized image must fit 48 KB of LDS)");
  hipLaunchKernelGGL(resize_normalize_kernel, dim3(N), dim3(256), (size_t)IH * OW, (hipStream_t)stream, src, IH, IW, OH, OW, x_bounds,
                     x_coeffs, x_ksize, y_bounds, y_coeffs, y_ksize, mean, stdv, dst);
  return launch_status("resize_normalize_kernel");
}
