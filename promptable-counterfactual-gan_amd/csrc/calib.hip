// calib.hip — what THIS box's fp32 matrix pipe and HBM sustain right now: the yardstick bench.py prints next to the step
// (`calib`), so that a number measured on a slower-clocked box can be told from a slower kernel.  Not on the step's path.
//
//   pcg_calib_mfma  one 256-thread block per (CU, round): every wave issues back-to-back v_mfma_f32_32x32x2_f32 on four
//                   independent accumulators — nothing but the matrix pipe is busy, i.e. the ceiling the implicit-GEMM
//                   family is priced against, at the clock the part holds for that load.  Per block it leaves the shader
//                   clock ticks (s_memtime) and the 100 MHz reference ticks (s_memrealtime) of its loop.
//   pcg_calib_copy  dst <- src, 16 bytes per lane, grid-stride: read + write bandwidth of HBM.
#include "pcg_common.h"

namespace pcg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) calib_mfma_kernel(const float* __restrict__ seed, float* __restrict__ sink, int iters,
                                                         unsigned long long* __restrict__ stamps) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int t = 0; t < 4; ++t) {
    a[t] = seed[threadIdx.x * 4 + t];
    b[t] = seed[1024 + threadIdx.x * 4 + t];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[3 - t], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 - t], b[t], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 - t], b[3 - t], acc[3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

__global__ void __launch_bounds__(256) calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cus = prop.multiProcessorCount;
  }
  return cus;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int32_t pcg_calib_mfma_blocks(int32_t rounds) {
  const int cus = cu_count();
  return cus > 0 && rounds > 0 ? cus * rounds : 0;
}

// 8 KB of seed operands, then blocks*256 floats of sink, then blocks*2 uint64 stamps
extern "C" size_t pcg_calib_mfma_workspace_bytes(int32_t rounds) {
  const size_t blocks = (size_t)pcg_calib_mfma_blocks(rounds);
  return 8192 + blocks * 256 * 4 + blocks * 16;
}

extern "C" int pcg_calib_mfma(int32_t iters, int32_t rounds, void* workspace, size_t workspace_bytes, double* flop_out,
                              uint64_t** stamps_out, pcg_stream_t stream) {
  const int32_t blocks = pcg_calib_mfma_blocks(rounds);
  PCG_REQUIRE(blocks > 0, "pcg_calib_mfma: no GPU (hipGetDeviceProperties failed) or rounds <= 0");
  PCG_REQUIRE(iters > 0, "pcg_calib_mfma: iters must be positive");
  PCG_REQUIRE(workspace != nullptr && workspace_bytes >= pcg_calib_mfma_workspace_bytes(rounds),
              "pcg_calib_mfma: workspace of %zu bytes needed (pcg_calib_mfma_workspace_bytes), got %zu",
              pcg_calib_mfma_workspace_bytes(rounds), workspace_bytes);
  char* base = static_cast<char*>(workspace);
  float* seed = reinterpret_cast<float*>(base);                 // caller-initialised: any finite values (2048 floats)
  float* sink = reinterpret_cast<float*>(base + 8192);
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(base + 8192 + (size_t)blocks * 1024);
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), seed, sink, iters, stamps);
  if (flop_out) *flop_out = (double)blocks * 4.0 * (double)iters * 16.0 * 4096.0;     // 4 waves x 16 MFMAs x 32*32*2*2 FLOP
  if (stamps_out) *stamps_out = reinterpret_cast<uint64_t*>(stamps);
  return launch_status("pcg_calib_mfma");
}

extern "C" int pcg_calib_copy(const void* src, void* dst, int64_t nbytes, pcg_stream_t stream) {
  PCG_REQUIRE(src && dst && nbytes > 0 && nbytes % 16 == 0, "pcg_calib_copy: need two buffers and a positive multiple of 16 bytes");
  PCG_REQUIRE((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0, "pcg_calib_copy: 16-byte alignment");
  const int cus = cu_count();
  PCG_REQUIRE(cus > 0, "pcg_calib_copy: no GPU");
  hipLaunchKernelGGL(calib_copy_kernel, dim3(cus * 8), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float4*>(src), static_cast<float4*>(dst), nbytes / 16);
  return launch_status("pcg_calib_copy");
}
