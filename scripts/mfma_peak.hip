// mfma_peak.hip — what the f32 MFMA pipe sustains on THIS box (diagnostic; not part of the library).
// One 256-thread block per CU (one wave per SIMD), `blocks_per_cu` rounds of blocks; every wave issues back-to-back
// v_mfma_f32_32x32x2_f32 on 4 independent accumulators.  Reports TFLOP/s by host events and the in-kernel shader
// clock (s_memtime ticks / s_memrealtime 100 MHz ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* stamps) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int t = 0; t < 4; ++t) { a[t] = in[threadIdx.x * 4 + t]; b[t] = in[1024 + threadIdx.x * 4 + t]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[3 - t], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 - t], b[t], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 - t], b[3 - t], acc[3], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 8192;     // 16 MFMAs per iteration per wave
  const int rounds = argc > 2 ? atoi(argv[2]) : 1;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = cus * rounds;
  std::vector<float> h(2048);
  for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX * 2 - 1;
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 8192); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
  hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> hs(blocks * 2);
  hipMemcpy(hs.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int b = 0; b < blocks; ++b) { clk.push_back((double)hs[2 * b] / hs[2 * b + 1] * 100.0); cyc.push_back((double)hs[2 * b] / (iters * 16.0)); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("CUs %d blocks %d iters %d zero %d: %.3f ms/launch  %.1f TFLOP/s | in-kernel clock MHz median %.0f (min %.0f max %.0f) | cycles per MFMA median %.2f\n",
         cus, blocks, iters, zero, ms, flop / ms / 1e9, clk[clk.size() / 2], clk.front(), clk.back(), cyc[cyc.size() / 2]);
  return 0;
}
