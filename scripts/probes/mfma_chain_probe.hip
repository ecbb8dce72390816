// mfma_chain_probe.hip — how fast ONE wave per SIMD can issue v_mfma_f32_32x32x2_f32 as a function of the number of independent
// accumulators it rotates over (the dependency distance of the chain), and the same with 2 / 3 waves per SIMD.  Question behind it
// (r03): a 64x32 wave tile has 2 accumulators — is such a wave capped below the pipe's rate?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(1024) chain(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* stamps) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int t = 0; t < 4; ++t) { a[t] = in[(threadIdx.x & 255) * 4 + t]; b[t] = in[1024 + (threadIdx.x & 255) * 4 + t]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + i) & 3], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int threads, int iters, const float* in, float* out, unsigned long long* st, int cus) {
  hipLaunchKernelGGL(chain<NACC>, dim3(cus), dim3(threads), 0, 0, in, out, iters, st);
  hipLaunchKernelGGL(chain<NACC>, dim3(cus), dim3(threads), 0, 0, in, out, iters, st);
  hipDeviceSynchronize();
  std::vector<unsigned long long> hs(cus);
  hipMemcpy(hs.data(), st, cus * 8, hipMemcpyDeviceToHost);
  std::sort(hs.begin(), hs.end());
  const double per = (double)hs[cus / 2] / (iters * 4.0 * NACC);   // shader clocks per MFMA of one wave
  const int waves_per_simd = threads / 256;
  printf("accumulators %d, waves per SIMD %d: %.1f clocks per MFMA per wave -> pipe busy %.0f %% (64 clocks per MFMA = 100 %%)\n", NACC, waves_per_simd, per,
         100.0 * 64.0 * waves_per_simd / per);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<float> h(2048);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2 - 1;
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 8192); hipMalloc(&out, cus * 1024 * 4); hipMalloc(&st, cus * 8);
  hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
  for (int threads : {256, 512, 768}) {
    run<1>(threads, 4096, in, out, st, cus);
    run<2>(threads, 4096, in, out, st, cus);
    run<4>(threads, 2048, in, out, st, cus);
  }
  return 0;
}
