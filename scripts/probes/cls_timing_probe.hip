// Phase timestamps inside the fused classifier forward (-DPCG_SEG_TIMING build of csrc/house_classifier_fused.hip), batch 4096.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPCG_SEG_TIMING -o cls_timing_probe cls_timing_probe.hip
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
__device__ unsigned long long* pcg_dbg_ts = nullptr;
#include "../../promptable-counterfactual-gan_amd/csrc/house_classifier_fused.hip"
namespace pcg {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int launch_status(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
}
int main() {
  const int B = 4096;
  auto dev = [&](size_t n, float v) { float* p; hipMalloc(&p, n * 4); std::vector<float> h(n, v); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
  const int Kp[5] = {20, 256, 256, 128, 64}, N[5] = {256, 256, 128, 64, 4};
  const float* wk[5]; const float* bs[5];
  for (int l = 0; l < 5; ++l) { wk[l] = dev((size_t)Kp[l] * N[l], 0.01f); bs[l] = dev(N[l], 0.f); }
  float* x = dev((size_t)B * 17, 0.5f);
  float *a1 = dev((size_t)B * 256, 0), *a2 = dev((size_t)B * 256, 0), *a3 = dev((size_t)B * 128, 0), *a4 = dev((size_t)B * 64, 0), *lg = dev((size_t)B * 4, 0);
  unsigned long long* ts; hipMalloc(&ts, 256 * 16 * 8); hipMemset(ts, 0, 256 * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(pcg_dbg_ts), &ts, sizeof(ts));
  hipStream_t s; hipStreamCreate(&s);
  for (int it = 0; it < 5; ++it) if (pcg_house_classifier_fwd(x, B, wk, bs, a1, a2, a3, a4, lg, s)) return 1;
  hipStreamSynchronize(s);
  std::vector<unsigned long long> h(256 * 16);
  hipMemcpy(h.data(), ts, 256 * 16 * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < 256; ++b) { if (h[b * 16] < t0) t0 = h[b * 16]; if (h[b * 16 + 6] > t1) t1 = h[b * 16 + 6]; }
  printf("classifier_fwd: first block entry -> last block exit: %.2f us\n", (t1 - t0) * 0.01);
  const char* names[7] = {"entry", "input rows", "17->256", "256->256", "256->128 (split 2)", "128->64 (split 4)", "64->4"};
  for (int b : {0, 128, 255}) {
    printf("block %3d: entry +%.2f;", b, (h[b * 16] - t0) * 0.01);
    for (int i = 1; i < 7; ++i) printf("  %s %.2f", names[i], (h[b * 16 + i] - h[b * 16 + i - 1]) * 0.01);
    printf("\n");
  }
  return 0;
}
