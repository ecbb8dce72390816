// Where the grouped weight-gradient kernel spends its time (phase timestamps, -DPCG_SEG_TIMING build of csrc/tabular.hip): the
// generator's 29 layers at batch 4096.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPCG_SEG_TIMING -o wgrad_timing_probe wgrad_timing_probe.hip
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
__device__ unsigned long long* pcg_dbg_ts = nullptr;
#include "../../promptable-counterfactual-gan_amd/csrc/tabular.hip"
namespace pcg {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int launch_status(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
}
int main() {
  const int B = 4096;
  auto dev = [&](size_t n, float v) { float* p; hipMalloc(&p, n * 4); std::vector<float> h(n, v); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
  std::vector<pcg_wgrad_item> items;
  auto add = [&](int O, int I) { pcg_wgrad_item it{}; it.dy = dev((size_t)B * O, 0.01f); it.x = dev((size_t)B * I, 0.5f); it.dW = dev((size_t)O * I, 0.f); it.db = dev(O, 0.f);
                                 it.ldy = O; it.ldx = I; it.O = O; it.I = I; it.accumulate_w = 1; it.accumulate_b = 1; items.push_back(it); };
  add(32, 38);
  for (int k = 0; k < 5; ++k) { add(32, 32); add(32, 32); add(32, 21); add(32, 21); }
  add(10, 32);
  for (int n : {9, 30, 6, 2, 5, 5, 13}) add(n, 32);
  const size_t wsb = pcg_linear_wgrad_grouped_workspace_bytes(B, items.data(), (int)items.size());
  void* ws; hipMalloc(&ws, wsb);
  int* tk; hipMalloc(&tk, 4096 * 4); hipMemset(tk, 0, 4096 * 4);
  const int NB = 8 * 64;
  unsigned long long* ts; hipMalloc(&ts, NB * 16 * 8); hipMemset(ts, 0, NB * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(pcg_dbg_ts), &ts, sizeof(ts));
  hipStream_t s; hipStreamCreate(&s);
  for (int it = 0; it < 5; ++it) if (pcg_linear_wgrad_grouped(items.data(), (int)items.size(), B, ws, wsb, tk, s)) return 1;
  hipStreamSynchronize(s);
  std::vector<unsigned long long> h(NB * 16);
  hipMemcpy(h.data(), ts, NB * 16 * 8, hipMemcpyDeviceToHost);
  // timestamps are indexed by blockIdx.x (the slab block, 0..7): the last writer among the tiles wins — a sample, not a census
  const char* names[8] = {"entry", "descriptors", "loads + MFMA", "combine in LDS", "partials stored", "stores complete + barrier", "ticket", "last block: reduce + write"};
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < 8; ++b) { if (h[b * 16] && h[b * 16] < t0) t0 = h[b * 16]; for (int i = 0; i < 8; ++i) if (h[b * 16 + i] > t1) t1 = h[b * 16 + i]; }
  printf("sampled span %.2f us\n", (t1 - t0) * 0.01);
  for (int b = 0; b < 8; ++b) {
    printf("slab block %d: entry +%.2f;", b, (h[b * 16] - t0) * 0.01);
    for (int i = 1; i < 8; ++i) if (h[b * 16 + i] >= h[b * 16 + i - 1] && h[b * 16 + i]) printf("  %s %.2f", names[i], (h[b * 16 + i] - h[b * 16 + i - 1]) * 0.01);
    printf("\n");
  }
  return 0;
}
