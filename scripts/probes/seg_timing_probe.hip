// Where a generator segment kernel spends its time: builds csrc/house_fused.hip with -DPCG_SEG_TIMING (phase timestamps from
// thread 0 of every block, 100 MHz clock) and runs the forward chain at batch 4096 on synthetic buffers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPCG_SEG_TIMING -o seg_timing_probe seg_timing_probe.hip
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
__device__ unsigned long long* pcg_dbg_ts = nullptr;
#include "../../promptable-counterfactual-gan_amd/csrc/house_fused.hip"
namespace pcg {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int launch_status(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
}
int main() {
  const int B = 4096, T = 70, NC = 10;
  pcg_house_g_desc d{};
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n + 3) / 4 * 4; return o; };
  d.fc_in_w = take(32 * 38); d.fc_in_b = take(32);
  for (int k = 0; k < 5; ++k) {
    d.fc1_w[k] = take(1024); d.fc1_b[k] = take(32); d.bn1_g[k] = take(32); d.bn1_b[k] = take(32);
    d.fc2_w[k] = take(1024); d.fc2_b[k] = take(32); d.bn2_g[k] = take(32); d.bn2_b[k] = take(32);
    d.film_gamma_w[k] = take(32 * 21); d.film_gamma_b[k] = take(32); d.film_beta_w[k] = take(32 * 21); d.film_beta_b[k] = take(32);
  }
  d.cont_w = take(NC * 32); d.cont_b = take(NC);
  const int sizes[7] = {9, 30, 6, 2, 5, 5, 13};
  int c = 0;
  for (int h = 0; h < 7; ++h) { d.head_w[h] = take(sizes[h] * 32); d.head_b[h] = take(sizes[h]); d.seg[h] = c; c += sizes[h]; }
  d.seg[7] = c; d.nheads = 7; d.ncont = NC; d.D = 17; d.NC = 4; d.hidden = 32; d.nblocks = 5;
  auto dev = [&](size_t n, float v) { float* p; hipMalloc(&p, n * 4); std::vector<float> h(n, v); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
  pcg_house_g_fwd_args a{};
  a.params = dev(off, 0.01f); a.x = dev(B * 17, 0.5f); a.onehot = dev(B * 4, 0.25f); a.mask = dev(B * 17, 1.f); a.noise = dev(B * T, 0.1f);
  a.inp = dev(B * 38, 0.f); a.H = dev(6 * B * 32, 0.f); a.Z1 = dev(5 * B * 32, 0.f); a.Z2 = dev(5 * B * 32, 0.f); a.P = dev(10 * 64 * 64, 0.f);
  a.SM = dev(10 * 64, 0.f); a.cont = dev(B * NC, 0.f); a.logits = dev(B * T, 0.f); a.soft = dev(B * T, 0.f); a.hard = nullptr;
  a.B = B; a.eps = 1e-5f; a.momentum = 0.1f; a.tau = 0.5f; a.res_scale = 0.1f;
  unsigned long long* ts; hipMalloc(&ts, 64 * 16 * 8); hipMemset(ts, 0, 64 * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(pcg_dbg_ts), &ts, sizeof(ts));
  hipStream_t s; hipStreamCreate(&s);
  for (int it = 0; it < 5; ++it) if (pcg_house_g_fwd(&d, &a, s)) return 1;
  hipStreamSynchronize(s);
  std::vector<unsigned long long> h(64 * 16);
  hipMemcpy(h.data(), ts, 64 * 16 * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < 64; ++b) { if (h[b * 16] < t0) t0 = h[b * 16]; if (h[b * 16 + 7] > t1) t1 = h[b * 16 + 7]; }
  printf("g_fwd_a4 (k = 4), 10 ns ticks; first block entry -> last block exit: %.2f us\n", (t1 - t0) * 0.01);
  const char* names[8] = {"entry", "burst issued", "in LDS (loads back)", "bn_finish", "FiLM products (MFMA) x2", "a1 parked + barrier", "fc2 product (MFMA)", "store + colsums"};
  for (int b : {0, 1, 31, 63}) {
    printf("block %2d: entry at +%.2f us;", b, (h[b * 16] - t0) * 0.01);
    for (int i = 1; i < 8; ++i) printf("  %s %.2f", names[i], (h[b * 16 + i] - h[b * 16 + i - 1]) * 0.01);
    printf("\n");
  }
  const char* hn[7] = {"entry", "burst issued", "parked + barrier", "matrix product (wave 0)", "softmax (wave 0)", "barrier", "tiles out"};
  t0 = ~0ull; t1 = 0;
  for (int b = 0; b < 64; ++b) { if (h[b * 16 + 8] < t0) t0 = h[b * 16 + 8]; if (h[b * 16 + 14] > t1) t1 = h[b * 16 + 14]; }
  printf("g_heads4: first block entry -> last block exit: %.2f us\n", (t1 - t0) * 0.01);
  for (int b : {0, 31, 63}) {
    printf("block %2d: entry at +%.2f us;", b, (h[b * 16 + 8] - t0) * 0.01);
    for (int i = 1; i < 7; ++i) printf("  %s %.2f", hn[i], (h[b * 16 + 8 + i] - h[b * 16 + 8 + i - 1]) * 0.01);
    printf("\n");
  }
  return 0;
}
