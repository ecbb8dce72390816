// Cost of a grid barrier among 64 / 128 / 256 resident blocks of 256 threads on gfx950 (arrival counter, agent scope), against
// the cost of a kernel boundary between two dependent launches.   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) barriers(unsigned* st, int n, float* data, float* out) {
  unsigned epoch = 0;
  float acc = 0.f;
  for (int it = 0; it < n; ++it) {
    // a little cross-block traffic so the fences have something to do: every block writes a line, reads its neighbour's after the barrier
    if (threadIdx.x < 64) data[((size_t)(it & 1) * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = acc + it;
    epoch += 1;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) __hip_atomic_fetch_add(st, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_fetch_add(st, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      const unsigned target = epoch * gridDim.x;
      while (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (MODE != 2) __builtin_amdgcn_s_sleep(1); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (threadIdx.x < 64) acc += data[((size_t)(it & 1) * gridDim.x + (blockIdx.x + 1) % gridDim.x) * 64 + threadIdx.x];
  }
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) one_step(int it, float* data, float* out, int nb) {
  float acc = 0.f;
  if (it > 0 && threadIdx.x < 64) acc = data[((size_t)((it - 1) & 1) * nb + (blockIdx.x + 1) % nb) * 64 + threadIdx.x] + out[blockIdx.x * 64 + threadIdx.x];
  if (threadIdx.x < 64) { data[((size_t)(it & 1) * nb + blockIdx.x) * 64 + threadIdx.x] = acc + it; out[blockIdx.x * 64 + threadIdx.x] = acc; }
}

int main() {
  unsigned* st; float *data, *out;
  hipMalloc(&st, 16); hipMalloc(&data, 2 * 256 * 64 * 4); hipMalloc(&out, 256 * 64 * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 200;
  for (int nb : {64, 128, 256}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemsetAsync(st, 0, 16, s); hipMemsetAsync(data, 0, 2 * 256 * 64 * 4, s);
        hipEventRecord(e0, s);
        if (mode == 0) hipLaunchKernelGGL(barriers<0>, dim3(nb), dim3(256), 0, s, st, N, data, out);
        if (mode == 1) hipLaunchKernelGGL(barriers<1>, dim3(nb), dim3(256), 0, s, st, N, data, out);
        if (mode == 2) hipLaunchKernelGGL(barriers<2>, dim3(nb), dim3(256), 0, s, st, N, data, out);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("blocks %3d mode %d (0 release-rmw+sleep, 1 fence+relaxed+sleep, 2 no sleep): %.2f us per barrier\n", nb, mode, best * 1e3f / N);
    }
    // kernel boundaries: N dependent launches captured in a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(one_step, dim3(nb), dim3(256), 0, s, it, data, out, nb);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("blocks %3d kernel boundary (graph of %d dependent launches): %.2f us per launch\n", nb, N, best * 1e3f / N);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
