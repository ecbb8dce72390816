// ldsdma_probe.hip — buffer_load_dwordx4 ... lds on gfx950: (1) does the builtin compile with 16-byte width, (2) where do the lanes'
// bytes land (base + lane*16?), (3) what does a lane whose offset is out of range do to its LDS slot: write zeros or leave it?
#include <hip/hip_runtime.h>
#include <cstdio>
using rsrc_t = __amdgpu_buffer_rsrc_t;
__global__ void probe(const float* src, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 7.f;
  __syncthreads();
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 4096, 0x00020000);
  const unsigned lane = threadIdx.x;
  // even lanes: a valid, lane-dependent source (reversed order: lane l reads chunk 63-l); odd lanes: out of range
  const unsigned voff = (lane & 1) ? 0x80000000u : (63u - lane) * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  float *src, *out, h[1024], ho[512];
  hipMalloc(&src, 4096); hipMalloc(&out, 2048);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
  hipMemcpy(ho, out, 2048, hipMemcpyDeviceToHost);
  printf("lds[60..63] (before the destination)  = %g %g %g %g (7: untouched)\n", ho[60], ho[61], ho[62], ho[63]);
  printf("lane 0 slot  lds[64..67]   = %g %g %g %g (expect src chunk 63: 252 253 254 255)\n", ho[64], ho[65], ho[66], ho[67]);
  printf("lane 1 slot  lds[68..71]   = %g %g %g %g (out-of-range lane: 0 = zeros written, 7 = left alone)\n", ho[68], ho[69], ho[70], ho[71]);
  printf("lane 2 slot  lds[72..75]   = %g %g %g %g (expect chunk 61: 244..247)\n", ho[72], ho[73], ho[74], ho[75]);
  printf("lane 63 slot lds[316..319] = %g %g %g %g (out of range)\n", ho[316], ho[317], ho[318], ho[319]);
  printf("lds[320..323] (after)      = %g %g %g %g (7)\n", ho[320], ho[321], ho[322], ho[323]);
  return 0;
}
