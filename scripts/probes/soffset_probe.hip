// soffset_probe.hip — is the SGPR offset of a raw buffer access part of the hardware range check on gfx950?
// A 64-byte buffer resource over a 1 KB allocation; stores / loads with voffset + soffset on either side of num_records.
// Build: hipcc --offload-arch=gfx950 -o soffset_probe soffset_probe.hip ; prints which stores landed.
#include <hip/hip_runtime.h>
#include <cstdio>
using rsrc_t = __amdgpu_buffer_rsrc_t;
__global__ void probe(float* buf, float* out) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 64, 0x00020000);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(1.f), r, 0, 0, 0);      // voff 0, soff 0: in range
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(2.f), r, 16, 32, 0);    // 48 total: in range
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(3.f), r, 16, 64, 0);    // voff in range, total 80: out?
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(4.f), r, 0, 128, 0);    // voff 0, soff beyond
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(5.f), r, 96, 0, 0);     // voff beyond: must be dropped
    out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 16, 64, 0)); // reads buf[20] if not range-checked
    out[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 96, 0, 0));  // must be 0
  }
}
int main() {
  float *buf, *out, h[256], ho[2];
  hipMalloc(&buf, 1024); hipMalloc(&out, 8);
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  hipMemcpy(buf, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out);
  hipMemcpy(h, buf, 1024, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 8, hipMemcpyDeviceToHost);
  printf("buf[0]=%g (1: stored)  buf[12]=%g (2: stored)  buf[20]=%g (3 if soffset is NOT range-checked, 120 if it is)  "
         "buf[32]=%g (4 if NOT checked, 132 if it is)  buf[24]=%g (124: voffset beyond is dropped)\n", h[0], h[12], h[20], h[32], h[24]);
  printf("load voff16+soff64 = %g (0 if checked)  load voff96 = %g (0)\n", ho[0], ho[1]);
  return 0;
}
