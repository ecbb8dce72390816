// runtime.hip — library-level entry points: ABI version, target, thread-local error text.
#include "pcg_common.h"
#include <stdarg.h>
#include <stdio.h>

namespace pcg {
namespace {
thread_local char g_err[512] = "";
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_status(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PCG_OK;
  set_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return PCG_ERR_LAUNCH;
}
}  // namespace pcg

extern "C" int pcg_abi_version(void) { return 4; }   // v4 (r03): + pcg_calib_*.   v2 (r02): + pcg_conv2d_*_xf, pcg_bn_train_stats_coef, pcg_dp_*; pcg_bn_bwd_partial takes fp64 partial rows.  v3 (r02): pcg_adam_step_capturable scratch is 48 bytes; pcg_linear_wgrad_grouped takes whole layers; + the pcg_house_* / spectral-norm reps / seq entry points
extern "C" const char* pcg_last_error(void) { return pcg::g_err; }
extern "C" const char* pcg_target_arch(void) { return "gfx950"; }
