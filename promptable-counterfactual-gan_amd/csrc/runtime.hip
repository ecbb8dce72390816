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

// ABI history.  v5 (r04): + grouped batches (pcg_conv2d_fwd_bn_g, pcg_bn_apply_act_g, pcg_conv2d_dgrad_bn_phases, pcg_conv2d_dgrad_bnbwd_g,
//   pcg_bn_bwd_partial_g(+_workspace_bytes), pcg_bn_act_bwd_premask_g(+pcg_bn_act_bwd_g_workspace_bytes), pcg_bce_pair),
//   pcg_conv2d_fwd_bnbwd_thin(+_ok, +_workspace_bytes), pcg_conv_weight_adjoint_many, pcg_conv_reset_scratch, pcg_dp_barrier,
//   pcg_dp_rccl_version; the stream-K scratch registry is keyed by (device, stream).
// v4 (r03): + pcg_calib_*, pcg_conv_set_scratch / pcg_conv_scratch_*_bytes, pcg_tune_set, pcg_conv_plan_describe, pcg_debug_stamp_buffer,
//   pcg_instnorm_bwd_fused / _bwd_bwd_act, pcg_rowsum3, pcg_norm_sum, pcg_house_diag, pcg_house_batch_draws_counter, the fused
//   critic-stage entry points.  v3 (r02): pcg_adam_step_capturable scratch is 48 bytes; pcg_linear_wgrad_grouped takes whole layers;
//   + the pcg_house_* / spectral-norm reps / seq entry points.  v2 (r02): + pcg_conv2d_*_xf, pcg_bn_train_stats_coef, pcg_dp_*;
//   pcg_bn_bwd_partial takes fp64 partial rows.
extern "C" int pcg_abi_version(void) { return 5; }
extern "C" const char* pcg_last_error(void) { return pcg::g_err; }
extern "C" const char* pcg_target_arch(void) { return "gfx950"; }
