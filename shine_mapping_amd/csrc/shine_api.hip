// shine_api.hip — library-level entry points of libshine_hip.so (version, errors, kernel dispatch).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "shine_step_common.hpp"

namespace shine {

static thread_local char g_err[512] = "ok";
static thread_local int g_err_code = 0;

int set_error(int code, const char* msg) {
  g_err_code = code;
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}

int set_hip_error(hipError_t e, const char* what) {
  g_err_code = SHINE_E_HIP;
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what ? what : "?");
  return SHINE_E_HIP;
}

}  // namespace shine

// the lane-per-point reference kernel's TRAINING instantiations live in the CHECK library only
// (libshine_check.so = this library with shine_step_v0.hip compiled -DSHINE_V0_TRAIN=1): weak here
extern "C" int shine_train_step_v0(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int64_t*, int64_t, const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, void*) __attribute__((weak));

extern "C" int shine_train_step_v3(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                   const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, unsigned char* const*, void*, size_t, void*);

extern "C" int shine_version(void) { return 200; }

extern "C" const char* shine_error_string(int code) {
  if (code == SHINE_OK) return "ok";
  if (code == shine::g_err_code) return shine::g_err;
  switch (code) {
    case SHINE_E_INVALID: return "invalid argument";
    case SHINE_E_HIP: return "HIP runtime error";
    case SHINE_E_NOMEM: return "out of device memory";
    case SHINE_E_STATE: return "table handle in the wrong state";
    default: return "unknown error";
  }
}

extern "C" int shine_train_step(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                const float* sdf_label, const float* weight, const int32_t* perm,
                                const int32_t* slots, const int64_t* n_surf, int64_t n, const float* const* feats,
                                const int64_t* rows, const float* const* mlp, float* pred_out, float* grad_x_out,
                                float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                unsigned char* const* touched, void* workspace, size_t workspace_bytes, void* stream) {
  if (!cfg) return shine::set_error(SHINE_E_INVALID, "shine_train_step: null config");
  if (n == 0) {  // empty batch: nothing to add to the grads, loss terms are zero
    if (loss_parts && hipMemsetAsync(loss_parts, 0, 4 * sizeof(double), (hipStream_t)stream) != hipSuccess)
      return shine::set_error(SHINE_E_HIP, "hipMemsetAsync(loss_parts)");
    return SHINE_OK;
  }
  // kernel_variant (low byte): 0 the product kernel (shine_step_v3.hip: planned / pool batches, <= 4 featured levels — every
  // shipped yaml; 5 / 6 force / forbid its build for tables beyond the Infinity Cache, which 0 picks by table size).  The check library adds 1 = the lane-per-point reference kernel (any batch, up to 8 levels; the
  // on-device cross-check of the tests).
  const int variant = cfg->kernel_variant & 0xff;
  if (variant != 0 && variant != 1 && variant != 4 && variant != 5 && variant != 6)
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: unknown kernel_variant (0 / 4: fused step, 5 / 6: its far / near "
                                             "build whatever the table size, 1: check library)");
  if (cfg->defer_reduce && (variant == 1 || cfg->n_levels > shine::LCAP || !slots))
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: defer_reduce is for the product kernel on a planned batch");
  if (cfg->n_surf_parts > 1 && (variant == 1 || cfg->n_levels > shine::LCAP || !slots))
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: n_surf_parts > 1 is for the product kernel (sum the parts first)");
  if (variant == 1 || cfg->n_levels > shine::LCAP || !slots) {
    if (!shine_train_step_v0)
      return shine::set_error(SHINE_E_INVALID,
                              variant == 1 ? "shine_train_step: kernel_variant 1 is part of the check library (libshine_check.so)"
                              : !slots     ? "shine_train_step: the batch needs a plan (shine_plan_batch: node order + hash slots)"
                                           : "shine_train_step: more than 4 featured levels need the check library's kernel");
    if (touched) return shine::set_error(SHINE_E_INVALID, "shine_train_step: touched flags need a planned batch and <= 4 levels");
    return shine_train_step_v0(t, cfg, coord, sdf_label, weight, perm, n_surf, n, feats, rows, mlp, pred_out, grad_x_out,
                               grad_feats, grad_mlp, loss_parts, stream);
  }
  return shine_train_step_v3(t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out, grad_x_out,
                             grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes, stream);
}
