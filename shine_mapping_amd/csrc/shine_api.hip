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

extern "C" int shine_train_step_v0(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int64_t*, int64_t, const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, void*);

extern "C" int shine_train_step_v1(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                   const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, unsigned char* const*, void*, size_t, void*);

extern "C" int shine_train_step_v2(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                   const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, unsigned char* const*, void*, size_t, void*);

extern "C" int shine_train_step_v3(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                   const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, unsigned char* const*, void*, size_t, void*);

extern "C" int shine_train_step_v5(const shine_tables*, const shine_step_config*, const float*, const float*,
                                   const float*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                   const float* const*,
                                   const int64_t*, const float* const*, float*, float*, float* const*, float* const*,
                                   double*, unsigned char* const*, void*, size_t, void*);

namespace shine {
// which configurations the 16-point-tile kernel serves when the caller leaves the choice to the library
// (kernel_variant 0): BCE steps (the eikonal build stays on the 32-point kernel).  SHINE_KERNEL=v1 forces the 32-point one.
bool v2_serves(const shine_step_config* cfg) {
  static const int force_v1 = []() {
    const char* e = getenv("SHINE_KERNEL");
    return (e && strcmp(e, "v1") == 0) ? 1 : 0;
  }();
  const int variant = cfg->kernel_variant & 0xff;
  if (variant == 3 || variant == 4) return !cfg->eikonal_on && cfg->n_levels <= LCAP;
  if (variant != 0 || force_v1) return false;
  return !cfg->eikonal_on && cfg->n_levels <= LCAP;
}
// the lane = (point, level) kernel: any planned / pool batch with <= 4 levels, unless another kernel is forced
bool v3_serves(const shine_step_config* cfg, bool planned) {
  static const int force_v1 = []() {
    const char* e = getenv("SHINE_KERNEL");
    return (e && strcmp(e, "v1") == 0) ? 1 : 0;
  }();
  const int variant = cfg->kernel_variant & 0xff;
  if (!planned || cfg->n_levels > LCAP) return false;
  if (variant == 4 || variant == 5) return true;
  return variant == 0 && !force_v1;
}
// the role-specialised form of the same kernel (shine_step_v5.hip): planned / pool batches large enough to fill its
// per-SIMD pipelines (SHINE_V5_MIN_POINTS, default below; kernel_variant 5 forces it, 4 forces shine_step_v3.hip)
bool v5_serves(const shine_step_config* cfg, bool planned, long long n) {
  static const long long min_points = []() {
    const char* e = getenv("SHINE_V5_MIN_POINTS");
    return e ? atoll(e) : (1ll << 62);
  }();
  const int variant = cfg->kernel_variant & 0xff;
  if (!planned || cfg->n_levels > LCAP) return false;
  if (variant == 5) return true;
  return variant == 0 && n >= min_points;
}
}  // namespace shine

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
  static const int force_v0 = []() {
    const char* e = getenv("SHINE_KERNEL");
    return (e && strcmp(e, "v0") == 0) ? 1 : 0;
  }();
  // kernel_variant (low byte): 0 auto, 1 the simple v0 kernel, 2 the 32-point-tile MFMA kernel (shine_step_v1.hip),
  // 3 the 16-point-tile MFMA kernel (shine_step_v2.hip).  The MFMA kernels handle up to 4 featured levels (every shipped
  // yaml); v0 is the cross-check kernel and the fallback for deeper trees.
  const int variant = cfg->kernel_variant & 0xff;
  const size_t v2_need = shine_train_step_workspace_bytes(cfg, n);
  if (variant == 3 && !shine::v2_serves(cfg))
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: kernel_variant 3 (16-point tiles, in-kernel probing) serves BCE steps only");
  if (variant == 4 && (!slots || cfg->n_levels > 4))
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: kernel_variant 4 needs a planned batch (slots) and <= 4 levels");
  // 16-point tiles: planned / pool batches (BCE or BCE + eikonal) run on the lane = (point, level) kernel
  // (shine_step_v3.hip); BCE batches without a plan (in-kernel probing) on shine_step_v2.hip; everything else that has
  // <= 4 levels on the 32-point kernel (shine_step_v1.hip).  kernel_variant 2 / 3 / 4 force v1 / v2 / v3.
  if (variant == 5 && (!slots || cfg->n_levels > 4))
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: kernel_variant 5 needs a planned batch (slots) and <= 4 levels");
  if (!force_v0 && workspace && workspace_bytes >= v2_need && shine::v5_serves(cfg, slots != nullptr, n))
    return shine_train_step_v5(t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                               grad_x_out, grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes,
                               stream);
  if (!force_v0 && workspace && workspace_bytes >= v2_need && shine::v3_serves(cfg, slots != nullptr))
    return shine_train_step_v3(t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                               grad_x_out, grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes,
                               stream);
  // the |weight|-scaled BCE (loss_weight_on, off in every shipped yaml) is built into the planned-batch kernel above and
  // into v0: a weighted batch WITHOUT a plan takes the simple kernel
  const bool weighted_unplanned = cfg->loss_weight_on != 0;
  if (weighted_unplanned && (variant == 2 || variant == 3))
    return shine::set_error(SHINE_E_INVALID, "shine_train_step: loss_weight_on runs on kernel_variant 0 / 1 / 4");
  if (!force_v0 && !weighted_unplanned && workspace && workspace_bytes >= v2_need && shine::v2_serves(cfg))
    return shine_train_step_v2(t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                               grad_x_out, grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes,
                               stream);
  if (!force_v0 && !weighted_unplanned && variant != 1 && cfg->n_levels <= 4)
    return shine_train_step_v1(t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                               grad_x_out, grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes,
                               stream);
  if (touched) return shine::set_error(SHINE_E_INVALID, "shine_train_step: touched flags need <= 4 featured levels");
  return shine_train_step_v0(t, cfg, coord, sdf_label, weight, perm, n_surf, n, feats, rows, mlp, pred_out,
                             grad_x_out, grad_feats, grad_mlp, loss_parts, stream);
}
