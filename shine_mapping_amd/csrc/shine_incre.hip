// shine_incre.hip — the incremental-mapping epilogues of the hot path (config ncd_incre_reg):
//
//   shine_regularize             FeatureOctree.cal_regularization   model/feature_octree.py:246-255
//       reg = sum_levels sum_{u in unique(hierarchical_indices)} importance[u] * (F[u] - F_last[u])^2
//       The reference finds the touched rows with a sort-based unique() of 8N int64 per level per iteration; here
//       the fused step leaves a byte flag per touched row (shine_train_step `touched`), and one row-parallel
//       pass evaluates value and gradient and clears the flags (same launch).  grad_on[s] = 0 reproduces the reference's
//       attached-clone quirk (:160): the term adds to the loss value but not to the gradient.
//   shine_importance_accumulate  the per-chunk epilogue of cal_feature_importance   utils/incre_learning.py:36-40
//       importance += |grad| ; grad = 0 ; importance[trash row] = 0
#include "shine_internal.hpp"

namespace shine {

struct RegArgs {
  const float* feat[SHINE_MAX_LEVELS];
  const float* last[SHINE_MAX_LEVELS];
  const float* imp[SHINE_MAX_LEVELS];
  float* grad[SHINE_MAX_LEVELS];
  unsigned char* touched[SHINE_MAX_LEVELS];
  long long rows[SHINE_MAX_LEVELS];
  long long start[SHINE_MAX_LEVELS + 1];  // prefix of rows over levels (work partition)
  int grad_on[SHINE_MAX_LEVELS];
  int n_levels;
  int keep_flags;  // 1: the flags are the optimiser's sticky active-row flags — bit 0 = this iteration; left as they are
  float lambda;
  double* out;  // out[0] += reg (unweighted)
};

// lane = row: the sweep reads one flag byte per row (coalesced) and only the touched rows (a few thousand of ~200 k at
// the reference's batch size) fetch their 3 x 32 bytes; the flag is cleared in the same pass.  (One thread per ELEMENT,
// with a level search and a flag read each, took 28 us per iteration — more than the fused step itself.)
__global__ __launch_bounds__(256) void k_regularize(RegArgs a) {
  __shared__ double s_red[4];
  double acc = 0.0;
  const long long total = a.start[a.n_levels];
  for (long long row_g = (long long)blockIdx.x * 256 + threadIdx.x; row_g < total; row_g += (long long)gridDim.x * 256) {
    int s = 0;
    while (s + 1 < a.n_levels && row_g >= a.start[s + 1]) ++s;
    const long long r = row_g - a.start[s];
    if (a.touched[s][r] & 1) {
      if (!a.keep_flags) a.touched[s][r] = 0;
      const float4* fp = reinterpret_cast<const float4*>(a.feat[s] + r * F);
      const float4* lp = reinterpret_cast<const float4*>(a.last[s] + r * F);
      const float4* ip = reinterpret_cast<const float4*>(a.imp[s] + r * F);
      const float4 f0 = fp[0], f1 = fp[1], l0 = lp[0], l1 = lp[1], w0 = ip[0], w1 = ip[1];
      const float d[F] = {f0.x - l0.x, f0.y - l0.y, f0.z - l0.z, f0.w - l0.w, f1.x - l1.x, f1.y - l1.y, f1.z - l1.z, f1.w - l1.w};
      const float w[F] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int q = 0; q < F; ++q) acc += (double)(w[q] * d[q] * d[q]);
      if (a.grad_on[s] && a.grad[s]) {  // one thread per row: no atomics
        float4* gp = reinterpret_cast<float4*>(a.grad[s] + r * F);
        float4 g0 = gp[0], g1 = gp[1];
        const float k = 2.0f * a.lambda;
        g0.x += k * w[0] * d[0]; g0.y += k * w[1] * d[1]; g0.z += k * w[2] * d[2]; g0.w += k * w[3] * d[3];
        g1.x += k * w[4] * d[4]; g1.y += k * w[5] * d[5]; g1.z += k * w[6] * d[6]; g1.w += k * w[7] * d[7];
        gp[0] = g0;
        gp[1] = g1;
      }
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(a.out, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(256) void k_importance(float* imp, float* grad, long long n_elems, long long trash_begin) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_elems; e += (long long)gridDim.x * 256) {
    const float g = grad[e];
    grad[e] = 0.f;
    imp[e] = e >= trash_begin ? 0.f : imp[e] + fabsf(g);
  }
}

// FeatureOctree.update's feature-side appends (model/feature_octree.py:147-160) for every level that grew, one launch:
//   hier_features[s]       = cat(old[:-1], std * randn(added + 1, 8) with its last row zeroed)
//   importance_weight[s]   = cat(old_imp[:-1], zeros(added + 1, 8))
//   features_last_frame[s] = hier_features[s].clone()
struct AppendArgs {
  const float4* old_feat[SHINE_MAX_LEVELS];
  const float4* old_imp[SHINE_MAX_LEVELS];  // or null (not incremental)
  const float4* noise[SHINE_MAX_LEVELS];    // randn(added + 1, 8)
  float4* feat[SHINE_MAX_LEVELS];
  float4* imp[SHINE_MAX_LEVELS];
  float4* last[SHINE_MAX_LEVELS];
  long long keep4[SHINE_MAX_LEVELS];   // float4s carried over: (old rows without the trash row) * 2
  long long total4[SHINE_MAX_LEVELS];  // float4s of the new table incl. its trash row
  float std;
};
__global__ __launch_bounds__(256) void k_append_rows(AppendArgs a) {
  const int s = blockIdx.y;
  const long long keep4 = a.keep4[s], total4 = a.total4[s];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
    float4 v;
    if (e < keep4) {
      v = a.old_feat[s][e];
    } else if (e >= total4 - 2) {
      v = make_float4(0.f, 0.f, 0.f, 0.f);  // the new trash row
    } else {
      const float4 r = a.noise[s][e - keep4];
      v = make_float4(a.std * r.x, a.std * r.y, a.std * r.z, a.std * r.w);
    }
    a.feat[s][e] = v;
    if (a.last[s]) a.last[s][e] = v;
    if (a.imp[s]) a.imp[s][e] = e < keep4 ? a.old_imp[s][e] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace shine

using namespace shine;

extern "C" int shine_append_rows(int32_t n_levels, const float* const* old_feat, const float* const* old_imp,
                                 const float* const* noise, const int64_t* old_rows, const int64_t* added, float feature_std,
                                 float* const* feat, float* const* imp, float* const* last, void* stream) {
  if (n_levels < 1 || n_levels > SHINE_MAX_LEVELS || !old_feat || !noise || !old_rows || !added || !feat)
    return set_error(SHINE_E_INVALID, "shine_append_rows: null argument");
  AppendArgs a = {};
  a.std = feature_std;
  long long max4 = 0;
  for (int s = 0; s < n_levels; ++s) {
    if (added[s] < 0 || old_rows[s] < 0) return set_error(SHINE_E_INVALID, "shine_append_rows: negative row count");
    if (!old_feat[s] || !noise[s] || !feat[s] || (imp && imp[s] && !(old_imp && old_imp[s])))
      return set_error(SHINE_E_INVALID, "shine_append_rows: null level tensor");
    if ((((size_t)old_feat[s] | (size_t)noise[s] | (size_t)feat[s] | (size_t)(imp ? imp[s] : nullptr) |
          (size_t)(old_imp ? old_imp[s] : nullptr) | (size_t)(last ? last[s] : nullptr)) & 15))
      return set_error(SHINE_E_INVALID, "shine_append_rows: level tensors must be 16-byte aligned");
    a.old_feat[s] = (const float4*)old_feat[s];
    a.old_imp[s] = (const float4*)(old_imp ? old_imp[s] : nullptr);
    a.noise[s] = (const float4*)noise[s];
    a.feat[s] = (float4*)feat[s];
    a.imp[s] = (float4*)(imp ? imp[s] : nullptr);
    a.last[s] = (float4*)(last ? last[s] : nullptr);
    a.keep4[s] = old_rows[s] * (F / 4);
    a.total4[s] = (old_rows[s] + added[s] + 1) * (F / 4);
    if (a.total4[s] > max4) max4 = a.total4[s];
  }
  long long blocks = (max4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_append_rows, dim3((unsigned)blocks, (unsigned)n_levels), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}


extern "C" int shine_regularize(int32_t n_levels, const float* const* feats, const float* const* feats_last,
                                const float* const* importance, float* const* grad_feats,
                                unsigned char* const* touched, const int64_t* rows, const int32_t* grad_on,
                                float lambda_forget, double* reg_out, int32_t out_zeroed, int32_t keep_flags, void* stream) {
  if (n_levels < 1 || n_levels > SHINE_MAX_LEVELS || !feats || !feats_last || !importance || !touched || !rows ||
      !reg_out)
    return set_error(SHINE_E_INVALID, "shine_regularize: null argument");
  RegArgs a = {};
  a.n_levels = n_levels;
  a.lambda = lambda_forget;
  a.out = reg_out;
  a.keep_flags = keep_flags ? 1 : 0;
  a.start[0] = 0;
  for (int s = 0; s < n_levels; ++s) {
    if (!feats[s] || !feats_last[s] || !importance[s] || !touched[s])
      return set_error(SHINE_E_INVALID, "shine_regularize: null level pointer");
    if ((((size_t)feats[s] | (size_t)feats_last[s] | (size_t)importance[s] |
          (size_t)(grad_feats && grad_feats[s] ? grad_feats[s] : nullptr)) & 15))
      return set_error(SHINE_E_INVALID, "shine_regularize: level tensors must be 16-byte aligned");
    a.feat[s] = feats[s];
    a.last[s] = feats_last[s];
    a.imp[s] = importance[s];
    a.grad[s] = grad_feats ? grad_feats[s] : nullptr;
    a.touched[s] = touched[s];
    a.rows[s] = rows[s];
    a.grad_on[s] = grad_on ? grad_on[s] : 1;
    a.start[s + 1] = a.start[s] + rows[s];  // the trash row is excluded: its importance is reset to 0 (incre_learning.py:40)
  }
  hipStream_t st = (hipStream_t)stream;
  if (!out_zeroed) SHINE_HIP_CHECK(hipMemsetAsync(reg_out, 0, sizeof(double), st));
  const long long total = a.start[n_levels];  // rows
  if (total == 0) return SHINE_OK;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_regularize, dim3((unsigned)blocks), dim3(256), 0, st, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_importance_accumulate(float* importance, float* grad, int64_t rows, void* stream) {
  if (!importance || !grad || rows < 0) return set_error(SHINE_E_INVALID, "shine_importance_accumulate: null argument");
  const long long n_elems = (rows + 1) * F;
  long long blocks = (n_elems + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_importance, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, importance, grad, n_elems,
                     (long long)rows * F);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
