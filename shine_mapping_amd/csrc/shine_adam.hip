// shine_adam.hip — fused dense Adam for the feature tables + decoder.
//
// Replaces `opt.step()` + `opt.zero_grad()` of the inner loop (shine_batch.py:208-210) for the optimiser the
// reference builds in setup_optimizer (utils/tools.py:57-83): torch.optim.Adam(betas=(0.9,0.99), eps=adam_eps),
// L2 weight decay on the decoder group only (:62), one learning rate per feature level (:68-72).
// torch's Adam (no amsgrad, not decoupled):   g' = g + wd*p ;  m = b1 m + (1-b1) g' ;  v = b2 v + (1-b2) g'^2 ;
//   p -= lr/(1-b1^t) * m / ( sqrt(v)/sqrt(1-b2^t) + eps )
// Dense semantics are kept on purpose: every row moves every step (momentum on untouched rows), exactly like the
// reference.  One launch over all tensors (segment table in kernel arguments), 16 B per lane, and the gradient is
// cleared in the same pass (28 -> 24 B of traffic per parameter, no separate zero-fill).
#include "shine_internal.hpp"

namespace shine {

constexpr int ADAM_MAX_SEG = 16;

struct AdamSeg {
  float* p;
  float* g;
  float* m;
  float* v;
  long long n;      // elements
  long long start;  // prefix in float4 units
  float lr, wd;
  unsigned char* flags;  // [n / 8] sticky touched-row flags of a [rows, 8] table (active rows: shine_hip.h), or null: dense
};
struct AdamArgs {
  AdamSeg seg[ADAM_MAX_SEG];
  int n_seg;
  long long total4;  // float4 units over all segments
  float b1, b2, eps, bc1, bc2_sqrt;  // bc1 = 1 - b1^t ; bc2_sqrt = sqrt(1 - b2^t)
  int zero_grad;
  // graph-replayable form (shine_adam_step_dev): step counter and learning rates live in device memory
  long long* step_state;  // int64[8]: [0] = steps taken so far, [1] = the two bias corrections (floats), [2] / [3] = beta^t as
                          // doubles (adam_advance, shine_internal.hpp)
  const float* lr_dev;    // [n_seg] or null
};

struct AdamScalars {
  float b1, b2, eps, bc1, bc2_sqrt;
};
__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamScalars& a, float lr, float wd) {
  const float gg = g + wd * p;
  m = a.b1 * m + (1.0f - a.b1) * gg;
  v = a.b2 * v + (1.0f - a.b2) * gg * gg;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (lr / a.bc1) * (m / denom);
}

// graph-replayable form: one thread counts the step and derives the bias corrections (adam_advance, shine_internal.hpp) —
// unless the fused step that precedes the optimiser in the iteration already did (cfg->adam_state): then there is no launch
__global__ void k_adam_prep(long long* step_state, float b1, float b2) { adam_advance(step_state, b1, b2); }

__global__ __launch_bounds__(256) void k_adam(const AdamArgs a) {
  __shared__ float s_lr[ADAM_MAX_SEG];
  // (the kernel argument itself is never written: a modified by-value struct would be copied to scratch memory)
  AdamScalars sc = {a.b1, a.b2, a.eps, a.bc1, a.bc2_sqrt};
  if (a.step_state) {  // wave-uniform: k_adam_prep counted the step and left the two bias corrections in step_state[1]
    const float* bc = reinterpret_cast<const float*>(a.step_state + 1);
    sc.bc1 = bc[0];
    sc.bc2_sqrt = bc[1];
    if (threadIdx.x < a.n_seg) s_lr[threadIdx.x] = a.lr_dev ? a.lr_dev[threadIdx.x] : a.seg[threadIdx.x].lr;
    __syncthreads();
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total4; i += (long long)gridDim.x * 256) {
    int s = 0;
    while (s + 1 < a.n_seg && i >= a.seg[s + 1].start) ++s;
    const AdamSeg& S = a.seg[s];
    const float lr_s = a.step_state ? s_lr[s] : S.lr;
    const long long e = (i - S.start) * 4;
    if (S.flags) {  // exact active rows: a row that never had a gradient has m = v = g = 0 and is left as it is (not read)
      const unsigned char f = S.flags[e >> 3];
      if (f == 0) continue;
      if ((f & 1) && (e & 7) == 0) S.flags[e >> 3] = 2;  // "touched by this iteration" becomes "touched earlier"
    }
    if (e + 4 <= S.n && ((((size_t)S.p | (size_t)S.g | (size_t)S.m | (size_t)S.v) & 15) == 0)) {
      float4 p = *reinterpret_cast<float4*>(S.p + e), g = *reinterpret_cast<float4*>(S.g + e),
             m = *reinterpret_cast<float4*>(S.m + e), v = *reinterpret_cast<float4*>(S.v + e);
      adam1(p.x, g.x, m.x, v.x, sc, lr_s, S.wd);
      adam1(p.y, g.y, m.y, v.y, sc, lr_s, S.wd);
      adam1(p.z, g.z, m.z, v.z, sc, lr_s, S.wd);
      adam1(p.w, g.w, m.w, v.w, sc, lr_s, S.wd);
      *reinterpret_cast<float4*>(S.p + e) = p;
      *reinterpret_cast<float4*>(S.m + e) = m;
      *reinterpret_cast<float4*>(S.v + e) = v;
      if (a.zero_grad) *reinterpret_cast<float4*>(S.g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long long k = e; k < S.n && k < e + 4; ++k) {
        float p = S.p[k], g = S.g[k], m = S.m[k], v = S.v[k];
        adam1(p, g, m, v, sc, lr_s, S.wd);
        S.p[k] = p;
        S.m[k] = m;
        S.v[k] = v;
        if (a.zero_grad) S.g[k] = 0.f;
      }
    }
  }
}

}  // namespace shine

using namespace shine;

static int adam_impl(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const int64_t* numel, const float* lr, const float* lr_dev,
                     const float* weight_decay, float beta1, float beta2, float eps, int64_t step, long long* step_state,
                     int32_t zero_grad, unsigned char* const* row_flags, void* stream) {
  if (n_tensors < 1 || n_tensors > ADAM_MAX_SEG || !params || !grads || !exp_avg || !exp_avg_sq || !numel ||
      (!lr && !lr_dev) || !weight_decay || (step < 1 && !step_state))
    return set_error(SHINE_E_INVALID, "shine_adam_step: bad argument");
  AdamArgs a = {};
  long long start = 0;
  for (int s = 0; s < n_tensors; ++s) {
    if (!params[s] || !grads[s] || !exp_avg[s] || !exp_avg_sq[s] || numel[s] < 0)
      return set_error(SHINE_E_INVALID, "shine_adam_step: null tensor");
    a.seg[s].p = params[s];
    a.seg[s].g = grads[s];
    a.seg[s].m = exp_avg[s];
    a.seg[s].v = exp_avg_sq[s];
    a.seg[s].n = numel[s];
    a.seg[s].start = start;
    a.seg[s].lr = lr ? lr[s] : 0.f;
    a.seg[s].wd = weight_decay[s];
    a.seg[s].flags = row_flags ? row_flags[s] : nullptr;
    if (a.seg[s].flags && (numel[s] % 8 != 0 || weight_decay[s] != 0.f))
      return set_error(SHINE_E_INVALID, "shine_adam_step: row_flags are for [rows, 8] feature tables without weight decay");
    start += (numel[s] + 3) / 4;
  }
  a.n_seg = n_tensors;
  a.total4 = start;
  a.b1 = beta1;
  a.b2 = beta2;
  a.eps = eps;
  a.bc1 = step_state ? 1.f : (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = step_state ? 1.f : (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.zero_grad = zero_grad & 1;
  a.step_state = step_state;
  a.lr_dev = lr_dev;
  if (start == 0) return SHINE_OK;
  long long blocks = (start + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (step_state && !(zero_grad & 2))  // (bit 1: the fused step of this iteration advanced the state already)
    hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(1), 0, (hipStream_t)stream, step_state, beta1, beta2);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_adam_step(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, const float* lr,
                               const float* weight_decay, float beta1, float beta2, float eps, int64_t step,
                               int32_t zero_grad, unsigned char* const* row_flags, void* stream) {
  return adam_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, nullptr, weight_decay, beta1, beta2, eps, step,
                   nullptr, zero_grad, row_flags, stream);
}

extern "C" int shine_adam_step_dev(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                                   float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev,
                                   const float* weight_decay, float beta1, float beta2, float eps, int64_t* step_state,
                                   int32_t zero_grad, unsigned char* const* row_flags, void* stream) {
  if (!lr_dev || !step_state) return set_error(SHINE_E_INVALID, "shine_adam_step_dev: null lr_dev/step_state");
  return adam_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, nullptr, lr_dev, weight_decay, beta1, beta2, eps, 0,
                   (long long*)step_state, zero_grad, row_flags, stream);
}
