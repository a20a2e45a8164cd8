// shine_mlp.hip — Tier A: Decoder.sdf (model/decoder.py:49-63) as a twice-differentiable op.
//
// The reference's decoder is three nn.Linear + ReLU; autograd differentiates it once for cur_loss.backward()
// (shine_batch.py:208-209) and twice when get_gradient(create_graph=True) feeds the eikonal term
// (utils/tools.py:175-185, shine_batch.py:141-142,182-185).  Here each of the three passes is one launch:
//   forward            pred = w3 . relu(W2 relu(W1 f + b1) + b2) + b3
//   backward           given g = d loss / d pred:  d/df = g J  (J = W1^T (m1 .* W2^T (m2 .* w3))) and the six weight grads
//   backward-backward  given r = d loss / d(d/df):  d/dg = r . J,  dW1 += g v1 (x) r,  dW2 += g v2 (x) a1,  dw3 += g a2
//                      (a1 = m1 .* W1 r,  a2 = m2 .* W2 a1; the ReLU masks are piecewise constant: nothing flows to f)
// lane = point, the weights are wave-uniform scalar loads (constant address space -> SGPR operands), activations are
// recomputed from `feat` in every pass (8 floats in, nothing saved between passes); weight grads contract over the
// wave's 64 points through LDS staging rows into lane-owned accumulators that live across the tile loop, summed over the
// workgroup's four waves and flushed once per workgroup with fp32 atomics.
// This tier keeps the reference's drivers unchanged; the benchmarked path is the fused step (shine_step_v3.hip).
#include "shine_internal.hpp"

namespace shine {

struct MlpArgs {
  const float* feat;    // [n][8]
  const float* g;       // [n] d loss / d pred                      (modes 1, 2)
  const float* r;       // [n][8] d loss / d(grad_feat)             (mode 2)
  const float* mlp[6];
  float* pred;          // [n]                                      (mode 0)
  float* grad_feat;     // [n][8] or null                           (mode 1)
  float* grad_g;        // [n] or null                              (mode 2)
  float* grad_mlp[6];   // accumulated into, or all null
  long long n;
  int want_wgrad;
};

// contract64 (device.hpp) with a bounded unroll: the fully unrolled 64-point loop keeps ~150 LDS results in flight
template <int NQ>
__device__ __forceinline__ void contract64r(const float* st, int j, int rb, float (&acc)[NQ], float& acc_left,
                                            bool with_bias) {
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(4)
  for (int p = 0; p < 64; ++p) {
    const float* row = st + p * ST;
    const float l = row[j];
    if (with_bias) acc_left += l;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = fmaf(l, row[rb + q], acc[q]);
  }
}

// a rolled loop over weight rows: opaque trip count (device.hpp), and keep the loop vectoriser away from it
#define ROW_LOOP_STR(x) #x
#define ROW_LOOP(n) _Pragma(ROW_LOOP_STR(clang loop vectorize(disable) interleave(disable) unroll_count(n)))

// ---- forward: lane = point, everything in registers; layer 1 unrolled (h1 is a register array), layer 2 a rolled loop
// over its weight rows
__global__ __launch_bounds__(256) void k_mlp_fwd(MlpArgs a) {
  cfloat *const W1 = uniform_ro(a.mlp[0]), *const B1 = uniform_ro(a.mlp[1]), *const W2 = uniform_ro(a.mlp[2]),
               *const B2 = uniform_ro(a.mlp[3]), *const W3 = uniform_ro(a.mlp[4]), *const B3 = uniform_ro(a.mlp[5]);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    const float4* row = reinterpret_cast<const float4*>(a.feat + i * F);
    const float4 r0 = row[0], r1 = row[1];
    const float f[F] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    float h1[H];
    cfloat *const W1i = relaunder(W1), *const B1i = relaunder(B1);
#pragma unroll
    for (int k = 0; k < H; ++k) {
      float z = B1i[k];
#pragma unroll
      for (int q = 0; q < F; ++q) z = fmaf(W1i[k * F + q], f[q], z);
      h1[k] = fmaxf(z, 0.f);
    }
    float y = B3[0];
    const int rows = opaque(H);
ROW_LOOP(2)
    for (int j = 0; j < rows; ++j) {
      float z = B2[j];
#pragma unroll
      for (int k = 0; k < H; ++k) z = fmaf(W2[j * H + k], h1[k], z);
      y = fmaf(W3[j], fmaxf(z, 0.f), y);
    }
    a.pred[i] = y;
  }
}

// ---- backward (MODE 1) and backward-backward (MODE 2): one wave = one 64-point tile at a time, lane = point.
// The layers are ROLLED loops over the weight rows (an unrolled 32x32 layer let the compiler keep a thousand loaded
// weights live and spill them); register arrays are only ever indexed statically (h1 / d1 / v1 / a1 inside the row
// loops), per-row results that a rolled loop produces go through the lane's own LDS staging row — the same rows the
// weight-grad contractions over the tile's 64 points read afterwards.  Nothing is shared between waves: DS operations of
// one wave execute in order, so the hand-offs need no workgroup barrier.
template <int MODE>
__global__ __launch_bounds__(256) void k_mlp_bwd(MlpArgs a) {
  __shared__ float s_stage[4 * 64 * ST];
  cfloat *const W1 = uniform_ro(a.mlp[0]), *const B1 = uniform_ro(a.mlp[1]), *const W2 = uniform_ro(a.mlp[2]),
               *const B2 = uniform_ro(a.mlp[3]), *const W3 = uniform_ro(a.mlp[4]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* st = s_stage + wv * 64 * ST;
  float* row = st + lane * ST;  // [0,32) left operands, [32,64) right operands, [64,75) spare

  float accW2[16], accW1[4];
  float accb2 = 0.f, accb1 = 0.f, accw3 = 0.f, accb3 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) accW2[q] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) accW1[q] = 0.f;
  const int jj = lane & 31, hi = lane >> 5;
  const bool wgrad = a.want_wgrad;

  const long long tiles = (a.n + 63) >> 6;
  const int rows = opaque(H);
  for (long long t = (long long)blockIdx.x * 4 + wv; t < tiles; t += (long long)gridDim.x * 4) {
    const long long i = t * 64 + lane;
    const bool valid = i < a.n;
    float f[F];
#pragma unroll
    for (int q = 0; q < F; ++q) f[q] = 0.f;
    float g = 0.f;  // padding lanes contribute zeros to the contractions
    if (valid) {
      const float4* fr = reinterpret_cast<const float4*>(a.feat + i * F);
      const float4 r0 = fr[0], r1 = fr[1];
      f[0] = r0.x, f[1] = r0.y, f[2] = r0.z, f[3] = r0.w, f[4] = r1.x, f[5] = r1.y, f[6] = r1.z, f[7] = r1.w;
      g = a.g[i];
    }
    // ---- forward, recomputed (8 floats in, nothing saved between passes): ReLU masks m1 / m2, h1 in registers
    unsigned int m1 = 0u, m2 = 0u;
ROW_LOOP(4)
    for (int k = 0; k < rows; ++k) {
      float z = B1[k];
#pragma unroll
      for (int q = 0; q < F; ++q) z = fmaf(W1[k * F + q], f[q], z);
      m1 |= (z > 0.f ? 1u : 0u) << k;
      row[32 + k] = fmaxf(z, 0.f);
    }
    wave_lds_fence();
    float h1[H];
#pragma unroll
    for (int k = 0; k < H; ++k) h1[k] = row[32 + k];
ROW_LOOP(2)
    for (int j = 0; j < rows; ++j) {
      float z = B2[j];
#pragma unroll
      for (int k = 0; k < H; ++k) z = fmaf(W2[j * H + k], h1[k], z);
      m2 |= (z > 0.f ? 1u : 0u) << j;
      if (MODE == 1 && wgrad) row[j] = g * fmaxf(z, 0.f);
    }

    if (MODE == 1) {
      if (wgrad) {  // dw3 += g h2
        wave_lds_fence();
        _Pragma("clang loop vectorize(disable) interleave(disable) unroll_count(8)")
        for (int pp = 0; pp < 64; ++pp) accw3 += st[pp * ST + jj];
        wave_lds_fence();
      }
      // ---- d2 = g (m2 .* w3);  d1 = m1 .* W2^T d2;  df = W1^T d1
      float d1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) d1[k] = 0.f;
ROW_LOOP(2)
      for (int j = 0; j < rows; ++j) {
        const float d2 = ((m2 >> j) & 1u) ? g * W3[j] : 0.f;
        if (wgrad) row[j] = d2;
#pragma unroll
        for (int k = 0; k < H; ++k) d1[k] = fmaf(W2[j * H + k], d2, d1[k]);
      }
      if (wgrad) {  // dW2 += d2 (x) h1 ; db2 += d2     (h1 is still staged at [32,64))
        wave_lds_fence();
        contract64r<16>(st, jj, 32 + hi * 16, accW2, accb2, true);
        wave_lds_fence();
      }
#pragma unroll
      for (int k = 0; k < H; ++k) row[k] = ((m1 >> k) & 1u) ? d1[k] : 0.f;
#pragma unroll
      for (int q = 0; q < F; ++q) row[32 + q] = f[q];
      row[40] = g;
      wave_lds_fence();
      float df[F];
#pragma unroll
      for (int q = 0; q < F; ++q) df[q] = 0.f;
ROW_LOOP(8)
      for (int k = 0; k < rows; ++k) {
        const float dk = row[k];
#pragma unroll
        for (int q = 0; q < F; ++q) df[q] = fmaf(W1[k * F + q], dk, df[q]);
      }
      if (valid && a.grad_feat) {
        float4* o = reinterpret_cast<float4*>(a.grad_feat + i * F);
        o[0] = make_float4(df[0], df[1], df[2], df[3]);
        o[1] = make_float4(df[4], df[5], df[6], df[7]);
      }
      if (wgrad) {  // dW1 += d1 (x) f ; db1 += d1 ; db3 += g
        contract64r<4>(st, jj, 32 + hi * 4, accW1, accb1, true);
        _Pragma("clang loop vectorize(disable) interleave(disable) unroll_count(8)")
        for (int pp = 0; pp < 64; ++pp) accb3 += st[pp * ST + 40];
        wave_lds_fence();
      }
    } else {
      // ---- v2 = m2 .* w3;  v1 = m1 .* W2^T v2;  J = W1^T v1;  a1 = m1 .* W1 r;  a2 = m2 .* W2 a1
      float rr[F];
#pragma unroll
      for (int q = 0; q < F; ++q) rr[q] = 0.f;
      if (valid) {
        const float4* rp = reinterpret_cast<const float4*>(a.r + i * F);
        const float4 r0 = rp[0], r1 = rp[1];
        rr[0] = r0.x, rr[1] = r0.y, rr[2] = r0.z, rr[3] = r0.w, rr[4] = r1.x, rr[5] = r1.y, rr[6] = r1.z, rr[7] = r1.w;
      }
      float v1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) v1[k] = 0.f;
ROW_LOOP(2)
      for (int j = 0; j < rows; ++j) {
        const float v2 = ((m2 >> j) & 1u) ? W3[j] : 0.f;
        if (wgrad) row[j] = g * v2;
#pragma unroll
        for (int k = 0; k < H; ++k) v1[k] = fmaf(W2[j * H + k], v2, v1[k]);
      }
      // t = W1 r, one row of W1 at a time, through the staging row (h1 is no longer needed there)
ROW_LOOP(4)
      for (int k = 0; k < rows; ++k) {
        float tk = 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) tk = fmaf(W1[k * F + q], rr[q], tk);
        row[32 + k] = tk;
      }
      wave_lds_fence();
      float gj = 0.f;  // r . J = sum_k v1[k] (W1 r)[k]
      float a1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const bool on = (m1 >> k) & 1u;
        const float tk = row[32 + k];
        v1[k] = on ? v1[k] : 0.f;
        gj = fmaf(v1[k], tk, gj);
        a1[k] = on ? tk : 0.f;
      }
      if (valid && a.grad_g) a.grad_g[i] = gj;
      if (wgrad) {
        // dW2 += (g v2) (x) a1
#pragma unroll
        for (int k = 0; k < H; ++k) row[32 + k] = a1[k];
        wave_lds_fence();
        float dummy = 0.f;
        contract64r<16>(st, jj, 32 + hi * 16, accW2, dummy, false);
        wave_lds_fence();
        // dW1 += (g v1) (x) r ; dw3 += g a2
#pragma unroll
        for (int k = 0; k < H; ++k) row[k] = g * v1[k];
#pragma unroll
        for (int q = 0; q < F; ++q) row[32 + q] = rr[q];
ROW_LOOP(2)
        for (int j = 0; j < rows; ++j) {
          float tj = 0.f;
#pragma unroll
          for (int k = 0; k < H; ++k) tj = fmaf(W2[j * H + k], a1[k], tj);
          row[40 + j] = ((m2 >> j) & 1u) ? g * tj : 0.f;
        }
        wave_lds_fence();
        contract64r<4>(st, jj, 32 + hi * 4, accW1, dummy, false);
        _Pragma("clang loop vectorize(disable) interleave(disable) unroll_count(8)")
        for (int pp = 0; pp < 64; ++pp) accw3 += st[pp * ST + 40 + jj];
        wave_lds_fence();
      }
    }
  }

  if (wgrad) {
    // One flush per WORKGROUP: the 1377 targets are the same for every wave of the launch, and same-address fp32 atomics
    // retire at ~3 per ns in total (measured: 1024 waves x 1377 atomics took 0.5 ms, the arithmetic ~20 us) — so the four
    // waves are summed through LDS first and the grid is kept small (mlp_bwd_grid).
    __syncthreads();  // every wave is done with its staging rows
    constexpr int NV = 24;
    float vals[NV];
#pragma unroll
    for (int q = 0; q < 16; ++q) vals[q] = accW2[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) vals[16 + q] = accW1[q];
    vals[20] = accw3, vals[21] = accb1, vals[22] = accb2, vals[23] = accb3;
#pragma unroll
    for (int v = 0; v < NV; ++v) s_stage[(wv * NV + v) * 64 + lane] = vals[v];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        vals[v] = (s_stage[v * 64 + lane] + s_stage[(NV + v) * 64 + lane]) +
                  (s_stage[(2 * NV + v) * 64 + lane] + s_stage[(3 * NV + v) * 64 + lane]);
#pragma unroll
      for (int q = 0; q < 16; ++q) atomic_add_f32(a.grad_mlp[2] + jj * H + hi * 16 + q, vals[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q) atomic_add_f32(a.grad_mlp[0] + jj * F + hi * 4 + q, vals[16 + q]);
      if (hi == 0) {
        atomic_add_f32(a.grad_mlp[4] + jj, vals[20]);
        if (MODE == 1) {  // the eikonal pass sends nothing to the biases
          atomic_add_f32(a.grad_mlp[1] + jj, vals[21]);
          atomic_add_f32(a.grad_mlp[3] + jj, vals[22]);
          if (lane == 0) atomic_add_f32(a.grad_mlp[5], vals[23]);
        }
      }
    }
  }
}

static int fill_args(MlpArgs* a, const char* what, const float* feat, int64_t n, const float* const* mlp,
                     float* const* grad_mlp) {
  if (n < 0 || !mlp || (n > 0 && !feat)) return set_error(SHINE_E_INVALID, what);
  a->feat = feat;
  a->n = n;
  a->want_wgrad = grad_mlp ? 1 : 0;
  for (int k = 0; k < 6; ++k) {
    if (!mlp[k]) return set_error(SHINE_E_INVALID, what);
    a->mlp[k] = mlp[k];
    if (grad_mlp) {
      if (!grad_mlp[k]) return set_error(SHINE_E_INVALID, what);
      a->grad_mlp[k] = grad_mlp[k];
    }
  }
  return SHINE_OK;
}

static unsigned mlp_grid(int64_t n) {
  const long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// backward launches that accumulate weight grads: every workgroup ends with 1377 atomics on the same addresses
// (~0.5 us per workgroup, serialised), a 64-point tile costs a wave ~10 us -> at most 96 workgroups
static unsigned mlp_bwd_grid(int64_t n, bool wgrad) {
  const unsigned g = mlp_grid(n);
  return (wgrad && g > 96u) ? 96u : g;
}

}  // namespace shine

using namespace shine;

extern "C" int shine_mlp_forward(const float* feat, int64_t n, const float* const* mlp, float* pred_out, void* stream) {
  MlpArgs a = {};
  int rc = fill_args(&a, "shine_mlp_forward: null argument", feat, n, mlp, nullptr);
  if (rc != SHINE_OK) return rc;
  if (n > 0 && !pred_out) return set_error(SHINE_E_INVALID, "shine_mlp_forward: null output");
  if (n == 0) return SHINE_OK;
  a.pred = pred_out;
  hipLaunchKernelGGL(k_mlp_fwd, dim3(mlp_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_mlp_backward(const float* feat, const float* grad_pred, int64_t n, const float* const* mlp,
                                  float* grad_feat_out, float* const* grad_mlp, void* stream) {
  MlpArgs a = {};
  int rc = fill_args(&a, "shine_mlp_backward: null argument", feat, n, mlp, grad_mlp);
  if (rc != SHINE_OK) return rc;
  if (n > 0 && !grad_pred) return set_error(SHINE_E_INVALID, "shine_mlp_backward: null grad_pred");
  if (n == 0 || (!grad_feat_out && !grad_mlp)) return SHINE_OK;
  a.g = grad_pred;
  a.grad_feat = grad_feat_out;
  hipLaunchKernelGGL(k_mlp_bwd<1>, dim3(mlp_bwd_grid(n, grad_mlp != nullptr)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_mlp_backward_backward(const float* feat, const float* grad_pred, const float* gg_feat, int64_t n,
                                           const float* const* mlp, float* grad_gpred_out, float* const* grad_mlp,
                                           void* stream) {
  MlpArgs a = {};
  int rc = fill_args(&a, "shine_mlp_backward_backward: null argument", feat, n, mlp, grad_mlp);
  if (rc != SHINE_OK) return rc;
  if (n > 0 && (!grad_pred || !gg_feat)) return set_error(SHINE_E_INVALID, "shine_mlp_backward_backward: null input");
  if (n == 0 || (!grad_gpred_out && !grad_mlp)) return SHINE_OK;
  a.g = grad_pred;
  a.r = gg_feat;
  a.grad_g = grad_gpred_out;
  hipLaunchKernelGGL(k_mlp_bwd<2>, dim3(mlp_bwd_grid(n, grad_mlp != nullptr)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
