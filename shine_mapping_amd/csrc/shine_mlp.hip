// shine_mlp.hip — Tier A: Decoder.sdf (model/decoder.py:49-63) as a twice-differentiable op.
//
// The reference's decoder is three nn.Linear + ReLU; autograd differentiates it once for cur_loss.backward()
// (shine_batch.py:208-209) and twice when get_gradient(create_graph=True) feeds the eikonal term
// (utils/tools.py:175-185, shine_batch.py:141-142,182-185).  Here each of the three passes is one launch:
//   forward            pred = w3 . relu(W2 relu(W1 f + b1) + b2) + b3
//   backward           given g = d loss / d pred:  d/df = g J  (J = W1^T (m1 .* W2^T (m2 .* w3))) and the six weight grads
//   backward-backward  given r = d loss / d(d/df):  d/dg = r . J,  dW1 += g v1 (x) r,  dW2 += g v2 (x) a1,  dw3 += g a2
//                      (a1 = m1 .* W1 r,  a2 = m2 .* W2 a1; the ReLU masks are piecewise constant: nothing flows to f)
// lane = point, weights broadcast from LDS, activations recomputed from `feat` in every pass (8 floats in, nothing
// saved between passes); weight grads contract over the wave's 64 points through LDS staging (contract64) into
// lane-owned accumulators that live across the grid-stride loop, flushed once per wave with fp32 atomics.
// This tier keeps the reference's drivers unchanged; the benchmarked path is the fused step (shine_step_v1.hip).
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

template <int MODE>
__global__ __launch_bounds__(256) void k_mlp(MlpArgs a) {
  __shared__ float s_mlp[1380];
  __shared__ float s_stage[MODE == 0 ? 1 : 4 * 64 * ST];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < H * F; i += 256) s_mlp[MLP_W1 + i] = a.mlp[0][i];
  for (int i = tid; i < H * H; i += 256) s_mlp[MLP_W2 + i] = a.mlp[2][i];
  if (tid < H) {
    s_mlp[MLP_B1 + tid] = a.mlp[1][tid];
    s_mlp[MLP_B2 + tid] = a.mlp[3][tid];
    s_mlp[MLP_W3 + tid] = a.mlp[4][tid];
  }
  if (tid == 0) s_mlp[MLP_B3] = a.mlp[5][0];
  __syncthreads();
  float* st = s_stage + (MODE == 0 ? 0 : wv * 64 * ST);

  float accW2[16], accW1[4];
  float accb2 = 0.f, accb1 = 0.f, accw3 = 0.f, accb3 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) accW2[q] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) accW1[q] = 0.f;
  const int jj = lane & 31, hi = lane >> 5;
  const bool wgrad = MODE != 0 && a.want_wgrad;

  for (long long base = (long long)blockIdx.x * 256; base < a.n; base += (long long)gridDim.x * 256) {
    const long long i = base + tid;
    const bool valid = i < a.n;
    float f[F];
    if (valid) {
      const float4* row = reinterpret_cast<const float4*>(a.feat + i * F);
      const float4 r0 = row[0], r1 = row[1];
      f[0] = r0.x, f[1] = r0.y, f[2] = r0.z, f[3] = r0.w, f[4] = r1.x, f[5] = r1.y, f[6] = r1.z, f[7] = r1.w;
    } else {
#pragma unroll
      for (int q = 0; q < F; ++q) f[q] = 0.f;
    }
    // ---- forward (recomputed in every mode): same association as the fused kernels' VALU decoder
    float h1[H], h2[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      float z = s_mlp[MLP_B1 + k];
#pragma unroll
      for (int q = 0; q < F; ++q) z = fmaf(s_mlp[MLP_W1 + k * F + q], f[q], z);
      h1[k] = fmaxf(z, 0.f);
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
      float z = s_mlp[MLP_B2 + j];
#pragma unroll
      for (int k = 0; k < H; ++k) z = fmaf(s_mlp[MLP_W2 + j * H + k], h1[k], z);
      h2[j] = fmaxf(z, 0.f);
    }
    if (MODE == 0) {
      float y = s_mlp[MLP_B3];
#pragma unroll
      for (int j = 0; j < H; ++j) y = fmaf(s_mlp[MLP_W3 + j], h2[j], y);
      if (valid) a.pred[i] = y;
      continue;
    }
    const float g = valid ? a.g[i] : 0.f;  // padding lanes contribute zeros to the contractions

    if (MODE == 1) {
      // ---- d2 = g (m2 .* w3);  d1 = m1 .* W2^T d2;  df = W1^T d1
      float d1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) d1[k] = 0.f;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const float d2 = h2[j] > 0.f ? g * s_mlp[MLP_W3 + j] : 0.f;
        if (wgrad) {
          st[lane * ST + j] = d2;
          st[lane * ST + 32 + j] = h1[j];
        }
#pragma unroll
        for (int k = 0; k < H; ++k) d1[k] = fmaf(s_mlp[MLP_W2 + j * H + k], d2, d1[k]);
      }
      float df[F];
#pragma unroll
      for (int q = 0; q < F; ++q) df[q] = 0.f;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        d1[k] = h1[k] > 0.f ? d1[k] : 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) df[q] = fmaf(s_mlp[MLP_W1 + k * F + q], d1[k], df[q]);
      }
      if (valid && a.grad_feat) {
        float4* o = reinterpret_cast<float4*>(a.grad_feat + i * F);
        o[0] = make_float4(df[0], df[1], df[2], df[3]);
        o[1] = make_float4(df[4], df[5], df[6], df[7]);
      }
      if (wgrad) {
        __syncthreads();
        contract64<16>(st, jj, 32 + hi * 16, accW2, accb2, true);  // dW2 += d2 (x) h1 ; db2 += d2
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) st[lane * ST + k] = d1[k];
#pragma unroll
        for (int q = 0; q < F; ++q) st[lane * ST + 32 + q] = f[q];
#pragma unroll
        for (int j = 0; j < H; ++j) st[lane * ST + 40 + j] = g * h2[j];
        st[lane * ST + 72] = g;
        __syncthreads();
        contract64<4>(st, jj, 32 + hi * 4, accW1, accb1, true);  // dW1 += d1 (x) f ; db1 += d1
        for (int pp = 0; pp < 64; ++pp) {
          accw3 += st[pp * ST + 40 + jj];  // dw3 += g h2
          accb3 += st[pp * ST + 72];       // db3 += g
        }
        __syncthreads();
      }
    } else {
      // ---- v2 = m2 .* w3;  v1 = m1 .* W2^T v2;  J = W1^T v1;  a1 = m1 .* W1 r;  a2 = m2 .* W2 a1
      float rr[F];
      if (valid) {
        const float4* row = reinterpret_cast<const float4*>(a.r + i * F);
        const float4 r0 = row[0], r1 = row[1];
        rr[0] = r0.x, rr[1] = r0.y, rr[2] = r0.z, rr[3] = r0.w, rr[4] = r1.x, rr[5] = r1.y, rr[6] = r1.z, rr[7] = r1.w;
      } else {
#pragma unroll
        for (int q = 0; q < F; ++q) rr[q] = 0.f;
      }
      float v1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) v1[k] = 0.f;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const float v2 = h2[j] > 0.f ? s_mlp[MLP_W3 + j] : 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) v1[k] = fmaf(s_mlp[MLP_W2 + j * H + k], v2, v1[k]);
      }
      float gj = 0.f;  // r . J = sum_k v1[k] (W1 r)[k]
      float a1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) {
        v1[k] = h1[k] > 0.f ? v1[k] : 0.f;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) t = fmaf(s_mlp[MLP_W1 + k * F + q], rr[q], t);
        gj = fmaf(v1[k], t, gj);
        a1[k] = h1[k] > 0.f ? t : 0.f;
      }
      if (valid && a.grad_g) a.grad_g[i] = gj;
      if (wgrad) {
        // dW2 += (g v2) (x) a1
#pragma unroll
        for (int j = 0; j < H; ++j) {
          st[lane * ST + j] = h2[j] > 0.f ? g * s_mlp[MLP_W3 + j] : 0.f;
          st[lane * ST + 32 + j] = a1[j];
        }
        __syncthreads();
        float dummy = 0.f;
        contract64<16>(st, jj, 32 + hi * 16, accW2, dummy, false);
        __syncthreads();
        // dW1 += (g v1) (x) r ; dw3 += g a2
#pragma unroll
        for (int k = 0; k < H; ++k) st[lane * ST + k] = g * v1[k];
#pragma unroll
        for (int q = 0; q < F; ++q) st[lane * ST + 32 + q] = rr[q];
#pragma unroll
        for (int j = 0; j < H; ++j) {
          float t = 0.f;
#pragma unroll
          for (int k = 0; k < H; ++k) t = fmaf(s_mlp[MLP_W2 + j * H + k], a1[k], t);
          st[lane * ST + 40 + j] = h2[j] > 0.f ? g * t : 0.f;
        }
        __syncthreads();
        contract64<4>(st, jj, 32 + hi * 4, accW1, dummy, false);
        for (int pp = 0; pp < 64; ++pp) accw3 += st[pp * ST + 40 + jj];
        __syncthreads();
      }
    }
  }

  if (wgrad) {
#pragma unroll
    for (int q = 0; q < 16; ++q) atomic_add_f32(a.grad_mlp[2] + jj * H + hi * 16 + q, accW2[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) atomic_add_f32(a.grad_mlp[0] + jj * F + hi * 4 + q, accW1[q]);
    if (hi == 0) {
      atomic_add_f32(a.grad_mlp[4] + jj, accw3);
      if (MODE == 1) {  // the eikonal pass sends nothing to the biases
        atomic_add_f32(a.grad_mlp[1] + jj, accb1);
        atomic_add_f32(a.grad_mlp[3] + jj, accb2);
        if (lane == 0) atomic_add_f32(a.grad_mlp[5], accb3);
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

}  // namespace shine

using namespace shine;

extern "C" int shine_mlp_forward(const float* feat, int64_t n, const float* const* mlp, float* pred_out, void* stream) {
  MlpArgs a = {};
  int rc = fill_args(&a, "shine_mlp_forward: null argument", feat, n, mlp, nullptr);
  if (rc != SHINE_OK) return rc;
  if (n > 0 && !pred_out) return set_error(SHINE_E_INVALID, "shine_mlp_forward: null output");
  if (n == 0) return SHINE_OK;
  a.pred = pred_out;
  hipLaunchKernelGGL(k_mlp<0>, dim3(mlp_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
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
  hipLaunchKernelGGL(k_mlp<1>, dim3(mlp_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
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
  hipLaunchKernelGGL(k_mlp<2>, dim3(mlp_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
