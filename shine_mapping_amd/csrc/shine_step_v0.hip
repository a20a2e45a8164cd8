// shine_step_v0.hip — first correct gfx950 path for the SHINE hot path: one lane per point.  CHECK LIBRARY ONLY since round 4.
//
//   query   : FeatureOctree.query_feature      model/feature_octree.py:199-244
//   decode  : Decoder.sdf                      model/decoder.py:49-63
//   loss    : sdf_bce_loss / eikonal           utils/loss.py:17-24, shine_batch.py:141-142,182-185
//   backward: cur_loss.backward()              shine_batch.py:208-209   (closed form, SURVEY.md §8a math contract)
//
// Structure (v0 = the simple, obviously-correct variant the optimised kernels are checked against):
//   * lane = point; per level: quantise -> Morton -> hash probe -> 8 row gathers (2 x dwordx4 each);
//   * decoder on the VALU with the 1377 weights broadcast-read from LDS;
//   * decoder weight grads: per-point vectors are staged through LDS (per wave, [64][75] fp32) and
//     contracted over the 64 points by lane-owned register accumulators that live across the whole
//     persistent loop; flushed once per wave with fp32 atomics;
//   * feature grads: fp32 hardware atomics (global_atomic_add_f32) per corner row; the per-level
//     trash row (index -1, :205,231) is wave-reduced first because every miss in the batch hits it.
#include "shine_internal.hpp"

namespace shine {


struct StepArgs {
  LevelSet ls;
  const float* coord;
  const float* label;
  const float* weight;
  const int* perm;
  const long long* n_surf;
  const float* mlp[6];
  float* pred;
  float* feat_out;
  float* grad_x;
  long long* idx_out[SHINE_MAX_LEVELS];  // bottom-up
  float* grad_mlp[6];
  double* loss_parts;
  long long n;
  int n_levels;
  int reduction_sum;
  int decoder_grad_on;
  int weighted;  // BCE terms scaled by |weight| (utils/loss.py:18-19)
  float sigma;
  float weight_e;
  float inv_n;
};

__device__ __forceinline__ void load_mlp_to_lds(const StepArgs& a, float* s_mlp, int tid, int nthreads) {
  for (int i = tid; i < H * F; i += nthreads) s_mlp[MLP_W1 + i] = a.mlp[0][i];
  for (int i = tid; i < H; i += nthreads) s_mlp[MLP_B1 + i] = a.mlp[1][i];
  for (int i = tid; i < H * H; i += nthreads) s_mlp[MLP_W2 + i] = a.mlp[2][i];
  for (int i = tid; i < H; i += nthreads) s_mlp[MLP_B2 + i] = a.mlp[3][i];
  for (int i = tid; i < H; i += nthreads) s_mlp[MLP_W3 + i] = a.mlp[4][i];
  if (tid == 0) s_mlp[MLP_B3] = a.mlp[5][0];
}

template <bool POLY, bool EIK, bool TRAIN>
__global__ __launch_bounds__(256) void k_step_v0(StepArgs a) {
  __shared__ float s_mlp[1380];
  __shared__ float s_stage[TRAIN ? 4 * 64 * ST : 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  load_mlp_to_lds(a, s_mlp, tid, 256);
  // FeatureOctree.set_zero (model/feature_octree.py:78-81, called at the top of query_feature): the trash row of every
  // level is re-zeroed here — nothing in this kernel reads it (a miss contributes nothing), so there is no ordering to keep
  if (!TRAIN && blockIdx.x == 0 && tid < a.n_levels * F)
    const_cast<float*>(a.ls.lv[tid / F].feat)[a.ls.lv[tid / F].rows * F + (tid % F)] = 0.f;
  __syncthreads();
  float* st = s_stage + (TRAIN ? wv * 64 * ST : 0);
  const int L = a.n_levels;
  const float sigma = a.sigma;

  // lane-owned decoder-gradient accumulators (persist over the grid-stride loop)
  float accW2[16], accW1[4];
  float accb2 = 0.f, accb1 = 0.f, accw3 = 0.f, accb3 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) accW2[q] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) accW1[q] = 0.f;
  double loss_acc = 0.0, eik_acc = 0.0, cnt_acc = 0.0;
  float inv_nsurf = 0.f;
  if (EIK && TRAIN) {
    long long ns = a.n_surf ? *a.n_surf : 0;
    inv_nsurf = ns > 0 ? 1.0f / (float)ns : 0.f;
  }

  for (long long base = (long long)blockIdx.x * 256; base < a.n; base += (long long)gridDim.x * 256) {
    const long long i = base + tid;
    const bool valid = i < a.n;
    const long long p = valid ? (a.perm ? (long long)a.perm[i] : i) : 0;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (valid) {
      x0 = a.coord[3 * p];
      x1 = a.coord[3 * p + 1];
      x2 = a.coord[3 * p + 2];
    }

    // ------------------------------------------------------------------ query (all levels)
    float f[F];
    float A[F][3];
#pragma unroll
    for (int q = 0; q < F; ++q) {
      f[q] = 0.f;
      A[q][0] = A[q][1] = A[q][2] = 0.f;
    }
    for (int s = 0; s < L; ++s) {
      const LevelDev& Lv = a.ls.lv[s];
      int slot = -1;
      if (valid) {
        unsigned long long key = morton3(quantize(x0, Lv.res), quantize(x1, Lv.res), quantize(x2, Lv.res));
        slot = probe(Lv, key);
      }
      int ids[8];
      if (slot >= 0) {
        int4 v0 = Lv.vals[2 * slot], v1 = Lv.vals[2 * slot + 1];
        ids[0] = v0.x; ids[1] = v0.y; ids[2] = v0.z; ids[3] = v0.w;
        ids[4] = v1.x; ids[5] = v1.y; ids[6] = v1.z; ids[7] = v1.w;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) ids[c] = -1;
      }
      if (!TRAIN && valid && a.idx_out[L - 1 - s]) {
        long long* o = a.idx_out[L - 1 - s] + p * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (long long)ids[c];
      }
      if (slot >= 0) {
        Axis X = axis_weight<POLY>(x0, Lv.res, Lv.dres), Y = axis_weight<POLY>(x1, Lv.res, Lv.dres),
             Z = axis_weight<POLY>(x2, Lv.res, Lv.dres);
        float w[8];
        corner_weights(X.t, Y.t, Z.t, w);
        float dw[8][3];
        if (EIK) corner_weight_grads(X, Y, Z, dw);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float4* row = reinterpret_cast<const float4*>(Lv.feat + (long long)ids[c] * F);
          float4 r0 = row[0], r1 = row[1];
          float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
          for (int q = 0; q < F; ++q) {
            f[q] = fmaf(w[c], r[q], f[q]);
            if (EIK) {
              A[q][0] = fmaf(dw[c][0], r[q], A[q][0]);
              A[q][1] = fmaf(dw[c][1], r[q], A[q][1]);
              A[q][2] = fmaf(dw[c][2], r[q], A[q][2]);
            }
          }
        }
      }
    }
    if (!TRAIN && valid && a.feat_out) {
      float4* o = reinterpret_cast<float4*>(a.feat_out + p * F);
      o[0] = make_float4(f[0], f[1], f[2], f[3]);
      o[1] = make_float4(f[4], f[5], f[6], f[7]);
    }

    // ------------------------------------------------------------------ decoder forward
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
    float y = s_mlp[MLP_B3];
#pragma unroll
    for (int j = 0; j < H; ++j) y = fmaf(s_mlp[MLP_W3 + j], h2[j], y);
    if (valid && a.pred) a.pred[p] = y;

    // ------------------------------------------------------------------ d pred / d coord (closed form)
    float v1[H];  // m1 .* (W2^T (m2 .* w3))
    float J[F];   // d y / d f
    float g[3] = {0.f, 0.f, 0.f};
    if (EIK) {
#pragma unroll
      for (int k = 0; k < H; ++k) v1[k] = 0.f;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        float v2 = h2[j] > 0.f ? s_mlp[MLP_W3 + j] : 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) v1[k] = fmaf(s_mlp[MLP_W2 + j * H + k], v2, v1[k]);
      }
#pragma unroll
      for (int q = 0; q < F; ++q) J[q] = 0.f;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        v1[k] = h1[k] > 0.f ? v1[k] : 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) J[q] = fmaf(s_mlp[MLP_W1 + k * F + q], v1[k], J[q]);
      }
#pragma unroll
      for (int q = 0; q < F; ++q) {
        g[0] = fmaf(J[q], A[q][0], g[0]);
        g[1] = fmaf(J[q], A[q][1], g[1]);
        g[2] = fmaf(J[q], A[q][2], g[2]);
      }
      g[0] *= sigma;
      g[1] *= sigma;
      g[2] *= sigma;
      if (valid && a.grad_x) {
        a.grad_x[3 * p] = g[0];
        a.grad_x[3 * p + 1] = g[1];
        a.grad_x[3 * p + 2] = g[2];
      }
    }

    if (TRAIN) {
      // ---------------------------------------------------------------- loss
      float delta = 0.f;
      float qv[3] = {0.f, 0.f, 0.f};
      if (valid) {
        float zt = sigmoidf_acc(a.label[p] / sigma);
        const float lw = a.weighted ? fabsf(a.weight[p]) : 1.0f;  // BCEWithLogitsLoss(weight=|weight|), utils/loss.py:18-19
        float li = lw * (fmaxf(y, 0.f) - y * zt + log1pf(expf(-fabsf(y))));
        loss_acc += (double)li;
        cnt_acc += 1.0;
        delta = lw * (sigmoidf_acc(y) - zt) * a.inv_n;
        if (EIK && a.weight[p] > 0.f) {
          float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
          float e = 1.0f - gn;
          eik_acc += (double)(e * e);
          float coef = gn > 0.f ? (-2.0f * e / gn) * (a.weight_e * inv_nsurf) : 0.f;
          qv[0] = coef * g[0];
          qv[1] = coef * g[1];
          qv[2] = coef * g[2];
        }
      }
      // ---------------------------------------------------------------- backward through the decoder
      float d1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) d1[k] = 0.f;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        float d2 = h2[j] > 0.f ? delta * s_mlp[MLP_W3 + j] : 0.f;
        if (a.decoder_grad_on) {
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
      // eikonal second-order pieces
      float r[F];
      float a1[H];
      if (EIK) {
#pragma unroll
        for (int q = 0; q < F; ++q) r[q] = sigma * (A[q][0] * qv[0] + A[q][1] * qv[1] + A[q][2] * qv[2]);
#pragma unroll
        for (int k = 0; k < H; ++k) {
          float t = 0.f;
#pragma unroll
          for (int q = 0; q < F; ++q) t = fmaf(s_mlp[MLP_W1 + k * F + q], r[q], t);
          a1[k] = h1[k] > 0.f ? t : 0.f;
        }
      }

      if (a.decoder_grad_on) {
        const int jj = lane & 31;
        // phase A1: dW2 += d2 (x) h1 ; db2 += d2
        __syncthreads();
        contract64<16>(st, jj, 32 + (lane >> 5) * 16, accW2, accb2, true);
        __syncthreads();
        if (EIK) {
          // phase A2: dW2 += v2 (x) a1
#pragma unroll
          for (int j = 0; j < H; ++j) {
            st[lane * ST + j] = h2[j] > 0.f ? s_mlp[MLP_W3 + j] : 0.f;
            st[lane * ST + 32 + j] = a1[j];
          }
          __syncthreads();
          float dummy = 0.f;
          contract64<16>(st, jj, 32 + (lane >> 5) * 16, accW2, dummy, false);
          __syncthreads();
        }
        // phase B1: dW1 += d1 (x) f ; db1 += d1 ; dw3 += delta*h2 (+ a2) ; db3 += delta
#pragma unroll
        for (int k = 0; k < H; ++k) st[lane * ST + k] = d1[k];
#pragma unroll
        for (int q = 0; q < F; ++q) st[lane * ST + 32 + q] = f[q];
#pragma unroll
        for (int j = 0; j < H; ++j) {
          float g3 = delta * h2[j];
          if (EIK) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < H; ++k) t = fmaf(s_mlp[MLP_W2 + j * H + k], a1[k], t);
            g3 += h2[j] > 0.f ? t : 0.f;
          }
          st[lane * ST + 40 + j] = g3;
        }
        st[lane * ST + 72] = delta;
        __syncthreads();
        contract64<4>(st, jj, 32 + (lane >> 5) * 4, accW1, accb1, true);
        for (int pp = 0; pp < 64; ++pp) {
          accw3 += st[pp * ST + 40 + jj];
          accb3 += st[pp * ST + 72];
        }
        __syncthreads();
        if (EIK) {
          // phase B2: dW1 += v1 (x) r
#pragma unroll
          for (int k = 0; k < H; ++k) st[lane * ST + k] = v1[k];
#pragma unroll
          for (int q = 0; q < F; ++q) st[lane * ST + 32 + q] = r[q];
          __syncthreads();
          float dummy = 0.f;
          contract64<4>(st, jj, 32 + (lane >> 5) * 4, accW1, dummy, false);
          __syncthreads();
        }
      }

      // ---------------------------------------------------------------- scatter into the feature grads
      for (int s = 0; s < L; ++s) {
        const LevelDev& Lv = a.ls.lv[s];
        if (!Lv.grad) continue;
        int slot = -1;
        if (valid) {
          unsigned long long key = morton3(quantize(x0, Lv.res), quantize(x1, Lv.res), quantize(x2, Lv.res));
          slot = probe(Lv, key);
        }
        Axis X = axis_weight<POLY>(x0, Lv.res, Lv.dres), Y = axis_weight<POLY>(x1, Lv.res, Lv.dres),
             Z = axis_weight<POLY>(x2, Lv.res, Lv.dres);
        float w[8];
        corner_weights(X.t, Y.t, Z.t, w);
        float cq[8];  // sigma * (d w_c / d x . q)
        if (EIK) {
          float dw[8][3];
          corner_weight_grads(X, Y, Z, dw);
#pragma unroll
          for (int c = 0; c < 8; ++c) cq[c] = sigma * (dw[c][0] * qv[0] + dw[c][1] * qv[1] + dw[c][2] * qv[2]);
        }
        float trash[F];
#pragma unroll
        for (int q = 0; q < F; ++q) trash[q] = 0.f;
        if (slot >= 0) {
          int4 v0 = Lv.vals[2 * slot], v1i = Lv.vals[2 * slot + 1];
          int ids[8] = {v0.x, v0.y, v0.z, v0.w, v1i.x, v1i.y, v1i.z, v1i.w};
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float* dst = Lv.grad + (long long)ids[c] * F;
#pragma unroll
            for (int q = 0; q < F; ++q) {
              float v = w[c] * df[q];
              if (EIK) v = fmaf(cq[c], J[q], v);
              atomic_add_f32(dst + q, v);
            }
          }
        } else if (valid) {
          // all eight corners of a miss address the trash row (:205,231): it receives sum_c of the same terms
          float sw = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
          float sc = 0.f;
          if (EIK) sc = ((cq[0] + cq[1]) + (cq[2] + cq[3])) + ((cq[4] + cq[5]) + (cq[6] + cq[7]));
#pragma unroll
          for (int q = 0; q < F; ++q) {
            trash[q] = sw * df[q];
            if (EIK) trash[q] = fmaf(sc, J[q], trash[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < F; ++q) {
          float tsum = wave_sum(trash[q]);
          if (lane == 0 && tsum != 0.f) atomic_add_f32(Lv.grad + Lv.rows * F + q, tsum);
        }
      }
    }
  }

  if (TRAIN) {
    if (a.decoder_grad_on) {
      const int jj = lane & 31, hi = lane >> 5;
#pragma unroll
      for (int q = 0; q < 16; ++q) atomic_add_f32(a.grad_mlp[2] + jj * H + hi * 16 + q, accW2[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q) atomic_add_f32(a.grad_mlp[0] + jj * F + hi * 4 + q, accW1[q]);
      if (hi == 0) {
        atomic_add_f32(a.grad_mlp[1] + jj, accb1);
        atomic_add_f32(a.grad_mlp[3] + jj, accb2);
        atomic_add_f32(a.grad_mlp[4] + jj, accw3);
        if (lane == 0) atomic_add_f32(a.grad_mlp[5], accb3);
      }
    }
    double ls = wave_sum_d(loss_acc), es = wave_sum_d(eik_acc), cs = wave_sum_d(cnt_acc);
    if (lane == 0 && a.loss_parts) {
      const double bce = a.reduction_sum ? ls : ls * (double)a.inv_n;
      const double eik = EIK ? es * (double)inv_nsurf : 0.0;
      atomicAdd(a.loss_parts + 0, bce);
      if (EIK) atomicAdd(a.loss_parts + 1, eik);
      atomicAdd(a.loss_parts + 2, cs);
      atomicAdd(a.loss_parts + 3, bce + (double)a.weight_e * eik);
    }
  }
}

static unsigned grid_for(long long n) {
  long long tiles = (n + 255) / 256;
  long long cap = 256 * 4;  // persistent: 4 blocks per CU
  return (unsigned)(tiles < cap ? (tiles < 1 ? 1 : tiles) : cap);
}

template <bool EIK, bool TRAIN>
static void launch_v0(const StepArgs& a, bool poly, hipStream_t st) {
  dim3 g(grid_for(a.n)), b(256);
  if (poly)
    hipLaunchKernelGGL((k_step_v0<true, EIK, TRAIN>), g, b, 0, st, a);
  else
    hipLaunchKernelGGL((k_step_v0<false, EIK, TRAIN>), g, b, 0, st, a);
}

}  // namespace shine

using namespace shine;

// This file is part of the CHECK library only (libshine_check.so, tests / tools): the lane-per-point kernel as the on-device
// cross-check of the fused step (StepOptions.kernel_variant = 1).  The product library's forward entry points live in
// shine_forward.hip.
extern "C" int shine_train_step_v0(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                   const float* sdf_label, const float* weight, const int32_t* perm,
                                   const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows,
                                   const float* const* mlp, float* pred_out, float* grad_x_out,
                                   float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                   void* stream) {
  if (n < 0 || !feats || !rows || !mlp || !grad_feats || (n > 0 && (!coord || !sdf_label)))
    return set_error(SHINE_E_INVALID, "shine_train_step: null argument");
  if (cfg && cfg->eikonal_on && (!weight || !n_surf))
    return set_error(SHINE_E_INVALID, "shine_train_step: eikonal needs weight and n_surf");
  StepArgs a = {};
  int rc = make_level_set(t, cfg, feats, rows, grad_feats, &a.ls);
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < cfg->n_levels; ++s)
    if (!feats[s]) return set_error(SHINE_E_INVALID, "shine_train_step: null feature level");
  for (int k = 0; k < 6; ++k) {
    if (!mlp[k]) return set_error(SHINE_E_INVALID, "shine_train_step: null decoder parameter");
    a.mlp[k] = mlp[k];
    if (cfg->decoder_grad_on) {
      if (!grad_mlp || !grad_mlp[k]) return set_error(SHINE_E_INVALID, "shine_train_step: null decoder grad");
      a.grad_mlp[k] = grad_mlp[k];
    }
  }
  if (n == 0) return SHINE_OK;
  a.coord = coord;
  a.label = sdf_label;
  a.weight = weight;
  a.perm = perm;
  a.n_surf = reinterpret_cast<const long long*>(n_surf);
  a.n = n;
  a.n_levels = cfg->n_levels;
  a.reduction_sum = cfg->reduction_sum;
  a.weighted = cfg->loss_weight_on ? 1 : 0;
  a.decoder_grad_on = cfg->decoder_grad_on;
  a.sigma = cfg->sigma;
  a.weight_e = cfg->weight_e;
  a.inv_n = (float)cfg->inv_n;
  a.pred = pred_out;
  a.grad_x = grad_x_out;
  a.loss_parts = loss_parts;
  if (loss_parts) SHINE_HIP_CHECK(hipMemsetAsync(loss_parts, 0, 4 * sizeof(double), (hipStream_t)stream));
  for (int s = 0; s < cfg->n_levels; ++s)  // FeatureOctree.set_zero (:78-81): the kernel never reads the trash row
    SHINE_HIP_CHECK(hipMemsetAsync(const_cast<float*>(feats[s]) + rows[s] * F, 0, F * sizeof(float),
                                   (hipStream_t)stream));
  if (cfg->eikonal_on)
    launch_v0<true, true>(a, cfg->poly_int_on != 0, (hipStream_t)stream);
  else
    launch_v0<false, true>(a, cfg->poly_int_on != 0, (hipStream_t)stream);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
