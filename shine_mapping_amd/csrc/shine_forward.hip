// shine_forward.hip — the strict drop-in tier's forward pass and its one-launch loss (Tier A):
//
//   shine_forward        FeatureOctree.query_feature (model/feature_octree.py:237-244: set_zero, get_indices, interpolation)
//                        [+ Decoder.sdf (model/decoder.py:49-63)] [+ get_gradient(coord, pred) * sigma (utils/tools.py:175-185)]
//   shine_query_indices  FeatureOctree.get_indices (:199-218)
//   shine_bce_loss       sdf_bce_loss (utils/loss.py:17-24) with its derivative in the same launch
//
// Unordered batches as the drivers hand them over (LiDARDataset.get_batch: torch.randint draws), so lane = point: quantise,
// Morton code, hash probe and the eight 32-B row gathers per level, all levels' probes in flight before the first gather.  The
// decoder runs on the VALU with its 1377 weights read through the constant address space (scalar loads, SGPR operands: no VGPR
// and no LDS read per weight, shine_device.hpp uniform_ro) and layer 2 as a ROLLED loop over its 32 weight rows — the round-1
// kernel behind this entry point (shine_step_v0.hip, lane = point with the weights broadcast from LDS) needed 512 VGPRs + 256
// AGPRs and still spilled 68 registers in its d pred / d coord form; this one stays below 128.
//   d pred / d coord in closed form: v2 = m2 .* w3 rides on layer 2's loop as t = W2^T v2 (the same scalar weight
//   row feeds z_j and t), v1 = m1 .* t, J = W1^T v1 = d y / d f, A = d f / d x is accumulated with the gathers, g = sigma J^T A.
#include "shine_internal.hpp"

namespace shine {

struct FwdArgs {
  LevelSet ls;
  const float* coord;
  const float* mlp[6];
  float* feat_out;                       // [n, 8] or null
  float* pred;                           // [n] or null
  float* grad_x;                         // [n, 3] = sigma * d pred / d coord, or null
  long long* idx_out[SHINE_MAX_LEVELS];  // bottom-up [n, 8] int64 each, or null
  long long n;
  int n_levels;
  float sigma;
  // cfg->reg_rider (include/shine_hip.h shine_reg_rider): cal_regularization's value on the rows this batch addresses
  const float* reg_last[SHINE_MAX_LEVELS];
  const float* reg_imp[SHINE_MAX_LEVELS];
  unsigned int* reg_stamp[SHINE_MAX_LEVELS];
  float* reg_acc;  // null: off
  unsigned int reg_epoch;
};

template <int LMAX, bool POLY, bool GRADX>
__global__ __launch_bounds__(256) void k_forward_points(FwdArgs a) {
  const int L = a.n_levels;
  // FeatureOctree.set_zero (model/feature_octree.py:78-81, at the top of query_feature): nothing in this kernel reads the trash
  // row (a miss contributes nothing), so there is no ordering to keep
  if (blockIdx.x == 0 && (int)threadIdx.x < L * F)
    const_cast<float*>(a.ls.lv[threadIdx.x / F].feat)[a.ls.lv[threadIdx.x / F].rows * F + (threadIdx.x % F)] = 0.f;
  const bool with_mlp = a.pred != nullptr || GRADX;  // wave-uniform
  cfloat *W1 = nullptr, *B1 = nullptr, *W2 = nullptr, *B2 = nullptr, *W3 = nullptr, *B3 = nullptr;
  if (with_mlp) {
    W1 = uniform_ro(a.mlp[0]), B1 = uniform_ro(a.mlp[1]), W2 = uniform_ro(a.mlp[2]), B2 = uniform_ro(a.mlp[3]);
    W3 = uniform_ro(a.mlp[4]), B3 = uniform_ro(a.mlp[5]);
  }
  const int rows = opaque(H);
  const long long stride = (long long)gridDim.x * 256;
  float reg_sum = 0.f;  // (rider) this thread's share of the regulariser
  if (a.reg_acc && blockIdx.x == 0 && threadIdx.x == 0) a.reg_acc[(a.reg_epoch + 1u) & 7u] = 0.f;  // the NEXT launch's accumulator
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < a.n; p += stride) {
    const float x0 = a.coord[3 * p], x1 = a.coord[3 * p + 1], x2 = a.coord[3 * p + 2];
    int slot[LMAX];
#pragma unroll
    for (int s = 0; s < LMAX; ++s) {
      slot[s] = -1;
      if (s < L) {
        const LevelDev& Lv = a.ls.lv[s];
        slot[s] = probe(Lv, morton3(quantize(x0, Lv.res), quantize(x1, Lv.res), quantize(x2, Lv.res)));
      }
    }
    float f[F];
    float A[GRADX ? F : 1][3];
#pragma unroll
    for (int q = 0; q < F; ++q) f[q] = 0.f;
#pragma unroll
    for (int q = 0; q < (GRADX ? F : 1); ++q) A[q][0] = A[q][1] = A[q][2] = 0.f;
#pragma unroll
    for (int s = 0; s < LMAX; ++s) {
      if (s >= L) continue;
      const LevelDev& Lv = a.ls.lv[s];
      const bool hit = slot[s] >= 0;
      const unsigned int sl = hit ? (unsigned int)slot[s] : 0u;
      int4 i0 = Lv.vals[2u * sl], i1 = Lv.vals[2u * sl + 1u];
      if (!hit) i0 = i1 = make_int4(-1, -1, -1, -1);  // python dict.get(key, [-1] * 8), :209
      if (a.idx_out[L - 1 - s]) {
        long long* o = a.idx_out[L - 1 - s] + p * 8;
        o[0] = i0.x, o[1] = i0.y, o[2] = i0.z, o[3] = i0.w, o[4] = i1.x, o[5] = i1.y, o[6] = i1.z, o[7] = i1.w;
      }
      const int ids[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
      Axis X = axis_weight<POLY>(x0, Lv.res, Lv.dres), Y = axis_weight<POLY>(x1, Lv.res, Lv.dres),
           Z = axis_weight<POLY>(x2, Lv.res, Lv.dres);
      float w[8];
      corner_weights(X.t, Y.t, Z.t, w);
      float4 r0[8], r1[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {  // a miss reads row 0 with weight 0 (no branches)
        const unsigned int off = (hit ? (unsigned int)ids[c] : 0u) * (unsigned int)F;
        r0[c] = *reinterpret_cast<const float4*>(Lv.feat + off);
        r1[c] = *reinterpret_cast<const float4*>(Lv.feat + off + 4u);
      }
      float dw[GRADX ? 8 : 1][3];
      if (GRADX) {
        if (!hit) X.dt = Y.dt = Z.dt = 0.f;  // every d w_c / d x carries exactly one of these factors
        float full[8][3];
        corner_weight_grads(X, Y, Z, full);
#pragma unroll
        for (int c = 0; c < 8; ++c) dw[c][0] = full[c][0], dw[c][1] = full[c][1], dw[c][2] = full[c][2];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float wz = hit ? w[c] : 0.f;
        const float rr[F] = {r0[c].x, r0[c].y, r0[c].z, r0[c].w, r1[c].x, r1[c].y, r1[c].z, r1[c].w};
#pragma unroll
        for (int q = 0; q < F; ++q) {
          f[q] = fmaf(wz, rr[q], f[q]);
          if (GRADX) {
            A[q][0] = fmaf(dw[c][0], rr[q], A[q][0]);
            A[q][1] = fmaf(dw[c][1], rr[q], A[q][1]);
            A[q][2] = fmaf(dw[c][2], rr[q], A[q][2]);
          }
        }
      }
    }
    if (a.feat_out) {
      float4* o = reinterpret_cast<float4*>(a.feat_out + p * F);
      o[0] = make_float4(f[0], f[1], f[2], f[3]);
      o[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (a.reg_acc) {  // (wave-uniform) the regulariser rides: every addressed row counts once per launch — whoever swaps its stamp
      // to this launch's epoch first adds the row's term (unique(hierarchical_indices) without -1, feature_octree.py:250-254)
      for (int s = 0; s < L; ++s) {
        if (slot[s] < 0) continue;
        const LevelDev& Lv = a.ls.lv[s];
        const int4 i0 = Lv.vals[2u * (unsigned int)slot[s]], i1 = Lv.vals[2u * (unsigned int)slot[s] + 1u];
        const int ids[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
        for (int c = 0; c < 8; ++c) {
          const unsigned int row = (unsigned int)ids[c];
          if (atomicExch(a.reg_stamp[s] + row, a.reg_epoch) == a.reg_epoch) continue;
          const float4* fr = reinterpret_cast<const float4*>(Lv.feat + row * (unsigned int)F);
          const float4* lr = reinterpret_cast<const float4*>(a.reg_last[s] + row * (unsigned int)F);
          const float4* ir = reinterpret_cast<const float4*>(a.reg_imp[s] + row * (unsigned int)F);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 fv = fr[h], lv = lr[h], iv = ir[h];
            const float d0 = fv.x - lv.x, d1 = fv.y - lv.y, d2 = fv.z - lv.z, d3 = fv.w - lv.w;
            reg_sum += (iv.x * (d0 * d0) + iv.y * (d1 * d1)) + (iv.z * (d2 * d2) + iv.w * (d3 * d3));
          }
        }
      }
    }
    if (!with_mlp) continue;
    float h1[H];
    {
      cfloat *const W1i = relaunder(W1), *const B1i = relaunder(B1);  // keep the loads inside this iteration
#pragma unroll
      for (int j = 0; j < H; ++j) {
        float z = B1i[j];
#pragma unroll
        for (int q = 0; q < F; ++q) z = fmaf(W1i[j * F + q], f[q], z);
        h1[j] = fmaxf(z, 0.f);
      }
    }
    float y = B3[0];
    float t[GRADX ? H : 1];  // W2^T (m2 .* w3): the decoder's backward of d y = 1 through layer 2
#pragma unroll
    for (int k = 0; k < (GRADX ? H : 1); ++k) t[k] = 0.f;
    // layer 2 + 3 as a ROLLED loop over the 32 weight rows (two s_load_dwordx16 each): h2 is consumed as produced
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(2)
    for (int j = 0; j < rows; ++j) {
      float z = B2[j];
#pragma unroll
      for (int k = 0; k < H; ++k) z = fmaf(W2[j * H + k], h1[k], z);
      const float w3 = W3[j];
      y = fmaf(w3, fmaxf(z, 0.f), y);
      if (GRADX) {
        const float v2 = z > 0.f ? w3 : 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) t[k] = fmaf(W2[j * H + k], v2, t[k]);
      }
    }
    if (a.pred) a.pred[p] = y;
    if (GRADX) {
      float J[F];
#pragma unroll
      for (int q = 0; q < F; ++q) J[q] = 0.f;
      cfloat* const W1j = relaunder(W1);
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float v1 = h1[k] > 0.f ? t[k] : 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) J[q] = fmaf(W1j[k * F + q], v1, J[q]);
      }
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
      for (int q = 0; q < F; ++q) {
        g0 = fmaf(J[q], A[q][0], g0);
        g1 = fmaf(J[q], A[q][1], g1);
        g2 = fmaf(J[q], A[q][2], g2);
      }
      a.grad_x[3 * p] = a.sigma * g0;
      a.grad_x[3 * p + 1] = a.sigma * g1;
      a.grad_x[3 * p + 2] = a.sigma * g2;
    }
  }
  if (a.reg_acc) {  // (every thread is here: the grid-stride loop has ended for all of them)
    const float r = wave_sum(reg_sum);
    if ((threadIdx.x & 63) == 0 && r != 0.f) atomicAdd(a.reg_acc + (a.reg_epoch & 7u), r);
  }
}

static unsigned fwd_grid(long long n) {
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 workgroups per CU
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <int LMAX>
static void launch_forward(const FwdArgs& a, bool poly, hipStream_t st) {
  const dim3 g(fwd_grid(a.n)), b(256);
  if (a.grad_x) {
    if (poly) hipLaunchKernelGGL((k_forward_points<LMAX, true, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_forward_points<LMAX, false, true>), g, b, 0, st, a);
  } else {
    if (poly) hipLaunchKernelGGL((k_forward_points<LMAX, true, false>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_forward_points<LMAX, false, false>), g, b, 0, st, a);
  }
}

// FeatureOctree.get_indices alone (model/feature_octree.py:199-218)
__global__ __launch_bounds__(256) void k_query_indices(FwdArgs a) {
  const int L = a.n_levels;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < a.n; p += (long long)gridDim.x * 256) {
    const float x0 = a.coord[3 * p], x1 = a.coord[3 * p + 1], x2 = a.coord[3 * p + 2];
    for (int s = 0; s < L; ++s) {
      const LevelDev& Lv = a.ls.lv[s];
      const int slot = probe(Lv, morton3(quantize(x0, Lv.res), quantize(x1, Lv.res), quantize(x2, Lv.res)));
      int4 v0 = make_int4(-1, -1, -1, -1), v1 = v0;
      if (slot >= 0) {
        v0 = Lv.vals[2 * slot];
        v1 = Lv.vals[2 * slot + 1];
      }
      long long* o = a.idx_out[L - 1 - s] + p * 8;
      o[0] = v0.x, o[1] = v0.y, o[2] = v0.z, o[3] = v0.w, o[4] = v1.x, o[5] = v1.y, o[6] = v1.z, o[7] = v1.w;
    }
  }
}

// sdf_bce_loss (utils/loss.py:17-24): BCEWithLogits(pred, sigmoid(label / sigma)) [x weight], mean or sum, AND its derivative
// d loss / d pred in the same launch (what autograd would compute in three more).  One workgroup at the reference's batch sizes
// (a few thousand points: a grid-wide sum would cost a second launch); large Tier-A batches (2^16 .. 2^20 points) run MULTI
// workgroups, whose fp64 partial sums meet in one fp32 atomic each on the pre-zeroed result (<= 256 terms).
template <bool MULTI>
__global__ __launch_bounds__(1024) void k_bce_loss(const float* pred, const float* label, const float* weight, long long n,
                                                   float sigma, float scale, float* loss_out, float* dpred_out) {
  __shared__ double s_red[16];
  double acc = 0.0;
  const long long stride = MULTI ? (long long)gridDim.x * 1024 : 1024;
  for (long long i = (MULTI ? (long long)blockIdx.x * 1024 : 0) + threadIdx.x; i < n; i += stride) {
    const float y = pred[i];
    const float z = 1.0f / (1.0f + expf(-(label[i] / sigma)));  // torch.sigmoid(label / sigma), utils/loss.py:23
    const float w = weight ? weight[i] : 1.0f;
    // max(y, 0) - y z + log1p(exp(-|y|)): torch's stable form of BCEWithLogits
    acc += (double)(w * (fmaxf(y, 0.f) - y * z + log1pf(expf(-fabsf(y)))));
    if (dpred_out) dpred_out[i] = w * (1.0f / (1.0f + expf(-y)) - z) * scale;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int k = 0; k < 16; ++k) tot += s_red[k];
    if (MULTI) atomicAdd(loss_out, (float)(tot * (double)scale));
    else *loss_out = (float)(tot * (double)scale);
  }
}

}  // namespace shine

using namespace shine;

extern "C" int shine_query_indices(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                                   int64_t* const* idx_out, void* stream) {
  if (n < 0 || !idx_out || (n > 0 && !coord)) return set_error(SHINE_E_INVALID, "shine_query_indices: null argument");
  FwdArgs a = {};
  int rc = make_level_set(t, cfg, nullptr, nullptr, nullptr, &a.ls);
  if (rc != SHINE_OK) return rc;
  if (n == 0) return SHINE_OK;
  for (int i = 0; i < cfg->n_levels; ++i) {
    if (!idx_out[i]) return set_error(SHINE_E_INVALID, "shine_query_indices: null output level");
    a.idx_out[i] = (long long*)idx_out[i];
  }
  a.coord = coord;
  a.n = n;
  a.n_levels = cfg->n_levels;
  hipLaunchKernelGGL(k_query_indices, dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_forward(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                             const float* const* feats, const int64_t* rows, const float* const* mlp, float* feat_out,
                             float* pred_out, int64_t* const* idx_out, float* grad_x_out, void* stream) {
  if (n < 0 || !feats || !rows || (n > 0 && !coord)) return set_error(SHINE_E_INVALID, "shine_forward: null argument");
  if ((pred_out || grad_x_out) && !mlp) return set_error(SHINE_E_INVALID, "shine_forward: pred / grad_x need the decoder");
  FwdArgs a = {};
  int rc = make_level_set(t, cfg, feats, rows, nullptr, &a.ls);
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < cfg->n_levels; ++s) {
    if (!feats[s]) return set_error(SHINE_E_INVALID, "shine_forward: null feature level");
    if (rows[s] >= (1ll << 29)) return set_error(SHINE_E_INVALID, "shine_forward: level exceeds 2^29 rows");
    if ((size_t)feats[s] & 15) return set_error(SHINE_E_INVALID, "shine_forward: feature tables must be 16-byte aligned");
  }
  if (pred_out || grad_x_out)
    for (int k = 0; k < 6; ++k) {
      if (!mlp[k]) return set_error(SHINE_E_INVALID, "shine_forward: null decoder parameter");
      a.mlp[k] = mlp[k];
    }
  if (feat_out && ((size_t)feat_out & 15)) return set_error(SHINE_E_INVALID, "shine_forward: feat_out must be 16-byte aligned");
  if (n == 0) return SHINE_OK;
  a.coord = coord;
  a.n = n;
  a.n_levels = cfg->n_levels;
  a.sigma = cfg->sigma;
  a.pred = pred_out;
  a.feat_out = feat_out;
  a.grad_x = grad_x_out;
  if (idx_out)
    for (int i = 0; i < cfg->n_levels; ++i) a.idx_out[i] = (long long*)idx_out[i];
  if (const shine_reg_rider* rr = cfg->reg_rider) {
    if (!rr->acc || rr->epoch == 0) return set_error(SHINE_E_INVALID, "shine_forward: reg_rider needs acc and an epoch > 0");
    for (int s = 0; s < cfg->n_levels; ++s) {
      if (!rr->last[s] || !rr->imp[s] || !rr->stamp[s] || ((size_t)rr->last[s] & 15) || ((size_t)rr->imp[s] & 15))
        return set_error(SHINE_E_INVALID, "shine_forward: reg_rider needs 16-byte aligned last / imp tables and a stamp array per level");
      a.reg_last[s] = rr->last[s];
      a.reg_imp[s] = rr->imp[s];
      a.reg_stamp[s] = rr->stamp[s];
    }
    a.reg_acc = rr->acc;
    a.reg_epoch = rr->epoch;
  }
  const bool poly = cfg->poly_int_on != 0;
  if (cfg->n_levels <= 4) launch_forward<4>(a, poly, (hipStream_t)stream);
  else launch_forward<SHINE_MAX_LEVELS>(a, poly, (hipStream_t)stream);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_bce_loss(const float* pred, const float* sdf_label, const float* weight, int64_t n, float sigma,
                              int32_t reduction_sum, float* loss_out, float* dpred_out, void* stream) {
  if (n < 1 || !pred || !sdf_label || !loss_out || !(sigma > 0.f))
    return set_error(SHINE_E_INVALID, "shine_bce_loss: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const float scale = reduction_sum ? 1.0f : 1.0f / (float)n;
  if (n <= 16384) {
    hipLaunchKernelGGL(k_bce_loss<false>, dim3(1), dim3(1024), 0, st, pred, sdf_label, weight, (long long)n, sigma, scale, loss_out,
                       dpred_out);
  } else {  // (2^18 .. 2^20 points on ONE compute unit otherwise)
    SHINE_HIP_CHECK(hipMemsetAsync(loss_out, 0, sizeof(float), st));
    const long long blocks = (n + 4095) / 4096;
    hipLaunchKernelGGL(k_bce_loss<true>, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(1024), 0, st, pred, sdf_label, weight,
                       (long long)n, sigma, scale, loss_out, dpred_out);
  }
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
