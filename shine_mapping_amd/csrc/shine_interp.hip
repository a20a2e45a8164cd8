// shine_interp.hip — Tier A (strict drop-in): FeatureOctree.query_feature as a twice-differentiable autograd op.
//
//   forward            feat[p]   = sum_l sum_c w_{l,c}(x_p) F_l[id_{l,c}]              model/feature_octree.py:222-234
//   backward           dF_l[id] += w g_p ;  dx_p[a] = sum_{l,c} dw_{l,c}/dx_a (F_l[id].g_p)   (what autograd derives)
//   backward-backward  (for get_gradient(create_graph=True) + eikonal, utils/tools.py:175-185, shine_batch.py:182-185)
//                      given gg = d loss / d(dx):  dg_p += sum_{l,c} (dw.gg) F_l[id] ;  dF_l[id] += (dw.gg) g_p
//                      (the second derivative wrt x itself is not produced: coord is a leaf whose .grad nobody reads)
// Forward is shine_forward(feat_out=...).  These two kernels are lane = point (unordered batches: no node runs to
// merge), the feature-grad atomics are issued with the lanes transposed to (point, feature); the throughput tier is the
// fused step.
#include "shine_internal.hpp"

namespace shine {

struct InterpArgs {
  LevelSet ls;
  const float* coord;
  const float* g;       // [N,8] upstream gradient wrt feat
  const float* gg;      // [N,3] (backward-backward only)
  float* out_x;         // [N,3] backward: d/dcoord ; or null
  float* out_g;         // [N,8] backward-backward: d/dg ; or null
  long long n;
  int n_levels;
};

template <bool POLY, bool SECOND>
__global__ __launch_bounds__(256) void k_interp_bwd(InterpArgs a) {
  __shared__ float s_g[4 * 64 * F];
  const int lane = threadIdx.x & 63;
  float* const gt = s_g + (threadIdx.x >> 6) * 64 * F;
  const int sub = lane >> 3, fi = lane & 7;
  for (long long base = (long long)blockIdx.x * 256; base < a.n; base += (long long)gridDim.x * 256) {
    const bool valid = base + threadIdx.x < a.n;  // the loop itself stays wave-uniform: the trash-row sums are wave-wide
    const long long p = valid ? base + threadIdx.x : a.n - 1;
    const float x0 = a.coord[3 * p], x1 = a.coord[3 * p + 1], x2 = a.coord[3 * p + 2];
    float g[F];
    {
      const float4* gp = reinterpret_cast<const float4*>(a.g + p * F);
      float4 g0 = gp[0], g1 = gp[1];
      g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
    }
    // Feature-grad scatter with the lanes TRANSPOSED: one atomic instruction covers 8 points x the 8 features of one
    // corner row each (8 x 32 contiguous bytes) instead of 64 different rows x 4 bytes — an eighth of the memory-side
    // transactions.  g goes through LDS once per tile ([point][feature] -> lane (point & 7 group, feature)), the corner
    // ids and coefficients of the source point come by shuffle.
    float gval[8];
    wave_lds_fence();  // the previous tile's reads are done
    reinterpret_cast<float4*>(gt + lane * F)[0] = make_float4(g[0], g[1], g[2], g[3]);
    reinterpret_cast<float4*>(gt + lane * F)[1] = make_float4(g[4], g[5], g[6], g[7]);
    wave_lds_fence();
#pragma unroll
    for (int G = 0; G < 8; ++G) gval[G] = gt[G * 64 + lane];
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    if (SECOND) {
      q0 = a.gg[3 * p];
      q1 = a.gg[3 * p + 1];
      q2 = a.gg[3 * p + 2];
    }
    float ox[3] = {0.f, 0.f, 0.f};
    float og[F];
#pragma unroll
    for (int i = 0; i < F; ++i) og[i] = 0.f;
    for (int s = 0; s < a.n_levels; ++s) {
      const LevelDev& Lv = a.ls.lv[s];
      unsigned long long key = morton3(quantize(x0, Lv.res), quantize(x1, Lv.res), quantize(x2, Lv.res));
      const int slot = probe(Lv, key);
      Axis X = axis_weight<POLY>(x0, Lv.res, Lv.dres), Y = axis_weight<POLY>(x1, Lv.res, Lv.dres),
           Z = axis_weight<POLY>(x2, Lv.res, Lv.dres);
      float w[8], dw[8][3];
      corner_weights(X.t, Y.t, Z.t, w);
      corner_weight_grads(X, Y, Z, dw);
      int ids[8];
      if (slot >= 0) {
        int4 v0 = Lv.vals[2 * slot], v1 = Lv.vals[2 * slot + 1];
        ids[0] = v0.x; ids[1] = v0.y; ids[2] = v0.z; ids[3] = v0.w;
        ids[4] = v1.x; ids[5] = v1.y; ids[6] = v1.z; ids[7] = v1.w;
      }
      float csum = 0.f;  // a miss: all eight corners address the trash row (index -1, :205,231), which receives sum_c
      float coefs[8];
      int sids[8];  // scatter targets: -1 = nothing (a miss goes to the trash row below, a padding lane nowhere)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float coef = SECOND ? (dw[c][0] * q0 + dw[c][1] * q1 + dw[c][2] * q2) : w[c];
        coefs[c] = coef;
        csum += coef;
        sids[c] = (slot >= 0 && valid) ? ids[c] : -1;
        if (slot >= 0 && (SECOND ? a.out_g != nullptr : a.out_x != nullptr)) {
          const float4* rp = reinterpret_cast<const float4*>(Lv.feat + (long long)ids[c] * F);
          float4 r0 = rp[0], r1 = rp[1];
          const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          if (SECOND) {
#pragma unroll
            for (int i = 0; i < F; ++i) og[i] = fmaf(coef, r[i], og[i]);
          } else {
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < F; ++i) dot = fmaf(r[i], g[i], dot);
            ox[0] = fmaf(dw[c][0], dot, ox[0]);
            ox[1] = fmaf(dw[c][1], dot, ox[1]);
            ox[2] = fmaf(dw[c][2], dot, ox[2]);
          }
        }
      }
      if (Lv.grad) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
          for (int G = 0; G < 8; ++G) {
            const int id = __shfl(sids[c], G * 8 + sub, 64);
            const float cf = __shfl(coefs[c], G * 8 + sub, 64);
            if (id >= 0) atomic_add_f32(Lv.grad + (long long)id * F + fi, cf * gval[G]);
          }
        }
      }
      // one atomic per wave and feature for the trash row instead of 64 per missing point on the same 32 bytes (free-space
      // samples outside the mapped shell miss at every level: same-address atomics serialise at the memory controller)
      if (Lv.grad && __any(valid && slot < 0)) {
        const float cm = (valid && slot < 0) ? csum : 0.f;
#pragma unroll
        for (int i = 0; i < F; ++i) {
          const float tsum = wave_sum(cm * g[i]);
          if (lane == 0 && tsum != 0.f) atomic_add_f32(Lv.grad + Lv.rows * F + i, tsum);
        }
      }
    }
    if (!valid) continue;
    if (!SECOND && a.out_x) {
      a.out_x[3 * p] = ox[0];
      a.out_x[3 * p + 1] = ox[1];
      a.out_x[3 * p + 2] = ox[2];
    }
    if (SECOND && a.out_g) {
      float4* o = reinterpret_cast<float4*>(a.out_g + p * F);
      o[0] = make_float4(og[0], og[1], og[2], og[3]);
      o[1] = make_float4(og[4], og[5], og[6], og[7]);
    }
  }
}

static int launch_interp(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                         const float* const* feats, const int64_t* rows, const float* g, const float* gg, float* out_x,
                         float* out_g, float* const* grad_feats, bool second, void* stream) {
  if (n < 0 || !feats || !rows || (n > 0 && (!coord || !g)) || (second && n > 0 && !gg))
    return set_error(SHINE_E_INVALID, "shine_interp_backward: null argument");
  InterpArgs a = {};
  int rc = make_level_set(t, cfg, feats, rows, grad_feats, &a.ls);
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < cfg->n_levels; ++s)
    if (!feats[s]) return set_error(SHINE_E_INVALID, "shine_interp_backward: null feature level");
  if (n == 0) return SHINE_OK;
  a.coord = coord;
  a.g = g;
  a.gg = gg;
  a.out_x = out_x;
  a.out_g = out_g;
  a.n = n;
  a.n_levels = cfg->n_levels;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (cfg->poly_int_on) {
    if (second) hipLaunchKernelGGL((k_interp_bwd<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_interp_bwd<true, false>), grid, block, 0, st, a);
  } else {
    if (second) hipLaunchKernelGGL((k_interp_bwd<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_interp_bwd<false, false>), grid, block, 0, st, a);
  }
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

}  // namespace shine

using namespace shine;

extern "C" int shine_interp_backward(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                                     const float* const* feats, const int64_t* rows, const float* grad_feat,
                                     float* grad_coord_out, float* const* grad_feats, void* stream) {
  return launch_interp(t, cfg, coord, n, feats, rows, grad_feat, nullptr, grad_coord_out, nullptr, grad_feats, false,
                       stream);
}

extern "C" int shine_interp_backward_backward(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                              int64_t n, const float* const* feats, const int64_t* rows,
                                              const float* grad_feat, const float* gg_coord, float* grad_gfeat_out,
                                              float* const* grad_feats, void* stream) {
  return launch_interp(t, cfg, coord, n, feats, rows, grad_feat, gg_coord, nullptr, grad_gfeat_out, grad_feats, true,
                       stream);
}
