// shine_finish.hip — the tail of a training iteration in ONE launch (shine_batch.py:208-210,
// shine_incre.py:152-181, model/feature_octree.py:246-255, utils/tools.py:57-83).
//
// At the reference's batch size (4096) an iteration is a chain of small launches, each costing its run time plus the
// dependency gap in front of it: sorted draw (7.6 us) -> fused step (14.9) -> reduction of the step's partial sums (4.8) ->
// regulariser (8.2) -> dense Adam (7.1), 52 us per iteration in a replayed HIP graph (profiles/r03_ab_experiments.txt block
// 11).  The last three walk the same tensors: with cfg->defer_reduce the fused step leaves its per-workgroup partial sums in
// the workspace, and this launch
//   * adds them up where they are consumed — the decoder's gradient elements and the trash rows' — and writes loss_parts,
//   * evaluates the regulariser (value and gradient) on the rows the step flagged, clearing the flags,
//   * applies torch's Adam to every tensor and clears the gradients (opt.step() + opt.zero_grad()),
//   * optionally draws the NEXT iteration's sorted batch in a few extra blocks (the one-launch sampler as a device function),
// one thread per feature row (8 floats), one wave per 8 elements whose gradient is a sum over the workgroups (trash rows,
// decoder): the lanes split the partial vectors.
#include "shine_finish_args.hpp"
#include "shine_sampler_dev.hpp"
#include "shine_tile16.hpp"

namespace shine {

struct FinScalars {
  float b1, b2, eps, bc1, bc2_sqrt;
};

__device__ __forceinline__ void fin_adam1(float& p, float g, float& m, float& v, const FinScalars& a, float lr, float wd) {
  const float gg = g + wd * p;
  m = a.b1 * m + (1.0f - a.b1) * gg;
  v = a.b2 * v + (1.0f - a.b2) * gg * gg;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (lr / a.bc1) * (m / denom);
}

// blocks [0, fb): feature rows, one thread per row; blocks [fb, fb + db): decoder, one wave per unit of 8 elements;
// block fb + db: the loss terms; blocks behind it: the sorted draw of the next iteration (one-launch form of the sampler)
__global__ __launch_bounds__(256) void k_finish(const FinArgs a, int fb, int db) {
  __shared__ float s_lr[FIN_MAX_SEG];
  __shared__ double s_red[4];
  __shared__ SampleShared s_sample;
  // Workgroups are dispatched in index order, so the next draw's blocks go FIRST: every one of them recomputes the block sums of
  // all spacings, the longest dependent chain of the launch — started last they were the last to finish, 1.5 us of a 10 us
  // launch (profiles/r04_ab_experiments.txt block 13).  vb = the index the code below is written in: rows [0, fb), decoder
  // units [fb, fb + db), loss fb + db, draw beyond.
  int vb = (int)blockIdx.x;
  vb = vb < a.nd_blocks ? fb + db + 1 + vb : vb - a.nd_blocks;
  if (vb > fb + db) {  // (nothing of this iteration reads the index buffer any more: the fused kernel is done)
    sample_fused_block(s_sample, vb - (fb + db + 1), a.nd_blocks, a.nd_n, a.nd_pool, a.nd_seed, 0ull, a.nd_stream,
                       a.nd_idx, a.nd_bits, a.nd_surf);
    return;
  }
  const float* bc = reinterpret_cast<const float*>(a.step_state + 1);
  const FinScalars sc = {a.b1, a.b2, a.eps, bc[0], bc[1]};
  if (threadIdx.x < a.n_seg) s_lr[threadIdx.x] = a.lr_dev[a.seg[threadIdx.x].lr_idx];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  if (vb < fb) {
    double acc = 0.0;
    // Rows are dealt round-robin to the launch's threads; a thread's (up to 4) flag bytes of one pass are requested together,
    // before any row is touched: on a map of 10^7 rows a thread walks ~10 rows, and flag -> row is a dependent round trip each.
    const long long stride = (long long)fb * 256;
    for (long long u0 = (long long)vb * 256 + threadIdx.x; u0 < a.feat_units; u0 += 4 * stride) {
      int sv[4];
      long long rv[4];
      unsigned char tv[4];
      bool live[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long u = u0 + k * stride;
        int s = 0;
        while (s + 1 < a.n_levels && u >= a.seg[s + 1].ustart) ++s;
        sv[k] = s;
        rv[k] = u - a.seg[s].ustart;
        // (the trash row, the last one, goes with the decoder units below: its gradient is a sum over the workgroups)
        live[k] = u < a.feat_units && rv[k] != a.seg[s].n / F - 1;
        tv[k] = (live[k] && a.touched[s]) ? a.touched[s][rv[k]] : (unsigned char)0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!live[k] || (a.active && tv[k] == 0)) continue;
        const int s = sv[k];
        const long long r = rv[k];
        const FinSeg& S = a.seg[s];
        float4* pp = reinterpret_cast<float4*>(S.p + r * F);
        float4* gp = reinterpret_cast<float4*>(S.g + r * F);
        float4* mp = reinterpret_cast<float4*>(S.m + r * F);
        float4* vp = reinterpret_cast<float4*>(S.v + r * F);
        const float4 p0 = pp[0], p1 = pp[1], g0 = gp[0], g1 = gp[1], m0 = mp[0], m1 = mp[1], v0 = vp[0], v1 = vp[1];
        float p[F] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        float g[F] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float m[F] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        float v[F] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (tv[k] & 1) {  // a row the step of THIS iteration touched
          a.touched[s][r] = a.active ? (unsigned char)2 : (unsigned char)0;
          if (a.lambda != 0.f) {
            // FeatureOctree.cal_regularization (:246-255) on a row the step touched: value, gradient
            const float4* lp = reinterpret_cast<const float4*>(a.last[s] + r * F);
            const float4* ip = reinterpret_cast<const float4*>(a.imp[s] + r * F);
            const float4 l0 = lp[0], l1 = lp[1], w0 = ip[0], w1 = ip[1];
            const float l[F] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            const float w[F] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const float kk = a.grad_on[s] ? 2.0f * a.lambda : 0.f;
#pragma unroll
            for (int q = 0; q < F; ++q) {
              const float d = p[q] - l[q];
              acc += (double)(w[q] * d * d);
              g[q] += kk * w[q] * d;
            }
          }
        }
        const float lr = s_lr[s];
#pragma unroll
        for (int q = 0; q < F; ++q) fin_adam1(p[q], g[q], m[q], v[q], sc, lr, S.wd);
        pp[0] = make_float4(p[0], p[1], p[2], p[3]);
        pp[1] = make_float4(p[4], p[5], p[6], p[7]);
        mp[0] = make_float4(m[0], m[1], m[2], m[3]);
        mp[1] = make_float4(m[4], m[5], m[6], m[7]);
        vp[0] = make_float4(v[0], v[1], v[2], v[3]);
        vp[1] = make_float4(v[4], v[5], v[6], v[7]);
        // most rows of a map were not touched by a 4096-point batch: their gradient is already zero and is not rewritten
        // (1/8 of this launch's traffic)
        if ((g0.x != 0.f) | (g0.y != 0.f) | (g0.z != 0.f) | (g0.w != 0.f) | (g1.x != 0.f) | (g1.y != 0.f) | (g1.z != 0.f) |
            (g1.w != 0.f))
          gp[0] = gp[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (a.lambda != 0.f) {
      acc = wave_sum_d(acc);
      if (lane == 0) s_red[threadIdx.x >> 6] = acc;
      __syncthreads();
      const double t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
      if (threadIdx.x == 0 && t != 0.0) atomicAdd(a.reg_out, t);
    }
    return;
  }
  if (vb < fb + db) {
    // wave = unit of 8 consecutive elements whose gradient is a sum over the workgroups' partial vectors: units [0, L) the
    // trash rows of the feature tables (every miss of the step lands there), then the decoder tensors, 8 elements each;
    // lane = (element q = lane & 7, share c = lane >> 3 of the partial vectors)
    const long long u = ((long long)vb - fb) * 4 + (threadIdx.x >> 6);
    if (u >= a.dec_units) return;
    const int q = lane & 7;
    int s, poff;
    long long e;
    if (u < a.n_levels) {
      s = (int)u;
      e = a.seg[s].n - F + q;
      poff = PART_TRASH + s * F - (int)(a.seg[s].n - F);  // (so that poff + e indexes the partial vector)
    } else {
      s = a.n_levels;
      const long long ug = a.feat_units + (u - a.n_levels);
      while (s + 1 < a.n_seg && ug >= a.seg[s + 1].ustart) ++s;
      e = (ug - a.seg[s].ustart) * F + q;
      poff = a.seg[s].part_off;
    }
    const FinSeg& S = a.seg[s];
    const bool live = e < S.n;
    float t = 0.f;
    if (live)
      for (int b = lane >> 3; b < a.nblocks; b += 8) t += a.partials[(long long)b * PART_STRIDE + poff + e];
    t += __shfl_xor(t, 8, 64);
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (live && lane < 8) {
      // FeatureOctree.set_zero (model/feature_octree.py:78-81) zeroes a trash row before the query, i.e. before this update
      float p = u < a.n_levels ? 0.f : S.p[e], m = S.m[e], v = S.v[e];
      const float g = S.g[e] + t;
      fin_adam1(p, g, m, v, sc, s_lr[s], S.wd);
      S.p[e] = p;
      S.m[e] = m;
      S.v[e] = v;
      S.g[e] = 0.f;
      if (a.op_image && u >= a.n_levels) {  // a decoder element: also where the fused kernel's operand image holds it
        int i0, i1;
        operand_image_slots(S.part_off, (int)e, i0, i1);
        if (i0 >= 0) a.op_image[i0] = p;
        if (i1 >= 0) a.op_image[i1] = p;
      }
    }
    return;
  }
  // the loss terms (as k_reduce_partials): doubles {bce sum, count, eikonal sum} of every workgroup, the surface count
  if (threadIdx.x < 64 && a.loss_parts) {
    double ls = 0.0, cs = 0.0, es = 0.0;
    for (int b = lane; b < a.nblocks; b += 64) {
      const double* d = reinterpret_cast<const double*>(a.partials + (long long)b * PART_STRIDE + PART_LOSS);
      ls += d[0];
      cs += d[1];
      es += d[2];
    }
    ls = wave_sum_d(ls);
    cs = wave_sum_d(cs);
    es = wave_sum_d(es);
    if (lane == 0) {
      // the surface count the fused kernel used (left by its workgroup 0 next to the loss terms): the sampler blocks of THIS
      // launch are already counting the next batch's surface samples into a.n_surf
      const long long ns = a.n_surf ? (long long)reinterpret_cast<const double*>(a.partials + PART_LOSS)[3] : 0;
      const double bce = a.reduction_sum ? ls : ls * (double)a.inv_n;
      const double eik = ns > 0 ? es * (double)(1.0f / (float)ns) : 0.0;
      a.loss_parts[0] = bce;
      a.loss_parts[1] = eik;
      a.loss_parts[2] = cs;
      a.loss_parts[3] = bce + (double)a.weight_e * eik;
    }
  }
}

}  // namespace shine

using namespace shine;

namespace shine {

int prepare_finish(FinLaunch* out, const shine_step_config* cfg, int64_t n, const void* workspace, const int64_t* n_surf,
                   double* loss_parts, const float* const* feats_last, const float* const* importance,
                   unsigned char* const* touched, const int32_t* grad_on, float lambda_forget, double* reg_out,
                   int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev, const int32_t* lr_index,
                   const float* weight_decay, float beta1, float beta2, float eps, const int64_t* step_state,
                   const shine_next_draw* next_draw, int32_t active_rows) {
  if (!cfg || n < 1 || !workspace || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr_dev || !lr_index ||
      !weight_decay || !step_state)
    return set_error(SHINE_E_INVALID, "shine_finish_iteration: null argument");
  const int L = cfg->n_levels;
  if (L < 1 || L > LCAP) return set_error(SHINE_E_INVALID, "shine_finish_iteration: 1..4 featured levels");
  if (n_tensors != L && n_tensors != L + 6)
    return set_error(SHINE_E_INVALID, "shine_finish_iteration: tensors = the L feature tables (top-down) [+ the 6 decoder tensors]");
  if ((cfg->kernel_variant & 0xff) != 0 && (cfg->kernel_variant & 0xff) != 4)
    return set_error(SHINE_E_INVALID, "shine_finish_iteration: follows the product kernel (kernel_variant 0)");
  if (cfg->eikonal_on && !n_surf) return set_error(SHINE_E_INVALID, "shine_finish_iteration: eikonal needs n_surf");
  if (lambda_forget != 0.f && (!feats_last || !importance || !touched || !reg_out))
    return set_error(SHINE_E_INVALID, "shine_finish_iteration: the regulariser needs feats_last, importance, touched, reg_out");
  if (active_rows && !touched)
    return set_error(SHINE_E_INVALID, "shine_finish_iteration: active_rows needs the touched-row flags of every feature level");
  FinArgs& a = out->a;
  a = FinArgs{};
  a.n_seg = n_tensors;
  a.n_levels = L;
  static const int dec_off[6] = {MLP_W1, MLP_B1, MLP_W2, MLP_B2, MLP_W3, MLP_B3};
  static const int dec_n[6] = {MLP_B1 - MLP_W1, MLP_W2 - MLP_B1, MLP_B2 - MLP_W2, MLP_W3 - MLP_B2, MLP_B3 - MLP_W3, 1};
  long long u = 0;
  for (int s = 0; s < n_tensors; ++s) {
    if (!params[s] || !grads[s] || !exp_avg[s] || !exp_avg_sq[s] || numel[s] < 0)
      return set_error(SHINE_E_INVALID, "shine_finish_iteration: null tensor");
    FinSeg& S = a.seg[s];
    S.p = params[s];
    S.g = grads[s];
    S.m = exp_avg[s];
    S.v = exp_avg_sq[s];
    S.n = numel[s];
    S.ustart = u;
    S.wd = weight_decay[s];
    S.lr_idx = lr_index[s];
    if (S.lr_idx < 0 || S.lr_idx >= FIN_MAX_SEG) return set_error(SHINE_E_INVALID, "shine_finish_iteration: bad lr_index");
    if (s < L) {
      if (numel[s] % F || numel[s] < F || (((size_t)S.p | (size_t)S.g | (size_t)S.m | (size_t)S.v) & 15))
        return set_error(SHINE_E_INVALID, "shine_finish_iteration: feature tables are [rows + 1][8] floats, 16-byte aligned");
      S.part_off = -1;
      u += numel[s] / F;
      if (touched) a.touched[s] = touched[s];
      if (active_rows) {
        // exactness of the skip: a row without gradient since the optimiser was created has m = v = 0, and with g = 0 and no
        // weight decay torch's Adam leaves p unchanged (0 / (0 + eps)); with weight decay every row moves every step
        if (!touched[s]) return set_error(SHINE_E_INVALID, "shine_finish_iteration: active_rows needs flags for every level");
        if (weight_decay[s] != 0.f)
          return set_error(SHINE_E_INVALID, "shine_finish_iteration: active_rows is exact only without weight decay on the feature tables");
      }
      if (lambda_forget != 0.f) {
        if (!feats_last[s] || !importance[s] || !touched[s] || (((size_t)feats_last[s] | (size_t)importance[s]) & 15))
          return set_error(SHINE_E_INVALID, "shine_finish_iteration: null or unaligned regulariser tensor");
        a.last[s] = feats_last[s];
        a.imp[s] = importance[s];
        a.grad_on[s] = grad_on ? grad_on[s] : 1;
      }
      if (s == L - 1) a.feat_units = u;
    } else {
      if (numel[s] != dec_n[s - L]) return set_error(SHINE_E_INVALID, "shine_finish_iteration: decoder tensors are W1, b1, W2, b2, w3, b3");
      S.part_off = dec_off[s - L];
      u += (numel[s] + F - 1) / F;
    }
  }
  a.dec_units = L + (u - a.feat_units);  // the L trash rows first, then the decoder
  V2Geometry g = v3_geometry(n, cfg->kernel_variant >> 8);
  if ((cfg->kernel_variant >> 8) & 64) g.blocks = 1;  // the deterministic single-workgroup launch
  a.partials = (const float*)workspace;
  a.nblocks = (int)g.blocks;
  a.n_surf = cfg->eikonal_on ? reinterpret_cast<const long long*>(n_surf) : nullptr;
  a.n_surf_parts = cfg->n_surf_parts > 1 ? cfg->n_surf_parts : 1;
  if (a.n_surf_parts > 64) return set_error(SHINE_E_INVALID, "shine_finish_iteration: at most 64 n_surf parts");
  a.reduction_sum = cfg->reduction_sum;
  a.inv_n = (float)cfg->inv_n;
  a.weight_e = cfg->weight_e;
  a.loss_parts = loss_parts;
  a.lambda = lambda_forget;
  a.reg_out = reg_out;
  a.active = active_rows ? 1 : 0;
  a.b1 = beta1;
  a.b2 = beta2;
  a.eps = eps;
  a.step_state = reinterpret_cast<const long long*>(step_state);
  a.lr_dev = lr_dev;
  if (next_draw) {
    const long long nb = (next_draw->n + 1 + SB - 1) / SB;
    if (next_draw->n < 1 || nb > FUSED_MAX_BLOCKS || next_draw->pool_size < 1 || next_draw->pool_size > 0x7fffffffll ||
        !next_draw->stream_state || !next_draw->idx_out || (next_draw->surf_parts && !next_draw->surf_bits))
      return set_error(SHINE_E_INVALID, "shine_finish_iteration: next_draw wants 1 <= n < 16 K draws, a device stream state and "
                                        "an index buffer");
    a.nd_blocks = (int)nb;
    a.nd_n = next_draw->n;
    a.nd_pool = next_draw->pool_size;
    a.nd_seed = next_draw->seed;
    a.nd_stream = reinterpret_cast<unsigned long long*>(next_draw->stream_state);
    a.nd_idx = next_draw->idx_out;
    a.nd_bits = next_draw->surf_bits;
    a.nd_surf = reinterpret_cast<long long*>(next_draw->surf_parts);
  }
  long long fb = (a.feat_units + 255) / 256;
  if (fb > 4096) fb = 4096;
  const long long db = (a.dec_units + 3) / 4;
  out->fn = (const void*)k_finish;
  out->grid = dim3((unsigned)(fb + db + 1 + a.nd_blocks));
  out->block = dim3(256);
  out->fb = (int)fb;
  out->db = (int)db;
  return SHINE_OK;
}

}  // namespace shine

extern "C" int shine_finish_iteration(const shine_step_config* cfg, int64_t n, const void* workspace, const int64_t* n_surf,
                                      double* loss_parts, const float* const* feats_last, const float* const* importance,
                                      unsigned char* const* touched, const int32_t* grad_on, float lambda_forget,
                                      double* reg_out, int32_t n_tensors, float* const* params, float* const* grads,
                                      float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                      const float* lr_dev, const int32_t* lr_index, const float* weight_decay, float beta1,
                                      float beta2, float eps, const int64_t* step_state, const shine_next_draw* next_draw,
                                      int32_t active_rows, void* stream) {
  FinLaunch fl;
  int rc = prepare_finish(&fl, cfg, n, workspace, n_surf, loss_parts, feats_last, importance, touched, grad_on, lambda_forget,
                          reg_out, n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr_dev, lr_index, weight_decay, beta1,
                          beta2, eps, step_state, next_draw, active_rows);
  if (rc != SHINE_OK) return rc;
  void* kp[] = {&fl.a, &fl.fb, &fl.db};
  SHINE_HIP_CHECK(hipLaunchKernel(fl.fn, fl.grid, fl.block, kp, 0, (hipStream_t)stream));
  return SHINE_OK;
}
