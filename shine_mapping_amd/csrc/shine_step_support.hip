// shine_step_support.hip — what surrounds the fused-step kernel (shine_step_v3.hip) in a step: the second stage that adds up
// the per-workgroup partial vectors, the touched-row marking pass, workspace sizing / launch facts for bench.py, and the
// MFMA lane-map self-test the GPU suite pins the operand layouts with.
#include "shine_sampler_dev.hpp"
#include "shine_tile16.hpp"

namespace shine {

// rows that receive gradient from this batch (= unique(hierarchical_indices) without -1, feature_octree.py:250):
// one byte flag per row for shine_regularize.  Kept out of the fused kernel's hot loop (only config 4 needs it).
// The decoder's MFMA operand image in global memory (what decoder_operands_issue / _store build in LDS): built here once per
// shine_iter_graph_launch, kept current by k_finish afterwards (FinArgs::op_image), copied by the small-batch builds of the step.
__global__ __launch_bounds__(256) void k_operand_image(V1Args a, float* image) {
  OperandRegs<256> R;
  decoder_operands_issue<256>(a, R, (int)threadIdx.x);
  decoder_operands_store<256>(R, image, image + V3_OPTOTAL, (int)threadIdx.x);
}

int launch_operand_image(const V1Args& a, float* image, hipStream_t st) {
  hipLaunchKernelGGL(k_operand_image, dim3(1), dim3(256), 0, st, a, image);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

__global__ __launch_bounds__(256) void k_mark_touched(V1Args a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int L = a.n_levels;
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  const long long p = a.perm ? (long long)a.perm[i] : i;
  // slots are in VISITING order for a planned batch, but indexed by pool sample id in pool mode (like the main kernel)
  const long long si = a.pool_mode ? p : i;
  if (!a.slots && a.pool_mode != 2) {
    x0 = a.coord[3 * p];
    x1 = a.coord[3 * p + 1];
    x2 = a.coord[3 * p + 2];
  }
  for (int s = 0; s < L; ++s) {
    if (!a.touched[s]) continue;
    int sl;
    if (a.pool_mode == 2) {  // a pool of 32-byte records (shine_step_body.hpp RecLayout): slots at dword 4 (L = 4) / 5
      sl = reinterpret_cast<const int*>(a.coord)[8 * p + (L == 4 ? 4 : 5) + s];
    } else if (a.slots) {
      sl = a.slots[si * L + s];
    } else {
      LevelDev Lv = {};
      Lv.keys = a.lv[s].keys;
      Lv.shift = a.lv[s].shift;
      Lv.mask = a.lv[s].mask;
      const float res = a.lv[s].res;
      sl = probe(Lv, morton3(quantize(x0, res), quantize(x1, res), quantize(x2, res)));
    }
    if (sl < 0) continue;
    const int4 v0 = a.lv[s].vals[2 * sl], v1 = a.lv[s].vals[2 * sl + 1];
    unsigned char* t = a.touched[s];
    t[v0.x] = 1; t[v0.y] = 1; t[v0.z] = 1; t[v0.w] = 1;
    t[v1.x] = 1; t[v1.y] = 1; t[v1.z] = 1; t[v1.w] = 1;
  }
}

// second stage: add the per-workgroup partial vectors into the gradient tensors / loss.
// One 1024-thread block per 64 entries: lane = entry (coalesced 256-B rows), the 16 waves split the blocks.
__global__ __launch_bounds__(1024) void k_reduce_partials(V1Args a, int nblocks, Pass1Args p1) {
  __shared__ float s_red[16][64];
  __shared__ double s_dred[16][3];
  constexpr int RB = (PART_FLOATS + 63) / 64;  // blocks of the reduction proper
  if ((int)blockIdx.x >= RB) {
    // pass 1 of the next sorted draw (k_sample_pass1's arithmetic and summation order: the draw is bit-identical to the
    // stand-alone form): each 256-thread quarter of this block is one sampler block of 1024 draws
    __shared__ double s_p1[16];
    const unsigned long long stream = p1.stream_dev[0];
    const int vb = ((int)blockIdx.x - RB) * 4 + (int)(threadIdx.x >> 8), t256 = threadIdx.x & 255;
    if (vb == 0 && t256 == 0) p1.stream_dev[2] = stream + 1ull;
    if (vb == 0 && p1.surf_parts && t256 < SURF_PARTS) p1.surf_parts[t256] = 0;
    const long long k0 = (long long)vb * SB + t256 * 4;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (vb < p1.nblocks && k0 + j < p1.n1) v += exp1v(p1.seed, stream, (unsigned long long)(k0 + j));
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) s_p1[threadIdx.x >> 6] = v;
    __syncthreads();
    if (t256 == 0 && vb < p1.nblocks) {
      const int w = (int)(threadIdx.x >> 6);
      p1.block_sum[vb] = s_p1[w] + s_p1[w + 1] + s_p1[w + 2] + s_p1[w + 3];
    }
    return;
  }
  __shared__ long long s_ns;  // the batch's surface count (eikonal)
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int L = a.n_levels;
  // where this entry's sum goes; its current value is requested NOW, together with the partial sums (the read-modify-write
  // at the end was a second dependent round trip of this ~7 us kernel)
  float* dst = nullptr;
  if (part == 0 && idx < PART_FLOATS) {
    if (idx < SHINE_MLP_PARAMS) {
      if (a.decoder_grad_on) {
        if (idx < MLP_B1) dst = a.grad_mlp[0] + idx;
        else if (idx < MLP_W2) dst = a.grad_mlp[1] + (idx - MLP_B1);
        else if (idx < MLP_B2) dst = a.grad_mlp[2] + (idx - MLP_W2);
        else if (idx < MLP_W3) dst = a.grad_mlp[3] + (idx - MLP_B2);
        else if (idx < MLP_B3) dst = a.grad_mlp[4] + (idx - MLP_W3);
        else dst = a.grad_mlp[5];
      }
    } else {
      const int t = idx - PART_TRASH, sl = t >> 3, q = t & 7;
      if (sl < L && a.lv[sl].grad) dst = a.lv[sl].grad + a.rows[sl] * F + q;
    }
  }
  const float old = dst ? *dst : 0.f;
  float s = 0.f;
  if (idx < PART_FLOATS) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = part;
    if (nblocks == 256) {  // the full-chip launch: all 16 loads of this thread in flight at once (one round trip)
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = a.partials[(long long)(part + 16 * k) * PART_STRIDE + idx];
#pragma unroll
      for (int k = 0; k < 16; k += 4) {  // same association as the generic loop below
        s0 += v[k];
        s1 += v[k + 1];
        s2 += v[k + 2];
        s3 += v[k + 3];
      }
      b = nblocks;
    }
    for (; b + 48 < nblocks; b += 64) {
      s0 += a.partials[(long long)b * PART_STRIDE + idx];
      s1 += a.partials[(long long)(b + 16) * PART_STRIDE + idx];
      s2 += a.partials[(long long)(b + 32) * PART_STRIDE + idx];
      s3 += a.partials[(long long)(b + 48) * PART_STRIDE + idx];
    }
    for (; b < nblocks; b += 16) s0 += a.partials[(long long)b * PART_STRIDE + idx];
    s = (s0 + s1) + (s2 + s3);
  }
  s_red[part][lane] = s;
  if (blockIdx.x == 0 && lane < 3) {  // loss / count / eikonal doubles ride along in block 0
    double d = 0.0;
    int b = part;
    if (nblocks == 256) {  // 16 independent loads instead of 16 dependent round trips (this was the kernel's tail)
      double v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        v[k] = reinterpret_cast<const double*>(a.partials + (long long)(part + 16 * k) * PART_STRIDE + PART_LOSS)[lane];
#pragma unroll
      for (int k = 0; k < 16; ++k) d += v[k];
      b = nblocks;
    }
    for (; b < nblocks; b += 16)
      d += reinterpret_cast<const double*>(a.partials + (long long)b * PART_STRIDE + PART_LOSS)[lane];
    s_dred[part][lane] = d;
  }
  // the surface count the fused kernel normalised the eikonal term with: workgroup 0 left it next to its loss terms (the
  // sampler's parts in a.n_surf may already belong to the NEXT batch: the pass-1 rider above clears them)
  if (blockIdx.x == 0 && threadIdx.x == 0)
    s_ns = a.n_surf ? (long long)reinterpret_cast<const double*>(a.partials + PART_LOSS)[3] : 0;
  __syncthreads();
  if (dst) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += s_red[k][lane];
    *dst = old + tot;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.loss_parts) {
    double ls = 0.0, cs = 0.0, es = 0.0;
    for (int k = 0; k < 16; ++k) {
      ls += s_dred[k][0];
      cs += s_dred[k][1];
      es += s_dred[k][2];
    }
    const long long ns = s_ns;
    const double bce = a.reduction_sum ? ls : ls * (double)a.inv_n;
    const double eik = ns > 0 ? es * (double)(1.0f / (float)ns) : 0.0;
    a.loss_parts[0] = bce;
    a.loss_parts[1] = eik;
    a.loss_parts[2] = cs;
    a.loss_parts[3] = bce + (double)a.weight_e * eik;  // total of the fused terms
  }
  // FeatureOctree.set_zero (model/feature_octree.py:78-81): the fused step never reads the trash row (a miss
  // contributes nothing), so re-zeroing it here is equivalent to zeroing it before the query
  if (blockIdx.x == 1 && threadIdx.x < L * F) {
    const int sl = threadIdx.x / F, q = threadIdx.x % F;
    if (a.feat_rw[sl]) a.feat_rw[sl][a.rows[sl] * F + q] = 0.f;
  }
  // iteration hooks (include/shine_hip.h, shine_step_config): scalar housekeeping of the launches that FOLLOW the step rides
  // here — the optimiser's step count + bias corrections, the regulariser's accumulator — one thread each, in a block that
  // has little else to do
  if (blockIdx.x == RB - 1) {
    if (threadIdx.x == 0 && a.adam_state) adam_advance(a.adam_state, a.adam_b1, a.adam_b2);
    if (threadIdx.x == 64 && a.zero_f64) *a.zero_f64 = 0.0;
  }
}

// D[16x16] = A[16x4] . B[4x16] through one v_mfma_f32_16x16x4_f32: pins the operand / accumulator lane maps
__global__ void k_selftest_mfma16(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x;
  f32x4 c = zero4();
  c = mfma16(A[(lane & 15) * 4 + (lane >> 4)], B[(lane >> 4) * 16 + (lane & 15)], c);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

long long* g_prof_buffer = nullptr;

// host: cfg->next_draw -> the pass-1 rider of the reduction launch
int fill_pass1_args(Pass1Args* p1, const shine_step_config* cfg) {
  *p1 = Pass1Args{};
  const shine_next_draw* d = cfg->next_draw;
  if (!d) return SHINE_OK;
  const long long nb = (d->n + 1 + SB - 1) / SB;
  if (d->n < 1 || !d->stream_state || !d->workspace || nb > 0x3fffff)
    return set_error(SHINE_E_INVALID, "shine_train_step: next_draw wants n >= 1, a device stream state and the draw's workspace");
  p1->block_sum = reinterpret_cast<double*>(d->workspace);
  p1->n1 = d->n + 1;
  p1->seed = d->seed;
  p1->stream_dev = reinterpret_cast<unsigned long long*>(d->stream_state);
  p1->surf_parts = reinterpret_cast<long long*>(d->surf_parts);
  p1->nblocks = (int)nb;
  return SHINE_OK;
}

}  // namespace shine

using namespace shine;

// measurement aid (include/shine_hip.h): per-wave phase cycle counters, [waves][8] int64, or NULL to disable
extern "C" void shine_debug_set_profile_buffer(int64_t* p) { g_prof_buffer = reinterpret_cast<long long*>(p); }

extern "C" int shine_selftest_mfma16(const float* a, const float* b, float* d, void* stream) {
  if (!a || !b || !d) return set_error(SHINE_E_INVALID, "shine_selftest_mfma16: null argument");
  hipLaunchKernelGGL(k_selftest_mfma16, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" size_t shine_train_step_workspace_bytes(const shine_step_config* cfg, int64_t n) {
  (void)cfg;
  if (n == 0) return 0;
  // n < 0: an upper bound for ANY batch size (callers that keep one buffer for the life of the process): the step kernels
  // launch at most 512 workgroups (the 4-wave form of shine_step_v3.hip below 2048 tiles)
  if (n < 0) return (size_t)512 * PART_STRIDE * sizeof(float);
  return (size_t)v3_geometry(n).blocks * PART_STRIDE * sizeof(float);
}

// what one launch does, for measurement (bench.py's roofline object): out[0] workgroups, out[1] waves, out[2] points per
// tile, out[3] MFMA FLOP issued per tile (padding included), out[4] LDS bytes per workgroup, out[5] useful decoder FLOP
// per point (3 x 2624 BCE, 6 x 2624 with the eikonal term), out[6] the kernel generation (3)
extern "C" int shine_train_step_info(const shine_step_config* cfg, int64_t n, int64_t* out) {
  if (!cfg || !out) return set_error(SHINE_E_INVALID, "shine_train_step_info: null argument");
  if (cfg->n_levels < 1 || cfg->n_levels > SHINE_MAX_LEVELS)
    return set_error(SHINE_E_INVALID, "shine_train_step_info: n_levels out of range");
  // 16-point tiles, v_mfma_f32_16x16x4_f32 = 2048 FLOP each: 68 MFMAs per tile, 88 with the eikonal chain
  const V2Geometry g = v3_geometry(n > 0 ? n : 1);
  out[0] = g.blocks;
  out[1] = g.waves;
  out[2] = V3_TP;
  out[3] = (cfg->eikonal_on ? 88ll : 68ll) * 2048;
  out[4] = v3_lds_bytes(g.wg_waves);
  out[5] = cfg->eikonal_on ? 6 * 2624 : 3 * 2624;
  out[6] = 3;
  out[7] = 0;
  return SHINE_OK;
}

// rows a set of pool samples touches (the unique() of their hierarchical_indices without -1), marked in byte flags:
// the stand-alone form of the pass shine_train_step runs when it is given `touched` — for data-parallel ranks that mark
// the rows of the GLOBAL draw, not only of their own slice (shine_mapping_amd/dp.py TouchedRowReducer).
extern "C" int shine_mark_touched(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                  const int32_t* perm, const int32_t* slots, int64_t n, const int64_t* rows,
                                  unsigned char* const* touched, void* stream) {
  if (!cfg || !touched || n < 0 || (!slots && !coord)) return set_error(SHINE_E_INVALID, "shine_mark_touched: null argument");
  if (cfg->n_levels > LCAP) return set_error(SHINE_E_INVALID, "shine_mark_touched: more than 4 featured levels");
  if (n == 0) return SHINE_OK;
  V1Args a = {};
  LevelSet ls = {};
  int rc = make_level_set(t, cfg, nullptr, rows, nullptr, &ls);
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < cfg->n_levels; ++s) {
    a.lv[s].keys = ls.lv[s].keys;
    a.lv[s].vals = ls.lv[s].vals;
    a.lv[s].shift = ls.lv[s].shift;
    a.lv[s].mask = ls.lv[s].mask;
    a.lv[s].res = ls.lv[s].res;
    a.touched[s] = touched[s];
  }
  a.coord = coord;
  a.perm = perm;
  a.slots = slots;
  a.n = n;
  a.n_levels = cfg->n_levels;
  a.pool_mode = cfg->sorted_input == 2 ? 1 : cfg->sorted_input == 3 ? 2 : 0;
  if (a.pool_mode && !perm) return set_error(SHINE_E_INVALID, "shine_mark_touched: pool mode needs the sample indices");
  hipLaunchKernelGGL(k_mark_touched, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

