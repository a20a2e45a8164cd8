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
// cfg->draw_rider: one 1024-thread block = four sampler blocks of 256 threads (quarter q = threadIdx.x >> 8).  The first
// ceil(nblocks / 4) rider blocks run pass 1 of the draw after next, the others pass 2 of the next draw — k_sample_pass1 /
// k_sample_pass2's arithmetic and summation order (shine_sampler.hip), so the draws are bit-identical to the stand-alone
// sampler's; every rider thread also takes its share of the zero-fill.  `rb`: index among the rider blocks.
__device__ __forceinline__ void draw_rider_block(const DrawRiderArgs& dr, int rb) {
  __shared__ double s_red[4][4];
  __shared__ double s_wave_pre[4][4];
  __shared__ int s_cnt[4][4];
  const int qb = (dr.nblocks + 3) / 4;
  const int q = threadIdx.x >> 8, t256 = threadIdx.x & 255, w = t256 >> 6, lane = threadIdx.x & 63;
  const unsigned long long sid = dr.state[dr.parity];  // stream id of the draw THIS step used
  for (long long z = (long long)rb * 1024 + threadIdx.x; z < dr.zero_n16; z += (long long)2 * qb * 1024)
    dr.zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rb < qb) {  // ---- pass 1 of draw sid + 2
    const unsigned long long stream = sid + 2ull;
    const int vb = rb * 4 + q;
    if (rb == 0 && threadIdx.x == 0) dr.state[1 - dr.parity] = sid + 1ull;
    if (rb == 0 && dr.parts_this && threadIdx.x < SURF_PARTS) dr.parts_this[threadIdx.x] = 0;
    const long long k0 = (long long)vb * SB + t256 * 4;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (vb < dr.nblocks && k0 + j < dr.n + 1) v += exp1v(dr.seed, stream, (unsigned long long)(k0 + j));
    v = wave_sum_d(v);
    if (lane == 0) s_red[q][w] = v;
    __syncthreads();
    if (t256 == 0 && vb < dr.nblocks) dr.bs_after[vb] = s_red[q][0] + s_red[q][1] + s_red[q][2] + s_red[q][3];
    return;
  }
  // ---- pass 2 of draw sid + 1
  const unsigned long long stream = sid + 1ull;
  const int vb = (rb - qb) * 4 + q;
  const bool on = vb < dr.nblocks;  // (padding quarters of the last block walk through the barriers only)
  double before = 0.0, total = 0.0;
  for (int b = t256; b < dr.nblocks; b += 256) {
    const double v = dr.bs_next[b];
    total += v;
    if (b < vb) before += v;
  }
  before = wave_sum_d(before);
  if (lane == 0) s_red[q][w] = before;
  __syncthreads();
  before = s_red[q][0] + s_red[q][1] + s_red[q][2] + s_red[q][3];
  __syncthreads();
  total = wave_sum_d(total);
  if (lane == 0) s_red[q][w] = total;
  __syncthreads();
  total = s_red[q][0] + s_red[q][1] + s_red[q][2] + s_red[q][3];
  __syncthreads();
  const long long k0 = (long long)vb * SB + t256 * 4;
  double e[4], run = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    e[j] = (on && k0 + j <= dr.n) ? exp1v(dr.seed, stream, (unsigned long long)(k0 + j)) : 0.0;
    run += e[j];
  }
  double inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_wave_pre[q][w] = inc;
  __syncthreads();
  double wpre = 0.0;
  for (int ww = 0; ww < w; ++ww) wpre += s_wave_pre[q][ww];
  double sacc = before + wpre + (inc - run);
  int surf = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sacc += e[j];
    if (on && k0 + j < dr.n) {
      long long v = (long long)((sacc / total) * (double)dr.pool);
      v = v < 0 ? 0 : (v >= dr.pool ? dr.pool - 1 : v);
      dr.idx[k0 + j] = (int)v;
      if (dr.parts_next) surf += (int)((dr.surf_bits[v >> 5] >> (v & 31)) & 1u);
    }
  }
  if (dr.parts_next) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) surf += __shfl_xor(surf, o, 64);
    if (lane == 0) s_cnt[q][w] = surf;
    __syncthreads();
    if (t256 == 0 && on)
      __hip_atomic_fetch_add(dr.parts_next + (vb & (SURF_PARTS - 1)), (long long)(s_cnt[q][0] + s_cnt[q][1] + s_cnt[q][2] + s_cnt[q][3]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(1024) void k_reduce_partials(V1Args a, int nblocks, Pass1Args p1, DrawRiderArgs dr) {
  __shared__ float s_red[16][64];
  __shared__ double s_dred[16][3];
  constexpr int RB = (PART_FLOATS + 63) / 64;  // blocks of the reduction proper
  if ((int)blockIdx.x >= RB && dr.nblocks > 0) {
    draw_rider_block(dr, (int)blockIdx.x - RB);
    return;
  }
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

// host: cfg->draw_rider -> the whole-draw rider of the reduction launch
int fill_draw_rider_args(DrawRiderArgs* dr, const shine_step_config* cfg) {
  *dr = DrawRiderArgs{};
  const shine_draw_rider* r = cfg->draw_rider;
  if (!r) return SHINE_OK;
  if (cfg->next_draw || cfg->defer_reduce)
    return set_error(SHINE_E_INVALID, "shine_train_step: draw_rider excludes next_draw and defer_reduce");
  const long long nb = (r->n + 1 + SB - 1) / SB;
  if (r->n < 1 || nb > 0x3fffff || r->pool_size < 1 || r->pool_size > 0x7fffffffll || !r->state || !r->idx_out ||
      !r->block_sum[0] || !r->block_sum[1] || (r->parity != 0 && r->parity != 1) ||
      (r->surf_bits && (!r->surf_parts[0] || !r->surf_parts[1])))
    return set_error(SHINE_E_INVALID, "shine_train_step: draw_rider wants n >= 1, a device state, both block-sum buffers, idx_out, "
                                      "parity 0 / 1 and (with surf_bits) both surf_parts");
  if (r->zero_ptr && (((size_t)r->zero_ptr | (size_t)r->zero_bytes) & 15))
    return set_error(SHINE_E_INVALID, "shine_train_step: draw_rider's zero buffer must be 16-byte aligned and sized");
  dr->bs_next = r->block_sum[1 - r->parity];
  dr->bs_after = r->block_sum[r->parity];
  dr->state = reinterpret_cast<unsigned long long*>(r->state);
  dr->parity = r->parity;
  dr->nblocks = (int)nb;
  dr->n = r->n;
  dr->pool = r->pool_size;
  dr->seed = r->seed;
  dr->idx = r->idx_out;
  dr->surf_bits = r->surf_bits;
  dr->parts_this = r->surf_bits ? reinterpret_cast<long long*>(r->surf_parts[r->parity]) : nullptr;
  dr->parts_next = r->surf_bits ? reinterpret_cast<long long*>(r->surf_parts[1 - r->parity]) : nullptr;
  dr->zero_ptr = reinterpret_cast<float4*>(r->zero_ptr);
  dr->zero_n16 = r->zero_ptr ? r->zero_bytes / 16 : 0;
  return SHINE_OK;
}

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

