// shine_sweep.hip — cal_feature_importance (utils/incre_learning.py:8-40) as a BATCH of steps per launch.
//
// The reference walks the frame's pool in chunks and, per chunk, runs query + decode + BCE + backward and adds
// |hier_features[i].grad| to importance_weight[i] (:27-40): the gradient is summed over a chunk BEFORE the abs, so the chunks
// cannot share a gradient table.  Rounds 2-3 ran the chunk loop on this side of the ABI as {fused step, epilogue} per chunk:
// 35-70 chunks x 2 small launches, 0.69 ms of a 3.3 ms frame at the incremental configuration, each launch using a quarter of
// the chip.  Here a GROUP of chunks (up to 64) is one launch: workgroup b of the grid works on chunk b / bpc as workgroup b % bpc of that
// chunk's step (the SLICED build of the step body: own sample range, own normaliser — sdf_bce_loss 'mean' divides by the
// chunk's size — own gradient tables and touched-row flags), and ONE epilogue launch per group folds the group's tables into
// importance_weight, visiting only the rows a chunk flagged, and leaves tables and flags zero for the next group.
#include <algorithm>

#include "shine_step_body.hpp"
#include "shine_sweep.hpp"

namespace shine {

// (4-wave workgroups whatever the chunk size: a chunk is the reference's batch size, a few thousand samples)
constexpr int SWEEP_WAVES = 4;
template <int L>
__global__ __launch_bounds__(SWEEP_WAVES * 64, 2) void k_step_sweep(SweepArgs s) {
  constexpr int WAVES = SWEEP_WAVES;
  __shared__ StepShared<WAVES> sm;
  const int c = (int)blockIdx.x / s.bpc, b = (int)blockIdx.x - c * s.bpc;
  const long long first = s.begin[c], n = s.begin[c + 1] - first;
  if (n <= 0) return;  // (the whole workgroup)
  StepSlice sl;
  sl.perm = s.a.perm + first;
  sl.n = n;
  sl.inv_n = s.a.reduction_sum ? 1.0f : (float)(1.0 / (double)n);
  sl.part_bid = (int)blockIdx.x;
#pragma unroll
  for (int l = 0; l < LCAP; ++l) {
    sl.grad[l] = l < L ? s.a.lv[l].grad + (long long)c * s.grad_stride : nullptr;
    sl.touched[l] = l < L ? s.a.touched[l] + (long long)c * s.flag_stride : nullptr;
  }
  step_body<L, WAVES, false, false, false, true, true>(s.a, sm, b, s.bpc, &sl);
}

// importance_weight[s] += sum over the group's chunks of |chunk gradient| on the rows a chunk flagged; tables and flags are
// left zero.  One thread per row: the group's flag bytes of the row are all requested at once (unrolled, branch-free) and
// folded into a 64-bit chunk mask; the flagged chunks' rows are then added in chunk order, as the reference's loop does.
// blockIdx.y = level.
__global__ __launch_bounds__(256) void k_sweep_fold(SweepFoldArgs a) {
  const int s = blockIdx.y;
  const long long rows = a.rows[s];  // without the trash row
  float4* const imp = a.imp[s];
  const unsigned char* const flags = a.flags[s];
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r <= rows; r += (long long)gridDim.x * 256) {
    if (r == rows) {
      // the trash row: importance_weight[i][-1] *= 0 (utils/incre_learning.py:40), and FeatureOctree.set_zero
      // (model/feature_octree.py:78-81) — every chunk's query_feature zeroes the features' trash row, which holds the last Adam
      // step's move after the training iterations (the fused step itself never reads it)
      imp[2 * r] = make_float4(0.f, 0.f, 0.f, 0.f);
      imp[2 * r + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
      a.feat[s][2 * r] = make_float4(0.f, 0.f, 0.f, 0.f);
      a.feat[s][2 * r + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    unsigned long long m = 0ull;
    const unsigned char* fp = flags + r;  // (a running per-lane pointer: 64 scalar base addresses would spill the SGPR file)
#pragma unroll
    for (int c = 0; c < SWEEP_GROUP; ++c) {
      const unsigned char f = c < a.chunks ? *fp : (unsigned char)0;
      fp += a.flag_stride;
      m |= f ? 1ull << c : 0ull;
    }
    if (!m) continue;
    float4 v0 = imp[2 * r], v1 = imp[2 * r + 1];
    while (m) {
      const int c = __builtin_ctzll(m);
      m &= m - 1;
      float4* g = reinterpret_cast<float4*>(a.grad[s] + (long long)c * a.grad_stride) + 2 * r;
      const float4 g0 = g[0], g1 = g[1];
      g[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      g[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      v0.x += fabsf(g0.x), v0.y += fabsf(g0.y), v0.z += fabsf(g0.z), v0.w += fabsf(g0.w);
      v1.x += fabsf(g1.x), v1.y += fabsf(g1.y), v1.z += fabsf(g1.z), v1.w += fabsf(g1.w);
      const_cast<unsigned char*>(flags)[(long long)c * a.flag_stride + r] = 0;
    }
    imp[2 * r] = v0;
    imp[2 * r + 1] = v1;
  }
}

// chunk membership (utils/incre_learning.py:27-31: chunk c = pool[c * interval : (c + 1) * interval : down_rate]) of the sample at
// sorted position j: key = its chunk, or n_chunks for a sample the stride skips; value = j
__global__ __launch_bounds__(256) void k_sweep_keys(const int* __restrict__ perm, long long n, long long interval, int down_rate,
                                                    int n_chunks, unsigned long long* __restrict__ key,
                                                    unsigned long long* __restrict__ val) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const long long p = perm[j];
  const long long c = p / interval, off = p - c * interval;
  key[j] = off % down_rate == 0 ? (unsigned long long)c : (unsigned long long)n_chunks;
  val[j] = (unsigned long long)j;
}

__global__ __launch_bounds__(256) void k_sweep_members(const unsigned long long* __restrict__ val, long long kept,
                                                       int* __restrict__ idx) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k < kept) idx[k] = (int)val[k];
}

// The same partition without a sort of the pool, for chunks that fit a workgroup's LDS (bs <= 16384: every shipped yaml has 4096):
// chunk c's members in ORIGINAL pool order are the kept samples p = c * interval + k * down_rate, so sample p's sorted position j
// goes to slot c * bs + k (k_sweep_place: one scattered store per kept sample) — and a chunk's segment then only has to be sorted
// ascending, which one workgroup does in LDS (bitonic network over the next power of two, padded with INT_MAX; the positions are
// distinct).  Two launches instead of the ~20 of a radix / merge sort of (key, value) pairs of the whole pool.
__global__ __launch_bounds__(256) void k_sweep_place(const int* __restrict__ perm, long long n, long long interval, int down_rate,
                                                     long long bs, int* __restrict__ idx) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const long long p = perm[j];
  const long long c = p / interval, off = p - c * interval;
  if (off % down_rate == 0) idx[c * bs + off / down_rate] = (int)j;
}

__global__ __launch_bounds__(1024) void k_sweep_sort_chunks(int* __restrict__ idx, long long bs, long long kept, int pow2) {
  extern __shared__ int s_v[];
  const long long first = (long long)blockIdx.x * bs;
  const long long cnt = kept - first < bs ? kept - first : bs;
  for (int i = threadIdx.x; i < pow2; i += 1024) s_v[i] = i < cnt ? idx[first + i] : 0x7fffffff;
  __syncthreads();
  for (int k = 2; k <= pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < pow2; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const int a = s_v[i], b = s_v[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            s_v[i] = b;
            s_v[l] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < cnt; i += 1024) idx[first + i] = s_v[i];
}

static const void* sweep_fn(int levels) {
  switch (levels) {
    case 1: return (const void*)k_step_sweep<1>;
    case 2: return (const void*)k_step_sweep<2>;
    case 3: return (const void*)k_step_sweep<3>;
    default: return (const void*)k_step_sweep<4>;
  }
}

// workgroups of one chunk's step: one tile per wave up to 256 workgroups (beyond that the waves walk several tiles)
static V2Geometry sweep_geometry(long long n) {
  V2Geometry g = v3_geometry(n > 0 ? n : 1);
  g.wg_waves = SWEEP_WAVES;
  long long blocks = (g.tiles + SWEEP_WAVES - 1) / SWEEP_WAVES;
  if (blocks > 256) blocks = 256;
  g.blocks = blocks;
  g.waves = blocks * SWEEP_WAVES;
  return g;
}

}  // namespace shine

using namespace shine;

// bytes of the zeroed scratch (private gradient tables + flags for `group` chunks) and of the step workspace
static void sweep_layout(const int64_t* rows, int L, size_t* grad_stride_f, size_t* flag_stride_b) {
  size_t gf = 0, fb = 0;
  for (int s = 0; s < L; ++s) {
    gf += (size_t)(rows[s] + 1) * F;
    fb += ((size_t)(rows[s] + 1) + 15) & ~(size_t)15;
  }
  *grad_stride_f = gf;  // floats per chunk (every level's table is a multiple of 8 floats: 16-byte aligned parts)
  *flag_stride_b = fb;
}

// The chunks of cal_feature_importance as segments of a node-ordered pool: perm[j] = pool index of the sample at sorted position
// j.  idx_out[chunk_begin[c] .. chunk_begin[c + 1]) = the sorted positions of chunk c's members, ascending (= node order inside
// a chunk).  chunk_begin (HOST, n_chunks + 1 entries) follows from the sizes alone.  bs <= 16384: two launches (place + per-chunk
// LDS sort); larger chunks: one stable radix pass over the chunk ids.
extern "C" int shine_importance_chunks(const int32_t* perm, int64_t n, int64_t bs, int32_t down_rate, int32_t* idx_out,
                                       int64_t* chunk_begin, int32_t n_chunks, void* workspace, size_t* workspace_bytes,
                                       void* stream) {
  if (n < 0 || bs < 1 || down_rate < 1 || !workspace_bytes || n_chunks < 0)
    return set_error(SHINE_E_INVALID, "shine_importance_chunks: bad argument");
  const long long interval = (long long)bs * down_rate;
  const long long want_chunks = (n + interval - 1) / interval;
  if (n_chunks != want_chunks)
    return set_error(SHINE_E_INVALID, "shine_importance_chunks: n_chunks must be ceil(n / (bs * down_rate))");
  if (n > 0x7fffffffll) return set_error(SHINE_E_INVALID, "shine_importance_chunks: pool exceeds the int32 index range");
  unsigned end_bit = 1;
  while ((1ll << end_bit) <= (long long)n_chunks) ++end_bit;  // keys 0 .. n_chunks
  const size_t ab = ((size_t)n * 8 + 255) & ~(size_t)255;
  size_t tmp_bytes = 0;
  if (bs > SWEEP_SORT_MAX)
    SHINE_HIP_CHECK(prim_sort_pairs_u64(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, (size_t)(n > 0 ? n : 1), 0u,
                                        end_bit, (hipStream_t)stream));
  const size_t need = bs <= SWEEP_SORT_MAX ? 256 : 4 * ab + tmp_bytes;  // (the two-launch form needs none)
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need || !perm || !idx_out || !chunk_begin)
    return set_error(SHINE_E_INVALID, "shine_importance_chunks: null argument or workspace too small");
  long long kept = 0;
  chunk_begin[0] = 0;
  for (int c = 0; c < n_chunks; ++c) {
    const long long head = (long long)c * interval, tail = std::min(head + interval, (long long)n);
    kept += (tail - head + down_rate - 1) / down_rate;
    chunk_begin[c + 1] = kept;
  }
  if (n == 0) return SHINE_OK;
  hipStream_t st = (hipStream_t)stream;
  if (bs <= SWEEP_SORT_MAX) {  // a chunk fits a workgroup's LDS: place, then sort every chunk's segment where it lies
    int pow2 = 2;
    while (pow2 < bs) pow2 <<= 1;
    hipLaunchKernelGGL(k_sweep_place, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, perm, (long long)n, interval,
                       (int)down_rate, (long long)bs, (int*)idx_out);
    SHINE_HIP_CHECK(hipGetLastError());
    if (bs <= 4096) {  // (every shipped yaml) a block radix sort over the positions' bits: ~10 us against ~48 for the network below
      unsigned pos_bits = 1;
      while (pos_bits < 32 && (1ll << pos_bits) < (long long)n) ++pos_bits;
      SHINE_HIP_CHECK(prim_sort_segments_4k((int*)idx_out, (long long)bs, kept, (int)n_chunks, pos_bits, st));
      return SHINE_OK;
    }
    hipLaunchKernelGGL(k_sweep_sort_chunks, dim3((unsigned)n_chunks), dim3(1024), (size_t)pow2 * sizeof(int), st, (int*)idx_out,
                       (long long)bs, kept, pow2);
    SHINE_HIP_CHECK(hipGetLastError());
    return SHINE_OK;
  }
  char* w = (char*)workspace;
  unsigned long long *k0 = (unsigned long long*)w, *k1 = (unsigned long long*)(w + ab);
  unsigned long long *v0 = (unsigned long long*)(w + 2 * ab), *v1 = (unsigned long long*)(w + 3 * ab);
  hipLaunchKernelGGL(k_sweep_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, perm, (long long)n, interval,
                     (int)down_rate, (int)n_chunks, k0, v0);
  SHINE_HIP_CHECK(hipGetLastError());
  SHINE_HIP_CHECK(prim_sort_pairs_u64(w + 4 * ab, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, end_bit, st));
  hipLaunchKernelGGL(k_sweep_members, dim3((unsigned)((kept + 255) / 256)), dim3(256), 0, st, v1, kept, (int*)idx_out);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_importance_sweep_sizes(int32_t n_levels, const int64_t* rows, int32_t n_chunks, int64_t max_chunk,
                                            size_t budget_bytes, int32_t* group_out, size_t* scratch_bytes,
                                            size_t* workspace_bytes) {
  if (n_levels < 1 || n_levels > LCAP || !rows || n_chunks < 0 || max_chunk < 0 || !group_out || !scratch_bytes ||
      !workspace_bytes)
    return set_error(SHINE_E_INVALID, "shine_importance_sweep_sizes: bad argument");
  size_t gf, fb;
  sweep_layout(rows, n_levels, &gf, &fb);
  const size_t per_chunk = gf * sizeof(float) + fb;
  long long group = n_chunks < SWEEP_GROUP ? n_chunks : SWEEP_GROUP;
  if (budget_bytes && per_chunk && (long long)(budget_bytes / per_chunk) < group) group = (long long)(budget_bytes / per_chunk);
  if (group < 1) group = 1;
  const V2Geometry g = sweep_geometry(max_chunk);
  *group_out = (int32_t)group;
  *scratch_bytes = (size_t)group * per_chunk;
  *workspace_bytes = (size_t)group * (size_t)g.blocks * PART_STRIDE * sizeof(float);
  return SHINE_OK;
}

extern "C" int shine_importance_sweep(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                      const float* sdf_label, const float* weight, const int32_t* idx,
                                      const int32_t* slots, const int64_t* chunk_begin, int32_t n_chunks,
                                      const float* const* feats, const int64_t* rows, const float* const* mlp,
                                      float* const* importance, int32_t group, void* scratch, size_t scratch_bytes,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!t || !cfg || !chunk_begin || n_chunks < 0 || !feats || !rows || !mlp || !importance || !scratch)
    return set_error(SHINE_E_INVALID, "shine_importance_sweep: null argument");
  const int L = cfg->n_levels;
  if (L < 1 || L > LCAP) return set_error(SHINE_E_INVALID, "shine_importance_sweep: bad level count");
  if (cfg->sorted_input != 2 || cfg->eikonal_on || cfg->decoder_grad_on)
    return set_error(SHINE_E_INVALID, "shine_importance_sweep: wants a pool-mode config, BCE only, decoder frozen");
  if (group < 1 || group > SWEEP_GROUP) return set_error(SHINE_E_INVALID, "shine_importance_sweep: 1 <= group <= 64");
  size_t gf, fb;
  sweep_layout(rows, L, &gf, &fb);
  if (scratch_bytes < (size_t)group * (gf * sizeof(float) + fb) || ((size_t)scratch & 15))
    return set_error(SHINE_E_INVALID, "shine_importance_sweep: scratch too small or unaligned (shine_importance_sweep_sizes)");
  int64_t max_chunk = 0;
  for (int c = 0; c < n_chunks; ++c) {
    const int64_t n = chunk_begin[c + 1] - chunk_begin[c];
    if (n < 0) return set_error(SHINE_E_INVALID, "shine_importance_sweep: chunk_begin must be non-decreasing");
    if (n > max_chunk) max_chunk = n;
  }
  if (max_chunk == 0) return SHINE_OK;
  // scratch = [group][levels' gradient tables] then [group][levels' flags]
  float* const grad0 = reinterpret_cast<float*>(scratch);
  unsigned char* const flag0 = reinterpret_cast<unsigned char*>(grad0 + (size_t)group * gf);
  float* grad_l[LCAP] = {};
  unsigned char* flag_l[LCAP] = {};
  SweepFoldArgs fa = {};
  {
    size_t go = 0, fo = 0;
    for (int s = 0; s < L; ++s) {
      if (!importance[s] || !feats[s] || rows[s] < 0 || (((size_t)importance[s] | (size_t)feats[s]) & 15))
        return set_error(SHINE_E_INVALID, "shine_importance_sweep: null or unaligned level tensor");
      grad_l[s] = grad0 + go;
      flag_l[s] = flag0 + fo;
      fa.imp[s] = (float4*)importance[s];
      fa.feat[s] = (float4*)const_cast<float*>(feats[s]);
      fa.grad[s] = grad_l[s];
      fa.flags[s] = flag_l[s];
      fa.rows[s] = rows[s];
      go += (size_t)(rows[s] + 1) * F;
      fo += ((size_t)(rows[s] + 1) + 15) & ~(size_t)15;
    }
  }
  fa.grad_stride = (long long)gf;
  fa.flag_stride = (long long)fb;

  SweepArgs sa = {};
  shine_step_config cc = *cfg;
  cc.n_global = max_chunk;
  cc.defer_reduce = 0;  // (no hooks: nothing follows the step but the fold)
  cc.adam_state = nullptr;
  cc.zero_f64 = nullptr;
  cc.next_draw = nullptr;
  int rc = fill_step_args(&sa.a, t, &cc, coord, sdf_label, weight, idx, slots, nullptr, max_chunk, feats, rows, mlp, nullptr,
                          nullptr, grad_l, nullptr, nullptr, flag_l);
  if (rc != SHINE_OK) return rc;
  V2Geometry g = sweep_geometry(max_chunk);
  if (sa.a.ablate & 64) {  // the test suite's deterministic accumulation: one wave per chunk
    g.blocks = 1;
    g.wg_waves = 4;
    g.waves = 4;
  }
  sa.a.tiles = g.tiles;
  sa.a.waves_total = g.waves;
  sa.a.partials = (float*)workspace;
  sa.bpc = (int)g.blocks;
  sa.grad_stride = (long long)gf;
  sa.flag_stride = (long long)fb;
  if (!workspace || workspace_bytes < (size_t)group * (size_t)g.blocks * PART_STRIDE * sizeof(float))
    return set_error(SHINE_E_INVALID, "shine_importance_sweep: workspace too small (shine_importance_sweep_sizes)");
  const void* fn = sweep_fn(L);

  long long max_rows = 0;
  for (int s = 0; s < L; ++s) max_rows = std::max(max_rows, (long long)rows[s] + 1);
  long long fold_blocks = (max_rows + 255) / 256;
  if (fold_blocks > 4096) fold_blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < n_chunks; c0 += group) {
    const int nc = n_chunks - c0 < group ? n_chunks - c0 : group;
    for (int c = 0; c <= nc; ++c) sa.begin[c] = chunk_begin[c0 + c];
    void* kp[] = {&sa};
    SHINE_HIP_CHECK(hipLaunchKernel(fn, dim3((unsigned)(nc * sa.bpc)), dim3((unsigned)(g.wg_waves * 64)), kp, 0, st));
    fa.chunks = nc;
    hipLaunchKernelGGL(k_sweep_fold, dim3((unsigned)fold_blocks, (unsigned)L), dim3(256), 0, st, fa);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}
