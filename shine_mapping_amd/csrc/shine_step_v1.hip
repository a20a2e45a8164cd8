// shine_step_v1.hip — the fused SHINE training step for gfx950, MI355X-first.
//
//   query    FeatureOctree.query_feature   model/feature_octree.py:199-244
//   decode   Decoder.sdf                   model/decoder.py:49-63
//   loss     sdf_bce_loss, eikonal term    utils/loss.py:17-24, shine_batch.py:141-142,182-185 (get_gradient: utils/tools.py:175-185)
//   backward cur_loss.backward()           shine_batch.py:208-209 (closed form, SURVEY.md §8a math contract)
//
// One wave owns a contiguous run of the node-ordered batch and walks it in tiles of 32 points.
//   lane = (pt = lane & 31, h = lane >> 5): the two half-waves hold features 4h..4h+3 of the same 32 points,
//   which is exactly the B-operand / C-accumulator shape of v_mfma_f32_32x32x2_f32 (exact fp32):
//     D[32 channels x 32 points] += A[32 x 2] . B[2 x 32],   lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//     acc reg r of lane l = D[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31].
//   * query: the point's hash slots come with the batch (plan / pool) or are probed here (leaf Morton key once,
//     parents are key >> 3); the corner ids of all levels are loaded together, then per level the 8 16-B row gathers
//     are issued as a batch; a miss reads row 0 with weight 0 (no branches).  The next tile's point data is
//     prefetched one tile ahead (its index two tiles ahead) — AFTER this tile's gathers in the BCE build, because
//     vmcnt counts in issue order and those loads are HBM misses;
//   * decoder forward/backward are chained MFMAs that never leave registers: the k-order of each product is permuted
//     to the accumulator row order of the previous one (k(t,h) = rowidx(t,h)), so the ReLU'd accumulator register t
//     IS the next B operand; the matching A operands are pre-permuted once per workgroup into LDS (s_opA).  In the
//     eikonal build the loss's backward IS the eikonal chain (v1, J) scaled by the point's delta (one decoder output);
//   * decoder weight grads contract over POINTS, so their operands are the transposes: they go through a padded
//     [32][36] LDS tile pair per wave (operands read back as ds_read_b128) and accumulate in MFMA accumulators that
//     live in registers for the whole kernel; BCE build: d2/h1 and d1/f, db1 in a spare MFMA column, db2 as the sum of
//     the transposed operands; eikonal build: v2/(delta h1 + a1) and v1/(delta f + r), one pass per matrix for both terms;
//   * feature grads: lane = (corner c = lane>>3, feature q = lane&7); node-run boundaries / hits of the ordered
//     stream are wave-uniform bit masks, the wave keeps a running sum while the node stays the same and issues
//     ONE 64-lane global_atomic_add_f32 (8 rows x 32 B) per node run; the trash row (index -1, :205,231) gets the plain
//     sum of df over the points that miss a level (its 8 corner weights sum to 1), kept in registers for the whole kernel;
//   * decoder / trash-row / loss sums leave the workgroup as one partial vector in the caller's workspace and a
//     second tiny kernel adds them up (deterministic, no hot-spot atomics), re-zeroes the trash rows
//     (set_zero, :78-81) and finalises the loss.
// Workgroup = 512 threads (8 waves, 2 per SIMD), one per CU: 157 KB of LDS, 18 KB of it private to each wave.
#include "shine_step_common.hpp"

namespace shine {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVES = 8;       // waves per workgroup
constexpr int NT = WAVES * 64;
constexpr int TP = 36;         // transpose tile pitch (floats): 16-B aligned rows for the b128 operand reads
constexpr int WP = 36;         // pitch of the [corner][point] / [feature][point] staging rows (conflict-free b128 reads)
constexpr int U_IDS = LCAP * 8 * WP;             // ids  [LCAP][8 corners][WP]  int32 (point-contiguous)
constexpr int U_W = LCAP * 8 * WP;               // w    [LCAP][8 corners][WP]
constexpr int R2_TL = 0, R2_TR = 32 * TP;        // region 2, first life: the two transpose tiles
constexpr int R2_DF = 0, R2_J = 8 * WP, R2_CQ = 16 * WP;  // second life: df [8][WP], J [8][WP], cq [LCAP][8][WP]
constexpr int R2_FLOATS = 2 * 32 * TP;           // 2304
constexpr int WAVE_FLOATS = U_IDS + U_W + R2_FLOATS;  // 4608 floats = 18,432 B per wave
constexpr int OP_A1 = 0, OP_A2 = 4 * 64, OP_A2T = 20 * 64, OP_A1T = 36 * 64, OP_TOTAL = 52 * 64;
constexpr int SB_B1 = 0, SB_B2 = 32, SB_W3 = 64, SB_B3 = 96;
static_assert(16 * WP + LCAP * 8 * WP <= R2_FLOATS, "df/J/cq must fit the transpose region");
static_assert((PART_LOSS * 4) % 8 == 0 && PART_LOSS >= PART_FLOATS && PART_LOSS + 6 <= PART_STRIDE, "loss slot");
static_assert(PART_STRIDE <= WAVE_FLOATS, "each wave's partial vector aliases its staging region at the end");

__device__ __forceinline__ int rowidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// one transposed weight-grad pass: acc[i][j] += sum_k L[i][k] * R[j][k] over the tile's 32 points.
// TL/TR hold the operands as [channel][point]; RCOLS < 32 zero-fills the unused B columns, ONES makes column
// RCOLS all ones (so acc[:, RCOLS] = sum_k L[i][k]); lsum, if given, accumulates this lane's L operands.
// EXTRA: what MFMA column RCOLS carries — 0 nothing, 1 all ones (acc[:, RCOLS] = sum_k L[i][k]), 2 row RCOLS of TR
// (acc[:, RCOLS] = sum_k L[i][k] TR[RCOLS][k]).  WSUM: *lsum accumulates sum_k L[i][k] * d[k] with d[0..31] read from
// the pad columns of TL's first 8 rows (d[k] at TL[(k >> 2) * TP + 32 + (k & 3)]), else the plain sum of L.
template <int RCOLS, int EXTRA, bool WSUM>
__device__ __forceinline__ f32x16 wgrad_pass(const float* TL, const float* TR, int pt, int h, f32x16 acc, float* lsum) {
  // MFMA t contracts over the two points k = 16 h + t (h = lane >> 5): any bijection onto the tile's 32 points will do,
  // and this one makes a lane's 16 operands contiguous: four ds_read_b128 per operand instead of sixteen ds_read_b32.
  const float4* pl = reinterpret_cast<const float4*>(TL + pt * TP + 16 * h);
  const float4* pr = reinterpret_cast<const float4*>(TR + pt * TP + 16 * h);
  float lv[16], bv[16], dv[WSUM ? 16 : 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 x = pl[j];
    lv[4 * j] = x.x, lv[4 * j + 1] = x.y, lv[4 * j + 2] = x.z, lv[4 * j + 3] = x.w;
    if (RCOLS == 32 || pt < RCOLS || (EXTRA == 2 && pt == RCOLS)) {
      const float4 y = pr[j];
      bv[4 * j] = y.x, bv[4 * j + 1] = y.y, bv[4 * j + 2] = y.z, bv[4 * j + 3] = y.w;
    } else {
      const float c = (EXTRA == 1 && pt == RCOLS) ? 1.f : 0.f;
      bv[4 * j] = bv[4 * j + 1] = bv[4 * j + 2] = bv[4 * j + 3] = c;
    }
    if (WSUM) {
      const float4 d = *reinterpret_cast<const float4*>(TL + (4 * h + j) * TP + 32);
      dv[4 * j] = d.x, dv[4 * j + 1] = d.y, dv[4 * j + 2] = d.z, dv[4 * j + 3] = d.w;
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if (lsum) *lsum = WSUM ? fmaf(lv[t], dv[t], *lsum) : *lsum + lv[t];
    acc = mfma32(lv[t], bv[t], acc);
  }
  return acc;
}

// software prefetch of the next tile's {index -> coord, label, weight, slots}; the index itself comes two tiles ahead.
// Streaming reads of the pool are marked non-temporal: they should not evict table rows from L2 (-1.2 % on the
// eikonal / KITTI-like workload, neutral on the MaiCity-like one).
#define SHINE_PREFETCH_NEXT_TILE() \
    { \
      const long long ni = base + 32 + pt; \
      nvalid = ni < end; \
      np = 0; \
      nx0 = nx1 = nx2 = nlabel = nweight = 0.f; \
      _Pragma("unroll") for (int s = 0; s < LCAP; ++s) nslot[s] = -1; \
      if (nvalid) { \
        np = a.perm ? (long long)np2 : ni; \
        if (a.slots) { \
          const long long si = a.pool_mode ? np : ni; \
          _Pragma("unroll") for (int s = 0; s < L; ++s) nslot[s] = __builtin_nontemporal_load(a.slots + si * L + s); \
        } \
        nx0 = __builtin_nontemporal_load(a.coord + 3 * np); \
        nx1 = __builtin_nontemporal_load(a.coord + 3 * np + 1); \
        nx2 = __builtin_nontemporal_load(a.coord + 3 * np + 2); \
        nlabel = __builtin_nontemporal_load(a.label + np); \
        if (EIK) nweight = a.weight[np]; \
      } \
      if (a.perm && ni + 32 < end) np2 = __builtin_nontemporal_load(a.perm + ni + 32); \
    }

// the 8 row gathers of one level for lane (pt, h): 16 B (features 4h..4h+3) of each corner row; a miss reads row 0
__device__ __forceinline__ void issue_row_gathers(const float* feat, const int4& a0, const int4& a1, bool hit, int h,
                                                  float4 (&dst)[8]) {
  const int ids[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const unsigned int off = (hit ? (unsigned int)ids[c] : 0u) * (unsigned int)F + 4u * (unsigned int)h;
    dst[c] = *reinterpret_cast<const float4*>(feat + off);  // SGPR base + 32-bit lane offset
  }
}

// The `ablate` measurement bits (tools/ablate.py) are read at run time, also by the product instantiation, where the host
// passes 0.  Making them a template parameter was tried in round 2 and REVERTED: with the tests folded away the
// register allocator of ROCm 7.2 lands the same source at 256 VGPRs + 152 B/lane of scratch instead of 238 VGPRs and no
// scratch (BCE, 4 levels: 83 -> 115 us; the eikonal builds went from 20-72 B to 440-716 B of scratch) — the wave-uniform
// branches happen to pin a better schedule.  They cost a handful of scalar compares per tile.
template <int L, bool EIK, bool PROF>
__global__ __launch_bounds__(NT, 2) void k_step_v1(V1Args a) {
  __shared__ float s_opA[OP_TOTAL];
  __shared__ float s_bias[100];
  __shared__ double s_loss[4];
  __shared__ float s_wave[WAVES][WAVE_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pt = lane & 31, h = lane >> 5;
  const bool poly = a.poly != 0;
  const int ablate = a.ablate;
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk = PROF ? clk() : 0;
#define SHINE_STAMP(k)            \
  if (PROF) {                     \
    long long now__ = clk();      \
    pc[k] += now__ - tk;          \
    tk = now__;                   \
  }

  // ---- per-workgroup setup: A operands in MFMA lane order, biases
#pragma unroll
  for (int it = 0; it < (OP_TOTAL + NT - 1) / NT; ++it) {  // branch-free source select: the loads of all passes overlap
    const int idx = it * NT + tid;
    if (idx < OP_TOTAL) {
      const int t = idx >> 6, l = idx & 63, li = l & 31, lh = l >> 5;
      const float* src = t < 4    ? a.mlp[0] + li * F + 4 * lh + t                   // W1[ch=li][k = 4h+t]
                         : t < 20 ? a.mlp[2] + li * H + rowidx((t - 4) & 15, lh)      // W2[out=li][in = rowidx]
                         : t < 36 ? a.mlp[2] + rowidx((t - 20) & 15, lh) * H + li     // W2^T[in=li][out = rowidx]
                                  : a.mlp[0] + rowidx((t - 36) & 15, lh) * F + (li & 7);  // W1^T[feat=li][ch], rows >= 8 zero
      const float v = *src;
      s_opA[idx] = (t >= 36 && li >= F) ? 0.f : v;
    }
  }
  if (tid < 32) {
    s_bias[SB_B1 + tid] = a.mlp[1][tid];
    s_bias[SB_B2 + tid] = a.mlp[3][tid];
    s_bias[SB_W3 + tid] = a.mlp[4][tid];
  }
  if (tid == 0) {
    s_bias[SB_B3] = a.mlp[5][0];
    s_loss[0] = s_loss[1] = s_loss[2] = s_loss[3] = 0.0;
  }
  __syncthreads();

  float* U = s_wave[wv];
  int* U_ids = reinterpret_cast<int*>(U);  // [LCAP][8][WP]
  float* U_w = U + U_IDS;                  // [LCAP][8][WP]
  float* R2 = U + U_IDS + U_W;             // transposes, then df / J / cq
  float* TL = R2 + R2_TL;
  float* TR = R2 + R2_TR;

  const float b3 = s_bias[SB_B3];
  const float sigma = a.sigma;
  float inv_nsurf = 0.f;
  if (EIK) {
    const long long ns = a.n_surf ? *a.n_surf : 0;
    inv_nsurf = ns > 0 ? 1.0f / (float)ns : 0.f;
  }

  f32x16 accW2 = zero16(), accW1 = zero16();
  float dw3c[16];      // sum_p delta_p h2[ch][p] (+ a2) for this lane's point column, reduced over lanes once per wave
  float db2acc = 0.f;  // sum_p d2[ch = lane&31][p] for the points of this half-wave's parity (transposed reads)
#pragma unroll
  for (int r = 0; r < 16; ++r) dw3c[r] = 0.f;
  float db3 = 0.f;
  double loss_acc = 0.0, cnt_acc = 0.0, eik_acc = 0.0;
  int run_id[LCAP];      // float offset (corner row * 8 + feature) of this lane's target in the node run in progress
  int last_slot[LCAP];   // node (hash slot) of the previous point, wave-uniform: carries runs across tiles
  float run_acc[LCAP];
  int run_hit[LCAP];     // wave-uniform: the run in progress belongs to an allocated node (else: a run of misses)
  // trash row (index -1, model/feature_octree.py:78-81,209): all 8 corners of a missed node address it and the corner
  // weights sum to one, so its gradient is the plain sum of df over the points that miss this level.  Lane
  // (sc, sq) keeps the sum for level sc, feature sq; lanes with sc >= L idle.
  float trash_sum = 0.f;
#pragma unroll
  for (int s = 0; s < LCAP; ++s) {
    run_id[s] = -1;
    last_slot[s] = -2;
    run_acc[s] = 0.f;
    run_hit[s] = 0;
  }
  const int sc = lane >> 3, sq = lane & 7;  // scatter role: corner, feature

  const long long wave_g = (long long)blockIdx.x * WAVES + wv;
  const long long begin = wave_g * a.chunk;
  const long long end = (begin + a.chunk < a.n) ? begin + a.chunk : a.n;

  // software prefetch of the {perm -> coord, label} chain: tile t+1's point is fetched while tile t computes
  long long np = 0;
  float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, nlabel = 0.f, nweight = 0.f;
  int nslot[LCAP];
#pragma unroll
  for (int s = 0; s < LCAP; ++s) nslot[s] = -1;
  bool nvalid = begin + pt < end;
  // the index chain perm -> {coord, label, slots} is two dependent round trips: the index itself is fetched TWO tiles
  // ahead (np2), so that at a tile's top only the second stage for tile t+1 is issued, with its address in hand
  int np2 = 0;
  if (a.perm && begin + 32 + pt < end) np2 = a.perm[begin + 32 + pt];
  if (nvalid) {
    np = a.perm ? (long long)a.perm[begin + pt] : begin + pt;
    if (a.slots) {
      const long long si = a.pool_mode ? np : begin + pt;
#pragma unroll
      for (int s = 0; s < L; ++s) nslot[s] = __builtin_nontemporal_load(a.slots + si * L + s);
    }
    nx0 = a.coord[3 * np];
    nx1 = a.coord[3 * np + 1];
    nx2 = a.coord[3 * np + 2];
    nlabel = a.label[np];
    if (EIK) nweight = a.weight[np];
  }
  SHINE_STAMP(0)  // setup

  for (long long base = begin; base < end; base += 32) {
    // the phases that ISSUE memory traffic (query, scatter) run at raised wave priority: when both waves of a SIMD are
    // ready, the one that can put loads / atomics in flight goes first.  (A/B: -2 % on the eikonal build, BCE neutral.)
    __builtin_amdgcn_s_setprio(2);
    const bool valid = nvalid;
    const long long p = np;
    const long long po = a.pool_mode ? base + pt : p;  // where this point's outputs go
    const float x0 = nx0, x1 = nx1, x2 = nx2, label = nlabel, wgt = nweight;
    int pslot[LCAP];
#pragma unroll
    for (int s = 0; s < LCAP; ++s) pslot[s] = nslot[s];
    if (EIK) SHINE_PREFETCH_NEXT_TILE()
    // ================================================================ phase 1: query (all levels)
    float f4[4] = {0.f, 0.f, 0.f, 0.f};
    float A4[4][3];  // EIK: d f_{4h+q} / d x_a
#pragma unroll
    for (int q = 0; q < 4; ++q) A4[q][0] = A4[q][1] = A4[q][2] = 0.f;
    unsigned int chgmask[LCAP], hitmask[LCAP];
    const unsigned int validmask = (unsigned int)__ballot(valid);
#pragma unroll
    for (int s = 0; s < LCAP; ++s) {
      chgmask[s] = 0;
      hitmask[s] = 0;
    }
    // leaf-level key once; a parent's Morton code is the child's >> 3 (quantisation is exact power-of-two scaling)
    int slot[LCAP];
    if (a.slots) {  // planned batch: the slots were found by shine_plan_batch (wave-uniform branch)
#pragma unroll
      for (int s = 0; s < L; ++s) slot[s] = (valid && !(ablate & 16)) ? pslot[s] : -1;
    } else {
    unsigned long long key[LCAP];
    unsigned int slot0[LCAP];
    unsigned long long k0[LCAP];
    {
      const float rl = a.lv[L - 1].res;
      const unsigned long long kleaf = morton3(quantize(x0, rl), quantize(x1, rl), quantize(x2, rl));
#pragma unroll
      for (int s = 0; s < L; ++s) {
        key[s] = kleaf >> (3 * (L - 1 - s));
        slot0[s] = hash_slot(key[s], a.lv[s].shift);
        k0[s] = a.lv[s].keys[slot0[s]];
      }
    }
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const V1Level& Lv = a.lv[s];
      int sl = -1;
      if (k0[s] == key[s]) {
        sl = (int)slot0[s];
      } else if (k0[s] != EMPTY_KEY) {  // collision on the first slot: rare, walk the probe sequence
        unsigned int q = (slot0[s] + 1) & Lv.mask;
        for (unsigned int n = 0; n < Lv.mask; ++n) {
          const unsigned long long kk = Lv.keys[q];
          if (kk == key[s]) {
            sl = (int)q;
            break;
          }
          if (kk == EMPTY_KEY) break;
          q = (q + 1) & Lv.mask;
        }
      }
      slot[s] = (valid && !(ablate & 16)) ? sl : -1;
    }
    }
    int4 i0[LCAP], i1[LCAP];
#pragma unroll
    for (int s = 0; s < L; ++s) {  // the corner ids of every level in flight together
      const V1Level& Lv = a.lv[s];
      const unsigned int sl = slot[s] >= 0 ? (unsigned int)slot[s] : 0u;
      i0[s] = Lv.vals[2u * sl];
      i1[s] = Lv.vals[2u * sl + 1u];
    }
    {
      // All 8 gathers of a level are issued before the first is consumed (issue_row_gathers).  Issuing level s+1's
      // as well before level s is consumed (a second row buffer) spills: +6 % time on the BCE build (A/B, tools/ab_build.py).
      float4 rowbuf[8];
      const bool gather_on = !(ablate & 8);
#pragma unroll
      for (int s = 0; s < L; ++s) {
        const V1Level& Lv = a.lv[s];
        const bool hit = slot[s] >= 0;
        if (!EIK && gather_on) issue_row_gathers(Lv.feat, i0[s], i1[s], hit, h, rowbuf);  // the ALU below runs under them
        // node-run boundaries of the sorted stream (wave-uniform bit masks over the 32 points of the tile)
        int prev = __shfl_up(slot[s], 1, 64);
        if (pt == 0) prev = last_slot[s];
        chgmask[s] = (unsigned int)__ballot(valid && slot[s] != prev);
        hitmask[s] = (unsigned int)__ballot(hit);
        last_slot[s] = __builtin_amdgcn_readlane(slot[s], 31);
        Axis X = axis_weight_rt(poly, x0, Lv.res), Y = axis_weight_rt(poly, x1, Lv.res),
             Z = axis_weight_rt(poly, x2, Lv.res);
        float w[8];
        corner_weights(X.t, Y.t, Z.t, w);
        if (!valid || !hit) {
#pragma unroll
          for (int c = 0; c < 8; ++c) w[c] = 0.f;  // padding lanes and misses contribute nothing to f or to the scatter
        }
        // staging for the scatter (h = 0 writes corners 0-3, h = 1 corners 4-7), point-contiguous rows
        {
          const int m = hit ? 0 : -1;  // a miss stages -1 (trash row), never the speculative ids
          U_ids[(s * 8 + 4 * h + 0) * WP + pt] = (h == 0 ? i0[s].x : i1[s].x) | m;
          U_ids[(s * 8 + 4 * h + 1) * WP + pt] = (h == 0 ? i0[s].y : i1[s].y) | m;
          U_ids[(s * 8 + 4 * h + 2) * WP + pt] = (h == 0 ? i0[s].z : i1[s].z) | m;
          U_ids[(s * 8 + 4 * h + 3) * WP + pt] = (h == 0 ? i0[s].w : i1[s].w) | m;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) U_w[(s * 8 + 4 * h + c) * WP + pt] = h == 0 ? w[c] : w[4 + c];
        if (gather_on) {
          if (EIK) issue_row_gathers(Lv.feat, i0[s], i1[s], hit, h, rowbuf);
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float wz = w[c];
            const float4 r = rowbuf[c];
            f4[0] = fmaf(wz, r.x, f4[0]);
            f4[1] = fmaf(wz, r.y, f4[1]);
            f4[2] = fmaf(wz, r.z, f4[2]);
            f4[3] = fmaf(wz, r.w, f4[3]);
            if (EIK) {
              const float rr[4] = {r.x, r.y, r.z, r.w};
              float dwc[3];
              corner_dw(X, Y, Z, c, dwc);
#pragma unroll
              for (int e = 0; e < 3; ++e) {
                const float dz = hit ? dwc[e] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) A4[q][e] = fmaf(dz, rr[q], A4[q][e]);
              }
            }
          }
        }
      }
    }
    // Prefetch of tile t+1's point data (BCE build).  Issued HERE, after every gather of this tile has been issued: vmcnt
    // counts in order, so loads put in flight at the tile's top have to return before the first corner ids can be
    // consumed — and these are random reads into the pool (HBM misses).  Their data is not needed before the next
    // tile's top.  -8 % kernel time; the eikonal build is 1 % faster with the prefetch at the top (A/B, both ways).
    if (!EIK) SHINE_PREFETCH_NEXT_TILE()
    __builtin_amdgcn_s_setprio(0);
    SHINE_STAMP(1)  // query

    float df4[4], J4[4] = {0.f, 0.f, 0.f, 0.f}, qv[3] = {0.f, 0.f, 0.f};
    if (!EIK) {
    // ---- BCE build: activations stay in registers until the transposed passes (fastest schedule measured)
    // ================================================================ phase 2: decoder forward (MFMA chain)
    f32x16 c1, c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c1[r] = s_bias[SB_B1 + rowidx(r, h)];
      c2[r] = s_bias[SB_B2 + rowidx(r, h)];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) c1 = mfma32(s_opA[OP_A1 + t * 64 + lane], f4[t], c1);
    float h1[16], h2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = fmaxf(c1[r], 0.f);
#pragma unroll
    for (int t = 0; t < 16; ++t) c2 = mfma32(s_opA[OP_A2 + t * 64 + lane], h1[t], c2);
    float yp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h2[r] = fmaxf(c2[r], 0.f);
      yp = fmaf(s_bias[SB_W3 + rowidx(r, h)], h2[r], yp);
    }
    const float y = yp + __shfl_xor(yp, 32, 64) + b3;
    if (valid && h == 0 && a.pred) __builtin_nontemporal_store(y, a.pred + po);

    SHINE_STAMP(2)  // decoder forward

    // ================================================================ phase 3: loss
    float delta = 0.f;
    if (valid) {
      const float zt = sigmoidf_acc(label / sigma);
      if (h == 0) {
        loss_acc += (double)(fmaxf(y, 0.f) - y * zt + log1pf(expf(-fabsf(y))));
        cnt_acc += 1.0;
      }
      delta = (sigmoidf_acc(y) - zt) * a.inv_n;
    }

    // ================================================================ phase 4: backward through the decoder
    float d2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      d2[r] = h2[r] > 0.f ? delta * s_bias[SB_W3 + rowidx(r, h)] : 0.f;
      dw3c[r] = fmaf(delta, h2[r], dw3c[r]);
    }
    if (h == 0) db3 += delta;
    f32x16 e1 = zero16(), e0 = zero16();
#pragma unroll
    for (int t = 0; t < 16; ++t) e1 = mfma32(s_opA[OP_A2T + t * 64 + lane], d2[t], e1);
    float d1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[r] = h1[r] > 0.f ? e1[r] : 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) e0 = mfma32(s_opA[OP_A1T + t * 64 + lane], d1[t], e0);
    // rows 0..7 of e0 = d loss / d f ; lane (pt,h) holds rows 4h..4h+3 in regs 0..3
#pragma unroll
    for (int q = 0; q < 4; ++q) df4[q] = e0[q];
    SHINE_STAMP(3)  // loss + decoder backward

    // ================================================================ phase 5: decoder weight grads (transposed MFMA)
    if (a.decoder_grad_on && !(ablate & 2)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        TL[rowidx(r, h) * TP + pt] = d2[r];
        TR[rowidx(r, h) * TP + pt] = h1[r];
      }
      wave_lds_fence();
      accW2 = wgrad_pass<32, 0, false>(TL, TR, pt, h, accW2, &db2acc);  // dW2[out][in] += d2[out][k] h1[in][k]; db2 rides
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 16; ++r) TL[rowidx(r, h) * TP + pt] = d1[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) TR[(4 * h + q) * TP + pt] = f4[q];
      wave_lds_fence();
      // B columns 0..7 = f, column 8 = ones: accW1[:,8] accumulates db1 = sum_k d1[ch][k] in the spare MFMA lanes
      accW1 = wgrad_pass<F, 1, false>(TL, TR, pt, h, accW1, nullptr);    // dW1[ch][feat] += d1[ch][k] f[feat][k]
      wave_lds_fence();
    }
    } else {
    // ---- eikonal build: same maths, activations retired early (register budget: v1/a1 need the room)
    // ================================================================ phase 2: decoder forward (MFMA chain)
    // Activations are retired as early as possible (register budget): h1 goes to its transpose tile (TR) as soon as
    // layer 2 has consumed it, d2 to TL right after the loss; only the ReLU masks stay, as 16-bit lane masks.
    const bool wg = a.decoder_grad_on && !(ablate & 2);
    f32x16 c1, c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c1[r] = s_bias[SB_B1 + rowidx(r, h)];
      c2[r] = s_bias[SB_B2 + rowidx(r, h)];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) c1 = mfma32(s_opA[OP_A1 + t * 64 + lane], f4[t], c1);
    unsigned int m1 = 0, m2 = 0;  // bit r: channel rowidx(r,h) of this lane's point is active
    {
      float h1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        h1[r] = fmaxf(c1[r], 0.f);
        m1 |= (h1[r] > 0.f ? 1u : 0u) << r;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) c2 = mfma32(s_opA[OP_A2 + t * 64 + lane], h1[t], c2);
      if (wg) {
#pragma unroll
        for (int r = 0; r < 16; ++r) TR[rowidx(r, h) * TP + pt] = h1[r];
      }
    }
    float h2[16];
    float yp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h2[r] = fmaxf(c2[r], 0.f);
      m2 |= (h2[r] > 0.f ? 1u : 0u) << r;
      yp = fmaf(s_bias[SB_W3 + rowidx(r, h)], h2[r], yp);
    }
    const float y = yp + __shfl_xor(yp, 32, 64) + b3;
    if (valid && h == 0 && a.pred) __builtin_nontemporal_store(y, a.pred + po);
    SHINE_STAMP(2)  // decoder forward

    // ================================================================ phase 3: BCE loss
    float delta = 0.f;
    if (valid) {
      const float zt = sigmoidf_acc(label / sigma);
      if (h == 0) {
        loss_acc += (double)(fmaxf(y, 0.f) - y * zt + log1pf(expf(-fabsf(y))));
        cnt_acc += 1.0;
      }
      delta = (sigmoidf_acc(y) - zt) * a.inv_n;
    }
    if (h == 0) db3 += delta;

    // ================================================================ phase 4: backward through the decoder
    // The decoder has ONE output, so everything the loss sends back is the eikonal chain scaled by the point's delta:
    //   d2 = delta * v2,  d1 = delta * v1,  d loss / d f = delta * J          (v2 = m2 .* w3, v1 = m1 .* W2^T v2, J = W1^T v1)
    // -> no separate backward MFMA chain, and the weight grads of both terms contract in ONE pass per matrix:
    //   dW2 += v2 (x) (delta * h1 + a1),   dW1 += v1 (x) (delta * f + r),   db2 += sum_k delta_k v2,  db1 += sum_k delta_k v1.
#pragma unroll
    for (int r = 0; r < 16; ++r) dw3c[r] = fmaf(delta, h2[r], dw3c[r]);
    SHINE_STAMP(3)  // loss + decoder backward

    // ---------------------------------------------------------------- eikonal term (closed form)
    {
      float v1[16], a1[16], r4[4], g[3];
      {
        f32x16 ev = zero16(), ej = zero16();
#pragma unroll
        for (int t = 0; t < 16; ++t)
          ev = mfma32(s_opA[OP_A2T + t * 64 + lane], ((m2 >> t) & 1u) ? s_bias[SB_W3 + rowidx(t, h)] : 0.f, ev);
#pragma unroll
        for (int r = 0; r < 16; ++r) v1[r] = ((m1 >> r) & 1u) ? ev[r] : 0.f;  // m1 .* (W2^T (m2 .* w3))
#pragma unroll
        for (int t = 0; t < 16; ++t) ej = mfma32(s_opA[OP_A1T + t * 64 + lane], v1[t], ej);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          J4[q] = ej[q];             // d y / d f_{4h+q}
          df4[q] = delta * ej[q];    // d loss_bce / d f_{4h+q}
        }
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sm = fmaf(J4[q], A4[q][e], sm);
        sm += __shfl_xor(sm, 32, 64);
        g[e] = sigma * sm;  // get_gradient(coord, pred) * sigma   (utils/tools.py:175-185, shine_batch.py:141-142)
      }
      if (valid && h == 0 && a.grad_x) {
        a.grad_x[3 * po] = g[0];
        a.grad_x[3 * po + 1] = g[1];
        a.grad_x[3 * po + 2] = g[2];
      }
      if (valid && wgt > 0.f) {  // surface samples only (shine_batch.py:137,183)
        const float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        const float e = 1.0f - gn;
        if (h == 0) eik_acc += (double)(e * e);
        const float coef = gn > 0.f ? (-2.0f * e / gn) * (a.weight_e * inv_nsurf) : 0.f;  // norm's sub-gradient 0 at 0
        qv[0] = coef * g[0];
        qv[1] = coef * g[1];
        qv[2] = coef * g[2];
      }
      {
        f32x16 t1 = zero16(), t2 = zero16();
#pragma unroll
        for (int q = 0; q < 4; ++q) r4[q] = sigma * (A4[q][0] * qv[0] + A4[q][1] * qv[1] + A4[q][2] * qv[2]);
#pragma unroll
        for (int t = 0; t < 4; ++t) t1 = mfma32(s_opA[OP_A1 + t * 64 + lane], r4[t], t1);
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[r] = ((m1 >> r) & 1u) ? t1[r] : 0.f;  // (W1 r) .* m1
#pragma unroll
        for (int t = 0; t < 16; ++t) t2 = mfma32(s_opA[OP_A2 + t * 64 + lane], a1[t], t2);
#pragma unroll
        for (int r = 0; r < 16; ++r) dw3c[r] += ((m2 >> r) & 1u) ? t2[r] : 0.f;  // a2 = (W2 a1) .* m2
      }
      // ============================================================== phase 5: decoder weight grads (transposed MFMA)
      if (wg) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rowidx(r, h);
          const float hv = TR[row * TP + pt];  // h1, parked here by this lane after layer 2
          TR[row * TP + pt] = fmaf(delta, hv, a1[r]);
          TL[row * TP + pt] = ((m2 >> r) & 1u) ? s_bias[SB_W3 + row] : 0.f;  // v2
        }
        if (h == 0) TL[(pt >> 2) * TP + 32 + (pt & 3)] = delta;  // delta_k in the pad columns (wgrad_pass WSUM)
        wave_lds_fence();
        accW2 = wgrad_pass<32, 0, true>(TL, TR, pt, h, accW2, &db2acc);  // dW2 += v2 (x) (delta h1 + a1); db2 rides
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 16; ++r) TL[rowidx(r, h) * TP + pt] = v1[r];
#pragma unroll
        for (int q = 0; q < 4; ++q) TR[(4 * h + q) * TP + pt] = fmaf(delta, f4[q], r4[q]);
        if (h == 0) TR[F * TP + pt] = delta;  // row 8: accW1[:, 8] = sum_k v1[ch][k] delta_k = db1
        wave_lds_fence();
        accW1 = wgrad_pass<F, 2, false>(TL, TR, pt, h, accW1, nullptr);    // dW1 += v1 (x) (delta f + r)
        wave_lds_fence();
      }
    }
    }
    SHINE_STAMP(5)  // weight grads

    // ================================================================ phase 6: feature-grad scatter (run-length)
    // stage df [feature][pt] (and, EIK, J and cq = sigma * (d w_c / d x . q)) in region 2, then
    // lane = (corner sc, feature sq): run boundaries / hits are wave-uniform bit masks (scalar branches only);
    // one 64-lane atomic (8 rows x 32 B) per node run, misses go to a register sum.
    if (!(ablate & 4)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) R2[R2_DF + (4 * h + q) * WP + pt] = df4[q];
      if (EIK) {
#pragma unroll
        for (int q = 0; q < 4; ++q) R2[R2_J + (4 * h + q) * WP + pt] = J4[q];
#pragma unroll
        for (int s = 0; s < L; ++s) {
          const float res = a.lv[s].res;
          Axis X = axis_weight_rt(poly, x0, res), Y = axis_weight_rt(poly, x1, res), Z = axis_weight_rt(poly, x2, res);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float d0[3], d1[3];
            corner_dw(X, Y, Z, c, d0);
            corner_dw(X, Y, Z, 4 + c, d1);
            const float v = h == 0 ? (d0[0] * qv[0] + d0[1] * qv[1] + d0[2] * qv[2])
                                   : (d1[0] * qv[0] + d1[1] * qv[1] + d1[2] * qv[2]);
            // a miss stages 0: its 8 terms would all land on the trash row, where they cancel (sum_c dw_c/dx = 0)
            R2[R2_CQ + (s * 8 + 4 * h + c) * WP + pt] = ((hitmask[s] >> pt) & 1u) ? sigma * v : 0.f;
          }
        }
      }
      wave_lds_fence();
      __builtin_amdgcn_s_setprio(2);
      constexpr int CH = EIK ? 8 : 16;  // points per chunk (register budget)
      // this lane's trash level: the points that miss level sc (valid points only; padding lanes carry df = 0 anyway)
      unsigned int mymiss = 0u;
#pragma unroll
      for (int s = 0; s < L; ++s)
        if (sc == s) mymiss = ~hitmask[s] & validmask;
#pragma unroll 1
      for (int ch = 0; ch < 32 / CH; ++ch) {
        float dfr[CH], jr[EIK ? CH : 1];
#pragma unroll
        for (int j = 0; j < CH / 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(R2 + R2_DF + sq * WP + CH * ch + 4 * j);
          dfr[4 * j] = v.x;
          dfr[4 * j + 1] = v.y;
          dfr[4 * j + 2] = v.z;
          dfr[4 * j + 3] = v.w;
          if (EIK) {
            const float4 u = *reinterpret_cast<const float4*>(R2 + R2_J + sq * WP + CH * ch + 4 * j);
            jr[4 * j] = u.x;
            jr[4 * j + 1] = u.y;
            jr[4 * j + 2] = u.z;
            jr[4 * j + 3] = u.w;
          }
        }
        {  // trash rows: one masked add per point, outside the per-level loop
          const unsigned int mm = mymiss >> (CH * ch);
#pragma unroll
          for (int p2 = 0; p2 < CH; ++p2) {
            // (0 - bit) is all-ones for a miss: AND keeps the float's bits or gives +0.0
            const unsigned int keep = 0u - ((mm >> p2) & 1u);
            trash_sum += __uint_as_float(__float_as_uint(dfr[p2]) & keep);
          }
        }
#pragma unroll
        for (int s = 0; s < L; ++s) {
          float* gbase = a.lv[s].grad;
          if (gbase) {
            float wr[CH], cqr[EIK ? CH : 1];
            int idr[CH];
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) {
              const float4 v = *reinterpret_cast<const float4*>(U_w + (s * 8 + sc) * WP + CH * ch + 4 * j);
              wr[4 * j] = v.x;
              wr[4 * j + 1] = v.y;
              wr[4 * j + 2] = v.z;
              wr[4 * j + 3] = v.w;
              const int4 u = *reinterpret_cast<const int4*>(U_ids + (s * 8 + sc) * WP + CH * ch + 4 * j);
              idr[4 * j] = u.x;
              idr[4 * j + 1] = u.y;
              idr[4 * j + 2] = u.z;
              idr[4 * j + 3] = u.w;
              if (EIK) {
                const float4 c4 = *reinterpret_cast<const float4*>(R2 + R2_CQ + (s * 8 + sc) * WP + CH * ch + 4 * j);
                cqr[4 * j] = c4.x;
                cqr[4 * j + 1] = c4.y;
                cqr[4 * j + 2] = c4.z;
                cqr[4 * j + 3] = c4.w;
              }
            }
            int rid = run_id[s], rhit = run_hit[s];
            float racc = run_acc[s];
            const unsigned int cm = (chgmask[s] & validmask) >> (CH * ch);
            // (debug bit 1, "no atomics", is folded into the hit mask: no test inside the loop)
            const unsigned int hm = (ablate & 1) ? 0u : hitmask[s] >> (CH * ch);
#pragma unroll
            for (int p2 = 0; p2 < CH; ++p2) {
              if (cm & (1u << p2)) {  // a new node (or a run of misses) starts here: close the open run
                if (rhit) atomic_add_f32(gbase + (unsigned int)rid, racc);  // scalar branch (wave-uniform hit mask)
                racc = 0.f;
                rid = (idr[p2] << 3) | sq;  // float offset of this lane's (corner row, feature)
                rhit = (int)((hm >> p2) & 1u);
              }
              // misses and padding lanes staged w = 0 (and cq = 0): they leave racc alone
              racc = fmaf(wr[p2], dfr[p2], racc);
              if (EIK) racc = fmaf(cqr[p2], jr[p2], racc);
            }
            run_id[s] = rid;
            run_hit[s] = rhit;
            run_acc[s] = racc;
          }
        }
      }
      wave_lds_fence();
    }
    __builtin_amdgcn_s_setprio(0);
    SHINE_STAMP(4)  // scatter
  }

  // ---- end of the wave's run: flush the open node runs
#pragma unroll
  for (int s = 0; s < L; ++s) {
    float* gbase = a.lv[s].grad;
    if (gbase && run_hit[s]) atomic_add_f32(gbase + (unsigned int)run_id[s], run_acc[s]);
  }
  __syncthreads();  // every wave is done with its staging region: it now holds the wave's partial vector
  // Each wave writes its sums with plain stores into ITS OWN region (no LDS atomics: 8 waves adding into the
  // same 1377 addresses cost ~40k cycles), then the workgroup adds the 8 vectors in one pass.
  float* wvec = s_wave[wv];
  if (sc < L) wvec[PART_TRASH + sc * 8 + sq] = trash_sum;  // lane (sc, sq) owns level sc, feature sq
  if (a.decoder_grad_on) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rowidx(r, h);
      wvec[MLP_W2 + row * H + pt] = accW2[r];
      if (pt <= F) wvec[pt < F ? MLP_W1 + row * F + pt : MLP_B1 + row] = accW1[r];  // column 8 of accW1 is db1
      float w3v = row16_sum(dw3c[r]);
      w3v += __shfl_xor(w3v, 16, 64);  // the half-wave's second DPP row
      if (pt == 0) wvec[MLP_W3 + row] = w3v;
    }
    const float b2v = db2acc + __shfl_xor(db2acc, 32, 64);  // the two half-waves hold the two point parities
    if (h == 0) wvec[MLP_B2 + pt] = b2v;
    const float b3v = wave_sum(db3);
    if (lane == 0) wvec[MLP_B3] = b3v;
  }
  {
    double ls = wave_sum_d(loss_acc), cs = wave_sum_d(cnt_acc);
    double es = EIK ? wave_sum_d(eik_acc) : 0.0;
    if (lane == 0) {
      atomicAdd(&s_loss[0], ls);
      atomicAdd(&s_loss[1], cs);
      if (EIK) atomicAdd(&s_loss[2], es);
    }
  }
  SHINE_STAMP(6)  // flush
  __syncthreads();
  SHINE_STAMP(7)  // wait for the workgroup
  if (PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a.prof[wave_g * 8 + k] = pc[k];
  }

  const int mlp_lo = a.decoder_grad_on ? 0 : SHINE_MLP_PARAMS;  // a frozen decoder has no sums to move
  if (a.partials) {
    float* dst = a.partials + (long long)blockIdx.x * PART_STRIDE;
    for (int idx = tid; idx < PART_TRASH + L * 8; idx += NT) {
      float v = 0.f;
      if (idx >= mlp_lo) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += s_wave[w][idx];
      }
      dst[idx] = v;
    }
    for (int idx = PART_TRASH + L * 8 + tid; idx < PART_FLOATS; idx += NT) dst[idx] = 0.f;
    if (tid == 0) {
      double* dl = reinterpret_cast<double*>(dst + PART_LOSS);
      dl[0] = s_loss[0];
      dl[1] = s_loss[1];
      dl[2] = s_loss[2];
    }
  } else {
    // no workspace: hot-spot atomics (slower, not deterministic); loss_parts was zeroed by the host side
    for (int idx = mlp_lo + tid; idx < PART_TRASH + L * 8; idx += NT) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += s_wave[w][idx];
      if (v == 0.f) continue;
      float* d;
      if (idx < MLP_B1) d = a.grad_mlp[0] + idx;
      else if (idx < MLP_W2) d = a.grad_mlp[1] + (idx - MLP_B1);
      else if (idx < MLP_B2) d = a.grad_mlp[2] + (idx - MLP_W2);
      else if (idx < MLP_W3) d = a.grad_mlp[3] + (idx - MLP_B2);
      else if (idx < MLP_B3) d = a.grad_mlp[4] + (idx - MLP_W3);
      else if (idx < PART_TRASH) d = a.grad_mlp[5];
      else {
        const int sl = (idx - PART_TRASH) >> 3, q = (idx - PART_TRASH) & 7;
        d = a.lv[sl].grad ? a.lv[sl].grad + a.rows[sl] * F + q : nullptr;
      }
      if (d) atomic_add_f32(d, v);
    }
    if (tid == 0 && a.loss_parts) {
      const double bce = a.reduction_sum ? s_loss[0] : s_loss[0] * (double)a.inv_n;
      const double eik = EIK ? s_loss[2] * (double)inv_nsurf : 0.0;
      atomicAdd(a.loss_parts + 0, bce);
      atomicAdd(a.loss_parts + 1, eik);
      atomicAdd(a.loss_parts + 2, s_loss[1]);
      atomicAdd(a.loss_parts + 3, bce + (double)a.weight_e * eik);
    }
  }
}

// rows that receive gradient from this batch (= unique(hierarchical_indices) without -1, feature_octree.py:250):
// one byte flag per row for shine_regularize.  Kept out of the fused kernel's hot loop (only config 4 needs it).
__global__ __launch_bounds__(256) void k_mark_touched(V1Args a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int L = a.n_levels;
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  const long long p = a.perm ? (long long)a.perm[i] : i;
  // slots are in VISITING order for a planned batch, but indexed by pool sample id in pool mode (like the main kernel)
  const long long si = a.pool_mode ? p : i;
  if (!a.slots) {
    x0 = a.coord[3 * p];
    x1 = a.coord[3 * p + 1];
    x2 = a.coord[3 * p + 2];
  }
  for (int s = 0; s < L; ++s) {
    if (!a.touched[s]) continue;
    int sl;
    if (a.slots) {
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
__global__ __launch_bounds__(1024) void k_reduce_partials(V1Args a, int nblocks) {
  __shared__ float s_red[16][64];
  __shared__ double s_dred[16][3];
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
    const long long ns = a.n_surf ? *a.n_surf : 0;
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
}

// D[32x32] = A[32x2] . B[2x32] through one v_mfma_f32_32x32x2_f32: pins the operand / accumulator lane maps
__global__ void k_selftest_mfma(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x;
  f32x16 c = zero16();
  c = mfma32(A[(lane & 31) * 2 + (lane >> 5)], B[(lane >> 5) * 32 + (lane & 31)], c);
#pragma unroll
  for (int r = 0; r < 16; ++r) D[rowidx(r, lane >> 5) * 32 + (lane & 31)] = c[r];
}

struct V1Geometry {
  long long waves, chunk, blocks;
};
static V1Geometry v1_geometry(long long n) {
  V1Geometry g;
  long long tiles = (n + 31) / 32;
  long long max_waves = 256 * WAVES;  // one resident 8-wave workgroup per CU
  g.waves = tiles < max_waves ? (tiles < 1 ? 1 : tiles) : max_waves;
  long long per = (n + g.waves - 1) / g.waves;
  g.chunk = ((per + 31) / 32) * 32;
  g.waves = (n + g.chunk - 1) / g.chunk;
  if (g.waves < 1) g.waves = 1;
  g.blocks = (g.waves + WAVES - 1) / WAVES;
  return g;
}

template <bool EIK>
static void launch_v1(const V1Args& a0, int levels, dim3 grid, hipStream_t st) {
  V1Args a = a0;
  if (a.prof && levels == 4) {  // debug build of the same kernel with s_memtime stamps per phase
    hipLaunchKernelGGL((k_step_v1<4, EIK, true>), grid, dim3(NT), 0, st, a);
    return;
  }
  a.prof = nullptr;
  switch (levels) {  // the level count is a template parameter: straight-line query code, no guards
    case 1: hipLaunchKernelGGL((k_step_v1<1, EIK, false>), grid, dim3(NT), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_step_v1<2, EIK, false>), grid, dim3(NT), 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_step_v1<3, EIK, false>), grid, dim3(NT), 0, st, a); break;
    default: hipLaunchKernelGGL((k_step_v1<4, EIK, false>), grid, dim3(NT), 0, st, a); break;
  }
}

}  // namespace shine

using namespace shine;

extern "C" size_t shine_train_step_workspace_bytes(const shine_step_config* cfg, int64_t n) {
  (void)cfg;
  if (n == 0) return 0;
  // n < 0: an upper bound for ANY batch size (callers that keep one buffer for the life of the process): the 32-point
  // kernel launches at most 256 workgroups, the 16-point kernel at most 512 (its 4-wave form below 2048 tiles)
  if (n < 0) return (size_t)512 * PART_STRIDE * sizeof(float);
  const long long b1 = v1_geometry(n).blocks, b2 = v2_geometry(n).blocks, b3 = v3_geometry(n).blocks;
  const long long b = b1 > b2 ? (b1 > b3 ? b1 : b3) : (b2 > b3 ? b2 : b3);
  return (size_t)b * PART_STRIDE * sizeof(float);
}

// what one launch does, for measurement (bench.py's roofline object): out[0] workgroups, out[1] waves, out[2] points per
// tile, out[3] MFMA FLOP issued per tile (padding included), out[4] LDS bytes per workgroup, out[5] useful decoder FLOP
// per point (3 x 2624 BCE, 6 x 2624 with the eikonal term)
extern "C" int shine_train_step_info(const shine_step_config* cfg, int64_t n, int64_t* out) {
  if (!cfg || !out) return set_error(SHINE_E_INVALID, "shine_train_step_info: null argument");
  if (cfg->n_levels < 1 || cfg->n_levels > SHINE_MAX_LEVELS)
    return set_error(SHINE_E_INVALID, "shine_train_step_info: n_levels out of range");
  out[5] = cfg->eikonal_on ? 6 * 2624 : 3 * 2624;
  out[6] = out[7] = 0;
  if (v3_serves(cfg, true)) {  // 16-point tiles, v_mfma_f32_16x16x4_f32 = 2048 FLOP each (bench.py's steps are pool
    // batches: the lane = (point, level) kernel, shine_step_v3.hip); 68 MFMAs per tile, 88 with the eikonal chain
    const V2Geometry g2 = v3_geometry(n > 0 ? n : 1);
    out[0] = g2.blocks;
    out[1] = g2.waves;
    out[2] = 16;
    out[3] = (cfg->eikonal_on ? 88ll : 68ll) * 2048;
    out[4] = v3_lds_bytes(g2.wg_waves);
    out[6] = 3;
    return SHINE_OK;
  }
  const V1Geometry g = v1_geometry(n > 0 ? n : 1);
  out[0] = g.blocks;
  out[1] = g.waves;
  out[2] = 32;
  out[3] = (long long)(cfg->eikonal_on ? MFMA_PER_TILE_EIK : MFMA_PER_TILE_BCE) * 4096;  // 32x32x2 MFMA = 4096 FLOP
  out[4] = (long long)(sizeof(float) * (OP_TOTAL + 100 + WAVES * WAVE_FLOATS) + 4 * sizeof(double));
  out[6] = 1;
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
  a.pool_mode = cfg->sorted_input == 2 ? 1 : 0;
  if (a.pool_mode && !perm) return set_error(SHINE_E_INVALID, "shine_mark_touched: pool mode needs the sample indices");
  hipLaunchKernelGGL(k_mark_touched, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

namespace shine {
long long* g_prof_buffer = nullptr;
}
// measurement aid (include/shine_hip.h): per-wave phase cycle counters, [waves][8] int64, or NULL to disable
extern "C" void shine_debug_set_profile_buffer(int64_t* p) { g_prof_buffer = reinterpret_cast<long long*>(p); }

extern "C" int shine_selftest_mfma(const float* a, const float* b, float* d, void* stream) {
  if (!a || !b || !d) return set_error(SHINE_E_INVALID, "shine_selftest_mfma: null argument");
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_train_step_v1(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                   const float* sdf_label, const float* weight, const int32_t* perm,
                                   const int32_t* slots, const int64_t* n_surf, int64_t n, const float* const* feats,
                                   const int64_t* rows, const float* const* mlp, float* pred_out, float* grad_x_out,
                                   float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                   unsigned char* const* touched, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  V1Args a = {};
  int rc = fill_step_args(&a, t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                          grad_x_out, grad_feats, grad_mlp, loss_parts, touched);
  if (rc != SHINE_OK) return rc;
  if (n == 0) return SHINE_OK;
  V1Geometry g = v1_geometry(n);
  a.chunk = g.chunk;
  a.prof = g_prof_buffer;
  const size_t need = (size_t)g.blocks * PART_STRIDE * sizeof(float);
  a.partials = (workspace && workspace_bytes >= need) ? (float*)workspace : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (!a.partials && loss_parts) SHINE_HIP_CHECK(hipMemsetAsync(loss_parts, 0, 4 * sizeof(double), st));
  const dim3 grid((unsigned)g.blocks);
  if (touched) {
    hipLaunchKernelGGL(k_mark_touched, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  if (cfg->eikonal_on)
    launch_v1<true>(a, cfg->n_levels, grid, st);
  else
    launch_v1<false>(a, cfg->n_levels, grid, st);
  SHINE_HIP_CHECK(hipGetLastError());
  if (a.partials && !(a.ablate & 32)) {  // (ablate bit 32: measurement only — time the dominant kernel by itself)
    hipLaunchKernelGGL(k_reduce_partials, dim3((PART_FLOATS + 63) / 64), dim3(1024), 0, st, a, (int)g.blocks);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}
